#!/usr/bin/env python3
"""Per-wavefront phase timeline of the fused kernel from in-kernel s_memtime stamps (trk_debug_set_stamp_buffer)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
from torch_robotics_amd._lib import lib

dev = torch.device("cuda:0")
ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
task.set_ee_target(Ht)
B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4096
H = 64
q = robot.random_q(B * H).reshape(B, H, 7).contiguous()
model, cm = task._fused_handles(dev)
NO_POS = "--no-pos" in sys.argv
W = tuple(float(v) for v in sys.argv[sys.argv.index("--weights") + 1].split(",")) if "--weights" in sys.argv else (0, 1, 0, 1)
TAG = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else "default"
plan = ops.RolloutPlan(model, cm, W, q, want_pos=not NO_POS)
nb = ops.n_blocks(B * H)
bs = torch.zeros(nb, **ta)
for _ in range(200):
    plan.launch(bs.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000):
    plan.launch(bs.data_ptr())
e1.record()
torch.cuda.synchronize()
print(f"[{TAG}] launch time by events: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us")
stamps = torch.zeros((nb, 8), device=dev, dtype=torch.int64)
lib().trk_debug_set_stamp_buffer(stamps.data_ptr())
# default: the stamped launch is the LAST of a back-to-back train (steady state: the XCDs are awake; an isolated launch after a
# sync shows them starting ~0.17 us apart, tools/dispatch_ramp.hip); --isolated stamps a single launch on an idle GPU
for _ in range(1 if "--isolated" in sys.argv else 6):
    plan.launch(bs.data_ptr())
torch.cuda.synchronize()
lib().trk_debug_set_stamp_buffer(None)
raw = stamps.cpu().numpy()
out_dir = ROOT / "gpurun_out"
out_dir.mkdir(exist_ok=True)
np.save(out_dir / f"phase_stamps_{TAG}.npy", raw)                      # raw ticks, for offline analysis
s = raw.astype(np.float64)
t0 = s[:, 0].min()
s -= t0
names = ["entry", "q loaded", "kernargs arrived", "pos staged", "objects done", "objectives done", "reverse done", "exit"]
order = [0, 2, 1, 3, 4, 5, 6, 7]
print("stamp clock: s_memtime ticks (100 MHz REFCLK on gfx9-family = 10 ns per tick, or shader clock; see spread below)")
print(f"{'phase':18s} {'mean start':>12s} {'p5':>10s} {'p95':>10s} {'mean dur to next':>18s}")
rel = s - s[:, :1]
for j, k in enumerate(order):
    nxt = order[j + 1] if j < 7 else None
    dur = (s[:, nxt] - s[:, k]).mean() if nxt is not None else 0.0
    print(f"{names[k]:18s} since entry: mean {rel[:, k].mean():9.1f}  p5 {np.percentile(rel[:, k], 5):9.1f}  p95 {np.percentile(rel[:, k], 95):9.1f}   to next: {dur:9.1f}")
print("kernel span (first entry -> last exit):", s[:, 7].max(), "ticks;  wave lifetime mean:", (s[:, 7] - s[:, 0]).mean())
