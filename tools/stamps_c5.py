#!/usr/bin/env python3
"""Phase stamps of config 5's fused launch (k_rollout_gpt: dual Panda, fp16 I/O, obstacles + both EEs + GP prior, 2048 x 128): where a wavefront's
9 - 12 us go.  Writes gpurun_out/phase_stamps_c5.npy (tools/phase_analyze.py reads it) and prints the per-phase means of the two generations."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd._lib import lib
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D

dev = torch.device("cuda:0")
B, H = 2048, 128
kin, tmpl = codegen.template_for("dual_panda")
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))
spec = CostModelSpec(n_links_in=kin.n_links)
spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
spec.objects = [o.as_object() for o in env.obj_fixed_list]
spec.ee_link = tmpl.ee_link
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
spec.ee2_link = tmpl.ee2_link
Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
spec.validate()
h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
D = kin.n_dofs
dt = 5.0 / H
q = (torch.cumsum(torch.randn(B, H, D, device=dev) * 0.02, 1) + (torch.rand(B, 1, D, device=dev) - 0.5) * 2.0)
qd = torch.zeros_like(q); qd[:, :-1] = (q[:, 1:] - q[:, :-1]) / dt
tq, tqd = q.half().contiguous(), qd.half().contiguous()
plan = ops.RolloutGpPlan(h, cm, (0, 1, 0, 1), tq, tqd, dt, 0.1, 1.0, want_pos=True, grad_scale=2.0 ** -12)
for _ in range(20): plan.launch()
torch.cuda.synchronize()
nb = ops.n_blocks(B * H)
stamps = torch.zeros((nb, 8), device=dev, dtype=torch.int64)
lib().trk_debug_set_stamp_buffer(stamps.data_ptr())
for _ in range(6): plan.launch()
torch.cuda.synchronize()
lib().trk_debug_set_stamp_buffer(None)
r = stamps.cpu().numpy()
out = ROOT / "gpurun_out"; out.mkdir(exist_ok=True)
np.save(out / "phase_stamps_c5.npy", r)
t = r.astype(np.int64)
names = {1: "rows in, prior, gqd out", 3: "FK + position staging", 4: "scene + workspace", 5: "EE terms", 6: "reverse pass", 7: "gq out, exit"}
prev = {1: 0, 3: 1, 4: 3, 5: 4, 6: 5, 7: 6}
entry = (t[:, 2] & ((1 << 44) - 1)); entry = (entry - entry.min()) / 100.0
gen2 = entry > 2.0
GHZ = 2.1
for g, m in (("first generation", ~gen2), ("second generation", gen2)):
    print(f"{g}: {int(m.sum())} wavefronts, entry p50 {np.percentile(entry[m], 50):.2f} us, lifetime {((t[m, 7] - t[m, 0]).mean()):.0f} ticks")
    for k in (1, 3, 4, 5, 6, 7):
        d = (t[m, k] - t[m, prev[k]])
        print(f"   {names[k]:28s} {d.mean():8.0f} ticks  ({d.mean() / (t[m, 7] - t[m, 0]).mean() * 100:4.1f} %)")
