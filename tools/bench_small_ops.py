#!/usr/bin/env python3
"""The small streaming ops either side of the hot path at planner sizes (4096 x 64 x 7 unless stated), through Python (so ~3 us of every
figure is the wrapper): interpolate_traj_via_points, finite_difference, traj_diff_norm_sum, sdf_points, gp_prior_cost_grad.
usage: tools/bench_small_ops.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops

dev = torch.device("cuda:0")
TA = dict(device=dev, dtype=torch.float32)


def t(fn, n=200, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def line(name, us, b):
    print(f"{name:64s} {us:8.1f} us  {b / 1e6:7.1f} MB  {b / us / 1e3 / 8000 * 100:5.1f} % of 8 TB/s")


B, H, D = 4096, 64, 7
x = torch.randn(B, H, D, **TA)
line("interpolate_traj_via_points (5 per segment)", t(lambda: ops.interpolate_traj_via_points(x, 5)), 4 * B * D * (H + (H - 1) * 5))
for m in ("forward", "central"):
    line(f"finite_difference ({m})", t(lambda: ops.finite_difference(x, 0.1, m)), 8 * B * H * D)
xs = torch.randn(B, H, 2 * D, **TA)
line("traj_diff_norm_sum (path length, positions of a 14-column state)", t(lambda: ops.traj_diff_norm_sum(xs, 0, D)), 4 * B * H * 2 * D + 4 * B)
robot = tra.RobotPanda(tensor_args=TA)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
_, cm = task._fused_handles(dev)
pts = torch.rand(B * H, 3, **TA) * 2 - 1
line("sdf_points (262 144 points, 1 object of 10 spheres, value + gradient)", t(lambda: ops.sdf_points(cm, pts, want_grad=True)), 4 * B * H * (3 + 1 + 3))
q, qd = torch.randn(2048, 128, 14, **TA), torch.randn(2048, 128, 14, **TA)
line("gp_prior_cost_grad fp32 (2048 x 128 x 14)", t(lambda: ops.gp_prior_cost_grad(q, qd, 5 / 128, 0.1)), 4 * 2048 * 128 * 14 * 4 + 4 * 2048)
qh, qdh = q.half(), qd.half()
line("gp_prior_cost_grad fp16 (2048 x 128 x 14)", t(lambda: ops.gp_prior_cost_grad(qh, qdh, 5 / 128, 0.1, grad_scale=2.0 ** -12)), 2 * 2048 * 128 * 14 * 4 + 4 * 2048)
