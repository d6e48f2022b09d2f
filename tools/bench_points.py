#!/usr/bin/env python3
"""Attached-point paths at 4096 x 64: Panda with the 45-sphere link model and with a grasped box.
Reports the fused point rollout, fk_map_collision and its backward against their HBM rooflines.
`--only fused|plan|positions|backward` runs one row per model: under `rocprofv3 --kernel-trace --stats` the fused rollout and the positions-only
launch are the SAME kernel (the positions-only exit), so a trace of all rows averages them; one row per run separates them."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops

dev = torch.device("cuda:0")
TA = dict(device=dev, dtype=torch.float32)


def t(fn, n=100, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, H = 4096, 64
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
for tag, kw in (("45 link spheres", dict(link_sphere_model="panda")),
                ("grasped box", dict(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))),
                ("spheres + grasped box", dict(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA)))):
    robot = tra.RobotPanda(tensor_args=TA, **kw)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = torch.eye(4); T[:3, 3] = torch.tensor([0.4, 0.2, 0.5]); task.set_ee_target(T)
    q = robot.random_q(B * H).reshape(B, H, 7)
    ps = robot._point_set(dev)
    P, D = ps.n_points, 7
    model, cm = task._fused_handles(dev)
    n = B * H
    gpos = torch.randn(n, P, 3, device=dev)
    q2 = q.reshape(n, D)
    plan = ops.PointsRolloutPlan(ps, cm, (1, 1, 1, 1), q.contiguous())
    rows = (("fused", "fused rollout (self+obj+ws+ee)", lambda: ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 1), q), 8 * D + 12 * P + 4),
            ("plan", "fused rollout, pre-bound (PointsRolloutPlan)", plan.launch, 8 * D + 12 * P + 4),
            ("positions", "fk_map_collision", lambda: ops.fk_points(ps, q2), 4 * D + 12 * P),
            ("backward", "fk_map_collision backward", lambda: ops.fk_points_backward(ps, q2, gpos), 8 * D + 12 * P))
    for key, name, fn, bytes_per in rows:
        if ONLY is not None and key != ONLY:
            continue
        us = t(fn)
        print(f"{tag:22s} P={P:3d} {name:44s} {us:8.1f} us  {bytes_per * n / us / 1e3 / 8e3 * 100:5.1f} % of 8 TB/s  ({n / us * 1e6:.3g} /s)")
