#!/usr/bin/env python3
"""Run-time ablation of the fused attached-point rollout at 4096 x 64: which part of the launch is what?  Weights (self, obj, ws, ee) zeroed one
at a time, with and without the position output."""
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
for tag, kw in (("45 link spheres", dict(link_sphere_model="panda")),
                ("grasped box", dict(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))),
                ("spheres + grasped box", dict(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA)))):
    robot = tra.RobotPanda(tensor_args=TA, **kw)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = torch.eye(4); T[:3, 3] = torch.tensor([0.4, 0.2, 0.5]); task.set_ee_target(T)
    q = robot.random_q(B * H).reshape(B, H, 7)
    ps = robot._point_set(dev)
    model, cm = task._fused_handles(dev)
    print(f"{tag} (P = {ps.n_points})")
    for name, w in (("all terms", (1, 1, 1, 1)), ("no self pairs", (0, 1, 1, 1)), ("no objects", (1, 0, 1, 1)), ("no workspace box", (1, 1, 0, 1)),
                    ("no EE", (1, 1, 1, 0)), ("objects only", (0, 1, 0, 0)), ("self only", (1, 0, 0, 0)), ("EE only", (0, 0, 0, 1))):
        a = t(lambda: ops.rollout_points_cost_grad(ps, cm, w, q, want_pos=True))
        b = t(lambda: ops.rollout_points_cost_grad(ps, cm, w, q, want_pos=False))
        print(f"   {name:18s} with positions {a:7.1f} us   without {b:7.1f} us")
    print(f"   {'positions only':18s} {t(lambda: ops.fk_points(ps, q.reshape(-1, 7))):7.1f} us")
