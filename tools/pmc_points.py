#!/usr/bin/env python3
"""Minimal driver for PMC passes over the attached-point fused kernel (Panda + 45 link spheres, 4096 x 64)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
TA = dict(device=dev, dtype=torch.float32)
kw = dict(link_sphere_model="panda") if len(sys.argv) < 2 or sys.argv[1] == "spheres" else dict(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))
robot = tra.RobotPanda(tensor_args=TA, **kw)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
T = torch.eye(4); T[:3, 3] = torch.tensor([0.4, 0.2, 0.5]); task.set_ee_target(T)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7)
ps = robot._point_set(dev)
model, cm = task._fused_handles(dev)
for _ in range(20):
    ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 1), q)
torch.cuda.synchronize()
