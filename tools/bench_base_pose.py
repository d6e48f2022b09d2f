#!/usr/bin/env python3
"""Headline workload with the robot's base at the identity (k_rollout_bi) and at a general pose (k_rollout_bg), same box."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
task.set_ee_target(Ht)
B, H = 4096, 64
q = robot.random_q(B * H).reshape(B, H, 7).contiguous()
model, cm = task._fused_handles(dev)
kin = robot.diff_panda._kin
for name, pose in (("identity base", None), ("general base pose", np.array([0.1234, -0.2345, 0.0567, 0.9238795, 0.0, 0.3826834, 0.0], np.float32))):
    if pose is not None:
        kin.set_base_pose(pose)
        model.set_base_pose(kin.base_R, kin.base_t)
    for w, wn in (((0, 1, 0, 1), "obj+ee"), ((1, 1, 1, 1), "all four")):
        plan = ops.RolloutPlan(model, cm, w, q)
        for _ in range(50): plan.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(1000): plan.launch()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:20s} {wn:9s} {e0.elapsed_time(e1):7.2f} us")
