#!/usr/bin/env python3
"""Experiment: does the per-wavefront cost-sum store (a plain 4-byte store per wave) or the cost store change the launch time?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); task.set_ee_target(Ht)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
model, cm = task._fused_handles(dev)
plan = ops.RolloutPlan(model, cm, (0, 1, 0, 1), q)
bs = torch.zeros(ops.n_blocks(4096 * 64), **ta)
def t(ptr, n=3000):
    for _ in range(300): plan.launch(ptr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): plan.launch(ptr)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print(f"with cost_sum {t(bs.data_ptr()):.2f} us | without {t(None):.2f} us")
