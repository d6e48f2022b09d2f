#!/usr/bin/env python3
"""Experiment: the headline workload with fp16 I/O (half the bytes, same arithmetic) next to fp32 -- is the kernel bound by bytes?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
task.set_ee_target(Ht)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
model, cm = task._fused_handles(dev)
for name, qq in (("fp32", q), ("fp16", q.half())):
    for want_pos in (True, False):
        plan = ops.RolloutPlan(model, cm, (0, 1, 0, 1), qq, want_pos=want_pos)
        for _ in range(200):
            plan.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3000):
            plan.launch()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} I/O, link positions {'written' if want_pos else 'not written'}: {e0.elapsed_time(e1) / 3000 * 1e3:.2f} us")
