#!/usr/bin/env python3
"""trk_rollout_collision kernel time (pre-allocated output, direct C call) per field mask at 4096 x 64."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd._lib import lib
dev = torch.device("cuda:0"); ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=ta), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=ta)
q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
model, cm = task._fused_handles(dev)
out = torch.empty(4096 * 64, device=dev, dtype=torch.bool)
st = torch.cuda.current_stream().cuda_stream
L = lib()
for name, fl in (("self", 1), ("objects", 2), ("workspace", 4), ("objects + ws", 6), ("all three", 7)):
    for mg, mname in ((float("nan"), "own margins"), (0.0, "margin 0")):
        args = (model._h, cm._h, fl, q.data_ptr(), 4096, 64, mg, out.data_ptr(), None, st)
        for _ in range(20): L.trk_rollout_collision(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500): L.trk_rollout_collision(*args)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:14s} {mname:12s} {e0.elapsed_time(e1) / 500 * 1e3:6.2f} us   in collision: {float(out.float().mean()):.3f}")
