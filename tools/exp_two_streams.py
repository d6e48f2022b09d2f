#!/usr/bin/env python3
"""Experiment: independent evaluations alternated over 1 / 2 / 3 HIP streams (two output buffer sets per stream are not needed:
each stream owns its plan).  With one stream a launch waits for the previous one to drain; with two the next launch's dispatch,
kernarg / q fetch and FK overlap the write tail of the previous one.  NOT the headline measurement (bench.py keeps one stream:
a planner's iterations depend on each other); this is the throughput available to independent batches."""
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
model, cm = task._fused_handles(dev)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    plans, sums = [], []
    for _ in range(ns):
        q = robot.random_q(4096 * 64).reshape(4096, 64, 7).contiguous()
        plans.append(ops.RolloutPlan(model, cm, (0, 1, 0, 1), q))
        sums.append(torch.zeros(ops.n_blocks(4096 * 64), **ta))
    torch.cuda.synchronize()
    def run(n):
        for i in range(n):
            k = i % ns
            plans[k].launch(sums[k].data_ptr(), streams[k].cuda_stream)
    run(300); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); run(3000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {dt / 3000 * 1e6:.2f} us per evaluation ({4096 * 64 * 3000 / dt:.3g} rollouts/s, {192 * 4096 * 64 * 3000 / dt / 8e12 * 100:.1f} % of 8 TB/s)")
