#!/usr/bin/env python3
"""Experiment: ONE batch of 4096 x 64 evaluated as ns sub-batches of whole trajectories, each on its own HIP stream (the
trajectories of a batch are independent, so a planner can carry each sub-batch through its iterations on its own stream:
iteration i+1 of a sub-batch waits only for iteration i of the same sub-batch).  One "step" = all ns sub-launches."""
import sys, time
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
B, H = 4096, 64
q = robot.random_q(B * H).reshape(B, H, 7).contiguous()
for ns in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    bounds = [B * k // ns for k in range(ns + 1)]
    plans = [ops.RolloutPlan(model, cm, (0, 1, 0, 1), q[bounds[k]:bounds[k + 1]].contiguous()) for k in range(ns)]
    sums = [torch.zeros(ops.n_blocks((bounds[k + 1] - bounds[k]) * H), **ta) for k in range(ns)]
    def run(n):
        for _ in range(n):
            for k in range(ns):
                plans[k].launch(sums[k].data_ptr(), streams[k].cuda_stream)
    run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{ns} sub-batch(es) / stream(s): {dt / 2000 * 1e6:.2f} us per step of {B} x {H} ({B * H * 2000 / dt:.3g} rollouts/s, {192 * B * H * 2000 / dt / 8e12 * 100:.1f} % of 8 TB/s)")
