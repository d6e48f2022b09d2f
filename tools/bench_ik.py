#!/usr/bin/env python3
"""trk_ik_step (one fused IK iteration: FK + SE3 distance + joint-limit hinge + Adam, in place): time per iteration."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
kin = robot.diff_panda._kin
h = ops.ModelHandle(kin)
lo, hi = torch.as_tensor(kin.lower_dof if hasattr(kin, "lower_dof") else robot.q_min.cpu().numpy(), **ta), torch.as_tensor(kin.upper_dof if hasattr(kin, "upper_dof") else robot.q_max.cpu().numpy(), **ta)
Ht = torch.eye(4, **ta); Ht[:3, 3] = torch.tensor([0.4, 0.2, 0.5])
for n in (1024, 16384, 262144):
    q = robot.random_q(n).contiguous(); m = torch.zeros_like(q); v = torch.zeros_like(q)
    loss = torch.empty(n, **ta); valid = torch.empty(n, device=dev, dtype=torch.uint8)
    for it in range(20): ops.ik_step(h, kin.n_links - 1, Ht, lo, hi, q, m, v, it + 1, loss=loss, valid=valid)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(200): ops.ik_step(h, kin.n_links - 1, Ht, lo, hi, q, m, v, it + 21, loss=loss, valid=valid)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    e0.record()
    for it in range(10): ops.ik_steps(h, kin.n_links - 1, Ht, lo, hi, q, m, v, 221 + 20 * it, 20, loss=loss, valid=valid)
    e1.record(); torch.cuda.synchronize()
    us20 = e0.elapsed_time(e1) / 200 * 1e3
    print(f"n = {n:7d}: {us:8.2f} us per iteration, one per launch | {us20:8.2f} us per iteration, 20 per call (trk_ik_steps)")
