#!/usr/bin/env python3
"""GP-prior kernel at config 5's per-GPU share (2048 x 128 x 14, fp16 and fp32 I/O, accumulating into existing gradients)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
B, H, D = 2048, 128, 14
for dt_, esz in ((torch.float16, 2), (torch.float32, 4)):
    q = (torch.randn(B, H, D, device=dev) * 0.1).to(dt_); qd = (torch.randn(B, H, D, device=dev) * 0.1).to(dt_)
    a0, a1 = torch.zeros_like(q), torch.zeros_like(q)
    for acc in (True, False):
        fn = (lambda: ops.gp_prior_cost_grad(q, qd, 5.0 / H, 0.1, 1.0, accumulate_into=(a0, a1))) if acc else (lambda: ops.gp_prior_cost_grad(q, qd, 5.0 / H, 0.1, 1.0))
        for _ in range(50): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 500 * 1e3
        nbytes = B * H * D * esz * (6 if acc else 4) + 4 * B
        print(f"{str(dt_):14s} accumulate={acc!s:5s}: {us:6.2f} us  {nbytes / 1e6:5.1f} MB  {nbytes / us / 1e3:6.0f} GB/s ({nbytes / us / 8e6 * 100:4.1f} % of 8 TB/s)")
