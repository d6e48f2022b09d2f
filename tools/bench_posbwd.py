#!/usr/bin/env python3
"""trk_fk_positions / trk_fk_positions_backward (generated Panda kernels) at 4096 x 64, kernel time from HIP events."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
ta = dict(device=dev, dtype=torch.float32)
robot = tra.RobotPanda(tensor_args=ta)
h = ops.ModelHandle(robot.diff_panda._kin)
N, D, L = 4096 * 64, 7, 11
q = robot.random_q(N).contiguous()
g = torch.randn(N, L, 3, **ta)
def t(fn, n=500):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
pos = torch.empty(N, L, 3, **ta); gq = torch.empty(N, D, **ta)
for name, fn, nbytes in (("fk_positions", lambda: ops.fk_positions(h, q, out=pos), 4 * D + 12 * L),
                         ("fk_positions_backward", lambda: ops.fk_positions_backward(h, q, g, out=gq), 8 * D + 12 * L)):
    try:
        us = t(fn)
    except TypeError:
        fn2 = (lambda: ops.fk_positions(h, q)) if name == "fk_positions" else (lambda: ops.fk_positions_backward(h, q, g))
        us = t(fn2)
    print(f"{name:24s} {us:7.2f} us  ({nbytes * N / us / 8e4:5.1f} % of 8 TB/s)")
