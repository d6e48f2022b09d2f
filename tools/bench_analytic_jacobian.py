#!/usr/bin/env python3
"""trk_fk_analytic_jacobian (robot_tree.py `compute_analytical_jacobian_all_links`: d [pos, quat] / d q of EVERY link, (N, L, 7, D)) at
4096 x 64 configurations (and at sizes whose output stays inside the 256 MB Infinity Cache): a write stream of 28 L D bytes per sample;
the generated kernel (k_ajac) against the table-driven one.  usage: tools/bench_analytic_jacobian.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd import ops
from torch_robotics_amd.kinmodel import KinModel

dev = torch.device("cuda:0")
ROOT = Path(__file__).resolve().parent.parent


def t(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N in (("panda_arm_no_gripper", 262144), ("ur10", 262144), ("iiwa7", 262144), ("panda_arm_no_gripper", 98304), ("ur10", 131072),
                ("panda_arm_no_gripper", 32768)):
    m = KinModel.from_urdf(str(ROOT / "torch_robotics_amd" / "data" / "urdf" / f"{name}.urdf")) if (ROOT / "torch_robotics_amd" / "data" / "urdf" / f"{name}.urdf").exists() else None
    if m is None:
        from tests.helpers import model
        m = model(name)
    h = ops.ModelHandle(m)
    q = (torch.rand(N, m.n_dofs, device=dev) - 0.5) * 2.0
    us = t(lambda: ops.fk_analytic_jacobian(h, q))
    h.enable_specialized(False)
    us_t = t(lambda: ops.fk_analytic_jacobian(h, q), n=10, w=2)
    h.enable_specialized(True)
    b = N * (4 * m.n_dofs + 28 * m.n_links * m.n_dofs)
    print(f"{name:24s} N {N:7d}  {m.n_links:2d} links {m.n_dofs:2d} DOF  {us:9.1f} us  {b / 1e6:7.1f} MB  {b / us / 1e3 / 8000 * 100:5.1f} % of 8 TB/s   "
          f"(table-driven kernel: {us_t:7.1f} us)")
