#!/usr/bin/env python3
"""trk_jtj (J^T J, J^T r of the geometric Jacobian) at 4096 x 64: the per-lane FMA kernel against the
v_mfma_f32_4x4x1_16b_f32 kernel, kernel-bound timing through the C ABI, with the op's HBM roofline beside it."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops
dev = torch.device("cuda:0")
for name, tree in (("Panda (7 DOF)", tra.DifferentiableFrankaPanda(device=dev)), ("UR10 (6 DOF)", tra.DifferentiableUR10(device=dev))):
    h = tree._handle
    D = h.n_dofs
    N = 4096 * 64
    q = torch.rand(N, D, device=dev) * 2 - 1
    _, _, lin, ang = ops.fk_jacobian(h, q, None, h.n_links - 1)
    r = torch.randn(N, 6, device=dev)
    bytes_ = 4 * N * (6 * D + 6 + D * D + D)
    for mfma in (False, True):
        for _ in range(20): ops.jtj(lin, ang, r, mfma=mfma)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for _ in range(n): ops.jtj(lin, ang, r, mfma=mfma)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name:14s} {'MFMA 4x4x1_16b' if mfma else 'VALU per-lane '}  {dt * 1e6:7.1f} us per call   {bytes_ / dt / 1e9:7.0f} GB/s "
              f"= {bytes_ / dt / 8e12 * 100:4.1f} % of the HBM peak ({bytes_ / 1e6:.0f} MB per call)")
