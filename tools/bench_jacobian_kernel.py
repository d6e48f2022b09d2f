#!/usr/bin/env python3
"""trk_fk_jacobian with pre-allocated outputs (kernel time, not the Python wrapper's): batch sweep on the generated kernels."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd._lib import lib
dev = torch.device("cuda:0")
kw = dict(device=dev, dtype=torch.float32)
for ident in sys.argv[1:] or ["panda", "dual_panda", "ur10_allegro"]:
    kin, _ = codegen.template_for(ident)
    h = ops.ModelHandle(kin)
    D = kin.n_dofs
    link = kin.name_to_idx.get("ee_link", kin.n_links - 1)
    for B in (3328, 4096, 8192):
        n = B * 64
        q = (torch.rand(n, D, **kw) - 0.5) * 3.0
        pos, quat = torch.empty((n, 3), **kw), torch.empty((n, 4), **kw)
        lin, ang = torch.empty((n, 3, D), **kw), torch.empty((n, 3, D), **kw)
        st = torch.cuda.current_stream().cuda_stream
        args = (h._h, q.data_ptr(), None, n, int(link), pos.data_ptr(), quat.data_ptr(), lin.data_ptr(), ang.data_ptr(), None, None, st)
        L = lib()
        for _ in range(20): L.trk_fk_jacobian(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): L.trk_fk_jacobian(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        nbytes = (4 * D + 12 + 16 + 24 * D) * n
        print(f"{ident:14s} batch {B:5d} x 64: {us:7.2f} us  {nbytes / us / 1e6:6.2f} TB/s ({nbytes / us / 8e4:4.1f} % of 8 TB/s)  {us / B * 4096:6.2f} us per 4096")
