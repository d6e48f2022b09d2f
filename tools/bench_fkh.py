#!/usr/bin/env python3
"""trk_fk_forward, all links (compute_forward_kinematics_all_links): generated k_fkh vs the table-driven kernel, kernel time."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd._lib import lib
dev = torch.device("cuda:0")
kw = dict(device=dev, dtype=torch.float32)
for ident in [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["panda", "dual_panda", "ur10_allegro"]:
    kin, _ = codegen.template_for(ident)
    h = ops.ModelHandle(kin)
    D, L = kin.n_dofs, kin.n_links
    n = int(sys.argv[sys.argv.index("--batch") + 1]) * 64 if "--batch" in sys.argv else 4096 * 64
    q = (torch.rand(n, D, **kw) - 0.5) * 3.0
    H = torch.empty((n, L, 4, 4), **kw)
    st = torch.cuda.current_stream().cuda_stream
    for on in (True, False):
        h.enable_specialized(on)
        args = (h._h, q.data_ptr(), n, None, 0, H.data_ptr(), st)
        Lb = lib()
        for _ in range(10): Lb.trk_fk_forward(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): Lb.trk_fk_forward(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        nbytes = (4 * D + 64 * L) * n
        print(f"{ident:14s} {'generated' if on else 'table-driven':12s} {us:8.2f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 8e4:5.1f} % of 8 TB/s")
        gq = torch.empty((n, D), **kw)
        bargs = (h._h, q.data_ptr(), H.data_ptr(), n, None, 0, gq.data_ptr(), st)
        for _ in range(10): Lb.trk_fk_backward(*bargs)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100): Lb.trk_fk_backward(*bargs)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        nbytes = (8 * D + 64 * L) * n
        print(f"{ident:14s} {'generated' if on else 'table-driven':12s} {us:8.2f} us  {nbytes / 1e6:7.1f} MB  {nbytes / us / 8e4:5.1f} % of 8 TB/s   (reverse mode)")
