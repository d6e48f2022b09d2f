#!/usr/bin/env python3
"""trk_fk_positions / trk_fk_positions_backward, generated vs table-driven kernels, kernel time (pre-allocated outputs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd._lib import lib
dev = torch.device("cuda:0")
kw = dict(device=dev, dtype=torch.float32)
Lb = lib()
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for ident in sys.argv[1:] or ["panda", "dual_panda", "ur10_allegro"]:
    kin, _ = codegen.template_for(ident)
    h = ops.ModelHandle(kin)
    D, L = kin.n_dofs, kin.n_links
    n = 4096 * 64
    q = (torch.rand(n, D, **kw) - 0.5) * 3.0
    pos = torch.empty((n, L, 3), **kw); gq = torch.empty((n, D), **kw)
    st = torch.cuda.current_stream().cuda_stream
    for on in (True, False):
        h.enable_specialized(on)
        f = t(lambda: Lb.trk_fk_positions(h._h, q.data_ptr(), n, None, 0, pos.data_ptr(), st))
        b = t(lambda: Lb.trk_fk_positions_backward(h._h, q.data_ptr(), pos.data_ptr(), n, None, 0, gq.data_ptr(), st))
        print(f"{ident:14s} {'generated' if on else 'table-driven':12s} positions {f:7.2f} us ({(4 * D + 12 * L) * n / f / 8e4:4.1f} %)   reverse mode {b:7.2f} us ({(8 * D + 12 * L) * n / b / 8e4:4.1f} %)")
