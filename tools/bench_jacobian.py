import sys; from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, numpy as np
from torch_robotics_amd import codegen, ops
dev = torch.device("cuda:0")
for ident in ("panda", "dual_panda", "ur10_allegro"):
    kin, tmpl = codegen.template_for(ident)
    h = ops.ModelHandle(kin)
    n = 4096 * 64
    q = (torch.rand(n, kin.n_dofs, device=dev) - 0.5) * 3.0
    link = kin.name_to_idx.get("ee_link", kin.n_links - 1)
    for on in (True, False):
        h.enable_specialized(on)
        for _ in range(5): ops.fk_jacobian(h, q, None, link)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.fk_jacobian(h, q, None, link)
        e1.record(); torch.cuda.synchronize()
        print(ident, "generated" if on else "table-driven", f"{e0.elapsed_time(e1) / 30 * 1e3:.1f} us", "link", link)
