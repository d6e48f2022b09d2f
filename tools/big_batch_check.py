#!/usr/bin/env python3
"""Multi-GiB outputs (33.5 M Panda samples = 4.1 GiB of positions; 8.4 M for the tree robots): 64-bit addressing and large grids,
checked by sharding invariance (slices of the big batch evaluated alone give the same bits) and matrices == positions."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import torch_robotics_amd as tra
from torch_robotics_amd import ops, codegen
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D
dev = torch.device("cuda:0"); TA = dict(device=dev, dtype=torch.float32)
env = EnvSpheres3D(tensor_args=TA)
for ident, B in (("panda", 1 << 19), ("ur10_allegro", 1 << 17), ("dual_panda", 1 << 17)):
    kin, tmpl = codegen.template_for(ident)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.1, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = tmpl.ee_link
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    if tmpl.ee2_link >= 0: spec.ee2_link = tmpl.ee2_link; spec.ee2_target = Ht
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
    H, D, L = 64, kin.n_dofs, kin.n_links
    q = (torch.rand(B, H, D, **TA) - 0.5) * 4
    pos, c, gq = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), q)
    torch.cuda.synchronize()
    print(ident, "n =", B * H, "positions", pos.numel() * 4 / 2**30, "GiB", "finite:", bool(torch.isfinite(c).all()), bool(torch.isfinite(gq).all()))
    for b0 in (0, B // 2 - 3, B - 5):
        p2, c2, g2 = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), q[b0:b0 + 5].contiguous())
        assert torch.equal(p2, pos[b0:b0 + 5]) and torch.equal(c2, c[b0:b0 + 5]) and torch.equal(g2, gq[b0:b0 + 5]), (ident, b0)
    Hm = ops.fk_forward(h, q[: B // 8].reshape(-1, D))
    assert float((Hm[..., :3, 3].reshape(B // 8, H, L, 3) - pos[: B // 8]).abs().max()) < 4e-6
    del pos, c, gq, q, Hm
    torch.cuda.empty_cache()
print("ok")
