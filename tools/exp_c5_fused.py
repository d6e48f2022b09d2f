#!/usr/bin/env python3
"""Where the time of config 5's fused launch goes: fused / rollout-only / prior-only, with and without positions, objectives off."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D

dev = torch.device("cuda:0")
ident = sys.argv[1] if len(sys.argv) > 1 else "dual_panda"
B, H = (2048, 128) if ident == "dual_panda" else (4096, 64)
kin, tmpl = codegen.template_for(ident)
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))
spec = CostModelSpec(n_links_in=kin.n_links)
spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
spec.objects = [o.as_object() for o in env.obj_fixed_list]
spec.ee_link = tmpl.ee_link
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
if tmpl.ee2_link >= 0:
    spec.ee2_link = tmpl.ee2_link
    Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
spec.validate()
h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
D = kin.n_dofs
dt = 5.0 / H
q = (torch.cumsum(torch.randn(B, H, D, device=dev) * 0.02, 1) + (torch.rand(B, 1, D, device=dev) - 0.5) * 2.0)
qd = torch.zeros_like(q); qd[:, :-1] = (q[:, 1:] - q[:, :-1]) / dt


def t_us(fn, n=300, w=30):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for io in ("f16", "f32"):
    tq, tqd = (q.half().contiguous(), qd.half().contiguous()) if io == "f16" else (q.contiguous(), qd.contiguous())
    gs = 2.0 ** -12 if io == "f16" else 1.0
    for wts in ((0, 1, 0, 1), (0, 0, 0, 0)):
        for pos in (True, False):
            fused = ops.RolloutGpPlan(h, cm, wts, tq, tqd, dt, 0.1, 1.0, want_pos=pos, grad_scale=gs)
            roll = ops.RolloutPlan(h, cm, wts, tq, want_pos=pos, grad_scale=gs)
            print(f"{ident} {io} weights {wts} positions {int(pos)}:  fused {t_us(fused.launch):6.2f} us   rollout alone {t_us(roll.launch):6.2f} us")
    gp = ops.GPPriorPlan(tq, tqd, dt, 0.1, 1.0, grad_scale=gs)
    print(f"{ident} {io} GP prior alone {t_us(gp.launch):6.2f} us")
