#!/usr/bin/env python3
"""Fused rollout on the robots of BASELINE configs 4 / 5 (and Panda): specialised vs table-driven kernel, 4096 x 64."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd.costmodel import CostModelSpec, make_object, sphere_prims
from torch_robotics_amd.environments import EnvSpheres3D

dev = torch.device("cuda:0")
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))


def t(fn, n=300, w=30):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for ident in ("panda", "dual_panda", "ur10_allegro"):
    kin, tmpl = codegen.template_for(ident)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
    sl = sorted({a for p in tmpl.self_pairs for a in p})
    spec.self_link_idx = np.asarray(sl, np.int32)
    spec.self_pairs = np.asarray([(sl.index(a), sl.index(b)) for a, b in tmpl.self_pairs], np.int32).reshape(-1, 2)
    spec.self_margin = np.full(len(tmpl.self_pairs), 0.05, np.float32)
    spec.ee_link = tmpl.ee_link
    if tmpl.ee2_link >= 0:          # two-arm template: the second arm tracks its own target
        spec.ee2_link = tmpl.ee2_link
        Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
    B, H, D, L = 4096, 64, kin.n_dofs, kin.n_links
    q = (torch.rand(B, H, D, device=dev) - 0.5) * 3.0
    nbytes = (8 * D + 12 * L + 4) * B * H
    for w, tag in (((0, 1, 0, 1), "obj+ee"), ((1, 1, 1, 1), "all")):
        plan = ops.RolloutPlan(h, cm, w, q)
        h.enable_specialized(True); ts = t(lambda: plan.launch())
        h.enable_specialized(False); tg = t(lambda: plan.launch(), n=50, w=5)
        h.enable_specialized(True)
        print(f"{ident:14s} L={L:2d} D={D:2d} {tag:7s} specialised {ts:7.1f} us ({nbytes / ts / 1e3 / 8e3 * 100:4.1f} % of 8 TB/s, {B * H / ts * 1e6:.3g} rollouts/s)   table-driven {tg:7.1f} us")
