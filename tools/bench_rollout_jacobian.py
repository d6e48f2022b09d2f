#!/usr/bin/env python3
"""trk_rollout_jacobian_cost_grad (round 6): the fused rollout + geometric Jacobian of the tracked link in ONE launch against the two launches
(RolloutPlan + JacobianPlan), per robot, 4096 x 64, pre-bound calls timed with HIP events."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from torch_robotics_amd import codegen, ops
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D

dev = torch.device("cuda:0")
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))


def t_us(fn, n=300, w=30):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for ident in ("panda", "iiwa7", "ur10", "dual_panda", "ur10_allegro"):
    kin, tmpl = codegen.template_for(ident)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = tmpl.ee_link
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    if tmpl.ee2_link >= 0:
        spec.ee2_link, spec.ee2_target = tmpl.ee2_link, Ht
    spec.validate()
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
    B, H, D, L = 4096, 64, kin.n_dofs, kin.n_links
    q = ((torch.rand(B, H, D, device=dev) - 0.5) * 3.0).contiguous()
    fused = ops.RolloutJacobianPlan(h, cm, (0, 1, 0, 1), q, tmpl.ee_link)
    roll = ops.RolloutPlan(h, cm, (0, 1, 0, 1), q)
    jac = ops.JacobianPlan(h, q.reshape(-1, D), tmpl.ee_link)
    fused.launch(); torch.cuda.synchronize()
    one = ops.last_dispatch() == "generated"

    def two():
        roll.launch(); jac.launch()
    bps = 4 * D + 12 * L + 4 + 4 * D + 28 + 24 * D
    a, b = t_us(fused.launch), t_us(two)
    print(f"{ident:14s} {L:2d} links {D:2d} DOF  {bps:5d} B/sample   one call {a:6.2f} us ({'ONE launch' if one else 'two launches'}, "
          f"{bps * B * H / a / 1e3 / 8000:.3f} of 8 TB/s)   rollout + Jacobian as two calls {b:6.2f} us", flush=True)
