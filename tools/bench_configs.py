#!/usr/bin/env python3
"""BASELINE configs 4 and 5 on ONE GPU (config 5: one rank's share of the 4-way batch shard).

c4: UR10 + Allegro hand (22 DOF, 30 links), batch 4096 x horizon 64: fused FK + SDF-obstacle + EE cost + gradient,
    then the geometric Jacobian of `ee_link` (trk_fk_jacobian).
c5: dual Panda (14 DOF, 23 links), batch 8192 x horizon 128 sharded 4-way -> 2048 x 128 per GPU, fp16 q / qd / link
    positions / gradients in HBM, fp32 arithmetic and cost: fused FK + SDF-obstacle + EE cost + gradient, then the
    constant-velocity GP prior accumulated into the same gradient (trk_gp_prior_cost_grad).
Prints one JSON object per config (kernel time from HIP events, algorithmic bytes, fraction of the 8 TB/s HBM peak)."""
import json
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
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))


def t_us(fn, n=200, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def setup(ident):
    kin, tmpl = codegen.template_for(ident)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = tmpl.ee_link
    if tmpl.ee2_link >= 0:          # two-arm template: the second arm tracks its own target
        spec.ee2_link = tmpl.ee2_link
        Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    return kin, ops.ModelHandle(kin), ops.CostHandle(spec, dev)


def line(name, us, nbytes, n):
    return {"op": name, "us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 1),
            "hbm_frac": round(nbytes / us / 1e3 / 8e3, 3), "samples_per_s": n / us * 1e6}


out = {}
# ---- c4
kin, h, cm = setup("ur10_allegro")
B, H, D, L = 4096, 64, kin.n_dofs, kin.n_links
n = B * H
q = (torch.rand(B, H, D, device=dev) - 0.5) * 3.0
plan = ops.RolloutPlan(h, cm, (0, 1, 0, 1), q)
ee = kin.name_to_idx["ee_link"]
q2 = q.reshape(n, D)
out["c4"] = {"workload": f"UR10+Allegro ({L} links, {D} DOF), batch {B} x horizon {H}, fp32",
             "kernel": "specialized" if h.specialized else "table-driven",
             "ops": [line("fused FK + obstacle + EE cost + grad", t_us(plan.launch), (8 * D + 12 * L + 4) * n, n),
                     line("geometric Jacobian of ee_link (pos, quat, lin, ang)", t_us(lambda: ops.fk_jacobian(h, q2, None, ee), n=30, w=3),
                          (4 * D + 28 + 24 * D) * n, n)]}
# ---- c5
kin, h, cm = setup("dual_panda")
B, H, D, L = 2048, 128, kin.n_dofs, kin.n_links
n = B * H
qh = ((torch.rand(B, H, D, device=dev) - 0.5) * 3.0).half()
qdh = (torch.randn(B, H, D, device=dev) * 0.3).half()
plan = ops.RolloutPlan(h, cm, (0, 1, 0, 1), qh)
gqd = torch.zeros_like(qh)
dt, sigma = 5.0 / H, 0.1


gp_plan = ops.GPPriorPlan(qh, qdh, dt, sigma, 1.0, accumulate_into=(plan.gq, gqd))      # pre-bound, like the rollout


def gp():
    gp_plan.launch()


def both():
    plan.launch(); gp()


out["c5"] = {"workload": f"dual Panda ({L} links, {D} DOF), one rank's share {B} x {H} of the 4-way sharded 8192 x 128 batch, "
                         f"fp16 I/O, fp32 arithmetic + cost",
             "kernel": "specialized" if h.specialized else "table-driven",
             "ops": [line("fused FK + obstacle + EE cost + grad (fp16 I/O)", t_us(plan.launch), (4 * D + 6 * L + 4) * n, n),
                     line("GP prior cost + grad, accumulated (fp16 I/O)", t_us(gp), (2 * 2 * D + 2 * 2 * 2 * D) * n + 4 * B, n),
                     line("both (one objective evaluation)", t_us(both), (4 * D + 6 * L + 4 + 12 * D) * n, n)]}
plan32 = ops.RolloutPlan(h, cm, (0, 1, 0, 1), qh.float())
out["c5"]["ops"].append(line("fused rollout, fp32 I/O (for comparison)", t_us(plan32.launch), (8 * D + 12 * L + 4) * n, n))
print(json.dumps(out, indent=1))
