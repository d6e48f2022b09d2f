#!/usr/bin/env python3
"""Config 5's fused launch (trk_rollout_gp_cost_grad, dual Panda, fp16 I/O): the arm-per-lane kernel (k_rollout_gpa, round 6) against the
whole-robot-per-lane kernel (k_rollout_gpt) in ONE process -- TRK_GP_ARM_LANES=0 / 1 is read per launch -- on the same buffers:
 1. outputs compared element by element (cost, both gradients, positions) at the bench's size and at ragged sizes,
 2. alternating timings (HIP events around a run of pre-bound launches).
usage: tools/ab_c5_arm_lanes.py [reps]"""
import os
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
kin, tmpl = codegen.template_for("dual_panda")
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))
spec = CostModelSpec(n_links_in=kin.n_links)
spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
spec.obj_link_margin = (0.10 + 0.01 * np.arange(len(tmpl.obj_links))).astype(np.float32)       # a different margin per link and arm
spec.objects = [o.as_object() for o in env.obj_fixed_list]
spec.ws_min, spec.ws_max = np.asarray([-1.0, -1.0, -1.0], np.float32), np.asarray([1.0, 1.0, 1.0], np.float32)
spec.ee_link = tmpl.ee_link
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
spec.ee2_link = tmpl.ee2_link
Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
spec.validate()
h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, dev)
D, L = kin.n_dofs, kin.n_links
gen = torch.Generator(device=dev).manual_seed(3)


def trajectories(B, H):
    dt = 5.0 / H
    q = torch.cumsum(torch.randn(B, H, D, device=dev, generator=gen) * 0.02, 1) + (torch.rand(B, 1, D, device=dev, generator=gen) - 0.5) * 2.0
    qd = torch.zeros_like(q)
    if H > 1:
        qd[:, :-1] = (q[:, 1:] - q[:, :-1]) / dt
        qd[:, -1] = qd[:, -2] if H > 2 else 0.0
    return q, qd, dt


def run(which, plan, sums=None):
    os.environ["TRK_GP_ARM_LANES"] = "1" if which == "arm" else "0"
    plan.launch(None if sums is None else sums.data_ptr())
    torch.cuda.synchronize()
    return [None if t is None else t.clone() for t in (plan.link_pos, plan.cost, plan.gq, plan.gqd)] + [None if sums is None else sums.clone()]


def t_us(which, plan, n=400, w=40):
    os.environ["TRK_GP_ARM_LANES"] = "1" if which == "arm" else "0"
    for _ in range(w):
        plan.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


worst = {}
for (B, H, wts, io, gs, pos) in ((2048, 128, (0, 1, 0, 1), "f16", 2.0 ** -12, True), (2048, 128, (0, 1, 1, 1), "f32", 1.0, True),
                                 (37, 128, (0, 1, 0, 1), "f16", 2.0 ** -12, True), (5, 16, (0, 1, 1, 1), "f32", 1.0, True),
                                 (3, 7, (0, 1, 0, 1), "f16", 2.0 ** -10, True), (1, 1, (0, 1, 0, 1), "f32", 1.0, True),
                                 (129, 64, (0, 0, 0, 1), "f16", 2.0 ** -12, False), (64, 33, (0, 1, 1, 0), "g32", 1.0, True)):
    q, qd, dt = trajectories(B, H)
    if io == "f32":
        tq, tqd, gdt = q.contiguous(), qd.contiguous(), None
    else:
        tq, tqd, gdt = q.half().contiguous(), qd.half().contiguous(), (torch.float32 if io == "g32" else None)
    plan = ops.RolloutGpPlan(h, cm, wts, tq, tqd, dt, 0.1, 1.0, want_pos=pos, grad_dtype=gdt, grad_scale=gs)
    sums = torch.zeros(ops.n_blocks(B * H), device=dev)
    a, b = run("arm", plan, sums), run("robot", plan, sums)
    assert ops.last_dispatch() == "generated"
    names = ("positions", "cost", "gq", "gqd", "block sums")
    line = []
    for nm, x, y in zip(names, a, b):
        if x is None:
            continue
        x, y = x.double(), y.double()
        assert torch.isfinite(x).all(), (nm, B, H, io)
        err = float((x - y).abs().max()) / max(1e-30, float(y.abs().max()))
        line.append(f"{nm} {err:.1e}")
        worst[nm] = max(worst.get(nm, 0.0), err)
    print(f"B {B:5d} H {H:4d} weights {wts} io {io} pos {int(pos)}: max |arm-lane - robot-lane| / max |.|:  " + "  ".join(line), flush=True)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
q, qd, dt = trajectories(2048, 128)
for io, pos in (("f16", True), ("f16", False), ("g32", True), ("f32", True)):
    tq, tqd = (q.contiguous(), qd.contiguous()) if io == "f32" else (q.half().contiguous(), qd.half().contiguous())
    plan = ops.RolloutGpPlan(h, cm, (0, 1, 0, 1), tq, tqd, dt, 0.1, 1.0, want_pos=pos, grad_dtype=torch.float32 if io == "g32" else None,
                             grad_scale=2.0 ** -12 if io == "f16" else 1.0)
    for r in range(reps):
        print(f"2048 x 128 {io} positions {int(pos)} rep {r}:  robot-per-lane {t_us('robot', plan):6.2f} us   arm-per-lane {t_us('arm', plan):6.2f} us", flush=True)
# occupancy scan (DESIGN: "one wavefront's latency is the unit"): the kernel's time against the batch -- 256 trajectories x 128 = one
# 64-sample wavefront per SIMD of the robot-per-lane kernel = TWO 32-sample wavefronts per SIMD of the arm-per-lane kernel
print("occupancy scan, fp16, positions on (us per launch):")
for Bs in (64, 128, 256, 512, 1024, 2048, 4096):
    qs, qds, dts = trajectories(Bs, 128)
    plan = ops.RolloutGpPlan(h, cm, (0, 1, 0, 1), qs.half().contiguous(), qds.half().contiguous(), dts, 0.1, 1.0, want_pos=True, grad_scale=2.0 ** -12)
    print(f"  batch {Bs:5d} x 128 ({Bs * 128 // 64:5d} / {Bs * 128 // 32:5d} wavefronts):  robot-per-lane {t_us('robot', plan, 300, 30):6.2f}   arm-per-lane {t_us('arm', plan, 300, 30):6.2f}", flush=True)
print("worst relative differences:", {k: f"{v:.1e}" for k, v in worst.items()})
