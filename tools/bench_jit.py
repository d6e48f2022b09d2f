#!/usr/bin/env python3
"""Table-driven vs run-time compiled fused rollout on robots without an ahead-of-time unit, 4096 x 64."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from helpers import model
from torch_robotics_amd import jit, ops
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.environments import EnvSpheres3D

dev = torch.device("cuda:0")
env = EnvSpheres3D(tensor_args=dict(device=dev, dtype=torch.float32))


def t(fn, n=100, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for robot in ("iiwa7", "ur10", "shadow_hand", "tiago_dual_holobase_minimal_holonomic", "hab_stretch", "iiwa7_allegro"):
    m = model(robot)
    leaves = [i for i in range(m.n_links) if not (m.parent == i).any()]
    obj = sorted(set(leaves[:5] + [m.n_links // 2, m.n_links // 3]))
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.asarray(obj, np.int32)
    spec.obj_link_margin = np.full(len(obj), 0.07, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = leaves[-1]
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.3, 0.1, 0.8); spec.ee_target = T
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, dev)
    B, H, D, L = 4096, 64, m.n_dofs, m.n_links
    q = (torch.rand(B, H, D, device=dev) - 0.5) * 2.0
    plan = ops.RolloutPlan(h, cm, (0, 1, 0, 1), q)
    tg = t(plan.launch, n=30, w=3)
    t0 = time.perf_counter(); jit.specialize_for_cost_spec(m, spec); tc = time.perf_counter() - t0
    ts = t(plan.launch)
    nbytes = (8 * D + 12 * L + 4) * B * H
    print(f"{robot:40s} L={L:2d} D={D:2d}  table-driven {tg:7.1f} us   run-time compiled {ts:6.1f} us ({nbytes / ts / 1e3 / 8e3 * 100:4.1f} % of 8 TB/s)"
          f"   x{tg / ts:4.1f}   (compile+load {tc:5.1f} s)")
