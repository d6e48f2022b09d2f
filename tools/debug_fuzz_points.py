#!/usr/bin/env python3
"""Debug aid for tests/test_gpu_fuzz_robots.py::test_random_point_sets_*: rebuild one draw, compare the generated and the table-driven kernels with
the fp64 oracle for the self-collision term alone, then bisect the pair list.   python tools/debug_fuzz_points.py <seed> <robot>"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
from helpers import model
from torch_robotics_amd import jit, ops
from torch_robotics_amd.costmodel import CostModelSpec, make_object, sphere_prims
from oracle import oracle as oracle_lib
oracle_lib.lib()

seed, robot = int(sys.argv[1]), sys.argv[2]
DEV = torch.device("cuda:0")
dev = lambda a: torch.as_tensor(a, device=DEV)
rng = np.random.default_rng(7700 + seed)
m = model(robot)
pl, po = [], []
for i in (int(v) for v in m.order):
    pl.append(i); po.append((0.0, 0.0, 0.0))
    for _ in range(int(rng.integers(0, 6 if seed != 1 else 3))):
        off = rng.uniform(-0.12, 0.12, 3) * (rng.random(3) < 0.7)
        pl.append(i); po.append(tuple(off))
pl, po = np.asarray(pl, np.int32), np.asarray(po, np.float32)
P = len(pl)
obj = np.sort(rng.choice(np.arange(P), size=min(P, int(rng.integers(6, 20))), replace=False)).astype(np.int32)
obj_mg = rng.uniform(0.02, 0.1, len(obj)).astype(np.float32)
centers = rng.uniform(-0.7, 0.7, (8, 3)).astype(np.float32)
sl = np.sort(rng.choice(np.arange(P), size=min(P, 12), replace=False))
pairs = [(a, b) for a in range(len(sl)) for b in range(a) if rng.random() < 0.45][:30]
same = [(a, b) for a in range(len(sl)) for b in range(a) if pl[sl[a]] == pl[sl[b]]]
pairs = (same[:2] + pairs)[:30]
margins = rng.uniform(0.03, 0.08, len(pairs)).astype(np.float32)
clamp = int(rng.integers(0, 8))
q = rng.uniform(-2.8, 2.8, size=(130, m.n_dofs)).astype(np.float32)
print("pairs", pairs, "clamp_fields", clamp, "cols", sl.tolist(), "links", pl[sl].tolist())


def run(sub):
    spec = CostModelSpec(n_links_in=P)
    spec.obj_link_idx, spec.obj_link_margin = obj, obj_mg
    spec.objects = [make_object(sphere_prims(centers, np.full(8, 0.1, np.float32)))]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -0.5]), np.float32([1, 1, 1.5])
    spec.self_link_idx = sl.astype(np.int32)
    spec.self_pairs = np.asarray([pairs[k] for k in sub], np.int32).reshape(-1, 2)
    spec.self_margin = margins[list(sub)]
    spec.ee_link = int(m.n_links - 1)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.3, 0.2, 0.6); spec.ee_target = T
    spec.clamp_fields = clamp
    spec.validate()
    h = ops.ModelHandle(m)
    cm = ops.CostHandle(spec, DEV)
    o = oracle_lib.Oracle(m, spec)
    w = (1, 0, 0, 0)
    rp, rc, rg = o.rollout_points(pl, po, q.astype(np.float64), w, "f64")
    ps0 = ops.PointSetHandle(h, pl, po, DEV)
    was = ps0.specialized
    out = {}
    if not was:
        out["table-driven"] = [t.cpu().numpy() for t in ops.rollout_points_cost_grad(ps0, cm, w, dev(q))]
    assert jit.specialize_points(m, pl, po, spec) is not None
    ps = ops.PointSetHandle(h, pl, po, DEV)
    out["generated"] = [t.cpu().numpy() for t in ops.rollout_points_cost_grad(ps, cm, w, dev(q))]
    res = {}
    for name, (p_, c_, g_) in out.items():
        bad = ~np.isclose(g_, rg, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(rg).max())).all(-1)
        res[name] = (int(bad.sum()), float(np.abs(g_ - rg).max()), float(np.abs(c_ - rc).max()), float(np.abs(rg).max()))
    return res


print("all pairs:", run(range(len(pairs))))
for k in range(len(pairs)):
    r = run([k])
    g = r["generated"]
    if g[0] > 2:
        print("pair", k, pairs[k], "cols", (int(sl[pairs[k][0]]), int(sl[pairs[k][1]])), "links", (int(pl[sl[pairs[k][0]]]), int(pl[sl[pairs[k][1]]])), "->", r)
print("done")
