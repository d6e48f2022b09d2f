"""Random robots through the model compiler: seeded random URDF trees (revolute / continuous / prismatic / fixed joints; axes +-x, +-y,
+-z, non-axis-aligned ones -- which the reference degenerates to +-z, rigid_body.py:162-168 -- and missing ones; joints with and
without limits; arbitrary origins; a file order that is or is not a pre-order walk), a random collision model on each (object links,
self pairs, tracked link, sphere and box scenes, hinge flags), and then: run-time generated kernels == table-driven kernels == fp64
oracle for the fused rollout (positions, cost, gradient), the boolean fields, the FK matrices and the positions' reverse mode.

The reference-pinned robots are in test_gpu_parity.py; this file is about the generator's corner cases (structural zeros, sign folds,
constant links, branch bookkeeping) on shapes nobody wrote by hand."""
import os

import numpy as np
import pytest
import torch

from helpers import grad_close, rel_err
from torch_robotics_amd.costmodel import CostModelSpec, box_prims, make_object, sphere_prims
from torch_robotics_amd.kinmodel import KinModel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL_H, TOL_C = 2e-6, 1e-5


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


@pytest.fixture(scope="module")
def ops():
    from torch_robotics_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def oracle_lib():
    from oracle import oracle as _o
    return _o


def random_urdf(path, rng, n_links, preorder):
    """a random tree; returns the text's link count.  Parents are earlier links (a valid tree); with preorder=False the LINK lines are
    shuffled (the joint list keeps parents before children), so the file order is not a walk order."""
    axes = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1", "0.574 0 0.819", "0.3 -0.4 0.5", None]
    parents = [None] + [int(rng.integers(max(0, i - 3), i)) for i in range(1, n_links)]
    joints = []
    for i in range(1, n_links):
        kind = rng.choice(["revolute", "revolute", "revolute", "continuous", "prismatic", "fixed", "fixed"])
        xyz = " ".join(f"{v:.4f}" for v in rng.uniform(-0.15, 0.15, 3))
        rpy = " ".join(f"{v:.4f}" for v in rng.choice([0.0, 0.0, 1.5707963, -1.5707963, 3.14159265, 0.3, -0.7], 3))
        lines = [f'  <joint name="j{i}" type="{kind}">', f'    <parent link="l{parents[i]}"/><child link="l{i}"/>',
                 f'    <origin xyz="{xyz}" rpy="{rpy}"/>']
        if kind != "fixed":
            ax = axes[int(rng.integers(0, len(axes)))]
            if ax is not None:
                lines.append(f'    <axis xyz="{ax}"/>')
            if kind == "continuous":
                pass
            elif kind == "prismatic" or rng.random() < 0.8:
                lo, hi = sorted(rng.uniform(-2.5, 2.5, 2)) if kind == "revolute" else sorted(rng.uniform(-0.2, 0.3, 2))
                lines.append(f'    <limit lower="{lo:.4f}" upper="{hi:.4f}" effort="1" velocity="1"/>')
        lines.append("  </joint>")
        joints.append("\n".join(lines))
    order = list(range(1, n_links))
    if not preorder:
        rng.shuffle(order)
    text = ['<?xml version="1.0"?>', '<robot name="fuzz">', '  <link name="l0"/>'] + [f'  <link name="l{i}"/>' for i in order] + joints + ["</robot>"]
    path.write_text("\n".join(text))


def random_cost_spec(m, rng, scene):
    L = m.n_links
    n_obj = int(rng.integers(1, min(L, 7)))
    obj = np.sort(rng.choice(np.arange(L), size=n_obj, replace=False)).astype(np.int32)
    spec = CostModelSpec(n_links_in=L)
    spec.obj_link_idx = obj
    spec.obj_link_margin = rng.uniform(0.03, 0.12, n_obj).astype(np.float32)
    objects = []
    if scene in ("spheres", "mixed"):
        r = np.full(6, 0.12, np.float32) if scene == "spheres" else rng.uniform(0.05, 0.2, 6).astype(np.float32)
        objects.append(make_object(sphere_prims(rng.uniform(-0.6, 0.6, (6, 3)).astype(np.float32), r)))
    if scene in ("boxes", "mixed"):
        c, sz = rng.uniform(-0.5, 0.5, (4, 3)).astype(np.float32), rng.uniform(0.1, 0.4, (4, 3)).astype(np.float32)
        qw = rng.standard_normal(4); qw /= np.linalg.norm(qw)
        from torch_robotics_amd.kinmodel import quat_wxyz_to_rot
        objects.append(make_object(box_prims(c, sz, rounded=bool(rng.integers(0, 2))), rng.uniform(-0.2, 0.2, 3).astype(np.float32),
                                   quat_wxyz_to_rot(qw.astype(np.float32))))
    spec.objects = objects
    if rng.random() < 0.7:
        spec.ws_min, spec.ws_max = np.float32([-0.8, -0.9, -0.7]), np.float32([0.9, 0.8, 1.0])
    if L >= 4 and rng.random() < 0.8:
        sl = np.sort(rng.choice(np.arange(L), size=min(L, int(rng.integers(2, 6))), replace=False))
        pairs = [(a, b) for a in range(len(sl)) for b in range(a) if rng.random() < 0.6][:6]
        if pairs:
            spec.self_link_idx = sl.astype(np.int32)
            spec.self_pairs = np.asarray(pairs, np.int32)
            spec.self_margin = rng.uniform(0.03, 0.08, len(pairs)).astype(np.float32)
    spec.ee_link = int(rng.integers(1, L))
    T = np.eye(4, dtype=np.float32); T[:3, 3] = rng.uniform(-0.4, 0.4, 3); spec.ee_target = T
    spec.ee_w_pos, spec.ee_w_rot, spec.ee_square = float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 2)), bool(rng.integers(0, 2))
    spec.clamp_fields = int(rng.integers(0, 8))
    spec.validate()
    return spec


def _robot_seeds():
    """eight draws keep the suite short (each is a run-time compile); TRK_FUZZ_ROBOT_SEEDS="8-40" adds more for a one-off soak"""
    seeds = list(range(8))
    extra = os.environ.get("TRK_FUZZ_ROBOT_SEEDS", "")
    if extra:
        a, b = (int(v) for v in extra.split("-"))
        seeds += list(range(a, b + 1))
    return seeds


@pytest.mark.parametrize("seed", _robot_seeds())
def test_random_robot_generated_vs_table_driven_vs_oracle(ops, oracle_lib, tmp_path, seed):
    from torch_robotics_amd import jit
    rng = np.random.default_rng(9000 + seed)
    n_links = int(rng.integers(4, 15))
    urdf = tmp_path / f"fuzz{seed}.urdf"
    random_urdf(urdf, rng, n_links, preorder=bool(seed % 3))
    m = KinModel.from_urdf(str(urdf))
    if m.n_dofs == 0:
        pytest.skip("the draw has no movable joint")
    scene = ("spheres", "boxes", "mixed")[seed % 3]
    spec = random_cost_spec(m, rng, scene)
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert jit.specialize_for_cost_spec(m, spec) is not None and h.specialized
    D, L = m.n_dofs, m.n_links
    weights = [(1, 1, 1, 1), (0, 1, 0, 1), (1, 0, 1, 0)][seed % 3]
    for n in (1, 64, 130):
        q = rng.uniform(-3.0, 3.0, size=(n, D)).astype(np.float32)              # beyond most limits: the clamps and their zero gradients
        rp, rc, rg = o.rollout(q.astype(np.float64), weights, "f64")
        scale = max(1.0, float(np.abs(rp).max()))
        H64 = o.fk(q.astype(np.float64), "f64")
        got = {}
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            pos, cost, gq = ops.rollout_cost_grad(h, cm, weights, dev(q))
            tag = (seed, n, use_spec)
            assert np.abs(pos.cpu().numpy() - rp).max() / scale < TOL_H, tag
            assert rel_err(cost.cpu().numpy(), rc) < TOL_C or np.abs(cost.cpu().numpy() - rc).max() < 1e-5, tag
            # kinks (arg-min ties between primitives, box faces, hinges at zero) may flip single samples: bound their number
            bad = ~np.isclose(gq.cpu().numpy(), rg, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(rg).max())).all(-1)
            assert bad.sum() <= max(1, n // 50), (tag, int(bad.sum()))
            Hm = ops.fk_forward(h, dev(q)).cpu().numpy()
            assert np.abs(Hm - H64).max() / scale < TOL_H, tag
            got[use_spec] = ops.rollout_collision(h, cm, 7, dev(q)).cpu().numpy()
        assert (got[True] != got[False]).sum() <= max(1, n // 60), seed          # a byte differs only within rounding of a margin
        h.enable_specialized(True)
        if n > 1:
            # round 6: rollout + geometric Jacobian of the tracked link behind ONE call.  Whether the unit serves it in one launch depends
            # on the tree (the generator demands that the Jacobian's stateful walk coincide with the rollout's on the link's chain: limits,
            # axes, joint types); either way the call must equal the two ops and the fp64 oracle.
            link = int(spec.ee_link)
            plan = ops.RolloutJacobianPlan(h, cm, weights, dev(q).reshape(1, n, D), link, strict=False)
            plan.lin_jac.fill_(7.0); plan.ang_jac.fill_(7.0)
            plan.launch()
            torch.cuda.synchronize()
            assert ops.last_dispatch() in ("generated", "generated + prior launches"), (seed, ops.last_dispatch())
            if os.environ.get("TRK_FUZZ_VERBOSE"):
                print(f"[fuzz] seed {seed} n {n}: rollout + Jacobian of link {link} served by: {ops.last_dispatch()}")
            jp, jq, jl, ja = o.jacobian(q.astype(np.float64), None, link, "f64")[:4]
            assert np.abs(plan.pos.reshape(n, 3).cpu().numpy() - jp).max() / scale < TOL_H, (seed, n)
            assert np.abs(plan.lin_jac.reshape(n, 3, D).cpu().numpy() - jl).max() / scale < 3 * TOL_H, (seed, n, ops.last_dispatch())
            assert np.abs(plan.ang_jac.reshape(n, 3, D).cpu().numpy() - ja).max() < 3 * TOL_H, (seed, n, ops.last_dispatch())
            qd_ = plan.quat.reshape(n, 4).cpu().numpy()
            assert np.minimum(np.abs(qd_ - jq).max(-1), np.abs(qd_ + jq).max(-1)).max() < 5e-6, (seed, n)
            assert np.abs(plan.link_pos.reshape(n, L, 3).cpu().numpy() - rp).max() / scale < TOL_H, (seed, n)
            assert rel_err(plan.cost.reshape(-1).cpu().numpy(), rc) < TOL_C or np.abs(plan.cost.reshape(-1).cpu().numpy() - rc).max() < 1e-5, (seed, n)
        if n > 1:
            # round 6: the analytic Jacobian of every link has a generated kernel too (prismatic joints, reversed axes, joints without
            # limits, pre-order != file order are what the bundled robots do not exercise): generated == table-driven == fp64 oracle,
            # except where fp32 and fp64 pick different quaternion candidates (a discontinuity of rotation_matrix_to_q)
            J64 = o.analytic_jacobian(q.astype(np.float64), "f64")
            for use_spec in (True, False):
                h.enable_specialized(use_spec)
                Jg = ops.fk_analytic_jacobian(h, dev(q)).cpu().numpy()
                sw = np.abs(Jg - J64).max(axis=(2, 3)) > 1e-4 * scale
                assert sw.mean() < 0.03, (seed, n, use_spec, float(sw.mean()))
                assert sw.all() or np.abs((Jg - J64)[~sw]).max() < 6e-6 * scale, (seed, n, use_spec)
            h.enable_specialized(True)
        w = rng.standard_normal((n, L, 3)).astype(np.float32)
        gH = np.zeros((n, L, 4, 4)); gH[..., :3, 3] = w
        ref_b = o.fk_backward(q.astype(np.float64), gH, "f64")
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            gb = ops.fk_positions_backward(h, dev(q), dev(w)).cpu().numpy()
            # (a draw whose gradient is zero everywhere -- one sample, every joint beyond its limit or turning about the only point it
            # carries -- leaves grad_close no scale: the wrench form's rounding, ~1e-8 |t| |f|, is then all there is; soak seeds 12, 20, 21, 31, 42)
            assert grad_close(gb, ref_b, scale=scale) or np.abs(gb - ref_b).max() < 1e-6 * scale, (seed, n, use_spec)
        h.enable_specialized(True)


def _point_set_draws():
    """four draws keep the suite short (each is a run-time compile); TRK_FUZZ_POINT_SEEDS="10-21" adds more for a one-off soak"""
    draws = [(0, "panda_arm_no_gripper"), (1, "dual_panda"), (2, "panda_arm_no_gripper"),
             (30, "iiwa7")]        # two paired points 1 mm apart on one link: what the first factorised pair accumulation (V - p S) got wrong
    extra = os.environ.get("TRK_FUZZ_POINT_SEEDS", "")
    if extra:
        a, b = (int(v) for v in extra.split("-"))
        draws += [(k, ("panda_arm_no_gripper", "ur10", "iiwa7", "panda_arm_no_gripper")[k % 4]) for k in range(a, b + 1)]
    return draws


@pytest.mark.parametrize("seed,robot", _point_set_draws())
def test_random_point_sets_generated_vs_table_driven_vs_oracle(ops, oracle_lib, seed, robot):
    """Attached-point units compiled at run time for RANDOM point sets (points per link 0 .. 5 with random offsets incl. exact zeros,
    walk-ordered columns) and random self pairs -- pairs between two points of ONE link, pairs whose earlier point sits on a fixed link,
    points paired many times (the factorised pair phase accumulates per point) --: generated == table-driven == fp64 oracle."""
    from helpers import model
    from torch_robotics_amd import jit
    rng = np.random.default_rng(7700 + seed)
    m = model(robot)
    pl, po = [], []
    for i in (int(v) for v in m.order):                       # walk order, origins first like link_sorted_point_set
        pl.append(i); po.append((0.0, 0.0, 0.0))
        for _ in range(int(rng.integers(0, 6 if seed != 1 else 3))):
            off = rng.uniform(-0.12, 0.12, 3) * (rng.random(3) < 0.7)      # some components exactly zero
            pl.append(i); po.append(tuple(off))
    pl, po = np.asarray(pl, np.int32), np.asarray(po, np.float32)
    P = len(pl)
    spec = CostModelSpec(n_links_in=P)
    obj = np.sort(rng.choice(np.arange(P), size=min(P, int(rng.integers(6, 20))), replace=False)).astype(np.int32)
    spec.obj_link_idx, spec.obj_link_margin = obj, rng.uniform(0.02, 0.1, len(obj)).astype(np.float32)
    spec.objects = [make_object(sphere_prims(rng.uniform(-0.7, 0.7, (8, 3)).astype(np.float32), np.full(8, 0.1, np.float32)))]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -0.5]), np.float32([1, 1, 1.5])
    sl = np.sort(rng.choice(np.arange(P), size=min(P, 12), replace=False))
    pairs = [(a, b) for a in range(len(sl)) for b in range(a) if rng.random() < 0.45][:30]
    # a pair inside one link, if the draw has a link with two selected points
    same = [(a, b) for a in range(len(sl)) for b in range(a) if pl[sl[a]] == pl[sl[b]]]
    pairs = (same[:2] + pairs)[:30]
    spec.self_link_idx, spec.self_pairs = sl.astype(np.int32), np.asarray(pairs, np.int32).reshape(-1, 2)
    spec.self_margin = rng.uniform(0.03, 0.08, len(pairs)).astype(np.float32)
    spec.ee_link = int(m.n_links - 1)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.3, 0.2, 0.6); spec.ee_target = T
    spec.clamp_fields = int(rng.integers(0, 8))
    spec.validate()
    h = ops.ModelHandle(m)
    ps0, cm = ops.PointSetHandle(h, pl, po, DEV), ops.CostHandle(spec, DEV)
    assert not ps0.specialized
    o = oracle_lib.Oracle(m, spec)
    q = rng.uniform(-2.8, 2.8, size=(130, m.n_dofs)).astype(np.float32)
    ref = {w: o.rollout_points(pl, po, q.astype(np.float64), w, "f64") for w in ((1, 1, 1, 1), (1, 0, 0, 0), (0, 1, 1, 0))}
    table = {w: [t.cpu().numpy() for t in ops.rollout_points_cost_grad(ps0, cm, w, dev(q))] for w in ref}
    assert jit.specialize_points(m, pl, po, spec) is not None
    ps = ops.PointSetHandle(h, pl, po, DEV)
    assert ps.specialized
    for w, (rp, rc, rg) in ref.items():
        pos, cost, gq = (t.cpu().numpy() for t in ops.rollout_points_cost_grad(ps, cm, w, dev(q)))
        for name, (p_, c_, g_) in (("generated", (pos, cost, gq)), ("table-driven", table[w])):
            assert np.abs(p_ - rp).max() < 2 * TOL_H, (seed, w, name)
            # a hinged cost is a small DIFFERENCE (margin - distance, both ~0.05): its error scales with those, not with what the hinge
            # leaves -- seeds whose self-only costs are all < 2e-3 otherwise fail on the 1e-7 a fp32 position is worth (soak seeds 15, 35, 40)
            assert np.abs(c_ - rc).max() < TOL_C * max(float(np.abs(rc).max()), 0.05), (seed, w, name)
            bad = ~np.isclose(g_, rg, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(rg).max())).all(-1)
            assert bad.sum() <= 2, (seed, w, name, int(bad.sum()))              # arg-min ties / hinges at zero
    wpt = rng.standard_normal((130, P, 3)).astype(np.float32)
    a = ops.fk_points_backward(ps, dev(q), dev(wpt)).cpu().numpy()
    b = ops.fk_points_backward(ps0, dev(q), dev(wpt)).cpu().numpy()
    assert grad_close(a, b)
