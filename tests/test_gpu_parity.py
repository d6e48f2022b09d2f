"""GPU parity: the HIP kernels, called through the C ABI (ctypes -> libtrk.so), against
(a) golden vectors produced by the reference itself and (b) the fp64 CPU oracle on fresh seeded inputs.

Tolerances (fp32): |dH| <= 2e-6 * max(1,|t|), cost rel 1e-5, gradient rel 1e-4."""
import os

import numpy as np
import pytest
import torch

from helpers import ROBOTS, gold, grad_close, grad_close_kinks, kink_rows_ok, model, panda_cost_spec, rel_err
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS

pytestmark = pytest.mark.gpu

TOL_H, TOL_C, TOL_G = 2e-6, 1e-5, 1e-4
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from torch_robotics_amd import ops as _ops
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


@pytest.mark.parametrize("robot", ROBOTS)
def test_fk_forward_backward_vs_golden(ops, robot):
    g = gold(f"fk_{robot}")
    h = ops.ModelHandle(model(robot))
    for tag in ("in", "out"):
        Hg = g[f"H_{tag}"]
        scale = max(1.0, float(np.abs(Hg[..., :3, 3]).max()))
        H = ops.fk_forward(h, dev(g[f"q_{tag}"])).cpu().numpy()
        assert np.abs(H - Hg).max() / scale < TOL_H
        np.testing.assert_array_equal(H[..., 3, :], np.broadcast_to([0, 0, 0, 1], H[..., 3, :].shape))
        gq = ops.fk_backward(h, dev(g[f"q_{tag}"]), dev(g[f"w_{tag}"])).cpu().numpy()
        assert grad_close(gq, g[f"gq_{tag}"], scale=scale)
        if tag == "out":
            assert np.all(gq[g["gq_out"] == 0] == 0)      # clamp kills the gradient exactly
    # link subset, in the caller's order
    m = h.kin
    sel = [m.name_to_idx[str(s)] for s in g["sel_names"]]
    Hs = ops.fk_forward(h, dev(g["q_in"]), sel).cpu().numpy()
    assert np.abs(Hs - g["H_sel"]).max() / scale < TOL_H
    pos = ops.fk_positions(h, dev(g["q_in"]), sel).cpu().numpy()
    assert np.abs(pos - g["H_sel"][..., :3, 3]).max() / scale < TOL_H
    # backward restricted to the subset == full backward with zeros elsewhere
    w = g["w_in"][:, sel]
    wfull = np.zeros_like(g["w_in"]); wfull[:, sel] = w
    gq_a = ops.fk_backward(h, dev(g["q_in"]), dev(w), sel).cpu().numpy()
    gq_b = ops.fk_backward(h, dev(g["q_in"]), dev(wfull)).cpu().numpy()
    # (the full form may run a generated kernel, the subset the table-driven one: equal to fp32 rounding of terms of size |t| -- the
    # tolerance convention of DESIGN.md section 2: the floor scales with max(1, |t|); Tiago's base moves up to 100 m from the origin)
    np.testing.assert_allclose(gq_a, gq_b, rtol=1e-5, atol=1e-5 * scale)
    gq_p = ops.fk_positions_backward(h, dev(g["q_in"]), dev(w[..., :3, 3]), sel).cpu().numpy()
    wpos = np.zeros_like(wfull)
    for c, li in enumerate(sel):
        wpos[:, li, :3, 3] = w[:, c, :3, 3]
    gq_pb = ops.fk_backward(h, dev(g["q_in"]), dev(wpos)).cpu().numpy()
    np.testing.assert_allclose(gq_p, gq_pb, rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "ur10_allegro", "hab_stretch"])
def test_fk_vs_fp64_oracle_ragged(ops, oracle_lib, robot):
    """Fresh seeded inputs, batch sizes that are not multiples of the 64-lane wavefront."""
    m = model(robot)
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    rng = np.random.default_rng(7)
    for n in (1, 63, 65, 1000):
        q = rng.uniform(-3.2, 3.2, size=(n, m.n_dofs)).astype(np.float32)
        w = rng.standard_normal((n, m.n_links, 4, 4)).astype(np.float32)
        H64 = o.fk(q.astype(np.float64), "f64")
        scale = max(1.0, float(np.abs(H64[..., :3, 3]).max()))
        H = ops.fk_forward(h, dev(q)).cpu().numpy()
        assert np.abs(H - H64).max() / scale < TOL_H
        gq = ops.fk_backward(h, dev(q), dev(w)).cpu().numpy()
        assert grad_close(gq, o.fk_backward(q.astype(np.float64), w.astype(np.float64), "f64"))
    assert ops.fk_forward(h, torch.empty((0, m.n_dofs), device=DEV)).shape == (0, m.n_links, 4, 4)


def test_fk_autograd_matches_golden(ops):
    g = gold("fk_panda_arm_no_gripper")
    h = ops.ModelHandle(model("panda_arm_no_gripper"))
    q = dev(g["q_out"]).requires_grad_(True)
    H = ops.fk(h, q)
    (H * dev(g["w_out"])).sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq_out"])


def test_base_pose(ops, oracle_lib):
    m = model("panda_arm_no_gripper")
    m.set_base_pose([0.3, -0.2, 0.1, 0.9238795, 0.0, 0.0, 0.3826834])
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    q = np.random.default_rng(1).uniform(-2, 2, (17, 7)).astype(np.float32)
    assert np.abs(ops.fk_forward(h, dev(q)).cpu().numpy() - o.fk(q.astype(np.float64), "f64")).max() < TOL_H
    m.reset_base_pose()
    h.set_base_pose(m.base_R, m.base_t)
    o.refresh_model()
    assert np.abs(ops.fk_forward(h, dev(q)).cpu().numpy() - o.fk(q.astype(np.float64), "f64")).max() < TOL_H


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "ur10", "iiwa7"])
def test_stateful_fk_and_jacobian(ops, robot):
    g = gold(f"jac_{robot}")
    m = model(robot)
    h = ops.ModelHandle(m)
    for k, link in enumerate(g["links"]):
        pos, quat, lin, ang, vl, va = [t.cpu().numpy() for t in
                                       ops.fk_jacobian(h, dev(g["q"]), dev(g["qd"]), m.name_to_idx[str(link)], want_vel=True)]
        assert np.abs(pos - g[f"pos_{k}"]).max() < 2e-6
        assert np.abs(quat - g[f"quat_{k}"]).max() < 2e-6
        assert np.abs(lin - g[f"lin_{k}"]).max() < 3e-6
        assert np.abs(ang - g[f"ang_{k}"]).max() < 2e-6
        assert np.abs(vl - g[f"vel_lin_{k}"]).max() < 2e-6
        assert np.abs(va - g[f"vel_ang_{k}"]).max() < 2e-6


def test_jacobian_tree_vs_oracle(ops, oracle_lib):
    m = model("ur10_allegro")
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    rng = np.random.default_rng(3)
    q = rng.uniform(-1.5, 1.5, (70, m.n_dofs)).astype(np.float32)
    qd = rng.standard_normal((70, m.n_dofs)).astype(np.float32)
    for link in (m.n_links - 1, 12, 5):
        got = [t.cpu().numpy() for t in ops.fk_jacobian(h, dev(q), dev(qd), link, want_vel=True)]
        ref = o.jacobian(q.astype(np.float64), qd.astype(np.float64), link, "f64")
        for a, b in zip(got, ref):
            assert np.abs(a - b).max() < 5e-6 * max(1.0, float(np.abs(b).max()))


def test_rotmat_to_quat(ops):
    g = gold("quat")
    q = ops.rotmat_to_quat(dev(g["R"])).cpu().numpy()
    assert np.abs(q[:-4] - g["q_wxyz"][:-4]).max() < 2e-6
    err = np.minimum(np.abs(q - g["q_wxyz"]).max(-1), np.abs(q + g["q_wxyz"]).max(-1))
    assert err.max() < 2e-6
    H = np.tile(np.eye(4, dtype=np.float32), (len(g["R"]), 1, 1)); H[:, :3, :3] = g["R"]
    q2 = ops.rotmat_to_quat(dev(H)).cpu().numpy()
    np.testing.assert_array_equal(q, q2)


ENVS = ["spheres3d", "spheres3d_grid", "table_shelf", "maze_boxes3d", "spheres3d_extra"]


@pytest.mark.parametrize("env", ENVS)
def test_collision_fields_vs_golden(ops, env):
    robot, g = gold("panda_robot"), gold(f"cost_{env}")
    spec = panda_cost_spec(g, robot)
    cm = ops.CostHandle(spec, DEV)
    h = ops.ModelHandle(model("panda_arm_no_gripper"))
    q = dev(g["q"].reshape(-1, 7))
    pos = ops.fk_positions(h, q)
    assert np.abs(pos.cpu().numpy() - robot["fk_map_collision"].reshape(-1, 11, 3)).max() < TOL_H
    pos_g = dev(robot["fk_map_collision"].reshape(-1, 11, 3))
    for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS)):
        for use_unit in (True, False):           # the Panda unit's field kernel (the fused kernel's code on given positions) / table-driven
            cm.enable_specialized(use_unit)
            c, gp = ops.cost_fields(cm, fl, pos_g, want_grad=True)
            assert rel_err(c.cpu().numpy(), g[f"cost_{fname}"].reshape(-1)) < TOL_C, (fname, use_unit)
            assert grad_close(gp.cpu().numpy(), g[f"gpos_{fname}"].reshape(-1, 11, 3)), (fname, use_unit)
            gc = torch.linspace(0.5, 2.0, pos_g.shape[0], device=DEV)              # an upstream gradient per sample
            _, gp2 = ops.cost_fields(cm, fl, pos_g, want_grad=True, gcost=gc)
            assert rel_err(gp2.cpu().numpy(), (gp * gc[:, None, None]).cpu().numpy()) < 1e-6
        cm.enable_specialized(True)
        np.testing.assert_array_equal(ops.collision_fields(cm, fl, pos_g).cpu().numpy(), g[f"coll_{fname}"].reshape(-1))
        np.testing.assert_array_equal(ops.collision_fields(cm, fl, pos_g, margin=0.0).cpu().numpy(),
                                      g[f"coll0_{fname}"].reshape(-1))
        # autograd chain FK -> field -> sum, like the reference's cost.sum().backward()
        qq = q.clone().requires_grad_(True)
        cost = ops.cost_fields_ad(cm, fl, ops.fk_pos(h, qq))
        cost.sum().backward()
        assert grad_close(qq.grad.cpu().numpy(), g[f"gq_{fname}"].reshape(-1, 7)), fname
    allf = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
    c = ops.cost_fields(cm, allf, pos_g)
    assert rel_err(c.cpu().numpy(), g["cost_total"].reshape(-1)) < TOL_C
    np.testing.assert_array_equal(ops.collision_fields(cm, allf, pos_g).cpu().numpy(), g["coll_total"].reshape(-1))
    np.testing.assert_array_equal(ops.collision_fields(cm, allf, pos_g, margin=0.0).cpu().numpy(), g["coll0_total"].reshape(-1))
    # fused rollout == PlanningTask.compute_collision_cost + backward
    pos_r, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 0), q)
    assert np.abs(pos_r.cpu().numpy() - robot["fk_map_collision"].reshape(-1, 11, 3)).max() < TOL_H
    assert rel_err(cost.cpu().numpy(), g["cost_total"].reshape(-1)) < TOL_C
    assert grad_close(gq.cpu().numpy(), g["gq_total"].reshape(-1, 7))
    if "cost_extra" in g:
        cm2 = ops.CostHandle(panda_cost_spec(g, robot, which="extra"), DEV)
        c, gp = ops.cost_fields(cm2, FIELD_OBJECTS, pos_g, want_grad=True)
        assert rel_err(c.cpu().numpy(), g["cost_extra"].reshape(-1)) < TOL_C
        assert grad_close(gp.cpu().numpy(), g["gpos_extra"].reshape(-1, 11, 3))


@pytest.mark.parametrize("env", ENVS)
def test_fused_collision_vs_golden_and_two_step(ops, oracle_lib, env):
    """trk_rollout_collision (FK + boolean fields in one launch, one byte per sample) == the reference's booleans on the goldens,
    == table-driven FK followed by trk_collision_fields byte for byte on fresh ragged inputs, for every field mask, with the
    fields' own margins and with the margin=0 override of get_trajs_collision_and_free; generated kernel (identity and general
    base pose) and the scratch fallback."""
    robot, g = gold("panda_robot"), gold(f"cost_{env}")
    spec = panda_cost_spec(g, robot)
    m = model("panda_arm_no_gripper")
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    assert h.specialized
    qg = dev(g["q"])                                           # (8, 8, 7)
    allf = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
    for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS), ("total", allf)):
        got = ops.rollout_collision(h, cm, fl, qg)
        assert got.shape == (8, 8) and got.dtype == torch.bool
        np.testing.assert_array_equal(got.cpu().numpy(), g[f"coll_{fname}"].astype(bool), err_msg=fname)
        np.testing.assert_array_equal(ops.rollout_collision(h, cm, fl, qg, margin=0.0).cpu().numpy(), g[f"coll0_{fname}"].astype(bool))
    rng = np.random.default_rng(11)
    seen = set()
    for n in (1, 63, 65, 1000, 4133):
        q = dev(rng.uniform(-3.0, 3.9, (n, 7)).astype(np.float32))
        pos = ops.fk_positions(h, q)
        for fl in (FIELD_SELF, FIELD_OBJECTS, FIELD_WS, FIELD_OBJECTS | FIELD_WS, allf):
            for margin in (None, 0.0, 0.07):
                want = ops.collision_fields(cm, fl, pos, margin=margin).cpu().numpy().astype(bool)
                seen |= set(np.unique(want).tolist())
                h.enable_specialized(True)
                got = ops.rollout_collision(h, cm, fl, q, margin=margin).cpu().numpy()
                h.enable_specialized(False)                    # table-driven FK into the scratch + the field kernel
                got_fb = ops.rollout_collision(h, cm, fl, q, margin=margin).cpu().numpy()
                h.enable_specialized(True)
                # the generated FK differs from the table-driven one in the last ulp of a position, so a byte may differ only
                # where some signed distance sits within 1e-6 of its margin; the fallback is the two-step path itself
                np.testing.assert_array_equal(got_fb, want)
                if not np.array_equal(got, want):
                    bad = np.flatnonzero(got != want)
                    assert len(bad) <= max(1, n // 2000), (env, n, fl, margin, len(bad))
    assert seen == {False, True}                               # both outcomes occurred
    # general base pose (the _bg variant of the generated kernel)
    m.set_base_pose([0.1234, -0.2345, 0.0567, 0.9659258, 0.0, 0.0, 0.2588190])
    h.set_base_pose(m.base_R, m.base_t)
    q = dev(rng.uniform(-2.5, 2.5, (700, 7)).astype(np.float32))
    want = ops.collision_fields(cm, allf, ops.fk_positions(h, q)).cpu().numpy().astype(bool)
    got = ops.rollout_collision(h, cm, allf, q).cpu().numpy()
    assert (got != want).sum() <= 1
    assert ops.rollout_collision(h, cm, allf, torch.empty((0, 64, 7), device=DEV)).shape == (0, 64)
    with pytest.raises(ValueError):
        ops.rollout_collision(h, cm, allf, torch.zeros(4, 14, device=DEV))


@pytest.mark.parametrize("name", ["spheres3d", "table_shelf", "spheres3d_tight"])
def test_clamp_sdf_vs_golden_and_oracle(ops, oracle_lib, name):
    """clamp_sdf=True -- relu(margin - sdf) per link / pair (distance_fields.py:114-117): the field kernel, the generated and the
    table-driven fused kernels against the reference's clamped fields (goldens) and the fp64 oracle on ragged random inputs."""
    from helpers import clamp_cost_spec
    g, robot = gold("cost_clamp"), gold("panda_robot")
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    spec = clamp_cost_spec(name, ee_target=Ht)
    m = model("panda_arm_no_gripper")
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    pos_g = dev(robot["fk_map_collision"].reshape(-1, 11, 3))
    qg = dev(g["q"].reshape(-1, 7))
    for fname, fl, w in (("self", FIELD_SELF, (1, 0, 0, 0)), ("objects", FIELD_OBJECTS, (0, 1, 0, 0)), ("ws", FIELD_WS, (0, 0, 1, 0))):
        c, gp = ops.cost_fields(cm, fl, pos_g, want_grad=True)
        ref_c, ref_g = g[f"{name}_cost_{fname}"].reshape(-1), g[f"{name}_gpos_{fname}"].reshape(-1, 11, 3)
        assert np.abs(c.cpu().numpy() - ref_c).max() < 1e-5 * max(1.0, np.abs(ref_c).max()), fname
        assert np.abs(gp.cpu().numpy() - ref_g).max() < 1e-4 * max(1.0, np.abs(ref_g).max()), fname
        for spec_on in (True, False):
            h.enable_specialized(spec_on)
            _, c2, gq = ops.rollout_cost_grad(h, cm, w, qg)
            ref_q = g[f"{name}_gq_{fname}"].reshape(-1, 7)
            assert np.abs(c2.cpu().numpy() - ref_c).max() < 1e-5 * max(1.0, np.abs(ref_c).max()), (fname, spec_on)
            assert np.abs(gq.cpu().numpy() - ref_q).max() < 1e-4 * max(1.0, np.abs(ref_q).max()), (fname, spec_on)
    h.enable_specialized(True)
    _, c, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 0), qg)
    assert rel_err(c.cpu().numpy(), g[f"{name}_cost_total"].reshape(-1)) < TOL_C
    assert grad_close(gq.cpu().numpy(), g[f"{name}_gq_total"].reshape(-1, 7))
    assert (c >= 0).all()
    rng = np.random.default_rng(21)
    for n in (37, 64, 1000):
        q = rng.uniform(-3.0, 3.9, (n, 7)).astype(np.float32)
        for w in ((1, 1, 1, 1), (0.5, 2.0, 0.25, 0.0)):
            p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
            for spec_on in (True, False):
                h.enable_specialized(spec_on)
                pos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
                assert np.abs(pos.cpu().numpy() - p64).max() < TOL_H
                assert rel_err(c.cpu().numpy(), c64) < TOL_C, (n, w, spec_on)
                # a hinge switching on exactly at the fp32 / fp64 rounding boundary flips a whole gradient term: allow a few samples
                bad = (np.abs(gq.cpu().numpy() - g64).max(1) > TOL_G * max(1.0, np.abs(g64).max())).sum()
                assert bad <= max(1, n // 300), (n, w, spec_on, bad)
    h.enable_specialized(True)


def _voxel_centres(dims, lim):
    """torch.linspace per axis in fp32 (start + i*step in the first half, end - (n-1-i)*step in the second), grid_map_sdf.py:34-41."""
    axes = []
    for k in range(3):
        n, lo, hi = int(dims[k]), np.float32(lim[0][k]), np.float32(lim[1][k])
        step = np.float32((hi - lo) / np.float32(n - 1)) if n > 1 else np.float32(0)
        i = np.arange(n)
        axes.append(np.where(i < n // 2, lo + step * i.astype(np.float32), hi - step * (n - 1 - i).astype(np.float32)).astype(np.float32))
    return np.stack(np.meshgrid(*axes, indexing="ij"), -1)


_KINK_PROBES = 2e-6 * np.array([[0, 0, 0]] + [[sx, sy, sz] for sx in (-1, 0, 1) for sy in (-1, 0, 1) for sz in (-1, 0, 1)
                                            if (sx, sy, sz) != (0, 0, 0)], np.float64)


def test_grid_precompute_and_sdf_points(ops, oracle_lib):
    """SDF values everywhere to 2e-6; gradients to 1e-5 everywhere EXCEPT on kinks of the distance function, where fp32 and the
    reference may pick different branches -- and there the result must be the gradient of one of the tied branches (no
    unbounded outliers): for a sphere union, the unit vector from a sphere whose distance is within 2e-6 of the minimum; for box
    scenes, the fp64 oracle's gradient at some point of a 2e-6 neighbourhood (face / edge / corner switches, arg-min box ties)."""
    robot, g, ga = gold("panda_robot"), gold("cost_spheres3d_grid"), gold("cost_spheres3d")
    cm = ops.CostHandle(panda_cost_spec(ga, robot), DEV)
    sdf, grad = ops.grid_precompute(cm, g["grid_cmap_dim"], g["limits"][0], g["limits"][1])
    assert np.abs(sdf.cpu().numpy() - g["grid_sdf"]).max() < 2e-6
    grad = grad.cpu().numpy()
    diff = np.abs(grad - g["grid_grad"]).max(-1)
    assert (diff < 1e-5).mean() > 0.999
    out = np.argwhere(diff >= 1e-5)
    ctr = _voxel_centres(g["grid_cmap_dim"], g["limits"]).astype(np.float64)
    c, r = ga["fixed0_f0_centers"].astype(np.float64), ga["fixed0_f0_radii"].astype(np.float64)
    assert np.array_equal(ga["fixed0_pos"], np.zeros(3))        # the scene object sits at the origin, unrotated
    for ix, iy, iz in out:
        d = ctr[ix, iy, iz] - c
        dist = np.linalg.norm(d, axis=1)
        tied = np.flatnonzero(dist - r <= (dist - r).min() + 2e-6)
        assert len(tied) >= 2, "gradient outlier away from any arg-min tie"
        assert min(np.abs(grad[ix, iy, iz] - d[s] / dist[s]).max() for s in tied) < 1e-5
    for env in ("table_shelf", "maze_boxes3d"):
        ge = gold(f"cost_{env}")
        spec = panda_cost_spec(ge, robot)
        cmh, o = ops.CostHandle(spec, DEV), oracle_lib.Oracle(model("panda_arm_no_gripper"), spec)
        pts = np.random.default_rng(0).uniform(-1, 1.2, (20000, 3)).astype(np.float32)
        s, gr = ops.sdf_points(cmh, dev(pts), want_grad=True)
        s64, g64 = o.sdf_points(pts.astype(np.float64), "f64")
        gr = gr.cpu().numpy()
        assert np.abs(s.cpu().numpy() - s64).max() < 2e-6
        d = np.abs(gr - g64).max(-1)
        assert (d < 1e-5).mean() > 0.995
        for n, ob in np.argwhere(d >= 1e-5):
            _, gp = o.sdf_points(pts[n].astype(np.float64) + _KINK_PROBES, "f64")
            if np.abs(gp[:, ob] - gr[n, ob]).max(-1).min() < 1e-5:
                continue
            # not a branch switch: an ill-conditioned normal.  Just inside a rounded edge / corner region the normal is r / |r| with |r|
            # small, and the fp32 rounding of the point's offsets (~6e-8) turns it by 6e-8 / |r| (found by the seed soak: 1.7e-5 at
            # |r| = 3e-3).  The result must then lie inside the envelope of the fp64 gradients over the point's rounding neighbourhood.
            _, gn = o.sdf_points(pts[n].astype(np.float64) + 0.125 * _KINK_PROBES, "f64")
            lo, hi = gn[:, ob].min(0) - 1e-5, gn[:, ob].max(0) + 1e-5
            assert ((gr[n, ob] >= lo) & (gr[n, ob] <= hi)).all(), (env, n, ob, gr[n, ob], lo, hi)


def test_ee_cost_vs_golden(ops):
    g, robot, gs = gold("cost_ee"), gold("panda_robot"), gold("cost_spheres3d")
    h = ops.ModelHandle(model("panda_arm_no_gripper"))
    q = dev(g["q"].reshape(-1, 7))
    for k in range(4):
        target = g[f"target_{k}"]
        for sq in (True, False):
            for wp, wr in ((1.0, 1.0), (2.0, 0.5)):
                key = f"t{k}_sq{int(sq)}_w{wp}_{wr}"
                spec = panda_cost_spec(gs, robot, ee_target=target if target.ndim == 2 else np.eye(4, dtype=np.float32),
                                       ee_kw=dict(ee_w_pos=wp, ee_w_rot=wr, ee_square=sq))
                cm = ops.CostHandle(spec, DEV)
                qq = q.clone().requires_grad_(True)
                H = ops.fk(h, qq)
                cost = ops.ee_cost_ad(cm, H[:, -1], dev(target))
                assert rel_err(cost.detach().cpu().numpy(), g["cost_" + key]) < TOL_C, key
                cost.sum().backward()
                assert grad_close(qq.grad.cpu().numpy(), g["gq_" + key]), key
                _, gH = ops.ee_cost(cm, H.detach()[:, -1], dev(target), want_grad=True)
                assert grad_close(gH.cpu().numpy()[:, :3], g["gH_" + key][:, -1, :3]), key
                if target.ndim == 2:
                    _, c2, gq = ops.rollout_cost_grad(h, cm, (0, 0, 0, 1), q)
                    assert rel_err(c2.cpu().numpy(), g["cost_" + key]) < TOL_C, key
                    assert grad_close(gq.cpu().numpy(), g["gq_" + key]), key


def test_rollout_bench_shapes_vs_golden(ops):
    g, robot, gs = gold("rollout_panda"), gold("panda_robot"), gold("cost_spheres3d")
    cm = ops.CostHandle(panda_cost_spec(gs, robot, ee_target=g["target"]), DEV)
    h = ops.ModelHandle(model("panda_arm_no_gripper"))
    q = dev(g["q"])                                     # (6, 64, 7): one wavefront per trajectory
    csum = torch.zeros(ops.n_blocks(6 * 64), device=DEV)
    pos, c2, g2 = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), q, cost_sum=csum)
    assert pos.shape == (6, 64, 11, 3) and c2.shape == (6, 64) and g2.shape == (6, 64, 7)
    assert np.abs(pos.cpu().numpy() - g["pos"]).max() < TOL_H
    assert rel_err(c2.cpu().numpy(), g["cost_c2"]) < TOL_C
    assert grad_close(g2.cpu().numpy(), g["gq_c2"])
    # per-wavefront partial sums == per-trajectory costs at horizon 64; deterministic scalar via trk_reduce_sum
    np.testing.assert_allclose(csum.cpu().numpy(), g["cost_c2"].astype(np.float64).sum(1), rtol=2e-5)
    tot = ops.reduce_sum(csum)
    assert abs(tot.item() - g["cost_c2"].astype(np.float64).sum()) < 1e-4 * abs(g["cost_c2"]).sum()
    assert tot.item() == ops.reduce_sum(csum).item()
    _, c3, g3 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, want_pos=False)
    assert rel_err(c3.cpu().numpy(), g["cost_c3"]) < TOL_C
    assert grad_close(g3.cpu().numpy(), g["gq_c3"])
    # differentiable wrapper
    qq = q.clone().requires_grad_(True)
    cost, _ = ops.rollout_ad(h, cm, (1, 1, 1, 1), qq)
    cost.sum().backward()
    assert grad_close(qq.grad.cpu().numpy(), g["gq_c3"])


@pytest.mark.parametrize("robot", ["ur10_allegro", "dual_panda", "hab_stretch"])
def test_rollout_trees_vs_fp64_oracle(ops, oracle_lib, robot):
    """Branched / deep models (BASELINE configs 4, 5): fused kernel vs fp64 oracle, ragged N."""
    from torch_robotics_amd.costmodel import CostModelSpec
    m = model(robot)
    gs, rb = gold("cost_spheres3d_extra"), gold("panda_robot")
    base = panda_cost_spec(gs, rb)
    rng = np.random.default_rng(11)
    L = m.n_links
    spec = CostModelSpec(n_links_in=L)
    spec.obj_link_idx = np.sort(rng.choice(L, size=min(L, 9), replace=False)).astype(np.int32)
    spec.obj_link_margin = rng.uniform(0.02, 0.12, len(spec.obj_link_idx)).astype(np.float32)
    spec.objects = base.objects
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1.2])
    spec.self_link_idx = np.arange(0, L, 2, dtype=np.int32)
    ns = len(spec.self_link_idx)
    pairs = [(a, b) for a in range(ns) for b in range(a + 2, ns)][:40]
    spec.self_pairs = np.asarray(pairs, np.int32)
    spec.self_margin = rng.uniform(0.02, 0.08, len(pairs)).astype(np.float32)
    spec.ee_link = L - 1
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    spec.ee_target = Ht
    spec.validate()
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    for n in (5, 64, 200):
        q = rng.uniform(-2.5, 2.5, (n, m.n_dofs)).astype(np.float32)
        for w in ((1, 1, 1, 1), (0, 1, 0, 1), (0.5, 2.0, 0.0, 0.25)):
            pos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
            p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
            scale = max(1.0, float(np.abs(p64).max()))
            assert np.abs(pos.cpu().numpy() - p64).max() / scale < TOL_H
            assert rel_err(c.cpu().numpy(), c64) < TOL_C
            assert grad_close(gq.cpu().numpy(), g64)


def test_error_behaviour(ops):
    h = ops.ModelHandle(model("panda_arm_no_gripper"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.fk_forward(h, torch.zeros(4, 7))
    with pytest.raises(ValueError):
        ops.fk_forward(h, torch.zeros(4, 7, device=DEV), sel=[0, 0])
    with pytest.raises(ValueError):
        ops.fk_forward(h, torch.zeros(4, 7, device=DEV), sel=[99])


@pytest.mark.parametrize("env", ENVS)
def test_specialized_rollout_kernel(ops, oracle_lib, env):
    """The generated (model-specialised) Panda kernel == the table-driven kernel == the fp64 oracle,
    for every scene type, ragged sizes, all weight combinations, with and without a base pose."""
    robot, g = gold("panda_robot"), gold(f"cost_{env}")
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    spec = panda_cost_spec(g, robot, ee_target=Ht)
    m = model("panda_arm_no_gripper")
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert h.specialized, "no specialised kernel registered for the Panda tables (model hash mismatch?)"
    rng = np.random.default_rng(5)
    for n in (64, 448, 1000, 37):
        q = rng.uniform(-3.0, 3.9, (n, 7)).astype(np.float32)
        for w in ((1, 1, 1, 1), (0, 1, 0, 1), (1, 0, 0, 0), (0, 0, 1, 0), (0.5, 2.0, 0.25, 3.0)):
            h.enable_specialized(True)
            pos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
            h.enable_specialized(False)
            pos_g, c_g, gq_g = ops.rollout_cost_grad(h, cm, w, dev(q))
            p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
            assert np.abs(pos.cpu().numpy() - p64).max() < TOL_H
            assert rel_err(c.cpu().numpy(), c64) < TOL_C, (n, w)
            assert grad_close(gq.cpu().numpy(), g64), (n, w)
            assert rel_err(c.cpu().numpy(), c_g.cpu().numpy()) < TOL_C
            assert grad_close(gq.cpu().numpy(), gq_g.cpu().numpy())
    # golden check through the specialised path, incl. want_pos=False and the cost-sum atomic
    h.enable_specialized(True)
    csum = torch.zeros(ops.n_blocks(64), device=DEV)
    _, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 0), dev(g["q"].reshape(-1, 7)), want_pos=False, cost_sum=csum)
    assert rel_err(cost.cpu().numpy(), g["cost_total"].reshape(-1)) < TOL_C
    assert grad_close(gq.cpu().numpy(), g["gq_total"].reshape(-1, 7))
    assert abs(csum.sum().item() - float(g["cost_total"].astype(np.float64).sum())) < 1e-4 * float(np.abs(g["cost_total"]).sum())
    # base pose: the general-base variant of the generated kernel
    m.set_base_pose([0.1234, -0.2345, 0.0567, 0.9659258, 0.0, 0.0, 0.2588190])
    h.set_base_pose(m.base_R, m.base_t); o.refresh_model()
    q = rng.uniform(-2.5, 2.5, (130, 7)).astype(np.float32)
    pos, c, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), dev(q))
    p64, c64, g64 = o.rollout(q.astype(np.float64), (1, 1, 1, 1), "f64")
    assert np.abs(pos.cpu().numpy() - p64).max() < TOL_H
    assert rel_err(c.cpu().numpy(), c64) < TOL_C and grad_close(gq.cpu().numpy(), g64)
    # a cost model whose link sets differ from the baked template silently uses the table-driven kernel
    spec2 = panda_cost_spec(g, robot, ee_target=Ht)
    spec2.obj_link_idx = np.asarray([1, 4, 6], np.int32); spec2.obj_link_margin = np.float32([0.1, 0.1, 0.1])
    cm2, o2 = ops.CostHandle(spec2, DEV), oracle_lib.Oracle(m, spec2)
    _, c, gq = ops.rollout_cost_grad(h, cm2, (0, 1, 1, 0), dev(q))
    _, c64, g64 = o2.rollout(q.astype(np.float64), (0, 1, 1, 0), "f64")
    assert rel_err(c.cpu().numpy(), c64) < TOL_C and grad_close(gq.cpu().numpy(), g64)


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "panda_arm_hand", "allegro_hand"])
def test_analytic_jacobian_all_links(ops, oracle_lib, robot):
    """A16 vs the reference's autograd Jacobian (golden) and vs the fp64 oracle on ragged fresh inputs."""
    g = gold(f"ajac_{robot}")
    m = model(robot)
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    J = ops.fk_analytic_jacobian(h, dev(g["q"])).cpu().numpy()
    assert J.shape == g["J"].shape
    assert np.abs(J - g["J"]).max() < 3e-6
    q = np.random.default_rng(2).uniform(-2.5, 2.5, (131, m.n_dofs)).astype(np.float32)
    J = ops.fk_analytic_jacobian(h, dev(q)).cpu().numpy()
    J64 = o.analytic_jacobian(q.astype(np.float64), "f64")
    # quaternion candidate switches are discontinuous: compare where fp32 and fp64 pick the same one
    bad = np.abs(J - J64).max(axis=(2, 3)) > 1e-4
    assert bad.mean() < 0.01
    assert np.abs((J - J64)[~bad]).max() < 5e-6
    # round 6: the call above ran the GENERATED kernel (k_ajac: unrolled walk, the row staged through the ring in memory order); the
    # table-driven kernel is the same function -- also with a moved base (the other instantiation), ragged and multi-wavefront sizes
    assert h.specialized
    h.enable_specialized(False)
    Jt = ops.fk_analytic_jacobian(h, dev(q)).cpu().numpy()
    h.enable_specialized(True)
    sw = np.abs(J - Jt).max(axis=(2, 3)) > 1e-4
    assert sw.mean() < 0.01 and np.abs((J - Jt)[~sw]).max() < 5e-6
    c, s_ = np.cos(0.4), np.sin(0.4)
    m.base_R = np.asarray([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
    m.base_t = np.asarray([0.2, -0.1, 0.3], np.float32)
    hb, ob = ops.ModelHandle(m), oracle_lib.Oracle(m)
    hb.set_base_pose(m.base_R, m.base_t)
    for n in (1, 64, 200):
        qb = np.random.default_rng(3 + n).uniform(-2.5, 2.5, (n, m.n_dofs)).astype(np.float32)
        Jb = ops.fk_analytic_jacobian(hb, dev(qb)).cpu().numpy()
        Jb64 = ob.analytic_jacobian(qb.astype(np.float64), "f64")
        swb = np.abs(Jb - Jb64).max(axis=(2, 3)) > 1e-4
        assert swb.mean() < 0.02 and (swb.all() or np.abs((Jb - Jb64)[~swb]).max() < 5e-6), n


@pytest.mark.parametrize("ident,urdf", [("ur10_allegro", "ur10_allegro"), ("dual_panda", "dual_panda")])
def test_specialized_tree_kernels(ops, oracle_lib, ident, urdf):
    """Generated kernels for the tree robots of BASELINE configs 4 / 5, with their baked collision templates,
    against the table-driven kernel and the fp64 oracle."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd.costmodel import CostModelSpec
    kin, tmpl = codegen.template_for(ident)
    m = model(urdf)
    gs, rb = gold("cost_spheres3d_extra"), gold("panda_robot")
    base = panda_cost_spec(gs, rb)
    rng = np.random.default_rng(21)
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = rng.uniform(0.02, 0.12, len(tmpl.obj_links)).astype(np.float32)
    spec.objects = base.objects
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1.2, 1.2, 1.5])
    self_links = sorted({a for p in tmpl.self_pairs for a in p})
    spec.self_link_idx = np.asarray(self_links, np.int32)
    spec.self_pairs = np.asarray([(self_links.index(a), self_links.index(b)) for a, b in tmpl.self_pairs], np.int32)
    spec.self_margin = rng.uniform(0.02, 0.08, len(tmpl.self_pairs)).astype(np.float32)
    spec.ee_link = tmpl.ee_link
    if tmpl.ee2_link >= 0:          # two-arm template: the second arm tracks its own target
        spec.ee2_link = tmpl.ee2_link
        Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.5, 0.1, 0.6)
    spec.ee_target = Ht
    spec.validate()
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert h.specialized
    for n in (64, 300):
        q = rng.uniform(-2.5, 2.5, (n, m.n_dofs)).astype(np.float32)
        for w in ((1, 1, 1, 1), (0, 1, 0, 1), (0.5, 2.0, 0.25, 0.0)):
            h.enable_specialized(True)
            pos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
            h.enable_specialized(False)
            _, c_g, gq_g = ops.rollout_cost_grad(h, cm, w, dev(q))
            p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
            scale = max(1.0, float(np.abs(p64).max()))
            assert np.abs(pos.cpu().numpy() - p64).max() / scale < TOL_H
            assert rel_err(c.cpu().numpy(), c64) < TOL_C, (n, w)
            assert grad_close(gq.cpu().numpy(), g64), (n, w)
            assert grad_close(gq.cpu().numpy(), gq_g.cpu().numpy())


def test_ik_step_vs_reference_adam(ops, oracle_lib):
    """8f rank 2: the fused IK iteration == loss_fn_ik_per_q + ik_termination + torch.optim.Adam (reference goldens)."""
    g = gold("ik_panda")
    m = model("panda_arm_no_gripper")
    h = ops.ModelHandle(m)
    lo, hi = dev(g["lower"]), dev(g["upper"])
    for tag, Ht in (("per_sample", g["H_target"]), ("single", g["H_target"][0])):
        q = dev(g["q0"]).clone()
        mom, vel = torch.zeros_like(q), torch.zeros_like(q)
        loss = torch.empty(48, device=DEV); valid = torch.empty(48, device=DEV, dtype=torch.uint8)
        for it in range(5):
            ops.ik_step(h, 10, dev(Ht), lo, hi, q, mom, vel, it + 1, lr=1e-2, loss=loss, valid=valid)
            if it == 0:
                assert rel_err(loss.cpu().numpy(), g[f"loss0_{tag}"]) < TOL_C
                np.testing.assert_array_equal(valid.cpu().numpy().astype(bool), g[f"valid0_{tag}"])
            assert rel_err(loss.cpu().numpy(), g[f"err_steps_{tag}"][it]) < 2e-4
            assert np.abs(q.cpu().numpy() - g[f"q_steps_{tag}"][it]).max() < 2e-4
        # the persistent form: K iterations in one call == K calls (bit for bit), loss / valid from before the first update;
        # 40 iterations span two launches of at most 32
        for K in (5, 40):
            qa, qb = dev(g["q0"]).clone(), dev(g["q0"]).clone()
            ma, va, mb, vb = torch.zeros_like(qa), torch.zeros_like(qa), torch.zeros_like(qa), torch.zeros_like(qa)
            la = torch.empty(48, device=DEV); lb = torch.empty(48, device=DEV); l0 = None
            for it in range(K):
                ops.ik_step(h, 10, dev(Ht), lo, hi, qa, ma, va, it + 1, lr=1e-2, loss=la)
                l0 = la.clone() if it == 0 else l0
            ops.ik_steps(h, 10, dev(Ht), lo, hi, qb, mb, vb, 1, K, lr=1e-2, loss=lb)
            assert torch.equal(qa, qb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(l0, lb)
    # ragged size on a tree robot vs the fp64 oracle
    mt = model("ur10_allegro")
    ht, ot = ops.ModelHandle(mt), oracle_lib.Oracle(mt)
    rng = np.random.default_rng(8)
    q0 = rng.uniform(-1.5, 1.5, (77, mt.n_dofs)).astype(np.float32)
    Htg = ot.fk(rng.uniform(-1.0, 1.0, (77, mt.n_dofs)).astype(np.float64), "f64")[:, 14].astype(np.float32)
    lo_t = np.full(mt.n_dofs, -1.2, np.float32); hi_t = np.full(mt.n_dofs, 1.2, np.float32)
    q = dev(q0).clone(); mom, vel = torch.zeros_like(q), torch.zeros_like(q)
    q64 = q0.astype(np.float64); m64, v64 = np.zeros_like(q64), np.zeros_like(q64)
    loss = torch.empty(77, device=DEV)
    for it in range(3):
        ops.ik_step(ht, 14, dev(Htg), dev(lo_t), dev(hi_t), q, mom, vel, it + 1, lr=5e-3, loss=loss)
        l64, _, _ = ot.ik_step(14, Htg.astype(np.float64), lo_t, hi_t, q64, m64, v64, it + 1, lr=5e-3, prec="f64")
        assert rel_err(loss.cpu().numpy(), l64) < 1e-4
        assert np.abs(q.cpu().numpy() - q64).max() < 2e-4


def test_gauss_newton_ik_steps_vs_fp64_oracle(ops, oracle_lib):
    """trk_ik_gn_steps (build-defined): every iteration == the fp64 oracle's damped Gauss-Newton step (stateful FK + the reference's
    geometric Jacobian, pose residual, J^T J + lambda I, Cholesky, clamped step) on the SAME configurations; K iterations in one
    launch == K launches, bit for bit; err / valid are ik_termination's metric for q as passed in; the reference's IK problem
    (tests/golden/ik_panda.npz: its q0, targets and shrunk limits) converges."""
    g = gold("ik_panda")
    m = model("panda_arm_no_gripper")
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    ee = m.name_to_idx["ee_link"]
    lo, hi = g["lower"], g["upper"]
    for tag, Ht in (("per_sample", g["H_target"]), ("single", g["H_target"][0])):
        q = dev(g["q0"]).clone()
        err = torch.empty(48, device=DEV); valid = torch.empty(48, device=DEV, dtype=torch.uint8)
        for it in range(6):
            q_in = q.cpu().numpy().astype(np.float64)
            q64, e64 = o.ik_gn_step(ee, np.asarray(Ht, np.float64), lo, hi, q_in, 1e-4, 0.1, 1.0, "f64")
            ops.ik_gn_steps(h, ee, dev(Ht), dev(lo), dev(hi), q, 1, err=err, valid=valid)
            dq = np.abs(q64 - q_in)
            # the step solves a system of condition <= (|J|^2 + lambda) / lambda in fp32: relative to the step, plus rounding of q
            assert (np.abs(q.cpu().numpy() - q64) <= 1e-4 + 5e-3 * dq).all(), (tag, it)
            assert rel_err(err.cpu().numpy(), e64) < 2e-5
            inside = ((q_in >= lo) & (q_in <= hi)).all(-1)
            sure = np.abs(e64 - 0.1) > 1e-5
            np.testing.assert_array_equal(valid.cpu().numpy().astype(bool)[sure], ((e64 < 0.1) & inside)[sure])
        for K in (3, 17):
            qa, qb = dev(g["q0"]).clone(), dev(g["q0"]).clone()
            ea, eb = torch.empty(48, device=DEV), torch.empty(48, device=DEV)
            e0 = None
            for it in range(K):
                ops.ik_gn_steps(h, ee, dev(Ht), dev(lo), dev(hi), qa, 1, err=ea)
                e0 = ea.clone() if it == 0 else e0
            ops.ik_gn_steps(h, ee, dev(Ht), dev(lo), dev(hi), qb, K, err=eb)
            assert torch.equal(qa, qb) and torch.equal(e0, eb)
    # convergence on the reference's problem: 40 iterations, then the termination metric of the result
    q = dev(g["q0"]).clone()
    err = torch.empty(48, device=DEV); valid = torch.empty(48, device=DEV, dtype=torch.uint8)
    ops.ik_gn_steps(h, ee, dev(g["H_target"]), dev(lo), dev(hi), q, 40)
    ops.ik_gn_steps(h, ee, dev(g["H_target"]), dev(lo), dev(hi), q.clone(), 1, err=err, valid=valid)
    Hq = o.fk(q.cpu().numpy().astype(np.float64), "f64")[:, ee]
    e_chk = np.array([1 - (np.trace(Hq[i, :3, :3].T @ g["H_target"][i, :3, :3].astype(np.float64)) - 1) / 2 +
                      np.linalg.norm(Hq[i, :3, 3] - g["H_target"][i, :3, 3]) for i in range(48)])
    assert rel_err(err.cpu().numpy(), e_chk) < 1e-4
    # random targets inside shrunk limits: about half of the 48 problems are solved from their q0 by a clamped LM iteration (the
    # others sit on a joint limit); the fp64 oracle iterated the same way says how many -- fp32 may differ on a few borderline ones
    q64 = g["q0"].astype(np.float64)
    for it in range(40):
        q64, _ = o.ik_gn_step(ee, g["H_target"].astype(np.float64), lo, hi, q64, 1e-4, 0.1, 1.0, "f64")
    _, e64 = o.ik_gn_step(ee, g["H_target"].astype(np.float64), lo, hi, q64, 1e-4, 0.1, 1.0, "f64")
    n64 = int((e64 < 0.1).sum())
    assert n64 >= 20 and abs(int(valid.sum()) - n64) <= 4 and int((e_chk < 1e-3).sum()) >= int((e64 < 1e-3).sum()) - 4
    # ragged sizes, random targets, other damping parameters, a base pose
    rng = np.random.default_rng(31)
    m.set_base_pose(np.array([0.1, -0.2, 0.05, 0.9238795, 0.0, 0.3826834, 0.0], np.float32))
    h2, o2 = ops.ModelHandle(m), oracle_lib.Oracle(m)
    for n in (1, 63, 65, 500):
        q0 = (lo + rng.random((n, 7)) * (hi - lo)).astype(np.float32)
        Ht = o2.fk((lo + rng.random((n, 7)) * (hi - lo)).astype(np.float64), "f64")[:, ee].astype(np.float32)
        q64, e64 = o2.ik_gn_step(ee, Ht.astype(np.float64), lo, hi, q0.astype(np.float64), 1e-3, 0.02, 0.7, "f64")
        q = dev(q0).clone(); err = torch.empty(n, device=DEV)
        ops.ik_gn_steps(h2, ee, dev(Ht), dev(lo), dev(hi), q, 1, damping=1e-3, lm_gain=0.02, step_scale=0.7, err=err)
        assert (np.abs(q.cpu().numpy() - q64) <= 1e-4 + 5e-3 * np.abs(q64 - q0)).all() and rel_err(err.cpu().numpy(), e64) < 2e-5
    m.set_base_pose(np.array([0, 0, 0, 1, 0, 0, 0], np.float32))
    # a link no unit tracks: refused loudly, never a silent other path
    with pytest.raises(Exception, match="no generated unit"):
        ops.ik_gn_steps(h, 5, dev(g["H_target"]), dev(lo), dev(hi), dev(g["q0"]).clone(), 1)


# ---------------------------------------------------------------------------------------------------------------------
# points fixed in link frames (grasped-object points, per-link spheres)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("robot", ["ur10_allegro", "dual_panda", "hab_stretch"])
def test_attached_points_vs_golden(ops, robot):
    g = gold(f"points_{robot}")
    h = ops.ModelHandle(model(robot))
    ps = ops.PointSetHandle(h, g["point_link"], g["point_offset"], DEV)
    pos = ops.fk_points(ps, dev(g["q"])).cpu().numpy()
    assert np.abs(pos - g["pos"]).max() / max(1.0, float(np.abs(g["pos"]).max())) < TOL_H
    gq = ops.fk_points_backward(ps, dev(g["q"]), dev(g["w"])).cpu().numpy()
    assert grad_close(gq, g["gq"])
    # autograd wrapper
    q = dev(g["q"]).requires_grad_(True)
    (ops.fk_points_ad(ps, q) * dev(g["w"])).sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq"])


def test_attached_points_vs_fp64_oracle_ragged(ops, oracle_lib):
    m = model("ur10_allegro")
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m)
    rng = np.random.default_rng(11)
    for n, P in ((1, 1), (63, 7), (65, 64), (700, 100)):
        pl = rng.integers(0, m.n_links, size=P).astype(np.int32)
        po = rng.uniform(-0.3, 0.3, size=(P, 3)).astype(np.float32)
        ps = ops.PointSetHandle(h, pl, po, DEV)
        q = rng.uniform(-3.2, 3.2, size=(n, m.n_dofs)).astype(np.float32)
        w = rng.standard_normal((n, P, 3)).astype(np.float32)
        ref = o.fk_points(pl, po, q.astype(np.float64), "f64")
        pos = ops.fk_points(ps, dev(q)).cpu().numpy()
        assert np.abs(pos - ref).max() / max(1.0, float(np.abs(ref).max())) < TOL_H
        gq = ops.fk_points_backward(ps, dev(q), dev(w)).cpu().numpy()
        assert grad_close(gq, o.fk_points_backward(pl, po, q.astype(np.float64), w.astype(np.float64), "f64"))
    with pytest.raises(NotImplementedError):
        ops.PointSetHandle(h, np.zeros(500, np.int32), np.zeros((500, 3), np.float32), DEV)
    with pytest.raises(ValueError):
        ops.PointSetHandle(h, np.array([m.n_links], np.int32), np.zeros((1, 3), np.float32), DEV)
    from torch_robotics_amd._lib import check, lib
    other = ops.ModelHandle(model("ur10"))
    ps = ops.PointSetHandle(h, np.zeros(1, np.int32), np.zeros((1, 3), np.float32), DEV)
    out, qq = torch.empty((1, 1, 3), device=DEV), torch.zeros((1, other.n_dofs), device=DEV)
    with pytest.raises(ValueError):       # a point set belongs to the model it was built for
        check(lib().trk_fk_points(other._h, ps._h, qq.data_ptr(), 1, out.data_ptr(), None), "trk_fk_points")


def test_grasped_object_vs_golden(ops):
    """fk_map_collision of RobotPanda + GraspedObjectPandaBox, the three fields on links + grasped points, fused."""
    from helpers import grasp_panda_setup
    g = gold("grasp_panda")
    m, pl, po, spec = grasp_panda_setup()
    h = ops.ModelHandle(m)
    ps = ops.PointSetHandle(h, pl, po, DEV)
    for tag in ("", "_out"):
        q = g["q" + tag].reshape(-1, 7)
        pos = ops.fk_points(ps, dev(q)).cpu().numpy()
        assert np.abs(pos - g["link_pos" + tag].reshape(-1, 26, 3)).max() < TOL_H
        gq = ops.fk_points_backward(ps, dev(q), dev(g["w" + tag].reshape(-1, 26, 3))).cpu().numpy()
        assert grad_close(gq, g["gq" + tag].reshape(-1, 7))
    cm = ops.CostHandle(spec, DEV)
    lp = dev(g["link_pos"].reshape(-1, 26, 3))
    total_g = np.zeros((32, 26, 3), np.float32)
    for key, f in (("self", FIELD_SELF), ("obj", FIELD_OBJECTS), ("ws", FIELD_WS)):
        c, gl = ops.cost_fields(cm, f, lp, want_grad=True)
        assert rel_err(c.cpu().numpy(), g[f"cost_{key}"].reshape(-1)) < TOL_C
        total_g += gl.cpu().numpy()
        np.testing.assert_array_equal(ops.collision_fields(cm, f, lp).cpu().numpy(), g[f"coll_{key}"].reshape(-1))
        np.testing.assert_array_equal(ops.collision_fields(cm, f, lp, 0.0).cpu().numpy(), g[f"coll0_{key}"].reshape(-1))
    assert grad_close(total_g, g["g_link_pos"].reshape(-1, 26, 3))
    pos, cost, gq = ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 0), dev(g["q"]))
    assert pos.shape == (4, 8, 26, 3) and cost.shape == (4, 8) and gq.shape == (4, 8, 7)
    assert np.abs(pos.cpu().numpy() - g["link_pos"]).max() < TOL_H
    assert rel_err(cost.cpu().numpy(), g["cost_self"] + g["cost_obj"] + g["cost_ws"]) < TOL_C
    assert grad_close(gq.cpu().numpy(), g["gq_cost"])


def test_rollout_points_vs_fp64_oracle(ops, oracle_lib):
    """Fused rollout over attached points on a tree robot: random spheres per link + EE tracking, ragged sizes."""
    from torch_robotics_amd.costmodel import CostModelSpec, make_object, sphere_prims
    m = model("dual_panda")
    rng = np.random.default_rng(5)
    P = 60
    pl = rng.integers(1, m.n_links, size=P).astype(np.int32)
    po = rng.uniform(-0.1, 0.1, size=(P, 3)).astype(np.float32)
    spec = CostModelSpec(n_links_in=P)
    spec.obj_link_idx = np.arange(P, dtype=np.int32)
    spec.obj_link_margin = rng.uniform(0.02, 0.1, size=P).astype(np.float32)
    spec.objects = [make_object(sphere_prims(rng.uniform(-0.8, 0.8, size=(12, 3)), rng.uniform(0.05, 0.2, size=12)))]
    spec.ws_min, spec.ws_max = np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32)
    spec.self_link_idx = np.arange(0, P, 3, dtype=np.int32)
    spec.self_pairs = np.array([(i, j) for i in range(20) for j in range(i) if (i + j) % 5 == 0], np.int32)
    spec.self_margin = np.full(len(spec.self_pairs), 0.05, np.float32)
    spec.ee_link = m.name_to_idx["left_ee_link"]
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.3, 0.5)
    spec.ee_target = T
    spec.validate()
    h, o = ops.ModelHandle(m), oracle_lib.Oracle(m, spec)
    ps, cm = ops.PointSetHandle(h, pl, po, DEV), ops.CostHandle(spec, DEV)
    for n in (1, 65, 500):
        q = rng.uniform(-2.5, 2.5, size=(n, m.n_dofs)).astype(np.float32)
        for wts in ((1, 1, 1, 1), (0, 1, 0, 0), (0.5, 0, 2, 1)):
            rp, rc, rg = o.rollout_points(pl, po, q.astype(np.float64), wts, "f64")
            pos, cost, gq = ops.rollout_points_cost_grad(ps, cm, wts, dev(q))
            assert np.abs(pos.cpu().numpy() - rp).max() < 2 * TOL_H
            assert rel_err(cost.cpu().numpy(), rc) < TOL_C
            assert grad_close(gq.cpu().numpy(), rg)


def test_gp_prior_vs_fp64_oracle(ops, oracle_lib):
    """Constant-velocity GP prior (build-defined, BASELINE config 5): fp32 and fp16 I/O against the fp64 oracle."""
    rng = np.random.default_rng(22)
    for (B, H, D, dt, sigma, w) in ((5, 64, 7, 0.08, 0.1, 1.0), (3, 128, 14, 5.0 / 128, 0.5, 0.3), (2, 1, 5, 0.1, 0.2, 1.0),
                                    (300, 37, 22, 0.05, 0.3, 1.0), (6, 128, 14, 5.0 / 128, 0.1, 1.0)):
        q = np.cumsum(rng.standard_normal((B, H, D)) * 0.05, axis=1).astype(np.float32)
        qd = (rng.standard_normal((B, H, D)) * 0.3).astype(np.float32)
        rc, rgq, rgqd = oracle_lib.gp_prior(q.astype(np.float64), qd.astype(np.float64), dt, sigma, w, "f64")
        c, gq, gqd = ops.gp_prior_cost_grad(dev(q), dev(qd), dt, sigma, w)
        assert c.dtype == torch.float32 and c.shape == (B,)
        if H > 1:
            assert rel_err(c.cpu().numpy(), rc) < 2e-5
            assert rel_err(gq.cpu().numpy(), rgq) < 1e-4 and rel_err(gqd.cpu().numpy(), rgqd) < 1e-4
        else:
            assert not c.any() and not gq.any() and not gqd.any()
        # accumulate into existing gradients
        g0, g1 = dev(rng.standard_normal((B, H, D)).astype(np.float32)), dev(rng.standard_normal((B, H, D)).astype(np.float32))
        a0, a1 = g0.clone(), g1.clone()
        ops.gp_prior_cost_grad(dev(q), dev(qd), dt, sigma, w, accumulate_into=(a0, a1))
        np.testing.assert_allclose((a0 - g0).cpu().numpy(), gq.cpu().numpy(), rtol=0, atol=1e-4 * max(1.0, float(gq.abs().max())))
        # fp16 I/O: the oracle sees the same fp16-rounded inputs.  The gradient leaves as grad_scale x d cost (product in fp32),
        # rounded to fp16 once; the scale from the bounds of the trajectories keeps EVERY element finite (config 5's own
        # parameters sigma = 0.1, dt = 5 / 128 are among the cases: a = 12 / (sigma^2 dt^3) = 2e7)
        qh, qdh = dev(q).half(), dev(qd).half()
        rc, rgq, rgqd = oracle_lib.gp_prior(qh.cpu().numpy().astype(np.float64), qdh.cpu().numpy().astype(np.float64), dt, sigma, w, "f64")
        gs = ops.gp_grad_scale(dt, sigma, w, float(qh.abs().max()), float(qdh.abs().max()))
        assert gs <= 1.0 and np.log2(gs) == np.round(np.log2(gs))
        c, gq, gqd = ops.gp_prior_cost_grad(qh, qdh, dt, sigma, w, grad_scale=gs)
        assert gq.dtype == torch.float16 and gqd.dtype == torch.float16 and c.dtype == torch.float32
        c32, gq32, gqd32 = ops.gp_prior_cost_grad(qh, qdh, dt, sigma, w, grad_dtype=torch.float32)      # mixed mode, unscaled
        assert gq32.dtype == torch.float32 and torch.equal(c32, c)
        if H > 1:
            assert rel_err(c.cpu().numpy(), rc) < 2e-5
            assert torch.isfinite(gq.float()).all() and torch.isfinite(gqd.float()).all()          # 100 % finite, no mask
            assert np.abs(rgq).max() * gs < 65504 and np.abs(rgqd).max() * gs < 65504
            for got, ref in ((gq, rgq), (gqd, rgqd)):
                # one fp16 rounding of the scaled value (2^-11 relative; 2^-25 absolute in the subnormal range) + fp32 arithmetic
                err = np.abs(got.float().cpu().numpy().astype(np.float64) / gs - ref)
                assert (err <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -24 / gs + 2e-5 * np.abs(ref).max()).all()
            assert rel_err(gq32.cpu().numpy(), rgq) < 1e-4 and rel_err(gqd32.cpu().numpy(), rgqd) < 1e-4
            # accumulate into a scaled fp16 gradient: one more fp16 rounding
            base16 = (torch.randn(B, H, D, device=DEV) * 3.0 * gs).half()
            a0, a1 = base16.clone(), torch.zeros_like(base16)
            ops.gp_prior_cost_grad(qh, qdh, dt, sigma, w, accumulate_into=(a0, a1), grad_scale=gs)
            ref = base16.double().cpu().numpy() / gs + rgq
            err = np.abs(a0.double().cpu().numpy() / gs - ref)
            assert torch.isfinite(a0.float()).all() and (err <= 2.0 ** -9 * (np.abs(ref) + np.abs(rgq)) + 2.0 ** -23 / gs + 4e-5 * np.abs(rgq).max()).all()
            # no scale: whatever exceeds the fp16 range SATURATES at +-65504 (never inf), everything else is the rounded value
            _, gsat, gdsat = ops.gp_prior_cost_grad(qh, qdh, dt, sigma, w)
            for got, ref in ((gsat, rgq), (gdsat, rgqd)):
                g = got.float().cpu().numpy().astype(np.float64)
                assert np.isfinite(g).all()
                big = np.abs(ref) > 65504.0 * (1 + 1e-4)
                assert (g[big] == np.sign(ref[big]) * 65504.0).all()
                small = np.abs(ref) < 65504.0 * (1 - 1e-3)
                assert (np.abs(g[small] - ref[small]) <= 2.0 ** -10 * np.abs(ref[small]) + 2.0 ** -24 + 2e-5 * np.abs(ref).max()).all()
    # autograd wrapper
    q = dev(rng.standard_normal((4, 16, 7)).astype(np.float32)).requires_grad_(True)
    qd = dev(rng.standard_normal((4, 16, 7)).astype(np.float32)).requires_grad_(True)
    wv = dev(np.array([1.0, 2.0, 0.5, 0.0], np.float32))
    (ops.gp_prior_cost(q, qd, 0.1, 0.4) * wv).sum().backward()
    _, rgq, rgqd = oracle_lib.gp_prior(q.detach().cpu().numpy().astype(np.float64), qd.detach().cpu().numpy().astype(np.float64), 0.1, 0.4, 1.0)
    assert rel_err(q.grad.cpu().numpy(), rgq * wv.cpu().numpy()[:, None, None]) < 1e-4
    with pytest.raises(ValueError):
        ops.gp_prior_cost_grad(dev(np.zeros((2, 3), np.float32)), dev(np.zeros((2, 3), np.float32)), 0.1, 0.1)


@pytest.mark.parametrize("robot,ident", [("panda_arm_no_gripper", "panda"), ("dual_panda", "dual_panda"), ("ur10_allegro", "ur10_allegro")])
def test_rollout_fp16_io(ops, oracle_lib, robot, ident):
    """fp16 q / link_pos / gq in HBM, fp32 arithmetic and cost (BASELINE config 5; build-defined): specialised and
    table-driven kernels against the fp64 oracle evaluated on the same fp16-rounded q.  Tolerance = one fp16 rounding of
    each output (2^-11 relative) plus the fp32 evaluation error."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    kin, tmpl = codegen.template_for(ident)
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
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
    spec.validate()
    h, cm, o = ops.ModelHandle(kin), ops.CostHandle(spec, DEV), oracle_lib.Oracle(kin, spec)
    assert h.specialized
    rng = np.random.default_rng(9)
    for shape in ((3, 64), (5, 37), (1, 1)):
        q = dev((rng.uniform(-2.5, 2.5, size=shape + (kin.n_dofs,))).astype(np.float32)).half()
        q64 = q.cpu().numpy().astype(np.float64).reshape(-1, kin.n_dofs)
        rp, rc, rg = o.rollout(q64, (1, 1, 1, 1), "f64")
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            nb = ops.n_blocks(shape[0] * shape[1])
            sums = torch.zeros(nb, device=DEV)
            pos, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, cost_sum=sums)
            assert pos.dtype == torch.float16 and gq.dtype == torch.float16 and cost.dtype == torch.float32
            assert pos.shape == shape + (kin.n_links, 3) and gq.shape == shape + (kin.n_dofs,)
            assert np.abs(pos.float().cpu().numpy().reshape(rp.shape) - rp).max() < 1e-3 * max(1.0, np.abs(rp).max())
            assert rel_err(cost.cpu().numpy().reshape(-1), rc) < TOL_C
            assert rel_err(gq.float().cpu().numpy().reshape(rg.shape), rg) < 1e-3
            assert abs(float(sums.sum()) - rc.sum()) < 1e-4 * max(1.0, abs(rc.sum()))
            # the loss scale and the mixed mode (fp16 trajectories / positions, fp32 gradient): gq = grad_scale * d cost / d q, the
            # product formed in fp32 -- a power-of-two scale changes nothing but the exponent; a scale that pushes the gradient
            # out of range saturates at +-65504, never inf
            _, c2, g2 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, want_pos=False, grad_scale=2.0 ** -6)
            assert torch.equal(c2, cost) and g2.dtype == torch.float16
            assert rel_err(g2.float().cpu().numpy().reshape(rg.shape) * 64.0, rg) < 1e-3
            p3, c3, g3 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, grad_dtype=torch.float32, grad_scale=0.25)
            assert g3.dtype == torch.float32 and p3.dtype == torch.float16 and torch.equal(c3, cost) and torch.equal(p3, pos)
            assert grad_close(g3.cpu().numpy().reshape(rg.shape) * 4.0, rg)
            _, _, g4 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, want_pos=False, grad_scale=2.0 ** 20)
            g4 = g4.float().cpu().numpy().reshape(rg.shape)
            big = np.abs(rg) * 2.0 ** 20 > 65504.0 * 1.01
            assert np.isfinite(g4).all() and big.any() and (g4[big] == np.sign(rg[big]) * 65504.0).all()
        h.enable_specialized(True)
    with pytest.raises(ValueError):
        ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q.float(), grad_scale=0.5)          # fp32 trajectories take no scale
    with pytest.raises(ValueError):
        ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q, grad_scale=0.0)
    # pre-bound plan in fp16
    q = dev(rng.uniform(-2, 2, size=(4, 64, kin.n_dofs)).astype(np.float32)).half()
    plan = ops.RolloutPlan(h, cm, (0, 1, 0, 1), q)
    plan.launch()
    torch.cuda.synchronize()
    _, rc, rg = o.rollout(q.cpu().numpy().astype(np.float64).reshape(-1, kin.n_dofs), (0, 1, 0, 1), "f64")
    assert plan.gq.dtype == torch.float16 and rel_err(plan.cost.cpu().numpy().reshape(-1), rc) < TOL_C
    assert rel_err(plan.gq.float().cpu().numpy().reshape(rg.shape), rg) < 1e-3


@pytest.mark.parametrize("ident", ["panda", "dual_panda", "ur10_allegro"])
def test_rollout_gp_fused_vs_fp64_oracle(ops, oracle_lib, ident):
    """trk_rollout_gp_cost_grad (build-defined: config 5's objective in one launch) == the fp64 oracle's rollout + GP prior on the same
    (fp16-rounded) trajectories: per-sample cost incl. the prior's factor costs, gq = rollout gradient + prior gradient, gqd, positions,
    block sums; fp32 / fp16 / mixed I/O; horizons that are not multiples of 64 (trajectories cross wavefront blocks anywhere), H = 1;
    the generated kernel (Panda: one segment incl. its self pairs; dual Panda: the two arms one after the other) against the two-launch
    form (what UR10 + Allegro -- ring-staged positions -- and the dual Panda with arm-vs-arm pairs take)."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    kin, tmpl = codegen.template_for(ident)
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.linspace(0.08, 0.14, len(tmpl.obj_links)).astype(np.float32)       # distinct margins: the segments' bases matter
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
    sl = sorted({a for p in tmpl.self_pairs for a in p})
    spec.self_link_idx = np.asarray(sl, np.int32)
    spec.self_pairs = np.asarray([(sl.index(a), sl.index(b)) for a, b in tmpl.self_pairs], np.int32).reshape(-1, 2)
    spec.self_margin = np.linspace(0.04, 0.07, len(tmpl.self_pairs)).astype(np.float32)
    spec.ee_link = tmpl.ee_link
    if tmpl.ee2_link >= 0:
        spec.ee2_link = tmpl.ee2_link
        Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    spec.validate()
    h, cm, o = ops.ModelHandle(kin), ops.CostHandle(spec, DEV), oracle_lib.Oracle(kin, spec)
    D, L = kin.n_dofs, kin.n_links
    rng = np.random.default_rng(41)
    for (B, H, dt, sigma, gw) in ((3, 64, 0.08, 0.3, 1.0), (2, 128, 5.0 / 128, 0.1, 1.0), (5, 37, 0.05, 0.2, 0.5), (70, 3, 0.1, 0.5, 1.0), (4, 1, 0.1, 0.2, 1.0)):
        q = (rng.uniform(-1.0, 1.0, (B, 1, D)) + np.cumsum(rng.standard_normal((B, H, D)) * 0.03, axis=1)).astype(np.float32)
        qd = (rng.standard_normal((B, H, D)) * 0.3).astype(np.float32)
        for wts in ((0, 1, 0, 1), (1, 1, 1, 1), (0.5, 0, 2, 0)):
            for io, gdt in (("f32", None), ("f16", None), ("f16", torch.float32)):
                tq, tqd = dev(q), dev(qd)
                if io == "f16":
                    tq, tqd = tq.half(), tqd.half()
                q64, qd64 = tq.double().cpu().numpy(), tqd.double().cpu().numpy()
                rp, rc, rg = o.rollout(q64.reshape(-1, D), wts, "f64")
                gc, ggq, ggqd = oracle_lib.gp_prior(q64, qd64, dt, sigma, gw, "f64")
                fc = oracle_lib.gp_factor_cost(q64, qd64, dt, sigma, gw, "f64")
                assert np.allclose(fc.sum(1), gc, rtol=1e-12, atol=1e-9)
                ref_c = rc.reshape(B, H) + fc
                ref_g = rg.reshape(B, H, D) + ggq
                gs = 1.0
                if io == "f16" and gdt is None:
                    gs = ops.gp_grad_scale(dt, sigma, gw, float(np.abs(q64).max()), float(np.abs(qd64).max()), extra=float(np.abs(rg).max()))
                nb = ops.n_blocks(B * H)
                # the dual Panda has two generated schedules (round 6): one ROBOT per lane (k_rollout_gpt) and one ARM per lane
                # (k_rollout_gpa: two lanes per sample, sphere scenes without arm-vs-arm pairs); the launch picks by I/O mode,
                # TRK_GP_ARM_LANES forces either -- both are held to the oracle in every I/O mode
                variants = [(True, None), (False, None)] + ([(True, "1"), (True, "0")] if ident == "dual_panda" else [])
                for use_spec, arm_lanes in variants:
                    h.enable_specialized(use_spec)
                    os.environ.pop("TRK_GP_ARM_LANES", None)
                    if arm_lanes is not None:
                        os.environ["TRK_GP_ARM_LANES"] = arm_lanes
                    sums = torch.zeros(nb, device=DEV)
                    pos, cost, gq, gqd = ops.rollout_gp_cost_grad(h, cm, wts, tq, tqd, dt, sigma, gw, cost_sum=sums, grad_dtype=gdt, grad_scale=gs)
                    os.environ.pop("TRK_GP_ARM_LANES", None)
                    tag = (ident, B, H, wts, io, gdt, use_spec, arm_lanes)
                    assert cost.dtype == torch.float32 and pos.dtype == tq.dtype and gq.dtype == (gdt or tq.dtype) and gqd.dtype == gq.dtype
                    ptol = (2 * TOL_H if io == "f32" else 1e-3) * max(1.0, np.abs(rp).max())
                    assert np.abs(pos.float().cpu().numpy().reshape(rp.shape) - rp).max() < ptol, tag
                    assert rel_err(cost.cpu().numpy(), ref_c) < 2e-5, tag
                    c_np = np.concatenate([cost.cpu().numpy().reshape(-1), np.zeros(nb * 64 - B * H, np.float32)]).reshape(nb, 64)
                    assert np.abs(sums.cpu().numpy() - c_np.sum(1)).max() <= 1e-5 * max(1.0, np.abs(c_np).sum(1).max()), tag
                    for got, ref in ((gq, ref_g), (gqd, ggqd)):
                        g = got.double().cpu().numpy() / gs
                        assert np.isfinite(g).all(), tag
                        if got.dtype == torch.float16:      # one fp16 rounding of the scaled sum (the two-launch form: of each term) + fp32 arithmetic
                            assert (np.abs(g - ref) <= 2.0 ** -9 * (np.abs(ref) + np.abs(ggq).max() * 2.0 ** -11) + 2.0 ** -22 / gs
                                    + 4e-5 * max(np.abs(ref).max(), 1e-3)).all(), tag
                        else:
                            assert grad_close(g, ref), tag
                h.enable_specialized(True)
    # which kernel served what: the generated unit for Panda and (without arm-vs-arm pairs) the dual Panda
    assert h.specialized


def test_build_defined_terms_vs_independent_goldens(ops):
    """The two build-defined terms pinned a second way (oracle/gen_golden.py builddef; SURVEY.md 8a has no reference counterpart):
    * GP prior: `trk_gp_prior_cost_grad` and the fused `trk_rollout_gp_cost_grad` (objective weights 0: what is left IS the prior,
      factor by factor) against the DENSE constant-velocity definition in torch fp64 with a numerically inverted Q (gp_prior.npz);
    * damped Gauss-Newton IK: `trk_ik_gn_steps` against one step built from the REFERENCE's geometric Jacobian (robot_tree.py:218-248)
      with `torch.linalg.solve` + scipy's rotation vector, error metric from the reference's SE3_distance (ik_gn_panda.npz)."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    g = gold("gp_prior")
    for k in range(int(g["n_cases"])):
        q, qd = g[f"q_{k}"], g[f"qd_{k}"]
        B, H, D = q.shape
        dt, sigma, w = (float(v) for v in g[f"params_{k}"])
        big = max(np.abs(g[f"gq_{k}"]).max(), np.abs(g[f"gqd_{k}"]).max(), 1e-30)
        c, gq, gqd = ops.gp_prior_cost_grad(dev(q), dev(qd), dt, sigma, w)
        assert rel_err(c.cpu().numpy(), g[f"cost_{k}"]) < 2e-5 or H == 1, k
        assert np.abs(gq.cpu().numpy() - g[f"gq_{k}"]).max() < 2e-5 * big and np.abs(gqd.cpu().numpy() - g[f"gqd_{k}"]).max() < 2e-5 * big, k
        ident = {7: "panda", 14: "dual_panda"}.get(D)
        if ident is None:
            continue
        kin, tmpl = codegen.template_for(ident)
        spec = CostModelSpec(n_links_in=kin.n_links)
        spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
        spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.1, np.float32)
        spec.objects = [o.as_object() for o in EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32)).obj_fixed_list]
        spec.ee_link, spec.ee_target = tmpl.ee_link, np.eye(4, dtype=np.float32)
        if tmpl.ee2_link >= 0:
            spec.ee2_link, spec.ee2_target = tmpl.ee2_link, np.eye(4, dtype=np.float32)
        spec.validate()
        h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, DEV)
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            _, cost, fgq, fgqd = ops.rollout_gp_cost_grad(h, cm, (0.0, 0.0, 0.0, 0.0), dev(q), dev(qd), dt, sigma, w, want_pos=False)
            assert cost.shape == (B, H)
            assert np.abs(cost.cpu().numpy() - g[f"factor_{k}"]).max() <= 2e-5 * max(np.abs(g[f"factor_{k}"]).max(), 1e-30), (k, use_spec)
            assert np.abs(fgq.cpu().numpy() - g[f"gq_{k}"]).max() < 2e-5 * big and np.abs(fgqd.cpu().numpy() - g[f"gqd_{k}"]).max() < 2e-5 * big, (k, use_spec)
        h.enable_specialized(True)
    g = gold("ik_gn_panda")
    m = model("panda_arm_no_gripper")
    h = ops.ModelHandle(m)
    ee = m.name_to_idx[str(g["link"])]
    n = g["q0"].shape[0]
    for tag in ("a", "b"):
        damping, lm_gain, step = (float(v) for v in g[f"params_{tag}"])
        for Ht in (g["H_target"],):
            q = dev(g["q0"]).clone()
            err = torch.empty(n, device=DEV)
            ops.ik_gn_steps(h, ee, dev(Ht), dev(g["lower"]), dev(g["upper"]), q, 1, damping=damping, lm_gain=lm_gain, step_scale=step, err=err)
            dq = np.abs(g[f"q_new_{tag}"] - g["q0"])
            assert (np.abs(q.cpu().numpy() - g[f"q_new_{tag}"]) <= 1e-4 + 5e-3 * dq).all(), tag
            assert rel_err(err.cpu().numpy(), g[f"err_{tag}"]) < 2e-5, tag


@pytest.mark.parametrize("ident", ["panda", "dual_panda", "ur10_allegro"])
def test_generated_fk_positions_and_backward(ops, oracle_lib, ident):
    """trk_fk_positions / trk_fk_positions_backward with all links selected run the generated kernels for the robots
    that have one: against the fp64 oracle and the table-driven kernels, ragged sizes, with and without a base pose."""
    from torch_robotics_amd import codegen
    kin, _ = codegen.template_for(ident)
    h, o = ops.ModelHandle(kin), oracle_lib.Oracle(kin)
    assert h.specialized
    rng = np.random.default_rng(17)
    L, D = kin.n_links, kin.n_dofs
    for base in (False, True):
        if base:
            kin.set_base_pose(np.array([0.1234, -0.2345, 0.0567, 0.9238795, 0.0, 0.3826834, 0.0], np.float32))
            h.set_base_pose(kin.base_R, kin.base_t)
            o.refresh_model()
        for n in (1, 64, 65, 777):
            q = rng.uniform(-3.2, 3.2, size=(n, D)).astype(np.float32)
            w = rng.standard_normal((n, L, 3)).astype(np.float32)
            H64 = o.fk(q.astype(np.float64), "f64")
            gH = np.zeros((n, L, 4, 4)); gH[..., :3, 3] = w
            g64 = o.fk_backward(q.astype(np.float64), gH, "f64")
            wH = rng.standard_normal((n, L, 4, 4)).astype(np.float32)           # a full adjoint: rotations too
            gH64 = o.fk_backward(q.astype(np.float64), wH.astype(np.float64), "f64")
            res = {}
            for use_spec in (True, False):
                h.enable_specialized(use_spec)
                assert grad_close(ops.fk_backward(h, dev(q), dev(wH)).cpu().numpy(), gH64)      # k_fkhbwd / table-driven
                Hm = ops.fk_forward(h, dev(q)).cpu().numpy()          # all links: the generated k_fkh / the table-driven kernel
                assert Hm.shape == (n, L, 4, 4)
                assert np.abs(Hm - H64).max() / max(1.0, float(np.abs(H64).max())) < TOL_H
                np.testing.assert_array_equal(Hm[..., 3, :], np.broadcast_to(np.float32([0, 0, 0, 1]), (n, L, 4)))
                for li in (0, L // 2, L - 1):                         # one link: the generated k_fk1 (walk cut after the target)
                    H1 = ops.fk_forward(h, dev(q), [li]).cpu().numpy()
                    assert H1.shape == (n, 1, 4, 4)
                    assert np.abs(H1[:, 0] - H64[:, li]).max() / max(1.0, float(np.abs(H64).max())) < TOL_H
                pos = ops.fk_positions(h, dev(q)).cpu().numpy()
                gq = ops.fk_positions_backward(h, dev(q), dev(w)).cpu().numpy()
                scale = max(1.0, float(np.abs(H64[..., :3, 3]).max()))
                assert np.abs(pos - H64[..., :3, 3]).max() / scale < TOL_H
                assert grad_close(gq, g64)
                res[use_spec] = (pos, gq)
            h.enable_specialized(True)
            np.testing.assert_allclose(res[True][0], res[False][0], rtol=0, atol=4e-6)


def test_runtime_compiled_kernel_for_another_robot(ops, oracle_lib):
    """A robot without an ahead-of-time unit (KUKA iiwa7): table-driven until jit.specialize() loads a generated kernel;
    then the fused rollout, positions and positions-backward all take the generated path and agree with the fp64 oracle."""
    from torch_robotics_amd import jit
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    m = model("iiwa7")
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.array([3, 5, 7], np.int32)
    spec.obj_link_margin = np.array([0.1, 0.09, 0.08], np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
    spec.self_link_idx = np.array([1, 2, 6, 7], np.int32)
    spec.self_pairs = np.array([[3, 0], [2, 1]], np.int32)        # (7,1), (6,2)
    spec.self_margin = np.array([0.05, 0.05], np.float32)
    spec.ee_link = m.n_links - 1
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = T
    spec.validate()
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    rng = np.random.default_rng(31)
    q = rng.uniform(-2.5, 2.5, size=(5, 64, m.n_dofs)).astype(np.float32)
    rp, rc, rg = o.rollout(q.reshape(-1, m.n_dofs).astype(np.float64), (1, 1, 1, 1), "f64")
    pos_g, cost_g, gq_g = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), dev(q))            # table-driven (or an earlier unit)
    ident = jit.specialize_for_cost_spec(m, spec)
    assert ident is not None and h.specialized
    pos_s, cost_s, gq_s = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), dev(q))
    for pos, cost, gq in ((pos_g, cost_g, gq_g), (pos_s, cost_s, gq_s)):
        assert np.abs(pos.cpu().numpy().reshape(rp.shape) - rp).max() < TOL_H
        assert rel_err(cost.cpu().numpy().reshape(-1), rc) < TOL_C and grad_close(gq.cpu().numpy().reshape(rg.shape), rg)
    # a different collision template of the same robot: no matching unit -> table-driven, still correct
    spec2 = CostModelSpec(n_links_in=m.n_links)
    spec2.obj_link_idx = np.array([2, 4], np.int32); spec2.obj_link_margin = np.array([0.1, 0.1], np.float32)
    spec2.objects = spec.objects
    cm2, o2 = ops.CostHandle(spec2, DEV), oracle_lib.Oracle(m, spec2)
    _, c2, g2 = ops.rollout_cost_grad(h, cm2, (0, 1, 0, 0), dev(q))
    _, rc2, rg2 = o2.rollout(q.reshape(-1, m.n_dofs).astype(np.float64), (0, 1, 0, 0), "f64")
    assert rel_err(c2.cpu().numpy().reshape(-1), rc2) < TOL_C and grad_close(g2.cpu().numpy().reshape(rg2.shape), rg2)
    # FK positions / backward of all links through the generated unit
    w = rng.standard_normal((320, m.n_links, 3)).astype(np.float32)
    pos = ops.fk_positions(h, dev(q.reshape(-1, m.n_dofs))).cpu().numpy()
    assert np.abs(pos - rp).max() < TOL_H
    gH = np.zeros((320, m.n_links, 4, 4)); gH[..., :3, 3] = w
    gq = ops.fk_positions_backward(h, dev(q.reshape(-1, m.n_dofs)), dev(w)).cpu().numpy()
    assert grad_close(gq, o.fk_backward(q.reshape(-1, m.n_dofs).astype(np.float64), gH, "f64"))


def test_pipeline_generator_on_a_tree(ops, oracle_lib):
    """The per-link pipeline generator (running wrench, prefix-sum gradients, late forces across branches) on a TREE:
    Allegro hand with fingertip-vs-fingertip pairs, compiled at run time, against the fp64 oracle."""
    from torch_robotics_amd import jit
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    m = model("allegro_hand")
    idx = m.name_to_idx
    tips = [i for n, i in idx.items() if n.endswith("tip")]
    assert len(tips) == 4
    obj = sorted([idx["palm_link"]] + tips)
    pairs = [(tips[a], tips[b]) for a in range(4) for b in range(a)] + [(tips[0], idx["palm_link"])]
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.asarray(obj, np.int32)
    spec.obj_link_margin = np.full(len(obj), 0.05, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
    sl = sorted({a for p in pairs for a in p})
    spec.self_link_idx = np.asarray(sl, np.int32)
    spec.self_pairs = np.asarray([(sl.index(a), sl.index(b)) for a, b in pairs], np.int32)
    spec.self_margin = np.full(len(pairs), 0.03, np.float32)
    spec.ee_link = tips[2]
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.05, 0.02, 0.15); spec.ee_target = T
    spec.validate()
    ident = jit.specialize(m, obj, pairs, tips[2], pipeline=True)
    assert ident.endswith("_p")
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert h.specialized
    rng = np.random.default_rng(41)
    for n in (64, 129):
        q = rng.uniform(-0.3, 1.5, size=(n, m.n_dofs)).astype(np.float32)
        for w in ((1, 1, 1, 1), (1, 0, 0, 0), (0, 1, 1, 0)):
            rp, rc, rg = o.rollout(q.astype(np.float64), w, "f64")
            pos, cost, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
            assert np.abs(pos.cpu().numpy() - rp).max() < TOL_H
            assert rel_err(cost.cpu().numpy(), rc) < TOL_C, (n, w)
            assert grad_close(gq.cpu().numpy(), rg), (n, w)


def test_two_tracked_end_effectors(ops, oracle_lib):
    """Dual Panda, EE tracking on BOTH arms (BASELINE config 5): generated kernel (baked ee_link + ee2_link), table-driven
    kernel and fp64 oracle; a cost model tracking only the left arm does not match the unit and still agrees."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    kin, tmpl = codegen.template_for("dual_panda")
    assert tmpl.ee2_link == kin.name_to_idx["right_ee_link"]
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))

    def make(ee2):
        spec = CostModelSpec(n_links_in=kin.n_links)
        spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
        spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.1, np.float32)
        spec.objects = [o.as_object() for o in env.obj_fixed_list]
        spec.ee_link = tmpl.ee_link
        T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.5, 0.5); spec.ee_target = T
        if ee2:
            spec.ee2_link = tmpl.ee2_link
            T2 = np.array([[0, -1, 0, 0.4], [1, 0, 0, -0.5], [0, 0, 1, 0.45], [0, 0, 0, 1]], np.float32); spec.ee2_target = T2
        spec.validate()
        return spec
    h = ops.ModelHandle(kin)
    rng = np.random.default_rng(51)
    q = rng.uniform(-2.5, 2.5, size=(3, 64, kin.n_dofs)).astype(np.float32)
    for ee2 in (True, False):
        spec = make(ee2)
        cm, o = ops.CostHandle(spec, DEV), oracle_lib.Oracle(kin, spec)
        rp, rc, rg = o.rollout(q.reshape(-1, kin.n_dofs).astype(np.float64), (0, 1, 0, 1), "f64")
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            _, cost, gq = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), dev(q))
            assert rel_err(cost.cpu().numpy().reshape(-1), rc) < TOL_C, (ee2, use_spec)
            assert grad_close(gq.cpu().numpy().reshape(rg.shape), rg), (ee2, use_spec)
        h.enable_specialized(True)
        if ee2:      # the right arm's joints must feel the second target
            assert np.abs(rg[:, 7:]).max() > 1e-3
            T3 = np.eye(4, dtype=np.float32); T3[:3, 3] = (0.3, -0.4, 0.6)
            cm.set_ee2_target(T3); spec.ee2_target = T3
            o2 = oracle_lib.Oracle(kin, spec)
            _, rc2, rg2 = o2.rollout(q.reshape(-1, kin.n_dofs).astype(np.float64), (0, 1, 0, 1), "f64")
            _, cost, gq = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), dev(q))
            assert rel_err(cost.cpu().numpy().reshape(-1), rc2) < TOL_C and grad_close(gq.cpu().numpy().reshape(rg2.shape), rg2)


@pytest.mark.parametrize("robot", ROBOTS)
def test_runtime_compiled_kernels_on_trees_with_prismatic_joints(ops, oracle_lib, robot):
    """jit.specialize on every robot of the golden set: chains, trees, prismatic joints, axes that degenerate to +-z, a URDF
    whose file order is not a pre-order walk (hab_stretch) -- generated kernel vs table-driven vs fp64 oracle."""
    from torch_robotics_amd import jit
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    m = model(robot)
    leaves = [i for i in range(m.n_links) if not (m.parent == i).any()]
    obj = sorted(set(leaves[:5] + [m.n_links // 2]))
    # a serial chain has one leaf: pair it with the root (a pair (a, a) would be the single-link self distance)
    pairs = [(leaves[0], leaves[-1] if len(leaves) > 1 else 0)] + ([(leaves[1], leaves[0])] if len(leaves) > 1 else [])
    ee = leaves[-1]
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.asarray(obj, np.int32)
    spec.obj_link_margin = np.full(len(obj), 0.07, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1.5])
    sl = sorted({a for p in pairs for a in p})
    spec.self_link_idx = np.asarray(sl, np.int32)
    spec.self_pairs = np.asarray([(sl.index(a), sl.index(b)) for a, b in pairs], np.int32)
    spec.self_margin = np.full(len(pairs), 0.04, np.float32)
    spec.ee_link = ee
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.3, 0.1, 0.8); spec.ee_target = T
    spec.validate()
    assert jit.specialize_for_cost_spec(m, spec) is not None
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert h.specialized
    rng = np.random.default_rng(61)
    q = rng.uniform(-1.5, 1.5, size=(130, m.n_dofs)).astype(np.float32)       # beyond many of the limits: clamps
    rp, rc, rg = o.rollout(q.astype(np.float64), (1, 1, 1, 1), "f64")
    scale = max(1.0, float(np.abs(rp).max()))
    for use_spec in (True, False):
        h.enable_specialized(use_spec)
        pos, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), dev(q))
        assert np.abs(pos.cpu().numpy() - rp).max() / scale < TOL_H, use_spec
        assert rel_err(cost.cpu().numpy(), rc) < TOL_C, use_spec
        assert grad_close(gq.cpu().numpy(), rg), use_spec
    h.enable_specialized(True)
    w = rng.standard_normal((130, m.n_links, 3)).astype(np.float32)
    gH = np.zeros((130, m.n_links, 4, 4)); gH[..., :3, 3] = w
    assert grad_close(ops.fk_positions_backward(h, dev(q), dev(w)).cpu().numpy(), o.fk_backward(q.astype(np.float64), gH, "f64"))


def test_empty_batches_everywhere(ops):
    """N = 0 through every entry point: no launch, correctly shaped empty outputs, no error."""
    m = model("panda_arm_no_gripper")
    h = ops.ModelHandle(m)
    spec = panda_cost_spec(gold("cost_spheres3d"), gold("panda_robot"), ee_target=np.eye(4, dtype=np.float32))
    cm = ops.CostHandle(spec, DEV)
    q0 = torch.empty((0, 7), device=DEV)
    assert ops.fk_forward(h, q0).shape == (0, 11, 4, 4)
    assert ops.fk_positions(h, q0).shape == (0, 11, 3)
    assert ops.fk_backward(h, q0, torch.empty((0, 11, 4, 4), device=DEV)).shape == (0, 7)
    assert ops.fk_positions_backward(h, q0, torch.empty((0, 11, 3), device=DEV)).shape == (0, 7)
    pos, quat, lin, ang = ops.fk_jacobian(h, q0, None, 10)[:4]
    assert pos.shape == (0, 3) and lin.shape == (0, 3, 7)
    assert ops.fk_analytic_jacobian(h, q0).shape == (0, 11, 7, 7)
    assert ops.rotmat_to_quat(torch.empty((0, 3, 3), device=DEV)).shape[0] == 0
    lp = torch.empty((0, 11, 3), device=DEV)
    assert ops.cost_fields(cm, 7, lp).shape == (0,)
    assert ops.collision_fields(cm, 7, lp).shape == (0,)
    assert ops.ee_cost(cm, torch.empty((0, 4, 4), device=DEV)).shape == (0,)
    pos, cost, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), torch.empty((0, 64, 7), device=DEV))
    assert pos.shape == (0, 64, 11, 3) and cost.shape == (0, 64) and gq.shape == (0, 64, 7)
    ps = ops.PointSetHandle(h, np.arange(11, dtype=np.int32), np.zeros((11, 3), np.float32), DEV)
    assert ops.fk_points(ps, q0).shape == (0, 11, 3)
    assert ops.fk_points_backward(ps, q0, lp).shape == (0, 7)
    assert ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 1), q0)[1].shape == (0,)
    c, gq_, gqd = ops.gp_prior_cost_grad(torch.empty((0, 8, 7), device=DEV), torch.empty((0, 8, 7), device=DEV), 0.1, 0.1)
    assert c.shape == (0,) and gq_.shape == (0, 8, 7)
    assert ops.interpolate_traj_via_points(torch.empty((0, 8, 7), device=DEV), 5).shape == (0, 35, 7)
    assert float(ops.reduce_sum(torch.empty((0,), device=DEV))) == 0.0


def _chain_urdf(path, n_links):
    """Serial chain in file order == walk order: every third joint revolute (axes cycling x, y, z), the others fixed."""
    axes = ["1 0 0", "0 1 0", "0 0 1"]
    lines = ['<?xml version="1.0"?>', f'<robot name="chain{n_links}">', '  <link name="l0"/>']
    nj = 0
    for i in range(1, n_links):
        lines.append(f'  <link name="l{i}"/>')
        xyz = f"{0.03 + 0.01 * (i % 3):.3f} {0.02 * ((i % 5) - 2):.3f} {0.05 + 0.005 * (i % 4):.3f}"
        rpy = f"{0.1 * (i % 3):.2f} {-0.07 * (i % 4):.2f} {0.05 * (i % 5):.2f}"
        if i % 3 == 1:
            lines += [f'  <joint name="j{i}" type="revolute">', f'    <parent link="l{i - 1}"/><child link="l{i}"/>',
                      f'    <origin xyz="{xyz}" rpy="{rpy}"/><axis xyz="{axes[nj % 3]}"/>',
                      '    <limit lower="-2.5" upper="2.5" effort="1" velocity="1"/>', '  </joint>']
            nj += 1
        else:
            lines += [f'  <joint name="j{i}" type="fixed">', f'    <parent link="l{i - 1}"/><child link="l{i}"/>',
                      f'    <origin xyz="{xyz}" rpy="{rpy}"/>', '  </joint>']
    lines.append("</robot>")
    path.write_text("\n".join(lines))


@pytest.mark.parametrize("n_links", [27, 28, 32, 34, 43, 44])
def test_ring_staging_geometries(ops, oracle_lib, tmp_path, n_links):
    """The ring staging of the link positions (RingFlusher) on rows of 81 floats (odd row: heads of period 8, tail units that end
    inside a 2-float vector), 84 (heads 0 / 4), 96 (rows that are whole sectors: no heads, a full-chunk tail), 102 (three whole
    chunks, the last of them complete only with the last float), 129 (the tail unit would not fit: unaligned ring, 1-float pieces)
    and 132 (four whole chunks, tail units of 8 and 0 floats): full and ragged wavefronts, fp32 and fp16 output, against the
    fp64 oracle."""
    from torch_robotics_amd import codegen, jit
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    from torch_robotics_amd.kinmodel import KinModel
    urdf = tmp_path / f"chain{n_links}.urdf"
    _chain_urdf(urdf, n_links)
    m = KinModel.from_urdf(str(urdf))
    assert m.n_links == n_links and 3 * n_links > codegen.CHUNKED_STAGING_MIN_FLOATS
    rp = codegen.ring_plan(3 * n_links)
    assert (rp.V, rp.aligned, rp.hx, rp.n_full, rp.tail) == {27: (2, True, 7, 2, 17), 28: (2, True, 4, 2, 20), 32: (2, True, 0, 2, 32),
                                                            34: (2, True, 6, 3, 6), 43: (1, False, 0, 4, 1),
                                                            44: (2, True, 4, 4, 4)}[n_links]
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    spec = CostModelSpec(n_links_in=m.n_links)
    spec.obj_link_idx = np.array([5, 11, n_links - 1], np.int32)
    spec.obj_link_margin = np.array([0.1, 0.09, 0.08], np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = m.n_links - 1
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = T
    spec.validate()
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    assert jit.specialize_for_cost_spec(m, spec) is not None and h.specialized
    rng = np.random.default_rng(n_links)
    for n in (64, 65, 300):
        q = rng.uniform(-2.5, 2.5, size=(n, m.n_dofs)).astype(np.float32)
        p64, c64, g64 = o.rollout(q.astype(np.float64), (0, 1, 0, 1), "f64")
        scale = max(1.0, float(np.abs(p64).max()))
        pos, c, gq = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), dev(q))
        assert np.abs(pos.cpu().numpy() - p64).max() / scale < TOL_H, n
        assert rel_err(c.cpu().numpy(), c64) < TOL_C and grad_close(gq.cpu().numpy(), g64)
        assert np.abs(ops.fk_positions(h, dev(q)).cpu().numpy() - p64).max() / scale < TOL_H        # positions-only exit
        _, c_np, gq_np = ops.rollout_cost_grad(h, cm, (0, 1, 0, 1), dev(q), want_pos=False)        # the kernel without staging
        assert rel_err(c_np.cpu().numpy(), c64) < TOL_C and grad_close(gq_np.cpu().numpy(), g64)
        plan = ops.RolloutPlan(h, cm, (0, 1, 0, 1), dev(q).half().reshape(1, n, m.n_dofs))         # fp16 I/O: 2-byte elements
        plan.launch(); torch.cuda.synchronize()
        p16, _, _ = o.rollout(q.astype(np.float16).astype(np.float64), (0, 1, 0, 1), "f64")
        assert np.abs(plan.link_pos.float().cpu().numpy().reshape(p16.shape) - p16).max() / scale < 2e-3


@pytest.mark.parametrize("ident", ["panda", "dual_panda", "ur10_allegro"])
def test_randomised_generated_vs_table_driven_and_oracle(ops, oracle_lib, ident):
    """A seeded sweep over what a caller can vary around one generated unit -- batch size (ragged, one wavefront, several
    workgroups), weights (terms on and off), margins, hinge flags, workspace box, a base pose, positions wanted or not, fp16 I/O --
    each drawn at random: generated kernel == table-driven kernel == fp64 oracle on every draw."""
    from torch_robotics_amd import codegen
    from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import EnvSpheres3D
    kin, tmpl = codegen.template_for(ident)
    env = EnvSpheres3D(tensor_args=dict(device=DEV, dtype=torch.float32))
    rng = np.random.default_rng({"panda": 101, "dual_panda": 202, "ur10_allegro": 303}[ident])
    self_links = sorted({a for p in tmpl.self_pairs for a in p})
    for draw in range(40):
        spec = CostModelSpec(n_links_in=kin.n_links)
        spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
        spec.obj_link_margin = rng.uniform(0.0, 0.15, len(tmpl.obj_links)).astype(np.float32)
        spec.objects = [o.as_object() for o in env.obj_fixed_list]
        if rng.random() < 0.5:
            spec.ws_min, spec.ws_max = rng.uniform(-1.5, -0.3, 3).astype(np.float32), rng.uniform(0.3, 1.5, 3).astype(np.float32)
        spec.self_link_idx = np.asarray(self_links, np.int32)
        spec.self_pairs = np.asarray([(self_links.index(a), self_links.index(b)) for a, b in tmpl.self_pairs], np.int32).reshape(-1, 2)
        spec.self_margin = rng.uniform(0.02, 0.3, len(tmpl.self_pairs)).astype(np.float32)
        spec.ee_link = tmpl.ee_link
        Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = rng.uniform(-0.6, 0.6, 3); spec.ee_target = Ht
        if tmpl.ee2_link >= 0:
            spec.ee2_link = tmpl.ee2_link
            Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = rng.uniform(-0.6, 0.6, 3); spec.ee2_target = Ht2
        spec.clamp_fields = int(rng.choice([0, FIELD_OBJECTS, FIELD_SELF | FIELD_WS, FIELD_OBJECTS | FIELD_SELF | FIELD_WS]))
        spec.validate()
        if rng.random() < 0.4:
            ang = rng.uniform(-1.0, 1.0)
            kin.set_base_pose(np.array([*rng.uniform(-0.3, 0.3, 3), np.cos(ang / 2), 0.0, 0.0, np.sin(ang / 2)], np.float32))
        else:
            kin.set_base_pose(np.array([0, 0, 0, 1, 0, 0, 0], np.float32))
        h, cm, o = ops.ModelHandle(kin), ops.CostHandle(spec, DEV), oracle_lib.Oracle(kin, spec)
        assert h.specialized
        n = int(rng.choice([1, 7, 64, 65, 129, 300, 1024]))
        w = tuple(float(v) for v in rng.choice([0.0, 0.5, 1.0, 2.0], 4))
        q = rng.uniform(-3.0, 3.0, (n, kin.n_dofs)).astype(np.float32)
        want_pos = bool(rng.random() < 0.7)
        p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
        scale = max(1.0, float(np.abs(p64).max()))
        res = {}
        for use_spec in (True, False):
            h.enable_specialized(use_spec)
            pos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q), want_pos=want_pos)
            if want_pos:
                assert np.abs(pos.cpu().numpy() - p64).max() / scale < TOL_H, (ident, draw, n, use_spec)
            else:
                assert pos is None
            # a single sample's cost can be a small difference of O(1) terms (unclamped margin - sdf): absolute floor of 1
            assert np.abs(c.cpu().numpy() - c64).max() / max(1.0, float(np.abs(c64).max())) < TOL_C, (ident, draw, n, w, use_spec)
            # per sample; a sample on a kink (arg-min tie within fp32 rounding) must carry the gradient of one of its tied branches
            gerr = np.abs(gq.cpu().numpy() - g64).max(-1) / max(1.0, float(np.abs(g64).max()))
            assert kink_rows_ok(gq.cpu().numpy(), g64, q, lambda qq: o.rollout(qq, w, "f64")[2], gerr >= TOL_G), (ident, draw, n, w, use_spec)
            res[use_spec] = (c, gq, gerr >= TOL_G)
        h.enable_specialized(True)
        # the same fields on GIVEN positions: the unit's field kernels against the table-driven ones
        posd = ops.fk_positions(h, dev(q))
        fl = int(rng.choice([1, 2, 4, 3, 6, 7]))
        mg = None if rng.random() < 0.5 else float(rng.uniform(0.0, 0.1))
        outs = {}
        for use_unit in (True, False):
            cm.enable_specialized(use_unit)
            outs[use_unit] = ops.cost_fields(cm, fl, posd, want_grad=True) + (ops.collision_fields(cm, fl, posd, margin=mg),)
        cm.enable_specialized(True)
        (c1, g1, b1), (c0, g0, b0) = outs[True], outs[False]
        assert float((c1 - c0).abs().max()) <= TOL_C * max(1.0, float(c0.abs().max())), (ident, draw, fl)
        off = ((g1 - g0).abs().flatten(1).max(1)[0] > TOL_G * max(1.0, float(g0.abs().max()))).sum()
        assert int(off) <= max(1, n // 500), (ident, draw, fl)          # the two kernel families rank the primitives by different arithmetic: ties may split
        assert int((b1 != b0).sum()) <= max(1, n // 2000), (ident, draw, fl, mg)
        if rng.random() < 0.5:                                        # fp16 I/O through the generated kernel
            p16, c16, g16 = o.rollout(q.astype(np.float16).astype(np.float64), w, "f64")
            pos_h, c_h, gq_h = ops.rollout_cost_grad(h, cm, w, dev(q).half(), want_pos=True)
            assert np.abs(pos_h.float().cpu().numpy() - p16).max() / scale < 2e-3
            assert np.abs(c_h.cpu().numpy() - c16).max() / max(1.0, float(np.abs(c16).max())) < 1e-4
    kin.set_base_pose(np.array([0, 0, 0, 1, 0, 0, 0], np.float32))


@pytest.mark.parametrize("generated", [False, True], ids=["table_driven", "generated"])
def test_interpolated_link_points(ops, oracle_lib, generated):
    """interpolate_link_pos (distance_fields.py:66-69, 145-147): the fields on points interpolated along the selected links --
    virtual columns of the cost model, evaluated (and their adjoints scattered back) inside the field kernels and the fused
    rollout -- against the reference's field code on interpolate_points_v1 of the links (goldens) and the fp64 oracle."""
    from helpers import interp_cost_spec
    g, robot = gold("cost_interp"), gold("panda_robot")
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    spec = interp_cost_spec(ee_target=Ht)
    m = model("panda_arm_no_gripper")
    h, cm, o = ops.ModelHandle(m), ops.CostHandle(spec, DEV), oracle_lib.Oracle(m, spec)
    pos = dev(robot["fk_map_collision"].reshape(-1, 11, 3))
    qg = dev(g["q"].reshape(-1, 7))
    if generated:
        # a unit with THIS interpolation table baked in (the lerp of two link registers, the adjoint scattered back to them)
        from torch_robotics_amd import jit
        assert jit.specialize_for_cost_spec(m, spec) is not None and jit.has_matching_unit(m, spec)
    else:
        h.enable_specialized(False)
        cm.enable_specialized(False)
    for fname, fl, w in (("self", FIELD_SELF, (1, 0, 0, 0)), ("objects", FIELD_OBJECTS, (0, 1, 0, 0)), ("ws", FIELD_WS, (0, 0, 1, 0))):
        c, gp = ops.cost_fields(cm, fl, pos, want_grad=True)
        assert gp.shape == (64, 11, 3)                              # gradients come back on the REAL columns
        assert rel_err(c.cpu().numpy(), g[f"cost_{fname}"].reshape(-1)) < TOL_C, fname
        assert grad_close(gp.cpu().numpy(), g[f"gpos_{fname}"].reshape(-1, 11, 3)), fname
        gc = torch.linspace(0.5, 2.0, 64, device=DEV)
        _, gp2 = ops.cost_fields(cm, fl, pos, gcost=gc, want_grad=True)
        assert torch.allclose(gp2, gp * gc[:, None, None], rtol=1e-6, atol=1e-7)
        _, c2, gq = ops.rollout_cost_grad(h, cm, w, qg)
        assert rel_err(c2.cpu().numpy(), g[f"cost_{fname}"].reshape(-1)) < TOL_C, fname
        assert grad_close(gq.cpu().numpy(), g[f"gq_{fname}"].reshape(-1, 7)), fname
        assert np.array_equal(ops.collision_fields(cm, fl, pos).cpu().numpy(), g[f"coll_{fname}"].reshape(-1)), fname
        assert np.array_equal(ops.collision_fields(cm, fl, pos, margin=0.0).cpu().numpy(), g[f"coll0_{fname}"].reshape(-1)), fname
        assert np.array_equal(ops.rollout_collision(h, cm, fl, qg).cpu().numpy(), g[f"coll_{fname}"].reshape(-1)), fname
    _, c, gq = ops.rollout_cost_grad(h, cm, (1, 1, 1, 0), qg)
    assert rel_err(c.cpu().numpy(), g["cost_total"].reshape(-1)) < TOL_C and grad_close(gq.cpu().numpy(), g["gq_total"].reshape(-1, 7))
    if generated:       # the generated and the table-driven kernels are the same function through two code paths
        h2, cm2 = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
        h2.enable_specialized(False)
        _, c_t, g_t = ops.rollout_cost_grad(h2, cm2, (1, 1, 1, 1), qg)
        _, c_g, g_g = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), qg)
        assert rel_err(c_g.cpu().numpy(), c_t.cpu().numpy()) < TOL_C and grad_close(g_g.cpu().numpy(), g_t.cpu().numpy())
    rng = np.random.default_rng(11)
    for n in (1, 63, 1000):
        q = rng.uniform(-3.0, 3.9, (n, 7)).astype(np.float32)
        for w in ((1, 1, 1, 1), (0.5, 2.0, 0.25, 0.0)):
            p64, c64, g64 = o.rollout(q.astype(np.float64), w, "f64")
            ppos, c, gq = ops.rollout_cost_grad(h, cm, w, dev(q))
            assert np.abs(ppos.cpu().numpy() - p64).max() < TOL_H
            # (random batches: a sample on an arg-min tie may take the other branch in fp32 -- helpers.kink_rows_ok bounds what it may return)
            assert rel_err(c.cpu().numpy(), c64) < TOL_C, (n, w)
            assert grad_close_kinks(gq.cpu().numpy(), g64, q, lambda qq: o.rollout(qq, w, "f64")[2]), (n, w)
            fl3 = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
            c_f, gp_f = ops.cost_fields(cm, fl3, ppos, want_grad=True)
            c_o, gp_o = o.cost_fields(fl3, p64, "f64")
            assert rel_err(c_f.cpu().numpy(), c_o) < TOL_C, (n, w)
            assert grad_close_kinks(gp_f.cpu().numpy(), gp_o, p64, lambda pp: o.cost_fields(fl3, pp.reshape(-1, 11, 3), "f64")[1].reshape(len(pp), -1),
                                    radius=1e-6), (n, w)


def test_interpolate_points_v1_op(ops):
    """trk_interpolate_columns(_backward) against the reference's interpolate_points_v1 and its autograd (goldens)."""
    from torch_robotics_amd.fields import interpolate_points_v1
    g = gold("cost_interp")
    for L, K in g["ip_shapes"]:
        x = dev(g[f"ip_{L}_{K}_in"]).requires_grad_(True)
        out = interpolate_points_v1(x, int(K))
        assert out.shape == (6, K, 3)
        assert np.abs(out.detach().cpu().numpy() - g[f"ip_{L}_{K}_out"]).max() < 5e-7
        (out * dev(g[f"ip_{L}_{K}_w"])).sum().backward()
        assert np.abs(x.grad.cpu().numpy() - g[f"ip_{L}_{K}_gin"]).max() < 2e-6
    x4 = torch.randn(3, 4, 5, 3, device=DEV)                         # any leading shape (the reference needs exactly 3-D)
    assert torch.equal(interpolate_points_v1(x4, 9).reshape(12, 9, 3), interpolate_points_v1(x4.reshape(12, 5, 3), 9))


def test_single_link_self_distance(ops, oracle_lib):
    """distance_fields.py:195-198: one self-collision link -> "distance" |p|_1 * 1e9 (cost = margin - that, gradient -1e9 sign(p))."""
    from helpers import single_link_self_spec
    g, robot = gold("cost_interp"), gold("panda_robot")
    spec = single_link_self_spec()
    m = model("panda_arm_no_gripper")
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    pos = dev(robot["fk_map_collision"].reshape(-1, 11, 3))
    c, gp = ops.cost_fields(cm, FIELD_SELF, pos, want_grad=True)
    assert rel_err(c.cpu().numpy(), g["single_cost"].reshape(-1)) < 1e-6
    assert np.array_equal(gp.cpu().numpy(), g["single_gpos"].reshape(-1, 11, 3))
    for on in (True, False):                                        # a generated unit must not take this cost model
        h.enable_specialized(on)
        _, c2, gq = ops.rollout_cost_grad(h, cm, (1, 0, 0, 0), dev(g["q"].reshape(-1, 7)))
        assert rel_err(c2.cpu().numpy(), g["single_cost"].reshape(-1)) < 1e-6
        assert rel_err(gq.cpu().numpy(), g["single_gq"].reshape(-1, 7)) < 1e-5
    assert np.array_equal(ops.collision_fields(cm, FIELD_SELF, pos).cpu().numpy(), g["single_coll"].reshape(-1))
    assert ops.collision_fields(cm, FIELD_SELF, torch.zeros(2, 11, 3, device=DEV)).all()       # |0|_1 * 1e9 < margin


@pytest.mark.parametrize("robot,link", [("panda_arm_no_gripper", "ee_link"), ("ur10", "ee_link"), ("iiwa7", None), ("dual_panda", None)])
def test_jtj_normal_equations(ops, oracle_lib, robot, link):
    """trk_jtj: J^T J and J^T r of the geometric Jacobian (robot_tree.py:238-246), per-lane FMA kernel and the
    v_mfma_f32_4x4x1_16b_f32 kernel, against fp64 numpy on the Jacobians of trk_fk_jacobian (goldens pin those), ragged sizes."""
    m = model(robot)
    h = ops.ModelHandle(m)
    li = m.name_to_idx[link] if link else m.n_links - 1
    D = m.n_dofs
    rng = np.random.default_rng(8)
    for n in (1, 63, 64, 257, 4099):
        q = rng.uniform(-2.0, 2.0, (n, D)).astype(np.float32)
        _, _, lin, ang = ops.fk_jacobian(h, dev(q), None, li)
        r = dev(rng.standard_normal((n, 6)).astype(np.float32))
        J = np.concatenate([lin.cpu().numpy(), ang.cpu().numpy()], 1).astype(np.float64)          # (n, 6, D)
        ref = np.einsum("nki,nkj->nij", J, J)
        ref_r = np.einsum("nki,nk->ni", J, r.cpu().numpy().astype(np.float64))
        for mfma in ((False, True) if D <= 8 else (False,)):
            JtJ, Jtr = ops.jtj(lin, ang, r, mfma=mfma)
            assert JtJ.shape == (n, D, D) and Jtr.shape == (n, D)
            assert np.abs(JtJ.cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (n, mfma)
            assert np.abs(Jtr.cpu().numpy() - ref_r).max() <= 2e-6 * max(1.0, np.abs(ref_r).max()), (n, mfma)
            assert torch.equal(JtJ, JtJ.transpose(1, 2))                        # exactly symmetric (a product commutes)
            assert torch.equal(ops.jtj(lin, ang, mfma=mfma), JtJ)               # without a residual: JtJ alone
            # the damped step, solved inside the kernel (Cholesky per sample), against numpy's fp64 solve
            for lam_t in (torch.tensor([0.05], device=DEV), torch.linspace(0.01, 0.2, n, device=DEV)):
                J2, r2, dq = ops.jtj(lin, ang, r, mfma=mfma, damping=lam_t, solve=True)
                assert torch.equal(J2, JtJ) and torch.equal(r2, Jtr)
                lam = lam_t.cpu().numpy().astype(np.float64).reshape(-1)
                A = ref + (lam[:, None, None] if lam.size == n else lam[0]) * np.eye(D)
                x = np.linalg.solve(A, ref_r[..., None])[..., 0]
                assert np.abs(dq.cpu().numpy() - x).max() <= 2e-4 * max(1.0, np.abs(x).max()), (n, mfma)
    if D > 8:
        with pytest.raises(NotImplementedError):
            ops.jtj(lin, ang, r, mfma=True)
    with pytest.raises(ValueError):
        ops.jtj(lin, ang[:, :2], r)


@pytest.mark.parametrize("name", ["ur10_allegro", "dual_panda"])
def test_tree_robot_costs_vs_reference_goldens(ops, name):
    """BASELINE configs 4 / 5: the generated units of UR10 + Allegro and dual Panda (and the table-driven kernels) against costs and
    gradients the REFERENCE computed for these robots -- its field classes on its own FK, autograd through the FK recursion
    (tests/golden/cost_tree_<robot>.npz); fused rollout, fp16 I/O within fp16 rounding, boolean fields."""
    from helpers import tree_cost_spec
    m, spec, g = tree_cost_spec(name)
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    q = dev(g["q"])
    for use_spec in (True, False):
        h.enable_specialized(use_spec)
        assert h.specialized == use_spec
        for fname, w in (("self", (1, 0, 0, 0)), ("objects", (0, 1, 0, 0)), ("ws", (0, 0, 1, 0)), ("ee", (0, 0, 0, 1)), ("total", (1, 1, 1, 1))):
            pos, c, gq = ops.rollout_cost_grad(h, cm, w, q)
            assert rel_err(c.cpu().numpy(), g[f"cost_{fname}"]) < TOL_C, (fname, use_spec)
            assert grad_close(gq.cpu().numpy(), g[f"gq_{fname}"]), (fname, use_spec)
        assert np.abs(pos.cpu().numpy() - g["link_pos"]).max() < TOL_H
        for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS)):
            assert np.array_equal(ops.rollout_collision(h, cm, fl, q).cpu().numpy(), g[f"coll_{fname}"]), (fname, use_spec)
            assert np.array_equal(ops.rollout_collision(h, cm, fl, q, margin=0.0).cpu().numpy(), g[f"coll0_{fname}"]), (fname, use_spec)
    h.enable_specialized(True)
    # fp16 I/O (config 5's storage format): the reference's fp32 values within the rounding of q, positions and gradient to fp16
    pos16, c16, g16 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q.half())
    _, c32, g32 = ops.rollout_cost_grad(h, cm, (1, 1, 1, 1), q.half().float())
    assert pos16.dtype == torch.float16 and torch.equal(c16, c32)
    assert rel_err(g16.float().cpu().numpy(), g32.cpu().numpy()) < 2.0 ** -10
    # the field kernels on the reference's link positions
    lp = dev(g["link_pos"])
    for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS)):
        assert rel_err(ops.cost_fields(cm, fl, lp).cpu().numpy(), g[f"cost_{fname}"]) < TOL_C, fname
