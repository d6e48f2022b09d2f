"""Pins the CPU oracle (oracle/oracle.c) against golden vectors produced by the reference itself.

CPU-only.  Every check compares the C restatement, in fp32 and fp64, with the reference's fp32
outputs stored under tests/golden/ (generator: oracle/gen_golden.py).  Tolerances are fp32
round-off: the reference composes poses right-to-left through O(L^2) products
(rigid_body.py:200-211), the oracle walks the chain once, so the last ulps differ.
"""
import numpy as np
import pytest

from helpers import ROBOTS, gold, model, panda_cost_spec, rel_err
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS

TOL_H = 2e-6        # |dH| relative to max(1, |translation|)
TOL_G = 5e-6        # gradient, relative to its max magnitude
TOL_C = 2e-6        # costs, relative


@pytest.mark.parametrize("robot", ROBOTS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_fk_forward_backward(oracle_lib, robot, prec):
    g = gold(f"fk_{robot}")
    o = oracle_lib.Oracle(model(robot))
    for tag in ("in", "out"):       # "out": q beyond the joint limits -> exercises clamp + zero gradient
        q, Hg = g[f"q_{tag}"], g[f"H_{tag}"]
        scale = max(1.0, float(np.abs(Hg[..., :3, 3]).max()))
        H = o.fk(q, prec)
        assert np.abs(H - Hg).max() / scale < TOL_H
        np.testing.assert_array_equal(H[..., 3, :], np.broadcast_to([0, 0, 0, 1], H[..., 3, :].shape))
        gq = o.fk_backward(q, g[f"w_{tag}"], prec)
        assert rel_err(gq, g[f"gq_{tag}"]) < TOL_G
        if tag == "out":            # clamped joints get exactly zero gradient, like torch.clamp
            m = o.model
            lo, hi = m.lower[m.controlled], m.upper[m.controlled]
            cl = m.clamp[m.controlled].astype(bool)
            outside = ((q < lo) | (q > hi)) & cl
            assert outside.any()
            assert np.all(gq[outside] == 0) and np.all(g["gq_out"][outside] == 0)


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "allegro_hand", "hab_stretch"])
def test_fk_reference_association_order(oracle_lib, robot):
    """The O(L^2) right-to-left variant (the reference's own association) agrees as well."""
    g = gold(f"fk_{robot}")
    o = oracle_lib.Oracle(model(robot))
    Hg = g["H_in"]
    scale = max(1.0, float(np.abs(Hg[..., :3, 3]).max()))
    assert np.abs(o.fk(g["q_in"], "f32", ref_order=True) - Hg).max() / scale < TOL_H


def test_panda_known_answers(oracle_lib):
    """KATs from SURVEY.md section 4 (q = 0; joint 4 clamps to -0.0698)."""
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"))
    H = o.fk(np.zeros((1, 7), np.float32), "f32")[0]
    exp = {1: (0, 0, 0.333), 3: (0, 0, 0.649), 4: (0.0825, 0, 0.649), 5: (0.026982, 0, 1.037819),
           7: (0.114768, 0, 1.031681), 9: (0.107306, 0, 0.924942), 10: (0.100331, 0, 0.825185)}
    for link, p in exp.items():
        np.testing.assert_allclose(H[link, :3, 3], p, atol=2e-6)
    np.testing.assert_allclose(H[10, 0, :3], (-0.704823, 0.705946, -0.069743), atol=2e-6)


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "ur10", "iiwa7"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_stateful_fk_and_geometric_jacobian(oracle_lib, robot, prec):
    g = gold(f"jac_{robot}")
    m = model(robot)
    o = oracle_lib.Oracle(m)
    for k, link in enumerate(g["links"]):
        pos, quat, lin, ang, vl, va = o.jacobian(g["q"], g["qd"], m.name_to_idx[str(link)], prec)
        assert np.abs(pos - g[f"pos_{k}"]).max() < 2e-6
        assert np.abs(quat - g[f"quat_{k}"]).max() < 2e-6
        assert np.abs(lin - g[f"lin_{k}"]).max() < 3e-6
        assert np.abs(ang - g[f"ang_{k}"]).max() < 2e-6
        assert np.abs(vl - g[f"vel_lin_{k}"]).max() < 2e-6
        assert np.abs(va - g[f"vel_ang_{k}"]).max() < 2e-6


def test_rotation_matrix_to_quaternion(oracle_lib):
    g = gold("quat")
    for prec in ("f32", "f64"):
        q = oracle_lib.Oracle.rotmat_to_quat(g["R"], prec)
        # the 4 exact 180-degree matrices at the end sit on argmax ties; compare up to sign there
        err = np.minimum(np.abs(q - g["q_wxyz"]).max(-1), np.abs(q + g["q_wxyz"]).max(-1))
        assert err[:-4].max() < 2e-6 and np.abs(q[:-4] - g["q_wxyz"][:-4]).max() < 2e-6
        assert err[-4:].max() < 2e-6


ENVS = ["spheres3d", "spheres3d_grid", "table_shelf", "maze_boxes3d", "spheres3d_extra"]


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_collision_fields(oracle_lib, env, prec):
    robot, g = gold("panda_robot"), gold(f"cost_{env}")
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), panda_cost_spec(g, robot))
    pos = robot["fk_map_collision"].reshape(-1, 11, 3)
    for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS)):
        c, gp = o.cost_fields(fl, pos, prec)
        assert rel_err(c, g[f"cost_{fname}"].reshape(-1)) < TOL_C, fname
        assert rel_err(gp, g[f"gpos_{fname}"].reshape(-1, 11, 3)) < TOL_G, fname
        if prec == "f32":           # booleans: bit-exact in the reference's own precision
            np.testing.assert_array_equal(o.collision_fields(fl, pos), g[f"coll_{fname}"].reshape(-1))
            np.testing.assert_array_equal(o.collision_fields(fl, pos, margin=0.0), g[f"coll0_{fname}"].reshape(-1))
    if "cost_extra" in g:
        o2 = oracle_lib.Oracle(model("panda_arm_no_gripper"), panda_cost_spec(g, robot, which="extra"))
        c, gp = o2.cost_fields(FIELD_OBJECTS, pos, prec)
        assert rel_err(c, g["cost_extra"].reshape(-1)) < TOL_C
        assert rel_err(gp, g["gpos_extra"].reshape(-1, 11, 3)) < TOL_G
    # PlanningTask.compute_collision_cost (+ backward to q) == fused rollout with weights (1,1,1,0)
    pos_r, c, gq = o.rollout(g["q"].reshape(-1, 7), (1, 1, 1, 0), prec)
    assert np.abs(pos_r - pos).max() < 2e-6
    assert rel_err(c, g["cost_total"].reshape(-1)) < TOL_C
    assert rel_err(gq, g["gq_total"].reshape(-1, 7)) < TOL_G
    for fname, w in (("self", (1, 0, 0, 0)), ("objects", (0, 1, 0, 0)), ("ws", (0, 0, 1, 0))):
        _, _, gq = o.rollout(g["q"].reshape(-1, 7), w, prec)
        assert rel_err(gq, g[f"gq_{fname}"].reshape(-1, 7)) < TOL_G, fname
    if prec == "f32":
        all_f = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
        np.testing.assert_array_equal(o.collision_fields(all_f, pos), g["coll_total"].reshape(-1))
        np.testing.assert_array_equal(o.collision_fields(all_f, pos, margin=0.0), g["coll0_total"].reshape(-1))


def test_grid_precompute_matches_reference_grid(oracle_lib):
    """GridMapSDF.precompute_sdf (grid_map_sdf.py:34-63): values and stored gradients."""
    robot, g = gold("panda_robot"), gold("cost_spheres3d_grid")
    ga = gold("cost_spheres3d")      # analytic scene with the same objects
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), panda_cost_spec(ga, robot))
    sdf, grad = o.grid_precompute(g["grid_cmap_dim"], g["limits"][0], g["limits"][1])
    assert np.abs(sdf - g["grid_sdf"]).max() < 2e-6
    # gradient flips between equidistant spheres on ties: compare where the argmin is unambiguous
    diff = np.abs(grad - g["grid_grad"]).max(-1)
    assert (diff < 1e-5).mean() > 0.999


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ee_se3_cost(oracle_lib, prec):
    g, robot = gold("cost_ee"), gold("panda_robot")
    m = model("panda_arm_no_gripper")
    gs = gold("cost_spheres3d")
    q = g["q"].reshape(-1, 7)
    for k in range(4):
        target = g[f"target_{k}"]
        for sq in (True, False):
            for wp, wr in ((1.0, 1.0), (2.0, 0.5)):
                key = f"t{k}_sq{int(sq)}_w{wp}_{wr}"
                spec = panda_cost_spec(gs, robot, ee_target=np.eye(4, dtype=np.float32),
                                       ee_kw=dict(ee_w_pos=wp, ee_w_rot=wr, ee_square=sq))
                o = oracle_lib.Oracle(m, spec)
                H = o.fk(q, prec)
                c, gH = o.ee_cost(H[:, -1], target, prec)
                assert rel_err(c, g["cost_" + key]) < 5e-6, key
                gH_ref = g["gH_" + key][:, -1]
                assert rel_err(gH[:, :3, :], gH_ref[:, :3, :]) < 1e-5, key
                if target.ndim == 2:     # single target: also through the fused rollout
                    spec.ee_target = target
                    o.set_cost(spec)
                    _, c2, gq = o.rollout(q, (0, 0, 0, 1), prec)
                    assert rel_err(c2, g["cost_" + key]) < 5e-6, key
                    assert rel_err(gq, g["gq_" + key]) < 2e-5, key


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_rollout_bench_shapes(oracle_lib, prec):
    """BASELINE configs 2 (objects + EE) and 3 (self + objects + ws + EE) on a (6, 64, 7) batch."""
    g, robot = gold("rollout_panda"), gold("panda_robot")
    gs = gold("cost_spheres3d")
    spec = panda_cost_spec(gs, robot, ee_target=g["target"])
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), spec)
    q = g["q"].reshape(-1, 7)
    pos, c2, g2 = o.rollout(q, (0, 1, 0, 1), prec)
    assert np.abs(pos - g["pos"].reshape(-1, 11, 3)).max() < 2e-6
    assert rel_err(c2, g["cost_c2"].reshape(-1)) < 5e-6
    assert rel_err(g2, g["gq_c2"].reshape(-1, 7)) < 2e-5
    _, c3, g3 = o.rollout(q, (1, 1, 1, 1), prec)
    assert rel_err(c3, g["cost_c3"].reshape(-1)) < 5e-6
    assert rel_err(g3, g["gq_c3"].reshape(-1, 7)) < 2e-5


@pytest.mark.parametrize("robot", ["panda_arm_no_gripper", "panda_arm_hand", "allegro_hand"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_analytic_jacobian_all_links(oracle_lib, robot, prec):
    """A16: compute_analytical_jacobian_all_links (autograd through FK + rotation_matrix_to_q)."""
    g = gold(f"ajac_{robot}")
    o = oracle_lib.Oracle(model(robot))
    J = o.analytic_jacobian(g["q"], prec)
    assert J.shape == g["J"].shape
    assert np.abs(J - g["J"]).max() < 3e-6
    # an independent check of the restated derivative: central differences of the fp64 FK + quaternion
    if prec == "f64":
        m = o.model
        q = g["q"].astype(np.float64)
        lo, hi = m.lower[m.controlled].astype(np.float64), m.upper[m.controlled].astype(np.float64)
        cl = m.clamp[m.controlled].astype(bool)
        inside = np.all(~cl | ((q > lo + 1e-3) & (q < hi - 1e-3)), axis=1)
        h = 1e-6
        for d in range(m.n_dofs):
            e = np.zeros(m.n_dofs); e[d] = h
            Hp, Hm = o.fk(q + e, "f64"), o.fk(q - e, "f64")
            dpos = (Hp[..., :3, 3] - Hm[..., :3, 3]) / (2 * h)
            assert np.abs(dpos[inside] - J[inside][:, :, :3, d]).max() < 1e-6


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_ik_loss_gradient_and_adam_steps(oracle_lib, prec):
    """8f rank 2: loss_fn_ik_per_q / ik_termination and five torch.optim.Adam steps on it (robot_tree.py:345-442)."""
    g = gold("ik_panda")
    m = model("panda_arm_no_gripper")
    o = oracle_lib.Oracle(m)
    dt = np.float32 if prec == "f32" else np.float64
    for tag, Ht in (("per_sample", g["H_target"]), ("single", g["H_target"][0])):
        q = g["q0"].astype(dt).copy(); mom = np.zeros_like(q); vel = np.zeros_like(q)
        for it in range(5):
            loss, grad, valid = o.ik_step(10, Ht, g["lower"], g["upper"], q, mom, vel, it + 1, prec=prec)
            if it == 0:
                assert rel_err(loss, g[f"loss0_{tag}"]) < 5e-6
                assert rel_err(grad, g[f"grad0_{tag}"]) < 2e-5
                np.testing.assert_array_equal(valid, g[f"valid0_{tag}"])
            assert rel_err(loss, g[f"err_steps_{tag}"][it]) < 2e-4
            assert np.abs(q - g[f"q_steps_{tag}"][it]).max() < 2e-4       # Adam's first steps are +-lr whatever |g| is


# ---------------------------------------------------------------------------------------------------------------------
# points fixed in link frames (grasped-object points; Frame.transform_point)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("robot", ["ur10_allegro", "dual_panda", "hab_stretch"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_attached_points_on_trees(oracle_lib, robot, prec):
    g = gold(f"points_{robot}")
    o = oracle_lib.Oracle(model(robot))
    pos = o.fk_points(g["point_link"], g["point_offset"], g["q"], prec)
    assert np.abs(pos - g["pos"]).max() / max(1.0, float(np.abs(g["pos"]).max())) < TOL_H
    gq = o.fk_points_backward(g["point_link"], g["point_offset"], g["q"], g["w"], prec)
    assert rel_err(gq, g["gq"]) < TOL_G


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_grasped_object_fk_map_collision(oracle_lib, prec):
    """RobotPanda(grasped_object=GraspedObjectPandaBox).fk_map_collision: 12 link origins + 14 box points."""
    from helpers import grasp_panda_setup
    g = gold("grasp_panda")
    m, pl, po, spec = grasp_panda_setup()
    o = oracle_lib.Oracle(m, spec)
    for tag in ("", "_out"):
        q, ref = g["q" + tag].reshape(-1, 7), g["link_pos" + tag].reshape(-1, 26, 3)
        pos = o.fk_points(pl, po, q, prec)
        assert np.abs(pos - ref).max() < TOL_H
        gq = o.fk_points_backward(pl, po, q, g["w" + tag].reshape(-1, 26, 3), prec)
        assert rel_err(gq, g["gq" + tag].reshape(-1, 7)) < TOL_G


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_grasped_object_costs(oracle_lib, prec):
    """The three collision fields over robot links + grasped points (reference field code on explicit columns)."""
    from helpers import grasp_panda_setup
    g = gold("grasp_panda")
    m, pl, po, spec = grasp_panda_setup()
    o = oracle_lib.Oracle(m, spec)
    lp = g["link_pos"].reshape(-1, 26, 3)
    total_g = np.zeros_like(lp)
    for key, f in (("self", FIELD_SELF), ("obj", FIELD_OBJECTS), ("ws", FIELD_WS)):
        c, gl = o.cost_fields(f, lp, prec)
        assert rel_err(c, g[f"cost_{key}"].reshape(-1)) < 5e-6
        total_g += gl
        np.testing.assert_array_equal(o.collision_fields(f, lp, None, prec), g[f"coll_{key}"].reshape(-1))
        np.testing.assert_array_equal(o.collision_fields(f, lp, 0.0, prec), g[f"coll0_{key}"].reshape(-1))
    assert rel_err(total_g, g["g_link_pos"].reshape(-1, 26, 3)) < 2e-5
    pos, cost, gq = o.rollout_points(pl, po, g["q"].reshape(-1, 7), (1, 1, 1, 0), prec)
    assert np.abs(pos - lp).max() < TOL_H
    assert rel_err(cost, (g["cost_self"] + g["cost_obj"] + g["cost_ws"]).reshape(-1)) < 5e-6
    assert rel_err(gq, g["gq_cost"].reshape(-1, 7)) < 2e-5


def test_rollout_points_with_link_origins_equals_rollout(oracle_lib):
    """A point set {every link, zero offset} reduces the point rollout to the link rollout (incl. the EE term)."""
    g, robot = gold("rollout_panda"), gold("panda_robot")
    spec = panda_cost_spec(gold("cost_spheres3d"), robot, ee_target=g["target"])
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), spec)
    q = g["q"].reshape(-1, 7)[:64]
    ref = o.rollout(q, (1, 1, 1, 1), "f64")
    got = o.rollout_points(np.arange(11), np.zeros((11, 3)), q, (1, 1, 1, 1), "f64")
    for a, b in zip(ref, got):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------
# GP prior (BUILD-DEFINED term of BASELINE config 5; no reference counterpart -> parity unpinned).  The oracle is
# checked against an independent statement of the same cost in torch fp64 with autograd gradients.
# ---------------------------------------------------------------------------------------------------------------------
def _gp_prior_torch(q, qd, dt, sigma, w):
    import torch
    q, qd = torch.tensor(q, dtype=torch.float64, requires_grad=True), torch.tensor(qd, dtype=torch.float64, requires_grad=True)
    D = q.shape[-1]
    Phi = torch.eye(2 * D, dtype=torch.float64)
    Phi[:D, D:] = dt * torch.eye(D, dtype=torch.float64)
    Qc_inv = torch.eye(D, dtype=torch.float64) / sigma ** 2
    Qinv = torch.cat([torch.cat([12 / dt ** 3 * Qc_inv, -6 / dt ** 2 * Qc_inv], 1),
                      torch.cat([-6 / dt ** 2 * Qc_inv, 4 / dt * Qc_inv], 1)], 0)
    x = torch.cat([q, qd], -1)                                   # (B, H, 2D)
    e = x[:, :-1] @ Phi.T - x[:, 1:]
    cost = w * 0.5 * torch.einsum("bti,ij,btj->b", e, Qinv, e)
    gq, gqd = torch.autograd.grad(cost.sum(), (q, qd))
    return cost.detach().numpy(), gq.numpy(), gqd.numpy()


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_gp_prior_against_independent_autograd(oracle_lib, prec):
    rng = np.random.default_rng(21)
    for (B, H, D, dt, sigma, w) in ((3, 64, 7, 0.08, 0.1, 1.0), (2, 128, 14, 5.0 / 128, 0.5, 0.3), (4, 2, 3, 1.0, 1.0, 2.0),
                                    (2, 1, 5, 0.1, 0.2, 1.0)):
        q = np.cumsum(rng.standard_normal((B, H, D)) * 0.05, axis=1)
        qd = rng.standard_normal((B, H, D)) * 0.3
        rc, rgq, rgqd = _gp_prior_torch(q, qd, dt, sigma, w)
        c, gq, gqd = oracle_lib.gp_prior(q, qd, dt, sigma, w, prec)
        tol = 1e-12 if prec == "f64" else 2e-5
        assert rel_err(c, rc) < tol or np.abs(rc).max() == 0
        assert rel_err(gq, rgq) < tol or np.abs(rgq).max() == 0
        assert rel_err(gqd, rgqd) < tol or np.abs(rgqd).max() == 0


# ---------------------------------------------------------------------------------------------------------------------
# A17: finite differences and trajectory metrics
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_finite_differences_and_trajectory_metrics(oracle_lib, prec):
    g = gold("traj")
    for m in ("forward", "backward", "central"):
        out = oracle_lib.finite_difference(g["x"], 0.25, m, prec)
        np.testing.assert_allclose(out, g["fd_" + m], rtol=0, atol=0 if prec == "f32" else 2e-6)
    gm = gold("metrics_panda")
    vel = oracle_lib.finite_difference(gm["trajs"], 1.0, "central", prec)
    np.testing.assert_allclose(vel, gm["vel_fd"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(oracle_lib.finite_difference(vel, 1.0, "central", prec), gm["acc_fd"], rtol=0, atol=1e-7)
    assert rel_err(oracle_lib.traj_diff_norm_sum(gm["trajs"], 0, 7, prec), gm["path_length"]) < 2e-6
    assert rel_err(oracle_lib.traj_diff_norm_sum(vel, 0, 7, prec), gm["smoothness_fd"]) < 2e-6
    assert rel_err(oracle_lib.traj_diff_norm_sum(gm["full"], 7, 7, prec), gm["smoothness_vel"]) < 2e-6


# -----------------------------------------------------------------------------------------------
# Frame algebra (geometrics/frame.py:55-121)
# -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_frame_algebra_vs_reference(prec):
    """orc_frame_* against the reference's own Frame: inverse / multiply_transform / multiply_inv_transform (+ broadcast),
    transform_point, get_quaternion (XYZW trace method) and get_euler, values and gradients."""
    from oracle.oracle import Oracle as O
    g = gold("frame_algebra")
    tol = 0 if prec == "f32" else 1e-6                           # the fp32 restatement is bit-exact on the forward values
    for op, name, a, b in ((1, "inv", ("Ra", "ta"), None), (0, "mul", ("Ra", "ta"), ("Rb", "tb")),
                           (2, "mulinv", ("Ra", "ta"), ("Rb", "tb")), (0, "mul1", ("R1", "t1"), ("Rb", "tb"))):
        R, t = O.frame_compose(op, g[a[0]], g[a[1]], None if b is None else g[b[0]], None if b is None else g[b[1]], prec=prec)
        assert np.abs(R - g[f"{name}_R"]).max() <= tol and np.abs(t - g[f"{name}_t"]).max() <= tol, name
        if name != "mul1":
            grads = O.frame_compose_backward(op, g["Ra"], g["ta"], g["Rb"], g["tb"], g["wR"], g["wt"], prec=prec)
            for gk, key in zip(grads, ["gRa", "gta", "gRb", "gtb"] if op != 1 else ["gRa", "gta"]):
                assert np.abs(gk - g[f"{name}_{key}"]).max() < 1e-6, (name, key)
    out, gR, gt = O.frame_transform_points(g["Ra"], g["ta"], g["pts"], g["wp"], prec=prec)
    assert np.abs(out - g["tp"]).max() <= tol and np.abs(gR - g["tp_gRa"]).max() < 2e-6 and np.abs(gt - g["tp_gta"]).max() < 2e-6
    for k, nm in enumerate("xyz"):
        assert np.abs(O.rotation_from(k, g["angle"], prec) - g[f"rot_{nm}"]).max() < 1e-7
    assert np.abs(O.rotation_from(3, g["quat_in"], prec) - g["quat_R"]).max() <= tol
    quat, eul = O.frame_quat_euler(g["Ra"], prec=prec)
    assert np.abs(quat - g["quat_xyzw"]).max() < 2e-7 and np.abs(eul - g["euler"]).max() < 2e-7


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("name", ["spheres3d", "table_shelf", "spheres3d_tight"])
def test_clamp_sdf_costs(oracle_lib, name, prec):
    """clamp_sdf=True (distance_fields.py:114-117): relu(margin - sdf) per link / pair for all three fields -- per-field costs,
    position gradients, gradients w.r.t. q and the PlanningTask total against the reference's own fields."""
    from helpers import clamp_cost_spec
    g, robot = gold("cost_clamp"), gold("panda_robot")
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), clamp_cost_spec(name))
    pos = robot["fk_map_collision"].reshape(-1, 11, 3)
    for fname, fl, w in (("self", FIELD_SELF, (1, 0, 0, 0)), ("objects", FIELD_OBJECTS, (0, 1, 0, 0)), ("ws", FIELD_WS, (0, 0, 1, 0))):
        c, gp = o.cost_fields(fl, pos, prec)
        ref_c, ref_g = g[f"{name}_cost_{fname}"].reshape(-1), g[f"{name}_gpos_{fname}"].reshape(-1, 11, 3)
        assert np.abs(c - ref_c).max() < 1e-5 * max(1.0, np.abs(ref_c).max()), fname
        assert np.abs(gp - ref_g).max() < 1e-4 * max(1.0, np.abs(ref_g).max()), fname
        assert ((c == 0) == (ref_c == 0)).mean() > 0.98                  # the hinge is off for the same samples
        _, _, gq = o.rollout(g["q"].reshape(-1, 7), w, prec)
        ref_q = g[f"{name}_gq_{fname}"].reshape(-1, 7)
        assert np.abs(gq - ref_q).max() < 1e-4 * max(1.0, np.abs(ref_q).max()), fname
    _, c, gq = o.rollout(g["q"].reshape(-1, 7), (1, 1, 1, 0), prec)
    assert rel_err(c, g[f"{name}_cost_total"].reshape(-1)) < TOL_C
    assert rel_err(gq, g[f"{name}_gq_total"].reshape(-1, 7)) < TOL_G
    assert (g[f"{name}_cost_total"] >= 0).all() and (g[f"{name}_cost_total"] > 0).any()


def test_interpolation_table_is_atens(oracle_lib):
    """costmodel.interpolation_table restates F.interpolate(mode='linear', align_corners=True): against the reference's
    interpolate_points_v1 outputs and input gradients (goldens), and against torch itself here."""
    import torch
    import torch.nn.functional as F
    from torch_robotics_amd.costmodel import interpolation_table
    g = gold("cost_interp")
    for L, K in g["ip_shapes"]:
        src, w = interpolation_table(L, K)
        assert src.shape == (K, 2) and w.shape == (K, 2) and src.min() >= 0 and src.max() < L
        x, wout = g[f"ip_{L}_{K}_in"], g[f"ip_{L}_{K}_w"]
        out = x[:, src[:, 0]] * w[None, :, 0, None] + x[:, src[:, 1]] * w[None, :, 1, None]
        assert np.abs(out - g[f"ip_{L}_{K}_out"]).max() < 5e-7          # one ulp: ATen contracts w0 a + w1 b into an FMA
        gin = np.zeros_like(x)
        for k in range(K):
            gin[:, src[k, 0]] += w[k, 0] * wout[:, k]
            gin[:, src[k, 1]] += w[k, 1] * wout[:, k]
        assert np.abs(gin - g[f"ip_{L}_{K}_gin"]).max() < 2e-6
        t = F.interpolate(torch.from_numpy(x).transpose(-2, -1), size=int(K), mode="linear", align_corners=True).transpose(-2, -1)
        assert np.abs(out - t.numpy()).max() < 5e-7
    assert "RuntimeError" in str(g["flag_result"])      # the reference's own flag trips over its second indexing (documented)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_interpolated_link_points_fields(oracle_lib, prec):
    """interpolate_link_pos: the three fields on interpolated link points (virtual columns of the cost model) against the
    reference's field code fed interpolate_points_v1 of the selected links (goldens), value / d pos / d q / booleans."""
    from helpers import interp_cost_spec
    g, robot = gold("cost_interp"), gold("panda_robot")
    spec = interp_cost_spec()
    assert spec.n_columns == 11 + 15 + 16
    m = model("panda_arm_no_gripper")
    o = oracle_lib.Oracle(m, spec)
    pos = robot["fk_map_collision"].reshape(-1, 11, 3)
    tol_c, tol_g = (1e-5, 1e-4) if prec == "f32" else (2e-6, 2e-5)
    for fname, fl, w in (("self", FIELD_SELF, (1, 0, 0, 0)), ("objects", FIELD_OBJECTS, (0, 1, 0, 0)),
                         ("ws", FIELD_WS, (0, 0, 1, 0))):
        c, gp = o.cost_fields(fl, pos, prec)
        assert rel_err(c, g[f"cost_{fname}"].reshape(-1)) < tol_c, fname
        assert rel_err(gp, g[f"gpos_{fname}"].reshape(-1, 11, 3)) < tol_g, fname
        _, c2, gq = o.rollout(g["q"].reshape(-1, 7), w, prec)
        assert rel_err(c2, g[f"cost_{fname}"].reshape(-1)) < tol_c and rel_err(gq, g[f"gq_{fname}"].reshape(-1, 7)) < tol_g, fname
        assert np.array_equal(o.collision_fields(fl, pos, None, prec), g[f"coll_{fname}"].reshape(-1)), fname
        assert np.array_equal(o.collision_fields(fl, pos, 0.0, prec), g[f"coll0_{fname}"].reshape(-1)), fname
    _, c, gq = o.rollout(g["q"].reshape(-1, 7), (1, 1, 1, 0), prec)
    assert rel_err(c, g["cost_total"].reshape(-1)) < tol_c and rel_err(gq, g["gq_total"].reshape(-1, 7)) < tol_g


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_single_link_self_distance(oracle_lib, prec):
    """distance_fields.py:195-198: with one self-collision link the "distance" is |p|_1 * 1e9."""
    from helpers import single_link_self_spec
    g, robot = gold("cost_interp"), gold("panda_robot")
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), single_link_self_spec())
    pos = robot["fk_map_collision"].reshape(-1, 11, 3)
    c, gp = o.cost_fields(FIELD_SELF, pos, prec)
    assert rel_err(c, g["single_cost"].reshape(-1)) < 1e-6
    assert np.array_equal(gp, g["single_gpos"].reshape(-1, 11, 3))                # +-1e9 on one link, zeros elsewhere
    _, c2, gq = o.rollout(g["q"].reshape(-1, 7), (1, 0, 0, 0), prec)
    assert rel_err(c2, g["single_cost"].reshape(-1)) < 1e-6 and rel_err(gq, g["single_gq"].reshape(-1, 7)) < 1e-5
    assert np.array_equal(o.collision_fields(FIELD_SELF, pos, None, prec), g["single_coll"].reshape(-1))


@pytest.mark.parametrize("name", ["ur10_allegro", "dual_panda"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_tree_robot_costs_of_configs_4_and_5(oracle_lib, name, prec):
    """BASELINE configs 4 / 5 robots: self / object / workspace / EE costs and d/dq against the reference's field classes + autograd
    through its FK on the authored UR10 + Allegro and dual-Panda trees (oracle/gen_golden.py treecost)."""
    from helpers import tree_cost_spec
    m, spec, g = tree_cost_spec(name)
    o = oracle_lib.Oracle(m, spec)
    q = g["q"]
    tol_c, tol_g = (1e-5, 1e-4) if prec == "f32" else (3e-6, 3e-5)
    for fname, w in (("self", (1, 0, 0, 0)), ("objects", (0, 1, 0, 0)), ("ws", (0, 0, 1, 0)), ("ee", (0, 0, 0, 1)), ("total", (1, 1, 1, 1))):
        pos, c, gq = o.rollout(q, w, prec)
        assert rel_err(c, g[f"cost_{fname}"]) < tol_c, fname
        assert rel_err(gq, g[f"gq_{fname}"]) < tol_g, fname
    assert np.abs(pos - g["link_pos"]).max() < 2e-6
    for fname, fl in (("self", FIELD_SELF), ("objects", FIELD_OBJECTS), ("ws", FIELD_WS)):
        assert np.array_equal(o.collision_fields(fl, g["link_pos"], None, prec), g[f"coll_{fname}"]), fname
        assert np.array_equal(o.collision_fields(fl, g["link_pos"], 0.0, prec), g[f"coll0_{fname}"]), fname


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_gp_prior_against_dense_definition_golden(oracle_lib, prec):
    """The build-defined GP prior pinned a SECOND way (tests/golden/gp_prior.npz, oracle/gen_golden.py builddef): the dense
    constant-velocity definition -- Phi(dt), Q = sigma^2 [[dt^3/3, dt^2/2], [dt^2/2, dt]] (x) I inverted NUMERICALLY, torch fp64
    autograd -- against the oracle's closed-form factors: cost per trajectory, per factor, both gradients."""
    g = gold("gp_prior")
    for k in range(int(g["n_cases"])):
        q, qd = g[f"q_{k}"], g[f"qd_{k}"]
        dt, sigma, w = (float(v) for v in g[f"params_{k}"])
        c, gq, gqd = oracle_lib.gp_prior(q, qd, dt, sigma, w, prec)
        fc = oracle_lib.gp_factor_cost(q, qd, dt, sigma, w, prec)
        tol = 2e-5 if prec == "f32" else 1e-9
        big = max(np.abs(g[f"gq_{k}"]).max(), np.abs(g[f"gqd_{k}"]).max(), 1e-30)
        assert rel_err(c, g[f"cost_{k}"]) < tol and rel_err(fc, g[f"factor_{k}"]) < tol, k
        # a state's gradient is the difference of two factors' terms, each of the size of the largest entry
        assert np.abs(gq - g[f"gq_{k}"]).max() < tol * big and np.abs(gqd - g[f"gqd_{k}"]).max() < tol * big, k
        if q.shape[1] == 1:
            assert np.all(c == 0) and np.all(gq == 0) and np.all(gqd == 0)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_gauss_newton_step_against_reference_jacobian_golden(oracle_lib, prec):
    """The build-defined damped Gauss-Newton IK step pinned a SECOND way (tests/golden/ik_gn_panda.npz): the REFERENCE's stateful FK +
    geometric Jacobian (robot_tree.py:218-248, fp32), scipy's rotation vector, `torch.linalg.solve` in fp64, the reference's
    SE3_distance as the termination metric -- against orc_ik_gn_step (own FK walk, own quaternion log, own Cholesky)."""
    g = gold("ik_gn_panda")
    m = model("panda_arm_no_gripper")
    o = oracle_lib.Oracle(m)
    ee = m.name_to_idx[str(g["link"])]
    for tag in ("a", "b"):
        damping, lm_gain, step = (float(v) for v in g[f"params_{tag}"])
        q_new, err = o.ik_gn_step(ee, g["H_target"], g["lower"], g["upper"], g["q0"], damping, lm_gain, step, prec)
        dq = np.abs(g[f"q_new_{tag}"] - g["q0"])
        # the golden's Jacobian and pose are the reference's fp32 numbers: the step agrees to fp32 accuracy x the system's conditioning
        assert (np.abs(q_new - g[f"q_new_{tag}"]) <= 2e-5 + 2e-3 * dq).all(), tag
        assert rel_err(err, g[f"err_{tag}"]) < 2e-5, tag
    assert (np.abs(g["q_new_a"] - g["q0"]).max(1) > 1e-2).sum() >= 30           # the steps are real steps, not clamped no-ops
