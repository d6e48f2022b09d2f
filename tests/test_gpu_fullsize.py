"""Parity at BASELINE.json's FULL sizes through size-independent properties of the domain (the oracle finishes only small
cases in seconds): linearity of the objective in its weights, rigid-motion equivariance of FK and the costs, a checksum of
checksums against the deterministic reduction, sharding invariance (what the N-GPU batch split relies on), a directional
finite-difference check of the gradient summed over the whole batch, monotonicity of the boolean fields in the margin --
plus an fp64-oracle spot check of a random subset of the full-size outputs.

configs[1]: Panda 4096 x 64, obstacles + EE      configs[2]: Panda, one GPU's 4096 x 64 share of 32768 x 64, all four terms
configs[3]: UR10 + Allegro 4096 x 64             configs[4]: dual Panda, one GPU's 2048 x 128 share, fp16 I/O + GP prior
"""
import numpy as np
import pytest
import torch

import torch_robotics_amd as tra
from helpers import grad_close, rel_err
from torch_robotics_amd import codegen
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
from torch_robotics_amd.costmodel import CostModelSpec
from torch_robotics_amd.kinmodel import quat_wxyz_to_rot

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TA = dict(device=DEV, dtype=torch.float32)


@pytest.fixture(scope="module")
def ops():
    from torch_robotics_amd import ops as o
    return o


@pytest.fixture(scope="module")
def oracle_lib():
    from oracle import oracle as orc
    return orc


def panda_task(all_terms):
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(Ht)
    gen = torch.Generator(device=DEV).manual_seed(1234)
    q = robot.random_q(4096 * 64, generator=gen).reshape(4096, 64, 7).contiguous()
    return robot, task, q, ((1.0, 1.0, 1.0, 1.0) if all_terms else (0.0, 1.0, 0.0, 1.0))


@pytest.mark.parametrize("all_terms", [False, True], ids=["config2_obj_ee", "config3_all_terms"])
def test_panda_full_size_properties(ops, oracle_lib, all_terms):
    robot, task, q, w = panda_task(all_terms)
    model, cm = task._fused_handles(DEV)
    assert model.specialized
    B, H = 4096, 64
    n = B * H
    sums = torch.zeros(ops.n_blocks(n), **TA)
    pos, cost, gq = ops.rollout_cost_grad(model, cm, w, q, cost_sum=sums)
    assert pos.shape == (B, H, 11, 3) and cost.shape == (B, H) and gq.shape == (B, H, 7)
    assert torch.isfinite(pos).all() and torch.isfinite(cost).all() and torch.isfinite(gq).all()

    # idempotence / determinism: a second evaluation gives the same bits
    pos2, cost2, gq2 = ops.rollout_cost_grad(model, cm, w, q)
    assert torch.equal(pos, pos2) and torch.equal(cost, cost2) and torch.equal(gq, gq2)

    # checksum of checksums: per-wavefront sums (= per-trajectory costs at horizon 64) and their deterministic total
    per_traj = cost.double().sum(1)
    np.testing.assert_allclose(sums.double().cpu().numpy(), per_traj.cpu().numpy(), rtol=3e-6, atol=1e-5)
    total = ops.reduce_sum(sums).item()
    assert abs(total - per_traj.sum().item()) <= 2e-6 * per_traj.abs().sum().item()
    assert ops.reduce_sum(sums).item() == total                 # fixed association order

    # linearity in the weights: cost(w) = sum_i w_i cost(e_i), same for the gradient
    acc_c, acc_g = torch.zeros_like(cost, dtype=torch.float64), torch.zeros_like(gq, dtype=torch.float64)
    coef = (0.5, 2.0, 0.25, 3.0)
    for i in range(4):
        e = [0.0] * 4
        e[i] = 1.0
        if not all_terms and i in (0, 2):
            continue
        _, ci, gi = ops.rollout_cost_grad(model, cm, e, q, want_pos=False)
        acc_c += coef[i] * ci.double(); acc_g += coef[i] * gi.double()
    wmix = tuple(c if (all_terms or i in (1, 3)) else 0.0 for i, c in enumerate(coef))
    _, cm_, gm_ = ops.rollout_cost_grad(model, cm, wmix, q, want_pos=False)
    assert rel_err(cm_.cpu().numpy(), acc_c.cpu().numpy()) < 1e-5
    assert rel_err(gm_.cpu().numpy(), acc_g.cpu().numpy()) < 2e-5

    # sharding invariance: the two halves of the batch evaluated separately == the full batch, bit for bit
    for lo, hi in ((0, B // 2), (B // 2, B), (1000, 1003)):
        p_s, c_s, g_s = ops.rollout_cost_grad(model, cm, w, q[lo:hi].contiguous())
        assert torch.equal(p_s, pos[lo:hi]) and torch.equal(c_s, cost[lo:hi]) and torch.equal(g_s, gq[lo:hi])

    # gradient: directional finite difference, summed over the whole batch (kinks at arg-min switches / joint limits are
    # measure-zero events; the batch sum averages them out)
    gen = torch.Generator(device=DEV).manual_seed(7)
    dq = torch.randn(q.shape, generator=gen, **TA)
    eps = 1e-3
    lo_, hi_ = robot.q_min.to(DEV), robot.q_max.to(DEV)
    inside = ((q - eps * dq.abs() > lo_) & (q + eps * dq.abs() < hi_)).all(-1)          # stay off the clamps
    _, cp, _ = ops.rollout_cost_grad(model, cm, w, q + eps * dq, want_pos=False)
    _, cn, _ = ops.rollout_cost_grad(model, cm, w, q - eps * dq, want_pos=False)
    fd_i = ((cp.double() - cn.double()) / (2 * eps))[inside]
    an_i = (gq.double() * dq.double()).sum(-1)[inside]
    # per sample: equal up to the fp32 rounding of the two costs (~1e-7 * cost / eps) except where the +-eps segment crosses
    # a kink of the objective (arg-min switch between spheres / planes: a small fraction of the samples)
    tol_i = 2e-3 * an_i.abs() + 2e-3 * (1.0 + cp[inside].abs().double())
    frac_bad = ((fd_i - an_i).abs() > tol_i).double().mean().item()
    assert frac_bad < (0.02 if all_terms else 0.01), frac_bad
    # whole batch: the kink errors have random sign, so the sums agree to a small fraction of sum |terms|
    assert abs(fd_i.sum().item() - an_i.sum().item()) <= 3e-5 * an_i.abs().sum().item(), (fd_i.sum().item(), an_i.sum().item())

    # rigid-motion equivariance: base pose T on the robot + the same T on scene, workspace-free terms and EE target
    if not all_terms:                                           # the axis-aligned workspace box is not rotation invariant
        pose = np.array([0.3, -0.2, 0.1, 0.9238795, 0.0, 0.0, 0.3826834], np.float32)     # 45 deg about z
        R, t = quat_wxyz_to_rot(pose[3:]), pose[:3]
        robot2 = tra.RobotPanda(tensor_args=TA)
        robot2.diff_panda.update_base_pose(torch.from_numpy(pose))
        env2 = tra.EnvSpheres3D(tensor_args=TA)
        env2.obj_fixed_list[0].set_position_orientation(pos=t, ori=pose[3:])
        task2 = tra.PlanningTask(env=env2, robot=robot2, obstacle_cutoff_margin=0.03, tensor_args=TA)
        Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = t
        task2.set_ee_target(T @ Ht)
        m2, c2 = task2._fused_handles(DEV)
        p_b, c_b, g_b = ops.rollout_cost_grad(m2, c2, w, q)
        want = pos.reshape(-1, 3) @ torch.from_numpy(R.T).to(DEV) + torch.from_numpy(t).to(DEV)
        assert (p_b.reshape(-1, 3) - want).abs().max().item() < 3e-6
        assert rel_err(c_b.cpu().numpy(), cost.cpu().numpy()) < 2e-5
        # the gradient jumps where two spheres are equidistant from a link (arg-min switch): a rotated scene rounds such a
        # near-tie the other way for a handful of the 262 144 samples; everywhere else the gradients agree
        bad = ((g_b - gq).abs().amax(-1) > 2e-4 * gq.abs().max()).sum().item()
        assert bad <= n // 20000, bad

    # boolean fields at full size: fused == positions + field kernel; monotone in the margin; consistent with the cost's sign
    fields = FIELD_OBJECTS | FIELD_WS | (FIELD_SELF if all_terms else 0)
    c_def = ops.rollout_collision(model, cm, fields, q)
    assert c_def.shape == (B, H)
    two_step = ops.collision_fields(cm, fields, pos.reshape(-1, 11, 3)).reshape(B, H).bool()
    assert (c_def != two_step).sum().item() <= 2                # last-ulp FK differences can only matter at a margin
    c0, c1 = ops.rollout_collision(model, cm, fields, q, margin=0.0), ops.rollout_collision(model, cm, fields, q, margin=0.05)
    assert not (c0 & ~c1).any()                                 # in collision at margin 0  =>  in collision at margin 0.05
    assert 0 < int(c0.sum()) < n

    # fp64 oracle on a random subset of the full-size outputs
    o = oracle_lib.Oracle(robot.diff_panda._kin, task.build_cost_spec())
    idx = np.random.default_rng(3).choice(n, 2048, replace=False)
    q_h = q.reshape(-1, 7)[idx].cpu().numpy()
    p64, c64, g64 = o.rollout(q_h.astype(np.float64), w, "f64")
    assert np.abs(pos.reshape(-1, 11, 3)[idx].cpu().numpy() - p64).max() < 2e-6
    assert rel_err(cost.reshape(-1)[idx].cpu().numpy(), c64) < 1e-5
    assert grad_close(gq.reshape(-1, 7)[idx].cpu().numpy(), g64)


def tree_setup(ident):
    kin, tmpl = codegen.template_for(ident)
    env = tra.EnvSpheres3D(tensor_args=TA)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.13, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = tmpl.ee_link
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5); spec.ee_target = Ht
    if tmpl.ee2_link >= 0:
        spec.ee2_link = tmpl.ee2_link
        Ht2 = np.eye(4, dtype=np.float32); Ht2[:3, 3] = (0.4, -0.3, 0.5); spec.ee2_target = Ht2
    return kin, spec


def test_ur10_allegro_full_size(ops, oracle_lib):
    """configs[3]: UR10 + Allegro (30 links, 22 DOF), 4096 x 64, FK + obstacles + EE + gradient and the geometric Jacobian."""
    kin, spec = tree_setup("ur10_allegro")
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, DEV)
    assert h.specialized
    B, H, D, L = 4096, 64, kin.n_dofs, kin.n_links
    gen = torch.Generator(device=DEV).manual_seed(5)
    q = ((torch.rand(B, H, D, generator=gen, **TA) - 0.5) * 3.0).contiguous()
    w = (0.0, 1.0, 0.0, 1.0)
    pos, cost, gq = ops.rollout_cost_grad(h, cm, w, q)
    assert pos.shape == (B, H, L, 3) and torch.isfinite(cost).all() and torch.isfinite(gq).all()
    p2, c2, g2 = ops.rollout_cost_grad(h, cm, w, q[B // 2:].contiguous())                 # sharding invariance
    assert torch.equal(p2, pos[B // 2:]) and torch.equal(c2, cost[B // 2:]) and torch.equal(g2, gq[B // 2:])
    # generated == table-driven on a slice (both full-size launches would only repeat the small-size parity tests)
    h.enable_specialized(False)
    pt, ct, gt = ops.rollout_cost_grad(h, cm, w, q[:64].contiguous())
    h.enable_specialized(True)
    assert (pt - pos[:64]).abs().max().item() < 3e-6 and rel_err(ct.cpu().numpy(), cost[:64].cpu().numpy()) < 1e-5
    assert rel_err(gt.cpu().numpy(), gq[:64].cpu().numpy()) < 1e-4
    # geometric Jacobian of ee_link: J_lin * dq == directional derivative of the position (finite difference), whole batch
    ee = kin.name_to_idx["ee_link"]
    qf = q.reshape(-1, D)
    p_ee, quat, lin, ang = ops.fk_jacobian(h, qf, None, ee)[:4]
    assert (p_ee - pos.reshape(-1, L, 3)[:, ee]).abs().max().item() < 3e-6
    dq = torch.randn(qf.shape, generator=gen, **TA)
    eps = 1e-3
    lo, hi = torch.as_tensor(kin.lower[kin.controlled], **TA), torch.as_tensor(kin.upper[kin.controlled], **TA)
    inside = ((qf - eps > lo) & (qf + eps < hi)).all(-1)
    pp = ops.fk_positions(h, qf + eps * dq, [ee])[:, 0]
    pn = ops.fk_positions(h, qf - eps * dq, [ee])[:, 0]
    fd = (pp - pn) / (2 * eps)
    an = torch.einsum("nkd,nd->nk", lin, dq)
    err = (fd - an)[inside].abs().max().item()
    assert err < 5e-3 * max(1.0, an[inside].abs().max().item()), err
    # oracle subset
    o = oracle_lib.Oracle(kin, spec)
    idx = np.random.default_rng(4).choice(B * H, 512, replace=False)
    p64, c64, g64 = o.rollout(qf[idx].cpu().numpy().astype(np.float64), w, "f64")
    assert np.abs(pos.reshape(-1, L, 3)[idx].cpu().numpy() - p64).max() < 3e-6
    assert rel_err(cost.reshape(-1)[idx].cpu().numpy(), c64) < 1e-5 and grad_close(gq.reshape(-1, D)[idx].cpu().numpy(), g64)
    # THE launch `bench.py --config c4` times: rollout + Jacobian in one kernel (trk_rollout_jacobian_cost_grad).  At this size the
    # launch's working set (287 MB) exceeds the Infinity Cache and the Jacobian tiles leave as non-temporal stores (SpecArgs::jac_stream).
    plan = ops.RolloutJacobianPlan(h, cm, w, q, ee)
    plan.lin_jac.fill_(7.0); plan.ang_jac.fill_(7.0)                # the structural zeros are written, not left over
    sums = torch.zeros(ops.n_blocks(B * H), **TA)
    plan.launch(sums.data_ptr())
    torch.cuda.synchronize()
    assert ops.last_dispatch() == "generated"
    assert (plan.link_pos - pos).abs().max().item() <= 3e-6 and rel_err(plan.cost.cpu().numpy(), cost.cpu().numpy()) < 1e-5
    assert rel_err(plan.gq.cpu().numpy(), gq.cpu().numpy()) < 1e-4
    assert abs(float(sums.double().sum()) - float(cost.double().sum())) <= 1e-5 * float(cost.double().abs().sum())
    assert (plan.pos.reshape(-1, 3) - p_ee).abs().max().item() <= 3e-6
    assert (plan.lin_jac.reshape(lin.shape) - lin).abs().max().item() <= 4e-6 and (plan.ang_jac.reshape(ang.shape) - ang).abs().max().item() <= 2e-6
    dqt = torch.minimum((plan.quat.reshape(-1, 4) - quat).abs().amax(1), (plan.quat.reshape(-1, 4) + quat).abs().amax(1))
    assert dqt.max().item() <= 5e-6
    rp, _, rl, ra, _, _ = o.jacobian(qf[idx].cpu().numpy().astype(np.float64), None, ee, "f64")
    assert np.abs(plan.pos.reshape(-1, 3)[idx].cpu().numpy() - rp).max() <= 3e-6
    assert np.abs(plan.lin_jac.reshape(-1, 3, D)[idx].cpu().numpy() - rl).max() <= 6e-6
    assert np.abs(plan.ang_jac.reshape(-1, 3, D)[idx].cpu().numpy() - ra).max() <= 3e-6
    # the store policy must not change a bit: the first 64 trajectories as a launch of their own (in cache: write-through tiles)
    small = ops.RolloutJacobianPlan(h, cm, w, q[:64].contiguous(), ee)
    small.launch()
    torch.cuda.synchronize()
    n64 = 64 * H
    for a, b in ((small.link_pos, plan.link_pos), (small.cost, plan.cost), (small.gq, plan.gq), (small.pos, plan.pos), (small.quat, plan.quat),
                 (small.lin_jac, plan.lin_jac), (small.ang_jac, plan.ang_jac)):
        assert torch.equal(a.reshape(n64, -1), b.reshape(B * H, -1)[:n64])


def test_stream_store_instantiation_gives_the_same_bits(ops):
    """Launches whose working set exceeds the Infinity Cache take the F32Stream instantiation of the fused rollout (non-temporal output
    stores, chosen by spec_stream_stores from the launch's bytes): the cache policy of a store must not change a single bit.  Forced on
    and off at a small size, and crossed by size at 24576 x 64 (302 MB) against the write-through kernel."""
    import os
    kin, tmpl = codegen.template_for("panda")
    from helpers import gold, panda_cost_spec
    spec = panda_cost_spec(gold("cost_spheres3d"), gold("panda_robot"), ee_target=gold("rollout_panda")["target"])
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    w = (1.0, 1.0, 1.0, 1.0)
    try:
        for B, H in ((37, 64), (24576, 64)):
            q = (torch.rand(B, H, 7, generator=gen, **TA) - 0.5) * 5.0
            res = {}
            for mode in ("0", "1"):
                os.environ["TRK_STREAM_STORES"] = mode
                pos, cost, gq = ops.rollout_cost_grad(h, cm, w, q)
                torch.cuda.synchronize()
                res[mode] = (pos.clone(), cost.clone(), gq.clone())
            for a, b in zip(res["0"], res["1"]):
                assert torch.equal(a, b)
            del os.environ["TRK_STREAM_STORES"]
            pos, cost, gq = ops.rollout_cost_grad(h, cm, w, q)          # the launch's own choice (by bytes)
            assert torch.equal(pos, res["0"][0]) and torch.equal(cost, res["0"][1]) and torch.equal(gq, res["0"][2])
    finally:
        os.environ.pop("TRK_STREAM_STORES", None)


def test_dual_panda_fp16_gp_full_size(ops, oracle_lib):
    """configs[4]: dual Panda, one GPU's 2048 x 128 share of the 8192 x 128 batch, fp16 q / link positions / gradient in HBM,
    fp32 arithmetic and cost, GP prior accumulated into the same gradient."""
    kin, spec = tree_setup("dual_panda")
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, DEV)
    B, H, D, L = 2048, 128, kin.n_dofs, kin.n_links
    gen = torch.Generator(device=DEV).manual_seed(9)
    q = (torch.cumsum(torch.randn(B, H, D, generator=gen, **TA) * 0.02, 1) + (torch.rand(B, 1, D, generator=gen, **TA) - 0.5) * 2.0)
    qd = torch.randn(B, H, D, generator=gen, **TA) * 0.1
    qh, qdh = q.half().contiguous(), qd.half().contiguous()
    w = (0.0, 1.0, 0.0, 1.0)
    pos, cost, gq = ops.rollout_cost_grad(h, cm, w, qh)
    assert pos.dtype == torch.float16 and gq.dtype == torch.float16 and cost.dtype == torch.float32
    assert pos.shape == (B, H, L, 3) and torch.isfinite(cost).all() and torch.isfinite(gq.float()).all()
    # fp32 I/O on the fp16-rounded inputs: the outputs differ by one fp16 rounding only, the cost not at all beyond fp32 noise
    p32, c32, g32 = ops.rollout_cost_grad(h, cm, w, qh.float())
    assert rel_err(cost.cpu().numpy(), c32.cpu().numpy()) < 2e-6
    assert (pos.float() - p32).abs().max().item() <= 2.0 ** -11 * max(1.0, p32.abs().max().item()) * 1.01
    assert rel_err(gq.float().cpu().numpy(), g32.cpu().numpy()) < 2.0 ** -10
    # The objective of config 5 at its own parameters: sigma_gp = 0.1 (env_spheres_3d.py:57), dt = 5 / 128.  a = 12 / (sigma^2 dt^3)
    # = 2e7: the GP gradient reaches 1e5 .. 1e6, beyond the fp16 range -- both kernels write grad_scale x gradient (the loss scale
    # from the trajectories' bounds), and EVERY element of the sum is finite (no mask).
    dt, sg = 5.0 / H, 0.1
    gs = ops.gp_grad_scale(dt, sg, 1.0, float(qh.abs().max()), float(qdh.abs().max()), extra=float(g32.abs().max()))
    _, _, gq_s = ops.rollout_cost_grad(h, cm, w, qh, want_pos=False, grad_scale=gs)
    gqd = torch.zeros_like(gq_s)
    acc = gq_s.clone()
    c_gp, _, _ = ops.gp_prior_cost_grad(qh, qdh, dt, sg, 1.0, accumulate_into=(acc, gqd), grad_scale=gs)
    c_gp2, g_gp, gd_gp = ops.gp_prior_cost_grad(qh, qdh, dt, sg, 1.0, grad_scale=gs)
    assert torch.equal(c_gp, c_gp2) and c_gp.shape == (B,) and torch.equal(gd_gp, gqd)
    acc2, gqd2 = gq_s.clone(), torch.zeros_like(gq_s)                   # the pre-bound launch gives the same bits
    gp_plan = ops.GPPriorPlan(qh, qdh, dt, sg, 1.0, accumulate_into=(acc2, gqd2), grad_scale=gs)
    gp_plan.launch()
    assert torch.equal(gp_plan.cost, c_gp) and torch.equal(acc2, acc) and torch.equal(gqd2, gqd)
    for t_ in (gq_s, acc, gqd, g_gp):
        assert torch.isfinite(t_.float()).all()                          # 100 % finite
    assert float(acc.float().abs().max()) < 65504.0 and float(gqd.float().abs().max()) < 65504.0      # and nothing saturated
    ref = gq_s.float() + g_gp.float()
    assert ((acc.float() - ref).abs() <= 2.0 ** -10 * ref.abs() + 2.0 ** -24).all()      # fp16(a + b) of two fp16 numbers: one rounding
    # fp64 oracle on 8 whole trajectories (1024 samples; the oracle sees the same fp16-rounded q, qd): GP cost, and the UNSCALED
    # total gradient to one fp16 rounding of each stored term
    tsel = np.random.default_rng(8).choice(B, 8, replace=False)
    q8, qd8 = qh[tsel].cpu().numpy().astype(np.float64), qdh[tsel].cpu().numpy().astype(np.float64)
    rc_gp, rg_gp, rgd_gp = oracle_lib.gp_prior(q8, qd8, dt, sg, 1.0, "f64")
    o8 = oracle_lib.Oracle(kin, spec)
    _, _, rg_ro = o8.rollout(q8.reshape(-1, D), w, "f64")
    assert rel_err(c_gp[tsel].cpu().numpy(), rc_gp) < 2e-5
    tot = rg_gp + rg_ro.reshape(8, H, D)
    got = acc[tsel].double().cpu().numpy() / gs
    assert (np.abs(got - tot) <= 2.0 ** -9 * (np.abs(rg_gp) + np.abs(rg_ro.reshape(8, H, D))) + 2.0 ** -22 / gs + 4e-5 * np.abs(rg_gp).max()).all()
    gotd = gqd[tsel].double().cpu().numpy() / gs
    assert (np.abs(gotd - rgd_gp) <= 2.0 ** -10 * np.abs(rgd_gp) + 2.0 ** -24 / gs + 2e-5 * np.abs(rgd_gp).max()).all()
    # the mixed mode (fp16 trajectories and positions, fp32 gradients, no scale): fp32 accuracy on the same subset
    _, _, gq_m = ops.rollout_cost_grad(h, cm, w, qh, want_pos=False, grad_dtype=torch.float32)
    gqd_m = torch.zeros_like(gq_m)
    ops.gp_prior_cost_grad(qh, qdh, dt, sg, 1.0, accumulate_into=(gq_m, gqd_m))
    assert gq_m.dtype == torch.float32 and rel_err(gq_m[tsel].cpu().numpy(), tot) < 1e-4 and rel_err(gqd_m[tsel].cpu().numpy(), rgd_gp) < 1e-4
    # THE KERNEL bench.py --config c5 TIMES -- the fused one-launch form (trk_rollout_gp_cost_grad -> k_rollout_gpt) -- at this very size
    # against the fp64 oracle on the same 8 trajectories: cost per sample (rollout + GP factor), positions, both UNSCALED gradients; loss-
    # scaled fp16 gradients and the mixed mode (fp32 gradients).  (Until round 5 the fused op met the oracle only at B <= 70.)
    assert h.specialized
    rp8, rc8, _ = o8.rollout(q8.reshape(-1, D), w, "f64")
    fc8 = oracle_lib.gp_factor_cost(q8, qd8, dt, sg, 1.0, "f64")
    ref_c8 = rc8.reshape(8, H) + fc8
    for gdt, gsc in ((None, gs), (torch.float32, 1.0)):
        sums = torch.zeros(ops.n_blocks(B * H), device=DEV)
        fpos, fcost, fgq, fgqd = ops.rollout_gp_cost_grad(h, cm, w, qh, qdh, dt, sg, 1.0, cost_sum=sums, grad_dtype=gdt, grad_scale=gsc)
        assert fpos.dtype == torch.float16 and fcost.dtype == torch.float32 and fgq.dtype == (gdt or torch.float16) and fgqd.dtype == fgq.dtype
        for t_ in (fcost, fgq, fgqd, fpos):
            assert torch.isfinite(t_.float()).all()
        if gdt is None:                    # fp16 gradients: the loss scale keeps every element below the saturation value
            assert float(fgq.float().abs().max()) < 65504.0 and float(fgqd.float().abs().max()) < 65504.0
        assert rel_err(fcost[tsel].cpu().numpy(), ref_c8) < 2e-5
        assert np.abs(fpos[tsel].float().cpu().numpy().reshape(-1, L, 3) - rp8).max() < 2.0 ** -10 * max(1.0, np.abs(rp8).max())
        # whole-batch consistency with the two-launch form: the per-trajectory cost and the per-wavefront sums (a checksum of checksums)
        two = cost.double().sum(1) + c_gp.double()
        assert rel_err(fcost.double().sum(1).cpu().numpy(), two.cpu().numpy()) < 1e-5
        assert rel_err(sums.double().reshape(B, H // 64).sum(1).cpu().numpy(), fcost.double().sum(1).cpu().numpy()) < 1e-5
        got, gotd = fgq[tsel].double().cpu().numpy() / gsc, fgqd[tsel].double().cpu().numpy() / gsc
        if gdt is None:       # ONE fp16 rounding of the scaled sum (+ fp32 arithmetic on terms of the prior's size)
            assert (np.abs(got - tot) <= 2.0 ** -9 * (np.abs(tot) + np.abs(rg_gp).max() * 2.0 ** -11) + 2.0 ** -22 / gsc + 4e-5 * np.abs(tot).max()).all()
            assert (np.abs(gotd - rgd_gp) <= 2.0 ** -9 * (np.abs(rgd_gp) + np.abs(rgd_gp).max() * 2.0 ** -11) + 2.0 ** -22 / gsc + 4e-5 * np.abs(rgd_gp).max()).all()
            # the collision / EE component under the loss scale: at gs ~ 2^-13 the O(1) collision gradient is ~1e-4 in the
            # scaled fp16 domain -- it is part of the SUM that is rounded once, so what is left of it is bounded by the sum's ulp:
            # |(total - GP) - collision| <= ulp_fp16(scaled total) / gs, element by element
            coll = rg_ro.reshape(8, H, D)
            ulp = np.maximum(2.0 ** (np.floor(np.log2(np.maximum(np.abs(tot) * gsc, 2.0 ** -24))) - 10), 2.0 ** -24) / gsc
            assert (np.abs((got - rg_gp) - coll) <= 1.01 * ulp + 4e-5 * np.abs(rg_gp).max()).all()
        else:
            assert rel_err(got, tot) < 1e-4 and rel_err(gotd, rgd_gp) < 1e-4
            # the collision / EE component inside the fp32 sum: a few fp32 ulps of the prior's terms (which are 1e5 x larger)
            assert (np.abs((got - rg_gp) - rg_ro.reshape(8, H, D)) <= 4e-7 * np.abs(tot) + 2e-6 * np.abs(rg_gp).max()).all()
    # the two generated schedules of the fused launch at full size (round 6): one ARM per lane (k_rollout_gpa, forced here; the default
    # for fp32 I/O) against one ROBOT per lane (what just ran) -- the same positions, gradients within one rounding of the stored type
    import os
    os.environ["TRK_GP_ARM_LANES"] = "1"
    try:
        sums_a = torch.zeros(ops.n_blocks(B * H), device=DEV)
        apos, acost, agq, agqd = ops.rollout_gp_cost_grad(h, cm, w, qh, qdh, dt, sg, 1.0, cost_sum=sums_a, grad_dtype=torch.float32)
    finally:
        os.environ.pop("TRK_GP_ARM_LANES", None)
    assert torch.equal(apos, fpos) and rel_err(acost.cpu().numpy(), fcost.cpu().numpy()) < 2e-6
    assert rel_err(sums_a.cpu().numpy(), sums.cpu().numpy()) < 2e-6
    assert rel_err(agq.cpu().numpy(), fgq.cpu().numpy()) < 2e-6 and rel_err(agqd.cpu().numpy(), fgqd.cpu().numpy()) < 2e-6
    # sharding invariance of the fused launch: the second half of the batch alone gives the same bits
    _, c_h, g_h, gd_h = ops.rollout_gp_cost_grad(h, cm, w, qh[B // 2:].contiguous(), qdh[B // 2:].contiguous(), dt, sg, 1.0, want_pos=False,
                                                  grad_dtype=torch.float32)
    assert torch.equal(c_h, fcost[B // 2:]) and torch.equal(g_h, fgq[B // 2:]) and torch.equal(gd_h, fgqd[B // 2:])
    # sharding invariance of both kernels (whole trajectories stay on one rank)
    _, c_s, g_s = ops.rollout_cost_grad(h, cm, w, qh[B // 2:].contiguous(), want_pos=False)
    assert torch.equal(c_s, cost[B // 2:]) and torch.equal(g_s, gq[B // 2:])
    assert torch.equal(ops.gp_prior_cost_grad(qh[:7].contiguous(), qdh[:7].contiguous(), dt, sg, 1.0, grad_scale=gs)[0], c_gp[:7])
    # oracle subset (the oracle sees the same fp16-rounded q)
    o = oracle_lib.Oracle(kin, spec)
    idx = np.random.default_rng(6).choice(B * H, 512, replace=False)
    p64, c64, g64 = o.rollout(qh.reshape(-1, D)[idx].float().cpu().numpy().astype(np.float64), w, "f64")
    assert rel_err(cost.reshape(-1)[idx].cpu().numpy(), c64) < 1e-5
    assert np.abs(pos.reshape(-1, L, 3)[idx].float().cpu().numpy() - p64).max() < 2.0 ** -10 * max(1.0, np.abs(p64).max())


@pytest.mark.parametrize("ident", ["panda", "dual_panda"])
def test_generated_fk_family_full_size(ops, oracle_lib, ident):
    """The generated FK family at 4096 x 64 -- all-links matrices (k_fkh), their reverse mode (k_fkhbwd), one link (k_fk1), the
    geometric Jacobian (k_jac), positions and their reverse mode: consistency properties over the WHOLE batch (orthonormal
    rotations, constant bottom rows, matrices == positions, single link == that link of the full result, linearity of the reverse
    modes in the adjoint, generated == table-driven to fp32 rounding) and an fp64-oracle spot check of a random subset."""
    kin, _ = codegen.template_for(ident)
    h, o = ops.ModelHandle(kin), oracle_lib.Oracle(kin)
    assert h.specialized
    n, L, D = 4096 * 64, kin.n_links, kin.n_dofs
    gen = torch.Generator(device=DEV).manual_seed(7)
    q = (torch.rand(n, D, generator=gen, **TA) - 0.5) * 5.0
    Hm = ops.fk_forward(h, q)
    assert Hm.shape == (n, L, 4, 4) and torch.isfinite(Hm).all()
    R = Hm[..., :3, :3]
    eye = torch.eye(3, **TA)
    assert float((R @ R.transpose(-1, -2) - eye).abs().max()) < 2e-6            # every rotation of every sample is orthonormal
    assert bool((Hm[..., 3, :] == torch.tensor([0.0, 0.0, 0.0, 1.0], **TA)).all())
    pos = ops.fk_positions(h, q)
    assert float((pos - Hm[..., :3, 3]).abs().max()) < 4e-6
    li = L - 1
    assert float((ops.fk_forward(h, q, [li])[:, 0] - Hm[:, li]).abs().max()) < 4e-6
    # reverse modes: linear in the adjoint
    w1, w2 = torch.randn(n, L, 4, 4, generator=gen, **TA), torch.randn(n, L, 4, 4, generator=gen, **TA)
    g1, g2, g12 = ops.fk_backward(h, q, w1), ops.fk_backward(h, q, w2), ops.fk_backward(h, q, w1 + 2.0 * w2)
    scale = float(g12.abs().max())
    assert float((g12 - (g1 + 2.0 * g2)).abs().max()) < 2e-5 * scale
    wp = torch.zeros_like(w1); wp[..., :3, 3] = w1[..., :3, 3]
    assert float((ops.fk_backward(h, q, wp) - ops.fk_positions_backward(h, q, w1[..., :3, 3].contiguous())).abs().max()) < 2e-5 * scale
    # the geometric Jacobian's position / quaternion agree with the matrices (the stateful walk differs from the stateless one
    # only where a joint has limits that the sample violates or a negative axis: none for these robots' sampled range)
    jp, jq, lin, ang = ops.fk_jacobian(h, q, None, li)
    assert torch.isfinite(lin).all() and torch.isfinite(ang).all() and lin.shape == (n, 3, D)
    # generated == table-driven (another instruction order), whole batch
    h.enable_specialized(False)
    Ht, gt = ops.fk_forward(h, q), ops.fk_backward(h, q, w1)
    jp_t, jq_t, lin_t, ang_t = ops.fk_jacobian(h, q, None, li)
    h.enable_specialized(True)
    assert float((Ht - Hm).abs().max()) < 4e-6 and float((gt - g1).abs().max()) < 2e-5 * scale
    assert float((lin_t - lin).abs().max()) < 1e-5 and float((ang_t - ang).abs().max()) < 1e-5 and float((jp_t - jp).abs().max()) < 4e-6
    # fp64 oracle on a random subset
    idx = torch.randint(0, n, (2048,), generator=torch.Generator().manual_seed(3))
    qs = q[idx.to(DEV)].cpu().numpy().astype(np.float64)
    H64 = o.fk(qs, "f64")
    assert np.abs(Hm[idx.to(DEV)].cpu().numpy() - H64).max() / max(1.0, float(np.abs(H64).max())) < 2e-6
    g64 = o.fk_backward(qs, w1[idx.to(DEV)].cpu().numpy().astype(np.float64), "f64")
    assert rel_err(g1[idx.to(DEV)].cpu().numpy(), g64) < 2e-5


def test_ik_loop_full_size(ops):
    """trk_ik_steps on 4096 x 64 configurations: 12 iterations in one call == 12 single calls (bit for bit), generated ==
    table-driven after the same iterations on all but the noise-gradient elements, and the loss of (almost) every sample decreases."""
    robot = tra.RobotPanda(tensor_args=TA)
    kin = robot.diff_panda._kin
    h = ops.ModelHandle(kin)
    n, D = 4096 * 64, 7
    lo, hi = robot.q_min.to(DEV).float().contiguous(), robot.q_max.to(DEV).float().contiguous()
    Ht = torch.eye(4, **TA); Ht[:3, 3] = torch.tensor([0.4, 0.2, 0.5], **TA)
    q0 = robot.random_q(n, generator=torch.Generator(device=DEV).manual_seed(5)).contiguous()
    link = kin.n_links - 1

    def run(single, spec):
        h.enable_specialized(spec)
        q, m, v = q0.clone(), torch.zeros_like(q0), torch.zeros_like(q0)
        loss0 = torch.empty(n, **TA)
        if single:
            for it in range(12):
                ops.ik_step(h, link, Ht, lo, hi, q, m, v, it + 1, lr=1e-2, loss=loss0 if it == 0 else None)
        else:
            ops.ik_steps(h, link, Ht, lo, hi, q, m, v, 1, 12, lr=1e-2, loss=loss0)
        loss1 = torch.empty(n, **TA)
        ops.ik_step(h, link, Ht, lo, hi, q, m, v, 13, lr=0.0, loss=loss1)       # lr = 0: evaluate only
        h.enable_specialized(True)
        return q, loss0, loss1
    qa, la0, la1 = run(False, True)
    qb, lb0, lb1 = run(True, True)
    assert torch.equal(qa, qb) and torch.equal(la0, lb0)
    qc, lc0, lc1 = run(False, False)
    # Adam normalises the step by sqrt(v): a DOF whose gradient is rounding noise moves by +-lr per iteration whatever the sign
    # of the noise, so two instruction orders cannot agree on THOSE elements; everywhere else they do
    assert float((la0 - lc0).abs().max()) < 1e-4 * float(la0.abs().max())
    assert float(((qa - qc).abs() < 1e-4).float().mean()) > 0.999
    assert abs(float(la1.mean()) - float(lc1.mean())) < 1e-3 * float(la1.mean())
    assert float((la1 < la0).float().mean()) > 0.99


@pytest.mark.parametrize("scene", ["grid", "shelf", "maze"])
def test_panda_full_size_other_scenes(ops, oracle_lib, scene):
    """BASELINE configs[1] at 4096 x 64 on the scenes SURVEY 8(d) names next to the analytic spheres: the 200^3 voxel SDF of the
    same spheres, EnvTableShelf and EnvMazeBoxes3D (boxes in posed objects).  Size-independent properties + an fp64-oracle subset."""
    robot = tra.RobotPanda(tensor_args=TA)
    env = {"grid": lambda: tra.EnvSpheres3D(precompute_sdf_obj_fixed=True, sdf_cell_size=0.01, tensor_args=TA),
           "shelf": lambda: tra.EnvTableShelf(tensor_args=TA), "maze": lambda: tra.EnvMazeBoxes3D(tensor_args=TA)}[scene]()
    task = tra.PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(Ht)
    B, H = 4096, 64
    n = B * H
    q = robot.random_q(n, generator=torch.Generator(device=DEV).manual_seed(77)).reshape(B, H, 7).contiguous()
    model, cm = task._fused_handles(DEV)
    assert model.specialized
    if scene == "grid":
        assert tuple(env.grid_map_sdf_obj_fixed.sdf_tensor.shape) == (200, 200, 200)
    w = (1.0, 1.0, 1.0, 1.0)
    sums = torch.zeros(ops.n_blocks(n), **TA)
    pos, cost, gq = ops.rollout_cost_grad(model, cm, w, q, cost_sum=sums)
    assert torch.isfinite(pos).all() and torch.isfinite(cost).all() and torch.isfinite(gq).all()
    # determinism; a checksum of checksums (per-wavefront sums == per-trajectory costs == the total)
    pos2, cost2, gq2 = ops.rollout_cost_grad(model, cm, w, q)
    assert torch.equal(pos, pos2) and torch.equal(cost, cost2) and torch.equal(gq, gq2)
    assert torch.allclose(sums, cost.sum(1), rtol=2e-5, atol=1e-3)
    assert abs(ops.reduce_sum(sums).item() - cost.double().sum().item()) <= 1e-5 * abs(cost.double().sum().item())
    # linearity in the weights: cost(w) = sum_k w_k cost(e_k), same for the gradient
    acc_c, acc_g = torch.zeros_like(cost, dtype=torch.float64), torch.zeros_like(gq, dtype=torch.float64)
    for k, wk in enumerate((0.5, 2.0, 0.25, 1.5)):
        e = [0.0, 0.0, 0.0, 0.0]; e[k] = 1.0
        _, c_k, g_k = ops.rollout_cost_grad(model, cm, e, q, want_pos=False)
        acc_c += wk * c_k.double(); acc_g += wk * g_k.double()
    _, c_w, g_w = ops.rollout_cost_grad(model, cm, (0.5, 2.0, 0.25, 1.5), q, want_pos=False)
    assert rel_err(c_w.cpu().numpy(), acc_c.cpu().numpy()) < 1e-5 and rel_err(g_w.cpu().numpy(), acc_g.cpu().numpy()) < 1e-5
    # sharding invariance: any contiguous block of trajectories evaluated alone gives the same bits
    for lo, hi in ((0, 512), (1000, 1003), (4095, 4096)):
        p_s, c_s, g_s = ops.rollout_cost_grad(model, cm, w, q[lo:hi].contiguous())
        assert torch.equal(p_s, pos[lo:hi]) and torch.equal(c_s, cost[lo:hi]) and torch.equal(g_s, gq[lo:hi])
    # generated == table-driven kernel to fp32 rounding (the same function through two code paths); the gradient jumps at SDF kinks
    # (box faces / edges, arg-min ties, voxel boundaries), where the two roundings may take different branches: a handful of samples
    model.enable_specialized(False)
    idx = torch.arange(0, B, 16, device=DEV)
    _, c_t, g_t = ops.rollout_cost_grad(model, cm, w, q[idx].contiguous(), want_pos=False)
    model.enable_specialized(True)
    # (a link within fp32 rounding of a voxel boundary reads the neighbouring cell in one of the two kernels: the VALUE jumps too)
    bad_c = ((c_t - cost[idx]).abs() > 2e-5 * cost.abs().max()).sum().item()
    assert bad_c <= (idx.numel() * H // 2000 if scene == "grid" else 0), bad_c
    bad = ((g_t - gq[idx]).abs().amax(-1) > 2e-4 * gq.abs().max()).sum().item()
    assert bad <= idx.numel() * H // 2000, bad
    # boolean fields: fused == positions + field kernel; monotone in the margin
    fields = FIELD_OBJECTS | FIELD_WS | FIELD_SELF
    c_def = ops.rollout_collision(model, cm, fields, q)
    two_step = ops.collision_fields(cm, fields, pos.reshape(-1, 11, 3)).reshape(B, H).bool()
    assert (c_def != two_step).sum().item() <= 4
    c0, c1 = ops.rollout_collision(model, cm, fields, q, margin=0.0), ops.rollout_collision(model, cm, fields, q, margin=0.05)
    assert not (c0 & ~c1).any() and 0 < int(c0.sum()) < n
    # fp64 oracle on a random subset (the grid's host copy for the oracle)
    spec = task.build_cost_spec()
    if spec.grid is not None:
        spec.grid = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in spec.grid.items()}
    o = oracle_lib.Oracle(robot.diff_panda._kin, spec)
    sub = np.random.default_rng(5).choice(n, 1024, replace=False)
    p64, c64, g64 = o.rollout(q.reshape(-1, 7)[sub].cpu().numpy().astype(np.float64), w, "f64")
    assert np.abs(pos.reshape(-1, 11, 3)[sub].cpu().numpy() - p64).max() < 2e-6
    # a link within fp32 rounding of a voxel boundary / box face reads the neighbouring cell or branch: allow a few such samples
    dc = np.abs(cost.reshape(-1)[sub].cpu().numpy() - c64)
    assert (dc > 1e-5 * np.abs(c64).max()).sum() <= 4
    dg = np.abs(gq.reshape(-1, 7)[sub].cpu().numpy() - g64).max(1)
    assert (dg > 1e-4 * np.abs(g64).max()).sum() <= 8


@pytest.mark.parametrize("kind", ["link_spheres", "grasped_box", "both"])
def test_attached_points_full_size(ops, oracle_lib, kind):
    """SURVEY 8(f) ranks 3 / 4 at the size tools/bench_points.py times: Panda with the 45-sphere link model, with a grasped box, and with
    both (56 / 26 / 71 attached points), 4096 x 64, all four terms.  The pre-bound launch (PointsRolloutPlan) == the op; sharding
    invariance; the positions-only launch of the same kernel writes the same positions; the per-wavefront cost sums are a checksum of the
    costs; an fp64-oracle subset."""
    kw = dict(link_spheres=dict(link_sphere_model="panda"), grasped_box=dict(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA)),
              both=dict(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA)))[kind]
    robot = tra.RobotPanda(tensor_args=TA, **kw)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(Ht)
    B, H, D = 4096, 64, 7
    gen = torch.Generator(device=DEV).manual_seed(77)
    q = robot.random_q(B * H, generator=gen).reshape(B, H, D).contiguous()
    ps = robot._point_set(DEV)
    P = ps.n_points
    assert ps.specialized and P == dict(link_spheres=56, grasped_box=26, both=71)[kind]
    model, cm = task._fused_handles(DEV)
    w = (1.0, 1.0, 1.0, 1.0)
    plan = ops.PointsRolloutPlan(ps, cm, w, q)
    assert plan.generated
    sums = torch.zeros(ops.n_blocks(B * H), **TA)
    plan.launch(sums.data_ptr())
    torch.cuda.synchronize()
    assert ops.last_dispatch() == "generated"
    pos, cost, gq = plan.link_pos.reshape(B, H, P, 3), plan.cost.reshape(B, H), plan.gq.reshape(B, H, D)
    assert torch.isfinite(cost).all() and torch.isfinite(gq).all() and torch.isfinite(pos).all()
    p1, c1, g1 = ops.rollout_points_cost_grad(ps, cm, w, q)                              # the op allocates and calls the same entry point
    assert torch.equal(p1.reshape(pos.shape), pos) and torch.equal(c1.reshape(cost.shape), cost) and torch.equal(g1.reshape(gq.shape), gq)
    p2, c2, g2 = ops.rollout_points_cost_grad(ps, cm, w, q[B // 2:].contiguous())        # sharding invariance
    assert torch.equal(p2.reshape(B // 2, H, P, 3), pos[B // 2:]) and torch.equal(c2.reshape(B // 2, H), cost[B // 2:])
    assert torch.equal(g2.reshape(B // 2, H, D), gq[B // 2:])
    assert torch.equal(ops.fk_points(ps, q.reshape(-1, D)).reshape(pos.shape), pos)      # the positions-only exit of the same kernel
    assert abs(float(sums.double().sum()) - float(cost.double().sum())) <= 1e-5 * float(cost.double().abs().sum())
    pl, po = robot.collision_point_set()
    o = oracle_lib.Oracle(robot.diff_panda._kin, task.build_cost_spec())
    idx = np.random.default_rng(8).choice(B * H, 384, replace=False)
    qs = q.reshape(-1, D)[idx].cpu().numpy().astype(np.float64)
    rp, rc, rg = o.rollout_points(pl, po, qs, w, "f64")
    assert np.abs(pos.reshape(-1, P, 3)[idx].cpu().numpy() - rp).max() < 3e-6
    assert rel_err(cost.reshape(-1)[idx].cpu().numpy(), rc) < 1e-5
    from helpers import grad_close_kinks
    assert grad_close_kinks(gq.reshape(-1, D)[idx].cpu().numpy(), rg, qs, lambda qq: o.rollout_points(pl, po, qq, w, "f64")[2])


def test_trajectory_validation_full_size(ops, oracle_lib):
    """SURVEY 8(f) rank 1 at the size tools/bench_task_api.py times: get_trajs_collision_and_free on 4096 trajectories x 64 states, 5 via
    points per segment (1.29 M configurations).  The one-launch form (way-point collisions + per-trajectory flags) against the three-launch
    form bit for bit; the partition against the bookkeeping of tasks.py:253-284 stated in numpy; the way-point booleans of a random
    subset of trajectories against the fp64 oracle on the interpolated configurations."""
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T, H, D, n_interp = 4096, 64, 7, 5
    gen = torch.Generator(device=DEV).manual_seed(21)
    # smooth trajectories (a straight line between two random configurations + a little noise): about half are collision free
    a, b = robot.random_q(T, generator=gen), robot.random_q(T, generator=gen)
    s = torch.linspace(0.0, 1.0, H, **TA)[None, :, None]
    trajs = (a[:, None] * (1 - s) + (0.35 * a + 0.65 * b)[:, None] * s).contiguous()
    trajs[::97, 13, 2] += 9.0                                             # some leave the joint limits
    trajs = torch.cat([trajs, torch.zeros(T, H, D, **TA)], -1).contiguous()      # states with velocities
    tc, ci, tf, fi, wp = task.get_trajs_collision_and_free(trajs, return_indices=True, num_interpolation=n_interp)
    W = (H - 1) * n_interp
    assert wp.shape == (T, W)
    # the three-launch form of round 5 (way-point collisions, then flags from the bytes, then partition + gathers)
    qmin, qmax = robot.q_min.to(DEV, torch.float32).contiguous(), robot.q_max.to(DEV, torch.float32).contiguous()
    wp3 = task._waypoint_collisions(trajs, n_interp)
    assert torch.equal(wp3.reshape(T, W), wp.reshape(T, W))
    part3 = ops.traj_validate(wp3, trajs, D, qmin, qmax)
    nf, nc, no = part3.counts()
    assert nf > T // 10 and nc > T // 10 and no > 0                         # a mixed batch, or the test says little
    assert fi.numel() == nf and torch.equal(fi.reshape(-1), part3.idx[:nf].reshape(-1))
    assert torch.equal(tf, part3.gathered[:nf]) and torch.equal(tf, trajs[fi.reshape(-1)])
    assert ci.numel() == nc + no and torch.equal(tc, trajs[ci.reshape(-1)])
    coll_any = wp.reshape(T, W).bool().any(1).cpu().numpy()
    x = trajs[..., :D]
    outside = ((x < qmin) | (x > qmax)).any(-1).any(-1).cpu().numpy()
    np.testing.assert_array_equal(fi.reshape(-1).cpu().numpy(), np.flatnonzero(~coll_any & ~outside))
    np.testing.assert_array_equal(ci.reshape(-1).cpu().numpy(), np.concatenate([np.flatnonzero(coll_any), np.flatnonzero(~coll_any & outside)]))
    # oracle subset: the interpolated configurations of 24 trajectories, the boolean fields at margin 0 (tasks.py:296-306)
    sub = np.random.default_rng(5).choice(T, 24, replace=False)
    dense = ops.interpolate_traj_via_points(trajs[torch.as_tensor(sub, device=DEV)][..., :D].contiguous(), n_interp)
    assert dense.shape[1] == W
    o = oracle_lib.Oracle(robot.diff_panda._kin, task.build_cost_spec())
    qd = dense.reshape(-1, D).cpu().numpy().astype(np.float64)
    p64 = o.rollout(qd, (0, 0, 0, 0), "f64")[0]
    ref = o.collision_fields(FIELD_SELF | FIELD_OBJECTS | FIELD_WS, p64, 0.0, "f64").reshape(24, W)
    got = wp.reshape(T, W)[torch.as_tensor(sub, device=DEV)].cpu().numpy()
    assert (got != ref).mean() < 2e-3                                       # fp32 against fp64 at the threshold
