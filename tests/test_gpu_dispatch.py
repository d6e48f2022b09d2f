"""Which kernel family serves a rollout call (round 6): a weight on a term the cost model does not have must never send the call to
the table-driven kernels (the silent 28 x cliff of round 5: UR10 + Allegro 26.8 -> 743 us with w_self = 1 on a cost model without
self pairs), `trk_last_dispatch` says which family ran, and the strict mode turns a REAL mismatch into an error.
Reference call path that is being dispatched: PlanningTask._compute_collision_or_cost, tasks.py:139-232."""
import numpy as np
import pytest
import torch

from helpers import gold, model, panda_cost_spec, tree_cost_spec
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from torch_robotics_amd import ops as o
    return o


def _strip(spec, self_pairs=False, ws=False, ee=False):
    """the same cost model without some of its terms"""
    import copy
    s = copy.deepcopy(spec)
    if self_pairs:
        s.self_link_idx = np.zeros((0,), np.int32)
        s.self_pairs = np.zeros((0, 2), np.int32)
        s.self_margin = np.zeros((0,), np.float32)
    if ws:
        s.ws_min = s.ws_max = None
    if ee:
        s.ee_link, s.ee2_link = -1, -1
    s.validate()
    return s


@pytest.mark.parametrize("name", ["ur10_allegro", "panda"])
def test_vacuous_weights_stay_on_the_generated_kernel(ops, name):
    if name == "panda":
        g, robot = gold("cost_spheres3d"), gold("panda_robot")
        T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5)
        m, spec = model("panda_arm_no_gripper"), panda_cost_spec(g, robot, ee_target=T)
    else:
        m, spec, _ = tree_cost_spec(name)
    h = ops.ModelHandle(m)
    assert h.specialized
    rng = np.random.default_rng(5)
    q = torch.as_tensor(rng.uniform(-1.5, 1.5, size=(64, 64, m.n_dofs)).astype(np.float32), device=DEV)
    full = ops.CostHandle(spec, DEV)
    assert ops.rollout_is_specialized(h, full, (1, 1, 1, 1))
    for strip in (dict(self_pairs=True), dict(ws=True), dict(ee=True), dict(self_pairs=True, ws=True, ee=True)):
        cm = ops.CostHandle(_strip(spec, **strip), DEV)
        w_all = (1.0, 1.0, 1.0, 1.0)
        w_eff = (0.0 if strip.get("self_pairs") else 1.0, 1.0, 0.0 if strip.get("ws") else 1.0, 0.0 if strip.get("ee") else 1.0)
        assert ops.rollout_is_specialized(h, cm, w_all), strip
        pos, c, g_ = ops.rollout_cost_grad(h, cm, w_all, q)
        assert ops.last_dispatch() == "generated", strip
        pos2, c2, g2 = ops.rollout_cost_grad(h, cm, w_eff, q)          # the explicit zero weight: the same launch
        assert ops.last_dispatch() == "generated"
        assert torch.equal(c, c2) and torch.equal(g_, g2) and torch.equal(pos, pos2), strip
        # ... and the same function as the table-driven kernel computes for these weights (the vacuous terms contribute nothing there)
        h.enable_specialized(False)
        _, ct, gt = ops.rollout_cost_grad(h, cm, w_all, q)
        assert ops.last_dispatch() == "table-driven"
        h.enable_specialized(True)
        assert float((c - ct).abs().max()) <= 2e-5 * max(1.0, float(ct.abs().max())), strip
        assert float((g_ - gt).abs().max()) <= 2e-4 * max(1.0, float(gt.abs().max())), strip
        # the boolean exit: a field with nothing to test is dropped from the mask, not sent to the two-step table-driven form
        hit = ops.rollout_collision(h, cm, FIELD_SELF | FIELD_OBJECTS | FIELD_WS, q)
        assert ops.last_dispatch() == "generated", strip
        keep = (0 if strip.get("self_pairs") else FIELD_SELF) | FIELD_OBJECTS | (0 if strip.get("ws") else FIELD_WS)
        assert torch.equal(hit, ops.rollout_collision(h, cm, keep, q)), strip
    # a cost model with NOTHING to test: nobody collides, nothing is launched
    s0 = _strip(spec, self_pairs=True, ws=True)
    s0.objects = []
    s0.validate()
    cm0 = ops.CostHandle(s0, DEV)
    assert not bool(ops.rollout_collision(h, cm0, FIELD_SELF | FIELD_OBJECTS | FIELD_WS, q).any())
    assert ops.last_dispatch() == "none"


def test_strict_mode_refuses_a_real_mismatch(ops):
    """another collision-link set than the unit bakes: table-driven by default, TRK_ERR_UNSUPPORTED (NotImplementedError) in strict mode;
    the pre-bound plans are strict by default"""
    g, robot = gold("cost_spheres3d"), gold("panda_robot")
    m = model("panda_arm_no_gripper")
    spec = panda_cost_spec(g, robot)
    spec.obj_link_idx = np.asarray(spec.obj_link_idx)[:-1].copy()            # one collision link less than RobotPanda's template
    spec.obj_link_margin = np.asarray(spec.obj_link_margin)[:-1].copy()
    spec.validate()
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    q = torch.zeros(2, 64, 7, device=DEV)
    assert h.specialized and not ops.rollout_is_specialized(h, cm, (0, 1, 0, 0))
    _, c, _ = ops.rollout_cost_grad(h, cm, (0, 1, 0, 0), q)
    assert ops.last_dispatch() == "table-driven"
    with ops.strict_specialized():
        with pytest.raises(NotImplementedError, match="strict mode"):
            ops.rollout_cost_grad(h, cm, (0, 1, 0, 0), q)
        with pytest.raises(NotImplementedError, match="strict mode"):
            ops.rollout_collision(h, cm, FIELD_OBJECTS, q)
        # terms the unit does serve are not affected: the self pairs are RobotPanda's
        ops.rollout_cost_grad(h, cm, (1, 0, 0, 0), q)
        assert ops.last_dispatch() == "generated"
        # a model WITHOUT generated units is served as before (strict mode is about silent fall-backs, not about coverage)
        h.enable_specialized(False)
        _, c2, _ = ops.rollout_cost_grad(h, cm, (0, 1, 0, 0), q)
        h.enable_specialized(True)
        assert torch.equal(c, c2)
    assert ops.set_strict_specialized(False) is False                          # the context manager restored the default
    with pytest.raises(NotImplementedError, match="strict=True"):
        ops.RolloutPlan(h, cm, (0, 1, 0, 0), q)
    plan = ops.RolloutPlan(h, cm, (0, 1, 0, 0), q, strict=False)
    assert not plan.generated
    plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(plan.cost, c)


def test_points_rollout_plan_matches_the_eager_call(ops):
    """F3 / F4 (robot_panda.py:154-168, the link-sphere table): the attached-point models are pre-bound like the link models"""
    import torch_robotics_amd as tra
    TA = dict(device=DEV, dtype=torch.float32)
    robot = tra.RobotPanda(tensor_args=TA, grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    q = robot.random_q(3 * 64).reshape(3, 64, 7).contiguous()
    plan = task.rollout_plan(q)
    assert isinstance(plan, ops.PointsRolloutPlan) and plan.generated
    for _ in range(2):
        plan.launch()
        pos, cost, gq = task.rollout_cost_grad(q)
        torch.cuda.synchronize()
        assert ops.last_dispatch() == "generated"
        assert torch.equal(plan.cost, cost) and torch.equal(plan.gq, gq) and torch.equal(plan.link_pos, pos)
        q.copy_(robot.random_q(3 * 64).reshape(3, 64, 7))
    # the sharded exchange takes it like any plan
    sums = torch.zeros(ops.n_blocks(3 * 64), **TA)
    plan.launch(sums.data_ptr())
    pk = ops.PackedSums(plan, sums)
    out = torch.empty(pk.size, **TA)
    pk.pack(out)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(plan.cost.double().sum())) <= 2e-6 * abs(float(plan.cost.double().sum()))


@pytest.mark.parametrize("T,H,n_interp", [(4096, 64, 5), (37, 64, 5), (1, 2, 1), (203, 3, 2), (50, 17, 7), (1027, 9, 3)])
def test_trajectory_flags_folded_into_the_via_launch(ops, T, H, n_interp):
    """Round 6 (F1; tasks.py:244-299): `rollout_collision_via(..., limits=...)` produces the per-trajectory flags in the launch that
    evaluates the via points -- the SAME flags, partition and gathers as the three-launch form, for trajectories that are free, colliding,
    outside the joint limits (incl. NaN way points) or both; a wavefront may span one trajectory or dozens."""
    g, robot = gold("cost_spheres3d"), gold("panda_robot")
    m = model("panda_arm_no_gripper")
    spec = panda_cost_spec(g, robot)
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    rng = np.random.default_rng(T + H)
    lo, hi = np.asarray(m.lower[m.dof_idx >= 0], np.float32), np.asarray(m.upper[m.dof_idx >= 0], np.float32)
    lo, hi = lo[np.argsort(m.dof_idx[m.dof_idx >= 0])], hi[np.argsort(m.dof_idx[m.dof_idx >= 0])]
    S = 14                                                       # positions + velocities, like the planners' states
    x = np.zeros((T, H, S), np.float32)
    base = rng.uniform(lo * 0.5, hi * 0.5, size=(T, 1, 7))
    x[..., :7] = np.clip(base + np.cumsum(rng.standard_normal((T, H, 7)) * 0.01, axis=1), lo + 1e-3, hi - 1e-3)
    x[..., 7:] = rng.standard_normal((T, H, 7)) * 100.0          # velocities are not positions: never tested against the limits
    viol = rng.random(T) < 0.2                                   # a way point outside the limits, somewhere (also the last one)
    for t in np.nonzero(viol)[0]:
        hh, d = (H - 1 if t % 3 == 0 else rng.integers(0, H)), rng.integers(0, 7)
        x[t, hh, d] = hi[d] + 0.01 if t % 2 else (np.nan if t % 5 == 0 else lo[d] - 0.01)
    xt = torch.as_tensor(x, device=DEV)
    qmin, qmax = torch.as_tensor(lo, device=DEV), torch.as_tensor(hi, device=DEV)
    fields = FIELD_SELF | FIELD_OBJECTS | FIELD_WS
    wp0 = ops.rollout_collision_via(h, cm, fields, xt, n_interp, margin=0.0)
    wp1, buf = ops.rollout_collision_via(h, cm, fields, xt, n_interp, margin=0.0, limits=(qmin, qmax))
    assert torch.equal(wp0, wp1)
    ref = ops.traj_validate(wp0, xt, 7, qmin, qmax)
    got = ops.traj_validate(None, xt, 7, qmin, qmax, flags=buf)
    assert torch.equal(ref.flags, got.flags)
    expect = (wp0.any(1).to(torch.uint8) | (torch.as_tensor(viol, device=DEV).to(torch.uint8) << 1))
    assert torch.equal(got.flags, expect)
    assert ref.counts() == got.counts()
    n = sum(got.counts())
    assert n == T and torch.equal(ref.idx, got.idx) and torch.equal(ref.gathered.nan_to_num(7.0), got.gathered.nan_to_num(7.0))
    # a cost model with nothing to test still reports the joint limits
    s0 = _strip(spec, self_pairs=True, ws=True)
    s0.objects = []
    s0.validate()
    cm0 = ops.CostHandle(s0, DEV)
    wp2, buf2 = ops.rollout_collision_via(h, cm0, fields, xt, n_interp, margin=0.0, limits=(qmin, qmax))
    assert not bool(wp2.any()) and torch.equal(ops.traj_validate(None, xt, 7, qmin, qmax, flags=buf2).flags,
                                               torch.as_tensor(viol, device=DEV).to(torch.uint8) << 1)


@pytest.mark.parametrize("name,base_pose", [("ur10_allegro", False), ("ur10_allegro", True), ("panda", False)])
def test_rollout_and_jacobian_in_one_launch(ops, name, base_pose):
    """Round 6, BASELINE config 4 ("FK + Jacobian + cost"): trk_rollout_jacobian_cost_grad == trk_rollout_cost_grad followed by
    trk_fk_jacobian (robot_tree.py:218-248) == the fp64 oracle.  ONE launch of the generated unit (the Jacobian's columns are read out of
    the poses the rollout holds) for UR10 + Allegro (ring-staged positions), also with a moved base, and for the Panda (whole-row staging);
    a link the unit does not track, or a model without generated kernels, runs the two launches behind the same call and says so."""
    from oracle import oracle as orc
    if name == "panda":
        g, robot = gold("cost_spheres3d"), gold("panda_robot")
        T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5)
        m, spec = model("panda_arm_no_gripper"), panda_cost_spec(g, robot, ee_target=T)
    else:
        m, spec, _ = tree_cost_spec(name)
    if base_pose:
        c, s_ = np.cos(0.3), np.sin(0.3)
        m.base_R = np.asarray([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
        m.base_t = np.asarray([0.1, -0.2, 0.05], np.float32)
    h, cm = ops.ModelHandle(m), ops.CostHandle(spec, DEV)
    if base_pose:
        h.set_base_pose(m.base_R, m.base_t)
    o = orc.Oracle(m, spec)
    link = int(spec.ee_link)
    w = (0.0, 1.0, 0.0, 1.0)
    rng = np.random.default_rng(17)
    for B, H in ((5, 64), (3, 37), (1, 1)):
        q = torch.as_tensor(rng.uniform(-2.0, 2.0, size=(B, H, m.n_dofs)).astype(np.float32), device=DEV)
        plan = ops.RolloutJacobianPlan(h, cm, w, q, link)
        sums = torch.zeros(ops.n_blocks(B * H), device=DEV)
        plan.launch(sums.data_ptr())
        torch.cuda.synchronize()
        assert ops.last_dispatch() == "generated"
        pos, cost, gq = ops.rollout_cost_grad(h, cm, w, q)
        jp, jq, lin, ang = ops.fk_jacobian(h, q.reshape(-1, m.n_dofs), None, link)[:4]
        scale = max(1.0, float(pos.abs().max()))
        assert float((plan.link_pos - pos).abs().max()) <= 2e-6 * scale
        assert float((plan.cost - cost).abs().max()) <= 1e-5 * max(1.0, float(cost.abs().max()))
        assert float((plan.gq - gq).abs().max()) <= 1e-4 * max(1e-3, float(gq.abs().max()))
        assert float((plan.pos.reshape(-1, 3) - jp).abs().max()) <= 2e-6 * scale
        assert float((plan.lin_jac.reshape(lin.shape) - lin).abs().max()) <= 4e-6 * scale and float((plan.ang_jac.reshape(ang.shape) - ang).abs().max()) <= 2e-6
        # a quaternion and its negative are the same rotation
        dq = torch.minimum((plan.quat.reshape(-1, 4) - jq).abs().amax(1), (plan.quat.reshape(-1, 4) + jq).abs().amax(1))
        assert float(dq.max()) <= 5e-6
        rp, rq, rl, ra, _, _ = o.jacobian(q.reshape(-1, m.n_dofs).cpu().numpy().astype(np.float64), None, link, "f64")
        assert np.abs(plan.pos.reshape(-1, 3).cpu().numpy() - rp).max() <= 3e-6 * scale
        assert np.abs(plan.lin_jac.reshape(rl.shape).cpu().numpy() - rl).max() <= 6e-6 * scale
        assert np.abs(plan.ang_jac.reshape(ra.shape).cpu().numpy() - ra).max() <= 3e-6
        # another link than the tracked one: the same call, two launches, the same function
        other = link - 1
        plan2 = ops.RolloutJacobianPlan(h, cm, w, q, other)
        plan2.launch()
        torch.cuda.synchronize()
        assert ops.last_dispatch() == "generated + prior launches"
        rp2, _, rl2, ra2, _, _ = o.jacobian(q.reshape(-1, m.n_dofs).cpu().numpy().astype(np.float64), None, other, "f64")
        assert np.abs(plan2.pos.reshape(-1, 3).cpu().numpy() - rp2).max() <= 3e-6 * scale
        assert np.abs(plan2.lin_jac.reshape(rl2.shape).cpu().numpy() - rl2).max() <= 6e-6 * scale
        assert torch.equal(plan2.cost, plan.cost) or float((plan2.cost - plan.cost).abs().max()) <= 1e-5 * max(1.0, float(cost.abs().max()))
        # the structural zeros are written, not left over: poison the buffers and evaluate again
        plan.lin_jac.fill_(7.0); plan.ang_jac.fill_(7.0)
        plan.launch()
        torch.cuda.synchronize()
        assert np.abs(plan.lin_jac.reshape(rl.shape).cpu().numpy() - rl).max() <= 6e-6 * scale
        assert np.abs(plan.ang_jac.reshape(ra.shape).cpu().numpy() - ra).max() <= 3e-6
