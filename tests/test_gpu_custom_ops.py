"""The dispatcher ops `torch.ops.trk.*` (torch_robotics_amd/custom_ops.py): schema / fake-tensor / autograd-registration checks
with `torch.library.opcheck`, parity with the reference goldens when the Python API is routed through them, and
`torch.compile` of the reference's call sites (robot_tree.py:267-301, tasks.py:135-137)."""
import numpy as np
import pytest
import torch

import torch_robotics_amd as tra
from helpers import gold, grad_close, model, panda_cost_spec, rel_err
from torch_robotics_amd import custom_ops, ops  # noqa: F401  (registers torch.ops.trk.*)
from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TA = dict(device=DEV, dtype=torch.float32)
CHECKS = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


@pytest.fixture(scope="module")
def handles():
    g, robot, gs = gold("rollout_panda"), gold("panda_robot"), gold("cost_spheres3d")
    m = ops.ModelHandle(model("panda_arm_no_gripper"))
    cm = ops.CostHandle(panda_cost_spec(gs, robot, ee_target=g["target"]), DEV)
    return m, cm, g, robot


def test_ops_are_registered_with_the_dispatcher(handles):
    names = ["fk", "fk_backward", "fk_positions", "fk_positions_backward", "cost_fields", "cost_fields_backward", "ee_cost",
             "ee_cost_backward", "rollout_cost_grad", "scale_rows"]
    for n in names:
        op = getattr(torch.ops.trk, n)
        assert op.default._schema.name == f"trk::{n}"
    m, cm, _, _ = handles
    assert ops.handle_of(m.uid) is m and ops.handle_of(cm.uid) is cm
    with pytest.raises(ValueError, match="not a live"):
        torch.ops.trk.fk(torch.zeros(2, 7, device=DEV), 12345, None)
    with pytest.raises(ValueError, match="expected a ModelHandle"):      # a cost model's id where a model's is expected
        torch.ops.trk.fk(torch.zeros(2, 7, device=DEV), cm.uid, None)


def test_opcheck(handles):
    m, cm, g, robot = handles
    q = dev(g["q"][:1].reshape(-1, 7))                                          # (64, 7)
    qg = q.clone().requires_grad_(True)
    oc = torch.library.opcheck
    oc(torch.ops.trk.fk.default, (qg, m.uid, None), test_utils=CHECKS)
    oc(torch.ops.trk.fk.default, (qg, m.uid, [10, 3]), test_utils=CHECKS)
    oc(torch.ops.trk.fk_positions.default, (qg, m.uid, None), test_utils=CHECKS)
    oc(torch.ops.trk.fk_backward.default, (q, torch.randn(64, 11, 4, 4, device=DEV), m.uid, None), test_utils=CHECKS)
    oc(torch.ops.trk.fk_positions_backward.default, (q, torch.randn(64, 2, 3, device=DEV), m.uid, [10, 3]), test_utils=CHECKS)
    pos = dev(robot["fk_map_collision"].reshape(-1, 11, 3)).requires_grad_(True)
    fields = FIELD_OBJECTS | FIELD_SELF | FIELD_WS
    oc(torch.ops.trk.cost_fields.default, (pos, cm.uid, fields), test_utils=CHECKS)
    oc(torch.ops.trk.cost_fields_backward.default, (pos.detach(), torch.rand(64, device=DEV), cm.uid, fields), test_utils=CHECKS)
    H = ops.fk_forward(m, q, [10]).reshape(-1, 4, 4).requires_grad_(True)
    oc(torch.ops.trk.ee_cost.default, (H, None, cm.uid), test_utils=CHECKS)
    oc(torch.ops.trk.ee_cost.default, (H, dev(g["target"]), cm.uid), test_utils=CHECKS)
    oc(torch.ops.trk.ee_cost_backward.default, (H.detach(), None, torch.rand(64, device=DEV), cm.uid), test_utils=CHECKS)
    q3 = dev(g["q"][:2]).requires_grad_(True)                                   # (2, 64, 7)
    oc(torch.ops.trk.rollout_cost_grad.default, (q3, m.uid, cm.uid, [1.0, 1.0, 1.0, 1.0], True, 0), test_utils=CHECKS)
    oc(torch.ops.trk.rollout_cost_grad.default, (q3, m.uid, cm.uid, [0.0, 1.0, 0.0, 1.0], False, 0), test_utils=CHECKS)
    oc(torch.ops.trk.rollout_cost_grad.default, (q3.detach().half(), m.uid, cm.uid, [0.0, 1.0, 0.0, 1.0], True, 0),
       test_utils=("test_schema", "test_faketensor"))
    oc(torch.ops.trk.scale_rows.default, (torch.randn(2, 64, 7, device=DEV), torch.rand(2, 64, device=DEV)), test_utils=CHECKS)


def test_gradients_through_dispatcher_ops_match_goldens(handles, monkeypatch):
    """The Python API routed through torch.ops.trk.* (TRK_DISPATCHER_OPS=1) reproduces the reference's values and gradients."""
    monkeypatch.setattr(ops, "_ALWAYS_DISPATCH", True)
    m, cm, g, robot = handles
    gf = gold("fk_panda_arm_no_gripper")
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    q = dev(gf["q_in"]).requires_grad_(True)
    H = tree.compute_forward_kinematics_all_links(q)
    assert np.abs(H.detach().cpu().numpy() - gf["H_in"]).max() < 2e-6
    (H * dev(gf["w_in"])).sum().backward()
    assert grad_close(q.grad.cpu().numpy(), gf["gq_in"])
    # PlanningTask.compute_collision_cost(q).sum().backward() -- one trk::rollout_cost_grad node, backward = trk::scale_rows
    gc = gold("cost_spheres3d")
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=tra.RobotPanda(tensor_args=TA),
                            obstacle_cutoff_margin=float(gc["cutoff"]), tensor_args=TA)
    q = dev(gc["q"]).requires_grad_(True)
    cost = task.compute_collision_cost(q)
    assert "Rollout" in cost.grad_fn.name()          # ONE node: the native op's C++ node (or the Python-registered op's)
    assert rel_err(cost.detach().cpu().numpy(), gc["cost_total"]) < 1e-5
    (cost * 1.0).sum().backward()
    assert grad_close(q.grad.cpu().numpy(), gc["gq_total"])
    # a non-uniform upstream gradient goes through the scale kernel
    q2 = dev(gc["q"]).requires_grad_(True)
    wgt = torch.linspace(0.5, 2.0, 64, device=DEV).reshape(8, 8)
    (task.compute_collision_cost(q2) * wgt).sum().backward()
    assert torch.allclose(q2.grad, q.grad * wgt.unsqueeze(-1), rtol=1e-6, atol=1e-7)
    # unfused field + FK chain as the reference writes it
    q3 = dev(gc["q"]).requires_grad_(True)
    pos = task.robot.fk_map_collision(q3)
    task.df_collision_objects.compute_cost(q3, pos, field_type="sdf").sum().backward()
    assert grad_close(q3.grad.cpu().numpy(), gc["gq_objects"])


def test_torch_compile_of_the_reference_call_sites(handles):
    """`torch.compile` traces through the ops (fake implementations) and the compiled callables give the eager results."""
    gc = gold("cost_spheres3d")
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=tra.RobotPanda(tensor_args=TA),
                            obstacle_cutoff_margin=float(gc["cutoff"]), tensor_args=TA)
    q = dev(gc["q"])
    eager = task.compute_collision_cost(q)
    compiled_cost = torch.compile(task.compute_collision_cost)
    out = compiled_cost(q)
    assert torch.equal(out, eager)
    assert rel_err(out.cpu().numpy(), gc["cost_total"]) < 1e-5
    qg = q.clone().requires_grad_(True)
    compiled_cost(qg).sum().backward()
    assert grad_close(qg.grad.cpu().numpy(), gc["gq_total"])

    m, cm, g, _ = handles

    def objective(q, scale):                    # an op in the middle of ordinary torch code: one graph, no breaks
        cost, gq, pos = torch.ops.trk.rollout_cost_grad(q * scale, m.uid, cm.uid, [1.0, 1.0, 1.0, 1.0], True, 0)
        return cost.sum() + 1e-3 * pos.square().sum()
    q3 = dev(g["q"][:2])
    fn = torch.compile(objective, fullgraph=True)
    a, b = fn(q3, 1.0), objective(q3, 1.0)
    assert torch.allclose(a, b, rtol=1e-6)
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    gf = gold("fk_panda_arm_no_gripper")
    fkc = torch.compile(lambda x: tree.compute_forward_kinematics_all_links(x))
    assert np.abs(fkc(dev(gf["q_in"])).cpu().numpy() - gf["H_in"]).max() < 2e-6


def test_native_rollout_op(handles):
    """csrc/trk_torch_ops.cpp: trk::rollout as a C++ dispatcher op with a C++ autograd node (backward = trk_scale_rows): opcheck, the
    reference's idiom eager and under torch.compile(fullgraph=True) against the goldens, a gradient on a by-product output is an error."""
    from torch_robotics_amd import _lib
    native = _lib.torch_ops()
    assert native is not None, "libtrk_torch.so is part of the build"
    m, cm, g, _ = handles
    q3 = dev(g["q"][:2]).requires_grad_(True)                                   # (2, 64, 7)
    for args in ((q3, m.ptr, cm.ptr, 1.0, 1.0, 1.0, 1.0, True), (q3, m.ptr, cm.ptr, 0.0, 1.0, 0.0, 1.0, False)):
        torch.library.opcheck(native.rollout.default, args, test_utils=CHECKS)
    torch.library.opcheck(native.rollout.default, (q3.detach().half(), m.ptr, cm.ptr, 0.0, 1.0, 0.0, 1.0, True), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(native.scale_rows_native.default, (torch.randn(2, 64, 7, device=DEV), torch.rand(2, 64, device=DEV)),
                          test_utils=("test_schema", "test_faketensor"))
    # handles travel as integers: a stale / foreign integer is an ERROR from the op, not a segfault inside it (libtrk.so's registry)
    from torch_robotics_amd.costmodel import CostModelSpec
    tmp = ops.CostHandle(CostModelSpec(n_links_in=11), DEV)
    stale = tmp.ptr
    assert _lib.lib().trk_handle_kind(stale) == 2 and _lib.lib().trk_handle_kind(m.ptr) == 1
    del tmp
    import gc as _gc
    _gc.collect()
    assert _lib.lib().trk_handle_kind(stale) == 0 and _lib.lib().trk_handle_kind(12345) == 0
    for bad_m, bad_c in ((m.ptr, stale), (m.ptr, 12345), (cm.ptr, cm.ptr), (0, cm.ptr)):
        with pytest.raises(RuntimeError, match="not a live"):
            native.rollout(q3.detach(), bad_m, bad_c, 1.0, 1.0, 1.0, 1.0, False)
    # == the ctypes path, value and gradient
    pos_r, cost_r, gq_r = ops.rollout_cost_grad(m, cm, (1, 1, 1, 1), q3.detach())
    cost, gq, pos = native.rollout(q3, m.ptr, cm.ptr, 1.0, 1.0, 1.0, 1.0, True)
    assert torch.equal(cost, cost_r) and torch.equal(gq, gq_r) and torch.equal(pos, pos_r) and "Rollout" in cost.grad_fn.name()
    wgt = torch.linspace(0.5, 2.0, 128, device=DEV).reshape(2, 64)
    (cost * wgt).sum().backward()
    assert torch.allclose(q3.grad, gq_r * wgt.unsqueeze(-1), rtol=1e-6, atol=1e-7)
    q3.grad = None
    native.rollout(q3, m.ptr, cm.ptr, 1.0, 1.0, 1.0, 1.0, False)[0].sum().backward()      # an expanded scalar comes down: read as one value
    assert torch.equal(q3.grad, gq_r)
    # gq and link_pos are by-products: flagged non-differentiable (no made-up zero gradients) -- also on the Python-registered twin
    c2, g2, p2 = native.rollout(q3, m.ptr, cm.ptr, 1.0, 1.0, 1.0, 1.0, True)
    assert c2.requires_grad and not g2.requires_grad and not p2.requires_grad
    with pytest.raises(RuntimeError, match="does not require grad"):
        p2.sum().backward()
    c4, g4, p4 = torch.ops.trk.rollout_cost_grad(q3, m.uid, cm.uid, [1.0, 1.0, 1.0, 1.0], True, 0)
    assert c4.requires_grad and not g4.requires_grad and not p4.requires_grad
    # the reference's call site: routed through the native op by default, traceable as ONE graph
    gc = gold("cost_spheres3d")
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=tra.RobotPanda(tensor_args=TA),
                            obstacle_cutoff_margin=float(gc["cutoff"]), tensor_args=TA)
    q = dev(gc["q"]).requires_grad_(True)
    cost = task.compute_collision_cost(q)
    assert "Rollout" in cost.grad_fn.name() and rel_err(cost.detach().cpu().numpy(), gc["cost_total"]) < 1e-5
    cost.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), gc["gq_total"])
    task.compute_collision_cost(q.detach())                     # handles exist before tracing
    fn = torch.compile(task.compute_collision_cost, fullgraph=True)
    q2 = dev(gc["q"]).requires_grad_(True)
    out = fn(q2)
    assert torch.equal(out.detach(), cost.detach())
    out.sum().backward()
    assert grad_close(q2.grad.cpu().numpy(), gc["gq_total"])


def test_graphed_cost_backward_replays_the_eager_result():
    """PlanningTask.capture_cost_backward: `compute_collision_cost(x).sum().backward()` (tasks.py:135-137 under autograd) captured once as a
    hipGraph; a replay must give what the eager idiom gives -- at the captured q, and after q was updated in place."""
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    gen = torch.Generator(device=DEV).manual_seed(7)
    q = robot.random_q(256 * 16, generator=gen).reshape(256, 16, 7).contiguous().requires_grad_(True)

    def eager(x):
        x = x.detach().clone().requires_grad_(True)
        c = task.compute_collision_cost(x)
        c.sum().backward()
        return c.detach().clone(), x.grad.detach().clone()

    g = task.capture_cost_backward(q)
    assert g.grad is q.grad and g.cost.shape == (256, 16)
    for trip in range(3):
        c_ref, g_ref = eager(q)
        c = g.replay()
        torch.cuda.synchronize()
        assert torch.equal(c, c_ref) and torch.equal(q.grad, g_ref), f"replay {trip} differs from the eager evaluation"
        assert torch.equal(g.total, c_ref.sum())
        with torch.no_grad():                               # what an optimiser step does: q moves in place
            q.add_(0.05 * torch.randn(q.shape, device=DEV, generator=gen))
    assert g.plan is not None and g.graph is None           # reduce=torch.sum is recognised: a replay is ONE launch of the fused rollout
    assert q.grad.data_ptr() == g.plan.gq.data_ptr()         # ... writing straight into q.grad's storage
    q.grad = None                                            # an optimiser's zero_grad(set_to_none=True): the next replay rebinds it
    c = g.replay()
    c_ref, g_ref = eager(q)
    assert q.grad is g.grad and torch.equal(q.grad, g_ref) and torch.equal(c, c_ref)
    # any other reduction is captured as a graph (torch's whole-network recipe) and replays the eager result too
    q2 = q.detach().clone().requires_grad_(True)
    g2 = task.capture_cost_backward(q2, reduce=torch.mean)
    assert g2.plan is None and g2.graph is not None
    for trip in range(2):
        x = q2.detach().clone().requires_grad_(True)
        c_e = task.compute_collision_cost(x)
        c_e.mean().backward()
        c = g2.replay()
        torch.cuda.synchronize()
        assert torch.equal(c, c_e.detach()) and torch.equal(q2.grad, x.grad) and torch.equal(g2.total, c_e.mean())
        with torch.no_grad():
            q2.add_(0.05 * torch.randn(q2.shape, device=DEV, generator=gen))
    # (N, D) input: cost comes back as (N, 1) like the eager call
    q3 = robot.random_q(300, generator=gen).contiguous().requires_grad_(True)
    g3 = task.capture_cost_backward(q3)
    x = q3.detach().clone().requires_grad_(True)
    c_e = task.compute_collision_cost(x)
    c_e.sum().backward()
    assert g3.plan is not None and torch.equal(g3.replay(), c_e.detach()) and torch.equal(q3.grad, x.grad)
    with pytest.raises(ValueError, match="leaf"):
        task.capture_cost_backward(q.detach() * 1.0)
