"""GPU parity through the drop-in Python interface: these read like calls into the reference
(`DifferentiableTree`, `RobotPanda`, `PlanningTask`, the distance fields) and are checked against
vectors the reference produced for the same calls."""
import numpy as np
import pytest
import torch

import torch_robotics_amd as tra
from helpers import gold, grad_close, rel_err

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TA = dict(device=DEV, dtype=torch.float32)
TOL_H, TOL_C, TOL_G = 2e-6, 1e-5, 1e-4


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def test_forward_kinematics_example_plumbing():
    """examples/forward_kinematics.py with batch 32, seed 1 (BASELINE config 1) + SURVEY section 4 KATs."""
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    torch.manual_seed(1)
    q = torch.rand(32, 7).to(DEV).requires_grad_(True)
    H = tree.compute_forward_kinematics_all_links(q)
    assert H.shape == (32, 11, 4, 4)
    np.testing.assert_allclose(H[0, -1, :3].detach().cpu().numpy(),
                               [[-0.9966, 0.0175, 0.0803, 0.2024], [0.0528, 0.8851, 0.4624, 0.3237],
                                [-0.0629, 0.4651, -0.8830, 0.8609]], atol=6e-5)
    H[..., :3, 3].sum().backward()
    np.testing.assert_allclose(q.grad[0].cpu().numpy(), [-0.3785, 4.2100, -0.3222, 0.0, -0.2273, 0.8131, 0.0], atol=6e-5)
    # 1-D q gets a batch dimension; dict output holds Frame-like objects
    d = tree.compute_forward_kinematics_all_links(q.detach()[0], return_dict=True, link_list=["ee_link", "panda_link3"])
    assert set(d) == {"ee_link", "panda_link3"}
    # a link subset runs the table-driven kernel, all links the generated one: equal up to fp32 rounding of different FMA orders
    np.testing.assert_allclose(d["ee_link"].get_transform_matrix().cpu().numpy(), H[:1, -1].detach().cpu().numpy(), rtol=0, atol=2e-6)
    Hs = tree.compute_forward_kinematics_all_links(q.detach(), link_list=["ee_link", "panda_link3", "ee_link"])
    np.testing.assert_array_equal(Hs[:, 2].cpu().numpy(), Hs[:, 0].cpu().numpy())           # a repeated link is the same matrix
    np.testing.assert_allclose(Hs[:, 2].cpu().numpy(), H[:, 10].detach().cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(Hs[:, 1].cpu().numpy(), H[:, 3].detach().cpu().numpy(), rtol=0, atol=2e-6)
    # state_less single-link call returns the SE(3) matrix (robot_tree.py:207-208)
    np.testing.assert_allclose(tree.compute_forward_kinematics(q.detach(), None, "ee_link", state_less=True).cpu().numpy(),
                               H[:, 10:11].detach().cpu().numpy(), rtol=0, atol=2e-6)


def test_geometric_jacobian_api():
    g = gold("jac_panda_arm_no_gripper")
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    pos, quat, lin, ang = tree.compute_forward_kinematics_and_geometric_jacobian(dev(g["q"]), dev(g["qd"]), "ee_link")
    assert np.abs(pos.cpu().numpy() - g["pos_0"]).max() < 2e-6
    assert np.abs(quat.cpu().numpy() - g["quat_0"]).max() < 2e-6
    assert np.abs(lin.cpu().numpy() - g["lin_0"]).max() < 3e-6
    assert np.abs(ang.cpu().numpy() - g["ang_0"]).max() < 2e-6
    p2, q2 = tree.compute_forward_kinematics(dev(g["q"]), dev(g["qd"]), "ee_link")
    np.testing.assert_array_equal(p2.cpu().numpy(), pos.cpu().numpy())
    # a later stateless call is not contaminated by the stateful one (the reference needs reset())
    H = tree.compute_forward_kinematics_all_links(dev(gold("fk_panda_arm_no_gripper")["q_in"]))
    assert np.abs(H.cpu().numpy() - gold("fk_panda_arm_no_gripper")["H_in"]).max() < TOL_H
    quat_all = tra.kinematics.link_quat_from_link_tensor(H)
    assert quat_all.shape == (32, 11, 4)
    ga = gold("ajac_panda_arm_no_gripper")
    Ja = tree.compute_analytical_jacobian_all_links(dev(ga["q"]))
    assert Ja.shape == (12, 11, 7, 7) and np.abs(Ja.cpu().numpy() - ga["J"]).max() < 3e-6


def test_update_base_pose():
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    q = dev(gold("fk_panda_arm_no_gripper")["q_in"])
    H0 = tree.compute_forward_kinematics_all_links(q)
    tree.update_base_pose(torch.tensor([0.5, 0.0, -0.25, 1.0, 0.0, 0.0, 0.0]))
    H1 = tree.compute_forward_kinematics_all_links(q)
    np.testing.assert_allclose((H1 - H0)[..., :3, 3].cpu().numpy(), np.broadcast_to([0.5, 0.0, -0.25], (32, 11, 3)), atol=1e-6)


ENVS = {"spheres3d": (tra.EnvSpheres3D, {}), "table_shelf": (tra.EnvTableShelf, {}), "maze_boxes3d": (tra.EnvMazeBoxes3D, {}),
        "spheres3d_extra": (tra.EnvSpheres3DExtraObjects, {}),
        "spheres3d_grid": (tra.EnvSpheres3D, dict(precompute_sdf_obj_fixed=True, sdf_cell_size=0.1))}


@pytest.mark.parametrize("env", sorted(ENVS))
def test_planning_task_like_the_reference(env):
    g, rg = gold(f"cost_{env}"), gold("panda_robot")
    cls, kw = ENVS[env]
    robot = tra.RobotPanda(tensor_args=TA)
    e = cls(tensor_args=TA, **kw)
    task = tra.PlanningTask(env=e, robot=robot, obstacle_cutoff_margin=float(g["cutoff"]), tensor_args=TA)
    if "grid_sdf" in g:
        grid = e.grid_map_sdf_obj_fixed
        np.testing.assert_array_equal(grid.cmap_dim.numpy(), g["grid_cmap_dim"])
        assert np.abs(grid.sdf_tensor.cpu().numpy() - g["grid_sdf"]).max() < 2e-6
    q0 = dev(g["q"])                                                   # (8, 8, 7)
    pos = robot.fk_map_collision(q0)
    assert pos.shape == (8, 8, 11, 3)
    assert np.abs(pos.cpu().numpy() - rg["fk_map_collision"]).max() < TOL_H
    names = ["self", "objects", "ws"]
    for fname, fld in zip(names, task.get_collision_fields()):
        q = q0.clone().requires_grad_(True)
        cost = fld.compute_cost(q, robot.fk_map_collision(q), field_type="sdf")
        assert cost.shape == (8, 8)
        assert rel_err(cost.detach().cpu().numpy(), g[f"cost_{fname}"]) < TOL_C, fname
        cost.sum().backward()
        assert grad_close(q.grad.cpu().numpy(), g[f"gq_{fname}"]), fname
        coll = fld.compute_cost(q0, pos, field_type="occupancy")
        np.testing.assert_array_equal(coll.cpu().numpy(), g[f"coll_{fname}"])
        coll0 = fld.compute_cost(q0, pos, field_type="occupancy", margin=0.0)
        np.testing.assert_array_equal(coll0.cpu().numpy(), g[f"coll0_{fname}"])
    if "cost_extra" in g:
        fld = task.get_collision_fields_extra_objects()[0]
        assert rel_err(fld.compute_cost(q0, pos, field_type="sdf").cpu().numpy(), g["cost_extra"]) < TOL_C
    q = q0.clone().requires_grad_(True)
    total = task.compute_collision_cost(q)
    assert total.shape == (8, 8)
    assert rel_err(total.detach().cpu().numpy(), g["cost_total"]) < TOL_C
    total.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq_total"])
    assert rel_err(task.compute_collision_cost(q0).cpu().numpy(), g["cost_total"]) < TOL_C      # no-grad fast path
    np.testing.assert_array_equal(task.compute_collision(q0).cpu().numpy(), g["coll_total"])
    np.testing.assert_array_equal(task.compute_collision(q0, margin=0.0).cpu().numpy(), g["coll0_total"])
    # q of rank 2 and 1 (tasks.py:146-155)
    assert task.compute_collision_cost(q0[0]).shape == (8, 1)
    assert task.compute_collision_cost(q0[0, 0]).shape == (1, 1)


def test_ee_field_like_the_reference():
    g = gold("cost_ee")
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    for k in (0, 1, 3):
        for sq in (True, False):
            key = f"t{k}_sq{int(sq)}_w2.0_0.5"
            fld = tra.EESE3DistanceField(dev(g[f"target_{k}"]), w_pos=2.0, w_rot=0.5, square=sq, tensor_args=TA)
            q = dev(g["q"].reshape(-1, 7)).requires_grad_(True)
            H = tree.compute_forward_kinematics_all_links(q)
            cost = fld.compute_costs_impl(q, H)
            assert rel_err(cost.detach().cpu().numpy(), g["cost_" + key]) < TOL_C
            cost.sum().backward()
            assert grad_close(q.grad.cpu().numpy(), g["gq_" + key])
    fld = tra.EESE3DistanceField(dev(g["target_0"]), tensor_args=TA)
    d = fld.compute_distance(tree.compute_forward_kinematics_all_links(dev(g["q"].reshape(-1, 7))))
    assert rel_err(d.cpu().numpy(), g["cost_t0_sq0_w1.0_1.0"]) < TOL_C
    fld.update_target(dev(g["target_1"]))
    d = fld.compute_distance(tree.compute_forward_kinematics_all_links(dev(g["q"].reshape(-1, 7))))
    assert rel_err(d.cpu().numpy(), g["cost_t1_sq0_w1.0_1.0"]) < TOL_C


def test_frame_algebra_on_fk_output():
    """Frame.transform_point / inverse / multiply_transform on the frames of `return_dict=True` (frame.py:55-78, 116-118):
    transform_point of the grasped-object frame reproduces the reference's grasped columns of fk_map_collision."""
    g = gold("grasp_panda")
    tree = tra.DifferentiableFrankaPanda(device=DEV, grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA))
    q = dev(g["q"].reshape(-1, 7))
    frames = tree.compute_forward_kinematics_all_links(q, return_dict=True)
    pts = frames["grasped_object"].transform_point(dev(g["base_points"]))
    assert np.abs(pts.cpu().numpy() - g["link_pos"].reshape(-1, 26, 3)[:, 12:]).max() < TOL_H
    f = frames["panda_hand"]
    ident = f.multiply_transform(f.inverse())
    assert np.abs(ident.rotation.cpu().numpy() - np.eye(3)).max() < 1e-6 and np.abs(ident.translation.cpu().numpy()).max() < 1e-6
    assert len(f.get_euler()) == 3


def test_frame_algebra_vs_reference_golden():
    """Frame.inverse / multiply_transform / multiply_inv_transform / transform_point / get_quaternion (XYZW, trace method) /
    get_euler and their gradients against the reference's own Frame (tests/golden/frame_algebra.npz, frame.py:55-121)."""
    from torch_robotics_amd.kinematics import Frame
    g = gold("frame_algebra")
    leaf = {k: dev(g[k]).requires_grad_(True) for k in ("Ra", "ta", "Rb", "tb")}
    fa, fb = Frame(leaf["Ra"], leaf["ta"]), Frame(leaf["Rb"], leaf["tb"])
    wR, wt, wp = dev(g["wR"]), dev(g["wt"]), dev(g["wp"])

    def check(name, frame, keys):
        assert np.abs(frame.rotation.detach().cpu().numpy() - g[f"{name}_R"]).max() < 1e-6, name
        assert np.abs(frame.translation.detach().cpu().numpy() - g[f"{name}_t"]).max() < 2e-6, name
        grads = torch.autograd.grad((frame.rotation * wR).sum() + (frame.translation * wt).sum(), [leaf[k] for k in keys])
        for k, gk in zip(keys, grads):
            assert np.abs(gk.cpu().numpy() - g[f"{name}_g{k}"]).max() < 5e-6, (name, k)

    check("inv", fa.inverse(), ["Ra", "ta"])
    check("mul", fa.multiply_transform(fb), ["Ra", "ta", "Rb", "tb"])
    check("mulinv", fa.multiply_inv_transform(fb), ["Ra", "ta", "Rb", "tb"])
    m1 = Frame(dev(g["R1"]), dev(g["t1"])).multiply_transform(Frame(dev(g["Rb"]), dev(g["tb"])))     # batch-1 broadcast
    assert np.abs(m1.rotation.cpu().numpy() - g["mul1_R"]).max() < 1e-6 and np.abs(m1.translation.cpu().numpy() - g["mul1_t"]).max() < 2e-6
    # a broadcast frame that needs a gradient gets the batch-summed adjoint
    R1 = dev(g["R1"]).requires_grad_(True)
    m1g = Frame(R1, dev(g["t1"])).multiply_transform(Frame(dev(g["Rb"]), dev(g["tb"])))
    (gR1,) = torch.autograd.grad((m1g.rotation * wR).sum(), [R1])
    ref = np.einsum("nij,nkj->ik", g["wR"].astype(np.float64), g["Rb"].astype(np.float64))
    assert gR1.shape == (1, 3, 3) and np.abs(gR1[0].cpu().numpy() - ref).max() < 2e-5
    tp = fa.transform_point(dev(g["pts"]))
    assert np.abs(tp.detach().cpu().numpy() - g["tp"]).max() < 2e-6
    gRa, gta = torch.autograd.grad((tp * wp).sum(), [leaf["Ra"], leaf["ta"]])
    assert np.abs(gRa.cpu().numpy() - g["tp_gRa"]).max() < 1e-5 and np.abs(gta.cpu().numpy() - g["tp_gta"]).max() < 1e-5
    quat = fa.get_quaternion().detach()                         # XYZW like the reference, not WXYZ (differentiable w.r.t. Ra)
    assert np.abs(quat.cpu().numpy() - g["quat_xyzw"]).max() < 1e-6
    np.testing.assert_array_equal(quat[-4:].cpu().numpy(), [[0, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])
    wxyz = fa.get_quaternion_wxyz().detach().cpu().numpy()
    assert np.abs(np.abs((wxyz[:, [1, 2, 3, 0]] * g["quat_xyzw"]).sum(1)) - 1).max() < 1e-5      # same rotation, other order
    assert np.abs(torch.stack(fa.get_euler(), -1).detach().cpu().numpy() - g["euler"]).max() < 2e-6
    assert np.abs(fa.get_transform_matrix().detach().cpu().numpy() - g["H"]).max() == 0


def test_rotation_builders_and_frame_constructor():
    """x_rot / y_rot / z_rot (spatial_vector.py:8-47) with their gradients, q_to_rotation_matrix (quaternion.py:102-120) and the
    reference's Frame constructor (identity defaults, pose = x y z qw qx qy qz) against vectors from the reference."""
    g = gold("frame_algebra")
    ang = dev(g["angle"]).requires_grad_(True)
    wR = dev(g["wR"])
    for nm, fn in (("x", tra.x_rot), ("y", tra.y_rot), ("z", tra.z_rot)):
        R = fn(ang.unsqueeze(1))                                  # (B,1), as rigid_body.py calls it
        assert R.shape == (64, 3, 3) and np.abs(R.detach().cpu().numpy() - g[f"rot_{nm}"]).max() < 3e-7
        (ga,) = torch.autograd.grad((R * wR).sum(), [ang])
        assert np.abs(ga.cpu().numpy() - g[f"rot_{nm}_gangle"]).max() < 2e-6
        assert fn(ang.detach()[0]).shape == (1, 3, 3)             # 0-d angle gets a batch dimension
    Rq = tra.q_to_rotation_matrix(dev(g["quat_in"]))
    assert np.abs(Rq.cpu().numpy() - g["quat_R"]).max() < 1e-6 * np.abs(g["quat_R"]).max() + 1e-6
    ident = tra.Frame(device=DEV)
    assert ident.batch_size == 1 and torch.equal(ident.rotation[0].cpu(), torch.eye(3)) and not ident.translation.any()
    pose = torch.tensor([0.1, -0.2, 0.3, 0.5, 0.5, -0.5, 0.5])
    fp = tra.Frame(pose=pose, device=DEV)
    from torch_robotics_amd.kinmodel import quat_wxyz_to_rot
    assert np.abs(fp.rotation[0].cpu().numpy() - quat_wxyz_to_rot(pose[3:].numpy())).max() < 1e-6
    np.testing.assert_allclose(fp.translation.cpu().numpy(), [[0.1, -0.2, 0.3]])
    one = tra.Frame(rot=torch.eye(3), trans=torch.tensor([1.0, 2.0, 3.0]), device=DEV)      # 2-D / 1-D inputs get a batch dim
    assert one.rotation.shape == (1, 3, 3) and one.get_transform_matrix().shape == (1, 4, 4)


def test_examples_run_like_the_reference_examples():
    """examples/forward_kinematics.py and examples/inverse_kinematics.py: the reference's two example scripts on this package."""
    import importlib.util
    from pathlib import Path
    ex = Path(__file__).resolve().parent.parent / "examples"

    def load(name):
        spec = importlib.util.spec_from_file_location(name, ex / f"{name}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    res = load("forward_kinematics").main(batch_size=10, device="cuda:0", verbose=False)
    assert set(res) == {"Panda", "UR10", "Habitat Stretch", "Tiago", "Shadow Hand", "Allegro Hand"}
    H, gq = res["Panda"]
    assert H.shape == (10, 11, 4, 4) and gq.shape == (10, 7) and torch.isfinite(gq).all()
    q_ik, idx_valid, err, H_target = load("inverse_kinematics").main(batch_size=16, device="cuda:0", verbose=False)
    # with the example's lr = 0.2 a few samples keep oscillating around the target (the reference's do too); how many are inside
    # se3_eps at the last test depends on rounding (11 - 12 of 16 here, 4 - 12 over other seeds: tools/ik_compare.py)
    assert q_ik.shape == (16, 7) and idx_valid.nelement() >= 4           # (the range over seeds; the example's own seed gives 11 - 12)
    # se3_eps of the call is 5e-2 -- tested, like the reference does (robot_tree.py:349-377), BEFORE the last Adam step: the returned q of a
    # valid sample is one step past the tested one (seed soak: 0.0512 for one of ten)
    assert float(err[idx_valid].median()) < 5e-2 and float(err[idx_valid].max()) < 1e-1
    np.testing.assert_allclose(H_target[0, :3, 3].cpu().numpy(), [0.2, 0.4, 0.1], atol=1e-7)
    # a batch trajectory optimiser on the fused kernels (hinge collision costs + GP prior), validated like the reference does
    q_opt, n_free, coll0 = load("plan_trajectories").main(batch=64, horizon=64, iters=300, device="cuda:0", verbose=False)
    assert q_opt.shape == (64, 64, 7) and torch.isfinite(q_opt).all()
    # most straight lines collide; the optimiser frees many of them -- how many depends on the drawn starts and goals (40+ of 64 at the
    # example's seed, 23 at the worst of thirteen others): a quarter of the batch more than before is the smoke bound
    assert coll0 > 0.5 and n_free >= 64 * (1.0 - coll0) + 16, (coll0, n_free)


def test_host_tensors_round_trip_like_the_reference_example():
    """The reference's own example constructs its trees with device="cpu" (examples/forward_kinematics.py:13-25; robot_tree.py:77
    defaults to it).  Here the COMPUTE stays on the GPU: host tensors are copied over, the HIP kernels run, results come back to the
    host with autograd intact -- bit-identical to the same call on GPU tensors.  Trees, robot, fields, task and scene queries."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("ex_fk", Path(__file__).resolve().parent.parent / "examples" / "forward_kinematics.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    res_cpu = mod.main(batch_size=10, device="cpu", verbose=False)
    res_gpu = mod.main(batch_size=10, device="cuda:0", verbose=False)
    assert set(res_cpu) == set(res_gpu) and len(res_cpu) == 6
    for k in res_cpu:
        H_c, g_c = res_cpu[k]
        H_g, g_g = res_gpu[k]
        assert H_c.device.type == "cpu" and g_c.device.type == "cpu" and H_g.is_cuda
        assert torch.equal(H_c, H_g.cpu()) and torch.equal(g_c, g_g.cpu())
    # the signature's default is the reference's
    tree = tra.DifferentiableFrankaPanda()
    assert tree._device.type == "cpu"
    q = torch.rand(5, 7)
    q.requires_grad_(True)
    d = tree.compute_forward_kinematics_all_links(q, return_dict=True, link_list=["ee_link"])
    assert d["ee_link"].rotation.device.type == "cpu" and d["ee_link"].get_transform_matrix().shape == (5, 4, 4)
    assert d["ee_link"].get_quaternion().device.type == "cpu"              # the Frame algebra round-trips too
    pos, quat, lin, ang = tree.compute_forward_kinematics_and_geometric_jacobian(q.detach(), torch.zeros(5, 7), "ee_link")
    assert all(t.device.type == "cpu" for t in (pos, quat, lin, ang)) and lin.shape == (5, 3, 7)
    assert tra.link_quat_from_link_tensor(tree.compute_forward_kinematics_all_links(q.detach())).device.type == "cpu"
    q_ik, idx = tree.inverse_kinematics(torch.eye(4), batch_size=8, max_iters=3, print_freq=-1)
    assert q_ik.device.type == "cpu" and q_ik.shape == (8, 7)
    # robot / fields / task built with the reference's default tensor_args (host)
    TAc = dict(device="cpu", dtype=torch.float32)
    robot_c, robot_g = tra.RobotPanda(tensor_args=TAc), tra.RobotPanda(tensor_args=TA)
    task_c = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TAc), robot=robot_c, obstacle_cutoff_margin=0.03, tensor_args=TAc)
    task_g = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot_g, obstacle_cutoff_margin=0.03, tensor_args=TA)
    qh = robot_c.random_q(64).reshape(4, 16, 7)
    assert qh.device.type == "cpu"
    qc = qh.clone().requires_grad_(True)
    qg = qh.to(DEV).requires_grad_(True)
    c_c, c_g = task_c.compute_collision_cost(qc), task_g.compute_collision_cost(qg)
    c_c.sum().backward(); c_g.sum().backward()
    assert c_c.device.type == "cpu" and torch.equal(c_c.detach(), c_g.detach().cpu()) and torch.equal(qc.grad, qg.grad.cpu())
    assert torch.equal(task_c.compute_collision(qh), task_g.compute_collision(qh.to(DEV)).cpu())
    lp_c = robot_c.fk_map_collision(qh)
    assert lp_c.device.type == "cpu" and torch.equal(lp_c, robot_g.fk_map_collision(qh.to(DEV)).cpu())
    f_c = task_c.df_collision_objects.compute_cost(qh, lp_c)
    assert f_c.device.type == "cpu" and torch.equal(f_c, task_g.df_collision_objects.compute_cost(qh.to(DEV), lp_c.to(DEV)).cpu())
    X = torch.rand(20, 3) - 0.5
    assert torch.equal(task_c.env.compute_sdf(X), task_g.env.compute_sdf(X.to(DEV)).cpu())
    assert torch.equal(tra.SE3_distance(torch.eye(4).repeat(3, 1, 1), torch.eye(4)), tra.SE3_distance(torch.eye(4, device=DEV).repeat(3, 1, 1), torch.eye(4, device=DEV)).cpu())
    tc, tf = task_c.get_trajs_collision_and_free(qh, num_interpolation=2)
    assert all(t is None or t.device.type == "cpu" for t in (tc, tf))


def test_moved_scene_object_is_seen_by_every_cached_cost_model():
    """`ObjectField.set_position_orientation` between two evaluations: the reference reads the pose on every call
    (primitives.py:387-405), so PlanningTask's fused cost model, the per-field one and EnvBase.compute_sdf must all follow."""
    robot = tra.RobotPanda(tensor_args=TA)
    env = tra.EnvSpheres3DExtraObjects(tensor_args=TA)
    task = tra.PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    q = dev(gold("cost_spheres3d_extra")["q"])
    X = dev(np.array([[0.25, 0.0, 0.0], [2.25, 3.0, 0.5]], np.float32))
    c0, k0, s0 = task.compute_collision_cost(q), task.compute_collision(q), env.compute_sdf(X)
    _, r0, _ = task.rollout_cost_grad(q, w_self=0, w_ws=0)
    f0 = task.df_collision_objects.compute_cost(q, robot.fk_map_collision(q))
    extra = env.obj_extra_list[0]
    extra.set_position_orientation(pos=(2.0, 3.0, 0.5))                   # far outside the robot's reach
    c1, k1, s1 = task.compute_collision_cost(q), task.compute_collision(q), env.compute_sdf(X)
    _, r1, _ = task.rollout_cost_grad(q, w_self=0, w_ws=0)
    f1 = task.df_collision_objects.compute_cost(q, robot.fk_map_collision(q))
    assert (c1 - c0).abs().max() > 1e-3 and (r1 - r0).abs().max() > 1e-3 and (f1 - f0).abs().max() > 1e-3
    assert s0[0] < -0.1 and s1[0] > 0.1 and s1[1] < -0.1                  # the sphere at (0.25,0,0) left, one arrived at X[1]
    assert k1.sum() <= k0.sum()
    # identical to a task built on the moved scene from scratch; the per-field path and the fused path agree
    env2 = tra.EnvSpheres3DExtraObjects(tensor_args=TA)
    env2.obj_extra_list[0].set_position_orientation(pos=(2.0, 3.0, 0.5))
    task2 = tra.PlanningTask(env=env2, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    np.testing.assert_array_equal(task2.compute_collision_cost(q).cpu().numpy(), c1.cpu().numpy())
    np.testing.assert_array_equal(task2.compute_collision(q).cpu().numpy(), k1.cpu().numpy())
    assert rel_err(r1.cpu().numpy(), f1.cpu().numpy()) < TOL_C
    extra.set_position_orientation(pos=(0.0, 0.0, 0.0))                   # and back
    np.testing.assert_array_equal(task.compute_collision_cost(q).cpu().numpy(), c0.cpu().numpy())


def test_reassigned_primitive_geometry_is_seen_by_object_sdf(oracle_lib):
    """`field.centers = new` (the supported edit) must reach `ObjectField.compute_signed_distance` and its autograd path --
    the object's own device cost model is keyed by the fields' geometry versions, not only by device and pose."""
    from torch_robotics_amd.costmodel import CostModelSpec
    from helpers import model
    f = tra.MultiSphereField(np.array([[0.3, 0.0, 0.2], [-0.2, 0.4, 0.5]], np.float32), np.array([0.1, 0.2], np.float32), tensor_args=TA)
    b = tra.MultiBoxField(np.array([[0.0, -0.4, 0.3]], np.float32), np.array([[0.2, 0.3, 0.4]], np.float32), tensor_args=TA)
    o = tra.ObjectField([f, b], "o", pos=(0.1, 0.0, 0.0), ori=(0.9238795, 0.0, 0.3826834, 0.0))
    gen = torch.Generator(device=DEV).manual_seed(21)
    X = (torch.rand(300, 3, device=DEV, generator=gen) - 0.5) * 2.0

    def check():
        x = X.clone().requires_grad_(True)
        s = o.compute_signed_distance(x)
        s.sum().backward()
        orc = oracle_lib.Oracle(model("panda_arm_no_gripper"), CostModelSpec(n_links_in=1, objects=[o.as_object()]))
        rs, rg = orc.sdf_points(X.cpu().numpy(), "f64")
        assert np.abs(s.detach().cpu().numpy() - rs[:, 0]).max() < 2e-6
        assert np.abs(o.compute_signed_distance(X).cpu().numpy() - rs[:, 0]).max() < 2e-6
        gerr = np.abs(x.grad.cpu().numpy() - rg[:, 0, :]).max(1)
        assert (gerr > 1e-4).sum() <= 3                        # arg-min ties between primitives / box faces flip in fp32
        return s.detach().clone()

    s0 = check()
    f.centers = f.centers + np.array([0.0, 0.3, -0.1], np.float32)
    s1 = check()
    assert (s1 - s0).abs().max() > 1e-2
    f.radii = f.radii * 0.5
    s2 = check()
    b.sizes = b.sizes * 1.5
    s3 = check()
    assert (s2 - s1).abs().max() > 1e-2 and (s3 - s2).abs().max() > 1e-3


def test_clamp_sdf_fields_and_task():
    """`clamp_sdf=True` on the fields (distance_fields.py:114-117) and on a PlanningTask: hinge costs like the reference's."""
    g, rg = gold("cost_clamp"), gold("panda_robot")
    robot = tra.RobotPanda(tensor_args=TA)
    env = tra.EnvSpheres3D(tensor_args=TA)
    task = tra.PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, clamp_sdf=True, tensor_args=TA)
    q0 = dev(g["q"])
    for fname, fld in (("objects", task.df_collision_objects), ("ws", task.df_collision_ws_boundaries)):
        assert fld.clamp_sdf
        q = q0.clone().requires_grad_(True)
        cost = fld.compute_cost(q, robot.fk_map_collision(q), field_type="sdf")
        assert rel_err(cost.detach().cpu().numpy(), g[f"spheres3d_cost_{fname}"]) < TOL_C and (cost >= 0).all()
        cost.sum().backward()
        ref = g[f"spheres3d_gq_{fname}"]
        assert np.abs(q.grad.cpu().numpy() - ref).max() < TOL_G * max(1.0, np.abs(ref).max())
    q = q0.clone().requires_grad_(True)
    total = task.compute_collision_cost(q)
    assert rel_err(total.detach().cpu().numpy(), g["spheres3d_cost_total"]) < TOL_C
    total.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["spheres3d_gq_total"])
    # the unclamped task on the same inputs is a different (unbounded below) objective
    plain = tra.PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA).compute_collision_cost(q0)
    assert (plain < 0).any() and (total >= 0).all()


def test_rollout_rejects_q_of_another_width():
    """A (B,H,D') tensor with D' != n_dofs (full state with velocities) must raise instead of being read with a wrong stride."""
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, tensor_args=TA)
    model, cm = task._fused_handles(DEV)
    x = torch.zeros(4, 8, 14, device=DEV)
    for bad in (x, x[..., :6], x.to(torch.float16)):
        with pytest.raises(ValueError, match="DOF"):
            tra.ops.rollout_cost_grad(model, cm, (1, 1, 1, 0), bad)
        with pytest.raises(ValueError, match="DOF"):
            tra.ops.RolloutPlan(model, cm, (1, 1, 1, 0), bad)
    q = x[..., :7].contiguous()
    n = 32
    good = (torch.empty(n, 11, 3, device=DEV), torch.empty(n, device=DEV), torch.empty(n, 7, device=DEV))
    tra.ops.rollout_cost_grad(model, cm, (1, 1, 1, 0), q, out=good)
    for k, bad_buf in ((0, torch.empty(n, 10, 3, device=DEV)), (1, torch.empty(n, device=DEV, dtype=torch.float64)),
                       (2, torch.empty(n, 7)), (2, torch.empty(n, 14, device=DEV)[:, ::2])):
        out = list(good)
        out[k] = bad_buf
        with pytest.raises(ValueError):
            tra.ops.rollout_cost_grad(model, cm, (1, 1, 1, 0), q, out=tuple(out))
    with pytest.raises(ValueError):
        tra.ops.rollout_cost_grad(model, cm, (1, 1, 1, 0), q, cost_sum=torch.zeros(0, device=DEV))
    task.rollout_cost_grad(x)                                   # the Task API slices the positions out of a full state itself


def test_se3_distance_function():
    """SE3_distance (geometrics/utils.py:130-178) as a function: values and gradient w.r.t. the poses vs the reference."""
    g = gold("cost_ee")
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    q = dev(g["q"].reshape(-1, 7))
    for t in range(4):
        for wp, wr in ((1.0, 1.0), (2.0, 0.5)):
            H = tree.compute_forward_kinematics_all_links(q, link_list=["ee_link"])[:, 0].detach().requires_grad_(True)
            d = tra.SE3_distance(H, dev(g[f"target_{t}"]), w_pos=wp, w_rot=wr)
            assert d.shape == (64,)
            assert rel_err(d.detach().cpu().numpy(), g[f"cost_t{t}_sq0_w{wp}_{wr}"]) < TOL_C
            d.sum().backward()
            assert grad_close(H.grad.cpu().numpy()[:, :3, :], g[f"gH_t{t}_sq0_w{wp}_{wr}"][:, -1, :3, :])
    with pytest.raises(NotImplementedError):
        tra.SE3_distance(H, dev(g["target_0"]), vel_batch=H)


def test_fused_task_rollout():
    g = gold("rollout_panda")
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    task.set_ee_target(g["target"])
    q = dev(g["q"])
    pos, c2, g2 = task.rollout_cost_grad(q, w_self=0, w_obj=1, w_ws=0, w_ee=1)
    assert rel_err(c2.cpu().numpy(), g["cost_c2"]) < TOL_C and grad_close(g2.cpu().numpy(), g["gq_c2"])
    assert np.abs(pos.cpu().numpy() - g["pos"]).max() < TOL_H
    _, c3, g3 = task.rollout_cost_grad(q, w_self=1, w_obj=1, w_ws=1, w_ee=1, want_pos=False)
    assert rel_err(c3.cpu().numpy(), g["cost_c3"]) < TOL_C and grad_close(g3.cpu().numpy(), g["gq_c3"])
    # ObjectField.compute_signed_distance on arbitrary points, differentiable
    obj = task.env.obj_fixed_list[0]
    x = dev(g["pos"][0, :, 5]).requires_grad_(True)
    sd = obj.compute_signed_distance(x)
    sd.sum().backward()
    assert sd.shape == (64,) and torch.isfinite(x.grad).all()
    assert np.abs(np.linalg.norm(x.grad.cpu().numpy(), axis=-1) - 1).max() < 1e-5


def test_rollout_plan_matches_rollout_cost_grad():
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(T)
    q = robot.random_q(5 * 64).reshape(5, 64, 7).contiguous()
    plan = task.rollout_plan(q, w_ee=1.0)
    for _ in range(2):
        plan.launch()
        pos, cost, gq = task.rollout_cost_grad(q, w_ee=1.0)
        torch.cuda.synchronize()
        assert torch.equal(plan.cost, cost) and torch.equal(plan.gq, gq) and torch.equal(plan.link_pos, pos)
        q.copy_(robot.random_q(5 * 64).reshape(5, 64, 7))      # the plan reads q in place


def test_trajectory_validation_like_the_reference():
    """8f rank 1: get_trajs_collision_and_free + stats (tasks.py:234-328) and interpolate_traj_via_points."""
    g, gt = gold("trajs_panda"), gold("traj")
    from torch_robotics_amd import ops
    # a trajectory too large for one workgroup's LDS takes the element-per-thread kernel: same bits as the per-trajectory kernel
    big = torch.randn(3, 2100, 7, device=DEV)
    ib = ops.interpolate_traj_via_points(big, num_interpolation=3)
    ref = torch.stack([ops.interpolate_traj_via_points(big[:, k:k + 2].contiguous(), num_interpolation=3) for k in (0, 1000, 2098)])
    assert ib.shape == (3, 2099 * 3, 7) and all(torch.equal(ib[:, 3 * k:3 * k + 3], ref[j]) for j, k in enumerate((0, 1000, 2098)))
    interp = ops.interpolate_traj_via_points(dev(gt["x"]), num_interpolation=5)
    np.testing.assert_array_equal(interp.cpu().numpy(), gt["interp5"])          # bit-exact
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    trajs = dev(g["trajs"])
    coll, coll_idx, free, free_idx, wp = task.get_trajs_collision_and_free(trajs, return_indices=True, num_interpolation=5)
    np.testing.assert_array_equal(wp.cpu().numpy(), g["waypoints_collisions"])
    np.testing.assert_array_equal(coll_idx.cpu().numpy(), g["coll_idx"])
    np.testing.assert_array_equal(free_idx.cpu().numpy(), g["free_idx"])
    assert coll.shape[0] == int(g["n_coll"]) and free.shape[0] == int(g["n_free"])
    assert task.compute_fraction_free_trajs(trajs) == pytest.approx(float(g["fraction_free"]))
    assert float(task.compute_collision_intensity_trajs(trajs)) == pytest.approx(float(g["collision_intensity"]))
    assert task.compute_success_free_trajs(trajs) == int(g["success"])
    c4, ci4, f4, fi4, wp4 = task.get_trajs_collision_and_free(trajs.reshape(3, 4, 16, 7), return_indices=True)
    np.testing.assert_array_equal(wp4.cpu().numpy(), g["waypoints_collisions4"])
    np.testing.assert_array_equal(ci4.cpu().numpy(), g["coll_idx4"])
    np.testing.assert_array_equal(fi4.cpu().numpy(), g["free_idx4"])


def test_inverse_kinematics_like_the_example():
    """examples/inverse_kinematics.py: a batch of IK problems converges to valid configurations."""
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    torch.manual_seed(0)
    q_goal = (torch.rand(1, 7, device=DEV) - 0.5) * 2.0
    H_target = tree.compute_forward_kinematics_all_links(q_goal, link_list=["ee_link"]).squeeze(1)
    q, idx_valid = tree.inverse_kinematics(H_target, link_name="ee_link", batch_size=64, max_iters=800, lr=2e-2,
                                           se3_eps=5e-2, print_freq=-1, check_every=10)
    assert q.shape == (64, 7)
    assert idx_valid.numel() >= 16                               # gradient IK has local minima: a good share of the restarts converge
    H = tree.compute_forward_kinematics_all_links(q[idx_valid], link_list=["ee_link"]).squeeze(1)
    assert (H[:, :3, 3] - H_target[0, :3, 3]).norm(dim=-1).max() < 5e-2
    lo, hi, _, _ = tree.get_joint_limit_array()
    qv = q[idx_valid].cpu().numpy()
    assert np.all(qv >= lo) and np.all(qv <= hi)
    # loss_fn_ik_per_q / ik_termination helpers agree with the golden of the reference
    g = gold("ik_panda")
    loss = tree.loss_fn_ik_per_q(dev(g["q0"]), dev(g["H_target"]), "ee_link", w_joint_limits=300.0,
                                 lower=dev(g["lower"]), upper=dev(g["upper"]))
    assert rel_err(loss.cpu().numpy(), g["loss0_per_sample"]) < TOL_C


def test_grasped_object_like_the_reference():
    """RobotPanda holding a box: fk_map_collision (pinned by the reference), the three fields and the
    task-level cost over robot links + grasped points (pinned by the reference's field code on explicit columns)."""
    g = gold("grasp_panda")
    robot = tra.RobotPanda(grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA), tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=float(g["cutoff"]),
                            tensor_args=TA)
    q0 = dev(g["q"])
    q = q0.clone().requires_grad_(True)
    pos = robot.fk_map_collision(q)
    assert pos.shape == (4, 8, 26, 3)
    assert np.abs(pos.detach().cpu().numpy() - g["link_pos"]).max() < TOL_H
    (pos * dev(g["w"])).sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq"])
    p2 = robot.fk_map_collision(dev(g["q_out"]))
    assert np.abs(p2.cpu().numpy() - g["link_pos_out"]).max() < TOL_H
    for key, fld in zip(("self", "obj", "ws"), task.get_collision_fields()):
        cost = fld.compute_cost(q0, pos.detach(), field_type="sdf")
        assert cost.shape == (4, 8)
        assert rel_err(cost.cpu().numpy(), g[f"cost_{key}"]) < TOL_C, key
        np.testing.assert_array_equal(fld.compute_cost(q0, pos.detach(), field_type="occupancy").cpu().numpy(), g[f"coll_{key}"])
        np.testing.assert_array_equal(fld.compute_cost(q0, pos.detach(), field_type="occupancy", margin=0.0).cpu().numpy(),
                                      g[f"coll0_{key}"])
    total_ref = g["cost_self"] + g["cost_obj"] + g["cost_ws"]
    q = q0.clone().requires_grad_(True)
    total = task.compute_collision_cost(q)
    assert rel_err(total.detach().cpu().numpy(), total_ref) < TOL_C
    total.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq_cost"])
    assert rel_err(task.compute_collision_cost(q0).cpu().numpy(), total_ref) < TOL_C
    np.testing.assert_array_equal(task.compute_collision(q0).cpu().numpy(), g["coll_self"] | g["coll_obj"] | g["coll_ws"])
    np.testing.assert_array_equal(task.compute_collision(q0, margin=0.0).cpu().numpy(),
                                  g["coll0_self"] | g["coll0_obj"] | g["coll0_ws"])
    ppos, cost, gq = task.rollout_cost_grad(q0)
    assert ppos.shape == (4, 8, 26, 3) and grad_close(gq.cpu().numpy(), g["gq_cost"])
    # generated kernel (spec_panda_grasp) vs table-driven kernel, both against the reference-derived golden
    assert robot._point_set(torch.device(DEV)).specialized
    model, _ = task._fused_handles(torch.device(DEV))
    model.enable_specialized(False)
    _, cost_g, gq_g = task.rollout_cost_grad(q0)
    model.enable_specialized(True)
    for c, gr in ((cost, gq), (cost_g, gq_g)):
        assert rel_err(c.cpu().numpy(), total_ref) < TOL_C and grad_close(gr.cpu().numpy(), g["gq_cost"])
    task.set_ee_target(np.array([[1, 0, 0, 0.4], [0, 1, 0, 0.2], [0, 0, 1, 0.5], [0, 0, 0, 1]], np.float32))
    model, _ = task._fused_handles(torch.device(DEV))
    _, c_s, g_s = task.rollout_cost_grad(q0, w_ee=1.0)
    model.enable_specialized(False)
    _, c_g, g_g = task.rollout_cost_grad(q0, w_ee=1.0)
    model.enable_specialized(True)
    assert rel_err(c_s.cpu().numpy(), c_g.cpu().numpy()) < TOL_C and grad_close(g_s.cpu().numpy(), g_g.cpu().numpy())


def test_link_sphere_model(oracle_lib):
    """RobotPanda(link_sphere_model="panda"): 45 link-frame collision spheres (the reference ships the table
    but has no code path for it, so the check is the fp64 oracle on the same tables -- parity build-defined)."""
    robot = tra.RobotPanda(link_sphere_model="panda", tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(T)
    spec = task.build_cost_spec()
    pl, po = robot.collision_point_set()
    o = oracle_lib.Oracle(robot.diff_panda._kin, spec)
    gen = torch.Generator(device=DEV).manual_seed(3)
    q0 = robot.random_q(6 * 64, generator=gen).reshape(6, 64, 7)
    qn = q0.cpu().numpy().astype(np.float64).reshape(-1, 7)
    rp, rc, rg = o.rollout_points(pl, po, qn, (1, 1, 1, 0), "f64")
    q = q0.clone().requires_grad_(True)
    pos = robot.fk_map_collision(q)
    assert pos.shape == (6, 64, 56, 3)
    assert np.abs(pos.detach().cpu().numpy().reshape(-1, 56, 3) - rp).max() < TOL_H
    total = task.compute_collision_cost(q)
    assert rel_err(total.detach().cpu().numpy().reshape(-1), rc) < TOL_C
    total.sum().backward()
    assert grad_close(q.grad.cpu().numpy().reshape(-1, 7), rg)
    # unfused chain through the drop-in classes == fused kernel
    q = q0.clone().requires_grad_(True)
    lp = robot.fk_map_collision(q)
    chain = sum(f.compute_cost(q, lp, field_type="sdf") for f in task.get_collision_fields())
    assert rel_err(chain.detach().cpu().numpy().reshape(-1), rc) < TOL_C
    chain.sum().backward()
    assert grad_close(q.grad.cpu().numpy().reshape(-1, 7), rg)
    # with the EE term
    _, rc4, rg4 = o.rollout_points(pl, po, qn, (1, 1, 1, 1), "f64")
    _, c4, g4 = task.rollout_cost_grad(q0, w_ee=1.0)
    assert rel_err(c4.cpu().numpy().reshape(-1), rc4) < TOL_C and grad_close(g4.cpu().numpy().reshape(-1, 7), rg4)
    coll = task.compute_collision(q0)
    ref = o.collision_fields(7, rp, None, "f64").reshape(6, 64)
    assert (coll.cpu().numpy() != ref).mean() < 0.01       # fp32 vs fp64 at the threshold
    # the generated kernel (this point set is baked into spec_panda_spheres) and the table-driven one agree
    ps = robot._point_set(torch.device(DEV))
    assert ps.specialized
    model, cm = task._fused_handles(torch.device(DEV))
    from torch_robotics_amd import ops
    for wts in ((1, 1, 1, 1), (0, 1, 0, 0), (1, 0, 0, 0), (0, 0, 1, 1)):
        _, rcw, rgw = o.rollout_points(pl, po, qn, wts, "f64")
        for n_rows in (6 * 64, 70, 1):
            qq = q0.reshape(-1, 7)[:n_rows].contiguous()
            model.enable_specialized(True)
            ps_, cs, gs = ops.rollout_points_cost_grad(ps, cm, wts, qq)
            model.enable_specialized(False)
            pg, cg, gg = ops.rollout_points_cost_grad(ps, cm, wts, qq)
            model.enable_specialized(True)
            assert np.abs(ps_.cpu().numpy() - rp[:n_rows]).max() < TOL_H and np.abs(pg.cpu().numpy() - rp[:n_rows]).max() < TOL_H
            assert rel_err(cs.cpu().numpy(), rcw[:n_rows]) < TOL_C and rel_err(cg.cpu().numpy(), rcw[:n_rows]) < TOL_C
            assert grad_close(gs.cpu().numpy(), rgw[:n_rows]) and grad_close(gg.cpu().numpy(), rgw[:n_rows])
        nb = ops.n_blocks(6 * 64)
        sums = torch.zeros(nb, device=DEV)
        _, cs, _ = ops.rollout_points_cost_grad(ps, cm, wts, q0, want_pos=False, cost_sum=sums)
        assert abs(float(sums.sum()) - float(cs.double().sum())) < 1e-3 * max(1.0, abs(float(cs.double().sum())))


def test_sphere_model_reduces_to_link_origin_goldens():
    """One zero-offset sphere per collision link with radius = link margin is the reference's own collision model:
    the point kernels must then reproduce the reference goldens of the link-origin path."""
    from torch_robotics_amd import ops
    from helpers import model, panda_cost_spec
    g, robot_g = gold("cost_spheres3d"), gold("panda_robot")
    spec = panda_cost_spec(g, robot_g)
    m = model("panda_arm_no_gripper")
    h = ops.ModelHandle(m)
    ps = ops.PointSetHandle(h, np.arange(11, dtype=np.int32), np.zeros((11, 3), np.float32), DEV)
    cm = ops.CostHandle(spec, DEV)
    _, cost, gq = ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 0), dev(g["q"]))
    assert rel_err(cost.cpu().numpy(), g["cost_total"]) < TOL_C
    assert grad_close(gq.cpu().numpy(), g["gq_total"])


def test_spheres_and_grasped_box_together(oracle_lib):
    """Link spheres + grasped box (71 columns): generated kernel spec_panda_spheres_grasp vs table-driven vs fp64 oracle."""
    from torch_robotics_amd import ops
    robot = tra.RobotPanda(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA), tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.4, 0.2, 0.5)
    task.set_ee_target(T)
    pl, po = robot.collision_point_set()
    o = oracle_lib.Oracle(robot.diff_panda._kin, task.build_cost_spec())
    gen = torch.Generator(device=DEV).manual_seed(4)
    q0 = robot.random_q(3 * 64 + 5, generator=gen)
    rp, rc, rg = o.rollout_points(pl, po, q0.cpu().numpy().astype(np.float64), (1, 1, 1, 1), "f64")
    ps = robot._point_set(torch.device(DEV))
    assert ps.specialized and ps.n_points == 71
    model, cm = task._fused_handles(torch.device(DEV))
    for use_spec in (True, False):
        model.enable_specialized(use_spec)
        pos, cost, gq = ops.rollout_points_cost_grad(ps, cm, (1, 1, 1, 1), q0)
        assert np.abs(pos.cpu().numpy() - rp).max() < TOL_H
        assert rel_err(cost.cpu().numpy(), rc) < TOL_C and grad_close(gq.cpu().numpy(), rg)
    model.enable_specialized(True)
    # fk_map_collision and its reverse mode: generated (positions-only exit / k_posbwd) vs table-driven vs fp64 oracle
    w = torch.randn(q0.shape[0], 71, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    rgq = o.fk_points_backward(pl, po, q0.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), "f64")
    for use_spec in (True, False):
        model.enable_specialized(use_spec)
        assert np.abs(ops.fk_points(ps, q0).cpu().numpy() - rp).max() < TOL_H
        assert grad_close(ops.fk_points_backward(ps, q0, w).cpu().numpy(), rgq)
    model.enable_specialized(True)


def test_planning_task_specialises_itself_at_run_time():
    """A task whose cost model no ahead-of-time unit serves (EE tracked on another link) compiles + loads its own fused kernel
    on first use; results equal the table-driven kernel's."""
    from torch_robotics_amd import jit
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.3, 0.3, 0.6)
    task.set_ee_target(T, link_name="panda_link7")
    assert not jit.has_matching_unit(robot.diff_panda._kin, task.build_cost_spec())
    q = robot.random_q(3 * 64).reshape(3, 64, 7)
    pos, cost, gq = task.rollout_cost_grad(q, w_ee=1.0)                       # triggers the run-time compile
    assert jit.has_matching_unit(robot.diff_panda._kin, task.build_cost_spec())
    ref = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA,
                           auto_specialize=False)
    ref.set_ee_target(T, link_name="panda_link7")
    model, _ = ref._fused_handles(torch.device(DEV))
    model.enable_specialized(False)
    pos_g, cost_g, gq_g = ref.rollout_cost_grad(q, w_ee=1.0)
    model.enable_specialized(True)
    assert np.abs(pos.cpu().numpy() - pos_g.cpu().numpy()).max() < 2 * TOL_H
    assert rel_err(cost.cpu().numpy(), cost_g.cpu().numpy()) < TOL_C and grad_close(gq.cpu().numpy(), gq_g.cpu().numpy())


def test_custom_sphere_table_gets_its_own_kernel(tmp_path, oracle_lib):
    """RobotPanda with ANOTHER link-sphere table: no ahead-of-time unit has that point set, so the task compiles one at
    run time (jit.specialize_points); generated vs table-driven vs fp64 oracle."""
    from torch_robotics_amd import jit, ops
    table = tmp_path / "spheres.yaml"
    table.write_text("panda_link2:\n- [0.0, 0.0, 0.05, 0.07]\n- [0.0, -0.1, 0.0, 0.06]\n"
                     "panda_link5:\n- [0.0, 0.06, 0.0, 0.06]\n- [0.0, 0.0, -0.2, 0.055]\n- [0.01, 0.08, -0.1, 0.03]\n"
                     "panda_hand:\n- [0.0, 0.04, 0.02, 0.03]\n- [0.0, -0.04, 0.02, 0.03]\n- [0.0, 0.0, 0.06, 0.025]\n")
    robot = tra.RobotPanda(link_sphere_model=str(table), tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.02, tensor_args=TA)
    pl, po = robot.collision_point_set()
    assert len(pl) == 11 + 8
    assert not jit.has_matching_points_unit(robot.diff_panda._kin, pl, po, task.build_cost_spec())
    q = robot.random_q(2 * 64 + 3)
    pos, cost, gq = task.rollout_cost_grad(q)                                 # compiles + loads the unit
    ps = robot._point_set(torch.device(DEV))
    assert ps.specialized and jit.has_matching_points_unit(robot.diff_panda._kin, pl, po, task.build_cost_spec())
    o = oracle_lib.Oracle(robot.diff_panda._kin, task.build_cost_spec())
    rp, rc, rg = o.rollout_points(pl, po, q.cpu().numpy().astype(np.float64), (1, 1, 1, 0), "f64")
    model, cm = task._fused_handles(torch.device(DEV))
    model.enable_specialized(False)
    pos_g, cost_g, gq_g = task.rollout_cost_grad(q)
    model.enable_specialized(True)
    for p_, c_, g_ in ((pos, cost, gq), (pos_g, cost_g, gq_g)):
        assert np.abs(p_.cpu().numpy() - rp).max() < TOL_H
        assert rel_err(c_.cpu().numpy(), rc) < TOL_C and grad_close(g_.cpu().numpy(), rg)


def test_trajectory_metrics_like_the_reference():
    """A17: get_velocity / get_acceleration (finite differences), compute_path_length, compute_smoothness vs the reference."""
    g, gm = gold("traj"), gold("metrics_panda")
    robot = tra.RobotPanda(tensor_args=TA)
    for m in ("forward", "backward", "central"):
        out = tra.finite_difference_vector(dev(g["x"]), dt=0.25, method=m)
        np.testing.assert_allclose(out.cpu().numpy(), g["fd_" + m], rtol=0, atol=0)
    trajs, full = dev(gm["trajs"]), dev(gm["full"])
    np.testing.assert_allclose(robot.get_velocity(trajs).cpu().numpy(), gm["vel_fd"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(robot.get_acceleration(trajs).cpu().numpy(), gm["acc_fd"], rtol=0, atol=1e-7)
    assert rel_err(tra.compute_path_length(trajs, robot).cpu().numpy(), gm["path_length"]) < 2e-6
    assert rel_err(tra.compute_smoothness(trajs, robot).cpu().numpy(), gm["smoothness_fd"]) < 2e-6
    assert rel_err(tra.compute_smoothness(full, robot).cpu().numpy(), gm["smoothness_vel"]) < 2e-6
    assert rel_err(tra.compute_smoothness(None, robot, trajs_vel=robot.get_velocity(full)).cpu().numpy(), gm["smoothness_vel"]) < 2e-6
    assert tra.compute_path_length(trajs[:0], robot).shape == (0,)


def test_grid_map_sdf_query(oracle_lib):
    """GridMapSDF(X): nearest-lower-cell value, gradient = the stored gradient of that cell (grid_map_sdf.py:84-114)."""
    from torch_robotics_amd.costmodel import CostModelSpec, grid_object
    env = tra.EnvSpheres3D(tensor_args=TA, precompute_sdf_obj_fixed=True, sdf_cell_size=0.1)
    grid = env.grid_map_sdf_obj_fixed
    gen = torch.Generator(device=DEV).manual_seed(9)
    X = ((torch.rand(500, 3, device=DEV, generator=gen) - 0.5) * 2.4).requires_grad_(True)     # some points outside the limits
    sdf = grid(X)
    assert sdf.shape == (500,)
    sdf.sum().backward()
    spec = CostModelSpec(n_links_in=1, objects=[grid_object()])
    d = grid.grid_dict()
    spec.grid = dict(dims=d["dims"], lim_min=d["lim_min"], map_dim=d["map_dim"], sdf=d["sdf"].cpu().numpy(), grad=d["grad"].cpu().numpy())
    from helpers import model
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), spec)
    rs, rg = o.sdf_points(X.detach().cpu().numpy(), "f32")
    np.testing.assert_array_equal(sdf.detach().cpu().numpy(), rs[:, 0])
    np.testing.assert_array_equal(X.grad.cpu().numpy(), rg[:, 0, :])
    assert grid(X.detach().reshape(20, 25, 3)).shape == (20, 25)


@pytest.mark.parametrize("grid", [False, True])
def test_env_compute_sdf(oracle_lib, grid):
    """EnvBase.compute_sdf (env_base.py:140-169): min over fixed objects (or their grid) and extra objects, with gradient."""
    from torch_robotics_amd.costmodel import CostModelSpec
    from torch_robotics_amd.environments import objects_to_spec_parts
    from helpers import model
    kw = dict(precompute_sdf_obj_fixed=True, sdf_cell_size=0.1) if grid else {}
    env = tra.EnvSpheres3DExtraObjects(tensor_args=TA, **kw)
    gen = torch.Generator(device=DEV).manual_seed(10)
    X = ((torch.rand(7, 60, 3, device=DEV, generator=gen) - 0.5) * 2.0).requires_grad_(True)
    sdf = env.compute_sdf(X)
    assert sdf.shape == (7, 60)
    sdf.sum().backward()
    spec = CostModelSpec(n_links_in=1)
    spec.objects, g = objects_to_spec_parts(env.get_df_obj_list())
    if g is not None:
        spec.grid = dict(dims=g["dims"], lim_min=g["lim_min"], map_dim=g["map_dim"], sdf=g["sdf"].cpu().numpy(), grad=g["grad"].cpu().numpy())
    o = oracle_lib.Oracle(model("panda_arm_no_gripper"), spec)
    rs, rg = o.sdf_points(X.detach().cpu().numpy().reshape(-1, 3), "f64")
    k = rs.argmin(axis=1)
    ref, refg = rs[np.arange(len(k)), k], rg[np.arange(len(k)), k]
    assert np.abs(sdf.detach().cpu().numpy().reshape(-1) - ref).max() < 2e-6
    gap = np.sort(rs, axis=1)
    clear = (gap[:, 1] - gap[:, 0] > 1e-5) if rs.shape[1] > 1 else np.ones(len(k), bool)     # away from arg-min ties between objects
    assert np.abs(X.grad.cpu().numpy().reshape(-1, 3)[clear] - refg[clear]).max() < (1e-5 if not grid else 1e-6)
    assert env.compute_sdf(X.detach(), reshape_shape=(420,)).shape == (420,)


def test_random_collision_free_configurations():
    """PlanningTask.random_coll_free_q / sample_q (tasks.py:97-129): every returned configuration is inside the limits
    and collision free according to the same boolean fields; shapes follow the reference (squeezed)."""
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    q = task.random_coll_free_q(n_samples=50, max_samples=200)
    assert q.shape == (50, 7) and q.device.type == "cuda"
    assert not task.compute_collision(q).any()
    assert bool(((q >= robot.q_min.to(q.device)) & (q <= robot.q_max.to(q.device))).all())
    assert task.random_coll_free_q(n_samples=1).shape == (7,)
    assert task.sample_q(n_samples=3).shape == (3, 7)
    assert task.sample_q(without_collision=False, n_samples=4).shape == (4, 7)


def test_point_mass_robot_like_the_reference():
    """RobotPointMass3D + PlanningTask (identity kinematics, one collision point): cost, gradient, booleans vs the reference."""
    g = gold("pointmass3d")
    robot = tra.RobotPointMass3D(tensor_args=TA)
    np.testing.assert_array_equal(robot.q_limits.cpu().numpy(), g["q_limits"])
    np.testing.assert_array_equal(robot.link_margins_for_object_collision_checking_tensor.numpy(), g["margins"])
    assert (robot.df_collision_self is not None) == bool(g["has_self"])
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=float(g["cutoff"]), tensor_args=TA)
    q = dev(g["q"]).requires_grad_(True)
    np.testing.assert_array_equal(robot.fk_map_collision(q.detach()).cpu().numpy(), g["fk"])
    cost = task.compute_collision_cost(q)
    assert cost.shape == (6, 16) and rel_err(cost.detach().cpu().numpy(), g["cost"]) < TOL_C
    cost.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq"])
    np.testing.assert_array_equal(task.compute_collision(q.detach()).cpu().numpy(), g["coll"])
    np.testing.assert_array_equal(task.compute_collision(q.detach(), margin=0.0).cpu().numpy(), g["coll0"])
    free = task.random_coll_free_q(n_samples=5)
    assert free.shape == (5, 3) and not task.compute_collision(free).any()


def test_ops_run_on_torchs_current_stream():
    """The wrappers launch on torch's CURRENT stream (the raw handle torch's own launchers use): work enqueued on a side stream
    -- a long matmul, then the copy that produces q -- is ordered before the op there, and the result is ordered before
    whatever follows on that stream.  On the default stream the op would read q before the copy had run."""
    from torch_robotics_amd import ops
    robot = tra.RobotPanda(tensor_args=TA)
    h = ops.ModelHandle(robot.diff_panda._kin)
    q_new = robot.random_q(4096)
    expect = ops.fk_positions(h, q_new).clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(DEV)
    q = torch.zeros_like(q_new)
    a = torch.randn(4096, 4096, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(8):
            a = a @ a * 1e-3                      # several milliseconds of work ahead of the copy
        q.copy_(q_new)
        pos = ops.fk_positions(h, q)
        got = pos.clone()
    side.synchronize()
    assert torch.equal(got, expect)


def _q_to_rotation_matrix_torch(q):
    """quaternion.py:102-120 as the reference writes it (plain torch ops; the floating-point reference of this check)."""
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    dc = 2.0 / (q ** 2).sum(-1)
    return torch.stack([1 - dc * (qy * qy + qz * qz), dc * (qx * qy - qz * qw), dc * (qx * qz + qy * qw),
                        dc * (qx * qy + qz * qw), 1 - dc * (qx * qx + qz * qz), dc * (qy * qz - qx * qw),
                        dc * (qx * qz - qy * qw), dc * (qy * qz + qx * qw), 1 - dc * (qx * qx + qy * qy)], -1).reshape(q.shape[:-1] + (3, 3))


def test_quaternion_and_euler_gradients_flow_like_the_reference():
    """`q_to_rotation_matrix` and `Frame.get_euler` are differentiable torch expressions in the reference; the
    kernels behind them have explicit reverse modes (they used to detach silently).  Checked against torch autograd in fp64."""
    from torch_robotics_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn(257, 4, generator=gen) * torch.tensor([1.0, 0.7, 1.3, 0.9])).to(DEV)       # NOT normalised
    w = torch.randn(257, 3, 3, generator=gen).to(DEV)
    qa = q.clone().requires_grad_(True)
    R = ops.quat_to_rotmat(qa)
    (R * w).sum().backward()
    qb = q.double().requires_grad_(True)
    Rb = _q_to_rotation_matrix_torch(qb)
    (Rb * w.double()).sum().backward()
    assert np.abs(R.detach().cpu().numpy() - Rb.detach().cpu().numpy()).max() < 2e-6
    err = (qa.grad.double() - qb.grad).abs() / (qb.grad.abs() + 1e-3 * qb.grad.abs().max())
    assert float(err.max()) < 1e-4
    # Frame.set_pose routes through it: the pose's quaternion receives a gradient
    pose = torch.cat([torch.zeros(4, 3, device=DEV), q[:4]], 1).requires_grad_(True)
    f = tra.Frame(pose=pose, device=DEV)
    f.rotation.sum().backward()
    assert pose.grad is not None and float(pose.grad[:, 3:].abs().max()) > 0

    # Euler angles (frame.py:120-121) and the trace-method quaternion with its scale held constant (frame.py:112)
    Rn = _q_to_rotation_matrix_torch(q.double()).float().contiguous()
    we, wq = torch.randn(257, 3, generator=gen).to(DEV), torch.randn(257, 4, generator=gen).to(DEV)
    Ra = Rn.clone().requires_grad_(True)
    quat, eul = ops.frame_quat_euler(Ra, want_quat=True, want_euler=True)
    ((eul * we).sum() + (quat * wq).sum()).backward()
    Rd = Rn.double().requires_grad_(True)
    e_ref = torch.stack([torch.atan2(Rd[:, 2, 1], Rd[:, 2, 2]), torch.asin(-Rd[:, 2, 0]), torch.atan2(Rd[:, 1, 0], Rd[:, 0, 0])], -1)
    qv = quat.detach().double()                       # per-sample branch and scale as taken by the kernel, held constant
    t = Rd[:, 0, 0] + Rd[:, 1, 1] + Rd[:, 2, 2] + 1.0
    terms = []
    for n in range(Rd.shape[0]):
        M = Rd[n]
        if float(t[n].detach()) > 1.0:
            v = torch.stack([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1], t[n]])
            sc = 0.5 / float(t[n].detach()) ** 0.5
        else:
            i, j, k = 0, 1, 2
            Md = M.detach()
            if float(Md[1, 1]) > float(Md[0, 0]):
                i, j, k = 1, 2, 0
            if float(Md[2, 2]) > float(Md[i, i]):
                i, j, k = 2, 0, 1
            tn = M[i, i] - (M[j, j] + M[k, k]) + 1.0
            v = [None] * 4
            v[i], v[j], v[k], v[3] = tn, M[i, j] + M[j, i], M[k, i] + M[i, k], M[k, j] - M[j, k]
            v = torch.stack(v)
            sc = 0.5 / float(tn.detach()) ** 0.5
        terms.append(((v * sc) * wq[n].double()).sum())
        assert torch.allclose(v.detach() * sc, qv[n], atol=5e-6)
    (torch.stack(terms).sum() + (e_ref * we.double()).sum()).backward()
    assert np.abs(eul.detach().cpu().numpy() - e_ref.detach().cpu().numpy()).max() < 5e-6
    gerr = (Ra.grad.double() - Rd.grad).abs() / (Rd.grad.abs() + 1e-3 * Rd.grad.abs().max())
    # asin'(x) = 1 / sqrt(1 - x^2) amplifies fp32 rounding of x near |x| = 1: exclude those few samples from the tight bound
    ok = (Rn[:, 2, 0].abs() < 0.999)
    assert float(gerr[ok].max()) < 2e-4
    # get_euler of a Frame built from a differentiable pose carries the gradient
    f2 = tra.Frame(pose=torch.cat([torch.zeros(4, 3, device=DEV), q[:4]], 1).requires_grad_(True), device=DEV)
    assert all(e.requires_grad for e in f2.get_euler())


def test_empty_scene_object_field_is_zero_in_both_kernel_families():
    """a cost model with collision links but NO objects -- the generated field kernel returned -inf where the
    table-driven one returns 0; the object term is now switched off at the boundary for both."""
    from torch_robotics_amd import ops
    from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
    from helpers import model, panda_cost_spec
    g, robot = gold("cost_spheres3d"), gold("panda_robot")
    spec = panda_cost_spec(g, robot)
    spec.objects = []
    spec.validate()
    cm = ops.CostHandle(spec, DEV)
    pos = dev(robot["fk_map_collision"]).reshape(-1, 11, 3)
    outs = []
    for on in (True, False):
        cm.enable_specialized(on)
        c_obj = ops.cost_fields(cm, FIELD_OBJECTS, pos)
        c_all, g_all = ops.cost_fields(cm, FIELD_OBJECTS | FIELD_SELF | FIELD_WS, pos, want_grad=True)
        assert torch.isfinite(c_all).all() and torch.isfinite(g_all).all()
        assert float(c_obj.abs().max()) == 0.0
        outs.append((c_all.cpu().numpy(), g_all.cpu().numpy()))
    assert rel_err(outs[0][0], outs[1][0]) < TOL_C and grad_close(outs[0][1], outs[1][1])
    c_sw = ops.cost_fields(cm, FIELD_SELF | FIELD_WS, pos)
    assert rel_err(outs[0][0], c_sw.cpu().numpy()) < 1e-6
    # the fused rollout: same rule, specialised and table-driven
    m = ops.ModelHandle(model("panda_arm_no_gripper"))
    q = dev(gold("rollout_panda")["q"][:2])
    res = []
    for on in (True, False):
        m.enable_specialized(on)
        _, c, gq = ops.rollout_cost_grad(m, cm, (1, 1, 1, 0), q)
        assert torch.isfinite(c).all() and torch.isfinite(gq).all()
        res.append((c.cpu().numpy(), gq.cpu().numpy()))
    assert rel_err(res[0][0], res[1][0]) < TOL_C and grad_close(res[0][1], res[1][1])


def test_ik_buffers_are_validated():
    from torch_robotics_amd import ops
    tree = tra.DifferentiableFrankaPanda(device=DEV)
    lo, hi, _, _ = tree.get_joint_limit_array()
    lo, hi = dev(lo.astype(np.float32)), dev(hi.astype(np.float32))
    q = torch.zeros(8, 7, **TA)
    Ht = torch.eye(4, **TA)
    with pytest.raises(ValueError, match="adam_m"):
        ops.ik_steps(tree._handle, 10, Ht, lo, hi, q, torch.zeros(8, 6, **TA), torch.zeros(8, 7, **TA), 1, 2)
    with pytest.raises(ValueError, match="lower"):
        ops.ik_steps(tree._handle, 10, Ht, lo[:5], hi, q, torch.zeros(8, 7, **TA), torch.zeros(8, 7, **TA), 1, 2)
    with pytest.raises(ValueError):
        ops.ik_step(tree._handle, 10, Ht, lo, hi, q.double(), None, None, 1)
    with pytest.raises(ValueError, match="valid"):
        ops.ik_step(tree._handle, 10, Ht, lo, hi, q, None, None, 1, lr=0.0, valid=torch.zeros(7, device=DEV, dtype=torch.uint8))


def test_fields_with_interpolated_link_points_like_the_reference():
    """`interpolate_link_pos=True`, used the way the reference's constructors are (robot_base.py:57-73 lays out the margins,
    distance_fields.py:145-147 interpolates): 15 points along the Panda's 5 object-collision links."""
    g, gr = gold("cost_interp"), gold("panda_robot")
    robot = tra.RobotPanda(tensor_args=TA)
    env = tra.EnvSpheres3D(tensor_args=TA)
    K = int(g["K_obj"])
    margins = torch.tensor(robot.link_margins_for_object_collision_checking, dtype=torch.float32).repeat_interleave(K // 5)
    fld = tra.CollisionObjectDistanceField(robot, df_obj_list_fn=env.get_df_obj_list,
                                           link_idxs_for_collision_checking=robot.link_idxs_for_object_collision_checking,
                                           num_interpolated_points=K, interpolate_link_pos=True,
                                           link_margins_for_object_collision_checking_tensor=margins,
                                           cutoff_margin=float(g["cutoff"]), tensor_args=TA)
    q = dev(g["q"]).requires_grad_(True)
    pos = robot.fk_map_collision(q)
    cost = fld.compute_cost(q, pos, field_type="sdf")
    assert cost.shape == (8, 8) and rel_err(cost.detach().cpu().numpy(), g["cost_objects"]) < TOL_C
    cost.sum().backward()
    assert grad_close(q.grad.cpu().numpy(), g["gq_objects"])
    assert np.array_equal(fld.compute_cost(q.detach(), pos.detach(), field_type="occupancy").cpu().numpy(), g["coll_objects"])
    ws = tra.CollisionWorkspaceBoundariesDistanceField(
        robot, ws_bounds_min=env.limits[0], ws_bounds_max=env.limits[1],
        link_idxs_for_collision_checking=robot.link_idxs_for_object_collision_checking, num_interpolated_points=K,
        interpolate_link_pos=True, link_margins_for_object_collision_checking_tensor=margins,
        cutoff_margin=float(g["cutoff"]), tensor_args=TA)
    assert rel_err(ws.compute_cost(q.detach(), pos.detach()).cpu().numpy(), g["cost_ws"]) < TOL_C
    # a robot declared with more interpolated points than links gets the interpolating fields from PlanningTask by itself
    robot15 = tra.RobotPanda(tensor_args=TA, num_interpolated_points_for_object_collision_checking=K,
                             num_interpolated_points_for_self_collision_checking=int(g["K_self"]))
    assert robot15.link_margins_for_object_collision_checking_tensor.shape == (K,)
    task = tra.PlanningTask(env=env, robot=robot15, obstacle_cutoff_margin=float(g["cutoff"]), tensor_args=TA)
    q2 = dev(g["q"]).requires_grad_(True)
    c = task.compute_collision_cost(q2)
    assert rel_err(c.detach().cpu().numpy(), g["cost_total"]) < TOL_C
    c.sum().backward()
    assert grad_close(q2.grad.cpu().numpy(), g["gq_total"])
    coll = g["coll_self"] | g["coll_objects"] | g["coll_ws"]
    assert np.array_equal(task.compute_collision(q2.detach()).cpu().numpy(), coll)


def _expected_partition(coll_any, outside, inner=0):
    """What get_trajs_collision_and_free's bookkeeping (tasks.py:253-284) yields for given per-trajectory facts, stated directly:
    free = collision free and inside the limits; the other list = colliding ones, then the collision-free limit violators --
    except that it is ONLY the violators when nothing is free but something was collision free."""
    t = np.arange(len(coll_any))
    free = t[~coll_any & ~outside]
    viol = t[~coll_any & outside]
    coll = t[coll_any]
    if len(free) + len(viol) == 0:
        other = coll
    elif len(free) == 0:
        other = viol
    else:
        other = np.concatenate([coll, viol])
    rows = (lambda a: np.stack([a // inner, a % inner], 1)) if inner else (lambda a: a.reshape(-1, 1))
    return rows(free), rows(other)


@pytest.mark.parametrize("case", ["mixed", "all_free", "none_free", "one_free", "only_violators", "ragged_big", "batched"])
def test_device_side_trajectory_partition(case):
    """ops.traj_validate (flags -> ordered index lists -> gathers, one host read) against the bookkeeping stated in numpy."""
    from torch_robotics_amd import ops
    rng = np.random.default_rng({"mixed": 1, "all_free": 2, "none_free": 3, "one_free": 4, "only_violators": 5, "ragged_big": 6, "batched": 7}[case])
    T, H, S, W, D = {"ragged_big": (5003, 9, 10, 37, 7), "batched": (12, 6, 7, 10, 7)}.get(case, (70, 8, 7, 21, 7))
    lo, hi = np.full(D, -1.0, np.float32), np.full(D, 1.0, np.float32)
    x = rng.uniform(-0.99, 0.99, (T, H, S)).astype(np.float32)
    p_c, p_o = {"mixed": (0.4, 0.3), "all_free": (0, 0), "none_free": (1.0, 0.2), "one_free": (0.6, 0.5), "only_violators": (0.5, 1.0),
                "ragged_big": (0.3, 0.2), "batched": (0.4, 0.3)}[case]
    coll_any, outside = rng.random(T) < p_c, rng.random(T) < p_o
    if case == "one_free":
        coll_any[:], outside[:] = True, False
        coll_any[41] = False
        outside[3] = True                       # a colliding trajectory outside the limits stays in the colliding group only
    wp = np.zeros((T, W), np.uint8)
    for t in np.flatnonzero(coll_any):
        wp[t, rng.integers(0, W, rng.integers(1, 4))] = 1
    for t in np.flatnonzero(outside):
        h, d = rng.integers(0, H), rng.integers(0, D)
        x[t, h, d] = [1.5, -1.5, np.nan][rng.integers(0, 3)]
    if S > D:
        x[:, :, D:] = 7.0                       # velocities beyond the limits' range are not positions: ignored
    inner = 4 if case == "batched" else 0
    part = ops.traj_validate(dev(wp).bool(), dev(x), D, dev(lo), dev(hi), inner=inner)
    nf, nc, no = part.counts()
    free_e, other_e = _expected_partition(coll_any, outside, inner)
    assert (nf, nc, no) == (int((~coll_any & ~outside).sum()), int(coll_any.sum()), int((~coll_any & outside).sum()))
    np.testing.assert_array_equal(part.flags.cpu().numpy(), coll_any.astype(np.uint8) | (outside.astype(np.uint8) << 1))
    np.testing.assert_array_equal(part.idx[:nf].cpu().numpy(), free_e)
    full = np.concatenate([np.flatnonzero(coll_any), np.flatnonzero(~coll_any & outside)])
    got = part.idx[nf:nf + nc + no].cpu().numpy()
    np.testing.assert_array_equal(got[:, 0] * inner + got[:, 1] if inner else got[:, 0], full)
    np.testing.assert_array_equal(part.gathered[:nf].cpu().numpy(), x[~coll_any & ~outside])
    np.testing.assert_array_equal(part.gathered[nf:nf + nc + no].cpu().numpy(), x[full])
    assert nf + nc + no == T

    # through PlanningTask: same lists, the reference's shapes (a single free trajectory gives a 1-D index row)
    class _Task(tra.PlanningTask):
        def _waypoint_collisions(self, flat, num_interpolation):
            return dev(wp).bool()
    robot = tra.RobotPanda(tensor_args=TA)
    robot.q_min, robot.q_max = dev(lo), dev(hi)
    task = _Task(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, tensor_args=TA)
    trajs = dev(x).reshape((T // inner, inner, H, S)) if inner else dev(x)
    tc, ci, tf, fi, w_out = task.get_trajs_collision_and_free(trajs, return_indices=True)
    assert w_out.shape == tuple(trajs.shape[:-2]) + (W,)
    np.testing.assert_array_equal(ci.cpu().numpy(), other_e)
    if nf == 1 and not inner:
        assert fi.shape == (1,) and int(fi[0]) == int(free_e[0, 0])
    else:
        np.testing.assert_array_equal(fi.cpu().numpy(), free_e)
    flat_rows = (lambda a: a[:, 0] * inner + a[:, 1]) if inner else (lambda a: a[:, 0])
    assert (tf is None) == (nf == 0) and (tc is None) == (len(other_e) == 0)
    if tf is not None:
        np.testing.assert_array_equal(tf.cpu().numpy(), x[flat_rows(free_e)])
    if tc is not None:
        np.testing.assert_array_equal(tc.cpu().numpy(), x[flat_rows(other_e)])


def test_fused_via_point_collision_equals_two_step():
    """trk_rollout_collision_via (interpolation inside the FK + boolean-field kernel) == trk_interpolate_via_points followed by
    trk_rollout_collision, bit for bit, incl. a state with velocities, ragged sizes and a trajectory count that splits wavefronts."""
    from torch_robotics_amd import ops
    from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    model, cm = task._fused_handles(DEV)
    fields = FIELD_OBJECTS | FIELD_WS | FIELD_SELF
    gen = torch.Generator(device=DEV).manual_seed(9)
    for T, H, S, n in ((12, 16, 7, 5), (7, 2, 7, 1), (33, 64, 14, 5), (257, 5, 9, 3), (4, 3, 7, 70)):
        x = torch.zeros(T, H, S, device=DEV)
        x[..., :7] = robot.random_q(T * H, generator=gen).reshape(T, H, 7)
        x[..., 7:] = 3.0
        for margin in (0.0, None):
            fused = ops.rollout_collision_via(model, cm, fields, x, n, margin=margin)
            assert fused is not None and fused.shape == (T, (H - 1) * n)
            two = ops.rollout_collision(model, cm, fields, ops.interpolate_traj_via_points(x, n)[..., :7].contiguous(), margin=margin)
            assert torch.equal(fused, two), (T, H, S, n, margin)
        assert 0 < int(fused.sum()) < fused.numel() or T < 8
    model.enable_specialized(False)             # no generated kernel: the call declines and the task falls back to two steps
    assert ops.rollout_collision_via(model, cm, fields, x, 3, margin=0.0) is None
    model.enable_specialized(True)


def test_empty_and_degenerate_inputs_of_the_round4_ops():
    """Zero-sized batches and the smallest legal shapes of the entry points added in round 4: the fused rollout + GP prior, the fp16
    rollout with a gradient scale, the GP prior alone, the Gauss-Newton IK and the packed sums."""
    from torch_robotics_amd import ops
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    model, cm = task._fused_handles(DEV)
    w = (0.0, 1.0, 0.0, 1.0)
    for dt_ in (torch.float32, torch.float16):
        e = torch.zeros(0, 16, 7, device=DEV, dtype=dt_)
        pos, cost, gq, gqd = ops.rollout_gp_cost_grad(model, cm, w, e, e.clone(), 0.1, 0.3, 1.0, grad_scale=0.5 if dt_ == torch.float16 else 1.0)
        assert pos.shape == (0, 16, 11, 3) and cost.shape == (0, 16) and gq.shape == (0, 16, 7) and gqd.shape == (0, 16, 7)
        pos, cost, gq = ops.rollout_cost_grad(model, cm, w, e, grad_scale=0.25 if dt_ == torch.float16 else 1.0)
        assert pos.shape == (0, 16, 11, 3) and cost.shape == (0, 16) and gq.dtype == dt_
        c, g, gd = ops.gp_prior_cost_grad(e, e.clone(), 0.1, 0.3)
        assert c.shape == (0,) and g.shape == e.shape and gd.shape == e.shape
    # one trajectory of one time step: no GP factor at all -> the fused call equals the plain rollout, gqd is zero
    q1 = robot.random_q(1).reshape(1, 1, 7).contiguous()
    pos, cost, gq, gqd = ops.rollout_gp_cost_grad(model, cm, w, q1, torch.ones_like(q1), 0.1, 0.3, 1.0)
    pos0, cost0, gq0 = ops.rollout_cost_grad(model, cm, w, q1)
    assert torch.equal(cost, cost0) and torch.equal(gq, gq0) and torch.equal(pos, pos0) and float(gqd.abs().max()) == 0.0
    # Gauss-Newton IK: an empty batch is a no-op, one problem converges
    lo, hi = robot.q_min.to(DEV).contiguous(), robot.q_max.to(DEV).contiguous()
    H_t = torch.eye(4, device=DEV); H_t[:3, 3] = torch.tensor([0.4, 0.1, 0.5])
    ops.ik_gn_steps(model, model.n_links - 1, H_t, lo, hi, torch.zeros(0, 7, device=DEV), 4)
    qs = robot.random_q(1).contiguous()
    err = torch.zeros(1, device=DEV)
    ops.ik_gn_steps(model, model.n_links - 1, H_t, lo, hi, qs, 64, err=err)
    ops.ik_gn_steps(model, model.n_links - 1, H_t, lo, hi, qs, 1, err=err)
    assert torch.isfinite(qs).all() and (qs >= lo - 1e-6).all() and (qs <= hi + 1e-6).all() and torch.isfinite(err).all()
    # packed sums of a one-sample plan
    plan = ops.RolloutPlan(model, cm, w, q1)
    sums = torch.zeros(ops.n_blocks(1), device=DEV)
    plan.launch(sums.data_ptr())
    pk = ops.PackedSums(plan, sums)
    out = torch.empty(pk.size, device=DEV)
    pk.pack(out)
    torch.cuda.synchronize()
    assert out[0].item() == pytest.approx(plan.cost.sum().item(), rel=1e-6) and torch.allclose(out[2:], plan.gq.reshape(-1), rtol=1e-6, atol=1e-7)


def test_empty_and_degenerate_inputs_of_the_round3_ops():
    """Zero-sized batches and the smallest legal shapes of the ops added in round 3 (the reference's functions accept them)."""
    from torch_robotics_amd import ops
    from torch_robotics_amd.fields import interpolate_points_v1
    from torch_robotics_amd._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    model, cm = task._fused_handles(DEV)
    fields = FIELD_OBJECTS | FIELD_WS | FIELD_SELF
    # no trajectories at all
    empty = torch.zeros(0, 16, 7, device=DEV)
    wp = ops.rollout_collision_via(model, cm, fields, empty, 5, margin=0.0)
    assert wp is not None and wp.shape == (0, 75)
    part = ops.traj_validate(wp, empty, 7, robot.q_min.to(DEV).contiguous(), robot.q_max.to(DEV).contiguous())
    assert part.counts() == (0, 0, 0) and part.idx.shape == (0, 1)
    tc, ci, tf, fi, w_out = task.get_trajs_collision_and_free(empty, return_indices=True)
    assert tc is None and tf is None and ci.shape == (0, 1) and fi.shape == (0, 1) and w_out.shape == (0, 75)
    # one trajectory of two way points, one via point
    one = robot.random_q(2).reshape(1, 2, 7).contiguous()
    tc, tf = task.get_trajs_collision_and_free(one, num_interpolation=1)
    assert (tc is None) != (tf is None)
    # jtj / interpolate / scale_rows on empty batches
    assert ops.jtj(torch.zeros(0, 3, 7, device=DEV), torch.zeros(0, 3, 7, device=DEV)).shape == (0, 7, 7)
    assert interpolate_points_v1(torch.zeros(0, 5, 3, device=DEV), 9).shape == (0, 9, 3)
    assert interpolate_points_v1(torch.ones(2, 1, 3, device=DEV), 4).eq(1).all()          # one link: every point is that link
    assert ops.scale_rows(torch.zeros(0, 7, device=DEV), torch.zeros(0, device=DEV)).shape == (0, 7)
    g = torch.randn(3, 5, 7, device=DEV)
    assert torch.equal(ops.scale_rows(g, torch.full((3, 5), 2.0, device=DEV)), g * 2.0)
    assert torch.equal(ops.scale_rows(g, torch.tensor(0.5, device=DEV).expand(3, 5)), g * 0.5)   # an expanded scalar (stride 0)
    assert torch.equal(ops.scale_rows(g.half(), torch.full((3, 5), 2.0, device=DEV)), (g.half().float() * 2.0).half())
    with pytest.raises(ValueError):
        ops.scale_rows(g, torch.ones(4, device=DEV))
    # a cost model whose fields interpolate, evaluated on an empty batch
    from helpers import interp_cost_spec
    cmi = ops.CostHandle(interp_cost_spec(), DEV)
    c, gp = ops.cost_fields(cmi, fields, torch.zeros(0, 11, 3, device=DEV), want_grad=True)
    assert c.shape == (0,) and gp.shape == (0, 11, 3)
    with pytest.raises(ValueError, match="virtual_src"):
        from torch_robotics_amd.costmodel import CostModelSpec
        bad = CostModelSpec(n_links_in=11)
        bad.add_virtual_columns(np.asarray([[2, 11]]), np.asarray([[0.5, 0.5]]))      # a source that is not a real column
        bad.validate()


def test_packed_sums_one_launch_equals_three():
    """trk_pack_sums: [sum cost | sum_b cost(b, h) | sum_b gq(b, h, d)] of an evaluation in ONE launch == the three separate
    reductions it replaces (deterministic total of the block sums, torch column sums), bit-reproducible, ticket returned to 0."""
    from torch_robotics_amd import ops
    robot = tra.RobotPanda(tensor_args=TA)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    for B, H in ((4096, 64), (37, 64), (5, 16), (1, 64)):
        q = robot.random_q(B * H).reshape(B, H, 7).contiguous()
        plan = task.rollout_plan(q, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0, want_pos=False)
        sums = torch.zeros(ops.n_blocks(B * H), **TA)
        plan.launch(sums.data_ptr())
        pk = ops.PackedSums(plan, sums)
        out = torch.empty(pk.size, **TA)
        pk.pack(out)
        assert pk.size == 1 + H + H * 7
        assert torch.equal(out[0:1], ops.reduce_sum(sums))                                      # the same association order
        ref_c, ref_g = plan.cost.double().sum(0), plan.gq.double().sum(0).reshape(-1)
        assert rel_err(out[1:1 + H].cpu().numpy(), ref_c.cpu().numpy()) < 2e-6
        assert np.abs(out[1 + H:].cpu().numpy() - ref_g.cpu().numpy()).max() <= 2e-6 * max(1.0, float(plan.gq.abs().sum(0).max()))
        out2 = torch.empty_like(out)
        for _ in range(3):
            pk.pack(out2)
            assert torch.equal(out, out2)                                                       # bit-reproducible
    # an fp16, loss-scaled gradient (config 5's exchange): widened, summed in fp32, divided by the scale once; + a per-trajectory cost
    for B, H in ((512, 64), (33, 128), (7, 6)):
        q = robot.random_q(B * H).reshape(B, H, 7).contiguous().half()
        gs = 2.0 ** -6
        model, cm = task._fused_handles(DEV)
        plan = ops.RolloutPlan(model, cm, (1.0, 1.0, 1.0, 0.0), q, want_pos=False, grad_scale=gs)
        sums = torch.zeros(ops.n_blocks(B * H), **TA)
        plan.launch(sums.data_ptr())
        tc = torch.rand(B, **TA)
        pk = ops.PackedSums(plan, sums, tc)
        out = torch.empty(pk.size, **TA)
        pk.pack(out)
        assert plan.gq.dtype == torch.float16
        assert rel_err(out[0:1].cpu().numpy(), (plan.cost.double().sum() + tc.double().sum()).reshape(1).cpu().numpy()) < 1e-5
        assert rel_err(out[1:1 + H].cpu().numpy(), plan.cost.double().sum(0).cpu().numpy()) < 2e-6
        ref_g = (plan.gq.double().sum(0) / gs).reshape(-1)
        assert np.abs(out[1 + H:].cpu().numpy() - ref_g.cpu().numpy()).max() <= 2e-6 * max(1.0, float(plan.gq.float().abs().sum(0).max()) / gs)


def test_interpolated_points_with_a_grasped_object(oracle_lib):
    """interpolate_link_pos on a robot that holds an object: the interpolated points replace the selected links, the grasped
    object's points follow un-interpolated (distance_fields.py:139-152), pair rows index both (robot_base.py:103-130).  No
    reference golden exists for this combination (the reference raises on the grasped-object gather): fp64 oracle."""
    from torch_robotics_amd import ops
    go = tra.GraspedObjectPandaBox(tensor_args=TA)
    robot = tra.RobotPanda(grasped_object=go, tensor_args=TA, num_interpolated_points_for_object_collision_checking=10,
                           num_interpolated_points_for_self_collision_checking=16)
    G = go.n_base_points_for_collision
    assert robot.link_margins_for_object_collision_checking_tensor.shape == (10 + G,)
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    assert task.interpolate_link_pos and robot.df_collision_self.interpolate_link_pos
    spec = task.build_cost_spec()
    P = len(robot.collision_point_set()[0])
    assert spec.n_links_in == P and spec.n_columns == P + 10 + 16
    assert list(spec.obj_link_idx[:10]) == list(range(P, P + 10)) and list(spec.obj_link_idx[10:]) == list(range(P - G, P))
    q = robot.random_q(300, generator=torch.Generator(device=DEV).manual_seed(3)).reshape(300, 1, 7)
    pos, cost, gq = task.rollout_cost_grad(q, w_self=1.0, w_obj=1.0, w_ws=1.0, w_ee=0.0)
    pl, po = robot.collision_point_set()
    o = oracle_lib.Oracle(robot.diff_panda._kin, spec)
    p64, c64, g64 = o.rollout_points(pl, po, q.reshape(-1, 7).cpu().numpy().astype(np.float64), (1, 1, 1, 0), "f64")
    assert np.abs(pos.reshape(-1, P, 3).cpu().numpy() - p64).max() < TOL_H
    assert rel_err(cost.reshape(-1).cpu().numpy(), c64) < TOL_C and grad_close(gq.reshape(-1, 7).cpu().numpy(), g64)
    # autograd through the task API and the boolean path agree with the fused call
    qg = q.clone().requires_grad_(True)
    c2 = task.compute_collision_cost(qg)
    c2.sum().backward()
    assert torch.allclose(c2, cost, rtol=1e-6, atol=1e-6) and torch.allclose(qg.grad, gq, rtol=1e-5, atol=1e-6)
    coll = task.compute_collision(q)
    ref = o.collision_fields(7, p64, None, "f64").reshape(300, 1)
    assert (coll.cpu().numpy() != ref).sum() <= 1


def test_inverse_kinematics_gn_api():
    """DifferentiableTree.inverse_kinematics_gn (extension: the reference's IK contract on trk_ik_gn_steps): (q, idx_valid) like
    `inverse_kinematics`, every returned valid configuration reaches the target inside the shrunk limits; a link no ahead-of-time
    unit tracks gets its unit compiled on first use."""
    torch.manual_seed(3)
    tree = tra.DifferentiableFrankaPanda(gripper=False, device=DEV)
    lo, hi, _, _ = tree.get_joint_limit_array()
    q_star = torch.as_tensor((lo + hi) / 2 + 0.3 * (hi - lo) * (np.random.default_rng(2).random(7) - 0.5), **TA).reshape(1, 7)
    for link_name in ("ee_link", "panda_link7"):
        H = tree.compute_forward_kinematics_all_links(q_star, link_list=[link_name])[0, 0]
        q, idx = tree.inverse_kinematics_gn(H, link_name=link_name, batch_size=256, max_iters=60, se3_eps=5e-2, check_every=15)
        assert q.shape == (256, 7) and idx.ndim == 1 and idx.numel() >= 64
        Hq = tree.compute_forward_kinematics_all_links(q[idx], link_list=[link_name])[:, 0]
        err = tra.SE3_distance(Hq, H)
        eps = np.pi / 100
        assert float(err.max()) < 5e-2 * 1.001
        assert (q[idx] >= torch.as_tensor(lo + eps, **TA) - 1e-6).all() and (q[idx] <= torch.as_tensor(hi - eps, **TA) + 1e-6).all()


def test_gauss_newton_ik_example_converges():
    """examples/gauss_newton_ik.py: the one-launch Gauss-Newton kernel (trk_ik_gn_steps) and the two-launch form (trk_fk_jacobian +
    trk_jtj + torch ops) reach a reachable pose from random starts."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("gn_ik", Path(__file__).resolve().parent.parent / "examples" / "gauss_newton_ik.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fracs = []
    for kw in (dict(), dict(two_launch=True), dict(two_launch=True, mfma=True)):      # trk_ik_gn_steps; the two-launch form, both jtj kernels
        q, err = mod.main(batch_size=512, max_iters=40, verbose=False, **kw)
        assert q.shape == (512, 7) and torch.isfinite(err).all()
        # from uniformly random starts about half of the problems reach the pose (the projection onto the Panda's tight joint
        # limits traps the rest); the Adam loop of the reference needs hundreds of iterations for the same.  The share depends on the
        # drawn target: 0.19 - 0.59 over twenty seeds (0.53 at the example's own), the same for both forms -- a smoke bound here, the
        # iteration itself is pinned by tests/golden/ik_gn_panda.npz
        frac = float((err < 1e-3).float().mean())
        assert frac > 0.15, frac
        fracs.append(frac)
    assert max(fracs) - min(fracs) < 0.02, fracs                   # one launch, two launches, MFMA J^T J: the same iteration


def test_reduce_sum_long_vectors_are_deterministic_and_accurate():
    """trk_reduce_sum beyond 65 536 elements takes the wide single-workgroup kernel: same bits run to run, ragged / unaligned lengths,
    fp64-accurate to fp32 summation error."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    for n in (65537, 262144, 262147, 1 << 20):
        x = torch.randn(n + 1, device=DEV, generator=gen)
        for v in (x[:n], x[1:n + 1]):                        # 16-byte aligned and not
            a, b = tra.ops.reduce_sum(v), tra.ops.reduce_sum(v)
            assert torch.equal(a, b)
            ref = v.double().sum().item()
            assert abs(a.item() - ref) <= 2e-6 * v.abs().double().sum().item()


def test_no_bundled_robot_takes_the_table_driven_kernels_without_a_compiler(tmp_path):
    """A box without hipcc and with an EMPTY run-time cache: every bundled URDF is served by an ahead-of-time unit of libtrk.so --
    the FK family by the model alone, the fused rollout on the robot's shipped collision template --, the Panda task needs no
    compile; a collision model that has no unit is compiled IN-PROCESS with hipRTC (the unit's device half as a code object,
    libtrk.so's generic launchers as its host half); and when that is not available either it is an ERROR under PlanningTask
    (not a silent 10 x slowdown) unless TRK_ALLOW_TABLE_DRIVEN=1."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.join(os.environ["TRK_ROOT"], "tests"))
import torch_robotics_amd as tra
from torch_robotics_amd import codegen, jit, ops
from torch_robotics_amd.costmodel import CostModelSpec
assert not jit.hipcc_available() and not any(jit.JIT_DIR.glob("*.so"))
TA = dict(device="cuda:0", dtype=torch.float32)
env = tra.EnvSpheres3D(tensor_args=TA)
for ident in codegen.SPEC_ROBOTS:
    kin, tmpl = codegen.template_for(ident)
    spec = CostModelSpec(n_links_in=kin.n_links)
    spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
    spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.1, np.float32)
    spec.objects = [o.as_object() for o in env.obj_fixed_list]
    spec.ee_link = tmpl.ee_link
    spec.ee_target = np.eye(4, dtype=np.float32)
    if tmpl.ee2_link >= 0:
        spec.ee2_link, spec.ee2_target = tmpl.ee2_link, np.eye(4, dtype=np.float32)
    h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, "cuda:0")
    assert h.specialized, ident
    assert ops.rollout_is_specialized(h, cm, (0, 1, 1, 1)), ident
    q = torch.zeros(2, 64, kin.n_dofs, **TA)
    pos, cost, gq = ops.rollout_cost_grad(h, cm, (0, 1, 1, 1), q)
    assert torch.isfinite(cost).all() and torch.isfinite(gq).all()
    print("ok", ident)
robot = tra.RobotPanda(tensor_args=TA)
task = tra.PlanningTask(env=env, robot=robot, tensor_args=TA)
q = robot.random_q(128)
task.compute_collision_cost(q)                       # the shipped Panda template: no compile needed
model, cm = task._fused_handles(torch.device("cuda:0"))
assert ops.rollout_is_specialized(model, cm, (1, 1, 1, 0))
# a robot declared with another collision model has no ahead-of-time unit.  (1) neither compiler: an error ...
robot2 = tra.RobotPanda(tensor_args=TA, num_interpolated_points_for_object_collision_checking=9)
os.environ["TRK_NO_HIPRTC"] = "1"
task2 = tra.PlanningTask(env=env, robot=robot2, tensor_args=TA)
try:
    task2.compute_collision_cost(q)
    raise SystemExit("expected a RuntimeError")
except RuntimeError as e:
    assert "TRK_ALLOW_TABLE_DRIVEN" in str(e)
for call in (lambda: task2.compute_collision_cost(q), lambda: task2.rollout_plan(q.reshape(2, 64, 7))):   # a retried call raises too
    try:
        call()
        raise SystemExit("expected a RuntimeError on the second call as well (no silent table-driven fall-back)")
    except RuntimeError as e:
        assert "TRK_ALLOW_TABLE_DRIVEN" in str(e)
# ... unless the caller accepts the table-driven kernels
os.environ["TRK_ALLOW_TABLE_DRIVEN"] = "1"
task3 = tra.PlanningTask(env=env, robot=robot2, tensor_args=TA)
with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter("always")
    c3 = task3.compute_collision_cost(q)
assert torch.isfinite(c3).all() and any("table-driven" in str(w.message) for w in rec)
m3, cm3 = task3._fused_handles(torch.device("cuda:0"))
assert not ops.rollout_is_specialized(m3, cm3, (1, 1, 1, 0))
# (2) hipRTC: the same collision model compiled in-process -- same values as the table-driven kernels, now from a generated unit
del os.environ["TRK_NO_HIPRTC"], os.environ["TRK_ALLOW_TABLE_DRIVEN"]
task4 = tra.PlanningTask(env=env, robot=robot2, tensor_args=TA)
c4 = task4.compute_collision_cost(q)
m4, cm4 = task4._fused_handles(torch.device("cuda:0"))
assert ops.rollout_is_specialized(m4, cm4, (1, 1, 1, 0)) and any(jit.JIT_DIR.glob("*.hsaco")) and not any(jit.JIT_DIR.glob("*.so"))
assert float((c4 - c3).abs().max()) <= 1e-5 * float(c3.abs().max())
print("rtc ok")
# (3) a robot NO unit exists for (a modified iiwa7: another model hash): every kernel family of its hipRTC unit, through libtrk.so's
# generic launchers, against the table-driven kernels
from torch_robotics_amd.kinematics import URDF_DIR
from torch_robotics_amd.kinmodel import KinModel
text = (URDF_DIR / "iiwa7.urdf").read_text().replace('xyz="0 0 0.15"', 'xyz="0 0 0.1625"', 1)
path = os.path.join(os.environ["TRK_JIT_DIR"], "iiwa7_mod.urdf")
open(path, "w").write(text)
kin = KinModel.from_urdf(path)
assert codegen.model_hash(kin) not in {mh for _i, mh, _t in codegen.aot_units()}
tmpl = codegen.default_template(kin)
spec = CostModelSpec(n_links_in=kin.n_links)
spec.obj_link_idx = np.asarray(tmpl.obj_links, np.int32)
spec.obj_link_margin = np.full(len(tmpl.obj_links), 0.1, np.float32)
spec.objects = [o.as_object() for o in env.obj_fixed_list]
spec.ws_min, spec.ws_max = np.float32([-1, -1, -1]), np.float32([1, 1, 1])
spec.ee_link = tmpl.ee_link
Ht = np.eye(4, dtype=np.float32); Ht[:3, 3] = (0.3, 0.1, 0.6); spec.ee_target = Ht
h, cm = ops.ModelHandle(kin), ops.CostHandle(spec, "cuda:0")
assert not h.specialized
jit.specialize(kin, tmpl.obj_links, (), ee_link=tmpl.ee_link)
assert h.specialized and ops.rollout_is_specialized(h, cm, (0, 1, 1, 1))
gen = torch.Generator(device="cuda:0").manual_seed(5)
D, L = kin.n_dofs, kin.n_links
qq = (torch.rand(3, 64, D, generator=gen, **TA) - 0.5) * 3.0
q2 = qq.reshape(-1, D)
def both(fn):
    h.enable_specialized(True); a = fn()
    h.enable_specialized(False); b = fn()
    h.enable_specialized(True)
    return a, b
def close(a, b, tol):
    a, b = (a.float(), b.float())
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), float((a - b).abs().max())
a, b = both(lambda: ops.rollout_cost_grad(h, cm, (0, 1, 1, 1), qq))
close(a[0], b[0], 3e-6); close(a[1], b[1], 1e-5); close(a[2], b[2], 1e-4)
a, b = both(lambda: ops.rollout_cost_grad(h, cm, (0, 1, 1, 1), qq.half()))
close(a[0], b[0], 2e-3); close(a[1], b[1], 1e-5); close(a[2], b[2], 2e-3)
a, b = both(lambda: ops.rollout_collision(h, cm, 6, qq))
assert int((a != b).sum()) <= 1
a, b = both(lambda: ops.fk_forward(h, q2)); close(a, b, 3e-6)
a, b = both(lambda: ops.fk_forward(h, q2, [tmpl.ee_link])); close(a, b, 3e-6)
gH = torch.randn(q2.shape[0], L, 4, 4, generator=gen, **TA)
a, b = both(lambda: ops.fk_backward(h, q2, gH)); close(a, b, 1e-4)
a, b = both(lambda: ops.fk_positions(h, q2)); close(a, b, 3e-6)
gp = torch.randn(q2.shape[0], L, 3, generator=gen, **TA)
a, b = both(lambda: ops.fk_positions_backward(h, q2, gp)); close(a, b, 1e-4)
a, b = both(lambda: ops.fk_jacobian(h, q2, None, tmpl.ee_link))
for x, y in zip(a, b): close(x, y, 1e-5)
pos = ops.fk_positions(h, q2)
a = ops.cost_fields(cm, 6, pos, want_grad=True)
cm.enable_specialized(False); b = ops.cost_fields(cm, 6, pos, want_grad=True); cm.enable_specialized(True)
close(a[0], b[0], 1e-5); close(a[1], b[1], 1e-4)
lo, hi = torch.full((D,), -2.0, **TA), torch.full((D,), 2.0, **TA)
Hq = ops.fk_forward(h, (torch.rand(1, D, generator=gen, **TA) - 0.5) * 2.0, [tmpl.ee_link]).reshape(4, 4).contiguous()
def ik():
    q_ = q2.clone(); m_, v_ = torch.zeros_like(q_), torch.zeros_like(q_)
    ops.ik_steps(h, tmpl.ee_link, Hq, lo, hi, q_, m_, v_, 1, 5)
    return q_
a, b = both(ik); close(a, b, 2e-3)
qg = q2.clone(); err = torch.empty(q2.shape[0], **TA)
ops.ik_gn_steps(h, tmpl.ee_link, Hq, lo, hi, qg, 20)
ops.ik_gn_steps(h, tmpl.ee_link, Hq, lo, hi, qg.clone(), 1, err=err)
assert torch.isfinite(qg).all() and float(err.median()) < 0.2
qd = torch.randn(3, 64, D, generator=gen, **TA) * 0.3
a, b = both(lambda: ops.rollout_gp_cost_grad(h, cm, (0, 1, 1, 1), qq, qd, 0.08, 0.3))
close(a[1], b[1], 2e-5); close(a[2], b[2], 1e-4); close(a[3], b[3], 1e-4)
print("generic launchers ok")
# (4) round 5: an ATTACHED-POINT unit through hipRTC -- a Panda holding another box (the grasp golden's point set, stretched: another
# points hash, no ahead-of-time unit): table-driven result first, then the unit compiled in-process, registered as a code object
from helpers import grasp_panda_setup
mg, pl, po, gspec = grasp_panda_setup()
po = (po * np.float32(1.25)).astype(np.float32)
hg = ops.ModelHandle(mg)
psg, cmg = ops.PointSetHandle(hg, pl, po, "cuda:0"), ops.CostHandle(gspec, "cuda:0")
assert not psg.specialized
qg3 = (torch.rand(3, 64, 7, generator=gen, **TA) - 0.5) * 4.0
ref = [t.clone() for t in ops.rollout_points_cost_grad(psg, cmg, (1, 1, 1, 0), qg3)]
ref_pos = ops.fk_points(psg, qg3.reshape(-1, 7)).clone()
wpt = torch.randn(192, len(pl), 3, generator=gen, **TA)
ref_bwd = ops.fk_points_backward(psg, qg3.reshape(-1, 7), wpt).clone()
ident = jit.specialize_points(mg, pl, po, gspec)
assert ident and any(jit.JIT_DIR.glob(f"spec_{ident}.hsaco")) and not any(jit.JIT_DIR.glob("*.so"))
psg2 = ops.PointSetHandle(hg, pl, po, "cuda:0")
assert psg2.specialized
got = ops.rollout_points_cost_grad(psg2, cmg, (1, 1, 1, 0), qg3)
close(got[0], ref[0], 3e-6); close(got[1], ref[1], 1e-5); close(got[2], ref[2], 1e-4)
close(ops.fk_points(psg2, qg3.reshape(-1, 7)), ref_pos, 3e-6)
close(ops.fk_points_backward(psg2, qg3.reshape(-1, 7), wpt), ref_bwd, 1e-4)
print("rtc points ok")
print("done")
"""
    root = str(ROOT) if "ROOT" in globals() else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRK_JIT_DIR=str(tmp_path), TRK_HIPCC=str(tmp_path / "no-such-hipcc"), TRK_ROOT=root,
               PATH=":".join(p for p in os.environ.get("PATH", "").split(":") if "rocm" not in p))
    for k in ("TRK_ALLOW_TABLE_DRIVEN", "TRK_NO_JIT", "TRK_NO_HIPRTC"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0 and "done" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    from torch_robotics_amd import codegen
    assert p.stdout.count("ok ") >= len(codegen.SPEC_ROBOTS) and "rtc ok" in p.stdout and "generic launchers ok" in p.stdout and "rtc points ok" in p.stdout
