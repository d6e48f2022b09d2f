"""CPU-only: the host-side mirror of the reference interface (model compiler, scene tables, cost specs,
error behaviour) against golden data dumped from the reference."""
import numpy as np
import pytest
import torch

import torch_robotics_amd as tra
from helpers import GOLD, ROBOTS, URDF, gold, model, objects_from_golden, panda_cost_spec
from torch_robotics_amd.kinmodel import KinModel

TA = dict(device=torch.device("cpu"), dtype=torch.float32)
TYPE_NAMES = {0: "fixed", 1: "revolute", 2: "continuous", 3: "prismatic"}


@pytest.mark.parametrize("robot", ROBOTS)
def test_model_compiler_matches_reference_build(robot):
    """A1: link/DOF order, parents, origins, fixed rotations (bit-exact), axes, types, limits."""
    g, m = gold(f"fk_{robot}"), model(robot)
    assert list(g["link_names"]) == m.link_names
    assert int(g["n_dofs"]) == m.n_dofs
    np.testing.assert_array_equal(g["controlled"], m.controlled)
    np.testing.assert_array_equal(g["parent"], m.parent)
    np.testing.assert_array_equal(g["trans"], m.trans)
    np.testing.assert_array_equal(g["R_fixed"], m.R_fixed)
    np.testing.assert_array_equal(g["axis"], m.axis)
    assert [TYPE_NAMES[int(t)] for t in m.joint_type] == list(g["joint_type"])
    np.testing.assert_array_equal(g["has_limits"].astype(np.int32), m.has_limits)
    np.testing.assert_array_equal(g["lower"].astype(np.float32), m.lower)
    np.testing.assert_array_equal(g["upper"].astype(np.float32), m.upper)
    # traversal tables are self-consistent
    assert sorted(m.order) == list(range(m.n_links)) and m.order[0] == 0
    pos = np.argsort(m.order)
    for p, i in enumerate(m.order):
        if p:
            assert pos[m.parent[i]] < p
            if m.parent_slot[p] < 0:
                assert m.order[p - 1] == m.parent[i]
        assert p < m.subtree_end[p] <= m.n_links


def test_reference_originals_give_the_same_model():
    """The kinematics-only URDFs shipped here compile to the same tables as the reference's own files
    (only checked where /root/reference exists, i.e. in the development container)."""
    from pathlib import Path
    ref = Path("/root/reference/torch_robotics/data/urdf/robots")
    if not ref.exists():
        pytest.skip("reference not present")
    pairs = {"panda_arm_no_gripper": "franka_description/robots/panda_arm_no_gripper.urdf",
             "ur10": "ur10/urdf/ur10.urdf", "allegro_hand": "allegro_hand/allegro_hand.urdf",
             "hab_stretch": "habitat_stretch/urdf/hab_stretch.urdf"}
    for name, rel in pairs.items():
        a, b = model(name), KinModel.from_urdf(str(ref / rel))
        for f in ("parent", "joint_type", "dof_idx", "R_fixed", "trans", "axis", "lower", "upper", "order"):
            np.testing.assert_array_equal(getattr(a, f), getattr(b, f))


def test_urdf_error_cases(tmp_path):
    bad = tmp_path / "r.urdf"
    bad.write_text('<robot name="r"><link name="a"/><link name="b"/></robot>')
    with pytest.raises(ValueError, match="not the child of any joint"):
        KinModel.from_urdf(str(bad))
    bad.write_text('<robot name="r"><link name="a"/><link name="b"/>'
                   '<joint name="j" type="floating"><parent link="a"/><child link="b"/></joint></robot>')
    m = KinModel.from_urdf(str(bad))
    assert m.has_unsupported_joint() == "b" and m.n_dofs == 1
    with pytest.raises(NotImplementedError):
        tra.DifferentiableTree("robot.xml")
    t = tra.DifferentiableTree(str(bad), device="cpu")
    with pytest.raises(NotImplementedError):
        t.compute_forward_kinematics_all_links(torch.zeros(2, 1))


def test_tree_api_surface():
    t = tra.DifferentiableFrankaPanda(device="cpu")
    assert t._n_dofs == 7 and t._name_to_idx_map["ee_link"] == 10
    assert t.get_link_names() == [f"panda_link{i}" for i in range(9)] + ["panda_hand", "ee_link"]
    lo, hi, vlo, vhi = t.get_joint_limit_array()
    np.testing.assert_allclose(lo, [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973])
    np.testing.assert_allclose(hi, [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973])
    np.testing.assert_allclose(vhi, [2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61])
    assert lo.dtype == np.float64
    assert tra.DifferentiableTiagoDualHoloMove(device="cpu").get_link_names()[0] == \
        model("tiago_dual_holobase_minimal_holonomic").link_names[3]


ENV_CLASSES = {"spheres3d": tra.EnvSpheres3D, "table_shelf": tra.EnvTableShelf, "maze_boxes3d": tra.EnvMazeBoxes3D,
               "spheres3d_extra": tra.EnvSpheres3DExtraObjects}


@pytest.mark.parametrize("env", sorted(ENV_CLASSES))
def test_scene_tables_match_reference(env):
    """Scene data (A10): primitive centres / sizes / rounding radii / object poses equal the reference's."""
    g = gold(f"cost_{env}")
    e = ENV_CLASSES[env](tensor_args=TA)
    np.testing.assert_array_equal(e.limits_np, g["limits"])
    for tag, objs in (("fixed", e.obj_fixed_list), ("extra", e.obj_extra_list)):
        ref = objects_from_golden(g, tag)
        mine = [o.as_object() for o in (objs or [])]
        assert len(ref) == len(mine)
        for a, b in zip(ref, mine):
            np.testing.assert_array_equal(a["pos"], b["pos"])
            np.testing.assert_array_equal(a["R"], b["R"])
            assert len(a["prims"]) == len(b["prims"])
            for pa, pb in zip(a["prims"], b["prims"]):
                assert pa["type"] == pb["type"]
                np.testing.assert_array_equal(pa["center"], pb["center"])
                np.testing.assert_array_equal(pa.get("half", 0), pb.get("half", 0))
                assert np.float32(pa["radius"]) == np.float32(pb["radius"])


@pytest.mark.parametrize("env", sorted(ENV_CLASSES))
def test_planning_task_cost_spec_matches_reference(env):
    """RobotPanda + PlanningTask assemble the same collision model as the reference (A6, A9, A14)."""
    g, robot_g = gold(f"cost_{env}"), gold("panda_robot")
    robot = tra.RobotPanda(tensor_args=TA)
    np.testing.assert_array_equal(robot.link_idxs_for_object_collision_checking, robot_g["obj_link_idxs"])
    np.testing.assert_array_equal(robot.link_idxs_for_self_collision_checking, robot_g["self_link_idxs"])
    np.testing.assert_array_equal(robot.df_collision_self.idxs_links_distance_matrix, robot_g["self_pairs"])
    np.testing.assert_array_equal(robot.q_limits.numpy(), robot_g["q_limits"])
    task = tra.PlanningTask(env=ENV_CLASSES[env](tensor_args=TA), robot=robot, obstacle_cutoff_margin=float(g["cutoff"]),
                            tensor_args=TA)
    mine, ref = task.build_cost_spec(), panda_cost_spec(g, robot_g)
    mine.validate()
    np.testing.assert_array_equal(mine.obj_link_idx, ref.obj_link_idx)
    np.testing.assert_array_equal(mine.obj_link_margin, ref.obj_link_margin)
    np.testing.assert_array_equal(mine.self_link_idx, ref.self_link_idx)
    np.testing.assert_array_equal(mine.self_pairs, ref.self_pairs)
    np.testing.assert_array_equal(mine.self_margin, ref.self_margin)
    np.testing.assert_array_equal(mine.ws_min, ref.ws_min)
    np.testing.assert_array_equal(mine.ws_max, ref.ws_max)
    assert len(mine.objects) == len(ref.objects)
    assert task.get_collision_fields() == [task.df_collision_self, task.df_collision_objects, task.df_collision_ws_boundaries]


def test_cpu_tensors_are_rejected_not_computed():
    robot = tra.RobotPanda(tensor_args=TA)
    if torch.cuda.is_available():
        pytest.skip("no-GPU behaviour")
    with pytest.raises(Exception):
        robot.fk_map_collision(torch.zeros(3, 7))


def test_trajectory_plumbing_has_no_cpu_path():
    """A17 (finite differences, path length, smoothness) runs in HIP kernels: CPU tensors are rejected, not computed."""
    from torch_robotics_amd.robots import compute_path_length, finite_difference_vector
    x = torch.as_tensor(gold("traj")["x"])
    with pytest.raises((ValueError, RuntimeError)):
        finite_difference_vector(x, dt=0.25, method="central")
    with pytest.raises((ValueError, RuntimeError)):
        compute_path_length(x, tra.RobotPanda(tensor_args=TA))
    with pytest.raises(NotImplementedError):
        from torch_robotics_amd import ops
        ops.finite_difference(x, method="nope")


def test_grasped_object_model_matches_reference():
    """RobotPanda(grasped_object=GraspedObjectPandaBox) -- SURVEY 8f-4: kinematic tables equal the reference's
    pre-generated URDF, the collision point set, margins and self-collision pair table equal RobotBase's."""
    from helpers import grasp_panda_setup
    g = gold("grasp_panda")
    go = tra.GraspedObjectPandaBox(tensor_args=TA)
    np.testing.assert_array_equal(go.base_points_for_collision.numpy(), g["base_points"])
    np.testing.assert_array_equal(go.pos, g["grasp_pos"])
    np.testing.assert_array_equal(go.ori, g["grasp_ori"])
    robot = tra.RobotPanda(grasped_object=go, tensor_args=TA)
    m_ref, pl, po, spec_ref = grasp_panda_setup()
    k = robot.diff_panda._kin
    assert k.link_names == m_ref.link_names == [str(s) for s in g["link_names"]]
    for name in ("parent", "joint_type", "dof_idx", "R_fixed", "trans", "axis", "lower", "upper", "order"):
        np.testing.assert_array_equal(getattr(k, name), getattr(m_ref, name), err_msg=name)
    mine_pl, mine_po = robot.collision_point_set()
    np.testing.assert_array_equal(mine_pl, pl)
    np.testing.assert_array_equal(mine_po, po)
    np.testing.assert_array_equal(robot.link_margins_for_object_collision_checking_tensor.numpy(), g["obj_margins"])
    np.testing.assert_array_equal(robot.df_collision_self.idxs_links_distance_matrix, g["self_pairs"])
    np.testing.assert_array_equal(robot.df_collision_self.cutoff_margin.numpy(), g["self_margins"])
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=float(g["cutoff"]),
                            tensor_args=TA)
    mine = task.build_cost_spec()
    mine.validate()
    assert mine.n_links_in == 26
    for name in ("obj_link_idx", "obj_link_margin", "self_link_idx", "self_pairs", "self_margin", "ws_min", "ws_max"):
        np.testing.assert_array_equal(getattr(mine, name), getattr(spec_ref, name), err_msg=name)


def test_link_sphere_model_tables():
    """RobotPanda(link_sphere_model="panda") -- SURVEY 8f-3: 45 link-frame spheres replace the 5 link-origin points.
    Columns are link-sorted: each link origin is followed by that link's spheres."""
    robot = tra.RobotPanda(link_sphere_model="panda", tensor_args=TA)
    pl, po = robot.collision_point_set()
    assert pl.shape == (56,) and po.shape == (56, 3)
    assert list(pl) == sorted(pl)                                   # a chain: walk order == file order
    origin_cols = [int(np.nonzero(pl == i)[0][0]) for i in range(11)]
    assert not po[origin_cols].any()
    obj = robot.link_idxs_for_object_collision_checking
    assert len(obj) == 45 and not set(obj) & set(origin_cols) and sorted(obj + origin_cols) == list(range(56))
    assert robot.link_names_for_object_collision_checking[0] == "panda_link0"
    assert pl[obj[0]] == 0 and np.allclose(po[obj[0]], [0, 0, 0.05])
    assert float(robot.link_margins_for_object_collision_checking_tensor[0]) == np.float32(0.08)
    plain = tra.RobotPanda(tensor_args=TA)
    assert robot.link_idxs_for_self_collision_checking == [origin_cols[i] for i in plain.link_idxs_for_self_collision_checking]
    task = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    spec = task.build_cost_spec()
    spec.validate()
    assert spec.n_links_in == 56 and len(spec.obj_link_idx) == 45 and len(spec.self_pairs) == 10
    both = tra.RobotPanda(link_sphere_model="panda", grasped_object=tra.GraspedObjectPandaBox(tensor_args=TA), tensor_args=TA)
    spec = tra.PlanningTask(env=tra.EnvSpheres3D(tensor_args=TA), robot=both, tensor_args=TA).build_cost_spec()
    spec.validate()
    assert spec.n_links_in == 12 + 45 + 14 and len(spec.obj_link_idx) == 59 and len(spec.self_pairs) == 66
    np.testing.assert_array_equal(spec.obj_link_idx[-14:], np.arange(57, 71))
    np.testing.assert_array_equal(spec.self_link_idx[-14:], np.arange(57, 71))


def test_scene_version_follows_object_poses():
    """Every cached device cost model is keyed by `scene_version`: identity and CURRENT pose of each object (a
    moved object must not keep evaluating the old scene)."""
    import torch
    from torch_robotics_amd.environments import EnvSpheres3DExtraObjects, scene_version
    ta = dict(device=torch.device("cpu"), dtype=torch.float32)
    env = EnvSpheres3DExtraObjects(tensor_args=ta)
    v0 = scene_version(env.get_df_obj_list())
    assert scene_version(env.get_df_obj_list()) == v0 and len(v0) == 2
    env.obj_extra_list[0].set_position_orientation(pos=(0.1, 0.0, 0.0))
    v1 = scene_version(env.get_df_obj_list())
    assert v1 != v0 and v1[0] == v0[0]
    env.obj_extra_list[0].set_position_orientation(ori=(0.0, 1.0, 0.0, 0.0))
    assert scene_version(env.get_df_obj_list()) != v1
    v2 = scene_version(env.get_df_obj_list())
    env.obj_extra_list[0].pos = (0.0, 0.0, 0.0)                 # plain attribute assignment is seen as well
    v3 = scene_version(env.get_df_obj_list())
    assert v3 != v2 and (env.obj_extra_list[0].pos == 0).all()
    # the key is integers only (it is built on every cost evaluation): (id, version) per object
    assert all(isinstance(x, int) for entry in v3 for x in entry)
    # compute_sdf lives on EnvBase only (the other scene classes have no object list to serve it)
    from torch_robotics_amd import environments as E
    assert hasattr(E.EnvBase, "compute_sdf")
    for cls in (E.PrimitiveShapeField, E.ObjectField, E.GridMapSDF):
        assert not hasattr(cls, "compute_sdf") and not hasattr(cls, "add_obj")


def test_q_width_check():
    import torch
    from torch_robotics_amd.ops import _check_buffer, _check_q_dofs
    _check_q_dofs(torch.zeros(3, 4, 7), 7, "x")
    for bad in (torch.zeros(3, 4, 14), torch.zeros(6), torch.zeros(())):
        with pytest.raises(ValueError):
            _check_q_dofs(bad, 7, "x")
    dev = torch.device("cpu")
    _check_buffer(torch.zeros(8), 8, torch.float32, dev, "b")
    _check_buffer(torch.zeros(9), 8, torch.float32, dev, "b", at_least=True)
    for bad in (torch.zeros(7), torch.zeros(8, dtype=torch.float64), torch.zeros(16)[::2]):
        with pytest.raises(ValueError):
            _check_buffer(bad, 8, torch.float32, dev, "b")


def test_gp_grad_scale_bounds_the_fp16_gradient():
    """ops.gp_grad_scale: a power of two that keeps the worst-case GP-prior gradient of bounded trajectories below half the fp16 range
    (config 5's parameters: sigma_gp = 0.1, dt = 5 / 128 -> a = 12 / (sigma^2 dt^3) = 2e7)."""
    import numpy as np
    from torch_robotics_amd import ops
    dt, sigma = 5.0 / 128, 0.1
    gs = ops.gp_grad_scale(dt, sigma, 1.0, q_abs_max=2.0, qd_abs_max=1.0)
    assert 0 < gs < 1 and np.log2(gs) == np.round(np.log2(gs))
    # brute force over the corners of the box |q| <= 2, |qd| <= 1 for one joint and three consecutive time steps
    s2 = 1.0 / sigma ** 2
    a, b, c = 12 * s2 / dt ** 3, -6 * s2 / dt ** 2, 4 * s2 / dt
    worst = 0.0
    for bits in range(64):
        pm, p0, pn, vm, v0, vn = [(1 if (bits >> k) & 1 else -1) * (2.0 if k < 3 else 1.0) for k in range(6)]
        ep, ev, em, fm = p0 + dt * v0 - pn, v0 - vn, pm + dt * vm - p0, vm - v0
        rp, rv = a * ep + b * ev, b * ep + c * ev
        worst = max(worst, abs(rp - (a * em + b * fm)), abs(dt * rp + rv - (b * em + c * fm)))
    assert worst * gs <= 32768.0 and worst * gs * 4 > 32768.0 * 0.2          # safe, and not needlessly small
    assert ops.gp_grad_scale(0.1, 10.0, 1.0, 1.0, 1.0) == 1.0               # a weak prior needs no scale
    assert ops.gp_grad_scale(dt, sigma, 1.0, 2.0, 1.0, extra=1e9) < gs       # the other terms' gradient counts too


def test_primitive_geometry_is_read_only_and_part_of_the_scene_key():
    """an in-place edit of a pose or of a primitive's geometry must not leave a stale device model behind -- it raises;
    a re-assigned array changes the key every CostHandle cache uses (environments.scene_version)."""
    from torch_robotics_amd.environments import scene_version
    f = tra.MultiSphereField(np.zeros((2, 3), np.float32), np.full(2, 0.1, np.float32))
    b = tra.MultiBoxField(np.zeros((1, 3), np.float32), np.full((1, 3), 0.2, np.float32))
    o = tra.ObjectField([f, b], "o")
    for arr in (f.centers, f.radii, b.centers, b.sizes, b.half_sizes, b.radius, o.pos, o.ori):
        with pytest.raises(ValueError):
            arr[0] = 1.0
    k0 = scene_version([o])
    f.centers = f.centers + 0.5
    k1 = scene_version([o])
    assert k1 != k0 and not f.centers.flags.writeable
    o.pos = (0.1, 0.0, 0.0)
    assert scene_version([o]) != k1
