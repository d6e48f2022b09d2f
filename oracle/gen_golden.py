#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference on CPU (dev container only).

TEST INFRASTRUCTURE -- never imported by the product.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 oracle/gen_golden.py

Imports `/root/reference` (read-only, never copied) through the
`oracle/refshim/urdf_parser_py` stand-in for the one missing third-party parser,
runs the reference's own FK / cost / autograd code on seeded inputs and writes

* `torch_robotics_amd/data/urdf/*.urdf` -- kinematics-only robot descriptions
  (link names + joint origin/axis/limit; no meshes, inertia, visuals) derived
  from the reference's data files, plus two authored ones (`ur10_allegro`,
  `dual_panda`).  The reference is run on these stripped files AND on its own
  originals and the two results are asserted bit-identical.
* `tests/golden/*.npz` -- inputs and reference outputs (fp32) for every row of
  SURVEY.md section 8a.

The reference cannot travel to the GPU box; these small files can.
"""
import os
import sys
import xml.etree.ElementTree as ET
from pathlib import Path

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent
REF = Path("/root/reference")
sys.path.insert(0, str(HERE / "refshim"))
sys.path.insert(0, str(REF))

import io
import contextlib
import numpy as np
import torch

from torch_robotics.torch_kinematics_tree.models.robot_tree import DifferentiableTree  # noqa: E402
from torch_robotics.torch_kinematics_tree.geometrics.quaternion import rotation_matrix_to_q  # noqa: E402
from torch_robotics.torch_kinematics_tree.geometrics.utils import SE3_distance  # noqa: E402

GOLD = REPO / "tests" / "golden"
URDF_OUT = REPO / "torch_robotics_amd" / "data" / "urdf"
REF_URDF = REF / "torch_robotics" / "data" / "urdf" / "robots"
TA = dict(device="cpu", dtype=torch.float32)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ----------------------------------------------------------------------------
# kinematics-only URDFs
# ----------------------------------------------------------------------------
def strip_urdf(src: Path, dst: Path, prefix: str = ""):
    """Keep element order; keep only what FK reads."""
    root = ET.parse(src).getroot()
    out = ET.Element("robot", {"name": root.get("name", src.stem)})
    for elem in root:
        if elem.tag == "link":
            ET.SubElement(out, "link", {"name": prefix + elem.get("name")})
        elif elem.tag == "joint":
            j = ET.SubElement(out, "joint", {"name": prefix + elem.get("name"), "type": elem.get("type")})
            origin = elem.find("origin")
            if origin is not None:
                ET.SubElement(j, "origin", dict(origin.attrib))
            ET.SubElement(j, "parent", {"link": prefix + elem.find("parent").get("link")})
            ET.SubElement(j, "child", {"link": prefix + elem.find("child").get("link")})
            for tag in ("axis", "limit"):
                sub = elem.find(tag)
                if sub is not None:
                    ET.SubElement(j, tag, dict(sub.attrib))
    ET.indent(out, space="  ")
    dst.parent.mkdir(parents=True, exist_ok=True)
    ET.ElementTree(out).write(dst, encoding="utf-8", xml_declaration=True)
    return out


def author_ur10_allegro(dst: Path):
    """UR10 arm with the Allegro hand rigidly mounted on `ee_link` (BASELINE config 4)."""
    ur10 = ET.parse(URDF_OUT / "ur10.urdf").getroot()
    hand = ET.parse(URDF_OUT / "allegro_hand.urdf").getroot()
    out = ET.Element("robot", {"name": "ur10_allegro"})
    hand_links = [e for e in hand if e.tag == "link"]
    for e in ur10:
        if e.tag == "link":
            out.append(e)
    for e in hand_links:
        e.set("name", "allegro_" + e.get("name"))
        out.append(e)
    for e in ur10:
        if e.tag == "joint":
            out.append(e)
    mount = ET.SubElement(out, "joint", {"name": "allegro_mount_joint", "type": "fixed"})
    ET.SubElement(mount, "origin", {"xyz": "0.02 0 0", "rpy": "0 1.5707963267948966 0"})
    ET.SubElement(mount, "parent", {"link": "ee_link"})
    ET.SubElement(mount, "child", {"link": hand_links[0].get("name")})
    for e in hand:
        if e.tag == "joint":
            e.set("name", "allegro_" + e.get("name"))
            e.find("parent").set("link", "allegro_" + e.find("parent").get("link"))
            e.find("child").set("link", "allegro_" + e.find("child").get("link"))
            out.append(e)
    ET.indent(out, space="  ")
    ET.ElementTree(out).write(dst, encoding="utf-8", xml_declaration=True)


def author_dual_panda(dst: Path):
    """`world` + two Panda arms side by side (BASELINE config 5)."""
    out = ET.Element("robot", {"name": "dual_panda"})
    ET.SubElement(out, "link", {"name": "world"})
    for prefix, y, yaw in (("left_", 0.35, "0"), ("right_", -0.35, "0")):
        arm = ET.parse(URDF_OUT / "panda_arm_no_gripper.urdf").getroot()
        first_link = None
        elems = list(arm)
        for e in elems:
            if e.tag == "link":
                e.set("name", prefix + e.get("name"))
                first_link = first_link or e.get("name")
            else:
                e.set("name", prefix + e.get("name"))
                e.find("parent").set("link", prefix + e.find("parent").get("link"))
                e.find("child").set("link", prefix + e.find("child").get("link"))
        mount = ET.SubElement(out, "joint", {"name": prefix + "mount", "type": "fixed"})
        ET.SubElement(mount, "origin", {"xyz": f"0 {y} 0", "rpy": f"0 0 {yaw}"})
        ET.SubElement(mount, "parent", {"link": "world"})
        ET.SubElement(mount, "child", {"link": first_link})
        for e in elems:
            out.append(e)
    ET.indent(out, space="  ")
    ET.ElementTree(out).write(dst, encoding="utf-8", xml_declaration=True)


ROBOTS = {
    # name: path relative to the reference's robots dir
    "panda_arm_no_gripper": "franka_description/robots/panda_arm_no_gripper.urdf",
    "panda_arm_hand": "franka_description/robots/panda_arm_hand.urdf",
    "ur10": "ur10/urdf/ur10.urdf",
    "allegro_hand": "allegro_hand/allegro_hand.urdf",
    "iiwa7": "kuka_iiwa/urdf/iiwa7.urdf",
    "iiwa7_allegro": "kuka_iiwa/urdf/iiwa7_allegro.urdf",
    "shadow_hand": "shadow_hand/shadow_hand.urdf",
    "tiago_dual_holobase_minimal_holonomic": "tiago_dual_description/tiago_dual_holobase_minimal_holonomic.urdf",
    "hab_stretch": "habitat_stretch/urdf/hab_stretch.urdf",
}
AUTHORED = {"ur10_allegro": author_ur10_allegro, "dual_panda": author_dual_panda}


# ----------------------------------------------------------------------------
# FK goldens
# ----------------------------------------------------------------------------
def limits_of(tree):
    lo, hi = [], []
    for idx in tree._controlled_joints:
        lim = tree._bodies[idx].joint_limits
        if lim is None:
            lo.append(-np.pi); hi.append(np.pi)
        else:
            lo.append(float(lim["lower"])); hi.append(float(lim["upper"]))
    return np.asarray(lo), np.asarray(hi)


def sample_q(tree, n, gen, widen=0.0):
    lo, hi = limits_of(tree)
    span = hi - lo
    lo, hi = lo - widen * span, hi + widen * span
    u = torch.rand(n, len(lo), generator=gen, dtype=torch.float64)
    return (torch.as_tensor(lo) + u * torch.as_tensor(hi - lo)).to(torch.float32)


def model_params(tree):
    """What the reference's model build produced (A1)."""
    out = dict(
        link_names=np.array(tree.get_link_names()),
        n_dofs=np.int32(tree._n_dofs),
        controlled=np.asarray(tree._controlled_joints, np.int32),
        trans=np.stack([b.trans.reshape(3).numpy() for b in tree._bodies]),
        R_fixed=np.stack([b.fixed_rotation.reshape(3, 3).numpy() for b in tree._bodies]),
        axis=np.stack([b.joint_axis.reshape(3).numpy() for b in tree._bodies]),
        joint_type=np.array([b.joint_type for b in tree._bodies]),
        has_limits=np.array([b.joint_limits is not None for b in tree._bodies]),
        lower=np.array([float(b.joint_limits["lower"]) if b.joint_limits else 0.0 for b in tree._bodies]),
        upper=np.array([float(b.joint_limits["upper"]) if b.joint_limits else 0.0 for b in tree._bodies]),
        parent=np.array([-1] + [tree._name_to_idx_map[tree._model.get_name_of_parent_body(b.name)]
                                for b in tree._bodies[1:]], np.int32),
    )
    return out


def fk_golden(name, tree, seed):
    gen = torch.Generator().manual_seed(seed)
    L = len(tree.get_link_names())
    out = model_params(tree)
    for tag, widen in (("in", 0.0), ("out", 0.2)):
        q = sample_q(tree, 32, gen, widen).requires_grad_(True)
        H = tree.compute_forward_kinematics_all_links(q)
        w = torch.randn(32, L, 4, 4, generator=gen)
        (gq,) = torch.autograd.grad((w * H).sum(), q)
        out[f"q_{tag}"] = q.detach().numpy()
        out[f"H_{tag}"] = H.detach().numpy()
        out[f"w_{tag}"] = w.numpy()
        out[f"gq_{tag}"] = gq.numpy()
    # link subset + dict path
    names = tree.get_link_names()
    sel = [names[-1], names[len(names) // 2], names[1]]
    q = torch.as_tensor(out["q_in"])
    out["sel_names"] = np.array(sel)
    out["H_sel"] = tree.compute_forward_kinematics_all_links(q, link_list=sel).numpy()
    np.savez_compressed(GOLD / f"fk_{name}.npz", **out)
    return out


def jacobian_golden(name, tree, links, seed):
    gen = torch.Generator().manual_seed(seed)
    q = sample_q(tree, 32, gen, 0.1)
    qd = 0.1 * torch.randn(32, tree._n_dofs, generator=gen)
    out = dict(q=q.numpy(), qd=qd.numpy(), links=np.array(links))
    for k, link in enumerate(links):
        tree.reset()
        pos, quat, lin, ang = tree.compute_forward_kinematics_and_geometric_jacobian(q, qd, link)
        out[f"pos_{k}"] = pos.numpy(); out[f"quat_{k}"] = quat.numpy()
        out[f"lin_{k}"] = lin.numpy(); out[f"ang_{k}"] = ang.numpy()
        body = tree._bodies[tree._name_to_idx_map[link]]
        out[f"vel_lin_{k}"] = body.vel.lin.numpy(); out[f"vel_ang_{k}"] = body.vel.ang.numpy()
        tree.reset()
    np.savez_compressed(GOLD / f"jac_{name}.npz", **out)


def analytic_jacobian_golden(name, tree, seed):
    """A16: compute_analytical_jacobian_all_links (autograd Jacobian of [pos, quat_wxyz] of every link)."""
    gen = torch.Generator().manual_seed(seed)
    q = sample_q(tree, 12, gen, 0.15)            # some samples beyond the limits: clamped joints give zero columns
    J = tree.compute_analytical_jacobian_all_links(q)
    np.savez_compressed(GOLD / f"ajac_{name}.npz", q=q.numpy(), J=J.numpy())


def quat_golden():
    gen = torch.Generator().manual_seed(77)
    # random rotations incl. near-180deg cases to hit every branch of the 4-candidate rule
    A = torch.randn(256, 3, 3, generator=gen, dtype=torch.float64)
    Q, _ = torch.linalg.qr(A)
    Q = Q * torch.sign(torch.linalg.det(Q)).reshape(-1, 1, 1)
    special = torch.stack([torch.diag(torch.tensor(d, dtype=torch.float64)) for d in
                           ([1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1])])
    R = torch.cat([Q, special]).to(torch.float32)
    np.savez_compressed(GOLD / "quat.npz", R=R.numpy(), q_wxyz=rotation_matrix_to_q(R).numpy())


# ----------------------------------------------------------------------------
# cost goldens (Panda)
# ----------------------------------------------------------------------------
def scene_arrays(env):
    """Flatten the env's analytic objects into data (so the build's own scene tables can be checked)."""
    from torch_robotics.environments.primitives import MultiSphereField, MultiBoxField, MultiSharpBoxField
    out = {}

    def dump(objs, tag):
        for oi, obj in enumerate(objs or []):
            out[f"{tag}{oi}_pos"] = obj.pos.numpy().astype(np.float32)
            out[f"{tag}{oi}_ori"] = obj.ori.numpy().astype(np.float32)
            for fi, f in enumerate(obj.fields):
                key = f"{tag}{oi}_f{fi}"
                if isinstance(f, MultiSphereField):
                    out[key + "_kind"] = np.array("sphere")
                    out[key + "_centers"] = f.centers.numpy(); out[key + "_radii"] = f.radii.numpy()
                elif isinstance(f, MultiBoxField):
                    out[key + "_kind"] = np.array("roundbox")
                    out[key + "_centers"] = f.centers.numpy(); out[key + "_sizes"] = f.sizes.numpy()
                    out[key + "_radius"] = f.radius.numpy()
                elif isinstance(f, MultiSharpBoxField):
                    out[key + "_kind"] = np.array("sharpbox")
                    out[key + "_centers"] = f.centers.numpy(); out[key + "_sizes"] = f.sizes.numpy()
    dump(env.obj_fixed_list, "fixed")
    dump(env.obj_extra_list, "extra")
    out["limits"] = env.limits.numpy()
    return out


def cost_goldens():
    from torch_robotics.robots.robot_panda import RobotPanda
    from torch_robotics.tasks.tasks import PlanningTask
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    from torch_robotics.environments.env_table_shelf import EnvTableShelf
    from torch_robotics.environments.env_maze_boxes_3d import EnvMazeBoxes3D
    from torch_robotics.environments.env_spheres_3d_extra_objects import EnvSpheres3DExtraObjects
    from torch_robotics.torch_planning_objectives.fields.distance_fields import EESE3DistanceField

    robot = quiet(RobotPanda, tensor_args=TA)
    tree = robot.diff_panda
    gen = torch.Generator().manual_seed(2024)
    q0 = sample_q(tree, 64, gen, 0.1).reshape(8, 8, 7)

    meta = dict(
        obj_link_idxs=np.asarray(robot.link_idxs_for_object_collision_checking, np.int32),
        obj_link_margins=robot.link_margins_for_object_collision_checking_tensor.numpy(),
        self_link_idxs=np.asarray(robot.link_idxs_for_self_collision_checking, np.int32),
        self_pairs=np.asarray(robot.df_collision_self.idxs_links_distance_matrix, np.int32),
        self_margins=robot.df_collision_self.cutoff_margin.numpy(),
        q_limits=robot.q_limits.numpy(),
        q=q0.numpy(),
        fk_map_collision=robot.fk_map_collision(q0).numpy(),
        ee_pose=robot.get_EE_pose(q0.reshape(-1, 7)).numpy(),
    )
    np.savez_compressed(GOLD / "panda_robot.npz", **meta)

    envs = {
        "spheres3d": (lambda: EnvSpheres3D(tensor_args=TA), 0.03),
        "spheres3d_grid": (lambda: quiet(EnvSpheres3D, tensor_args=TA, precompute_sdf_obj_fixed=True,
                                         sdf_cell_size=0.1), 0.03),
        "table_shelf": (lambda: EnvTableShelf(tensor_args=TA), 0.01),
        "maze_boxes3d": (lambda: EnvMazeBoxes3D(tensor_args=TA), 0.01),
        "spheres3d_extra": (lambda: EnvSpheres3DExtraObjects(tensor_args=TA), 0.01),
    }
    for name, (make_env, cutoff) in envs.items():
        env = make_env()
        task = PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=cutoff, tensor_args=TA)
        out = scene_arrays(env)
        out["cutoff"] = np.float32(cutoff)
        out["q"] = q0.numpy()
        if env.grid_map_sdf_obj_fixed is not None:
            g = env.grid_map_sdf_obj_fixed
            out["grid_sdf"] = g.sdf_tensor.numpy(); out["grid_grad"] = g.grad_sdf_tensor.numpy()
            out["grid_cmap_dim"] = g.cmap_dim.numpy().astype(np.int32)
            out["grid_cell"] = np.float32(g.cell_size)
        # per-field cost + grad w.r.t. link positions (unfused ops) and w.r.t. q (through FK)
        fields = dict(self=task.df_collision_self, objects=task.df_collision_objects,
                      ws=task.df_collision_ws_boundaries)
        if task._collision_fields_extra_objects:
            fields["extra"] = task._collision_fields_extra_objects[0]
        for fname, fld in fields.items():
            q = q0.clone().requires_grad_(True)
            pos = robot.fk_map_collision(q)
            pos_leaf = pos.detach().clone().requires_grad_(True)
            c = fld.compute_cost(q, pos_leaf, field_type="sdf")
            (gpos,) = torch.autograd.grad(c.sum(), pos_leaf)
            c2 = fld.compute_cost(q, pos, field_type="sdf")
            (gq,) = torch.autograd.grad(c2.sum(), q)
            out[f"cost_{fname}"] = c.detach().numpy()
            out[f"gpos_{fname}"] = gpos.numpy()
            out[f"gq_{fname}"] = gq.numpy()
            out[f"coll_{fname}"] = fld.compute_cost(q0, pos.detach(), field_type="occupancy").numpy()
            out[f"coll0_{fname}"] = fld.compute_cost(q0, pos.detach(), field_type="occupancy", margin=0.).numpy()
        q = q0.clone().requires_grad_(True)
        total = task.compute_collision_cost(q)
        (gq,) = torch.autograd.grad(total.sum(), q)
        out["cost_total"] = total.detach().numpy(); out["gq_total"] = gq.numpy()
        out["coll_total"] = task.compute_collision(q0).numpy()
        out["coll0_total"] = task.compute_collision(q0, margin=0.).numpy()
        np.savez_compressed(GOLD / f"cost_{name}.npz", **out)

    # EE SE(3) tracking (A13)
    out = dict(q=q0.numpy())
    targets = []
    gen = torch.Generator().manual_seed(5)
    Ht0 = torch.eye(4); Ht0[:3, 3] = torch.tensor([0.4, 0.2, 0.5])
    targets.append(Ht0)
    for _ in range(2):
        qq = sample_q(tree, 1, gen, 0.0)
        targets.append(robot.get_EE_pose(qq)[0, 0])
    per_sample = robot.get_EE_pose(sample_q(tree, 64, gen, 0.0))[:, 0]
    for k, Ht in enumerate(targets + [per_sample]):
        for sq in (True, False):
            for (wp, wr) in ((1.0, 1.0), (2.0, 0.5)):
                fld = EESE3DistanceField(Ht, w_pos=wp, w_rot=wr, square=sq, tensor_args=TA)
                q = q0.reshape(-1, 7).clone().requires_grad_(True)
                H = tree.compute_forward_kinematics_all_links(q)
                Hleaf = H.detach().clone().requires_grad_(True)
                c = fld.compute_costs_impl(q, Hleaf)
                (gH,) = torch.autograd.grad(c.sum(), Hleaf)
                c2 = fld.compute_costs_impl(q, H)
                (gq,) = torch.autograd.grad(c2.sum(), q)
                key = f"t{k}_sq{int(sq)}_w{wp}_{wr}"
                out["target_%d" % k] = Ht.numpy()
                out["cost_" + key] = c.detach().numpy()
                out["gH_" + key] = gH.numpy()
                out["gq_" + key] = gq.numpy()
    np.savez_compressed(GOLD / "cost_ee.npz", **out)

    # bench-shaped composite goldens (C2: objects + EE, C3: self + objects + ws + EE), B=6 x H=64
    env = EnvSpheres3D(tensor_args=TA)
    task = PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    gen = torch.Generator().manual_seed(1234)
    qb = sample_q(tree, 6 * 64, gen, 0.05).reshape(6, 64, 7)
    fld = EESE3DistanceField(Ht0, w_pos=1.0, w_rot=1.0, square=True, tensor_args=TA)
    out = dict(q=qb.numpy(), target=Ht0.numpy(), cutoff=np.float32(0.03))

    def ee_cost(q):
        H = tree.compute_forward_kinematics_all_links(q.reshape(-1, 7))
        return fld.compute_costs_impl(q, H).reshape(q.shape[:-1])

    q = qb.clone().requires_grad_(True)
    pos = robot.fk_map_collision(q)
    c_obj = task.df_collision_objects.compute_cost(q, pos, field_type="sdf")
    c2 = c_obj + ee_cost(q)
    (g2,) = torch.autograd.grad(c2.sum(), q)
    out["pos"] = pos.detach().numpy(); out["cost_c2"] = c2.detach().numpy(); out["gq_c2"] = g2.numpy()
    q = qb.clone().requires_grad_(True)
    c3 = task.compute_collision_cost(q) + ee_cost(q)
    (g3,) = torch.autograd.grad(c3.sum(), q)
    out["cost_c3"] = c3.detach().numpy(); out["gq_c3"] = g3.numpy()
    np.savez_compressed(GOLD / "rollout_panda.npz", **out)


def trajs_goldens():
    """8f rank 1: PlanningTask.get_trajs_collision_and_free (tasks.py:234-308) and the fraction / intensity stats."""
    from torch_robotics.robots.robot_panda import RobotPanda
    from torch_robotics.tasks.tasks import PlanningTask
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    robot = quiet(RobotPanda, tensor_args=TA)
    task = PlanningTask(env=EnvSpheres3D(tensor_args=TA), robot=robot, obstacle_cutoff_margin=0.03, tensor_args=TA)
    gen = torch.Generator().manual_seed(77)
    lo, hi = robot.q_min, robot.q_max
    # 12 random-walk trajectories of 16 via points; some leave the joint limits, some hit obstacles
    start = lo + torch.rand(12, 1, 7, generator=gen) * (hi - lo)
    steps = 0.08 * torch.randn(12, 16, 7, generator=gen)
    trajs = start + torch.cumsum(steps, dim=1)
    trajs[:4] = torch.clamp(trajs[:4], lo + 0.05, hi - 0.05)
    out = dict(trajs=trajs.numpy())
    coll, coll_idx, free, free_idx, wp = task.get_trajs_collision_and_free(trajs, return_indices=True, num_interpolation=5)
    out["waypoints_collisions"] = wp.numpy()
    out["coll_idx"] = coll_idx.numpy(); out["free_idx"] = free_idx.numpy()
    out["n_coll"] = np.int32(0 if coll is None else coll.shape[0]); out["n_free"] = np.int32(0 if free is None else free.shape[0])
    out["fraction_free"] = np.float64(task.compute_fraction_free_trajs(trajs))
    out["collision_intensity"] = np.float64(task.compute_collision_intensity_trajs(trajs))
    out["success"] = np.int32(task.compute_success_free_trajs(trajs))
    t4 = trajs.reshape(3, 4, 16, 7)
    coll4, coll_idx4, free4, free_idx4, wp4 = task.get_trajs_collision_and_free(t4, return_indices=True)
    out["waypoints_collisions4"] = wp4.numpy(); out["coll_idx4"] = coll_idx4.numpy(); out["free_idx4"] = free_idx4.numpy()
    np.savez_compressed(GOLD / "trajs_panda.npz", **out)


def metrics_goldens():
    """A17: compute_path_length / compute_smoothness (trajectory/metrics.py:7-12, 27-35), RobotBase.get_velocity /
    get_acceleration (robot_base.py:151-166).  `python oracle/gen_golden.py metrics` regenerates only this file."""
    from torch_robotics.robots.robot_panda import RobotPanda
    from torch_robotics.trajectory.metrics import compute_path_length, compute_smoothness
    robot = quiet(RobotPanda, tensor_args=TA)
    gen = torch.Generator().manual_seed(88)
    trajs = torch.cumsum(0.05 * torch.randn(9, 33, 7, generator=gen), dim=1)
    full = torch.cat([trajs, 0.3 * torch.randn(9, 33, 7, generator=gen)], dim=-1)      # positions + velocities
    np.savez_compressed(GOLD / "metrics_panda.npz", trajs=trajs.numpy(), full=full.numpy(),
                        path_length=compute_path_length(trajs, robot).numpy(),
                        smoothness_fd=compute_smoothness(trajs, robot).numpy(),       # velocities by finite differences
                        smoothness_vel=compute_smoothness(full, robot).numpy(),       # velocities carried in the state
                        vel_fd=robot.get_velocity(trajs).numpy(), acc_fd=robot.get_acceleration(trajs).numpy())
    print("metrics_panda: 9 x 33")


def pointmass_goldens():
    """RobotPointMass3D (robot_point_mass.py:101-109: identity FK, one collision point) through PlanningTask on
    EnvSpheres3D.  `python oracle/gen_golden.py pointmass` regenerates only this file."""
    from torch_robotics.robots.robot_point_mass import RobotPointMass3D
    from torch_robotics.tasks.tasks import PlanningTask
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    robot = quiet(RobotPointMass3D, tensor_args=TA)
    env = EnvSpheres3D(tensor_args=TA)
    task = PlanningTask(env=env, robot=robot, obstacle_cutoff_margin=0.02, tensor_args=TA)
    gen = torch.Generator().manual_seed(99)
    q = ((torch.rand(6, 16, 3, generator=gen) - 0.5) * 2.2).requires_grad_(True)       # some points outside the workspace box
    cost = task.compute_collision_cost(q)
    (gq,) = torch.autograd.grad(cost.sum(), q)
    np.savez_compressed(GOLD / "pointmass3d.npz", q=q.detach().numpy(), cost=cost.detach().numpy(), gq=gq.numpy(),
                        coll=task.compute_collision(q.detach()).numpy(), coll0=task.compute_collision(q.detach(), margin=0.0).numpy(),
                        fk=robot.fk_map_collision(q.detach()).numpy(), q_limits=robot.q_limits.numpy(),
                        margins=robot.link_margins_for_object_collision_checking_tensor.numpy(), cutoff=np.float32(0.02),
                        limits=env.limits.numpy(), has_self=np.bool_(robot.df_collision_self is not None))
    print("pointmass3d:", tuple(cost.shape), "self field:", robot.df_collision_self is not None)


def ik_goldens(trees):
    """8f rank 2: loss_fn_ik_per_q / ik_termination (robot_tree.py:386-442) and Adam steps on it."""
    tree = trees["panda_arm_no_gripper"]
    gen = torch.Generator().manual_seed(404)
    lower, upper, _, _ = tree.get_joint_limit_array()
    eps = np.pi / 100
    lower_t = torch.as_tensor(lower + eps, dtype=torch.float32); upper_t = torch.as_tensor(upper - eps, dtype=torch.float32)
    q0 = sample_q(tree, 48, gen, 0.1)                      # some joints outside the (shrunk) limits
    q_goal = sample_q(tree, 48, gen, 0.0)
    H_target = tree.compute_forward_kinematics_all_links(q_goal, link_list=["ee_link"]).squeeze(1).detach()
    out = dict(q0=q0.numpy(), H_target=H_target.numpy(), lower=lower_t.numpy(), upper=upper_t.numpy())
    for tag, Ht in (("per_sample", H_target), ("single", H_target[:1])):
        q = q0.clone().requires_grad_(True)
        opt = torch.optim.Adam([q], lr=1e-2)
        traj_q, traj_err = [], []
        for it in range(5):
            opt.zero_grad()
            idx_valid = tree.ik_termination(q, Ht, "ee_link", lower_t, upper_t, se3_eps=1e-1)
            err = tree.loss_fn_ik_per_q(q, Ht, "ee_link", w_se3=1.0, w_joint_limits=300.0, lower=lower_t, upper=upper_t)
            if it == 0:
                (g0,) = torch.autograd.grad(err.sum(), q, retain_graph=True)
                out[f"loss0_{tag}"] = err.detach().numpy(); out[f"grad0_{tag}"] = g0.numpy()
                valid = np.zeros(48, bool); valid[idx_valid.numpy()] = True
                out[f"valid0_{tag}"] = valid
            err.sum().backward(); opt.step()
            traj_q.append(q.detach().clone().numpy()); traj_err.append(err.detach().numpy())
        out[f"q_steps_{tag}"] = np.stack(traj_q); out[f"err_steps_{tag}"] = np.stack(traj_err)
    np.savez_compressed(GOLD / "ik_panda.npz", **out)


def misc_goldens():
    """finite differences / smoothness (A17) and via-point interpolation (8f rank 1)."""
    from torch_robotics.trajectory.utils import finite_difference_vector, interpolate_traj_via_points
    from torch_robotics.trajectory.metrics import compute_smoothness
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 16, 7, generator=gen)
    out = dict(x=x.numpy())
    for m in ("forward", "backward", "central"):
        out["fd_" + m] = finite_difference_vector(x, dt=0.25, method=m).numpy()
    out["interp5"] = interpolate_traj_via_points(x, num_interpolation=5).numpy()
    try:
        out["smoothness"] = compute_smoothness(x, None).numpy()
    except Exception:
        pass
    np.savez_compressed(GOLD / "traj.npz", **out)


# ----------------------------------------------------------------------------
# points fixed in link frames (grasped-object points; Frame.transform_point)
# ----------------------------------------------------------------------------
def points_goldens():
    """`python oracle/gen_golden.py points` regenerates only these files."""
    # (1) RobotPanda with a grasped box: fk_map_collision -> (b, h, 12 links + 14 points, 3) and its gradient.
    # The reference would write a modified URDF into its own data directory (robots.py:47-51); its tree ships that
    # file pre-generated, so the writer is replaced by a function returning the path of the shipped file.
    import torch_robotics.torch_kinematics_tree.models.robots as ref_robots
    from torch_robotics.environments.objects import GraspedObjectPandaBox
    ref_robots.modidy_franka_panda_urdf_grasped_object = \
        lambda robot_file, grasped_object: Path(str(robot_file).replace(".urdf", "_grasped_object.urdf"))
    from torch_robotics.robots.robot_panda import RobotPanda

    name = "panda_arm_no_gripper_grasped_object"
    rel = f"franka_description/robots/{name}.urdf"
    strip_urdf(REF_URDF / rel, URDF_OUT / f"{name}.urdf")
    t_strip = quiet(DifferentiableTree, str(URDF_OUT / f"{name}.urdf"), name)
    go = GraspedObjectPandaBox(tensor_args=TA)
    robot = quiet(RobotPanda, grasped_object=go, tensor_args=TA)
    torch.manual_seed(77)
    gen = torch.Generator().manual_seed(78)
    q = robot.random_q(32).view(4, 8, 7).clone().requires_grad_(True)
    lp = robot.fk_map_collision(q)
    w = torch.randn(lp.shape, generator=gen)
    (gq,) = torch.autograd.grad((w * lp).sum(), q)
    q2 = sample_q(robot.diff_panda, 16, gen, 0.2)
    assert torch.equal(robot.diff_panda.compute_forward_kinematics_all_links(q2),
                       t_strip.compute_forward_kinematics_all_links(q2))
    q2 = q2.requires_grad_(True)
    lp2 = robot.fk_map_collision(q2)
    w2 = torch.randn(lp2.shape, generator=gen)
    (gq2,) = torch.autograd.grad((w2 * lp2).sum(), q2)
    # The reference's cost path with a grasped object raises (compute_costs_impl distance_fields.py:134-155 concatenates
    # the grasped points onto the full link tensor, then margins (5+14) meet 5 columns; the self field's extra pair rows
    # index past its 8 selected links).  The evident intent -- robot collision links followed by the grasped points,
    # with the margins and pair tables RobotBase builds (robot_base.py:71-141) -- is evaluated here with the reference's
    # own field code by giving each field the point columns explicitly and calling compute_embodiment_cost directly.
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    from torch_robotics.torch_planning_objectives.fields.distance_fields import (
        CollisionObjectDistanceField, CollisionSelfField, CollisionWorkspaceBoundariesDistanceField)
    env = EnvSpheres3D(tensor_args=TA)
    L, G = len(robot.diff_panda.get_link_names()), go.n_base_points_for_collision
    grasp_cols = list(range(L, L + G))
    obj_cols = list(robot.link_idxs_for_object_collision_checking) + grasp_cols
    self_cols = list(robot.link_idxs_for_self_collision_checking) + grasp_cols
    cutoff = 0.03
    f_obj = CollisionObjectDistanceField(
        robot, df_obj_list_fn=env.get_df_obj_list, link_idxs_for_collision_checking=obj_cols,
        link_margins_for_object_collision_checking_tensor=robot.link_margins_for_object_collision_checking_tensor,
        cutoff_margin=cutoff, tensor_args=TA)
    f_ws = CollisionWorkspaceBoundariesDistanceField(
        robot, link_idxs_for_collision_checking=obj_cols,
        link_margins_for_object_collision_checking_tensor=robot.link_margins_for_object_collision_checking_tensor,
        cutoff_margin=cutoff, ws_bounds_min=env.limits[0], ws_bounds_max=env.limits[1], tensor_args=TA)
    f_self = CollisionSelfField(
        robot, link_idxs_for_collision_checking=self_cols,
        idxs_links_distance_matrix=robot.df_collision_self.idxs_links_distance_matrix,
        cutoff_margin=robot.df_collision_self.cutoff_margin, tensor_args=TA)
    qc = q.detach().clone().requires_grad_(True)
    lpc = robot.fk_map_collision(qc)
    flat = lpc.reshape(-1, L + G, 3)
    costs = {k: f.compute_embodiment_cost(None, flat).reshape(4, 8) for k, f in (("self", f_self), ("obj", f_obj), ("ws", f_ws))}
    coll = {k: f.compute_embodiment_cost(None, flat, field_type="occupancy").reshape(4, 8)
            for k, f in (("self", f_self), ("obj", f_obj), ("ws", f_ws))}
    coll0 = {k: f.compute_embodiment_cost(None, flat, field_type="occupancy", margin=0.0).reshape(4, 8)
             for k, f in (("self", f_self), ("obj", f_obj), ("ws", f_ws))}
    total = costs["self"] + costs["obj"] + costs["ws"]
    (g_lp,) = torch.autograd.grad(total.sum(), lpc, retain_graph=True)
    (g_q,) = torch.autograd.grad(total.sum(), qc)
    np.savez_compressed(
        GOLD / "grasp_panda.npz",
        cutoff=np.float32(cutoff), limits=env.limits.numpy(),
        cost_self=costs["self"].detach().numpy(), cost_obj=costs["obj"].detach().numpy(),
        cost_ws=costs["ws"].detach().numpy(), g_link_pos=g_lp.numpy(), gq_cost=g_q.numpy(),
        coll_self=coll["self"].numpy(), coll_obj=coll["obj"].numpy(), coll_ws=coll["ws"].numpy(),
        coll0_self=coll0["self"].numpy(), coll0_obj=coll0["obj"].numpy(), coll0_ws=coll0["ws"].numpy(),
        link_names=np.array(robot.diff_panda.get_link_names()),
        grasp_link=np.array(robot.link_name_grasped_object),
        base_points=go.base_points_for_collision.numpy(),
        grasp_pos=go.pos.numpy().reshape(3), grasp_ori=go.ori.numpy().reshape(4),
        obj_margins=robot.link_margins_for_object_collision_checking_tensor.numpy(),
        obj_link_idxs=np.asarray(robot.link_idxs_for_object_collision_checking, np.int32),
        self_link_idxs=np.asarray(robot.link_idxs_for_self_collision_checking, np.int32),
        self_pairs=np.asarray(robot.df_collision_self.idxs_links_distance_matrix, np.int32),
        self_margins=robot.df_collision_self.cutoff_margin.numpy(),
        q=q.detach().numpy(), link_pos=lp.detach().numpy(), w=w.numpy(), gq=gq.numpy(),
        q_out=q2.detach().numpy(), link_pos_out=lp2.detach().numpy(), w_out=w2.numpy(), gq_out=gq2.numpy())
    print("grasp_panda:", tuple(lp.shape))

    # (2) random points on random links of the tree robots
    for k, name in enumerate(("ur10_allegro", "dual_panda", "hab_stretch")):
        tree = quiet(DifferentiableTree, str(URDF_OUT / f"{name}.urdf"), name)
        names = tree.get_link_names()
        gen = torch.Generator().manual_seed(900 + k)
        P = 40
        plink = torch.randint(0, len(names), (P,), generator=gen)
        poff = (torch.rand(P, 3, generator=gen) - 0.5) * 0.4
        poff[::7] = 0.0                                     # some points at link origins
        q = sample_q(tree, 24, gen, 0.2).requires_grad_(True)
        frames = tree.compute_forward_kinematics_all_links(q, return_dict=True)
        cols = [frames[names[int(plink[j])]].transform_point(poff[j:j + 1])[:, 0, :] for j in range(P)]
        pos = torch.stack(cols, dim=1)                      # (N, P, 3)
        w = torch.randn(pos.shape, generator=gen)
        (gq,) = torch.autograd.grad((w * pos).sum(), q)
        np.savez_compressed(GOLD / f"points_{name}.npz", point_link=plink.numpy().astype(np.int32), point_offset=poff.numpy(),
                            q=q.detach().numpy(), pos=pos.detach().numpy(), w=w.numpy(), gq=gq.numpy())
        print(f"points_{name}:", tuple(pos.shape))


def sphere_config_data():
    """`python oracle/gen_golden.py spheres`: the Panda link-sphere table the reference ships but never reads
    (data/configs/panda/panda_sphere_config.yaml) -> torch_robotics_amd/data/configs/ (active entries only)."""
    import yaml
    src = REF / "torch_robotics" / "data" / "configs" / "panda" / "panda_sphere_config.yaml"
    table = yaml.safe_load(src.read_text())
    dst = REPO / "torch_robotics_amd" / "data" / "configs" / "panda_sphere_config.yaml"
    dst.parent.mkdir(parents=True, exist_ok=True)
    lines = ["# link -> [x, y, z, radius] in the link frame (collision spheres of the Franka Panda)"]
    n = 0
    for link, rows in table.items():
        if not isinstance(rows, list):                  # trailing scalar entry (`margin`), kept as is
            lines.append(f"{link}: {rows!r}")
            continue
        lines.append(f"{link}:")
        lines += [f"- [{', '.join(repr(float(v)) for v in row)}]" for row in rows]
        n += len(rows)
    dst.write_text("\n".join(lines) + "\n")
    print("panda_sphere_config:", n, "spheres")


def clamp_goldens():
    """clamp_sdf=True (distance_fields.py:114-117: relu(margin - sdf) before the max over objects and the sum over links) on the
    three fields a PlanningTask builds, for two scenes; same q as cost_<env>.npz, whose scene arrays the tests reuse."""
    from torch_robotics.robots.robot_panda import RobotPanda
    from torch_robotics.tasks.tasks import PlanningTask
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    from torch_robotics.environments.env_table_shelf import EnvTableShelf
    robot = quiet(RobotPanda, tensor_args=TA)
    gen = torch.Generator().manual_seed(2024)
    q0 = sample_q(robot.diff_panda, 64, gen, 0.1).reshape(8, 8, 7)
    out = dict(q=q0.numpy())
    tight_ws = torch.tensor([[-0.35, -0.35, 0.0], [0.35, 0.35, 0.75]], **TA)
    out["tight_ws"] = tight_ws.numpy()
    for name, make_env, cutoff, ws, self_margin in (
            ("spheres3d", lambda: EnvSpheres3D(tensor_args=TA), 0.03, None, None),
            ("table_shelf", lambda: EnvTableShelf(tensor_args=TA), 0.01, None, None),
            # a workspace the arm leaves and a self-collision margin it violates, so that all three hinges are active
            ("spheres3d_tight", lambda: EnvSpheres3D(tensor_args=TA), 0.03, tight_ws, 0.3)):
        task = PlanningTask(env=make_env(), robot=robot, obstacle_cutoff_margin=cutoff, tensor_args=TA,
                            **({} if ws is None else {"ws_limits": ws}))
        fields = dict(self=task.df_collision_self, objects=task.df_collision_objects, ws=task.df_collision_ws_boundaries)
        old_self_margin = task.df_collision_self.cutoff_margin
        if self_margin is not None:
            task.df_collision_self.cutoff_margin = torch.full_like(old_self_margin, self_margin)
            out["tight_self_margin"] = task.df_collision_self.cutoff_margin.numpy()
        for fld in fields.values():
            fld.clamp_sdf = True
        try:
            for fname, fld in fields.items():
                q = q0.clone().requires_grad_(True)
                pos = robot.fk_map_collision(q)
                pos_leaf = pos.detach().clone().requires_grad_(True)
                c = fld.compute_cost(q, pos_leaf, field_type="sdf")
                (gpos,) = torch.autograd.grad(c.sum(), pos_leaf)
                (gq,) = torch.autograd.grad(fld.compute_cost(q, pos, field_type="sdf").sum(), q)
                out[f"{name}_cost_{fname}"], out[f"{name}_gpos_{fname}"], out[f"{name}_gq_{fname}"] = c.detach().numpy(), gpos.numpy(), gq.numpy()
            q = q0.clone().requires_grad_(True)
            total = task.compute_collision_cost(q)
            (gq,) = torch.autograd.grad(total.sum(), q)
            out[f"{name}_cost_total"], out[f"{name}_gq_total"] = total.detach().numpy(), gq.numpy()
        finally:
            for fld in fields.values():
                fld.clamp_sdf = False          # robot.df_collision_self is shared
            task.df_collision_self.cutoff_margin = old_self_margin
    np.savez_compressed(GOLD / "cost_clamp.npz", **out)


def frame_goldens():
    """geometrics/frame.py:55-121 -- Frame.inverse / multiply_transform / multiply_inv_transform / transform_point /
    get_quaternion (trace method, xyzw) / get_euler on random poses, incl. the diagonal rotations that take every branch of
    the trace method, a batch-1 frame broadcast against a batch, and gradients of a fixed weighted sum of the outputs."""
    from torch_robotics.torch_kinematics_tree.geometrics.frame import Frame
    gen = torch.Generator().manual_seed(91)

    def rand_rot(n):
        A = torch.randn(n, 3, 3, generator=gen, dtype=torch.float64)
        Q, _ = torch.linalg.qr(A)
        return (Q * torch.sign(torch.linalg.det(Q)).reshape(-1, 1, 1)).to(torch.float32)

    special = torch.stack([torch.diag(torch.tensor(d, dtype=torch.float32)) for d in
                           ([1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1])])
    Ra = torch.cat([rand_rot(60), special]).requires_grad_(True)
    ta = torch.randn(64, 3, generator=gen).requires_grad_(True)
    Rb = rand_rot(64).requires_grad_(True)
    tb = torch.randn(64, 3, generator=gen).requires_grad_(True)
    R1, t1 = rand_rot(1), torch.randn(1, 3, generator=gen)
    pts = torch.randn(14, 3, generator=gen)
    wR, wt, wp = torch.randn(64, 3, 3, generator=gen), torch.randn(64, 3, generator=gen), torch.randn(64, 14, 3, generator=gen)
    fa, fb, f1 = Frame(Ra, ta), Frame(Rb, tb), Frame(R1, t1)
    out = dict(Ra=Ra.detach().numpy(), ta=ta.detach().numpy(), Rb=Rb.detach().numpy(), tb=tb.detach().numpy(),
               R1=R1.numpy(), t1=t1.numpy(), pts=pts.numpy(), wR=wR.numpy(), wt=wt.numpy(), wp=wp.numpy())

    def grads(loss):
        g = torch.autograd.grad(loss, [Ra, ta, Rb, tb], allow_unused=True)
        return [np.zeros(x.shape, np.float32) if gi is None else gi.numpy() for gi, x in zip(g, [Ra, ta, Rb, tb])]

    inv = fa.inverse()
    out["inv_R"], out["inv_t"] = inv.rotation.detach().numpy(), inv.translation.detach().numpy()
    out["inv_gRa"], out["inv_gta"], _, _ = grads((inv.rotation * wR).sum() + (inv.translation * wt).sum())
    mul = fa.multiply_transform(fb)
    out["mul_R"], out["mul_t"] = mul.rotation.detach().numpy(), mul.translation.detach().numpy()
    out["mul_gRa"], out["mul_gta"], out["mul_gRb"], out["mul_gtb"] = grads((mul.rotation * wR).sum() + (mul.translation * wt).sum())
    mi = fa.multiply_inv_transform(fb)                      # = fb^-1 o fa
    out["mulinv_R"], out["mulinv_t"] = mi.rotation.detach().numpy(), mi.translation.detach().numpy()
    out["mulinv_gRa"], out["mulinv_gta"], out["mulinv_gRb"], out["mulinv_gtb"] = \
        grads((mi.rotation * wR).sum() + (mi.translation * wt).sum())
    m1 = f1.multiply_transform(fb)                          # batch-1 frame broadcast over a batch
    out["mul1_R"], out["mul1_t"] = m1.rotation.detach().numpy(), m1.translation.detach().numpy()
    tp = fa.transform_point(pts)
    out["tp"] = tp.detach().numpy()
    out["tp_gRa"], out["tp_gta"], _, _ = grads((tp * wp).sum())
    # axis rotations and quaternion -> rotation (spatial_vector.py:8-47, quaternion.py:102-120), with d sum(w R) / d angle
    from torch_robotics.torch_kinematics_tree.geometrics.spatial_vector import x_rot, y_rot, z_rot
    from torch_robotics.torch_kinematics_tree.geometrics.quaternion import q_to_rotation_matrix
    ang = ((torch.rand(64, generator=gen) - 0.5) * 8.0).requires_grad_(True)
    out["angle"] = ang.detach().numpy()
    for nm, fn in (("x", x_rot), ("y", y_rot), ("z", z_rot)):
        Rr = fn(ang.unsqueeze(1))            # (B,1), as rigid_body.py calls it; a (B,) vector trips to_torch_2d_min
        out[f"rot_{nm}"] = Rr.detach().numpy()
        out[f"rot_{nm}_gangle"] = torch.autograd.grad((Rr * wR).sum(), [ang])[0].numpy()
    qu = torch.randn(64, 4, generator=gen)
    qu[:8] *= 3.0                                             # not normalised: the formula divides by |q|^2
    out["quat_in"], out["quat_R"] = qu.numpy(), q_to_rotation_matrix(qu).numpy()
    with torch.no_grad():
        out["quat_xyzw"] = fa.get_quaternion().numpy()
        out["euler"] = torch.stack(fa.get_euler(), -1).numpy()
        out["H"] = fa.get_transform_matrix().numpy()
    np.savez_compressed(GOLD / "frame_algebra.npz", **out)


def interp_goldens():
    """`python oracle/gen_golden.py interp`.  (1) interpolate_points_v1 itself (distance_fields.py:66-69).  (2) The three
    embodiment fields on interpolated link points, the way RobotBase lays them out (robot_base.py:57-73, 103-108: K = points
    per link x links, margins repeat_interleave'd, self pairs (i*p+m, j*p+n)).  The reference's own switch for this,
    `interpolate_link_pos=True`, indexes the interpolated tensor a second time with the LINK indices (:109 after :147), so it is
    driven here the way that makes it evaluate what it evidently means: the field is built over `arange(K)` columns and fed
    `interpolate_points_v1(link_pos[..., link_idxs, :], K)` -- the second indexing is then the identity.  What the flag itself
    does with the Panda's indices is recorded as `flag_result`.  (3) The single-link self distance (:195-198)."""
    import itertools
    from torch_robotics.robots.robot_panda import RobotPanda
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    from torch_robotics.torch_planning_objectives.fields.distance_fields import (
        interpolate_points_v1, CollisionObjectDistanceField, CollisionSelfField, CollisionWorkspaceBoundariesDistanceField)
    out = {}
    gen = torch.Generator().manual_seed(4242)
    shapes = [(5, 5), (5, 15), (5, 30), (8, 16), (1, 4), (7, 1), (3, 100), (11, 64)]
    out["ip_shapes"] = np.asarray(shapes, np.int32)
    for L, K in shapes:
        pts = torch.randn(6, L, 3, generator=gen).requires_grad_(True)
        w = torch.randn(6, K, 3, generator=gen)
        o = interpolate_points_v1(pts, K)
        (g,) = torch.autograd.grad((o * w).sum(), pts)
        out[f"ip_{L}_{K}_in"], out[f"ip_{L}_{K}_w"] = pts.detach().numpy(), w.numpy()
        out[f"ip_{L}_{K}_out"], out[f"ip_{L}_{K}_gin"] = o.detach().numpy(), g.numpy()

    robot = quiet(RobotPanda, tensor_args=TA)
    env = EnvSpheres3D(tensor_args=TA)
    cutoff = 0.03
    q0 = sample_q(robot.diff_panda, 64, torch.Generator().manual_seed(2024), 0.1).reshape(8, 8, 7)
    obj_idx = list(robot.link_idxs_for_object_collision_checking)
    self_idx = list(robot.link_idxs_for_self_collision_checking)
    p_obj, p_self = 3, 2
    K_obj, K_self = p_obj * len(obj_idx), p_self * len(self_idx)
    margins_obj = torch.tensor(robot.link_margins_for_object_collision_checking, **TA).repeat_interleave(p_obj)
    names = robot.link_names_for_self_collision_checking
    pairs = []
    for i, l1 in enumerate(names):
        if l1 in robot.link_names_pairs_for_self_collision_checking:
            for l2 in robot.link_names_pairs_for_self_collision_checking[l1]:
                j = names.index(l2)
                pairs.extend([(i * p_self + m, j * p_self + n) for m, n in itertools.product(range(p_self), range(p_self))])
    margins_self = torch.full((len(pairs),), float(robot.self_collision_margin_robot), **TA)
    f_obj = CollisionObjectDistanceField(robot, df_obj_list_fn=env.get_df_obj_list, link_idxs_for_collision_checking=list(range(K_obj)),
                                         num_interpolated_points=K_obj, link_margins_for_object_collision_checking_tensor=margins_obj,
                                         cutoff_margin=cutoff, tensor_args=TA)
    f_ws = CollisionWorkspaceBoundariesDistanceField(robot, ws_bounds_min=env.limits[0], ws_bounds_max=env.limits[1],
                                                     link_idxs_for_collision_checking=list(range(K_obj)), num_interpolated_points=K_obj,
                                                     link_margins_for_object_collision_checking_tensor=margins_obj,
                                                     cutoff_margin=cutoff, tensor_args=TA)
    f_self = CollisionSelfField(robot, link_idxs_for_collision_checking=list(range(K_self)), idxs_links_distance_matrix=pairs,
                                num_interpolated_points=K_self, cutoff_margin=margins_self, tensor_args=TA)
    out.update(q=q0.numpy(), cutoff=np.float32(cutoff), limits=env.limits.numpy(), obj_link_idxs=np.asarray(obj_idx, np.int32),
               self_link_idxs=np.asarray(self_idx, np.int32), K_obj=np.int32(K_obj), K_self=np.int32(K_self),
               obj_margins=margins_obj.numpy(), self_pairs=np.asarray(pairs, np.int32), self_margins=margins_self.numpy())

    def fields_on(link_pos):
        link_pos = link_pos.reshape(-1, link_pos.shape[-2], 3)      # F.interpolate(mode='linear') wants (N, C, L): fold b x h first,
        po = interpolate_points_v1(link_pos[..., obj_idx, :], K_obj)  # as DistanceField.compute_cost does before compute_costs_impl
        ps = interpolate_points_v1(link_pos[..., self_idx, :], K_self)
        return dict(objects=(f_obj, po), ws=(f_ws, po), self=(f_self, ps))

    q = q0.clone().requires_grad_(True)
    pos = robot.fk_map_collision(q)
    pos_leaf = pos.detach().clone().requires_grad_(True)
    total, gq_total = 0.0, 0.0
    for fname in ("self", "objects", "ws"):
        fld, pts = fields_on(pos_leaf)[fname]
        c = fld.compute_cost(q, pts, field_type="sdf")
        (gpos,) = torch.autograd.grad(c.sum(), pos_leaf)
        fld, pts = fields_on(pos)[fname]
        c2 = fld.compute_cost(q, pts, field_type="sdf")
        (gq,) = torch.autograd.grad(c2.sum(), q, retain_graph=True)
        c = c.reshape(8, 8)
        out[f"cost_{fname}"], out[f"gpos_{fname}"], out[f"gq_{fname}"] = c.detach().numpy(), gpos.numpy(), gq.numpy()
        out[f"coll_{fname}"] = fld.compute_cost(q, pts.detach(), field_type="occupancy").reshape(8, 8).numpy()
        out[f"coll0_{fname}"] = fld.compute_cost(q, pts.detach(), field_type="occupancy", margin=0.0).reshape(8, 8).numpy()
        total, gq_total = total + c.detach(), gq_total + gq
    out["cost_total"], out["gq_total"] = total.numpy(), gq_total.numpy()
    out["interp_points_obj"] = interpolate_points_v1(pos.detach().reshape(-1, 11, 3)[..., obj_idx, :], K_obj).numpy()

    # what the reference's own flag does with these link indices (for the record; not a parity target)
    try:
        f_flag = CollisionObjectDistanceField(robot, df_obj_list_fn=env.get_df_obj_list, link_idxs_for_collision_checking=obj_idx,
                                              num_interpolated_points=K_obj, interpolate_link_pos=True,
                                              link_margins_for_object_collision_checking_tensor=margins_obj,
                                              cutoff_margin=cutoff, tensor_args=TA)
        f_flag.compute_cost(q0, pos.detach(), field_type="sdf")
        out["flag_result"] = np.asarray("ran")
    except Exception as e:                                      # margins (K) meet the 5 re-indexed points
        out["flag_result"] = np.asarray(f"{type(e).__name__}: {str(e)[:160]}")

    # (3) one self-collision link: |p|_1 * 1e9 stands in for the distance (distance_fields.py:195-198)
    f_one = CollisionSelfField(robot, link_idxs_for_collision_checking=[7], idxs_links_distance_matrix=[(0, 0)],
                               num_interpolated_points=1, cutoff_margin=0.05, tensor_args=TA)
    q = q0.clone().requires_grad_(True)
    pos = robot.fk_map_collision(q)
    pos_leaf = pos.detach().clone().requires_grad_(True)
    c = f_one.compute_cost(q, pos_leaf, field_type="sdf")
    (gpos,) = torch.autograd.grad(c.sum(), pos_leaf)
    (gq,) = torch.autograd.grad(f_one.compute_cost(q, pos, field_type="sdf").sum(), q)
    out["single_link"], out["single_margin"] = np.int32(7), np.float32(0.05)
    out["single_cost"], out["single_gpos"], out["single_gq"] = c.detach().numpy(), gpos.numpy(), gq.numpy()
    out["single_coll"] = f_one.compute_cost(q, pos.detach(), field_type="occupancy").numpy()
    np.savez_compressed(GOLD / "cost_interp.npz", **out)
    print("cost_interp: flag_result =", out["flag_result"])


def tree_cost_goldens(trees):
    """`python oracle/gen_golden.py treecost`.  BASELINE configs 4 and 5 pinned to the reference: the authored UR10 + Allegro and
    dual-Panda robots have no Robot class in the reference, so its OWN pieces are assembled the way RobotBase / PlanningTask
    assemble them for the Panda (robot_base.py:57-141, tasks.py:44-80): DifferentiableTree FK (all links) -> link positions ->
    CollisionSelfField / CollisionObjectDistanceField / CollisionWorkspaceBoundariesDistanceField on the link sets of the build's
    collision templates (torch_robotics_amd/codegen.py: ur10_allegro_template, dual_panda_template -- data, passed in by index
    lists) + one EESE3DistanceField per tracked link; gradients by the reference's autograd through its FK recursion."""
    from types import SimpleNamespace
    from torch_robotics.environments.env_spheres_3d import EnvSpheres3D
    from torch_robotics.torch_planning_objectives.fields.distance_fields import (
        CollisionObjectDistanceField, CollisionSelfField, CollisionWorkspaceBoundariesDistanceField, EESE3DistanceField)
    from torch_robotics.torch_kinematics_tree.geometrics.utils import link_pos_from_link_tensor
    sys.path.insert(0, str(REPO))
    from torch_robotics_amd import codegen                      # the collision templates are DATA of the build (index lists)
    from torch_robotics_amd.kinmodel import KinModel
    env = EnvSpheres3D(tensor_args=TA)
    cutoff, self_margin, obj_margin = 0.03, 0.04, 0.07
    for name, tmpl_fn, n in (("ur10_allegro", codegen.ur10_allegro_template, 24), ("dual_panda", codegen.dual_panda_template, 24)):
        tree = trees[name]
        kin = KinModel.from_urdf(str(URDF_OUT / f"{name}.urdf"))
        tm = tmpl_fn(kin)
        assert kin.link_names == tree.get_link_names()
        robot = SimpleNamespace(grasped_object=None)             # what the field code reads of its robot (distance_fields.py:137)
        obj_idx = list(tm.obj_links)
        margins = torch.full((len(obj_idx),), obj_margin, **TA)
        self_links = sorted({a for p in tm.self_pairs for a in p})
        pairs = [(self_links.index(a), self_links.index(b)) for a, b in tm.self_pairs]
        f_obj = CollisionObjectDistanceField(robot, df_obj_list_fn=env.get_df_obj_list, link_idxs_for_collision_checking=obj_idx,
                                             num_interpolated_points=len(obj_idx), link_margins_for_object_collision_checking_tensor=margins,
                                             cutoff_margin=cutoff, tensor_args=TA)
        f_ws = CollisionWorkspaceBoundariesDistanceField(robot, ws_bounds_min=env.limits[0], ws_bounds_max=env.limits[1],
                                                         link_idxs_for_collision_checking=obj_idx, num_interpolated_points=len(obj_idx),
                                                         link_margins_for_object_collision_checking_tensor=margins,
                                                         cutoff_margin=cutoff, tensor_args=TA)
        f_self = CollisionSelfField(robot, link_idxs_for_collision_checking=self_links, idxs_links_distance_matrix=pairs,
                                    num_interpolated_points=len(self_links),
                                    cutoff_margin=torch.full((len(pairs),), self_margin, **TA), tensor_args=TA)
        gen = torch.Generator().manual_seed(808)
        q0 = sample_q(tree, n, gen, 0.1)
        # EE targets: poses the tracked links reach for another configuration, pushed a little
        q_t = sample_q(tree, 1, gen, 0.0)
        H_t = tree.compute_forward_kinematics_all_links(q_t)[0]
        tracked = [l for l in (tm.ee_link, tm.ee2_link) if l >= 0]
        targets = []
        for l in tracked:
            Ht = H_t[l].clone(); Ht[:3, 3] += torch.tensor([0.05, -0.03, 0.04])
            targets.append(Ht)
        f_ee = [EESE3DistanceField(Ht, w_pos=1.0, w_rot=1.0, square=True, tensor_args=TA) for Ht in targets]
        out = dict(q=q0.numpy(), cutoff=np.float32(cutoff), limits=env.limits.numpy(), obj_link_idxs=np.asarray(obj_idx, np.int32),
                   obj_margins=margins.numpy(), self_link_idxs=np.asarray(self_links, np.int32), self_pairs=np.asarray(pairs, np.int32),
                   self_margins=np.full(len(pairs), self_margin, np.float32), ee_links=np.asarray(tracked, np.int32),
                   ee_targets=torch.stack(targets).numpy())

        def parts(q):
            H = tree.compute_forward_kinematics_all_links(q)
            pos = link_pos_from_link_tensor(H)
            c = dict(self=f_self.compute_cost(q, pos, field_type="sdf").reshape(-1),
                     objects=f_obj.compute_cost(q, pos, field_type="sdf").reshape(-1),
                     ws=f_ws.compute_cost(q, pos, field_type="sdf").reshape(-1))
            ee = 0.0
            for l, f in zip(tracked, f_ee):
                ee = ee + f.compute_costs_impl(q, H[:, l:l + 1]).reshape(-1)
            c["ee"] = ee
            return H, pos, c
        for fname in ("self", "objects", "ws", "ee"):
            q = q0.clone().requires_grad_(True)
            _, _, c = parts(q)
            (gq,) = torch.autograd.grad(c[fname].sum(), q)
            out[f"cost_{fname}"], out[f"gq_{fname}"] = c[fname].detach().numpy(), gq.numpy()
        q = q0.clone().requires_grad_(True)
        H, pos, c = parts(q)
        total = c["self"] + c["objects"] + c["ws"] + c["ee"]
        (gq,) = torch.autograd.grad(total.sum(), q)
        out["cost_total"], out["gq_total"], out["link_pos"] = total.detach().numpy(), gq.numpy(), pos.detach().numpy()
        for fname, f in (("self", f_self), ("objects", f_obj), ("ws", f_ws)):
            out[f"coll_{fname}"] = f.compute_cost(q0, pos.detach(), field_type="occupancy").reshape(-1).numpy()
            out[f"coll0_{fname}"] = f.compute_cost(q0, pos.detach(), field_type="occupancy", margin=0.0).reshape(-1).numpy()
        np.savez_compressed(GOLD / f"cost_tree_{name}.npz", **out)
        print(f"cost_tree_{name}: {n} samples, {len(obj_idx)} collision links, {len(pairs)} pairs, {len(tracked)} tracked")


def builddef_goldens(trees=None):
    """SECOND, independent statements of the two BUILD-DEFINED terms (no reference counterpart: SURVEY.md 8a), so that the C oracle
    and the HIP kernels are not checked against a single restatement by the same author:

    * gp_prior.npz -- the constant-velocity GP prior from its DENSE definition in torch fp64: state x_t = [q_t; qd_t], transition
      Phi(dt) = [[I, dt I], [0, I]], process noise Q = sigma^2 [[dt^3/3 I, dt^2/2 I], [dt^2/2 I, dt I]] (white noise on the
      acceleration), Q^-1 by `torch.linalg.inv` of the 2D x 2D matrix (NOT the closed form 12/dt^3, -6/dt^2, 4/dt the kernels
      use), cost = w/2 sum_t e_t^T Q^-1 e_t with e_t = Phi x_t - x_t+1; gradients by autograd.
    * ik_gn_panda.npz -- one damped Gauss-Newton step built from the REFERENCE's own stateful FK + geometric Jacobian
      (compute_forward_kinematics_and_geometric_jacobian, robot_tree.py:218-248, fp32) with `torch.linalg.solve` in fp64 and
      scipy's rotation vector; the termination metric from the reference's SE3_distance (geometrics/utils.py:130-154)."""
    from scipy.spatial.transform import Rotation
    out = {}
    gen = torch.Generator().manual_seed(515)
    cases = [(3, 16, 7, 0.08, 0.3, 1.0), (2, 128, 14, 5.0 / 128, 0.1, 1.0), (4, 5, 3, 0.25, 1.5, 0.5), (2, 2, 7, 0.1, 0.2, 2.0), (3, 1, 7, 0.1, 0.2, 1.0)]
    out["n_cases"] = np.int32(len(cases))
    for k, (B, H, D, dt, sigma, w) in enumerate(cases):
        q32 = (torch.rand(B, 1, D, generator=gen) * 2 - 1 + torch.cumsum(torch.randn(B, H, D, generator=gen) * 0.03, 1)).float()
        qd32 = (torch.randn(B, H, D, generator=gen) * 0.3).float()
        q = q32.double().requires_grad_(True)
        qd = qd32.double().requires_grad_(True)
        I, Z = torch.eye(D, dtype=torch.float64), torch.zeros(D, D, dtype=torch.float64)
        Phi = torch.cat([torch.cat([I, dt * I], 1), torch.cat([Z, I], 1)], 0)
        Q = sigma ** 2 * torch.cat([torch.cat([dt ** 3 / 3 * I, dt ** 2 / 2 * I], 1), torch.cat([dt ** 2 / 2 * I, dt * I], 1)], 0)
        Qinv = torch.linalg.inv(Q)
        x = torch.cat([q, qd], -1)                                       # (B, H, 2D)
        if H > 1:
            e = x[:, :-1] @ Phi.T - x[:, 1:]                             # (B, H-1, 2D)
            fac = 0.5 * w * torch.einsum("bti,ij,btj->bt", e, Qinv, e)
            fac = torch.cat([fac, torch.zeros(B, 1, dtype=torch.float64)], 1)
        else:
            fac = torch.zeros(B, 1, dtype=torch.float64) + 0.0 * x.sum()
        cost = fac.sum(1)
        gq, gqd = torch.autograd.grad(cost.sum(), (q, qd), allow_unused=True)
        gq = torch.zeros_like(q) if gq is None else gq
        gqd = torch.zeros_like(qd) if gqd is None else gqd
        out.update({f"q_{k}": q32.numpy(), f"qd_{k}": qd32.numpy(), f"params_{k}": np.array([dt, sigma, w], np.float64),
                    f"cost_{k}": cost.detach().numpy(), f"factor_{k}": fac.detach().numpy(), f"gq_{k}": gq.numpy(), f"gqd_{k}": gqd.numpy()})
    np.savez_compressed(GOLD / "gp_prior.npz", **out)

    if trees is None:
        trees = {"panda_arm_no_gripper": quiet(DifferentiableTree, str(URDF_OUT / "panda_arm_no_gripper.urdf"), "panda_arm_no_gripper")}
    tree = trees["panda_arm_no_gripper"]
    gen = torch.Generator().manual_seed(616)
    lower, upper, _, _ = tree.get_joint_limit_array()
    eps = np.pi / 100
    lo, hi = (lower + eps).astype(np.float32), (upper - eps).astype(np.float32)
    n = 40
    q0 = sample_q(tree, n, gen, 0.0)
    q0[:8] = sample_q(tree, 8, gen, 0.1)                                # a few start outside the (shrunk) limits: the stateful FK clamps
    q_goal = sample_q(tree, n, gen, 0.0)
    H_target = tree.compute_forward_kinematics_all_links(q_goal, link_list=["ee_link"]).squeeze(1).detach()
    out = dict(q0=q0.numpy(), H_target=H_target.numpy(), lower=lo, upper=hi, link="ee_link")
    for tag, (damping, lm_gain, step_scale) in (("a", (1e-4, 0.1, 1.0)), ("b", (1e-3, 0.02, 0.7))):
        tree.reset()
        pos, quat, lin, ang = tree.compute_forward_kinematics_and_geometric_jacobian(q0, torch.zeros_like(q0), "ee_link")
        tree.reset()
        J = torch.cat([lin, ang], 1).double()                            # (n, 6, D): the reference's geometric Jacobian
        R = torch.as_tensor(Rotation.from_quat(quat.numpy()[:, [1, 2, 3, 0]].astype(np.float64)).as_matrix())     # wxyz -> scipy's xyzw
        Rt = H_target[:, :3, :3].double()
        rot = Rotation.from_matrix((Rt @ R.transpose(1, 2)).numpy()).as_rotvec()
        r = torch.cat([H_target[:, :3, 3].double() - pos.double(), torch.as_tensor(rot)], 1)       # (n, 6)
        lam = damping + lm_gain * (r * r).sum(1)
        A = J.transpose(1, 2) @ J + lam[:, None, None] * torch.eye(J.shape[-1], dtype=torch.float64)
        dq = torch.linalg.solve(A, (J.transpose(1, 2) @ r[:, :, None])).squeeze(-1)
        q_new = torch.minimum(torch.maximum(q0.double() + step_scale * dq, torch.as_tensor(lo).double()), torch.as_tensor(hi).double())
        Hq = torch.eye(4).repeat(n, 1, 1)
        Hq[:, :3, :3] = R.float(); Hq[:, :3, 3] = pos
        err = SE3_distance(Hq, H_target, w_pos=1.0, w_rot=1.0)
        out.update({f"params_{tag}": np.array([damping, lm_gain, step_scale], np.float64), f"q_new_{tag}": q_new.numpy(), f"dq_{tag}": dq.numpy(),
                    f"err_{tag}": err.reshape(n).numpy(), f"residual_{tag}": r.numpy()})
    np.savez_compressed(GOLD / "ik_gn_panda.npz", **out)


def main():
    GOLD.mkdir(parents=True, exist_ok=True)
    URDF_OUT.mkdir(parents=True, exist_ok=True)
    if sys.argv[1:] == ["points"]:
        points_goldens()
        return
    if sys.argv[1:] == ["pointmass"]:
        pointmass_goldens()
        return
    if sys.argv[1:] == ["metrics"]:
        metrics_goldens()
        return
    if sys.argv[1:] == ["clamp"]:
        clamp_goldens()
        return
    if sys.argv[1:] == ["frames"]:
        frame_goldens()
        return
    if sys.argv[1:] == ["spheres"]:
        sphere_config_data()
        return
    if sys.argv[1:] == ["interp"]:
        interp_goldens()
        return
    if sys.argv[1:] == ["builddef"]:
        builddef_goldens()
        return
    if sys.argv[1:] == ["treecost"]:
        tree_cost_goldens({n: quiet(DifferentiableTree, str(URDF_OUT / f"{n}.urdf"), n) for n in ("ur10_allegro", "dual_panda")})
        return
    trees = {}
    for name, rel in ROBOTS.items():
        strip_urdf(REF_URDF / rel, URDF_OUT / f"{name}.urdf")
        t_orig = quiet(DifferentiableTree, str(REF_URDF / rel), name)
        t_strip = quiet(DifferentiableTree, str(URDF_OUT / f"{name}.urdf"), name)
        q = sample_q(t_orig, 8, torch.Generator().manual_seed(0), 0.1)
        assert torch.equal(t_orig.compute_forward_kinematics_all_links(q),
                           t_strip.compute_forward_kinematics_all_links(q)), name
        trees[name] = t_strip
    for name, fn in AUTHORED.items():
        fn(URDF_OUT / f"{name}.urdf")
        trees[name] = quiet(DifferentiableTree, str(URDF_OUT / f"{name}.urdf"), name)
    for k, (name, tree) in enumerate(sorted(trees.items())):
        out = fk_golden(name, tree, 100 + k)
        print(f"fk_{name}: L={len(out['link_names'])} D={int(out['n_dofs'])}")
    jacobian_golden("panda_arm_no_gripper", trees["panda_arm_no_gripper"], ["ee_link", "panda_link5", "panda_link2"], 31)
    jacobian_golden("ur10", trees["ur10"], [trees["ur10"].get_link_names()[-1], trees["ur10"].get_link_names()[4]], 32)
    jacobian_golden("iiwa7", trees["iiwa7"], [trees["iiwa7"].get_link_names()[-1]], 33)
    analytic_jacobian_golden("panda_arm_no_gripper", trees["panda_arm_no_gripper"], 41)
    analytic_jacobian_golden("panda_arm_hand", trees["panda_arm_hand"], 42)
    analytic_jacobian_golden("allegro_hand", trees["allegro_hand"], 43)
    quat_golden()
    cost_goldens()
    trajs_goldens()
    ik_goldens(trees)
    misc_goldens()
    points_goldens()
    sphere_config_data()
    metrics_goldens()
    pointmass_goldens()
    frame_goldens()
    clamp_goldens()
    interp_goldens()
    tree_cost_goldens(trees)
    builddef_goldens(trees)
    total = sum(p.stat().st_size for p in GOLD.glob("*.npz"))
    print(f"golden dir: {len(list(GOLD.glob('*.npz')))} files, {total/1024:.0f} kB")


if __name__ == "__main__":
    main()
