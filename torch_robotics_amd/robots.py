"""Drop-in for `torch_robotics.robots` on the hot path: `RobotBase` (robot_base.py:13-188) data and
`RobotPanda` (robot_panda.py:21-184).  `fk_map_collision` is one `trk_fk_positions` launch."""
from __future__ import annotations

import itertools
from collections import OrderedDict
from math import ceil

import numpy as np
import torch

from . import ops
from .costmodel import link_sorted_point_set, load_link_spheres
from .environments import DEFAULT_TENSOR_ARGS
from .fields import CollisionSelfField
from .kinematics import DATA_DIR, DifferentiableFrankaPanda, link_pos_from_link_tensor, link_quat_from_link_tensor, \
    link_rot_from_link_tensor


def finite_difference_vector(x, dt=1.0, method="forward"):
    """trajectory/utils.py:53-64 (zero-padded finite differences along the horizon), one streaming kernel."""
    return ops.finite_difference(x, dt=dt, method=method)


def compute_path_length(trajs, robot):
    """trajectory/metrics.py:7-12: sum over the horizon of the joint-space step lengths, (B, H, S) -> (B,)."""
    assert trajs.ndim == 3
    return ops.traj_diff_norm_sum(trajs, 0, robot.q_dim)


def compute_smoothness(trajs, robot, trajs_vel=None):
    """trajectory/metrics.py:27-35: sum_t || v[t+1] - v[t] ||; velocities from the state, else by central differences."""
    if trajs_vel is None:
        assert trajs.ndim == 3
        if trajs.shape[-1] >= 2 * robot.q_dim:                       # velocities are carried in the state: read them in place
            return ops.traj_diff_norm_sum(trajs, robot.q_dim, robot.q_dim)
        trajs_vel = robot.get_velocity(trajs)
    else:
        assert trajs_vel.ndim == 3
    return ops.traj_diff_norm_sum(trajs_vel, 0, trajs_vel.shape[-1])


class RobotBase:
    def __init__(self, name="RobotBase", q_limits=None, grasped_object=None,
                 margin_for_grasped_object_collision_checking=0.001,
                 link_names_for_object_collision_checking=None, link_margins_for_object_collision_checking=None,
                 link_idxs_for_object_collision_checking=None, link_names_for_self_collision_checking=None,
                 link_names_pairs_for_self_collision_checking=None, link_idxs_for_self_collision_checking=None,
                 self_collision_margin_robot=0.001, link_names_for_self_collision_checking_with_grasped_object=None,
                 self_collision_margin_grasped_object=0.05, num_interpolated_points_for_self_collision_checking=1,
                 num_interpolated_points_for_object_collision_checking=1, dt=1.0, tensor_args=None, **kwargs):
        self.name = name
        self.tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args
        self.dt = dt
        assert q_limits is not None, "q_limits cannot be None"
        self.q_limits = q_limits
        self.q_min, self.q_max = q_limits[0], q_limits[1]
        self.q_min_np, self.q_max_np = self.q_min.cpu().numpy(), self.q_max.cpu().numpy()
        self.q_dim = len(self.q_min)
        self.grasped_object = grasped_object
        self.margin_for_grasped_object_collision_checking = margin_for_grasped_object_collision_checking

        # objects collision field (robot_base.py:57-82)
        n_obj = len(link_names_for_object_collision_checking)
        assert num_interpolated_points_for_object_collision_checking >= n_obj
        if num_interpolated_points_for_object_collision_checking % n_obj != 0:
            self.points_per_link_object_collision_checking = ceil(num_interpolated_points_for_object_collision_checking / n_obj)
            num_interpolated_points_for_object_collision_checking = self.points_per_link_object_collision_checking * n_obj
        else:
            self.points_per_link_object_collision_checking = int(num_interpolated_points_for_object_collision_checking / n_obj)
        self.self_collision_margin_robot = self_collision_margin_robot
        self.num_interpolated_points_for_object_collision_checking = num_interpolated_points_for_object_collision_checking
        self.link_names_for_object_collision_checking = link_names_for_object_collision_checking
        self.n_links_for_object_collision_checking = n_obj
        self.link_margins_for_object_collision_checking = link_margins_for_object_collision_checking
        self.link_margins_for_object_collision_checking_robot_tensor = torch.tensor(
            link_margins_for_object_collision_checking, dtype=torch.float32).repeat_interleave(
            int(num_interpolated_points_for_object_collision_checking / len(link_margins_for_object_collision_checking)))
        self.link_margins_for_object_collision_checking_tensor = self.link_margins_for_object_collision_checking_robot_tensor
        if self.grasped_object is not None:                     # robot_base.py:76-80: the grasped points' margins follow
            self.link_margins_for_object_collision_checking_tensor = torch.cat((
                self.link_margins_for_object_collision_checking_tensor,
                torch.ones(self.grasped_object.n_base_points_for_collision, dtype=torch.float32)
                * self.margin_for_grasped_object_collision_checking))
        self.link_idxs_for_object_collision_checking = link_idxs_for_object_collision_checking

        # self collision field: pair index table (robot_base.py:84-141)
        if link_names_for_self_collision_checking is None:
            self.df_collision_self = None
        else:
            n_self = len(link_names_for_self_collision_checking)
            assert num_interpolated_points_for_self_collision_checking >= n_self
            if num_interpolated_points_for_self_collision_checking % n_self != 0:
                self.points_per_link_self_collision_checking = ceil(num_interpolated_points_for_self_collision_checking / n_self)
                num_interpolated_points_for_self_collision_checking = self.points_per_link_self_collision_checking * n_self
            else:
                self.points_per_link_self_collision_checking = int(num_interpolated_points_for_self_collision_checking / n_self)
            self.link_names_for_self_collision_checking = link_names_for_self_collision_checking
            self.link_names_pairs_for_self_collision_checking = link_names_pairs_for_self_collision_checking
            self.link_idxs_for_self_collision_checking = link_idxs_for_self_collision_checking
            self.link_names_for_self_collision_checking_with_grasped_object = link_names_for_self_collision_checking_with_grasped_object
            self.self_collision_margin_grasped_object = self_collision_margin_grasped_object
            idxs, p = [], self.points_per_link_self_collision_checking
            for i, link_1 in enumerate(link_names_for_self_collision_checking):
                if link_1 in link_names_pairs_for_self_collision_checking:
                    for link_2 in link_names_pairs_for_self_collision_checking[link_1]:
                        j = link_names_for_self_collision_checking.index(link_2)
                        idxs.extend([(i * p + m, j * p + n) for m, n in itertools.product(range(p), range(p))])
            margin_list = [self.self_collision_margin_robot] * len(idxs)
            if self.grasped_object is not None:                 # robot_base.py:120-130: grasped points x listed links
                last_row = n_self * p
                n_grasped = self.grasped_object.n_base_points_for_collision
                n_before = len(idxs)
                for link_1 in link_names_for_self_collision_checking_with_grasped_object:
                    j = link_names_for_self_collision_checking.index(link_1)
                    idxs.extend([(last_row + m, j * p + n) for m, n in itertools.product(range(n_grasped), range(p))])
                margin_list.extend([self.self_collision_margin_grasped_object] * (len(idxs) - n_before))
            margins = torch.tensor(margin_list, dtype=torch.float32)
            # more points than links: the pair rows above index INTERPOLATED points, so the field has to interpolate (the
            # reference leaves the flag off and its pair rows would then index past the link tensor)
            self.df_collision_self = CollisionSelfField(
                self, link_idxs_for_collision_checking=self.link_idxs_for_self_collision_checking,
                idxs_links_distance_matrix=idxs,
                num_interpolated_points=num_interpolated_points_for_self_collision_checking,
                interpolate_link_pos=num_interpolated_points_for_self_collision_checking != n_self,
                cutoff_margin=margins, tensor_args=self.tensor_args)

    def random_q(self, n_samples=10, generator=None):
        """Uniform in the joint limits (robot_base.py:143-146), drawn on the robot's device."""
        dev = self.tensor_args["device"]
        lo, hi = self.q_min.to(dev), self.q_max.to(dev)
        u = torch.rand((n_samples, self.q_dim), device=dev, dtype=torch.float32, generator=generator)
        return lo + u * (hi - lo)

    def get_position(self, x):
        return x[..., :self.q_dim]

    def get_velocity(self, x):
        vel = x[..., self.q_dim:2 * self.q_dim]
        if x.nelement() != 0 and vel.nelement() == 0:
            return finite_difference_vector(x, dt=self.dt, method="central")
        return vel

    def get_acceleration(self, x):
        acc = x[..., 2 * self.q_dim:3 * self.q_dim]
        if x.nelement() != 0 and acc.nelement() == 0:
            return finite_difference_vector(self.get_velocity(x), dt=self.dt, method="central")
        return acc

    def distance_q(self, q1, q2):
        return torch.linalg.norm(q1 - q2, dim=-1)

    @ops.host_round_trip
    def fk_map_collision(self, q, **kwargs):                   # robot_base.py:171-174
        if q.ndim == 1:
            q = q.unsqueeze(0)
        return self.fk_map_collision_impl(q, **kwargs)

    def fk_map_collision_impl(self, q, **kwargs):
        raise NotImplementedError


class RobotPointMass(RobotBase):                               # robot_point_mass.py:13-32
    """A point in configuration space == task space: no kinematics, one collision point (margin 0.01)."""

    def __init__(self, name="RobotPointMass", q_limits=((-1, -1), (1, 1)), tensor_args=None, **kwargs):
        tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args
        super().__init__(name=name, q_limits=torch.as_tensor(q_limits, dtype=torch.float32).to(tensor_args["device"]),
                         link_names_for_object_collision_checking=["link_0"], link_margins_for_object_collision_checking=[0.01],
                         link_idxs_for_object_collision_checking=[0], num_interpolated_points_for_object_collision_checking=1,
                         tensor_args=tensor_args, **kwargs)

    def fk_map_collision_impl(self, q, **kwargs):
        return q.unsqueeze(-2)                                 # identity "kinematics": add the link dimension


class RobotPointMass3D(RobotPointMass):                        # robot_point_mass.py:101-109
    def __init__(self, tensor_args=None, **kwargs):
        super().__init__(name="RobotPointMass3D", q_limits=((-1, -1, -1), (1, 1, 1)), tensor_args=tensor_args, **kwargs)


class RobotPanda(RobotBase):                                   # robot_panda.py:21-184
    """`link_sphere_model` (an extension) replaces the five link-origin collision points by the link-frame
    spheres of a table such as data/configs/panda_sphere_config.yaml ("panda" = that file): fk_map_collision then
    returns, link by link in walk order, the link origin followed by that link's sphere centres (L + S columns; a grasped
    object's points still come last), and the object / workspace fields read the sphere columns with the radii as
    margins.  Self-collision keeps using the link origins (their columns)."""

    def __init__(self, use_self_collision_storm=False, grasped_object=None, tensor_args=None, link_sphere_model=None,
                 num_interpolated_points_for_object_collision_checking=None,
                 num_interpolated_points_for_self_collision_checking=None, **kwargs):
        # num_interpolated_points_for_*: the reference's RobotPanda fixes both to the number of links (robot_panda.py:117,121);
        # RobotBase's layout for more points per link (robot_base.py:57-73, 103-108) is reachable here by passing them
        tensor_args = DEFAULT_TENSOR_ARGS if tensor_args is None else tensor_args
        if use_self_collision_storm:
            raise NotImplementedError("the STORM self-collision network needs storm_kit weights (out of scope)")
        self.gripper = False
        self.link_name_ee = "ee_link"
        self.link_name_grasped_object = "grasped_object"
        self.diff_panda = DifferentiableFrankaPanda(gripper=self.gripper, device=tensor_args["device"],
                                                    grasped_object=grasped_object)
        self.jl_lower, self.jl_upper, _, _ = self.diff_panda.get_joint_limit_array()
        q_limits = torch.tensor(np.array([self.jl_lower, self.jl_upper]), **tensor_args)
        obj_links = ["panda_link2", "panda_link3", "panda_link5", "panda_link7", "panda_hand"]
        obj_margins = [0.125, 0.125, 0.13, 0.1, 0.08]
        obj_idxs = [self.diff_panda._name_to_idx_map[n] for n in obj_links]
        pairs = OrderedDict({"panda_link4": ["panda_link1"],
                             "panda_link5": ["panda_link0", "panda_link1", "panda_link2"],
                             "panda_link6": ["panda_link0", "panda_link1", "panda_link2"],
                             "panda_hand": ["panda_link0", "panda_link1", "panda_link2"]})
        with_grasped = ["panda_link0", "panda_link1", "panda_link2", "panda_link3"]
        self_links = []
        for k, v in pairs.items():
            self_links.append(k)
            self_links.extend(v)
        self_links.extend(with_grasped)
        self_links = sorted(list(set(self_links)))
        self_idxs = [self.diff_panda._name_to_idx_map[n] for n in self_links]
        self.link_spheres = None
        kin = self.diff_panda._kin
        self._base_points = link_sorted_point_set(kin.order)[:2]
        if link_sphere_model is not None:
            path = DATA_DIR / "configs" / "panda_sphere_config.yaml" if link_sphere_model == "panda" else link_sphere_model
            sl, so, sr, names = load_link_spheres(path, self.diff_panda._name_to_idx_map)
            self.link_spheres = (sl, so, sr)
            pl, po, origin_col, sphere_col = link_sorted_point_set(kin.order, sl, so)
            self._base_points = (pl, po)
            obj_links, obj_margins, obj_idxs = names, [float(r) for r in sr], [int(c) for c in sphere_col]
            self_idxs = [int(origin_col[i]) for i in self_idxs]       # link origins moved to their new columns
        super().__init__(
            name="RobotPanda", q_limits=q_limits, grasped_object=grasped_object,
            link_names_for_object_collision_checking=obj_links, link_margins_for_object_collision_checking=obj_margins,
            link_idxs_for_object_collision_checking=obj_idxs, margin_for_grasped_object_collision_checking=0.001,
            num_interpolated_points_for_object_collision_checking=(num_interpolated_points_for_object_collision_checking
                                                                   or len(obj_links)),
            link_names_for_self_collision_checking=self_links, link_names_pairs_for_self_collision_checking=pairs,
            link_idxs_for_self_collision_checking=self_idxs,
            num_interpolated_points_for_self_collision_checking=(num_interpolated_points_for_self_collision_checking
                                                                 or len(self_links)), self_collision_margin_robot=0.05,
            link_names_for_self_collision_checking_with_grasped_object=with_grasped,
            self_collision_margin_grasped_object=0.05, tensor_args=tensor_args, **kwargs)

    def collision_point_set(self):
        """(point_link, point_offset) of what fk_map_collision returns: every link origin (with a link-sphere model: each
        followed by its spheres), then the grasped object's collision points in the `grasped_object` link frame
        (robot_panda.py:154-168)."""
        link, off = [self._base_points[0]], [self._base_points[1]]
        if self.grasped_object is not None:
            pts = self.grasped_object.base_points_for_collision.detach().cpu().numpy().astype(np.float32)
            link.append(np.full(len(pts), self.diff_panda._name_to_idx_map[self.link_name_grasped_object], np.int32))
            off.append(pts)
        return np.concatenate(link).astype(np.int32), np.concatenate(off).astype(np.float32)

    @property
    def has_extra_points(self) -> bool:
        return self.grasped_object is not None or self.link_spheres is not None

    def _point_set(self, device) -> "ops.PointSetHandle":
        key = str(device)
        if getattr(self, "_ps", None) is None or self._ps[0] != key:
            self._ps = (key, ops.PointSetHandle(self.diff_panda._handle, *self.collision_point_set(), device))
        return self._ps[1]

    def fk_map_collision_impl(self, q, **kwargs):              # robot_panda.py:138-170
        shape = q.shape
        if len(shape) not in (2, 3):
            raise NotImplementedError
        if not self.has_extra_points:
            pos = ops.fk_pos(self.diff_panda._handle, q)        # all L link origins, (N, L, 3)
        else:
            pos = ops.fk_points_ad(self._point_set(q.device), q)    # (N, L + G, 3), one launch
        return pos.reshape(tuple(shape[:-1]) + (pos.shape[-2], 3))

    @ops.host_round_trip
    def get_EE_pose(self, q):
        return self.diff_panda.compute_forward_kinematics_all_links(q, link_list=[self.link_name_ee])

    @ops.host_round_trip
    def get_EE_position(self, q):
        return link_pos_from_link_tensor(self.get_EE_pose(q))

    @ops.host_round_trip
    def get_EE_orientation(self, q, rotation_matrix=True):
        ee = self.get_EE_pose(q)
        return link_rot_from_link_tensor(ee) if rotation_matrix else link_quat_from_link_tensor(ee)
