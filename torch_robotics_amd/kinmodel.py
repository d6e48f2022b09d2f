"""Host-side model compiler: URDF -> flat `KinModel` arrays for the HIP kernels.

Restates the reference's model build (reference:
torch_kinematics_tree/models/utils.py:199-313 `URDFRobotModel.get_body_parameters`,
robot_tree.py:77-126 `DifferentiableTree.__init__`, rigid_body.py:74-123
`DifferentiableRigidBody.__init__`) as data instead of an object graph:

* link order  = order of `<link>` elements in the file; link 0 is the root;
* DOF order   = file order of the links whose parent joint is not `fixed`;
* per link    = the joint whose child it is: `trans`, `rpy`,
  `R_fixed = Rz(yaw) @ Ry(pitch) @ Rx(roll)` (fp32), raw `axis`, type, limits;
* stateless FK picks the rotation axis as the first of x,y,z with |axis_k| == 1,
  else z, and rotates by `sign(axis_k) * q` (rigid_body.py:162-168);
* the stateful path picks x/y only when axis_k == +1 exactly, ignores the
  sign, and clamps whenever limits exist (rigid_body.py:101-106, 214-233).

The arrays are consumed by the C ABI (`include/trk.h: TrkKinModelDesc`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .urdf import UrdfModel, parse_urdf

# joint type codes shared with include/trk.h
JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_CONTINUOUS = 2
JOINT_PRISMATIC = 3
JOINT_UNSUPPORTED = 4   # floating / planar / ...: FK raises NotImplementedError like the reference

_TYPE_CODE = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE,
              "continuous": JOINT_CONTINUOUS, "prismatic": JOINT_PRISMATIC}

MAX_POSE_SLOTS = 8   # pose stack depth the generic kernels support (branch nesting)


def _axis_rot(kind: str, angle: torch.Tensor) -> torch.Tensor:
    """fp32 axis rotation built the way spatial_vector.py:8-47 builds it."""
    c, s = torch.cos(angle), torch.sin(angle)
    R = torch.zeros(3, 3, dtype=torch.float32)
    if kind == "x":
        R[0, 0] = 1.0; R[1, 1] = c; R[1, 2] = -s; R[2, 1] = s; R[2, 2] = c
    elif kind == "y":
        R[0, 0] = c; R[0, 2] = s; R[1, 1] = 1.0; R[2, 0] = -s; R[2, 2] = c
    else:
        R[0, 0] = c; R[0, 1] = -s; R[1, 0] = s; R[1, 1] = c; R[2, 2] = 1.0
    return R


def fixed_rotation_from_rpy(rpy) -> np.ndarray:
    """`(z_rot(yaw) @ y_rot(pitch)) @ x_rot(roll)` in fp32 (rigid_body.py:89-93).

    torch is used (not numpy) so that sin/cos/matmul round exactly as in the
    reference's one-off model build."""
    ang = torch.tensor(list(rpy), dtype=torch.float32)
    R = (_axis_rot("z", ang[2]) @ _axis_rot("y", ang[1])) @ _axis_rot("x", ang[0])
    return R.numpy().copy()


def quat_wxyz_to_rot(q) -> np.ndarray:
    """Reference quaternion.py:102-120 `q_to_rotation_matrix` (fp32)."""
    q = torch.tensor(np.asarray(q, np.float32).reshape(4))      # a copy: the caller's array may be read-only (ObjectField poses)
    w, x, y, z = q.unbind(-1)
    dc = 2.0 / (q ** 2).sum(-1)
    o = torch.stack((1 - dc * (y * y + z * z), dc * (x * y - z * w), dc * (x * z + y * w),
                     dc * (x * y + z * w), 1 - dc * (x * x + z * z), dc * (y * z - x * w),
                     dc * (x * z - y * w), dc * (y * z + x * w), 1 - dc * (x * x + y * y)))
    return o.reshape(3, 3).numpy().copy()


@dataclass
class KinModel:
    name: str
    link_names: List[str]
    joint_names: List[str]                 # per link: name of its parent joint ("base_joint" for the root)
    n_links: int
    n_dofs: int
    parent: np.ndarray                     # int32[L], -1 for the root (file-order indices)
    joint_type: np.ndarray                 # int32[L], JOINT_*
    dof_idx: np.ndarray                    # int32[L], -1 if fixed
    controlled: np.ndarray                 # int32[D], link index of each DOF
    R_fixed: np.ndarray                    # float32[L,3,3]
    trans: np.ndarray                      # float32[L,3]
    axis: np.ndarray                       # float32[L,3] raw <axis>, zeros if absent
    rot_axis: np.ndarray                   # int32[L]  stateless rotation axis 0/1/2
    rot_sign: np.ndarray                   # float32[L] sign(axis_k) in {-1,0,1}
    clamp: np.ndarray                      # int32[L]  stateless: clamp q to [lower, upper]
    sf_rot_axis: np.ndarray                # int32[L]  stateful rotation axis (sign ignored)
    sf_clamp: np.ndarray                   # int32[L]  stateful: clamp whenever limits exist
    jac_axis: np.ndarray                   # int32[L]  first non-zero component of <axis> (-1 if none)
    joint_list_idx: np.ndarray             # int32[L]  index of the parent joint in the file's joint list (-1 root)
    lower: np.ndarray                      # float32[L]
    upper: np.ndarray                      # float32[L]
    has_limits: np.ndarray                 # int32[L]
    lower64: np.ndarray                    # float64[L] limits as parsed (get_joint_limit_array returns doubles)
    upper64: np.ndarray
    velocity64: np.ndarray                 # float64[L] (nan when absent)
    # traversal tables for the kernels
    order: np.ndarray                      # int32[L] DFS pre-order (file indices), order[0] == 0
    subtree_end: np.ndarray                # int32[L] per pre-order position: one past the last descendant
    parent_slot: np.ndarray                # int32[L] per position: -1 = parent is the previous position, else pose slot
    store_slot: np.ndarray                 # int32[L] per position: pose slot to save this link's pose to, or -1
    n_slots: int = 0
    base_R: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    base_t: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=np.float32))
    name_to_idx: Dict[str, int] = field(default_factory=dict)

    # ------------------------------------------------------------------
    @classmethod
    def from_urdf(cls, path: str) -> "KinModel":
        return cls.from_parsed(parse_urdf(path))

    @classmethod
    def from_parsed(cls, urdf: UrdfModel) -> "KinModel":
        L = len(urdf.links)
        name_to_idx = {}
        for i, name in enumerate(urdf.links):
            name_to_idx[name] = i          # later duplicates win, as in robot_tree.py:119
        # first joint in file order whose child is the link (utils.py:188-192)
        joint_of: Dict[str, int] = {}
        for j, joint in enumerate(urdf.joints):
            joint_of.setdefault(joint.child, j)

        parent = np.full(L, -1, np.int32)
        joint_type = np.zeros(L, np.int32)
        dof_idx = np.full(L, -1, np.int32)
        R_fixed = np.tile(np.eye(3, dtype=np.float32), (L, 1, 1))
        trans = np.zeros((L, 3), np.float32)
        axis = np.zeros((L, 3), np.float32)
        rot_axis = np.full(L, 2, np.int32)
        rot_sign = np.zeros(L, np.float32)
        clamp = np.zeros(L, np.int32)
        sf_rot_axis = np.full(L, 2, np.int32)
        sf_clamp = np.zeros(L, np.int32)
        jac_axis = np.full(L, -1, np.int32)
        joint_list_idx = np.full(L, -1, np.int32)
        lower = np.zeros(L, np.float32)
        upper = np.zeros(L, np.float32)
        has_limits = np.zeros(L, np.int32)
        lower64 = np.zeros(L, np.float64)
        upper64 = np.zeros(L, np.float64)
        velocity64 = np.full(L, np.nan, np.float64)
        joint_names = ["base_joint"] * L
        controlled: List[int] = []

        for i, link in enumerate(urdf.links):
            if i == 0:
                continue                   # root: identity, fixed (utils.py:204-211)
            if link not in joint_of:
                raise ValueError(
                    f"link {link!r} is not the child of any joint; only links[0] "
                    f"({urdf.links[0]!r}) may be the tree root")
            j = joint_of[link]
            joint = urdf.joints[j]
            if joint.parent not in name_to_idx:
                raise ValueError(f"joint {joint.name!r}: unknown parent link {joint.parent!r}")
            joint_list_idx[i] = j
            joint_names[i] = joint.name
            parent[i] = name_to_idx[joint.parent]
            trans[i] = np.asarray(joint.xyz, np.float32)
            R_fixed[i] = fixed_rotation_from_rpy(joint.rpy)
            if joint.axis is not None:
                axis[i] = np.asarray(joint.axis, np.float32)
            jt = _TYPE_CODE.get(joint.type, JOINT_UNSUPPORTED)
            joint_type[i] = jt
            if jt != JOINT_FIXED:
                dof_idx[i] = len(controlled)
                controlled.append(i)
                if joint.has_limit:
                    has_limits[i] = 1
                    lo, hi = joint.lower, joint.upper
                    if joint.type == "continuous":          # utils.py:241-243
                        lo, hi = -math.pi, math.pi
                    lower64[i], upper64[i] = lo, hi
                    lower[i], upper[i] = np.float32(lo), np.float32(hi)
                    if joint.velocity is not None:
                        velocity64[i] = joint.velocity
                # stateless clamp: type != continuous and limits exist (rigid_body.py:157-160)
                clamp[i] = int(joint.type != "continuous" and joint.has_limit)
                sf_clamp[i] = int(joint.has_limit)           # rigid_body.py:218-224
            ax = axis[i]
            # stateless axis choice, rigid_body.py:163-168
            if abs(ax[0]) == 1:
                rot_axis[i], rot_sign[i] = 0, np.sign(ax[0])
            elif abs(ax[1]) == 1:
                rot_axis[i], rot_sign[i] = 1, np.sign(ax[1])
            else:
                rot_axis[i], rot_sign[i] = 2, np.sign(ax[2])
            # stateful axis choice, rigid_body.py:101-106
            if ax[0] == 1:
                sf_rot_axis[i] = 0
            elif ax[1] == 1:
                sf_rot_axis[i] = 1
            else:
                sf_rot_axis[i] = 2
            nz = np.nonzero(ax)[0]
            jac_axis[i] = int(nz[0]) if nz.size else -1      # rigid_body.py:97-99

        # tree traversal: children in file order (robot_tree.py:122-126), DFS pre-order from link 0
        children: List[List[int]] = [[] for _ in range(L)]
        for i in range(1, L):
            if parent[i] == i:
                raise ValueError(f"link {urdf.links[i]!r} is its own parent")
            children[parent[i]].append(i)
        order: List[int] = []
        stack = [0]
        seen = np.zeros(L, bool)
        while stack:
            n = stack.pop()
            if seen[n]:
                raise ValueError("kinematic loop in URDF")
            seen[n] = True
            order.append(n)
            stack.extend(reversed(children[n]))
        if len(order) != L:
            missing = [urdf.links[i] for i in range(L) if not seen[i]]
            raise ValueError(f"links not reachable from root {urdf.links[0]!r}: {missing}")
        pos_of = np.zeros(L, np.int32)
        pos_of[np.asarray(order)] = np.arange(L, dtype=np.int32)
        subtree_end = np.zeros(L, np.int32)
        size = np.ones(L, np.int64)
        for n in reversed(order):
            for c in children[n]:
                size[n] += size[c]
        for p, n in enumerate(order):
            subtree_end[p] = p + size[n]
        # pose slots: a link with children that do not directly follow it in
        # pre-order keeps its pose in a slot until its last child has started.
        parent_slot = np.full(L, -1, np.int32)
        store_slot = np.full(L, -1, np.int32)
        free = list(range(MAX_POSE_SLOTS - 1, -1, -1))
        slot_of: Dict[int, int] = {}
        release_at: Dict[int, List[int]] = {}
        n_slots = 0
        for p, n in enumerate(order):
            for s in release_at.pop(p, []):
                free.append(s)
            if p > 0 and order[p - 1] != parent[n]:
                parent_slot[p] = slot_of[int(parent[n])]
            later = [c for c in children[n] if pos_of[c] != p + 1]
            if later:
                if not free:
                    raise ValueError(f"kinematic tree needs more than {MAX_POSE_SLOTS} pose slots")
                s = free.pop()
                store_slot[p] = s
                slot_of[n] = s
                n_slots = max(n_slots, s + 1)
                # the slot is free again once the last such child has been visited
                release_at.setdefault(int(max(pos_of[c] for c in later)) + 1, []).append(s)

        return cls(name=urdf.name, link_names=list(urdf.links), joint_names=joint_names,
                   n_links=L, n_dofs=len(controlled), parent=parent, joint_type=joint_type,
                   dof_idx=dof_idx, controlled=np.asarray(controlled, np.int32), R_fixed=R_fixed,
                   trans=trans, axis=axis, rot_axis=rot_axis, rot_sign=rot_sign, clamp=clamp,
                   sf_rot_axis=sf_rot_axis, sf_clamp=sf_clamp, jac_axis=jac_axis,
                   joint_list_idx=joint_list_idx, lower=lower, upper=upper, has_limits=has_limits,
                   lower64=lower64, upper64=upper64, velocity64=velocity64,
                   order=np.asarray(order, np.int32), subtree_end=subtree_end,
                   parent_slot=parent_slot, store_slot=store_slot, n_slots=n_slots,
                   name_to_idx=name_to_idx)

    # ------------------------------------------------------------------
    def set_base_pose(self, pose_vec) -> None:
        """`update_base_pose([x,y,z,qw,qx,qy,qz])` (robot_tree.py:133-134, frame.py:41-49)."""
        pose = np.asarray(pose_vec, np.float32).reshape(7)
        self.base_t = pose[:3].copy()
        self.base_R = quat_wxyz_to_rot(pose[3:])

    def reset_base_pose(self) -> None:
        self.base_R = np.eye(3, dtype=np.float32)
        self.base_t = np.zeros(3, dtype=np.float32)

    def has_unsupported_joint(self) -> Optional[str]:
        bad = np.nonzero(self.joint_type == JOINT_UNSUPPORTED)[0]
        return self.link_names[int(bad[0])] if bad.size else None

    def is_serial_chain(self) -> bool:
        return self.n_slots == 0 and bool(np.all(self.parent_slot < 0))

    def ancestors_mask(self, link: int) -> np.ndarray:
        """bool[L]: links on the path root..link (inclusive)."""
        mask = np.zeros(self.n_links, bool)
        n = link
        while n >= 0:
            mask[n] = True
            n = int(self.parent[n])
        return mask
