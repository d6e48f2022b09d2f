"""Drop-in for the reference's kinematics classes, computed by the HIP kernels.

Mirrors `torch_robotics.torch_kinematics_tree.models.robot_tree.DifferentiableTree`
(robot_tree.py:75-492) and the robot subclasses of `models/robots.py:16-133`: same
constructor, method names, argument meaning, tensor layouts and exception types.  The
recursion over `torch.bmm` is replaced by one kernel launch per call (ops.py -> libtrk.so);
autograd sees a single node with an explicit backward kernel.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .kinmodel import KinModel
from .urdf import UrdfJoint, parse_urdf

DATA_DIR = Path(__file__).resolve().parent / "data"
URDF_DIR = DATA_DIR / "urdf"


class Frame:
    """Batch of rigid transforms (reference: geometrics/frame.py:12-129): same constructor -- `Frame(rot=None, trans=None,
    pose=None, device=...)`, identity by default, a single (3,3) / (3,) gets a batch dimension, `pose` = x y z qw qx qy qz."""

    def __init__(self, rot: Optional[torch.Tensor] = None, trans: Optional[torch.Tensor] = None,
                 pose: Optional[torch.Tensor] = None, device=None):
        if device is None:
            device = rot.device if rot is not None else (trans.device if trans is not None else
                                                         (pose.device if pose is not None else "cuda"))
        self.device = torch.device(device)
        if rot is None:
            self._rot = torch.eye(3, device=self.device).unsqueeze(0)
        else:
            self._rot = rot.to(self.device)
            if self._rot.dim() == 2:
                self._rot = self._rot.unsqueeze(0)
        if trans is None:
            self._trans = torch.zeros(1, 3, device=self.device)
        else:
            self._trans = trans.to(self.device)
            if self._trans.dim() == 1:
                self._trans = self._trans.unsqueeze(0)
        if pose is not None:
            self.set_pose(pose)
        assert self._trans.shape[0] == self._rot.shape[0]
        self.batch_size = self._trans.shape[0]

    def set_pose(self, pose: torch.Tensor) -> None:              # frame.py:41-49
        if pose.dim() == 1:
            pose = pose.unsqueeze(0)
        pose = pose.to(self.device)
        self._trans = pose[:, :3].clone()
        self._rot = ops.quat_to_rotmat(pose[:, 3:])
        self.batch_size = self._trans.shape[0]

    def set_translation(self, t: torch.Tensor) -> None:
        self._trans = t.to(self.device)

    def set_rotation(self, rot: torch.Tensor) -> None:
        self._rot = rot.to(self.device)

    @property
    def rotation(self) -> torch.Tensor:
        return self._rot

    @property
    def translation(self) -> torch.Tensor:
        return self._trans

    def get_transform_matrix(self) -> torch.Tensor:          # frame.py:81-85
        H = torch.zeros((self.batch_size, 4, 4), device=self._rot.device, dtype=self._rot.dtype)
        H[:, :3, :3] = self._rot
        H[:, :3, 3] = self._trans
        H[:, 3, 3] = 1.0
        return H

    def get_quaternion(self) -> torch.Tensor:
        """XYZW by the trace method, exactly as the reference's per-sample loop (frame.py:87-114; its callers convert with
        `q_convert_wxyz`, robot_tree.py:214-215).  One `trk_frame_quat_euler` launch for the whole batch."""
        return ops.frame_quat_euler(self._rot)[0]

    def get_quaternion_wxyz(self) -> torch.Tensor:
        """WXYZ through `rotation_matrix_to_q` (quaternion.py:135-166) -- what `link_quat_from_link_tensor` returns."""
        return ops.rotmat_to_quat(self._rot)

    # Pose algebra (frame.py:55-78, 116-121) runs in the trk_frame_* kernels with an explicit reverse mode, so frames taken
    # from `return_dict=True` compose and differentiate like the reference's.
    def inverse(self) -> "Frame":
        return Frame(*ops.frame_compose(ops.FRAME_INVERSE, self._rot, self._trans))

    def multiply_transform(self, frame: "Frame") -> "Frame":
        return Frame(*ops.frame_compose(ops.FRAME_COMPOSE, self._rot, self._trans, frame.rotation, frame.translation))

    def multiply_inv_transform(self, frame: "Frame") -> "Frame":
        """frame^-1 o self (frame.py:70-76 calls multiply_inv_transform(frame.rot, frame.trans, self.rot, self.trans))."""
        return Frame(*ops.frame_compose(ops.FRAME_INV_COMPOSE, self._rot, self._trans, frame.rotation, frame.translation))

    def transform_point(self, point: torch.Tensor) -> torch.Tensor:
        """point (P, 3) in this frame -> (B, P, 3) in the world (what fk_map_collision does for grasped-object points)."""
        return ops.frame_transform_points(self._rot, self._trans, point)

    def get_euler(self):
        e = ops.frame_quat_euler(self._rot, want_quat=False, want_euler=True)[1]
        return e[:, 0], e[:, 1], e[:, 2]


class DifferentiableTree(torch.nn.Module):

    def __init__(self, model_path: str, name="", link_list=None, device="cpu"):
        """Same signature and default as the reference (robot_tree.py:77).  `device` is where the tensors this object CREATES live
        (IK start configurations and results); computation is always on the GPU -- tensors handed in on the host are copied to
        `ops.compute_device()`, the kernels run there, the results come back to the host (`ops.host_round_trip`, differentiable)."""
        super().__init__()
        self.name = name
        self.link_list = link_list
        self._device = torch.device(device)
        self.model_type = str(model_path).split(".")[-1]
        if self.model_type != "urdf":
            # the reference's MJCF ('xml') reader is self-described as not working (models/utils.py:43)
            raise NotImplementedError(f"{self.model_type} is not supported!")
        self.model_path = str(model_path)
        self._kin = self._build_kin_model()
        self._n_dofs = self._kin.n_dofs
        self._controlled_joints = [int(i) for i in self._kin.controlled]
        self._name_to_idx_map: Dict[str, int] = dict(self._kin.name_to_idx)
        self._handle_cache: Optional[ops.ModelHandle] = None

    def _build_kin_model(self) -> KinModel:
        return KinModel.from_urdf(self.model_path)

    # -- device handle (created on first use so that construction works without a GPU) ----------
    @property
    def _handle(self) -> ops.ModelHandle:
        if self._handle_cache is None:
            with torch.cuda.device(self._device if self._device.type == "cuda" else None):
                self._handle_cache = ops.ModelHandle(self._kin)
        return self._handle_cache

    def reset(self):
        """The engine is stateless (no per-body pose cache to contaminate, cf. robot_tree.py:128-131)."""
        return None

    def update_base_pose(self, pose_vec):                     # robot_tree.py:133-134
        pose = torch.as_tensor(pose_vec, dtype=torch.float32).detach().cpu().reshape(-1)[:7].numpy()
        self._kin.set_base_pose(pose)
        if self._handle_cache is not None:
            self._handle_cache.set_base_pose(self._kin.base_R, self._kin.base_t)

    def _check_supported(self):
        bad = self._kin.has_unsupported_joint()
        if bad is not None:
            raise NotImplementedError(f"joint type of link {bad!r} is not supported")   # rigid_body.py:184,251

    def _sel_from_names(self, link_list) -> Optional[List[int]]:
        if link_list is None:
            return None
        try:
            return [self._name_to_idx_map[n] for n in link_list]
        except KeyError as e:
            raise KeyError(f"unknown link {e.args[0]!r}") from None

    def _fk_matrices(self, q: torch.Tensor, sel: Optional[List[int]]) -> torch.Tensor:
        if sel is not None and len(set(sel)) != len(sel):      # duplicates: compute unique, then gather
            uniq = sorted(set(sel))
            H = ops.fk(self._handle, q, uniq)
            return H[:, [uniq.index(s) for s in sel]]
        return ops.fk(self._handle, q, sel)

    # -- stateless FK (robot_tree.py:267-301) ---------------------------------------------------
    @ops.host_round_trip
    def compute_forward_kinematics_all_links(self, q: torch.Tensor, return_dict=False, link_list=None):
        self._check_supported()
        if q.ndim == 1:
            q = q.unsqueeze(0)
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        if link_list is None:
            link_list = self.link_list
        if not return_dict:
            if link_list is None:
                link_list = self.get_link_names()
                sel = None if len(link_list) == self._kin.n_links else self._sel_from_names(link_list)
            else:
                sel = self._sel_from_names(link_list)
            return self._fk_matrices(q, sel)
        names = self.get_link_names() if link_list is None else [n for n in self._kin.link_names if n in set(link_list)]
        H = self._fk_matrices(q, self._sel_from_names(names))
        return {n: Frame(H[:, k, :3, :3], H[:, k, :3, 3]) for k, n in enumerate(names)}

    # -- stateful path (robot_tree.py:192-248) --------------------------------------------------
    @ops.host_round_trip
    def compute_forward_kinematics(self, q: torch.Tensor, qd: torch.Tensor, link_name: str, state_less: bool = False):
        assert q.ndim == 2
        if state_less:
            return self.compute_forward_kinematics_all_links(q, link_list=[link_name])
        pos, quat, _, _ = self._stateful(q, qd, link_name)
        return pos, quat

    def _stateful(self, q, qd, link_name):
        self._check_supported()
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        if qd is not None:
            assert qd.ndim == 2 and qd.shape[1] == self._n_dofs
        return ops.fk_jacobian(self._handle, q, qd, self._name_to_idx_map[link_name])

    @ops.host_round_trip
    def compute_forward_kinematics_and_geometric_jacobian(self, q: torch.Tensor, qd: torch.Tensor, link_name: str):
        return self._stateful(q, qd, link_name)

    @ops.host_round_trip
    def compute_analytical_jacobian_all_links(self, q: torch.Tensor):
        """(N, L, 7, D) Jacobian of [pos, quat_wxyz] of every link (robot_tree.py:250-265): one kernel launch
        instead of 7L autograd traversals."""
        self._check_supported()
        if q.ndim == 1:
            q = q.unsqueeze(0)
        assert q.ndim == 2 and q.shape[1] == self._n_dofs
        return ops.fk_analytic_jacobian(self._handle, q.detach())

    # -- batched Adam IK (robot_tree.py:303-442): one fused kernel per iteration ------------------------
    def _ik_limits(self, eps_joint_lim, device):
        lower, upper, _, _ = self.get_joint_limit_array()
        lo = torch.as_tensor(lower + eps_joint_lim, dtype=torch.float32, device=device)
        hi = torch.as_tensor(upper - eps_joint_lim, dtype=torch.float32, device=device)
        return lo, hi

    @ops.host_round_trip
    def loss_fn_ik_per_q(self, q, H_target, link_name, w_se3=1.0, w_joint_limits=1.0, lower=None, upper=None,
                         w_q_rest=1.0, q_rest=None, debug=False):
        if w_se3 != 1.0 or q_rest is not None:
            raise NotImplementedError("only the configuration inverse_kinematics itself uses (w_se3 = 1, no rest pose)")
        q = q.detach().contiguous()
        loss = torch.empty(q.shape[0], device=q.device, dtype=torch.float32)
        ops.ik_step(self._handle, self._name_to_idx_map[link_name], H_target, lower, upper, q, None, None, 1, lr=0.0,
                    w_joint_limits=w_joint_limits, loss=loss)
        return loss

    @ops.host_round_trip
    def ik_termination(self, q, H_target, link_name, lower, upper, se3_eps=1e-1, debug=False):
        q = q.detach().contiguous()
        valid = torch.empty(q.shape[0], device=q.device, dtype=torch.uint8)
        ops.ik_step(self._handle, self._name_to_idx_map[link_name], H_target, lower, upper, q, None, None, 1, lr=0.0,
                    se3_eps=se3_eps, valid=valid)
        return torch.atleast_1d(torch.argwhere(valid.bool()).squeeze())

    def inverse_kinematics(self, H_target, link_name="ee_link", batch_size=1, max_iters=1000, lr=1e-2, se3_eps=1e-1,
                           q0=None, q0_noise=torch.pi / 8, eps_joint_lim=torch.pi / 100, print_freq=50, debug=False,
                           check_every=1):
        """Same contract as the reference: returns (q, idx_valid).  `check_every` > 1 tests the termination condition
        (a device->host sync) only every so many iterations instead of every iteration."""
        self._check_supported()
        cdev = ops.compute_device(self._device)          # tensors this call creates: on the GPU while it iterates, returned on self._device
        H_target = torch.as_tensor(H_target, dtype=torch.float32, device=cdev)
        if H_target.ndim == 2:
            H_target = H_target.unsqueeze(0)
        Ht = H_target[0].contiguous() if H_target.shape[0] == 1 else H_target.contiguous()
        lo, hi = self._ik_limits(eps_joint_lim, cdev)
        if q0 is None:
            q0 = lo + torch.rand(batch_size, self._n_dofs, device=cdev) * (hi - lo)
        else:
            q0 = torch.as_tensor(q0, dtype=torch.float32, device=cdev)
            q0 = torch.clamp(q0 + torch.randn(batch_size, self._n_dofs, device=cdev) * q0_noise, lo, hi)
            assert q0.shape == (batch_size, self._n_dofs)
        q = q0.clone().contiguous()
        m, v = torch.zeros_like(q), torch.zeros_like(q)
        loss = torch.empty(batch_size, device=cdev, dtype=torch.float32)
        valid = torch.empty(batch_size, device=cdev, dtype=torch.uint8)
        link = self._name_to_idx_map[link_name]
        it, converged = 0, False
        while it < max_iters:
            # iterations it .. it + k - 1 in one call: the termination test (a device->host sync) looks at the state before
            # the first of them, exactly as a loop that tests only every `check_every` iterations would
            k = max(1, min(int(check_every), max_iters - it))
            q_prev = q.clone()
            ops.ik_steps(self._handle, link, Ht, lo, hi, q, m, v, it + 1, k, lr=lr, w_joint_limits=300.0, se3_eps=se3_eps,
                         loss=loss, valid=valid)
            if bool(valid.all()):
                q = q_prev                      # the reference breaks BEFORE updating once every configuration is valid
                print(f"\nIK converged for all joint configurations in {it} iterations")
                converged = True
                break
            if print_freq != -1 and (it == 0 or any((j % print_freq) == 0 for j in range(it, it + k))):
                print(f"\n---> Iter {it}/{max_iters}")
                print(f"Error mean, std: {loss.mean():.3f}, {loss.std():.3f}")
                print(f"idx_valid: {int(valid.sum())}/{batch_size}")
            it += k
        if not converged and max_iters > 0:
            print("\nIK did not converge for all joint configurations!")
        idx_valid = torch.atleast_1d(torch.argwhere(valid.bool()).squeeze())
        return q.to(self._device), idx_valid.to(self._device)

    def inverse_kinematics_gn(self, H_target, link_name="ee_link", batch_size=1, max_iters=40, damping=1e-4, lm_gain=0.1, step_scale=1.0,
                              se3_eps=1e-1, q0=None, q0_noise=torch.pi / 8, eps_joint_lim=torch.pi / 100, check_every=10):
        """Batched IK by damped Gauss-Newton (Levenberg-Marquardt) on the geometric Jacobian -- an EXTENSION (the reference's
        `inverse_kinematics` is Adam, robot_tree.py:303-384): same arguments and return value `(q, idx_valid)`, same termination test
        (SE3 distance < se3_eps inside the shrunk joint limits, :419-442), evaluated every `check_every` iterations; the iterations
        between two tests are ONE launch of `trk_ik_gn_steps` (FK, Jacobian, normal equations, Cholesky and the clamped step per lane
        in registers).  A unit tracking `link_name` is compiled on first use when none is registered (robots up to 9 DOF)."""
        self._check_supported()
        cdev = ops.compute_device(self._device)          # tensors this call creates: on the GPU while it iterates, returned on self._device
        H_target = torch.as_tensor(H_target, dtype=torch.float32, device=cdev)
        if H_target.ndim == 2:
            H_target = H_target.unsqueeze(0)
        Ht = H_target[0].contiguous() if H_target.shape[0] == 1 else H_target.contiguous()
        lo, hi = self._ik_limits(eps_joint_lim, cdev)
        if q0 is None:
            q0 = lo + torch.rand(batch_size, self._n_dofs, device=cdev) * (hi - lo)
        else:
            q0 = torch.as_tensor(q0, dtype=torch.float32, device=cdev)
            q0 = torch.clamp(q0 + torch.randn(batch_size, self._n_dofs, device=cdev) * q0_noise, lo, hi)
            assert q0.shape == (batch_size, self._n_dofs)
        q = q0.clone().contiguous()
        err = torch.empty(batch_size, device=cdev, dtype=torch.float32)
        valid = torch.empty(batch_size, device=cdev, dtype=torch.uint8)
        link = self._name_to_idx_map[link_name]
        self._ensure_gn_unit(link)
        it = 0
        while it < max_iters:
            k = max(1, min(int(check_every), max_iters - it))
            q_prev = q.clone()
            ops.ik_gn_steps(self._handle, link, Ht, lo, hi, q, k, damping=damping, lm_gain=lm_gain, step_scale=step_scale,
                            se3_eps=se3_eps, err=err, valid=valid)
            if bool(valid.all()):
                q = q_prev                      # like the reference's loop: stop BEFORE updating once every configuration is valid
                break
            it += k
        else:
            ops.ik_gn_steps(self._handle, link, Ht, lo, hi, q.clone(), 1, damping=damping, lm_gain=lm_gain, step_scale=step_scale,
                            se3_eps=se3_eps, err=err, valid=valid)          # validity of the final configurations
        idx_valid = torch.atleast_1d(torch.argwhere(valid.bool()).squeeze())
        return q.to(self._device), idx_valid.to(self._device)

    def _ensure_gn_unit(self, link: int) -> None:
        """a generated unit of this robot that tracks `link` (the Gauss-Newton IK lives in generated kernels only)"""
        from . import codegen, jit
        kin = self._kin
        h = codegen.model_hash(kin)
        if any(mh == h and t.ee_link == link for _i, mh, t in codegen.aot_units()):
            return
        if any(mh == h and t.ee_link == link for mh, t in jit._loaded_templates.values()):
            return
        tmpl = codegen.default_template(kin)
        jit.specialize(kin, tmpl.obj_links, (), ee_link=link)

    # -- model queries ----------------------------------------------------------------------------
    def get_joint_limits(self) -> List[Optional[Dict[str, float]]]:
        k, out = self._kin, []
        for i in self._controlled_joints:
            if not k.has_limits[i]:
                out.append(None)
                continue
            vel = k.velocity64[i]
            out.append({"effort": None, "lower": float(k.lower64[i]), "upper": float(k.upper64[i]),
                        "velocity": None if np.isnan(vel) else float(vel)})
        return out

    def get_joint_limit_array(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:   # robot_tree.py:455-471
        k = self._kin
        idx = np.asarray(self._controlled_joints, int)
        if not np.all(k.has_limits[idx]):
            raise TypeError("'NoneType' object is not subscriptable")   # what the reference raises for limit-less joints
        return (k.lower64[idx].copy(), k.upper64[idx].copy(), -k.velocity64[idx], k.velocity64[idx].copy())

    def get_link_names(self) -> List[str]:
        return list(self._kin.link_names)

    def print_link_names(self) -> None:
        for n in self.get_link_names():
            print(n)


def _tree(urdf_name: str, name: str):
    class _Robot(DifferentiableTree):
        def __init__(self, link_list: Optional[List[str]] = None, device="cpu", **kwargs):
            self.model_path = (URDF_DIR / urdf_name).as_posix()
            super().__init__(self.model_path, name, link_list=link_list, device=device)
    return _Robot


def quat_wxyz_to_rpy(q) -> Tuple[float, float, float]:
    """Roll / pitch / yaw of a wxyz quaternion in fp32 (what the reference writes into the grasped object's fixed
    joint: q_to_euler quaternion.py:203-215 via robots.py:25-37)."""
    w, x, y, z = torch.tensor(np.asarray(q, np.float32).reshape(4)).unbind(-1)
    roll = torch.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = torch.asin(torch.clamp(2.0 * (w * y - z * x), -1.0, 1.0))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return float(roll), float(pitch), float(yaw)


class DifferentiableFrankaPanda(DifferentiableTree):          # robots.py:56-69
    def __init__(self, link_list=None, gripper=False, device="cpu", grasped_object=None):
        fname = "panda_arm_hand.urdf" if gripper else "panda_arm_no_gripper.urdf"
        self._grasped_object = grasped_object
        super().__init__((URDF_DIR / fname).as_posix(), "differentiable_franka_panda", link_list=link_list, device=device)

    def _build_kin_model(self) -> KinModel:
        """With a grasped object the reference rewrites the URDF on disk with one more link, `grasped_object`, fixed
        to `panda_hand` at the object's pose (modidy_franka_panda_urdf_grasped_object robots.py:24-53).  Here the same
        link and joint are appended to the parsed description in memory; nothing is written."""
        urdf = parse_urdf(self.model_path)
        go = self._grasped_object
        if go is not None:
            xyz = [float(v) for v in np.asarray(go.pos, np.float32).reshape(3)]
            urdf.joints.append(UrdfJoint(name="grasped_object_fixed_joint", type="fixed", parent="panda_hand",
                                         child="grasped_object", xyz=xyz, rpy=list(quat_wxyz_to_rpy(go.ori))))
            urdf.links.append("grasped_object")
        return KinModel.from_parsed(urdf)


DifferentiableKUKAiiwa = _tree("iiwa7.urdf", "differentiable_kuka_iiwa")
DifferentiableUR10 = _tree("ur10.urdf", "differentiable_ur10")
DifferentiableHabitatStretch = _tree("hab_stretch.urdf", "differentiable_stretch")
DifferentiableShadowHand = _tree("shadow_hand.urdf", "differentiable_shadow_hand")
DifferentiableAllegroHand = _tree("allegro_hand.urdf", "differentiable_allegro_hand")
DifferentiableUR10Allegro = _tree("ur10_allegro.urdf", "differentiable_ur10_allegro")
DifferentiableDualPanda = _tree("dual_panda.urdf", "differentiable_dual_panda")


class DifferentiableTiagoDualHoloMove(DifferentiableTree):     # robots.py:104-112
    def __init__(self, link_list=None, device="cpu"):
        super().__init__((URDF_DIR / "tiago_dual_holobase_minimal_holonomic.urdf").as_posix(),
                         "differentiable_tiago_dual_holo_move", link_list=link_list, device=device)

    def get_link_names(self):   # the reference pops the three virtual base links (robots.py:111-112)
        return super().get_link_names()[3:]


# tensor helpers with the reference's names (geometrics/utils.py:321-344)
def x_rot(angle):                                            # spatial_vector.py:8-20
    return ops.axis_rotation(0, angle)


def y_rot(angle):                                            # spatial_vector.py:22-34
    return ops.axis_rotation(1, angle)


def z_rot(angle):                                            # spatial_vector.py:36-47
    return ops.axis_rotation(2, angle)


def q_to_rotation_matrix(q):                                 # quaternion.py:102-120
    return ops.quat_to_rotmat(q)


def link_pos_from_link_tensor(link_tensor):
    if link_tensor.shape[-1] == 4:
        return link_tensor[..., :3, 3]
    if link_tensor.shape[-1] == 3:
        return link_tensor[..., :2, 2]
    raise ValueError


def link_rot_from_link_tensor(link_tensor):
    if link_tensor.shape[-1] == 4:
        return link_tensor[..., :3, :3]
    if link_tensor.shape[-1] == 3:
        return link_tensor[..., :2, :2]
    raise ValueError


def link_quat_from_link_tensor(link_tensor):
    return ops.rotmat_to_quat(link_rot_from_link_tensor(link_tensor).contiguous())
