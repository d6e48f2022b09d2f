"""Drop-in for `torch_robotics.torch_planning_objectives.fields.distance_fields` (distance_fields.py:12-359).

Same classes, constructor keywords and call signatures: `compute_cost(q, link_pos, field_type=..., margin=...)`,
`compute_costs_impl`, `EESE3DistanceField.compute_distance / update_target`.  Evaluation happens in the
`trk_cost_fields` / `trk_collision_fields` / `trk_ee_cost` kernels; autograd sees one node with an explicit
backward (ops.py).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from ._abi import FIELD_OBJECTS, FIELD_SELF, FIELD_WS
from .costmodel import CostModelSpec, interpolation_table
from .environments import _np, objects_to_spec_parts, scene_version


class DistanceField:
    def __init__(self, tensor_args=None):
        self.tensor_args = tensor_args

    @ops.host_round_trip
    def compute_cost(self, q, link_pos, *args, **kwargs):
        """Rank normaliser of distance_fields.py:26-55: (b,d) | (b,t,d) | (b,h,t,3) | (b,h,t,4,4) -> cost (b,h)."""
        shape = link_pos.shape
        if len(shape) == 2:
            b, h = shape[0], 1
            link_pos = link_pos.unsqueeze(1)
        elif len(shape) == 3:
            b, h = shape[0], 1
        elif len(shape) == 4:
            b, h = shape[0], shape[1]
            link_pos = link_pos.reshape((b * h,) + tuple(shape[2:]))
        elif len(shape) == 5:       # the reference raises an EinopsError here (duplicate axis name); the intent is clear
            b, h = shape[0], shape[1]
            link_pos = link_pos.reshape((b * h,) + tuple(shape[2:]))
        else:
            raise NotImplementedError
        cost = self.compute_costs_impl(q, link_pos, *args, **kwargs)
        if cost.ndim == 1:
            cost = cost.reshape(b, h)
        return cost

    def compute_costs_impl(self, *args, **kwargs):
        raise NotImplementedError

    def zero_grad(self):
        pass


class _InterpolateColumns(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, src, w, n_out):
        ctx.save_for_backward(src, w)
        ctx.n_in = int(x.shape[-2])
        return ops.interpolate_columns(x, src, w, n_out)

    @staticmethod
    def backward(ctx, g):
        src, w = ctx.saved_tensors
        return ops.interpolate_columns_backward(g.contiguous(), src, w, ctx.n_in), None, None, None


@ops.host_round_trip
def interpolate_points_v1(points, num_interpolated_points):
    """distance_fields.py:66-69: `F.interpolate(points.transpose(-2, -1), size=K, mode='linear', align_corners=True)` back
    transposed -- (..., L, C) -> (..., K, C), linear along the link axis; differentiable w.r.t. the points."""
    L, K = int(points.shape[-2]), int(num_interpolated_points)
    src, w = interpolation_table(L, K)
    src_t = torch.as_tensor(src.reshape(-1), device=points.device)
    w_t = torch.as_tensor(w.reshape(-1), device=points.device)
    if torch.is_grad_enabled() and points.requires_grad:
        return _InterpolateColumns.apply(points, src_t, w_t, K)
    return ops.interpolate_columns(points, src_t, w_t, K)


class EmbodimentDistanceFieldBase(DistanceField):
    _field = 0

    def __init__(self, robot, link_idxs_for_collision_checking=None, num_interpolated_points=30,
                 collision_margins=0.0, cutoff_margin=0.001, field_type="sdf", clamp_sdf=False,
                 interpolate_link_pos=False, **kwargs):
        super().__init__(**kwargs)
        assert robot is not None, "You need to pass a robot instance to the embodiment distance fields"
        # interpolate_link_pos=True (distance_fields.py:145-147): the field is evaluated on `num_interpolated_points` points
        # spread linearly along the chain of selected links (interpolate_points_v1, :66-69) instead of the link origins;
        # margins and self-collision pair rows then refer to those points (robot_base.py:70-73, 103-108).  The reference
        # indexes the interpolated tensor with `link_idxs_for_collision_checking` a second time (:109) -- with the Panda's link
        # indices it would evaluate interpolated points no. 2, 3, 5, 7, 9 of the K -- so what is built is the evident intent,
        # every interpolated point with its own margin, pinned by the reference's own functions called with the index set
        # that makes them do that (oracle/gen_golden.py: cost_interp.npz).
        self.interpolate_link_pos = bool(interpolate_link_pos)
        self.clamp_sdf = bool(clamp_sdf)        # relu(margin - sdf) per link / pair (distance_fields.py:114-117)
        self.robot = robot
        self.link_idxs_for_collision_checking = link_idxs_for_collision_checking
        self.num_interpolated_points = num_interpolated_points
        self.collision_margins = collision_margins
        self.cutoff_margin = cutoff_margin
        self.field_type = field_type
        self._handles = {}

    # subclasses fill the parts of the spec they own
    def _fill_spec(self, spec: CostModelSpec) -> None:
        raise NotImplementedError

    def _scene_version(self):
        return 0

    def _handle(self, n_links_in: int, device) -> ops.CostHandle:
        key = (n_links_in, str(device), self._scene_version(), bool(self.clamp_sdf), self.interpolate_link_pos,
               int(self.num_interpolated_points) if self.interpolate_link_pos else 0)
        if key not in self._handles:
            spec = CostModelSpec(n_links_in=n_links_in)
            self._fill_spec(spec)
            spec.clamp_fields = self._field if self.clamp_sdf else 0
            self._handles = {key: ops.CostHandle(spec, device)}
        return self._handles[key]

    def _columns(self, n_links_in: int, spec: Optional[CostModelSpec] = None) -> np.ndarray:
        """Columns of the position tensor this field reads: the selected robot links, then -- when the robot holds an
        object -- the grasped points, which fk_map_collision appends after the links (robot_panda.py:154-168).
        The reference's own gather for that case (distance_fields.py:134-155) raises on a shape mismatch; what it
        evidently means (robot links ++ grasped points, margins and pair rows as RobotBase builds them,
        robot_base.py:71-141) is what is implemented, and pinned with the reference's field code in
        tests/golden/grasp_panda.npz.
        With interpolate_link_pos the selected links are replaced by `num_interpolated_points` virtual columns of `spec`
        (linear interpolation along the selected links); grasped points are not interpolated (distance_fields.py:149-152)."""
        cols = list(self.link_idxs_for_collision_checking)
        if self.interpolate_link_pos:
            if spec is None:
                raise ValueError("interpolate_link_pos: the cost spec that receives the interpolated columns is required")
            src, w = interpolation_table(len(cols), int(self.num_interpolated_points))
            cols = list(spec.add_virtual_columns(np.asarray(cols, np.int32)[src], w))
        go = getattr(self.robot, "grasped_object", None)
        if go is not None:
            G = go.n_base_points_for_collision
            cols += list(range(n_links_in - G, n_links_in))
        return np.asarray(cols, np.int32)

    def _margin_vector(self, n_cols=None) -> np.ndarray:
        """collision_margins + cutoff_margin in fp32 (distance_fields.py:112)."""
        cm = torch.as_tensor(_np(self.collision_margins), dtype=torch.float32)
        co = self.cutoff_margin
        co = torch.as_tensor(_np(co), dtype=torch.float32) if not isinstance(co, (int, float)) else co
        out = cm + co
        n = len(self.link_idxs_for_collision_checking) if n_cols is None else n_cols
        return np.broadcast_to(out.numpy().astype(np.float32).reshape(-1), (n,)).copy() if out.ndim <= 1 and out.numel() in (1, n) \
            else out.numpy().astype(np.float32)

    def compute_costs_impl(self, q, link_pos, **kwargs):        # distance_fields.py:134-155
        return self.compute_embodiment_cost(q, link_pos, **kwargs)

    @ops.host_round_trip
    def compute_embodiment_cost(self, q, link_pos, field_type=None, **kwargs):   # distance_fields.py:107-130
        if field_type is None:
            field_type = self.field_type
        lead = link_pos.shape[:-2]
        cm = self._handle(link_pos.shape[-2], link_pos.device)
        if field_type == "sdf":
            return ops.cost_fields_ad(cm, self._field, link_pos).reshape(lead)
        if field_type == "occupancy":
            return self.compute_embodiment_collision(q, link_pos, **kwargs)
        raise NotImplementedError("field_type {} not implemented".format(field_type))

    @ops.host_round_trip
    def compute_embodiment_collision(self, q, link_pos, **kwargs):
        lead = link_pos.shape[:-2]
        cm = self._handle(link_pos.shape[-2], link_pos.device)
        margin = kwargs.get("margin", None)
        return ops.collision_fields(cm, self._field, link_pos, margin=margin).reshape(lead)

    def compute_embodiment_signed_distances(self, *args, **kwargs):
        raise NotImplementedError

    @ops.host_round_trip
    def compute_distance(self, q, link_pos, **kwargs):
        raise NotImplementedError


class CollisionSelfField(EmbodimentDistanceFieldBase):        # distance_fields.py:180-215
    _field = FIELD_SELF

    def __init__(self, *args, idxs_links_distance_matrix=None, **kwargs):
        super().__init__(*args, collision_margins=0.0, **kwargs)
        self.idxs_links_distance_matrix = idxs_links_distance_matrix
        self.idxs_links_distance_matrix_tuple = tuple(zip(*idxs_links_distance_matrix))

    def _fill_spec(self, spec):
        spec.self_link_idx = self._columns(spec.n_links_in, spec)
        cm = _np(self.cutoff_margin).astype(np.float32).reshape(-1)
        if len(spec.self_link_idx) == 1:
            # one self-collision point: the reference ignores the pair table and returns |p|_1 * 1e9 as "the distance"
            # (distance_fields.py:195-198); the degenerate pair (0, 0) is how the cost model says that
            spec.self_pairs = np.zeros((1, 2), np.int32)
            spec.self_margin = cm[:1].copy()
            return
        spec.self_pairs = np.asarray(self.idxs_links_distance_matrix, np.int32).reshape(-1, 2)
        spec.self_margin = np.broadcast_to(cm, (len(spec.self_pairs),)).copy()


class CollisionObjectBase(EmbodimentDistanceFieldBase):       # distance_fields.py:269-295
    def __init__(self, *args, link_margins_for_object_collision_checking_tensor=None, **kwargs):
        super().__init__(*args, collision_margins=link_margins_for_object_collision_checking_tensor, **kwargs)


class CollisionObjectDistanceField(CollisionObjectBase):      # distance_fields.py:298-316
    _field = FIELD_OBJECTS

    def __init__(self, *args, df_obj_list_fn=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.df_obj_list_fn = df_obj_list_fn

    def _scene_version(self):
        return scene_version(self.df_obj_list_fn() if self.df_obj_list_fn is not None else [])

    def _fill_spec(self, spec):
        spec.obj_link_idx = self._columns(spec.n_links_in, spec)
        spec.obj_link_margin = self._margin_vector(len(spec.obj_link_idx))
        objs = self.df_obj_list_fn() if self.df_obj_list_fn is not None else []
        spec.objects, spec.grid = objects_to_spec_parts(objs)


class CollisionWorkspaceBoundariesDistanceField(CollisionObjectBase):    # distance_fields.py:319-332
    _field = FIELD_WS

    def __init__(self, *args, ws_bounds_min=None, ws_bounds_max=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.ws_min, self.ws_max = ws_bounds_min, ws_bounds_max

    def _fill_spec(self, spec):
        spec.obj_link_idx = self._columns(spec.n_links_in, spec)
        spec.obj_link_margin = self._margin_vector(len(spec.obj_link_idx))
        spec.ws_min = _np(self.ws_min).astype(np.float32).reshape(3)
        spec.ws_max = _np(self.ws_max).astype(np.float32).reshape(3)


class EESE3DistanceField(DistanceField):                      # distance_fields.py:335-359
    def __init__(self, target_H, w_pos=1.0, w_rot=1.0, square=True, **kwargs):
        super().__init__(**kwargs)
        self.target_H = target_H
        self.square, self.w_pos, self.w_rot = square, w_pos, w_rot
        self._cm: Optional[ops.CostHandle] = None

    def update_target(self, target_H):
        self.target_H = target_H

    def _handle(self, device, square) -> ops.CostHandle:
        key = (str(device), bool(square), float(self.w_pos), float(self.w_rot))
        if self._cm is None or self._cm[0] != key:
            spec = CostModelSpec(n_links_in=1, ee_link=0, ee_w_pos=self.w_pos, ee_w_rot=self.w_rot, ee_square=square)
            self._cm = (key, ops.CostHandle(spec, device))
        return self._cm[1]

    def _eval(self, link_tensor, square):
        H = link_tensor[..., -1, :, :]                          # the LAST link is the end effector (:350)
        lead = H.shape[:-2]
        target = torch.as_tensor(self.target_H, dtype=torch.float32, device=H.device)
        if target.dim() == 2:
            tgt = target.contiguous()
        else:
            tgt = target.reshape(-1, 4, 4).contiguous()
        cost = ops.ee_cost_ad(self._handle(H.device, square), H, tgt)
        return cost.reshape(lead)

    @ops.host_round_trip
    def compute_distance(self, link_tensor):
        return self._eval(link_tensor, square=False)

    def compute_costs_impl(self, q, link_tensor, **kwargs):
        return self._eval(link_tensor, square=self.square).squeeze()

    def zero_grad(self):
        raise NotImplementedError


_se3_handles = {}


@ops.host_round_trip
def SE3_distance(H_batch, H_target, vel_batch=None, vel_target=None, w_pos=1.0, w_rot=1.0, **kwargs):
    """geometrics/utils.py:130-178 on the pose part: w_rot (1 - (tr(R R_t^T) - 1) / 2) + w_pos ||p - p_t|| for H_batch (..., 4, 4)
    against one (4, 4) or per-sample (..., 4, 4) target; differentiable w.r.t. H_batch (`trk_ee_cost`).  The velocity terms of the
    reference signature are not part of the planning path."""
    if vel_batch is not None or vel_target is not None:
        raise NotImplementedError("SE3_distance: velocity terms are outside the built path")
    key = (str(H_batch.device), float(w_pos), float(w_rot))
    if key not in _se3_handles:
        spec = CostModelSpec(n_links_in=1, ee_link=0, ee_w_pos=float(w_pos), ee_w_rot=float(w_rot), ee_square=False)
        _se3_handles[key] = ops.CostHandle(spec, H_batch.device)
    lead = H_batch.shape[:-2]
    target = torch.as_tensor(H_target, dtype=torch.float32, device=H_batch.device)
    tgt = target.contiguous() if target.dim() == 2 else target.reshape(-1, 4, 4).contiguous()
    return ops.ee_cost_ad(_se3_handles[key], H_batch, tgt).reshape(lead)
