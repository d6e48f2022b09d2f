"""`torch.ops.trk.*`: the hot-path operators registered with the PyTorch dispatcher (`torch.library.custom_op`).

Every op is a thin wrapper over one entry point of the C ABI (`include/trk.h`), has a fake (meta) implementation so that
`torch.compile` / `torch.export` can trace through it, and -- where the reference's call site is differentiable -- an autograd
formula whose backward is ANOTHER registered op running the explicit reverse-mode kernel (never a recorded graph of small ops):

    reference call site                                           op                                backward op
    compute_forward_kinematics_all_links  robot_tree.py:267-301   trk::fk                           trk::fk_backward
    fk_map_collision                      robot_panda.py:138-170  trk::fk_positions                 trk::fk_positions_backward
    compute_embodiment_cost               distance_fields.py:107  trk::cost_fields                  trk::cost_fields_backward
    EESE3DistanceField.compute_costs_impl distance_fields.py:347  trk::ee_cost                      trk::ee_cost_backward
    PlanningTask.compute_collision_cost   tasks.py:135-137        trk::rollout_cost_grad (fp32/fp16) trk::scale_rows

Kinematic models, cost models and point sets are passed as INTEGER handles (`ModelHandle.uid` ...: the dispatcher knows tensors
and scalars only); `handle_of(uid)` finds the live Python object, which owns the C handle.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops

handle_of = ops.handle_of


def _model(uid):
    return handle_of(uid, ops.ModelHandle)


def _cost(uid):
    return handle_of(uid, ops.CostHandle)


def _points(uid):
    return handle_of(uid, ops.PointSetHandle)


def _n_cols(model_uid: int, sel: Optional[Sequence[int]]) -> int:
    return len(sel) if sel is not None else _model(model_uid).n_links


def _rows(q: Tensor, d: int) -> int:
    return q.numel() // max(1, d)


# ----------------------------------------------------------------------------------------------------------------------
# forward kinematics
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("trk::fk", mutates_args=(), device_types="cuda")
def fk(q: Tensor, model: int, sel: Optional[List[int]]) -> Tensor:
    return ops.fk_forward(_model(model), q, sel)


@fk.register_fake
def _(q, model, sel):
    m = _model(model)
    return q.new_empty((_rows(q, m.n_dofs), _n_cols(model, sel), 4, 4), dtype=torch.float32)


@torch.library.custom_op("trk::fk_backward", mutates_args=(), device_types="cuda")
def fk_backward(q: Tensor, gH: Tensor, model: int, sel: Optional[List[int]]) -> Tensor:
    return ops.fk_backward(_model(model), q, gH.contiguous(), sel).reshape(q.shape)


@fk_backward.register_fake
def _(q, gH, model, sel):
    return torch.empty_like(q, dtype=torch.float32)


def _fk_setup(ctx, inputs, output):
    q, ctx.model, ctx.sel = inputs
    ctx.save_for_backward(q)


def _fk_bwd(ctx, gH):
    (q,) = ctx.saved_tensors
    return torch.ops.trk.fk_backward(q, gH, ctx.model, ctx.sel), None, None


fk.register_autograd(_fk_bwd, setup_context=_fk_setup)


@torch.library.custom_op("trk::fk_positions", mutates_args=(), device_types="cuda")
def fk_positions(q: Tensor, model: int, sel: Optional[List[int]]) -> Tensor:
    return ops.fk_positions(_model(model), q, sel)


@fk_positions.register_fake
def _(q, model, sel):
    m = _model(model)
    return q.new_empty((_rows(q, m.n_dofs), _n_cols(model, sel), 3), dtype=torch.float32)


@torch.library.custom_op("trk::fk_positions_backward", mutates_args=(), device_types="cuda")
def fk_positions_backward(q: Tensor, gpos: Tensor, model: int, sel: Optional[List[int]]) -> Tensor:
    return ops.fk_positions_backward(_model(model), q, gpos.contiguous(), sel).reshape(q.shape)


@fk_positions_backward.register_fake
def _(q, gpos, model, sel):
    return torch.empty_like(q, dtype=torch.float32)


def _fkp_bwd(ctx, gpos):
    (q,) = ctx.saved_tensors
    return torch.ops.trk.fk_positions_backward(q, gpos, ctx.model, ctx.sel), None, None


fk_positions.register_autograd(_fkp_bwd, setup_context=_fk_setup)


# ----------------------------------------------------------------------------------------------------------------------
# collision fields on given link positions, EE tracking on given transforms
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("trk::cost_fields", mutates_args=(), device_types="cuda")
def cost_fields(link_pos: Tensor, cm: int, fields: int) -> Tensor:
    return ops.cost_fields(_cost(cm), fields, link_pos)


@cost_fields.register_fake
def _(link_pos, cm, fields):
    return link_pos.new_empty((link_pos.numel() // (3 * _cost(cm).n_links_in),), dtype=torch.float32)


@torch.library.custom_op("trk::cost_fields_backward", mutates_args=(), device_types="cuda")
def cost_fields_backward(link_pos: Tensor, gcost: Tensor, cm: int, fields: int) -> Tensor:
    _, g = ops.cost_fields(_cost(cm), fields, link_pos, gcost=gcost.contiguous(), want_grad=True)
    return g.reshape(link_pos.shape)


@cost_fields_backward.register_fake
def _(link_pos, gcost, cm, fields):
    return torch.empty_like(link_pos, dtype=torch.float32)


def _cf_setup(ctx, inputs, output):
    link_pos, ctx.cm, ctx.fields = inputs
    ctx.save_for_backward(link_pos)


def _cf_bwd(ctx, gcost):
    (link_pos,) = ctx.saved_tensors
    return torch.ops.trk.cost_fields_backward(link_pos, gcost, ctx.cm, ctx.fields), None, None


cost_fields.register_autograd(_cf_bwd, setup_context=_cf_setup)


@torch.library.custom_op("trk::ee_cost", mutates_args=(), device_types="cuda")
def ee_cost(H: Tensor, target: Optional[Tensor], cm: int) -> Tensor:
    return ops.ee_cost(_cost(cm), H, target)


@ee_cost.register_fake
def _(H, target, cm):
    return H.new_empty((H.numel() // 16,), dtype=torch.float32)


@torch.library.custom_op("trk::ee_cost_backward", mutates_args=(), device_types="cuda")
def ee_cost_backward(H: Tensor, target: Optional[Tensor], gcost: Tensor, cm: int) -> Tensor:
    _, gH = ops.ee_cost(_cost(cm), H, target, gcost=gcost.contiguous(), want_grad=True)
    return gH.reshape(H.shape)


@ee_cost_backward.register_fake
def _(H, target, gcost, cm):
    return torch.empty_like(H, dtype=torch.float32)


def _ee_setup(ctx, inputs, output):
    H, target, ctx.cm = inputs
    ctx.has_target = target is not None
    ctx.save_for_backward(H, *([target] if target is not None else []))


def _ee_bwd(ctx, gcost):
    H = ctx.saved_tensors[0]
    target = ctx.saved_tensors[1] if ctx.has_target else None
    return torch.ops.trk.ee_cost_backward(H, target, gcost, ctx.cm), None, None


ee_cost.register_autograd(_ee_bwd, setup_context=_ee_setup)


# ----------------------------------------------------------------------------------------------------------------------
# the fused rollout (FK -> objectives -> d cost / d q), fp32 or fp16 I/O by the dtype of q
# ----------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("trk::rollout_cost_grad", mutates_args=(), device_types="cuda")
def rollout_cost_grad(q: Tensor, model: int, cm: int, weights: List[float], want_pos: bool, points: int) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (cost (...), gq (..., D), link_pos (..., L | P, 3) or an empty tensor).  points = 0 or a PointSetHandle uid."""
    if points:
        pos, cost, gq = ops.rollout_points_cost_grad(_points(points), _cost(cm), weights, q, want_pos=want_pos)
    else:
        pos, cost, gq = ops.rollout_cost_grad(_model(model), _cost(cm), weights, q, want_pos=want_pos)
    return cost, gq, (pos if pos is not None else q.new_empty((0,)))


@rollout_cost_grad.register_fake
def _(q, model, cm, weights, want_pos, points):
    lead = tuple(q.shape[:-1])
    io = q.dtype if q.dtype == torch.float16 and not points else torch.float32
    n_cols = _points(points).n_points if points else _model(model).n_links
    pos = q.new_empty(lead + (n_cols, 3), dtype=io) if want_pos else q.new_empty((0,))
    return q.new_empty(lead, dtype=torch.float32), q.new_empty(tuple(q.shape), dtype=io), pos


@torch.library.custom_op("trk::scale_rows", mutates_args=(), device_types="cuda")
def scale_rows(g: Tensor, scale: Tensor) -> Tensor:
    return ops.scale_rows(g, scale)


@scale_rows.register_fake
def _(g, scale):
    return torch.empty_like(g)


def _ro_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])        # d cost / d q, produced by the forward kernel
    ctx.set_materialize_grads(False)
    # gq and link_pos are by-products of the fused kernel: flagged non-differentiable (requires_grad False; differentiating through them
    # alone raises) instead of receiving made-up zero gradients -- fk_map_collision gives differentiable link positions
    ctx.mark_non_differentiable(output[1], output[2])


def _ro_bwd(ctx, gcost, _ggq, _gpos):
    # only `cost` carries a gradient back to q (gq and link_pos are by-products: the reference's autograd graph would not
    # differentiate its own gradient either -- no double backward)
    (gq,) = ctx.saved_tensors
    if gcost is None:
        return None, None, None, None, None, None
    return torch.ops.trk.scale_rows(gq, gcost), None, None, None, None, None


rollout_cost_grad.register_autograd(_ro_bwd, setup_context=_ro_setup)
