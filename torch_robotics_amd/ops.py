"""Raw operators: torch CUDA(ROCm) tensors in, torch tensors out, arithmetic in libtrk.so.

torch is used here for device memory, streams and autograd bookkeeping only; every
floating-point operation of the path happens in the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import weakref
from typing import Optional, Sequence

import numpy as np
import torch

from . import _abi
from . import _lib as _lib_mod
from ._lib import check, lib
from .costmodel import CostModelSpec
from .kinmodel import KinModel


# Host time matters for the small ops: a kernel of ~10 us is host-bound when its wrapper costs more.  Measured pieces
# (tools/wrapper_overhead.py): torch.cuda.current_stream(dev).cuda_stream 1.9 us, `with _on(dev)` 1.35 us,
# torch.empty 1.65 us each.  The two helpers below cost ~0.3 us each when the tensor's device is the current one.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # what torch's own kernel launchers use


def _stream_of(device: torch.device) -> int:
    """HIP stream handle of torch's current stream on `device`."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream(t: torch.Tensor) -> int:
    return _stream_of(t.device)


class _on:
    """`with _on(device):` -- torch.cuda.device(device), free when that device is already the current one."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = torch.device(device).index if not isinstance(device, torch.device) else device.index
        self.prev = -1

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                torch.cuda.set_device(self.idx)
                self.prev = cur
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
            self.prev = -1
        return False


def compute_device(device=None) -> torch.device:
    """The GPU that computes for tensors living on `device`: a CUDA device is itself; the HOST ('cpu' -- the reference's default,
    robot_tree.py:77, examples/forward_kinematics.py:15) maps to the current GPU.  There is no CPU compute path: host tensors are
    copied to that GPU, the HIP kernels run, the results are copied back (`host_round_trip`); without a GPU this raises."""
    d = torch.device("cpu" if device is None else device)
    if d.type == "cuda":
        return d
    if not torch.cuda.is_available():
        raise _lib_mod.TrkError("no HIP device visible: torch_robotics_amd computes on the GPU only -- tensors on the host are copied to it "
                                "and back, there is no CPU compute path (libtrk.so needs an MI355X / gfx950 GPU)")
    return torch.device("cuda", torch.cuda.current_device())


def _map_tensors(obj, fn):
    """fn applied to every tensor in a (nested) tuple / list / dict; `Frame`-like objects (rotation / translation pairs) are rebuilt."""
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, tuple):
        return tuple(_map_tensors(v, fn) for v in obj)
    if isinstance(obj, list):
        return [_map_tensors(v, fn) for v in obj]
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if hasattr(obj, "_rot") and hasattr(obj, "_trans") and hasattr(obj, "batch_size"):          # kinematics.Frame
        return type(obj)(fn(obj._rot), fn(obj._trans))
    return obj


def host_round_trip(fn):
    """Decorator of the reference-facing entry points.  The reference runs wherever its tensors live and defaults to the host
    (`device='cpu'`); this engine computes on the GPU only.  A call that receives HOST tensors is therefore a transparent round trip:
    the tensors are copied to `compute_device()`, the call runs there on the HIP kernels, and every tensor of the result is copied
    back to the host.  Both copies are ordinary differentiable torch ops, so `.backward()` of a host result fills the host leaf's
    `.grad`.  Calls on GPU tensors pass straight through (one isinstance test per argument).  This is a transport convenience --
    PCIe-bound, never what `bench.py` measures -- and NOT a CPU compute path: without a GPU `compute_device` raises."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        host = False
        for v in args:
            if isinstance(v, torch.Tensor) and v.is_cpu:
                host = True
                break
        if not host and kwargs:
            for v in kwargs.values():
                if isinstance(v, torch.Tensor) and v.is_cpu:
                    host = True
                    break
        if not host or not torch.cuda.is_available():
            # (no GPU: the call's own argument checks come first -- the reference's exception types --, then the raw ops refuse host tensors)
            return fn(*args, **kwargs)
        dev = compute_device("cpu")
        up = lambda t: t.to(dev) if t.is_cpu else t          # noqa: E731
        out = fn(*[_map_tensors(a, up) if isinstance(a, (torch.Tensor, tuple, list)) else a for a in args],
                 **{k: (_map_tensors(v, up) if isinstance(v, (torch.Tensor, tuple, list)) else v) for k, v in kwargs.items()})
        return _map_tensors(out, lambda t: t.cpu())
    return wrapper


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; torch_robotics_amd computes only on the GPU "
                           f"(there is no CPU path)")
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def _check_q_dofs(q: torch.Tensor, n_dofs: int, what: str) -> None:
    """The C ABI sees only a pointer plus (B, H): a q whose last dimension is not the model's DOF count (a full state with
    velocities, another robot's trajectory) would be read with the wrong stride, so it is refused here."""
    if q.dim() == 0 or int(q.shape[-1]) != int(n_dofs):
        raise ValueError(f"{what}: last dimension of q is {tuple(q.shape)[-1] if q.dim() else '()'}, the model has "
                         f"{n_dofs} DOF (pass robot.get_position(x) for a state with velocities)")


def _check_buffer(t: Optional[torch.Tensor], shape_numel: int, dtype, device, what: str, at_least: bool = False) -> None:
    """Caller-provided output buffer: right device, dtype, size, and contiguous (the kernels write raw pointers)."""
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or t.device != device:
        raise ValueError(f"{what}: must be a tensor on {device}")
    if t.dtype != dtype or not t.is_contiguous() or (t.numel() < shape_numel if at_least else t.numel() != shape_numel):
        raise ValueError(f"{what}: expected a contiguous {dtype} tensor of {shape_numel} elements, got "
                         f"{t.dtype} {tuple(t.shape)}{'' if t.is_contiguous() else ' (non-contiguous)'}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# Integer ids of the live handles, for the dispatcher ops (torch.ops.trk.*, custom_ops.py): an op schema carries tensors and
# scalars only, so a model / cost model / point set travels as the value of its C pointer.
_handles: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()


def _register_handle(obj) -> int:
    uid = int(obj._h.value)
    _handles[uid] = obj
    return uid


def handle_of(uid: int, kind=None):
    """The live ModelHandle / CostHandle / PointSetHandle behind an integer id (`handle.uid`).  `kind`: the class the caller
    expects -- an id is the value of a C pointer, and a freed handle's address can be handed out again to an object of another
    kind; a graph compiled against the old id must fail loudly, not run the wrong tables."""
    try:
        h = _handles[int(uid)]
    except KeyError:
        raise ValueError(f"torch.ops.trk: {uid} is not a live model / cost-model / point-set handle") from None
    if kind is not None and not isinstance(h, kind):
        raise ValueError(f"torch.ops.trk: handle {uid} is a {type(h).__name__}, expected a {kind.__name__}")
    return h


class ModelHandle:
    """Owns a TrkModel* (device copy of the kinematic tables)."""

    def __init__(self, kin: KinModel):
        self.kin = kin
        desc, keep = _abi.kin_desc(kin)
        h = C.c_void_p()
        check(lib().trk_model_create(C.byref(desc), C.byref(h)), "trk_model_create")
        self._h = h
        self.ptr = int(h.value)           # the C handle as an integer: what the native dispatcher ops take
        self.n_links, self.n_dofs = kin.n_links, kin.n_dofs
        self.uid = _register_handle(self)

    def set_base_pose(self, R: np.ndarray, t: np.ndarray) -> None:
        R = np.ascontiguousarray(R, np.float32).reshape(9)
        t = np.ascontiguousarray(t, np.float32).reshape(3)
        check(lib().trk_model_set_base_pose(self._h, R.ctypes.data, t.ctypes.data), "trk_model_set_base_pose")

    @property
    def specialized(self) -> bool:
        return bool(lib().trk_model_is_specialized(self._h))

    def enable_specialized(self, enable: bool) -> None:
        check(lib().trk_model_enable_specialized(self._h, int(bool(enable))), "trk_model_enable_specialized")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().trk_model_destroy(h)
            except Exception:
                pass


class CostHandle:
    """Owns a TrkCostModel* (device copy of the objective tables, incl. a tiled snapshot of the voxel grid)."""

    def __init__(self, spec: CostModelSpec, device):
        spec.validate()
        self.spec = spec
        self.device = torch.device(device)
        grid_ptrs = None
        self._grid = None
        if spec.grid is not None:
            sdf = torch.as_tensor(spec.grid["sdf"], dtype=torch.float32).to(self.device).contiguous()
            grad = torch.as_tensor(spec.grid["grad"], dtype=torch.float32).to(self.device).contiguous()
            self._grid = (sdf, grad)
            grid_ptrs = (sdf.data_ptr(), grad.data_ptr())
        desc, keep = _abi.cost_desc(spec, grid_ptrs)
        h = C.c_void_p()
        with _on(self.device):
            if self._grid is not None:
                # trk_cost_model_create snapshots the grid into its own record table on the NULL stream: the uploads above ran on
                # torch's current stream, which may be a non-blocking side stream -- wait for them first
                torch.cuda.current_stream(self.device).synchronize()
            check(lib().trk_cost_model_create(C.byref(desc), C.byref(h)), "trk_cost_model_create")
        self._grid = None          # the kernels read the handle's own copy: do not keep 2 x the grid in HBM
        self._h = h
        self.ptr = int(h.value)
        self.n_links_in = spec.n_links_in
        self.n_objects = len(spec.objects)
        self.uid = _register_handle(self)

    def set_ee_target(self, H) -> None:
        H = np.ascontiguousarray(np.asarray(H, np.float32).reshape(16))
        check(lib().trk_cost_model_set_ee_target(self._h, H.ctypes.data), "trk_cost_model_set_ee_target")
        self.spec.ee_target = H.reshape(4, 4).copy()

    def set_ee2_target(self, H) -> None:
        H = np.ascontiguousarray(np.asarray(H, np.float32).reshape(16))
        check(lib().trk_cost_model_set_ee2_target(self._h, H.ctypes.data), "trk_cost_model_set_ee2_target")
        self.spec.ee2_target = H.reshape(4, 4).copy()

    def enable_specialized(self, on: bool) -> None:
        """May `cost_fields` on this cost model run a generated unit's field kernel (default) or only the table-driven one."""
        check(lib().trk_cost_model_enable_specialized(self._h, int(bool(on))), "trk_cost_model_enable_specialized")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().trk_cost_model_destroy(h)
            except Exception:
                pass


class PointSetHandle:
    """Owns a TrkPointSet*: points rigidly attached to links of one model (link index + offset in the link frame)."""

    def __init__(self, model: ModelHandle, point_link, point_offset, device):
        pl = np.ascontiguousarray(point_link, np.int32).reshape(-1)
        po = np.ascontiguousarray(point_offset, np.float32).reshape(-1, 3)
        if pl.shape[0] != po.shape[0]:
            raise ValueError("PointSetHandle: point_link and point_offset disagree on the number of points")
        self.model, self.point_link, self.point_offset = model, pl, po
        self.n_points = int(pl.shape[0])
        h = C.c_void_p()
        with _on(torch.device(device)):
            check(lib().trk_point_set_create(model._h, pl.ctypes.data, po.ctypes.data, self.n_points, C.byref(h)),
                  "trk_point_set_create")
        self._h = h
        self.uid = _register_handle(self)

    @property
    def specialized(self) -> bool:
        return bool(lib().trk_point_set_is_specialized(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().trk_point_set_destroy(h)
            except Exception:
                pass


def _sel(sel: Optional[Sequence[int]]):
    if sel is None:
        return None, 0, None
    arr = np.ascontiguousarray(sel, np.int32)
    return arr.ctypes.data, int(arr.size), arr


# ----------------------------------------------------------------------------------------------
# plain (non-autograd) calls
# ----------------------------------------------------------------------------------------------
def fk_forward(model: ModelHandle, q: torch.Tensor, sel=None) -> torch.Tensor:
    q = _dev_f32(q, "fk_forward(q)").reshape(-1, model.n_dofs)
    n = q.shape[0]
    p, ns, keep = _sel(sel)
    ncol = ns if p else model.n_links
    H = torch.empty((n, ncol, 4, 4), device=q.device, dtype=torch.float32)
    with _on(q.device):
        check(lib().trk_fk_forward(model._h, q.data_ptr(), n, p, ns, H.data_ptr(), _stream(q)), "trk_fk_forward")
    return H


def fk_positions(model: ModelHandle, q: torch.Tensor, sel=None) -> torch.Tensor:
    q = _dev_f32(q, "fk_positions(q)").reshape(-1, model.n_dofs)
    n = q.shape[0]
    p, ns, keep = _sel(sel)
    ncol = ns if p else model.n_links
    pos = torch.empty((n, ncol, 3), device=q.device, dtype=torch.float32)
    with _on(q.device):
        check(lib().trk_fk_positions(model._h, q.data_ptr(), n, p, ns, pos.data_ptr(), _stream(q)), "trk_fk_positions")
    return pos


def fk_backward(model: ModelHandle, q: torch.Tensor, gH: torch.Tensor, sel=None) -> torch.Tensor:
    q = _dev_f32(q, "fk_backward(q)").reshape(-1, model.n_dofs)
    gH = _dev_f32(gH, "fk_backward(gH)")
    n = q.shape[0]
    p, ns, keep = _sel(sel)
    gq = torch.empty_like(q)            # the kernel writes every element (zeros where the reference clamps)
    with _on(q.device):
        check(lib().trk_fk_backward(model._h, q.data_ptr(), gH.data_ptr(), n, p, ns, gq.data_ptr(), _stream(q)),
              "trk_fk_backward")
    return gq


def fk_positions_backward(model: ModelHandle, q: torch.Tensor, gpos: torch.Tensor, sel=None) -> torch.Tensor:
    q = _dev_f32(q, "fk_positions_backward(q)").reshape(-1, model.n_dofs)
    gpos = _dev_f32(gpos, "fk_positions_backward(gpos)")
    n = q.shape[0]
    p, ns, keep = _sel(sel)
    gq = torch.empty_like(q)            # the kernel writes every element (zeros where the reference clamps)
    with _on(q.device):
        check(lib().trk_fk_positions_backward(model._h, q.data_ptr(), gpos.data_ptr(), n, p, ns, gq.data_ptr(),
                                              _stream(q)), "trk_fk_positions_backward")
    return gq


def fk_points(ps: PointSetHandle, q: torch.Tensor) -> torch.Tensor:
    """q (N,D) -> world positions (N,P,3) of the attached points (robot_panda.py:154-168, frame.py:116-118)."""
    model = ps.model
    q = _dev_f32(q, "fk_points(q)").reshape(-1, model.n_dofs)
    n = q.shape[0]
    pos = torch.empty((n, ps.n_points, 3), device=q.device, dtype=torch.float32)
    with _on(q.device):
        check(lib().trk_fk_points(model._h, ps._h, q.data_ptr(), n, pos.data_ptr(), _stream(q)), "trk_fk_points")
    return pos


def fk_points_backward(ps: PointSetHandle, q: torch.Tensor, gpos: torch.Tensor) -> torch.Tensor:
    model = ps.model
    q = _dev_f32(q, "fk_points_backward(q)").reshape(-1, model.n_dofs)
    gpos = _dev_f32(gpos, "fk_points_backward(gpos)")
    gq = torch.empty_like(q)            # the kernel writes every element (zeros where the reference clamps)
    with _on(q.device):
        check(lib().trk_fk_points_backward(model._h, ps._h, q.data_ptr(), gpos.data_ptr(), q.shape[0], gq.data_ptr(),
                                           _stream(q)), "trk_fk_points_backward")
    return gq


def fk_jacobian(model: ModelHandle, q: torch.Tensor, qd: Optional[torch.Tensor], link: int, want_vel=False):
    q = _dev_f32(q, "fk_jacobian(q)").reshape(-1, model.n_dofs)
    n, D = q.shape[0], model.n_dofs
    qd_t = None if qd is None else _dev_f32(qd, "fk_jacobian(qd)").reshape(n, D)
    kw = dict(device=q.device, dtype=torch.float32)
    pos, quat = torch.empty((n, 3), **kw), torch.empty((n, 4), **kw)
    lin, ang = torch.empty((n, 3, D), **kw), torch.empty((n, 3, D), **kw)
    vl = torch.empty((n, 3), **kw) if want_vel else None
    va = torch.empty((n, 3), **kw) if want_vel else None
    with _on(q.device):
        check(lib().trk_fk_jacobian(model._h, q.data_ptr(), _ptr(qd_t), n, int(link), pos.data_ptr(), quat.data_ptr(),
                                    lin.data_ptr(), ang.data_ptr(), _ptr(vl), _ptr(va), _stream(q)), "trk_fk_jacobian")
    return (pos, quat, lin, ang, vl, va) if want_vel else (pos, quat, lin, ang)


class JacobianPlan:
    """Pre-bound `trk_fk_jacobian` (stateful FK + geometric Jacobian of one link, robot_tree.py:218-248): outputs are allocated once,
    `launch()` is one C call -- `pos` (N,3), `quat` (N,4 wxyz), `lin_jac` / `ang_jac` (N,3,D) are rewritten in place."""

    def __init__(self, model: ModelHandle, q: torch.Tensor, link: int):
        q = _dev_f32(q, "JacobianPlan(q)")
        if q.dim() != 2 or not q.is_contiguous():
            raise ValueError("JacobianPlan: q must be a contiguous (N, dof) tensor")
        _check_q_dofs(q, model.n_dofs, "JacobianPlan(q)")
        n, D = q.shape[0], model.n_dofs
        kw = dict(device=q.device, dtype=torch.float32)
        self.model, self.q, self.device = model, q, q.device
        self.pos, self.quat = torch.empty((n, 3), **kw), torch.empty((n, 4), **kw)
        self.lin_jac, self.ang_jac = torch.empty((n, 3, D), **kw), torch.empty((n, 3, D), **kw)
        self._fn = lib().trk_fk_jacobian
        self._args = (model._h, q.data_ptr(), None, n, int(link), self.pos.data_ptr(), self.quat.data_ptr(), self.lin_jac.data_ptr(),
                      self.ang_jac.data_ptr(), None, None)

    def launch(self, stream: Optional[int] = None) -> None:
        with _on(self.device):
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, stream)
        if rc:
            check(rc, "trk_fk_jacobian")


def fk_analytic_jacobian(model: ModelHandle, q: torch.Tensor) -> torch.Tensor:
    """(N, L, 7, D) Jacobian of [pos, quat_wxyz] of every link."""
    q = _dev_f32(q, "fk_analytic_jacobian(q)").reshape(-1, model.n_dofs)
    n = q.shape[0]
    J = torch.empty((n, model.n_links, 7, model.n_dofs), device=q.device, dtype=torch.float32)
    with _on(q.device):
        check(lib().trk_fk_analytic_jacobian(model._h, q.data_ptr(), n, J.data_ptr(), _stream(q)), "trk_fk_analytic_jacobian")
    return J


def _check_ik_buffers(who, model, q, lower, upper, adam_m, adam_v, loss, valid):
    """The IK kernels update q / adam_m / adam_v in place through raw pointers: refuse anything that is not exactly the
    buffer they expect (a wrong size would be an out-of-bounds device write)."""
    if not isinstance(q, torch.Tensor) or not q.is_cuda or q.dim() != 2:
        raise ValueError(f"{who}: q must be a 2-D CUDA/HIP tensor (there is no CPU path)")
    n, D = int(q.shape[0]), int(q.shape[1])
    if D != model.n_dofs:
        raise ValueError(f"{who}: q has {D} columns, the model has {model.n_dofs} DOF")
    _check_buffer(q, n * D, torch.float32, q.device, f"{who}(q)")
    for name, t in (("lower", lower), ("upper", upper)):
        if t is None:
            raise ValueError(f"{who}: {name} is required")
        _check_buffer(t, D, torch.float32, q.device, f"{who}({name})")
    _check_buffer(adam_m, n * D, torch.float32, q.device, f"{who}(adam_m)")
    _check_buffer(adam_v, n * D, torch.float32, q.device, f"{who}(adam_v)")
    _check_buffer(loss, n, torch.float32, q.device, f"{who}(loss)")
    if valid is not None and (not isinstance(valid, torch.Tensor) or valid.device != q.device or valid.numel() != n or
                              valid.element_size() != 1 or not valid.is_contiguous()):
        raise ValueError(f"{who}(valid): expected a contiguous 1-byte tensor of {n} elements on {q.device}")
    return n, D


def ik_step(model: ModelHandle, link: int, H_target: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor,
            q: torch.Tensor, adam_m: Optional[torch.Tensor], adam_v: Optional[torch.Tensor], step: int, lr: float = 1e-2,
            w_joint_limits: float = 300.0, se3_eps: float = 1e-1, loss: Optional[torch.Tensor] = None,
            valid: Optional[torch.Tensor] = None) -> None:
    """One fused IK iteration IN PLACE on q / adam_m / adam_v (all (N, D) float32 contiguous CUDA tensors)."""
    n, D = _check_ik_buffers("ik_step", model, q, lower, upper, adam_m, adam_v, loss, valid)
    Ht = _dev_f32(H_target, "ik_step(H_target)")
    per_sample = int(Ht.dim() == 3)
    if per_sample and Ht.shape[0] != n:
        raise ValueError("ik_step: per-sample target batch mismatch")
    with _on(q.device):
        check(lib().trk_ik_step(model._h, int(link), Ht.data_ptr(), per_sample, lower.data_ptr(), upper.data_ptr(),
                                float(w_joint_limits), float(se3_eps), float(lr), int(step), n, q.data_ptr(),
                                _ptr(adam_m), _ptr(adam_v), _ptr(loss), _ptr(valid), _stream(q)), "trk_ik_step")


def ik_steps(model: ModelHandle, link: int, H_target: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor,
             q: torch.Tensor, adam_m: torch.Tensor, adam_v: torch.Tensor, first_step: int, n_steps: int, lr: float = 1e-2,
             w_joint_limits: float = 300.0, se3_eps: float = 1e-1, loss: Optional[torch.Tensor] = None,
             valid: Optional[torch.Tensor] = None) -> None:
    """`n_steps` fused IK iterations IN PLACE (one launch per 32 iterations; the configurations stay on the chip in between).
    Equal to n_steps calls of ik_step, except that loss / valid describe q as passed in."""
    n, D = _check_ik_buffers("ik_steps", model, q, lower, upper, adam_m, adam_v, loss, valid)
    if adam_m is None or adam_v is None:
        raise ValueError("ik_steps: adam_m and adam_v are required")
    Ht = _dev_f32(H_target, "ik_steps(H_target)")
    per_sample = int(Ht.dim() == 3)
    if per_sample and Ht.shape[0] != n:
        raise ValueError("ik_steps: per-sample target batch mismatch")
    with _on(q.device):
        check(lib().trk_ik_steps(model._h, int(link), Ht.data_ptr(), per_sample, lower.data_ptr(), upper.data_ptr(),
                                 float(w_joint_limits), float(se3_eps), float(lr), int(first_step), int(n_steps), n, q.data_ptr(),
                                 adam_m.data_ptr(), adam_v.data_ptr(), _ptr(loss), _ptr(valid), _stream(q)), "trk_ik_steps")


def ik_gn_steps(model: ModelHandle, link: int, H_target: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor, q: torch.Tensor,
                n_steps: int, damping: float = 1e-4, lm_gain: float = 0.1, step_scale: float = 1.0, se3_eps: float = 1e-1,
                err: Optional[torch.Tensor] = None, valid: Optional[torch.Tensor] = None) -> None:
    """`n_steps` damped Gauss-Newton (Levenberg-Marquardt) IK iterations IN PLACE, one launch (include/trk.h: trk_ik_gn_steps):
    q (N,D) <- clamp(q + step_scale (J^T J + (damping + lm_gain |r|^2) I)^-1 J^T r, lower, upper) per iteration, the Jacobian in
    registers.  err (N,) / valid (N, bool or uint8): SE3 distance / validity of q as passed in."""
    n, D = _check_ik_buffers("ik_gn_steps", model, q, lower, upper, None, None, err, valid)
    Ht = _dev_f32(H_target, "ik_gn_steps(H_target)")
    per_sample = int(Ht.dim() == 3)
    if per_sample and Ht.shape[0] != n:
        raise ValueError("ik_gn_steps: per-sample target batch mismatch")
    with _on(q.device):
        check(lib().trk_ik_gn_steps(model._h, int(link), Ht.data_ptr(), per_sample, lower.data_ptr(), upper.data_ptr(), float(damping),
                                    float(lm_gain), float(step_scale), float(se3_eps), int(n_steps), n, q.data_ptr(), _ptr(err),
                                    _ptr(valid), _stream(q)), "trk_ik_gn_steps")


@host_round_trip
def rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """rotation_matrix_to_q on (..., 3, 3) rotations or (..., 4, 4) transforms -> (..., 4) wxyz."""
    R = _dev_f32(R, "rotmat_to_quat(R)")
    if R.shape[-2:] == (3, 3):
        stride, pitch = 9, 3
    elif R.shape[-2:] == (4, 4):
        stride, pitch = 16, 4
    else:
        raise ValueError("rotmat_to_quat: expected (...,3,3) or (...,4,4)")
    batch = R.shape[:-2]
    n = int(np.prod(batch)) if len(batch) else 1
    out = torch.empty(tuple(batch) + (4,), device=R.device, dtype=torch.float32)
    with _on(R.device):
        check(lib().trk_rotmat_to_quat(R.data_ptr(), n, stride, pitch, out.data_ptr(), _stream(R)), "trk_rotmat_to_quat")
    return out


class _AxisRotation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, angle):
        n = int(angle.shape[0])
        R = torch.empty((n, 3, 3), device=angle.device, dtype=torch.float32)
        with _on(angle.device):
            check(lib().trk_rotation_from(kind, angle.data_ptr(), n, R.data_ptr(), _stream(angle)), "trk_rotation_from")
        ctx.kind = kind
        ctx.save_for_backward(angle)
        return R

    @staticmethod
    def backward(ctx, gR):
        (angle,) = ctx.saved_tensors
        gR = _dev_f32(gR, "axis rotation backward")
        ga = torch.empty_like(angle)
        with _on(angle.device):
            check(lib().trk_rotation_from_backward(ctx.kind, angle.data_ptr(), gR.data_ptr(), int(angle.shape[0]), ga.data_ptr(),
                                                   _stream(angle)), "trk_rotation_from_backward")
        return None, ga


@host_round_trip
def axis_rotation(kind: int, angle: torch.Tensor) -> torch.Tensor:
    """x_rot / y_rot / z_rot (kind 0 / 1 / 2; spatial_vector.py:8-47): angles (n,) | (n,1) | () -> (n,3,3); differentiable."""
    a = _dev_f32(angle, "axis_rotation(angle)").reshape(-1)
    return _AxisRotation.apply(int(kind), a)


def _quat_to_rotmat_raw(flat: torch.Tensor) -> torch.Tensor:
    R = torch.empty((flat.shape[0], 3, 3), device=flat.device, dtype=torch.float32)
    with _on(flat.device):
        check(lib().trk_rotation_from(3, flat.data_ptr(), int(flat.shape[0]), R.data_ptr(), _stream(flat)), "trk_rotation_from")
    return R


class _QuatRotation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat):
        ctx.save_for_backward(flat)
        return _quat_to_rotmat_raw(flat)

    @staticmethod
    def backward(ctx, gR):
        (flat,) = ctx.saved_tensors
        gR = _dev_f32(gR, "quat_to_rotmat backward")
        gq = torch.empty_like(flat)
        with _on(flat.device):
            check(lib().trk_rotation_from_backward(3, flat.data_ptr(), gR.data_ptr(), int(flat.shape[0]), gq.data_ptr(),
                                                   _stream(flat)), "trk_rotation_from_backward")
        return gq


@host_round_trip
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """q_to_rotation_matrix (quaternion.py:102-120): wxyz (..., 4), not necessarily normalised -> (..., 3, 3); differentiable
    w.r.t. q like the reference's torch expression (explicit backward kernel, incl. the 2 / |q|^2 normalisation)."""
    q = _dev_f32(q, "quat_to_rotmat(q)")
    if q.dim() == 1:
        q = q.unsqueeze(0)
    lead = q.shape[:-1]
    flat = q.reshape(-1, 4).contiguous()
    if torch.is_grad_enabled() and flat.requires_grad:
        R = _QuatRotation.apply(flat)
    else:
        R = _quat_to_rotmat_raw(flat)
    return R.reshape(tuple(lead) + (3, 3))


# ----------------------------------------------------------------------------------------------------------------------
# Frame algebra (geometrics/frame.py:55-121): trk_frame_* kernels, explicit reverse mode
# ----------------------------------------------------------------------------------------------------------------------
FRAME_COMPOSE, FRAME_INVERSE, FRAME_INV_COMPOSE = 0, 1, 2


def _pose_args(R: torch.Tensor, t: torch.Tensor, what: str):
    R, t = _dev_f32(R, what + "(rot)"), _dev_f32(t, what + "(trans)")
    if R.dim() != 3 or R.shape[1:] != (3, 3) or t.dim() != 2 or t.shape[1] != 3 or t.shape[0] != R.shape[0]:
        raise ValueError(f"{what}: expected rot (n,3,3) and trans (n,3), got {tuple(R.shape)} / {tuple(t.shape)}")
    return R, t


def _frame_compose_raw(op, Ra, ta, Rb, tb):
    na = int(Ra.shape[0])
    nb = int(Rb.shape[0]) if Rb is not None else na
    if op != FRAME_INVERSE and na != nb and na != 1 and nb != 1:
        raise ValueError(f"frame algebra: batch sizes {na} and {nb} do not broadcast")
    n = na if op == FRAME_INVERSE else max(na, nb)
    Ro = torch.empty((n, 3, 3), device=Ra.device, dtype=torch.float32)
    to = torch.empty((n, 3), device=Ra.device, dtype=torch.float32)
    with _on(Ra.device):
        check(lib().trk_frame_compose(op, Ra.data_ptr(), ta.data_ptr(), na, _ptr(Rb), _ptr(tb), nb, Ro.data_ptr(),
                                      to.data_ptr(), _stream(Ra)), "trk_frame_compose")
    return Ro, to


class _FrameCompose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, Ra, ta, Rb, tb):
        ctx.op = op
        ctx.save_for_backward(Ra, ta, Rb, tb)
        return _frame_compose_raw(op, Ra, ta, Rb, tb)

    @staticmethod
    def backward(ctx, gR, gt):
        Ra, ta, Rb, tb = ctx.saved_tensors
        n = int(Ra.shape[0])
        gR = torch.zeros_like(Ra) if gR is None else _dev_f32(gR, "frame backward")
        gt = torch.zeros_like(ta) if gt is None else _dev_f32(gt, "frame backward")
        gRa, gta = torch.empty_like(Ra), torch.empty_like(ta)
        two = ctx.op != FRAME_INVERSE
        gRb, gtb = (torch.empty_like(Rb), torch.empty_like(tb)) if two else (None, None)
        with _on(Ra.device):
            check(lib().trk_frame_compose_backward(ctx.op, Ra.data_ptr(), ta.data_ptr(), _ptr(Rb) if two else None,
                                                   _ptr(tb) if two else None, gR.data_ptr(), gt.data_ptr(), n, gRa.data_ptr(),
                                                   gta.data_ptr(), _ptr(gRb), _ptr(gtb), _stream(Ra)),
                  "trk_frame_compose_backward")
        return None, gRa, gta, gRb, gtb


@host_round_trip
def frame_compose(op: int, Ra, ta, Rb=None, tb=None):
    """(rot, trans) of `a o b` (FRAME_COMPOSE), `a^-1` (FRAME_INVERSE) or `b^-1 o a` (FRAME_INV_COMPOSE); differentiable
    w.r.t. every input pose.  A batch-1 frame broadcasts (it is expanded first when a gradient has to flow into it)."""
    Ra, ta = _pose_args(Ra, ta, "frame_compose(a)")
    if op == FRAME_INVERSE:
        Rb, tb = Ra, ta                                         # placeholders for the autograd signature
    else:
        Rb, tb = _pose_args(Rb, tb, "frame_compose(b)")
    needs_grad = torch.is_grad_enabled() and any(x.requires_grad for x in (Ra, ta, Rb, tb))
    if not needs_grad:
        return _frame_compose_raw(op, Ra, ta, None if op == FRAME_INVERSE else Rb, None if op == FRAME_INVERSE else tb)
    n = max(Ra.shape[0], Rb.shape[0])
    if Ra.shape[0] != n:
        Ra, ta = Ra.expand(n, 3, 3).contiguous(), ta.expand(n, 3).contiguous()
    if Rb.shape[0] != n:
        Rb, tb = Rb.expand(n, 3, 3).contiguous(), tb.expand(n, 3).contiguous()
    return _FrameCompose.apply(op, Ra, ta, Rb, tb)


class _FrameTransformPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, t, pts):
        n, P = int(R.shape[0]), int(pts.shape[0])
        out = torch.empty((n, P, 3), device=R.device, dtype=torch.float32)
        with _on(R.device):
            check(lib().trk_frame_transform_points(R.data_ptr(), t.data_ptr(), n, pts.data_ptr(), P, out.data_ptr(), _stream(R)),
                  "trk_frame_transform_points")
        ctx.save_for_backward(pts)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        (pts,) = ctx.saved_tensors
        g = _dev_f32(g, "transform_point backward")
        gR = torch.empty((ctx.n, 3, 3), device=g.device, dtype=torch.float32)
        gt = torch.empty((ctx.n, 3), device=g.device, dtype=torch.float32)
        with _on(g.device):
            check(lib().trk_frame_transform_points_backward(g.data_ptr(), ctx.n, pts.data_ptr(), int(pts.shape[0]),
                                                            gR.data_ptr(), gt.data_ptr(), _stream(g)),
                  "trk_frame_transform_points_backward")
        return gR, gt, None


@host_round_trip
def frame_transform_points(R, t, points):
    """points (P,3) in the frames -> (n,P,3) in their parent frame; differentiable w.r.t. the poses (the reference's callers
    transform constant point sets -- grasped-object vertices -- so no gradient is produced for `points`)."""
    R, t = _pose_args(R, t, "frame_transform_points")
    pts = _dev_f32(points, "frame_transform_points(points)").reshape(-1, 3)
    if torch.is_grad_enabled() and pts.requires_grad:
        raise NotImplementedError("frame_transform_points: no gradient w.r.t. the points (constant point sets only)")
    return _FrameTransformPoints.apply(R, t, pts)


def _rot_layout(R: torch.Tensor, what: str):
    if R.shape[-2:] == (3, 3):
        stride, pitch = 9, 3
    elif R.shape[-2:] == (4, 4):
        stride, pitch = 16, 4
    else:
        raise ValueError(f"{what}: expected (...,3,3) or (...,4,4)")
    batch = R.shape[:-2]
    return stride, pitch, tuple(batch), (int(np.prod(batch)) if len(batch) else 1)


def _frame_quat_euler_raw(R, want_quat, want_euler):
    stride, pitch, batch, n = _rot_layout(R, "frame_quat_euler")
    quat = torch.empty(batch + (4,), device=R.device, dtype=torch.float32) if want_quat else None
    eul = torch.empty(batch + (3,), device=R.device, dtype=torch.float32) if want_euler else None
    with _on(R.device):
        check(lib().trk_frame_quat_euler(R.data_ptr(), n, stride, pitch, _ptr(quat), _ptr(eul), _stream(R)), "trk_frame_quat_euler")
    return quat, eul


class _FrameQuatEuler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, want_quat, want_euler):
        quat, eul = _frame_quat_euler_raw(R, want_quat, want_euler)
        ctx.save_for_backward(R)
        ctx.want = (want_quat, want_euler)
        return tuple(t for t in (quat, eul) if t is not None)

    @staticmethod
    def backward(ctx, *grads):
        (R,) = ctx.saved_tensors
        grads = list(grads)
        gq = grads.pop(0) if ctx.want[0] else None
        ge = grads.pop(0) if ctx.want[1] else None
        gq = None if gq is None else _dev_f32(gq, "frame_quat_euler backward")
        ge = None if ge is None else _dev_f32(ge, "frame_quat_euler backward")
        stride, pitch, batch, n = _rot_layout(R, "frame_quat_euler")
        g9 = torch.empty((n, 3, 3), device=R.device, dtype=torch.float32)
        with _on(R.device):
            check(lib().trk_frame_quat_euler_backward(R.data_ptr(), n, stride, pitch, _ptr(gq), _ptr(ge), g9.data_ptr(),
                                                      _stream(R)), "trk_frame_quat_euler_backward")
        if pitch == 3:
            return g9.reshape(R.shape), None, None
        gR = torch.zeros_like(R)                      # 4x4 transforms: only the rotation block receives a gradient
        gR[..., :3, :3] = g9.reshape(batch + (3, 3))
        return gR, None, None


@host_round_trip
def frame_quat_euler(R: torch.Tensor, want_quat=True, want_euler=False):
    """Frame.get_quaternion (trace method, XYZW -- frame.py:87-114) / Frame.get_euler (frame.py:120-121) of (n,3,3) rotations
    or (n,4,4) transforms.  Differentiable w.r.t. R the way the reference is: the Euler angles through atan2 / asin, the
    quaternion with its normalising scale held constant (the reference computes it with `math.sqrt`, outside autograd)."""
    R = _dev_f32(R, "frame_quat_euler(R)")
    if torch.is_grad_enabled() and R.requires_grad and (want_quat or want_euler):
        outs = list(_FrameQuatEuler.apply(R, bool(want_quat), bool(want_euler)))
        return (outs.pop(0) if want_quat else None), (outs.pop(0) if want_euler else None)
    return _frame_quat_euler_raw(R, want_quat, want_euler)


def cost_fields(cm: CostHandle, fields: int, link_pos: torch.Tensor, gcost: Optional[torch.Tensor] = None,
                want_grad: bool = False):
    link_pos = _dev_f32(link_pos, "cost_fields(link_pos)").reshape(-1, cm.n_links_in, 3)
    n = link_pos.shape[0]
    cost = torch.empty((n,), device=link_pos.device, dtype=torch.float32)
    g = torch.empty_like(link_pos) if want_grad else None
    gc = None if gcost is None else _dev_f32(gcost, "cost_fields(gcost)").reshape(n)
    with _on(link_pos.device):
        check(lib().trk_cost_fields(cm._h, int(fields), link_pos.data_ptr(), n, _ptr(gc), cost.data_ptr(), _ptr(g),
                                    _stream(link_pos)), "trk_cost_fields")
    return (cost, g) if want_grad else cost


def collision_fields(cm: CostHandle, fields: int, link_pos: torch.Tensor, margin: Optional[float] = None) -> torch.Tensor:
    link_pos = _dev_f32(link_pos, "collision_fields(link_pos)").reshape(-1, cm.n_links_in, 3)
    n = link_pos.shape[0]
    out = torch.empty((n,), device=link_pos.device, dtype=torch.bool)       # the kernel writes 0 / 1 bytes
    with _on(link_pos.device):
        check(lib().trk_collision_fields(cm._h, int(fields), link_pos.data_ptr(), n,
                                         float("nan") if margin is None else float(margin), out.data_ptr(),
                                         _stream(link_pos)), "trk_collision_fields")
    return out


def ee_cost(cm: CostHandle, H: torch.Tensor, target: Optional[torch.Tensor] = None,
            gcost: Optional[torch.Tensor] = None, want_grad: bool = False):
    """H: (N,4,4) contiguous EE transforms (or a strided view made contiguous)."""
    H = _dev_f32(H, "ee_cost(H)").reshape(-1, 4, 4)
    n = H.shape[0]
    per_sample, tgt = 0, None
    if target is not None:
        tgt = _dev_f32(target, "ee_cost(target)")
        per_sample = int(tgt.dim() == 3)
        if per_sample and tgt.shape[0] != n:
            raise ValueError("ee_cost: per-sample target batch mismatch")
    cost = torch.empty((n,), device=H.device, dtype=torch.float32)
    gH = torch.empty_like(H) if want_grad else None      # all 16 entries are written (bottom row: zeros)
    gc = None if gcost is None else _dev_f32(gcost, "ee_cost(gcost)").reshape(n)
    with _on(H.device):
        check(lib().trk_ee_cost(cm._h, H.data_ptr(), n, 16, _ptr(tgt), per_sample, _ptr(gc), cost.data_ptr(),
                                _ptr(gH), 16, _stream(H)), "trk_ee_cost")
    return (cost, gH) if want_grad else cost


_weights_cache: dict = {}


def _weights_struct(weights):
    """TrkRolloutWeights for a 4-tuple of floats (read-only on the C side, so one struct per distinct tuple is kept)."""
    key = tuple(float(v) for v in weights)
    w = _weights_cache.get(key)
    if w is None:
        if len(_weights_cache) > 256:
            _weights_cache.clear()
        w = _weights_cache[key] = _abi.RolloutWeights(*key)
    return w


def _grad_mode(f16: bool, grad_dtype, grad_scale, who: str):
    """(torch dtype of the gradient, TRK_F32 / TRK_F16, scale) of a reduced-precision call; fp32 trajectories take no options."""
    if grad_dtype is None:
        grad_dtype = torch.float16 if f16 else torch.float32
    if grad_dtype not in (torch.float16, torch.float32) or (not f16 and grad_dtype != torch.float32):
        raise ValueError(f"{who}: grad_dtype must be torch.float32, or torch.float16 with float16 trajectories")
    gs = float(grad_scale)
    if not (gs > 0.0 and math.isfinite(gs)):
        raise ValueError(f"{who}: grad_scale must be a finite positive number")
    return grad_dtype, int(grad_dtype == torch.float16), gs


def gp_grad_scale(dt: float, sigma: float, weight: float = 1.0, q_abs_max: float = 3.0, qd_abs_max: float = 2.5,
                  extra: float = 0.0, limit: float = 32768.0) -> float:
    """A power-of-two loss scale that keeps an fp16 gradient of the GP prior (+ `extra`: a bound on the other terms' gradient)
    below `limit` in the worst case of trajectories bounded by |q| <= q_abs_max, |qd| <= qd_abs_max: with
    a = 12 / (sigma^2 dt^3), b = 6 / (sigma^2 dt^2), c = 4 / (sigma^2 dt) and residual bounds ep = 2 q_abs_max + dt qd_abs_max,
    ev = 2 qd_abs_max, the two neighbours of a time step give |d/dq| <= 2 (a ep + b ev) and |d/dqd| <= 2 (b ep + c ev) + dt (a ep + b ev)."""
    s2 = 1.0 / (sigma * sigma)
    a, b, c = 12.0 * s2 / dt ** 3, 6.0 * s2 / dt ** 2, 4.0 * s2 / dt
    ep, ev = 2.0 * q_abs_max + dt * qd_abs_max, 2.0 * qd_abs_max
    bound = abs(weight) * max(2.0 * (a * ep + b * ev), 2.0 * (b * ep + c * ev) + dt * (a * ep + b * ev)) + abs(extra)
    return 2.0 ** min(0, math.floor(math.log2(limit / bound))) if bound > 0.0 else 1.0


def rollout_cost_grad(model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, want_pos: bool = True,
                      cost_sum: Optional[torch.Tensor] = None, out=None, grad_dtype=None, grad_scale: float = 1.0):
    """q (B,H,D) or (N,D) -> (link_pos (…,L,3) or None, cost (…), gq (…,D)).
    A float16 q selects the fp16-I/O kernel: link_pos comes back as float16, cost stays float32, and gq is
    `grad_scale * d cost / d q` in `grad_dtype` (float16 by default, saturating at +-65504 instead of inf; float32 = the mixed mode)."""
    f16 = q.dtype == torch.float16
    if f16:
        if q.device.type != "cuda":
            raise ValueError("rollout_cost_grad(q): expected a CUDA/HIP tensor (there is no CPU path)")
        q = q.contiguous()
    else:
        q = _dev_f32(q, "rollout_cost_grad(q)")
    io = torch.float16 if f16 else torch.float32
    gio, gcode, gs = _grad_mode(f16, grad_dtype, grad_scale, "rollout_cost_grad")
    if not f16 and gs != 1.0:
        raise ValueError("rollout_cost_grad: grad_scale applies to float16 trajectories only")
    _check_q_dofs(q, model.n_dofs, "rollout_cost_grad(q)")
    lead = q.shape[:-1]
    if q.dim() == 3:
        B, Hh = int(q.shape[0]), int(q.shape[1])
    else:
        q = q.reshape(-1, model.n_dofs)
        B, Hh = int(q.shape[0]), 1
    n, L, D = B * Hh, model.n_links, model.n_dofs
    if out is None:         # allocated in their final shapes: a reshape of a fresh tensor is ~0.7 us of host time each
        lt = tuple(lead)
        pos = torch.empty(lt + (L, 3), device=q.device, dtype=io) if want_pos else None
        cost = torch.empty(lt, device=q.device, dtype=torch.float32)
        gq = torch.empty(lt + (D,), device=q.device, dtype=gio)
    else:
        pos, cost, gq = out
        _check_buffer(pos, n * L * 3, io, q.device, "rollout_cost_grad(out[0] = link_pos)")
        _check_buffer(cost, n, torch.float32, q.device, "rollout_cost_grad(out[1] = cost)")
        _check_buffer(gq, n * D, gio, q.device, "rollout_cost_grad(out[2] = gq)")
        if cost is None or gq is None:
            raise ValueError("rollout_cost_grad(out): cost and gq buffers are required (link_pos may be None)")
    _check_buffer(cost_sum, n_blocks(n), torch.float32, q.device, "rollout_cost_grad(cost_sum)", at_least=True)
    w = _weights_struct(weights)
    with _on(q.device):
        if f16:
            check(lib().trk_rollout_cost_grad_f16(model._h, cm._h, C.byref(w), q.data_ptr(), B, Hh, _ptr(pos), cost.data_ptr(),
                                                  gq.data_ptr(), gcode, gs, _ptr(cost_sum), _stream(q)), "trk_rollout_cost_grad_f16")
        else:
            check(lib().trk_rollout_cost_grad(model._h, cm._h, C.byref(w), q.data_ptr(), B, Hh, _ptr(pos), cost.data_ptr(),
                                              gq.data_ptr(), _ptr(cost_sum), _stream(q)), "trk_rollout_cost_grad")
    if out is None:
        return pos, cost, gq
    return (None if pos is None else pos.reshape(tuple(lead) + (L, 3)), cost.reshape(tuple(lead)),
            gq.reshape(tuple(lead) + (D,)))


def rollout_is_specialized(model: ModelHandle, cm: CostHandle, weights) -> bool:
    """True when `rollout_cost_grad(model, cm, weights, ...)` runs a generated kernel, False when the table-driven one would."""
    return bool(lib().trk_rollout_is_specialized(model._h, cm._h, C.byref(_weights_struct(weights))))


DISPATCH_NAMES = {0: "none", 1: "generated", 2: "table-driven", 3: "generated + prior launches"}


def rollout_points_is_specialized(ps: "PointSetHandle", cm: CostHandle, weights) -> bool:
    """The same question for `rollout_points_cost_grad(ps, cm, weights, ...)`."""
    return bool(lib().trk_rollout_points_is_specialized(ps._h, cm._h, C.byref(_weights_struct(weights))))


def _require_generated(who: str, model: ModelHandle, served: bool) -> None:
    """strict=True of the pre-bound plans: a model that has generated kernels must not be bound to the table-driven one silently."""
    if model.specialized and not served:
        raise NotImplementedError(
            f"{who}(strict=True): the model has generated kernels, but none bakes this cost model's link sets for the non-zero weights -- "
            "every launch of this plan would take the table-driven kernel (10 - 30 x slower).  Match the robot's collision template, "
            "compile a unit for this one (jit.specialize / PlanningTask.specialize), or pass strict=False.")


def last_dispatch() -> str:
    """Which kernel family served this thread's latest rollout call (`trk_last_dispatch`): 'generated', 'table-driven',
    'generated + prior launches' (the two-launch form of the GP-fused rollout) or 'none'."""
    return DISPATCH_NAMES[int(lib().trk_last_dispatch())]


def set_strict_specialized(on: bool) -> bool:
    """Strict mode (`trk_set_strict_specialized`; TRK_STRICT_SPECIALIZED=1 sets it from the environment): a rollout call on a model
    that HAS generated kernels raises NotImplementedError instead of silently taking the 10 - 30 x slower table-driven kernel when
    no unit bakes the cost model's link sets.  Returns the previous setting."""
    return bool(lib().trk_set_strict_specialized(1 if on else 0))


class strict_specialized:
    """`with ops.strict_specialized():` -- strict mode for a block (bench.py and RolloutPlan bind their launches under it)."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = set_strict_specialized(self.on)
        return self

    def __exit__(self, *exc):
        set_strict_specialized(self.prev)
        return False


def rollout_collision(model: ModelHandle, cm: CostHandle, fields: int, q: torch.Tensor, margin: Optional[float] = None) -> torch.Tensor:
    """Fused FK + boolean collision fields: q (B,H,D) or (N,D) -> bool (B,H) / (N,).  One launch, one byte per sample out
    (`PlanningTask.compute_collision`, tasks.py:131-133); `margin=None` uses the fields' own margins."""
    q = _dev_f32(q, "rollout_collision(q)")
    _check_q_dofs(q, model.n_dofs, "rollout_collision(q)")
    lead = q.shape[:-1]
    if q.dim() == 3:
        B, Hh = int(q.shape[0]), int(q.shape[1])
    else:
        q = q.reshape(-1, model.n_dofs)
        B, Hh = int(q.shape[0]), 1
    n = B * Hh
    out = torch.empty((n,), device=q.device, dtype=torch.bool)      # the kernel writes 0 / 1 bytes
    # scratch for the table-driven fallback only (a model / cost model that no generated kernel serves)
    ws = None if model.specialized else torch.empty((n, model.n_links, 3), device=q.device, dtype=torch.float32)
    m = float("nan") if margin is None else float(margin)
    with _on(q.device):
        rc = lib().trk_rollout_collision(model._h, cm._h, int(fields), q.data_ptr(), B, Hh, m, out.data_ptr(), _ptr(ws), _stream(q))
        if rc == _abi.TRK_ERR_INVALID_ARG and ws is None:       # a unit exists for the model, but not for this cost model
            ws = torch.empty((n, model.n_links, 3), device=q.device, dtype=torch.float32)
            rc = lib().trk_rollout_collision(model._h, cm._h, int(fields), q.data_ptr(), B, Hh, m, out.data_ptr(), ws.data_ptr(), _stream(q))
        check(rc, "trk_rollout_collision")
    return out.reshape(tuple(lead))


def rollout_points_cost_grad(ps: PointSetHandle, cm: CostHandle, weights, q: torch.Tensor, want_pos: bool = True,
                             cost_sum: Optional[torch.Tensor] = None):
    """Fused rollout with the collision fields on attached points: q (B,H,D) or (N,D) ->
    (point_pos (…,P,3) or None, cost (…), gq (…,D))."""
    model = ps.model
    q = _dev_f32(q, "rollout_points_cost_grad(q)")
    _check_q_dofs(q, model.n_dofs, "rollout_points_cost_grad(q)")
    lead = q.shape[:-1]
    if q.dim() == 3:
        B, Hh = int(q.shape[0]), int(q.shape[1])
    else:
        q = q.reshape(-1, model.n_dofs)
        B, Hh = int(q.shape[0]), 1
    n, P, D = B * Hh, ps.n_points, model.n_dofs
    pos = torch.empty((n, P, 3), device=q.device, dtype=torch.float32) if want_pos else None
    cost = torch.empty((n,), device=q.device, dtype=torch.float32)
    gq = torch.empty((n, D), device=q.device, dtype=torch.float32)
    _check_buffer(cost_sum, n_blocks(n), torch.float32, q.device, "rollout_points_cost_grad(cost_sum)", at_least=True)
    w = _abi.RolloutWeights(*[float(v) for v in weights])
    with _on(q.device):
        check(lib().trk_rollout_points_cost_grad(model._h, ps._h, cm._h, C.byref(w), q.data_ptr(), B, Hh, _ptr(pos),
                                                 cost.data_ptr(), gq.data_ptr(), _ptr(cost_sum), _stream(q)),
              "trk_rollout_points_cost_grad")
    return (None if pos is None else pos.reshape(tuple(lead) + (P, 3)), cost.reshape(tuple(lead)),
            gq.reshape(tuple(lead) + (D,)))


def gp_prior_cost_grad(q: torch.Tensor, qd: torch.Tensor, dt: float, sigma: float, weight: float = 1.0,
                       accumulate_into=None, grad_dtype=None, grad_scale: float = 1.0):
    """Constant-velocity GP prior over (B,H,D) trajectories (build-defined; include/trk.h): -> (cost (B,), gq, gqd).
    fp32 or fp16 tensors (fp32 arithmetic, fp32 cost).  gq / gqd hold `grad_scale` x the gradient in `grad_dtype` (the dtype of q
    by default; float32 with float16 trajectories = the mixed mode); a float16 gradient saturates at +-65504 instead of inf --
    `gp_grad_scale(dt, sigma, ...)` picks a scale that keeps it finite.  accumulate_into=(gq, gqd) adds into existing buffers
    (which hold gradients of the same scale)."""
    if q.device.type != "cuda" or qd.device != q.device:
        raise ValueError("gp_prior_cost_grad: q and qd must be tensors on the same GPU (there is no CPU path)")
    if q.dim() != 3 or qd.shape != q.shape or q.dtype != qd.dtype or q.dtype not in (torch.float32, torch.float16):
        raise ValueError("gp_prior_cost_grad: q, qd must be (batch, horizon, dof) of the same fp32 / fp16 dtype")
    q, qd = q.contiguous(), qd.contiguous()
    B, H, D = (int(v) for v in q.shape)
    if accumulate_into is not None and grad_dtype is None:
        grad_dtype = accumulate_into[0].dtype
    gio, gcode, gs = _grad_mode(q.dtype == torch.float16, grad_dtype, grad_scale, "gp_prior_cost_grad")
    cost = torch.empty((B,), device=q.device, dtype=torch.float32)
    if accumulate_into is None:
        gq, gqd, acc = torch.empty_like(q, dtype=gio), torch.empty_like(q, dtype=gio), 0
    else:
        gq, gqd = accumulate_into
        acc = 1
        if gq.shape != q.shape or gqd.shape != q.shape or gq.dtype != gio or gqd.dtype != gio or \
                not (gq.is_contiguous() and gqd.is_contiguous()):
            raise ValueError("gp_prior_cost_grad: accumulate_into buffers must match q (shape, contiguous) and grad_dtype")
    with _on(q.device):
        check(lib().trk_gp_prior_cost_grad(q.data_ptr(), qd.data_ptr(), B, H, D, int(q.dtype == torch.float16), float(dt),
                                           float(sigma), float(weight), cost.data_ptr(), gq.data_ptr(), gqd.data_ptr(), gcode, gs,
                                           acc, _stream(q)), "trk_gp_prior_cost_grad")
    return cost, gq, gqd


def _gp_args(model, q, qd, who):
    if q.device.type != "cuda" or qd.device != q.device:
        raise ValueError(f"{who}: q and qd must be tensors on the same GPU (there is no CPU path)")
    if q.dim() != 3 or qd.shape != q.shape or q.dtype != qd.dtype or q.dtype not in (torch.float32, torch.float16) or \
            not (q.is_contiguous() and qd.is_contiguous()):
        raise ValueError(f"{who}: q, qd must be contiguous (batch, horizon, dof) tensors of the same fp32 / fp16 dtype")
    _check_q_dofs(q, model.n_dofs, f"{who}(q)")
    return (int(v) for v in q.shape)


class RolloutGpPlan:
    """BASELINE config 5's objective as ONE pre-bound launch (include/trk.h: trk_rollout_gp_cost_grad): fused FK + collision / EE
    objectives + the constant-velocity GP prior + both gradients.  q, qd (B,H,D) fp32 or fp16 are read in place on every `launch()`;
    `cost` (B,H) fp32 is the rollout's cost plus the prior's factor t -> t+1 at sample t, `gq` / `gqd` hold grad_scale x the gradients
    in grad_dtype (fp16 by default for fp16 trajectories, saturating), `link_pos` (B,H,L,3) the positions (want_pos)."""

    def __init__(self, model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, qd: torch.Tensor, dt: float, sigma: float,
                 gp_weight: float = 1.0, want_pos: bool = True, grad_dtype=None, grad_scale: float = 1.0, strict: bool = True):
        self.generated = rollout_is_specialized(model, cm, weights)
        if strict:
            _require_generated("RolloutGpPlan", model, self.generated)
        B, H, D = _gp_args(model, q, qd, "RolloutGpPlan")
        f16 = q.dtype == torch.float16
        gio, gcode, gs = _grad_mode(f16, grad_dtype, grad_scale, "RolloutGpPlan")
        if not f16 and gs != 1.0:
            raise ValueError("RolloutGpPlan: grad_scale applies to float16 trajectories only")
        self.model, self.cm, self.q, self.qd, self.device, self.grad_scale = model, cm, q, qd, q.device, gs
        self.B, self.H = B, H
        self.link_pos = torch.empty((B, H, model.n_links, 3), device=q.device, dtype=q.dtype) if want_pos else None
        self.cost = torch.empty((B, H), device=q.device, dtype=torch.float32)
        self.gq = torch.empty((B, H, D), device=q.device, dtype=gio)
        self.gqd = torch.empty((B, H, D), device=q.device, dtype=gio)
        self._w = _abi.RolloutWeights(*[float(v) for v in weights])
        self._gp = _abi.GpPrior(float(dt), float(sigma), float(gp_weight))
        self._fn = lib().trk_rollout_gp_cost_grad
        self._args = (model._h, cm._h, C.byref(self._w), C.byref(self._gp), q.data_ptr(), qd.data_ptr(), B, H, int(f16), _ptr(self.link_pos),
                      self.cost.data_ptr(), self.gq.data_ptr(), self.gqd.data_ptr(), gcode, gs)

    def launch(self, cost_sum_ptr: Optional[int] = None, stream: Optional[int] = None) -> None:
        with _on(self.device):
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, cost_sum_ptr, stream)
        if rc:
            check(rc, "trk_rollout_gp_cost_grad")


def rollout_gp_cost_grad(model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, qd: torch.Tensor, dt: float, sigma: float,
                         gp_weight: float = 1.0, want_pos: bool = True, cost_sum: Optional[torch.Tensor] = None, grad_dtype=None,
                         grad_scale: float = 1.0):
    """One fused launch: -> (link_pos or None, cost (B,H), gq, gqd); see RolloutGpPlan."""
    plan = RolloutGpPlan(model, cm, weights, q.contiguous(), qd.contiguous(), dt, sigma, gp_weight, want_pos, grad_dtype, grad_scale)
    _check_buffer(cost_sum, n_blocks(plan.B * plan.H), torch.float32, q.device, "rollout_gp_cost_grad(cost_sum)", at_least=True)
    plan.launch(_ptr(cost_sum))
    return plan.link_pos, plan.cost, plan.gq, plan.gqd


class GPPriorPlan:
    """Pre-bound GP-prior launch (the counterpart of `RolloutPlan`): buffers and arguments are resolved once, `launch()` is one
    C call (~3 us of host time instead of ~12 us through `gp_prior_cost_grad`).  q, qd (B,H,D) are read in place on every launch;
    results land in `cost` (B,), `gq`, `gqd` -- or are ADDED into `accumulate_into=(gq, gqd)`, e.g. a RolloutPlan's `gq`."""

    def __init__(self, q: torch.Tensor, qd: torch.Tensor, dt: float, sigma: float, weight: float = 1.0, accumulate_into=None,
                 grad_dtype=None, grad_scale: float = 1.0):
        if q.device.type != "cuda" or qd.device != q.device:
            raise ValueError("GPPriorPlan: q and qd must be tensors on the same GPU (there is no CPU path)")
        if q.dim() != 3 or qd.shape != q.shape or q.dtype != qd.dtype or q.dtype not in (torch.float32, torch.float16) or \
                not (q.is_contiguous() and qd.is_contiguous()):
            raise ValueError("GPPriorPlan: q, qd must be contiguous (batch, horizon, dof) tensors of the same fp32 / fp16 dtype")
        B, H, D = (int(v) for v in q.shape)
        if accumulate_into is not None and grad_dtype is None:
            grad_dtype = accumulate_into[0].dtype
        gio, gcode, gs = _grad_mode(q.dtype == torch.float16, grad_dtype, grad_scale, "GPPriorPlan")
        self.q, self.qd, self.device, self.grad_scale = q, qd, q.device, gs
        self.cost = torch.empty((B,), device=q.device, dtype=torch.float32)
        if accumulate_into is None:
            self.gq, self.gqd, acc = torch.empty_like(q, dtype=gio), torch.empty_like(q, dtype=gio), 0
        else:
            self.gq, self.gqd = accumulate_into
            acc = 1
            for g in (self.gq, self.gqd):
                if g.numel() != q.numel() or g.dtype != gio or g.device != q.device or not g.is_contiguous():
                    raise ValueError("GPPriorPlan: accumulate_into buffers must match q (size, device, contiguous) and grad_dtype")
        self._fn = lib().trk_gp_prior_cost_grad
        self._args = (q.data_ptr(), qd.data_ptr(), B, H, D, int(q.dtype == torch.float16), float(dt), float(sigma), float(weight),
                      self.cost.data_ptr(), self.gq.data_ptr(), self.gqd.data_ptr(), gcode, gs, acc)

    def launch(self, stream: Optional[int] = None) -> None:
        with _on(self.device):                  # free when the device is already current
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, stream)
        if rc:
            check(rc, "trk_gp_prior_cost_grad")


class _GPPrior(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, qd, dt, sigma):
        cost, gq, gqd = gp_prior_cost_grad(q, qd, dt, sigma)
        ctx.save_for_backward(gq, gqd)
        return cost

    @staticmethod
    def backward(ctx, gcost):
        gq, gqd = ctx.saved_tensors
        sc = gcost.reshape(-1, 1, 1)
        return (gq * sc).to(gq.dtype), (gqd * sc).to(gqd.dtype), None, None


def gp_prior_cost(q: torch.Tensor, qd: torch.Tensor, dt: float, sigma: float) -> torch.Tensor:
    """Differentiable GP-prior cost per trajectory, (B,H,D) x 2 -> (B,)."""
    return _GPPrior.apply(q, qd, float(dt), float(sigma))


_FD_METHODS = {"forward": 0, "backward": 1, "central": 2}


@host_round_trip
def finite_difference(x: torch.Tensor, dt: float = 1.0, method: str = "forward") -> torch.Tensor:
    """Zero-padded finite differences along the horizon (trajectory/utils.py:53-64): x (..., H, D) -> same shape."""
    if method not in _FD_METHODS:
        raise NotImplementedError
    x = _dev_f32(x, "finite_difference(x)")
    H, D = int(x.shape[-2]), int(x.shape[-1])
    B = int(x.numel() // max(1, H * D))
    out = torch.empty_like(x)
    with _on(x.device):
        check(lib().trk_finite_difference(x.data_ptr(), B, H, D, float(dt), _FD_METHODS[method], out.data_ptr(), _stream(x)),
              "trk_finite_difference")
    return out


@host_round_trip
def traj_diff_norm_sum(x: torch.Tensor, c0: int, dim: int) -> torch.Tensor:
    """sum_t || x[b, t+1, c0:c0+dim] - x[b, t, c0:c0+dim] ||  for x (B, H, S) -> (B,)  (trajectory/metrics.py:7-12, 27-35)."""
    x = _dev_f32(x, "traj_diff_norm_sum(x)")
    if x.dim() != 3:
        raise ValueError("traj_diff_norm_sum: x must be (batch, horizon, state_dim)")
    B, H, S = (int(v) for v in x.shape)
    out = torch.empty((B,), device=x.device, dtype=torch.float32)
    with _on(x.device):
        check(lib().trk_traj_diff_norm_sum(x.data_ptr(), B, H, S, int(c0), int(dim), out.data_ptr(), _stream(x)),
              "trk_traj_diff_norm_sum")
    return out


@host_round_trip
def interpolate_traj_via_points(trajs: torch.Tensor, num_interpolation: int = 10) -> torch.Tensor:
    """trajectory/utils.py:37-50: (..., H, D) -> (..., (H-1)*num_interpolation, D); identity for num_interpolation <= 0."""
    if num_interpolation <= 0:
        return trajs
    assert trajs.ndim > 1
    x = _dev_f32(trajs, "interpolate_traj_via_points(trajs)")
    H, D = x.shape[-2:]
    lead = x.shape[:-2]
    T = int(np.prod(lead)) if len(lead) else 1
    alpha, beta = via_point_weights(num_interpolation, x.device)
    out = torch.empty(tuple(lead) + ((H - 1) * num_interpolation, D), device=x.device, dtype=torch.float32)
    with _on(x.device):
        check(lib().trk_interpolate_via_points(x.data_ptr(), T, H, D, int(num_interpolation), alpha.data_ptr(),
                                               beta.data_ptr(), out.data_ptr(), _stream(x)), "trk_interpolate_via_points")
    return out


def interpolate_columns(x: torch.Tensor, src: torch.Tensor, w: torch.Tensor, n_out: int) -> torch.Tensor:
    """x (..., L, C) -> (..., n_out, C): out[..., k, :] = w[2k] x[..., src[2k], :] + w[2k+1] x[..., src[2k+1], :]
    (interpolate_points_v1, distance_fields.py:66-69; src / w from costmodel.interpolation_table, on x's device)."""
    x = _dev_f32(x, "interpolate_columns(x)")
    if x.dim() < 2:
        raise ValueError("interpolate_columns: x must be (..., links, channels)")
    L, Cc = int(x.shape[-2]), int(x.shape[-1])
    lead = tuple(x.shape[:-2])
    n = int(np.prod(lead)) if lead else 1
    _check_buffer(src, 2 * int(n_out), torch.int32, x.device, "interpolate_columns(src)")
    _check_buffer(w, 2 * int(n_out), torch.float32, x.device, "interpolate_columns(w)")
    out = torch.empty(lead + (int(n_out), Cc), device=x.device, dtype=torch.float32)
    with _on(x.device):
        check(lib().trk_interpolate_columns(x.data_ptr(), n, L, Cc, int(n_out), src.data_ptr(), w.data_ptr(), out.data_ptr(),
                                            _stream(x)), "trk_interpolate_columns")
    return out


def interpolate_columns_backward(g: torch.Tensor, src: torch.Tensor, w: torch.Tensor, n_in: int) -> torch.Tensor:
    g = _dev_f32(g, "interpolate_columns_backward(g)")
    K, Cc = int(g.shape[-2]), int(g.shape[-1])
    lead = tuple(g.shape[:-2])
    n = int(np.prod(lead)) if lead else 1
    gx = torch.empty(lead + (int(n_in), Cc), device=g.device, dtype=torch.float32)
    with _on(g.device):
        check(lib().trk_interpolate_columns_backward(g.data_ptr(), n, int(n_in), Cc, K, src.data_ptr(), w.data_ptr(),
                                                     gx.data_ptr(), _stream(g)), "trk_interpolate_columns_backward")
    return gx


_via_weights: dict = {}


def via_point_weights(num_interpolation: int, device) -> tuple:
    """(alpha, beta) of interpolate_traj_via_points (trajectory/utils.py:43-44) on `device`, cached per (n, device)."""
    key = (int(num_interpolation), str(device))
    ab = _via_weights.get(key)
    if ab is None:
        alpha = torch.linspace(0, 1, key[0] + 2, dtype=torch.float32)[1:key[0] + 1]
        ab = _via_weights[key] = (alpha.to(device), (1 - alpha).to(device))
    return ab


def rollout_collision_via(model: ModelHandle, cm: CostHandle, fields: int, trajs: torch.Tensor, num_interpolation: int,
                          margin: Optional[float] = None, limits=None):
    """Boolean collision fields on the interpolated via points of trajs (T, H, S >= D) without materialising them:
    -> bool (T, (H-1) * num_interpolation), or None when no generated kernel serves this model / cost model (the caller then
    interpolates first).  tasks.py:244-251.
    limits=(q_min, q_max) (float32 device vectors of n_dofs): the launch also folds the per-trajectory flags of `traj_validate` --
    bit 0 some via point collides, bit 1 some way point lies outside the limits -- and the result is (bool tensor, flags buffer): hand
    the buffer to `traj_validate(..., flags=...)`, which then skips its own flags launch."""
    x = _dev_f32(trajs, "rollout_collision_via(trajs)")
    T, H, S = (int(v) for v in x.shape)
    if S < model.n_dofs:
        raise ValueError(f"rollout_collision_via: way points have {S} columns, the model has {model.n_dofs} DOF")
    n = int(num_interpolation)
    alpha, beta = via_point_weights(n, x.device)
    out = torch.empty((T, (H - 1) * n), device=x.device, dtype=torch.bool)
    m = float("nan") if margin is None else float(margin)
    with _on(x.device):
        if limits is None:
            rc = lib().trk_rollout_collision_via(model._h, cm._h, int(fields), x.data_ptr(), T, H, S, n, alpha.data_ptr(), beta.data_ptr(),
                                                 m, out.data_ptr(), _stream(x))
        else:
            q_min, q_max = limits
            _check_buffer(q_min, model.n_dofs, torch.float32, x.device, "rollout_collision_via(q_min)")
            _check_buffer(q_max, model.n_dofs, torch.float32, x.device, "rollout_collision_via(q_max)")
            # [counts 16 B | pad 16 B | flags T, padded to 16 | per-wavefront partial flags]: the layout traj_validate carries on with
            nb = int(lib().trk_via_partial_flags_bytes(T, H, n))
            small = torch.empty(32 + (T + 15) // 16 * 16 + nb, device=x.device, dtype=torch.uint8)
            rc = lib().trk_rollout_collision_via_flags(model._h, cm._h, int(fields), x.data_ptr(), T, H, S, n, alpha.data_ptr(), beta.data_ptr(),
                                                       m, q_min.data_ptr(), q_max.data_ptr(), out.data_ptr(),
                                                       small.data_ptr() + 32 + (T + 15) // 16 * 16, _stream(x))
    if rc == _abi.TRK_ERR_UNSUPPORTED:
        return None
    check(rc, "trk_rollout_collision_via" + ("" if limits is None else "_flags"))
    return out if limits is None else (out, (small, (H - 1) * n))


class _PinnedCounters:
    """A ring of pinned host int32[4] slots the partition kernel writes its counters into DIRECTLY (pinned host memory is
    device-addressable under HIP's unified addressing), the caller's ticket last: the host polls that word -- no device-to-host
    copy call (~20 us for 16 bytes through torch), no event (~5 us to create and record)."""
    _buf = None
    _np = None
    _next = 0
    SLOTS = 64
    _lock = threading.Lock()        # two threads validating trajectories at once must not be handed the same slot and ticket

    @classmethod
    def take(cls):
        """-> (slot pointer, numpy view of the slot, ticket)"""
        with cls._lock:
            if cls._buf is None:
                cls._buf = torch.zeros(cls.SLOTS * 4, dtype=torch.int32).pin_memory()
                cls._np = cls._buf.numpy()
            cls._next += 1
            n = cls._next
        k = n % cls.SLOTS
        ticket = (n % 0x7ffffff0) + 1                          # never 0 (the slots start zeroed), distinct over many ring turns
        return cls._buf.data_ptr() + 16 * k, cls._np[4 * k:4 * k + 4], ticket


class TrajPartition:
    """Result of `traj_validate` (device side of get_trajs_collision_and_free): buffers to slice after `counts()`.
    rows [0, n_free) of `idx` / `gathered` are the free trajectories, [n_free, n_free + n_coll) the colliding ones, the next
    n_out the collision-free ones outside the joint limits (each group in trajectory order)."""
    __slots__ = ("flags", "idx", "gathered", "_small", "_slot", "_ticket", "_host", "_stream")

    def counts(self):
        """(n_free, n_colliding, n_outside_limits): the single device -> host synchronisation of a validation."""
        if self._host is None:
            slot, spins = self._slot, 0
            while slot[3] != self._ticket:                      # the kernel stores the ticket last (system-scope release)
                spins += 1
                if spins > 2_000_000:                           # host memory not coherent on this system: wait for the stream
                    self._stream.synchronize()
                    if slot[3] != self._ticket:
                        raise RuntimeError("traj_validate: the partition kernel did not report its counters (a result must be "
                                           f"read before {_PinnedCounters.SLOTS} later validations reuse its pinned slot)")
            self._host = (int(slot[0]), int(slot[1]), int(slot[2]))
        return self._host


def traj_validate(waypoint_collisions: Optional[torch.Tensor], trajs: torch.Tensor, n_dofs: int, q_min: torch.Tensor, q_max: torch.Tensor,
                  inner: int = 0, gather: bool = True, flags: Optional[torch.Tensor] = None) -> TrajPartition:
    """Per-trajectory flags, the stable [free | colliding | outside-limits] partition (`idx`: int64 (T, 1 | 2)) and the
    trajectories gathered in that order (`gathered`: (T, H, S)) -- tasks.py:253-299 -- queued as three launches with no host
    synchronisation.  waypoint_collisions (T, W) bool / uint8; trajs (T, H, S) float32 contiguous; q_min / q_max float32
    (n_dofs,) on the same device.
    flags: the buffer `rollout_collision_via(..., limits=...)` returned -- the flags are then an input and two launches remain
    (waypoint_collisions is not read and may be None)."""
    x = trajs
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3):
        x = _dev_f32(trajs, "traj_validate(trajs)")
        if x.dim() != 3:
            raise ValueError("traj_validate: trajs must be (T, H, S)")
    T, H, S = x.shape
    _check_buffer(q_min, int(n_dofs), torch.float32, x.device, "traj_validate(q_min)")
    _check_buffer(q_max, int(n_dofs), torch.float32, x.device, "traj_validate(q_max)")
    r = TrajPartition()
    if flags is not None:
        buf, hi = flags
        if buf.device != x.device or buf.dtype != torch.uint8 or buf.numel() < 32 + (T + 15) // 16 * 16 or hi < 1:
            raise ValueError("traj_validate(flags): expected what rollout_collision_via(..., limits=...) returned for these trajectories")
        # the per-wavefront partial flags ride in as `waypoint_collisions`, n_waypoints = -(samples per trajectory)
        wp_ptr, W, r._small = buf.data_ptr() + 32 + (T + 15) // 16 * 16, -int(hi), buf
    else:
        wp = waypoint_collisions
        if wp.device != x.device or wp.element_size() != 1 or not wp.is_contiguous() or wp.numel() % max(T, 1):
            raise ValueError("traj_validate: waypoint_collisions must be a contiguous 1-byte tensor (T, W) on the trajectories' device")
        wp_ptr, W = wp.data_ptr(), wp.numel() // max(T, 1)
        r._small = torch.empty(T + 32, device=x.device, dtype=torch.uint8)      # [counts 16 B | pad | flags T]
    r.flags = r._small[32:32 + T]
    r.idx = torch.empty((T, 2 if inner else 1), device=x.device, dtype=torch.int64)
    r.gathered = torch.empty((T, H, S), device=x.device, dtype=torch.float32) if gather else None
    r._host = None
    slot_ptr, r._slot, r._ticket = _PinnedCounters.take()
    base = r._small.data_ptr()
    with _on(x.device):
        r._stream = torch.cuda.current_stream(x.device)
        check(lib().trk_traj_validate(wp_ptr, x.data_ptr(), T, H, S, W, int(n_dofs), q_min.data_ptr(), q_max.data_ptr(),
                                      int(inner), base + 32, r.idx.data_ptr(), base, slot_ptr, r._ticket, _ptr(r.gathered),
                                      r._stream.cuda_stream), "trk_traj_validate")
    return r


def jtj(lin_jac: torch.Tensor, ang_jac: torch.Tensor, residual: Optional[torch.Tensor] = None, mfma: bool = False,
        damping: Optional[torch.Tensor] = None, solve: bool = False):
    """Normal equations of the geometric Jacobian (robot_tree.py:238-246): lin_jac, ang_jac (N, 3, D) [+ residual (N, 6)] ->
    JtJ (N, D, D) [, Jtr (N, D)] with J = [lin_jac; ang_jac].  mfma=True runs the matrix-core kernel (D <= 8).
    solve=True also returns dq (N, D) = (JtJ + damping I)^-1 Jtr, factorised per sample inside the kernel; damping: a device
    tensor of one value or of N values (None: 0)."""
    lin, ang = _dev_f32(lin_jac, "jtj(lin_jac)"), _dev_f32(ang_jac, "jtj(ang_jac)")
    if lin.dim() != 3 or lin.shape[1] != 3 or ang.shape != lin.shape:
        raise ValueError("jtj: lin_jac and ang_jac must both be (N, 3, D)")
    n, D = int(lin.shape[0]), int(lin.shape[2])
    res = None
    if residual is not None:
        res = _dev_f32(residual, "jtj(residual)")
        if tuple(res.shape) != (n, 6):
            raise ValueError("jtj: residual must be (N, 6)")
    if solve and res is None:
        raise ValueError("jtj(solve=True) needs the residual")
    dmp, stride = None, 0
    if damping is not None:
        dmp = _dev_f32(damping, "jtj(damping)").reshape(-1)
        if dmp.numel() not in (1, n):
            raise ValueError("jtj: damping must hold one value or one per sample")
        stride = int(dmp.numel() == n and n > 1)
    JtJ = torch.empty((n, D, D), device=lin.device, dtype=torch.float32)
    Jtr = torch.empty((n, D), device=lin.device, dtype=torch.float32) if res is not None else None
    dq = torch.empty((n, D), device=lin.device, dtype=torch.float32) if solve else None
    with _on(lin.device):
        check(lib().trk_jtj(lin.data_ptr(), ang.data_ptr(), _ptr(res), n, D, int(bool(mfma)), JtJ.data_ptr(), _ptr(Jtr), _ptr(dmp), stride,
                            _ptr(dq), _stream(lin)), "trk_jtj")
    if solve:
        return JtJ, Jtr, dq
    return (JtJ, Jtr) if res is not None else JtJ


def scale_rows(g: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """g (..., D) fp32 / fp16 times scale (...) per row -> new tensor like g: the backward of the fused rollout's cost output.
    A scale that is one value expanded over the rows (what `.sum().backward()` hands down) is read as a scalar in the kernel."""
    if g.device.type != "cuda" or g.dtype not in (torch.float32, torch.float16):
        raise ValueError("scale_rows: g must be an fp32 / fp16 tensor on the GPU")
    if not g.is_contiguous():
        g = g.contiguous()
    D = int(g.shape[-1])
    rows = g.numel() // max(1, D)
    if scale.device != g.device or scale.numel() != rows:
        raise ValueError("scale_rows: scale must have one entry per row of g, on the same device")
    if scale.dtype == torch.float32 and (rows == 1 or not any(scale.stride())):
        sc, stride = scale, 0                       # every stride 0: one element behind data_ptr()
    else:
        sc, stride = _dev_f32(scale, "scale_rows(scale)"), 1
    out = torch.empty_like(g)
    with _on(g.device):
        check(lib().trk_scale_rows(g.data_ptr(), sc.data_ptr(), stride, rows, D, int(g.dtype == torch.float16), out.data_ptr(),
                                   _stream(g)), "trk_scale_rows")
    return out


def reduce_sum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Deterministic sum of a float32 device vector (fixed association order)."""
    x = _dev_f32(x, "reduce_sum(x)").reshape(-1)
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    with _on(x.device):
        check(lib().trk_reduce_sum(x.data_ptr(), x.numel(), out.data_ptr(), _stream(x)), "trk_reduce_sum")
    return out


class PackedSums:
    """The exchange buffer of a batch-sharded planner, produced by ONE launch: `pack(out)` writes
    [sum cost | sum_b cost(b, h) | sum_b gq(b, h, d)] of a RolloutPlan's latest evaluation into out (1 + H + H D floats, always
    unscaled fp32: an fp16 / loss-scaled gradient is widened and divided by the plan's grad_scale).  traj_cost (B,): a per-trajectory
    cost evaluated next to the rollout (the GP prior's) whose sum joins out[0]."""

    def __init__(self, plan: "RolloutPlan", block_sums: torch.Tensor, traj_cost: Optional[torch.Tensor] = None):
        self.B, self.H, self.D = plan.B, plan.H, int(plan.gq.shape[-1])
        self.size = 1 + self.H + self.H * self.D
        _check_buffer(block_sums, n_blocks(self.B * self.H), torch.float32, plan.device, "PackedSums(block_sums)", at_least=True)
        _check_buffer(traj_cost, self.B, torch.float32, plan.device, "PackedSums(traj_cost)")
        nbytes = int(lib().trk_pack_sums_scratch_bytes(self.H, self.D))
        self._scratch = torch.zeros(nbytes // 4, device=plan.device, dtype=torch.float32)     # zeroed once: holds the ticket
        self._args = (plan.cost.data_ptr(), plan.gq.data_ptr(), int(plan.gq.dtype == torch.float16), float(plan.grad_scale),
                      block_sums.data_ptr(), _ptr(traj_cost), self.B, self.H, self.D, self._scratch.data_ptr())
        self._keep = (plan, block_sums, traj_cost)
        self.device = plan.device

    def pack(self, out: torch.Tensor, stream: Optional[int] = None) -> None:
        _check_buffer(out, self.size, torch.float32, self.device, "PackedSums.pack(out)")
        with _on(self.device):
            check(lib().trk_pack_sums(*self._args, out.data_ptr(), _stream_of(self.device) if stream is None else stream), "trk_pack_sums")


def n_blocks(n_samples: int) -> int:
    """Length of the `cost_block_sums` output of the fused rollout (one entry per 64 samples)."""
    return (int(n_samples) + 63) // 64


def grid_precompute(cm: CostHandle, dims, lim_min, lim_max):
    dims_a = np.ascontiguousarray(dims, np.int32)
    lo, hi = np.ascontiguousarray(lim_min, np.float32), np.ascontiguousarray(lim_max, np.float32)
    shape = tuple(int(d) for d in dims_a)
    sdf = torch.empty(shape, device=cm.device, dtype=torch.float32)
    grad = torch.empty(shape + (3,), device=cm.device, dtype=torch.float32)
    with _on(cm.device):
        check(lib().trk_grid_precompute(cm._h, dims_a.ctypes.data, lo.ctypes.data, hi.ctypes.data, sdf.data_ptr(),
                                        grad.data_ptr(), _stream_of(cm.device)),
              "trk_grid_precompute")
    return sdf, grad


def sdf_points(cm: CostHandle, pts: torch.Tensor, want_grad=False):
    pts = _dev_f32(pts, "sdf_points(pts)").reshape(-1, 3)
    n = pts.shape[0]
    sdf = torch.empty((n, cm.n_objects), device=pts.device, dtype=torch.float32)
    grad = torch.empty((n, cm.n_objects, 3), device=pts.device, dtype=torch.float32) if want_grad else None
    with _on(pts.device):
        check(lib().trk_sdf_points(cm._h, pts.data_ptr(), n, sdf.data_ptr(), _ptr(grad), _stream(pts)), "trk_sdf_points")
    return (sdf, grad) if want_grad else sdf


# ----------------------------------------------------------------------------------------------
# autograd: explicit backward kernels instead of a recorded graph
# ----------------------------------------------------------------------------------------------
class _FK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, model, sel):
        ctx.model, ctx.sel = model, sel
        ctx.save_for_backward(q)
        return fk_forward(model, q, sel)

    @staticmethod
    def backward(ctx, gH):
        (q,) = ctx.saved_tensors
        gq = fk_backward(ctx.model, q, gH.contiguous(), ctx.sel)
        return gq.reshape(q.shape), None, None


class _FKPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, model, sel):
        ctx.model, ctx.sel = model, sel
        ctx.save_for_backward(q)
        return fk_positions(model, q, sel)

    @staticmethod
    def backward(ctx, gpos):
        (q,) = ctx.saved_tensors
        gq = fk_positions_backward(ctx.model, q, gpos.contiguous(), ctx.sel)
        return gq.reshape(q.shape), None, None


class _FKPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, ps):
        ctx.ps = ps
        ctx.save_for_backward(q)
        return fk_points(ps, q)

    @staticmethod
    def backward(ctx, gpos):
        (q,) = ctx.saved_tensors
        gq = fk_points_backward(ctx.ps, q, gpos.contiguous())
        return gq.reshape(q.shape), None


class _CostFields(torch.autograd.Function):
    @staticmethod
    def forward(ctx, link_pos, cm, fields):
        ctx.cm, ctx.fields = cm, fields
        ctx.save_for_backward(link_pos)
        return cost_fields(cm, fields, link_pos)

    @staticmethod
    def backward(ctx, gcost):
        (link_pos,) = ctx.saved_tensors
        _, g = cost_fields(ctx.cm, ctx.fields, link_pos, gcost=gcost.contiguous(), want_grad=True)
        return g.reshape(link_pos.shape), None, None


class _EECost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, cm, target):
        ctx.cm = cm
        ctx.save_for_backward(H, target if target is not None else torch.empty(0, device=H.device))
        ctx.has_target = target is not None
        return ee_cost(cm, H, target)

    @staticmethod
    def backward(ctx, gcost):
        H, target = ctx.saved_tensors
        _, gH = ee_cost(ctx.cm, H, target if ctx.has_target else None, gcost=gcost.contiguous(), want_grad=True)
        return gH.reshape(H.shape), None, None


class _Rollout(torch.autograd.Function):
    """cost (…), link_pos (…,L,3) from q; backward uses the gradient the fused kernel already produced."""

    @staticmethod
    def forward(ctx, q, model, cm, weights, ps, want_pos):
        if ps is None:
            pos, cost, gq = rollout_cost_grad(model, cm, weights, q, want_pos=want_pos)
        else:
            pos, cost, gq = rollout_points_cost_grad(ps, cm, weights, q, want_pos=want_pos)
        ctx.save_for_backward(gq)
        if pos is None:                              # cost only: the kernel instantiation / launch without the position stream
            return cost, None
        ctx.mark_non_differentiable(pos)
        return cost, pos

    @staticmethod
    def backward(ctx, gcost, _gpos):
        (gq,) = ctx.saved_tensors
        return scale_rows(gq, gcost), None, None, None, None, None


# The differentiable entry points below exist twice: as `torch.autograd.Function`s over the ctypes calls (the eager path: least
# host time per call) and as dispatcher ops `torch.ops.trk.*` (custom_ops.py: what `torch.compile`, `torch.export` and
# `torch.library.opcheck` see).  TRK_DISPATCHER_OPS=1 routes eager calls through the dispatcher as well; under a compiler the
# dispatcher ops are always used (ctypes calls cannot be traced).
_ALWAYS_DISPATCH = os.environ.get("TRK_DISPATCHER_OPS", "0") == "1"


def _dispatch() -> bool:
    return _ALWAYS_DISPATCH or torch.compiler.is_compiling()


def _trk_ops():
    from . import custom_ops  # noqa: F401  (registers torch.ops.trk.*)
    return torch.ops.trk


def _sel_list(sel):
    return None if sel is None else [int(i) for i in sel]


def fk(model: ModelHandle, q: torch.Tensor, sel=None) -> torch.Tensor:
    if _dispatch():
        return _trk_ops().fk(q.reshape(-1, model.n_dofs), model.uid, _sel_list(sel))
    q2 = _dev_f32(q, "fk(q)").reshape(-1, model.n_dofs)
    if torch.is_grad_enabled() and q.requires_grad:
        return _FK.apply(q2, model, sel)
    return fk_forward(model, q2, sel)


def fk_pos(model: ModelHandle, q: torch.Tensor, sel=None) -> torch.Tensor:
    if _dispatch():
        return _trk_ops().fk_positions(q.reshape(-1, model.n_dofs), model.uid, _sel_list(sel))
    q2 = _dev_f32(q, "fk_pos(q)").reshape(-1, model.n_dofs)
    if torch.is_grad_enabled() and q.requires_grad:
        return _FKPos.apply(q2, model, sel)
    return fk_positions(model, q2, sel)


def fk_points_ad(ps: PointSetHandle, q: torch.Tensor) -> torch.Tensor:
    q2 = _dev_f32(q, "fk_points(q)").reshape(-1, ps.model.n_dofs)
    if torch.is_grad_enabled() and q.requires_grad:
        return _FKPoints.apply(q2, ps)
    return fk_points(ps, q2)


def cost_fields_ad(cm: CostHandle, fields: int, link_pos: torch.Tensor) -> torch.Tensor:
    if _dispatch():
        return _trk_ops().cost_fields(link_pos.reshape(-1, cm.n_links_in, 3), cm.uid, int(fields))
    lp = _dev_f32(link_pos, "cost_fields(link_pos)").reshape(-1, cm.n_links_in, 3)
    if torch.is_grad_enabled() and link_pos.requires_grad:
        return _CostFields.apply(lp, cm, fields)
    return cost_fields(cm, fields, lp)


def ee_cost_ad(cm: CostHandle, H: torch.Tensor, target=None) -> torch.Tensor:
    if _dispatch():
        return _trk_ops().ee_cost(H.reshape(-1, 4, 4), target, cm.uid)
    Hc = _dev_f32(H, "ee_cost(H)").reshape(-1, 4, 4)
    if torch.is_grad_enabled() and H.requires_grad:
        return _EECost.apply(Hc, cm, target)
    return ee_cost(cm, Hc, target)


def rollout_ad(model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, ps: Optional[PointSetHandle] = None,
               want_pos: bool = True):
    """Differentiable fused op: returns (cost, link_pos) -- or (cost, point_pos) when a point set is given; want_pos=False
    returns (cost, None) and skips the position output (34.6 of the 50 MB a Panda evaluation writes)."""
    if ps is None and q.is_cuda and q.dtype in (torch.float32, torch.float16):
        native = _lib_mod.torch_ops()
        if native is not None:
            # the native dispatcher op (csrc/trk_torch_ops.cpp): one C++ autograd node whose backward is trk_scale_rows; the handles
            # travel as their C pointers -- eager, torch.compile (fullgraph) and export see the same op
            cost, _gq, pos = native.rollout(q, model.ptr, cm.ptr, float(weights[0]), float(weights[1]), float(weights[2]),
                                            float(weights[3]), bool(want_pos))
            return cost, (pos if want_pos else None)
    if _dispatch():
        cost, _gq, pos = _trk_ops().rollout_cost_grad(q, model.uid, cm.uid, [float(w) for w in weights], bool(want_pos),
                                                      ps.uid if ps is not None else 0)
        return cost, (pos if want_pos else None)
    return _Rollout.apply(_dev_f32(q, "rollout(q)"), model, cm, tuple(float(w) for w in weights), ps, bool(want_pos))


class RolloutPlan:
    """Pre-bound fused-rollout launch: every argument is resolved once, so a step costs one ctypes call
    (a planner's inner loop re-evaluates the same buffers thousands of times)."""

    def __init__(self, model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, want_pos: bool = True, grad_dtype=None,
                 grad_scale: float = 1.0, gq_out: Optional[torch.Tensor] = None, strict: bool = True):
        """gq_out: a caller-owned (B, H, D) buffer the gradient is written to (e.g. the storage of `q.grad`: the kernel then IS the
        backward of `sum(cost)`); default: a buffer of the plan's own.
        strict: raise (NotImplementedError) when the model has generated kernels but this cost model / weights would be served by the
        table-driven one; `self.generated` says which family the plan's launches take."""
        self.generated = rollout_is_specialized(model, cm, weights)
        if strict:
            _require_generated("RolloutPlan", model, self.generated)
        f16 = q.dtype == torch.float16
        gio, gcode, gs = _grad_mode(f16, grad_dtype, grad_scale, "RolloutPlan")
        if not f16 and gs != 1.0:
            raise ValueError("RolloutPlan: grad_scale applies to float16 trajectories only")
        self.grad_scale = gs
        q = q.contiguous() if (f16 and q.device.type == "cuda") else _dev_f32(q, "RolloutPlan(q)")
        if q.dim() != 3:
            raise ValueError("RolloutPlan: q must be (batch, horizon, dof)")
        _check_q_dofs(q, model.n_dofs, "RolloutPlan(q)")
        self.model, self.cm, self.q = model, cm, q
        self.B, self.H = int(q.shape[0]), int(q.shape[1])
        n, L, D = self.B * self.H, model.n_links, model.n_dofs
        kw = dict(device=q.device, dtype=q.dtype)
        self.link_pos = torch.empty((self.B, self.H, L, 3), **kw) if want_pos else None
        self.cost = torch.empty((self.B, self.H), device=q.device, dtype=torch.float32)
        if gq_out is not None:
            _check_buffer(gq_out, n * D, gio, q.device, "RolloutPlan(gq_out)")
            self.gq = gq_out.view(self.B, self.H, D)
        else:
            self.gq = torch.empty((self.B, self.H, D), device=q.device, dtype=gio)
        self._w = _abi.RolloutWeights(*[float(v) for v in weights])
        self._fn = lib().trk_rollout_cost_grad_f16 if f16 else lib().trk_rollout_cost_grad
        self._args = (model._h, cm._h, C.byref(self._w), q.data_ptr(), self.B, self.H, _ptr(self.link_pos),
                      self.cost.data_ptr(), self.gq.data_ptr()) + ((gcode, gs) if f16 else ())
        self.device = q.device

    def launch(self, cost_sum_ptr: Optional[int] = None, stream: Optional[int] = None) -> None:
        with _on(self.device):                  # free when the device is already current
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, cost_sum_ptr, stream)
        if rc:
            check(rc, "trk_rollout_cost_grad")


class RolloutJacobianPlan(RolloutPlan):
    """BASELINE config 4's step -- fused FK + objectives + gradient AND the stateful FK + geometric Jacobian of `link`
    (robot_tree.py:218-248) -- as one pre-bound call (`trk_rollout_jacobian_cost_grad`): ONE launch when a generated unit tracks
    `link` (UR10 + Allegro: the columns are read out of the poses the rollout already holds), else the two launches.  Outputs of a
    RolloutPlan plus `pos` (B, H, 3), `quat` (B, H, 4 wxyz), `lin_jac` / `ang_jac` (B, H, 3, D)."""

    def __init__(self, model: ModelHandle, cm: CostHandle, weights, q: torch.Tensor, link: int, strict: bool = True):
        if q.dtype != torch.float32:
            raise ValueError("RolloutJacobianPlan: float32 trajectories (the Jacobian is an fp32 output)")
        super().__init__(model, cm, weights, q, want_pos=True, strict=strict)
        kw = dict(device=self.q.device, dtype=torch.float32)
        B, H, D = self.B, self.H, model.n_dofs
        self.link = int(link)
        self.pos, self.quat = torch.empty((B, H, 3), **kw), torch.empty((B, H, 4), **kw)
        self.lin_jac, self.ang_jac = torch.empty((B, H, 3, D), **kw), torch.empty((B, H, 3, D), **kw)
        self._fn = lib().trk_rollout_jacobian_cost_grad
        self._args = (model._h, cm._h, C.byref(self._w), self.q.data_ptr(), B, H, self.link, _ptr(self.link_pos), self.cost.data_ptr(),
                      self.gq.data_ptr())
        self._tail = (self.pos.data_ptr(), self.quat.data_ptr(), self.lin_jac.data_ptr(), self.ang_jac.data_ptr())

    def launch(self, cost_sum_ptr: Optional[int] = None, stream: Optional[int] = None) -> None:
        with _on(self.device):
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, cost_sum_ptr, *self._tail, stream)
        if rc:
            check(rc, "trk_rollout_jacobian_cost_grad")


class PointsRolloutPlan:
    """RolloutPlan for the attached-point models (link spheres, grasped-object points; `trk_rollout_points_cost_grad`): the collision
    fields' columns are the points of `ps`, `link_pos` (B, H, P, 3) their world positions.  Same interface as RolloutPlan (launch(),
    link_pos / cost / gq), so step graphs, `ShardedRollout` and `PackedSums` take it unchanged."""

    def __init__(self, ps: PointSetHandle, cm: CostHandle, weights, q: torch.Tensor, want_pos: bool = True,
                 gq_out: Optional[torch.Tensor] = None, strict: bool = True):
        model = ps.model
        self.generated = rollout_points_is_specialized(ps, cm, weights)
        if strict and ps.specialized and not self.generated:
            _require_generated("PointsRolloutPlan", model, False)
        q = _dev_f32(q, "PointsRolloutPlan(q)")
        if q.dim() != 3:
            raise ValueError("PointsRolloutPlan: q must be (batch, horizon, dof)")
        _check_q_dofs(q, model.n_dofs, "PointsRolloutPlan(q)")
        self.model, self.ps, self.cm, self.q = model, ps, cm, q
        self.B, self.H = int(q.shape[0]), int(q.shape[1])
        n, P, D = self.B * self.H, ps.n_points, model.n_dofs
        self.link_pos = torch.empty((self.B, self.H, P, 3), device=q.device, dtype=torch.float32) if want_pos else None
        self.cost = torch.empty((self.B, self.H), device=q.device, dtype=torch.float32)
        if gq_out is not None:
            _check_buffer(gq_out, n * D, torch.float32, q.device, "PointsRolloutPlan(gq_out)")
            self.gq = gq_out.view(self.B, self.H, D)
        else:
            self.gq = torch.empty((self.B, self.H, D), device=q.device, dtype=torch.float32)
        self.grad_scale = 1.0
        self._w = _abi.RolloutWeights(*[float(v) for v in weights])
        self._fn = lib().trk_rollout_points_cost_grad
        self._args = (model._h, ps._h, cm._h, C.byref(self._w), q.data_ptr(), self.B, self.H, _ptr(self.link_pos),
                      self.cost.data_ptr(), self.gq.data_ptr())
        self.device = q.device

    def launch(self, cost_sum_ptr: Optional[int] = None, stream: Optional[int] = None) -> None:
        with _on(self.device):
            if stream is None:
                stream = _stream_of(self.device)
            rc = self._fn(*self._args, cost_sum_ptr, stream)
        if rc:
            check(rc, "trk_rollout_points_cost_grad")
