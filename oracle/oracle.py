"""TEST INFRASTRUCTURE: ctypes wrapper over oracle/_build/liboracle.so (the C restatement).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package never imports this module.

The model / cost descriptions are the same `TrkKinModelDesc` / `TrkCostModelDesc` bytes the
HIP library receives (torch_robotics_amd/_abi.py), so both sides see identical inputs.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from torch_robotics_amd import _abi

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None


def build() -> Path:
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _dt(prec):
    return (np.float32, C.c_float, "_f32") if prec == "f32" else (np.float64, C.c_double, "_f64")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def max_threads() -> int:
    return int(lib().orc_max_threads())


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


class Oracle:
    """Bound to one KinModel (+ optional CostModelSpec with host grid arrays)."""

    def __init__(self, model, cost_spec=None):
        self.model = model
        self.kd, self._keep = _abi.kin_desc(model)
        self.cd = None
        if cost_spec is not None:
            self.set_cost(cost_spec)

    def refresh_model(self):
        self.kd, self._keep = _abi.kin_desc(self.model)

    def set_cost(self, spec):
        grid_ptrs = None
        self._grid_keep = None
        if spec.grid is not None:
            sdf = np.ascontiguousarray(np.asarray(spec.grid["sdf"], np.float32))
            grad = np.ascontiguousarray(np.asarray(spec.grid["grad"], np.float32))
            self._grid_keep = (sdf, grad)
            grid_ptrs = (sdf.ctypes.data, grad.ctypes.data)
        self.spec = spec
        self.cd, self._ckeep = _abi.cost_desc(spec, grid_ptrs)

    # ------------------------------------------------------------------ FK
    def fk(self, q, prec="f32", ref_order=False):
        npdt, _, suf = _dt(prec)
        q = np.ascontiguousarray(q, npdt).reshape(-1, self.model.n_dofs)
        n = q.shape[0]
        H = np.empty((n, self.model.n_links, 4, 4), npdt)
        fn = getattr(lib(), ("orc_fk_ref_order" if ref_order else "orc_fk") + suf)
        fn(C.byref(self.kd), _p(q), C.c_int64(n), _p(H))
        return H

    def fk_backward(self, q, gH, prec="f32"):
        npdt, _, suf = _dt(prec)
        q = np.ascontiguousarray(q, npdt).reshape(-1, self.model.n_dofs)
        gH = np.ascontiguousarray(gH, npdt).reshape(q.shape[0], self.model.n_links, 4, 4)
        gq = np.empty_like(q)
        getattr(lib(), "orc_fk_backward" + suf)(C.byref(self.kd), _p(q), _p(gH), C.c_int64(q.shape[0]), _p(gq))
        return gq

    def jacobian(self, q, qd, link, prec="f32"):
        npdt, _, suf = _dt(prec)
        D = self.model.n_dofs
        q = np.ascontiguousarray(q, npdt).reshape(-1, D)
        n = q.shape[0]
        qd_a = None if qd is None else np.ascontiguousarray(qd, npdt).reshape(n, D)
        pos, quat = np.empty((n, 3), npdt), np.empty((n, 4), npdt)
        lin, ang = np.empty((n, 3, D), npdt), np.empty((n, 3, D), npdt)
        vl, va = np.empty((n, 3), npdt), np.empty((n, 3), npdt)
        getattr(lib(), "orc_fk_jacobian" + suf)(
            C.byref(self.kd), _p(q), None if qd_a is None else _p(qd_a), C.c_int64(n), C.c_int(int(link)),
            _p(pos), _p(quat), _p(lin), _p(ang), _p(vl), _p(va))
        return pos, quat, lin, ang, vl, va

    def analytic_jacobian(self, q, prec="f32"):
        npdt, _, suf = _dt(prec)
        D, L = self.model.n_dofs, self.model.n_links
        q = np.ascontiguousarray(q, npdt).reshape(-1, D)
        J = np.empty((q.shape[0], L, 7, D), npdt)
        getattr(lib(), "orc_fk_analytic_jacobian" + suf)(C.byref(self.kd), _p(q), C.c_int64(q.shape[0]), _p(J))
        return J

    def ik_step(self, link, H_target, lower, upper, q, mom, vel, step, lr=1e-2, w_jl=300.0, se3_eps=1e-1, prec="f32"):
        """One IK iteration IN PLACE on q / mom / vel; returns (loss, grad, valid) evaluated before the update."""
        npdt, ct, suf = _dt(prec)
        D = self.model.n_dofs
        n = q.shape[0]
        assert q.dtype == npdt and mom.dtype == npdt and vel.dtype == npdt and q.flags.c_contiguous
        Ht = np.ascontiguousarray(H_target, npdt)
        per_sample = int(Ht.ndim == 3)
        lo, hi = np.ascontiguousarray(lower, npdt), np.ascontiguousarray(upper, npdt)
        loss, grad, valid = np.empty(n, npdt), np.empty((n, D), npdt), np.empty(n, np.uint8)
        getattr(lib(), "orc_ik_step" + suf)(C.byref(self.kd), C.c_int(int(link)), _p(Ht), C.c_int(per_sample), _p(lo), _p(hi),
                                            ct(w_jl), ct(se3_eps), ct(lr), C.c_int(int(step)), C.c_int64(n), _p(q), _p(mom),
                                            _p(vel), _p(loss), _p(grad), _p(valid))
        return loss, grad, valid.astype(bool)

    def ik_gn_step(self, link, H_target, lower, upper, q, damping=1e-4, lm_gain=0.1, step_scale=1.0, prec="f64"):
        """One damped Gauss-Newton IK iteration (build-defined; oracle_impl.inc orc_ik_gn_step): -> (q_new, err of q as passed)."""
        npdt, ct, suf = _dt(prec)
        D = self.model.n_dofs
        q = np.ascontiguousarray(q, npdt).reshape(-1, D).copy()
        n = q.shape[0]
        Ht = np.ascontiguousarray(H_target, npdt).reshape(-1, 16)
        lo, hi = np.ascontiguousarray(lower, npdt), np.ascontiguousarray(upper, npdt)
        err = np.empty(n, npdt)
        getattr(lib(), "orc_ik_gn_step" + suf)(C.byref(self.kd), C.c_int(int(link)), _p(Ht), C.c_int(int(Ht.shape[0] > 1)), _p(lo), _p(hi),
                                               ct(damping), ct(lm_gain), ct(step_scale), C.c_int64(n), _p(q), _p(err))
        return q, err

    @staticmethod
    def rotmat_to_quat(R, prec="f32"):
        npdt, _, suf = _dt(prec)
        R = np.ascontiguousarray(R, npdt).reshape(-1, 9)
        out = np.empty((R.shape[0], 4), npdt)
        getattr(lib(), "orc_rotmat_to_quat" + suf)(_p(R), C.c_int64(R.shape[0]), _p(out))
        return out

    @staticmethod
    def rotation_from(kind, x, prec="f32"):
        """kind 0/1/2: x_rot / y_rot / z_rot of angles (n,); kind 3: q_to_rotation_matrix of wxyz quaternions (n,4)."""
        npdt, _, suf = _dt(prec)
        x = np.ascontiguousarray(x, npdt)
        n = x.shape[0]
        R = np.empty((n, 3, 3), npdt)
        getattr(lib(), "orc_rotation_from" + suf)(C.c_int(kind), _p(x), C.c_int64(n), _p(R))
        return R

    # --------------------------------------------------------------- Frame algebra (geometrics/frame.py:55-121)
    @staticmethod
    def frame_compose(op, Ra, ta, Rb=None, tb=None, prec="f32"):
        npdt, _, suf = _dt(prec)
        Ra, ta = np.ascontiguousarray(Ra, npdt).reshape(-1, 9), np.ascontiguousarray(ta, npdt).reshape(-1, 3)
        if Rb is None:
            Rb, tb = np.zeros((1, 9), npdt), np.zeros((1, 3), npdt)
        Rb, tb = np.ascontiguousarray(Rb, npdt).reshape(-1, 9), np.ascontiguousarray(tb, npdt).reshape(-1, 3)
        n = Ra.shape[0] if op == 1 else max(Ra.shape[0], Rb.shape[0])
        Ro, to = np.empty((n, 3, 3), npdt), np.empty((n, 3), npdt)
        getattr(lib(), "orc_frame_compose" + suf)(C.c_int(op), _p(Ra), _p(ta), C.c_int64(Ra.shape[0]), _p(Rb), _p(tb),
                                                  C.c_int64(Rb.shape[0]), _p(Ro), _p(to))
        return Ro, to

    @staticmethod
    def frame_compose_backward(op, Ra, ta, Rb, tb, gR, gt, prec="f32"):
        npdt, _, suf = _dt(prec)
        arrs = [np.ascontiguousarray(a, npdt) for a in (Ra, ta, Rb, tb, gR, gt)]
        n = arrs[0].reshape(-1, 9).shape[0]
        outs = [np.zeros((n, 3, 3), npdt), np.zeros((n, 3), npdt), np.zeros((n, 3, 3), npdt), np.zeros((n, 3), npdt)]
        getattr(lib(), "orc_frame_compose_backward" + suf)(C.c_int(op), *[_p(a) for a in arrs], C.c_int64(n), *[_p(o) for o in outs])
        return outs

    @staticmethod
    def frame_transform_points(R, t, pts, g=None, prec="f32"):
        npdt, _, suf = _dt(prec)
        R, t = np.ascontiguousarray(R, npdt).reshape(-1, 9), np.ascontiguousarray(t, npdt).reshape(-1, 3)
        pts = np.ascontiguousarray(pts, npdt).reshape(-1, 3)
        n, P = R.shape[0], pts.shape[0]
        out = np.empty((n, P, 3), npdt)
        gR, gt = np.zeros((n, 3, 3), npdt), np.zeros((n, 3), npdt)
        gg = None if g is None else np.ascontiguousarray(g, npdt)
        getattr(lib(), "orc_frame_transform_points" + suf)(_p(R), _p(t), C.c_int64(n), _p(pts), C.c_int(P), _p(out),
                                                           None if gg is None else _p(gg), _p(gR), _p(gt))
        return (out, gR, gt) if g is not None else out

    @staticmethod
    def frame_quat_euler(R, prec="f32"):
        npdt, _, suf = _dt(prec)
        R = np.ascontiguousarray(R, npdt).reshape(-1, 9)
        quat, eul = np.empty((R.shape[0], 4), npdt), np.empty((R.shape[0], 3), npdt)
        getattr(lib(), "orc_frame_quat_euler" + suf)(_p(R), C.c_int64(R.shape[0]), _p(quat), _p(eul))
        return quat, eul

    # --------------------------------------------------------------- costs
    def cost_fields(self, fields, link_pos, prec="f32", grad=True):
        npdt, _, suf = _dt(prec)
        Lin = self.spec.n_links_in
        pos = np.ascontiguousarray(link_pos, npdt).reshape(-1, Lin, 3)
        n = pos.shape[0]
        cost = np.empty(n, npdt)
        g = np.empty_like(pos) if grad else None
        getattr(lib(), "orc_cost_fields" + suf)(C.byref(self.cd), C.c_int(fields), _p(pos), C.c_int64(n),
                                                _p(cost), None if g is None else _p(g))
        return (cost, g) if grad else cost

    def collision_fields(self, fields, link_pos, margin=None, prec="f32"):
        npdt, ct, suf = _dt(prec)
        Lin = self.spec.n_links_in
        pos = np.ascontiguousarray(link_pos, npdt).reshape(-1, Lin, 3)
        out = np.empty(pos.shape[0], np.uint8)
        getattr(lib(), "orc_collision_fields" + suf)(C.byref(self.cd), C.c_int(fields), _p(pos),
                                                     C.c_int64(pos.shape[0]),
                                                     ct(float("nan") if margin is None else margin), _p(out))
        return out.astype(bool)

    def ee_cost(self, H, target=None, prec="f32", grad=True):
        npdt, _, suf = _dt(prec)
        H = np.ascontiguousarray(H, npdt).reshape(-1, 4, 4)
        n = H.shape[0]
        per_sample, tgt = 0, None
        if target is not None:
            tgt = np.ascontiguousarray(target, npdt)
            per_sample = int(tgt.ndim == 3)
        cost = np.empty(n, npdt)
        gH = np.empty_like(H) if grad else None
        getattr(lib(), "orc_ee_cost" + suf)(C.byref(self.cd), _p(H), C.c_int64(n),
                                            None if tgt is None else _p(tgt), C.c_int(per_sample),
                                            _p(cost), None if gH is None else _p(gH))
        return (cost, gH) if grad else cost

    def rollout(self, q, weights, prec="f32", want_pos=True):
        npdt, _, suf = _dt(prec)
        D, L = self.model.n_dofs, self.model.n_links
        q = np.ascontiguousarray(q, npdt).reshape(-1, D)
        n = q.shape[0]
        w = _abi.RolloutWeights(*[float(v) for v in weights])
        pos = np.empty((n, L, 3), npdt) if want_pos else None
        cost, gq = np.empty(n, npdt), np.empty((n, D), npdt)
        getattr(lib(), "orc_rollout" + suf)(C.byref(self.kd), C.byref(self.cd), C.byref(w), _p(q), C.c_int64(n),
                                            None if pos is None else _p(pos), _p(cost), _p(gq))
        return pos, cost, gq

    # ------------------------------------------------------ attached points
    @staticmethod
    def _points(point_link, point_offset):
        pl = np.ascontiguousarray(point_link, np.int32).reshape(-1)
        po = np.ascontiguousarray(point_offset, np.float32).reshape(-1, 3)
        assert pl.shape[0] == po.shape[0]
        return pl, po

    def fk_points(self, point_link, point_offset, q, prec="f32"):
        npdt, _, suf = _dt(prec)
        pl, po = self._points(point_link, point_offset)
        q = np.ascontiguousarray(q, npdt).reshape(-1, self.model.n_dofs)
        out = np.empty((q.shape[0], pl.shape[0], 3), npdt)
        getattr(lib(), "orc_fk_points" + suf)(C.byref(self.kd), _p(pl), _p(po), C.c_int(pl.shape[0]), _p(q),
                                               C.c_int64(q.shape[0]), _p(out))
        return out

    def fk_points_backward(self, point_link, point_offset, q, gpos, prec="f32"):
        npdt, _, suf = _dt(prec)
        pl, po = self._points(point_link, point_offset)
        q = np.ascontiguousarray(q, npdt).reshape(-1, self.model.n_dofs)
        g = np.ascontiguousarray(gpos, npdt).reshape(q.shape[0], pl.shape[0], 3)
        gq = np.empty_like(q)
        getattr(lib(), "orc_fk_points_backward" + suf)(C.byref(self.kd), _p(pl), _p(po), C.c_int(pl.shape[0]), _p(q), _p(g),
                                                        C.c_int64(q.shape[0]), _p(gq))
        return gq

    def rollout_points(self, point_link, point_offset, q, weights, prec="f32"):
        npdt, _, suf = _dt(prec)
        pl, po = self._points(point_link, point_offset)
        D = self.model.n_dofs
        q = np.ascontiguousarray(q, npdt).reshape(-1, D)
        n = q.shape[0]
        w = _abi.RolloutWeights(*[float(v) for v in weights])
        pos = np.empty((n, pl.shape[0], 3), npdt)
        cost, gq = np.empty(n, npdt), np.empty((n, D), npdt)
        getattr(lib(), "orc_rollout_points" + suf)(C.byref(self.kd), _p(pl), _p(po), C.c_int(pl.shape[0]), C.byref(self.cd),
                                                    C.byref(w), _p(q), C.c_int64(n), _p(pos), _p(cost), _p(gq))
        return pos, cost, gq

    def grid_precompute(self, dims, lim_min, lim_max, prec="f32"):
        npdt, _, suf = _dt(prec)
        dims_a = np.ascontiguousarray(dims, np.int32)
        lo, hi = np.ascontiguousarray(lim_min, npdt), np.ascontiguousarray(lim_max, npdt)
        sdf = np.empty(tuple(int(d) for d in dims_a), npdt)
        grad = np.empty(sdf.shape + (3,), npdt)
        getattr(lib(), "orc_grid_precompute" + suf)(C.byref(self.cd), _p(dims_a), _p(lo), _p(hi), _p(sdf), _p(grad))
        return sdf, grad

    def sdf_points(self, pts, prec="f32"):
        npdt, _, suf = _dt(prec)
        pts = np.ascontiguousarray(pts, npdt).reshape(-1, 3)
        no = len(self.spec.objects)
        sdf = np.empty((pts.shape[0], no), npdt)
        grad = np.empty((pts.shape[0], no, 3), npdt)
        getattr(lib(), "orc_sdf_points" + suf)(C.byref(self.cd), _p(pts), C.c_int64(pts.shape[0]), _p(sdf), _p(grad))
        return sdf, grad


def gp_prior(q, qd, dt, sigma, weight=1.0, prec="f64"):
    """Constant-velocity GP prior (build-defined term of BASELINE config 5): (cost [B], gq, gqd)."""
    npdt, ct, suf = _dt(prec)
    q = np.ascontiguousarray(q, npdt)
    qd = np.ascontiguousarray(qd, npdt)
    B, H, D = q.shape
    cost, gq, gqd = np.empty(B, npdt), np.empty_like(q), np.empty_like(q)
    getattr(lib(), "orc_gp_prior" + suf)(_p(q), _p(qd), C.c_int64(B), C.c_int(H), C.c_int(D), ct(dt), ct(sigma), ct(weight),
                                         _p(cost), _p(gq), _p(gqd))
    return cost, gq, gqd


def gp_factor_cost(q, qd, dt, sigma, weight=1.0, prec="f64"):
    """The prior's cost per factor: (B, H), out[b, t] = the factor between t and t + 1 (0 at t = H - 1)."""
    npdt, ct, suf = _dt(prec)
    q = np.ascontiguousarray(q, npdt)
    qd = np.ascontiguousarray(qd, npdt)
    B, H, D = q.shape
    out = np.empty((B, H), npdt)
    getattr(lib(), "orc_gp_factor_cost" + suf)(_p(q), _p(qd), C.c_int64(B), C.c_int(H), C.c_int(D), ct(dt), ct(sigma), ct(weight), _p(out))
    return out


def finite_difference(x, dt=1.0, method="forward", prec="f64"):
    npdt, ct, suf = _dt(prec)
    x = np.ascontiguousarray(x, npdt)
    H, D = x.shape[-2], x.shape[-1]
    out = np.empty_like(x)
    getattr(lib(), "orc_finite_difference" + suf)(_p(x), C.c_int64(x.size // (H * D)), C.c_int(H), C.c_int(D), ct(dt),
                                                  C.c_int({"forward": 0, "backward": 1, "central": 2}[method]), _p(out))
    return out


def traj_diff_norm_sum(x, c0, dim, prec="f64"):
    npdt, _, suf = _dt(prec)
    x = np.ascontiguousarray(x, npdt)
    B, H, S = x.shape
    out = np.empty(B, npdt)
    getattr(lib(), "orc_traj_diff_norm_sum" + suf)(_p(x), C.c_int64(B), C.c_int(H), C.c_int(S), C.c_int(c0), C.c_int(dim), _p(out))
    return out
