"""ctypes mirror of include/trk.h (struct layouts + helpers to fill them from host arrays).

Kept in one place so the product loader (`_lib.py`) and the test-only oracle wrapper
(`oracle/oracle.py`) describe models with the same bytes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

TRK_ABI_VERSION = 5
TRK_MAX_LINKS = 64
TRK_MAX_DOFS = 32
TRK_MAX_POSE_SLOTS = 8
TRK_MAX_OBJECTS = 16
TRK_MAX_PRIMS = 256
TRK_MAX_COLL_LINKS = 192
TRK_MAX_SELF_PAIRS = 1024

TRK_OK = 0
TRK_ERR_INVALID_ARG = -1
TRK_ERR_UNSUPPORTED = -2
TRK_ERR_HIP = -3
TRK_ERR_NO_DEVICE = -4

FIELD_SELF, FIELD_OBJECTS, FIELD_WS = 1, 2, 4
PRIM_SPHERE, PRIM_ROUNDED_BOX, PRIM_SHARP_BOX = 0, 1, 2

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


class KinModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_links", C.c_int32), ("n_dofs", C.c_int32), ("n_slots", C.c_int32),
        ("parent", _i32p), ("joint_type", _i32p), ("dof_idx", _i32p),
        ("R_fixed", _f32p), ("trans", _f32p), ("axis", _f32p),
        ("rot_axis", _i32p), ("rot_sign", _f32p), ("clamp", _i32p),
        ("lower", _f32p), ("upper", _f32p),
        ("sf_rot_axis", _i32p), ("sf_clamp", _i32p), ("jac_axis", _i32p), ("joint_list_idx", _i32p),
        ("order", _i32p), ("subtree_end", _i32p), ("parent_slot", _i32p), ("store_slot", _i32p),
        ("base_R", C.c_float * 9), ("base_t", C.c_float * 3),
    ]


class Primitive(C.Structure):
    _fields_ = [("type", C.c_int32), ("object", C.c_int32), ("center", C.c_float * 3),
                ("half", C.c_float * 3), ("radius", C.c_float), ("_pad", C.c_float)]


class Object(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("R", C.c_float * 9), ("prim_begin", C.c_int32),
                ("prim_end", C.c_int32), ("is_grid", C.c_int32), ("_pad", C.c_int32)]


class GridDesc(C.Structure):
    _fields_ = [("sdf", C.c_void_p), ("grad", C.c_void_p), ("dims", C.c_int32 * 3),
                ("lim_min", C.c_float * 3), ("map_dim", C.c_float * 3)]


class CostModelDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_links_in", C.c_int32),
        ("n_obj_links", C.c_int32), ("obj_link_idx", _i32p), ("obj_link_margin", _f32p),
        ("n_objects", C.c_int32), ("objects", C.POINTER(Object)),
        ("n_prims", C.c_int32), ("prims", C.POINTER(Primitive)),
        ("has_grid", C.c_int32), ("grid", GridDesc),
        ("has_ws", C.c_int32), ("ws_min", C.c_float * 3), ("ws_max", C.c_float * 3),
        ("n_self_links", C.c_int32), ("self_link_idx", _i32p),
        ("n_self_pairs", C.c_int32), ("self_pairs", _i32p), ("self_margin", _f32p),
        ("ee_link", C.c_int32), ("ee_w_pos", C.c_float), ("ee_w_rot", C.c_float), ("ee_square", C.c_int32),
        ("ee_target", C.c_float * 16),
        ("ee2_link", C.c_int32), ("ee2_target", C.c_float * 16),
        ("clamp_fields", C.c_int32), ("n_virtual", C.c_int32), ("virtual_src", _i32p), ("virtual_w", _f32p),
    ]


class RolloutWeights(C.Structure):
    _fields_ = [("w_self", C.c_float), ("w_obj", C.c_float), ("w_ws", C.c_float), ("w_ee", C.c_float)]


class ModuleUnitDesc(C.Structure):  # TrkModuleUnitDesc
    _fields_ = [("spec_abi_version", C.c_int32), ("sizeof_args", C.c_uint32), ("sizeof_cost_hdr", C.c_uint32), ("ident", C.c_char_p),
                ("model_hash", C.c_uint64), ("n_links", C.c_int32), ("n_dofs", C.c_int32),
                ("n_obj_links", C.c_int32), ("obj_link_idx", C.POINTER(C.c_int32)),
                ("n_self_pairs", C.c_int32), ("self_pairs", C.POINTER(C.c_int32)),
                ("ee_link", C.c_int32), ("ee2_link", C.c_int32),
                ("n_virtual", C.c_int32), ("virtual_src", C.POINTER(C.c_int32)), ("virtual_w", C.POINTER(C.c_float)),
                ("chunked", C.c_int32), ("fast_switch", C.c_int32), ("fkhbwd_ok", C.c_int32), ("fields_ok", C.c_int32), ("ik_ok", C.c_int32),
                ("ikgn_ok", C.c_int32), ("jac_ok", C.c_int32), ("jac_direct", C.c_int32), ("gp_ok", C.c_int32),
                ("code", C.c_void_p), ("code_size", C.c_uint64),
                ("n_kernels", C.c_int32), ("name_exprs", C.POINTER(C.c_char_p)), ("lowered_names", C.POINTER(C.c_char_p)),
                ("n_points", C.c_int32), ("points_hash", C.c_uint64)]


class GpPrior(C.Structure):         # TrkGpPrior
    _fields_ = [("dt", C.c_float), ("sigma", C.c_float), ("weight", C.c_float)]


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def kin_desc(model) -> Tuple[KinModelDesc, List[np.ndarray]]:
    """Fill a KinModelDesc from a `KinModel`; the returned list keeps the arrays alive."""
    keep: List[np.ndarray] = []
    d = KinModelDesc()
    d.abi_version = TRK_ABI_VERSION
    d.n_links, d.n_dofs, d.n_slots = model.n_links, model.n_dofs, model.n_slots
    for name in ("parent", "joint_type", "dof_idx", "rot_axis", "clamp", "sf_rot_axis", "sf_clamp",
                 "jac_axis", "joint_list_idx", "order", "subtree_end", "parent_slot", "store_slot"):
        arr = _i32(getattr(model, name)); keep.append(arr); setattr(d, name, _ptr(arr, _i32p))
    for name in ("R_fixed", "trans", "axis", "rot_sign", "lower", "upper"):
        arr = _f32(getattr(model, name)); keep.append(arr); setattr(d, name, _ptr(arr, _f32p))
    d.base_R = (C.c_float * 9)(*_f32(model.base_R).reshape(9))
    d.base_t = (C.c_float * 3)(*_f32(model.base_t).reshape(3))
    return d, keep


def cost_desc(spec, grid_ptrs=None) -> Tuple[CostModelDesc, list]:
    """Fill a CostModelDesc from a `CostModelSpec`.

    grid_ptrs: (sdf_ptr, grad_ptr) integers for the grid arrays in the address space of the
    consumer (device pointers for libtrk, host pointers for the oracle)."""
    keep: list = []
    d = CostModelDesc()
    d.abi_version = TRK_ABI_VERSION
    d.n_links_in = int(spec.n_links_in)

    def put_i(name, arr):
        a = _i32(arr); keep.append(a); setattr(d, name, _ptr(a, _i32p)); return a

    def put_f(name, arr):
        a = _f32(arr); keep.append(a); setattr(d, name, _ptr(a, _f32p)); return a

    d.n_obj_links = len(spec.obj_link_idx)
    put_i("obj_link_idx", spec.obj_link_idx)
    put_f("obj_link_margin", spec.obj_link_margin)
    objs = (Object * max(1, len(spec.objects)))()
    prims = (Primitive * max(1, sum(len(o["prims"]) for o in spec.objects)))()
    npr = 0
    for oi, o in enumerate(spec.objects):
        objs[oi].pos = (C.c_float * 3)(*_f32(o["pos"]).reshape(3))
        objs[oi].R = (C.c_float * 9)(*_f32(o["R"]).reshape(9))
        objs[oi].is_grid = int(o.get("is_grid", 0))
        objs[oi].prim_begin = npr
        for p in o["prims"]:
            prims[npr].type = int(p["type"]); prims[npr].object = oi
            prims[npr].center = (C.c_float * 3)(*_f32(p["center"]).reshape(3))
            prims[npr].half = (C.c_float * 3)(*_f32(p.get("half", [0, 0, 0])).reshape(3))
            prims[npr].radius = float(np.float32(p.get("radius", 0.0)))
            npr += 1
        objs[oi].prim_end = npr
    keep += [objs, prims]
    d.n_objects, d.objects = len(spec.objects), objs
    d.n_prims, d.prims = npr, prims
    if spec.grid is not None:
        if grid_ptrs is None:
            raise ValueError("cost model has a grid: pass grid_ptrs")
        d.has_grid = 1
        d.grid.sdf, d.grid.grad = int(grid_ptrs[0]), int(grid_ptrs[1])
        d.grid.dims = (C.c_int32 * 3)(*[int(v) for v in spec.grid["dims"]])
        d.grid.lim_min = (C.c_float * 3)(*_f32(spec.grid["lim_min"]).reshape(3))
        d.grid.map_dim = (C.c_float * 3)(*_f32(spec.grid["map_dim"]).reshape(3))
    if spec.ws_min is not None:
        d.has_ws = 1
        d.ws_min = (C.c_float * 3)(*_f32(spec.ws_min).reshape(3))
        d.ws_max = (C.c_float * 3)(*_f32(spec.ws_max).reshape(3))
    d.n_self_links = len(spec.self_link_idx)
    put_i("self_link_idx", spec.self_link_idx)
    d.n_self_pairs = len(spec.self_margin)
    put_i("self_pairs", np.asarray(spec.self_pairs, np.int32).reshape(-1))
    put_f("self_margin", spec.self_margin)
    d.ee_link = int(spec.ee_link)
    d.ee_w_pos, d.ee_w_rot, d.ee_square = float(spec.ee_w_pos), float(spec.ee_w_rot), int(bool(spec.ee_square))
    d.ee_target = (C.c_float * 16)(*_f32(spec.ee_target).reshape(16))
    d.ee2_link = int(spec.ee2_link)
    d.clamp_fields = int(spec.clamp_fields)
    vsrc = np.asarray(spec.virtual_src, np.int32).reshape(-1, 2)
    d.n_virtual = int(vsrc.shape[0])
    put_i("virtual_src", vsrc.reshape(-1))
    put_f("virtual_w", np.asarray(spec.virtual_w, np.float32).reshape(-1))
    d.ee2_target = (C.c_float * 16)(*_f32(spec.ee2_target).reshape(16))
    return d, keep
