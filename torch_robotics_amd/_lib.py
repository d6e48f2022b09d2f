"""Loader for csrc/libtrk.so (the C ABI in include/trk.h).

The library is the only compute path of this package.  If it is missing, cannot be loaded, or
finds no GPU, calls raise -- there is deliberately no CPU / eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import torch  # noqa: F401  (must be imported first: libtrk.so binds to the HIP runtime torch loaded)

from . import _abi

_CSRC = Path(__file__).resolve().parent / "csrc"
# TRK_LIBTRK=<path> loads another build of the same ABI (same-box A/B measurements of two kernel versions)
LIB_PATH = Path(os.environ["TRK_LIBTRK"]).resolve() if os.environ.get("TRK_LIBTRK") else _CSRC / "libtrk.so"
_lib = None

EXPORTS = [
    "trk_abi_version", "trk_last_error", "trk_model_create", "trk_model_destroy", "trk_model_set_base_pose",
    "trk_model_n_links", "trk_model_n_dofs", "trk_model_is_specialized", "trk_model_enable_specialized", "trk_spec_count",
    "trk_fk_forward", "trk_fk_positions", "trk_fk_backward", "trk_fk_positions_backward", "trk_fk_jacobian", "trk_fk_analytic_jacobian", "trk_ik_step", "trk_ik_steps",
    "trk_rotmat_to_quat", "trk_cost_model_create", "trk_cost_model_destroy", "trk_cost_model_set_ee_target", "trk_cost_model_set_ee2_target", "trk_cost_model_enable_specialized",
    "trk_cost_fields", "trk_collision_fields", "trk_ee_cost", "trk_rollout_cost_grad", "trk_reduce_sum", "trk_debug_set_stamp_buffer", "trk_interpolate_via_points",
    "trk_grid_precompute",
    "trk_frame_compose", "trk_frame_compose_backward", "trk_frame_transform_points", "trk_frame_transform_points_backward",
    "trk_frame_quat_euler", "trk_frame_quat_euler_backward", "trk_rotation_from", "trk_rotation_from_backward",
    "trk_sdf_points",
    "trk_point_set_create", "trk_point_set_destroy", "trk_point_set_size", "trk_point_set_is_specialized", "trk_fk_points", "trk_fk_points_backward",
    "trk_rollout_points_cost_grad", "trk_rollout_collision", "trk_gp_prior_cost_grad", "trk_rollout_cost_grad_f16", "trk_finite_difference", "trk_traj_diff_norm_sum",
    "trk_interpolate_columns", "trk_interpolate_columns_backward", "trk_rollout_collision_via", "trk_rollout_collision_via_flags", "trk_via_partial_flags_bytes", "trk_traj_validate",
    "trk_scale_rows", "trk_jtj", "trk_pack_sums", "trk_pack_sums_scratch_bytes", "trk_rollout_is_specialized", "trk_ik_gn_steps", "trk_rollout_gp_cost_grad",
    "trk_spec_register_module", "trk_spec_layout_stamp", "trk_last_dispatch", "trk_set_strict_specialized", "trk_rollout_points_is_specialized", "trk_rollout_jacobian_cost_grad",
    "trk_handle_kind", "trk_mailbox_create", "trk_mailbox_ipc_handle", "trk_mailbox_connect", "trk_mailbox_send", "trk_mailbox_recv", "trk_mailbox_exchange", "trk_mailbox_status", "trk_mailbox_destroy",
]


class TrkError(RuntimeError):
    pass


def build(verbose: bool = False) -> Path:
    """Compile libtrk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    from . import codegen
    codegen.generate_all(_CSRC / "generated")          # model-specialised kernels (source is a build product)
    res = subprocess.run(["make", "-C", str(_CSRC), "-j", jobs], capture_output=True, text=True)
    if res.returncode != 0:
        raise TrkError(f"building libtrk.so failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(res.stdout)
    # the native dispatcher ops are optional (torch_ops() returns None without them): a box without g++ or the torch headers keeps libtrk.so
    res2 = subprocess.run(["make", "-C", str(_CSRC), "torch_ops"], capture_output=True, text=True)
    if res2.returncode != 0:
        import warnings
        warnings.warn("libtrk_torch.so (native PyTorch dispatcher ops) was not built; the Python-registered ops serve instead:\n"
                      + (res2.stdout + res2.stderr)[-2000:])
    return LIB_PATH


TORCH_OPS_PATH = _CSRC / "libtrk_torch.so"
_torch_ops_loaded = None


def torch_ops():
    """`torch.ops.trk` with the NATIVE ops of csrc/trk_torch_ops.cpp registered (trk::rollout, trk::scale_rows_native), or None when
    libtrk_torch.so has not been built -- the callers then use the Python-registered ops / autograd Functions over the same kernels."""
    global _torch_ops_loaded
    if _torch_ops_loaded is None:
        _torch_ops_loaded = False
        if TORCH_OPS_PATH.exists() and os.environ.get("TRK_NO_NATIVE_OPS", "0") != "1":
            lib()                                   # libtrk.so first: the op library links against it
            try:
                torch.ops.load_library(str(TORCH_OPS_PATH))
                _torch_ops_loaded = True
            except OSError as e:                    # e.g. built against another torch: say so once, use the Python-registered ops
                import warnings
                warnings.warn(f"{TORCH_OPS_PATH.name} could not be loaded ({e}); using the Python-registered dispatcher ops")
    return torch.ops.trk if _torch_ops_loaded else None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise TrkError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {_CSRC}`. There is no fallback path.")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.trk_abi_version.restype = C.c_int
    L.trk_last_error.restype = C.c_char_p
    L.trk_model_create.argtypes = [C.POINTER(_abi.KinModelDesc), C.POINTER(vp)]
    L.trk_model_destroy.argtypes = [vp]
    L.trk_model_destroy.restype = None
    L.trk_model_set_base_pose.argtypes = [vp, vp, vp]
    L.trk_model_n_links.argtypes = [vp]
    L.trk_model_n_dofs.argtypes = [vp]
    L.trk_model_is_specialized.argtypes = [vp]
    L.trk_model_enable_specialized.argtypes = [vp, C.c_int]
    L.trk_spec_count.argtypes = []
    L.trk_spec_register_module.argtypes = [C.POINTER(_abi.ModuleUnitDesc)]
    L.trk_spec_layout_stamp.argtypes = [C.POINTER(C.c_int64)]
    L.trk_fk_forward.argtypes = [vp, vp, i64, vp, i32, vp, vp]
    L.trk_fk_positions.argtypes = [vp, vp, i64, vp, i32, vp, vp]
    L.trk_fk_backward.argtypes = [vp, vp, vp, i64, vp, i32, vp, vp]
    L.trk_fk_positions_backward.argtypes = [vp, vp, vp, i64, vp, i32, vp, vp]
    L.trk_fk_jacobian.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp]
    L.trk_fk_analytic_jacobian.argtypes = [vp, vp, i64, vp, vp]
    L.trk_ik_step.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, f32, i32, i64, vp, vp, vp, vp, vp, vp]
    L.trk_ik_steps.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, f32, i32, i32, i64, vp, vp, vp, vp, vp, vp]
    L.trk_rotmat_to_quat.argtypes = [vp, i64, i32, i32, vp, vp]
    L.trk_rotation_from.argtypes = [i32, vp, i64, vp, vp]
    L.trk_rotation_from_backward.argtypes = [i32, vp, vp, i64, vp, vp]
    L.trk_frame_compose.argtypes = [i32, vp, vp, i64, vp, vp, i64, vp, vp, vp]
    L.trk_frame_compose_backward.argtypes = [i32, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.trk_frame_transform_points.argtypes = [vp, vp, i64, vp, i32, vp, vp]
    L.trk_frame_transform_points_backward.argtypes = [vp, i64, vp, i32, vp, vp, vp]
    L.trk_frame_quat_euler.argtypes = [vp, i64, i32, i32, vp, vp, vp]
    L.trk_frame_quat_euler_backward.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp]
    L.trk_cost_model_create.argtypes = [C.POINTER(_abi.CostModelDesc), C.POINTER(vp)]
    L.trk_cost_model_destroy.argtypes = [vp]
    L.trk_cost_model_destroy.restype = None
    L.trk_cost_model_set_ee_target.argtypes = [vp, vp]
    L.trk_cost_model_set_ee2_target.argtypes = [vp, vp]
    L.trk_cost_model_enable_specialized.argtypes = [vp, i32]
    L.trk_cost_fields.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp]
    L.trk_collision_fields.argtypes = [vp, i32, vp, i64, f32, vp, vp]
    L.trk_ee_cost.argtypes = [vp, vp, i64, i64, vp, i32, vp, vp, vp, i64, vp]
    L.trk_rollout_cost_grad.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights), vp, i64, i32, vp, vp, vp, vp, vp]
    L.trk_rollout_collision.argtypes = [vp, vp, i32, vp, i64, i32, f32, vp, vp, vp]
    L.trk_rollout_gp_cost_grad.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights), C.POINTER(_abi.GpPrior), vp, vp, i64, i32, i32, vp, vp, vp, vp,
                                           i32, f32, vp, vp]
    L.trk_ik_gn_steps.argtypes = [vp, i32, vp, i32, vp, vp, f32, f32, f32, f32, i32, i64, vp, vp, vp, vp]
    L.trk_rollout_is_specialized.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights)]
    L.trk_last_dispatch.argtypes = []
    L.trk_rollout_jacobian_cost_grad.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights), vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.trk_rollout_points_is_specialized.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights)]
    L.trk_set_strict_specialized.argtypes = [i32]
    L.trk_interpolate_via_points.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp]
    L.trk_debug_set_stamp_buffer.argtypes = [vp]
    L.trk_reduce_sum.argtypes = [vp, i64, vp, vp]
    L.trk_grid_precompute.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.trk_sdf_points.argtypes = [vp, vp, i64, vp, vp, vp]
    L.trk_rollout_cost_grad_f16.argtypes = [vp, vp, C.POINTER(_abi.RolloutWeights), vp, i64, i32, vp, vp, vp, i32, f32, vp, vp]
    L.trk_finite_difference.argtypes = [vp, i64, i32, i32, f32, i32, vp, vp]
    L.trk_traj_diff_norm_sum.argtypes = [vp, i64, i32, i32, i32, i32, vp, vp]
    L.trk_gp_prior_cost_grad.argtypes = [vp, vp, i64, i32, i32, i32, f32, f32, f32, vp, vp, vp, i32, f32, i32, vp]
    L.trk_interpolate_columns.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp]
    L.trk_interpolate_columns_backward.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp]
    L.trk_rollout_collision_via.argtypes = [vp, vp, i32, vp, i64, i32, i32, i32, vp, vp, f32, vp, vp]
    L.trk_rollout_collision_via_flags.argtypes = [vp, vp, i32, vp, i64, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp]
    L.trk_via_partial_flags_bytes.argtypes = [i64, i32, i32]
    L.trk_via_partial_flags_bytes.restype = C.c_int64
    L.trk_traj_validate.argtypes = [vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, i32, vp, vp]
    L.trk_pack_sums.argtypes = [vp, vp, i32, f32, vp, vp, i64, i32, i32, vp, vp, vp]
    L.trk_pack_sums_scratch_bytes.argtypes = [i32, i32]
    L.trk_pack_sums_scratch_bytes.restype = C.c_int64
    L.trk_handle_kind.argtypes = [vp]
    L.trk_mailbox_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.trk_mailbox_ipc_handle.argtypes = [vp, vp]
    L.trk_mailbox_connect.argtypes = [vp, vp]
    L.trk_mailbox_exchange.argtypes = [vp, vp, vp, vp]
    L.trk_mailbox_send.argtypes = [vp, vp, vp]
    L.trk_mailbox_recv.argtypes = [vp, vp, vp]
    L.trk_mailbox_status.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    L.trk_mailbox_destroy.argtypes = [vp]
    L.trk_mailbox_destroy.restype = None
    L.trk_jtj.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, i32, vp, vp]
    L.trk_scale_rows.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp]
    L.trk_point_set_create.argtypes = [vp, vp, vp, i32, C.POINTER(vp)]
    L.trk_point_set_destroy.argtypes = [vp]
    L.trk_point_set_destroy.restype = None
    L.trk_point_set_size.argtypes = [vp]
    L.trk_point_set_is_specialized.argtypes = [vp]
    L.trk_fk_points.argtypes = [vp, vp, vp, i64, vp, vp]
    L.trk_fk_points_backward.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.trk_rollout_points_cost_grad.argtypes = [vp, vp, vp, C.POINTER(_abi.RolloutWeights), vp, i64, i32, vp, vp, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(L, name)        # AttributeError here = the library does not export the ABI
        if name not in ("trk_last_error", "trk_model_destroy", "trk_cost_model_destroy", "trk_point_set_destroy",
                        "trk_pack_sums_scratch_bytes", "trk_mailbox_destroy", "trk_via_partial_flags_bytes"):
            fn.restype = C.c_int
    if L.trk_abi_version() != _abi.TRK_ABI_VERSION:
        raise TrkError("libtrk.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def check(rc: int, what: str = "libtrk") -> None:
    if rc == _abi.TRK_OK:
        return
    msg = lib().trk_last_error().decode("utf-8", "replace")
    if rc == _abi.TRK_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == _abi.TRK_ERR_INVALID_ARG:
        raise ValueError(f"{what}: {msg}")
    raise TrkError(f"{what}: {msg} (status {rc})")
