"""Run-time model compiler: a generated fused kernel for ANY robot + collision template.

`csrc/generated/` holds the units built ahead of time for the robots of the benchmark configs.  For any other
`KinModel` (or another collision-link template of the same robot) `specialize()` runs the same generator
(`codegen.generate_rollout_source`), compiles the unit with hipcc for gfx950 into `csrc/jit/spec_<ident>.so`, and loads it;
the unit's static initialiser registers it with libtrk.so (`trk_spec_register`), and `trk_rollout_cost_grad` /
`trk_fk_positions(_backward)` pick it up by the model hash.  Nothing changes for the caller except the speed
(table-driven -> generated: ~10x on the fused rollout).  Compilation takes tens of seconds per robot and is cached on disk
by (model hash, template hash, generator source hash).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib, codegen
from .kinmodel import KinModel

_CSRC = Path(__file__).resolve().parent / "csrc"
# Units are cached in the package tree by default (they then travel with it, like libtrk.so); TRK_JIT_DIR moves the cache,
# e.g. to a per-user directory when the package is installed read-only or shared between users.
JIT_DIR = Path(os.environ["TRK_JIT_DIR"]).resolve() if os.environ.get("TRK_JIT_DIR") else _CSRC / "jit"
# same code-generation flags as csrc/Makefile uses for the ahead-of-time units
GENFLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-Xarch_device", "-fno-slp-vectorize"]
_REPO_INCLUDE = _CSRC.parent.parent / "include"
_loaded: Dict[str, C.CDLL] = {}


HIPCC = os.environ.get("TRK_HIPCC", "/opt/rocm/bin/hipcc")      # the compiler driver of the run-time units


def _compile_cmd(src: str, out: str):
    return [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
            "-Wno-unused-variable", "-Wno-pass-failed", "-ffp-contract=off", f"-I{_CSRC}", "-Xarch_device", "-fno-honor-nans",
            "-Xarch_device", "-mno-amdgpu-ieee", *GENFLAGS, "-shared", src, "-o", out, f"-L{_CSRC}", "-ltrk",
            "-Wl,-rpath,$ORIGIN/.." if JIT_DIR == _CSRC / "jit" else f"-Wl,-rpath,{_CSRC}"]


def template_hash(tmpl: codegen.CollisionTemplate) -> str:
    h = hashlib.sha1()
    h.update(np.asarray(tmpl.obj_links, np.int32).tobytes())
    h.update(np.asarray(tmpl.self_pairs, np.int32).reshape(-1).tobytes())
    h.update(np.asarray([tmpl.ee_link, tmpl.ee2_link], np.int32).tobytes())
    if tmpl.virtual:
        h.update(np.asarray([r[:2] for r in tmpl.virtual], np.int32).tobytes())
        h.update(np.asarray([r[2:] for r in tmpl.virtual], np.float32).tobytes())
    return h.hexdigest()[:8]


def _generator_stamp() -> str:
    """Everything a cached unit's machine code and struct layouts depend on: the generator, EVERY header the unit includes
    (trk_device.h pulls in include/trk.h: TrkRolloutWeights, the TRK_MAX_* limits, TRK_ABI_VERSION), and the full compile
    command.  A unit whose stamp differs is recompiled; should one slip through anyway (hand-copied cache), its SpecEntry
    carries the layout stamp and `trk_spec_register` refuses it (see `_load_unit`)."""
    h = hashlib.sha1()
    for f in (Path(codegen.__file__), _CSRC / "trk_spec_common.h", _CSRC / "trk_device.h", _CSRC / "trk_launch.h",
              _REPO_INCLUDE / "trk.h"):
        h.update(f.read_bytes())
    h.update(" ".join(_compile_cmd("<src>", "<out>")).replace(str(_CSRC), "<csrc>").encode())   # location-independent
    # generation-time experiment knobs change the text the generator writes: a unit generated under one is not the default unit
    knobs = sorted((k, v) for k, v in os.environ.items() if k.startswith(("TRK_EXP_", "TRK_GP_SCHEDULE")))
    if knobs:
        h.update(repr(knobs).encode())
    return h.hexdigest()[:12]


def _load_unit(so: Path, ident: str) -> C.CDLL:
    """dlopen a unit and make sure libtrk.so accepted its registration (it refuses another struct layout)."""
    L = _lib.lib()                                  # libtrk.so first: the unit's initialiser calls into it
    before = L.trk_spec_count()
    handle = C.CDLL(str(so))
    if L.trk_spec_count() != before + 1:
        raise _lib.TrkError(f"{so.name}: libtrk.so refused the unit (compiled against another SpecArgs/SpecEntry layout); "
                            f"delete {so} and retry")
    return handle


def _compile_unit(source: str, ident: str, verbose: bool) -> Path:
    """source -> csrc/jit/spec_<ident>.so.  Every file appears under its final name by an atomic rename, so several
    processes (one per GPU) compiling the same unit at the same time cannot hand each other a half-written object."""
    JIT_DIR.mkdir(parents=True, exist_ok=True)
    src, so, stamp = JIT_DIR / f"spec_{ident}.hip", JIT_DIR / f"spec_{ident}.so", JIT_DIR / f"spec_{ident}.stamp"
    want = _generator_stamp()
    if so.exists() and stamp.exists() and stamp.read_text() == want:
        return so
    if not _lib.LIB_PATH.exists():
        raise _lib.TrkError(f"{_lib.LIB_PATH} not found: build libtrk.so first")
    tag = f".tmp{os.getpid()}"
    src_tmp = JIT_DIR / f"spec_{ident}{tag}.hip"
    so_tmp = JIT_DIR / f"spec_{ident}{tag}.so"
    src_tmp.write_text(source)
    cmd = _compile_cmd(str(src_tmp), str(so_tmp))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        src_tmp.unlink(missing_ok=True)
        raise _lib.TrkError(f"compiling spec_{ident}.hip failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(res.stderr)
    os.replace(src_tmp, src)
    os.replace(so_tmp, so)
    stamp_tmp = JIT_DIR / f"spec_{ident}{tag}.stamp"
    stamp_tmp.write_text(want)
    os.replace(stamp_tmp, stamp)
    return so


# ----------------------------------------------------------------------------------------------------------------------
# hipRTC fall-back: no hipcc on the box (a runtime-only ROCm installation has libhiprtc.so but no compiler driver).  The unit's
# DEVICE half is compiled in-process to a code object; libtrk.so's generic launchers play its host half
# (trk_spec_register_module).  Same generator, same headers, same device flags; the code object is cached next to the .so units.
# ----------------------------------------------------------------------------------------------------------------------
RTC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-honor-nans",
             "-mno-amdgpu-ieee", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
_rtc_keep = []          # descriptors' buffers of the registered code-object units (libtrk.so copies what it needs; the code stays mapped)


def hipcc_available() -> bool:
    return os.path.exists(HIPCC) and os.access(HIPCC, os.X_OK)


def _hiprtc():
    if os.environ.get("TRK_NO_HIPRTC", "0") == "1":
        raise _lib.TrkError("no hipcc, and the hipRTC fall-back is switched off (TRK_NO_HIPRTC=1)")
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so", os.path.join(os.path.dirname(__import__("torch").__file__), "lib", "libhiprtc.so")):
        try:
            L = C.CDLL(name)
            L.hiprtcGetErrorString.restype = C.c_char_p
            return L
        except OSError:
            continue
    raise _lib.TrkError("neither hipcc nor libhiprtc.so is available: a generated unit cannot be compiled on this box")


def _rtc_compile(source: str, ident: str, kernels) -> Tuple[bytes, list]:
    """device half of a unit -> (code object, lowered kernel names in the order of `kernels`)"""
    R = _hiprtc()

    def ck(rc, what):
        if rc != 0:
            raise _lib.TrkError(f"hipRTC {what}: {R.hiprtcGetErrorString(rc).decode()}")
    prog = C.c_void_p()
    ck(R.hiprtcCreateProgram(C.byref(prog), source.encode(), f"spec_{ident}.hip".encode(), 0, None, None), "hiprtcCreateProgram")
    try:
        for k in kernels:
            ck(R.hiprtcAddNameExpression(prog, k.encode()), f"hiprtcAddNameExpression({k})")
        opts = [o.encode() for o in RTC_FLAGS + [f"-I{_CSRC}", f"-I{_REPO_INCLUDE}"]]
        arr = (C.c_char_p * len(opts))(*opts)
        rc = R.hiprtcCompileProgram(prog, len(opts), arr)
        if rc != 0:
            n = C.c_size_t()
            R.hiprtcGetProgramLogSize(prog, C.byref(n))
            log = C.create_string_buffer(max(1, n.value))
            R.hiprtcGetProgramLog(prog, log)
            raise _lib.TrkError(f"hipRTC compilation of spec_{ident}.hip failed:\n{log.value.decode()[-4000:]}")
        sz = C.c_size_t()
        ck(R.hiprtcGetCodeSize(prog, C.byref(sz)), "hiprtcGetCodeSize")
        code = C.create_string_buffer(sz.value)
        ck(R.hiprtcGetCode(prog, code), "hiprtcGetCode")
        lowered = []
        for k in kernels:
            p = C.c_char_p()
            ck(R.hiprtcGetLoweredName(prog, k.encode(), C.byref(p)), f"hiprtcGetLoweredName({k})")
            lowered.append(p.value.decode())
        return code.raw, lowered
    finally:
        R.hiprtcDestroyProgram(C.byref(prog))


def _rtc_stamp() -> str:
    h = hashlib.sha1(_generator_stamp().encode())
    h.update(" ".join(RTC_FLAGS).encode())
    return "rtc-" + h.hexdigest()[:12]


def _rtc_code_object(source: str, ident: str, kernels) -> tuple:
    """hipRTC-compile `source` (or take the code object from the cache): (code bytes, lowered kernel names)"""
    import json
    JIT_DIR.mkdir(parents=True, exist_ok=True)
    co, js = JIT_DIR / f"spec_{ident}.hsaco", JIT_DIR / f"spec_{ident}.rtc.json"
    want = _rtc_stamp()
    code = lowered = None
    if co.exists() and js.exists():
        try:
            rec = json.loads(js.read_text())
            if rec.get("stamp") == want and rec.get("kernels") == list(kernels):
                code, lowered = co.read_bytes(), rec["lowered"]
        except (OSError, ValueError):
            pass
    if code is None:
        code, lowered = _rtc_compile(source, ident, list(kernels))
        tag = f".tmp{os.getpid()}"
        (JIT_DIR / f"spec_{ident}{tag}.hsaco").write_bytes(code)
        os.replace(JIT_DIR / f"spec_{ident}{tag}.hsaco", co)
        (JIT_DIR / f"spec_{ident}{tag}.json").write_text(json.dumps({"stamp": want, "kernels": list(kernels), "lowered": lowered}))
        os.replace(JIT_DIR / f"spec_{ident}{tag}.json", js)
    return code, lowered


def _load_points_unit_rtc(kin: KinModel, pt: "codegen.PointsTemplate", ident: str) -> object:
    """An ATTACHED-POINT unit (link spheres, grasped-object points) without hipcc: its device half compiled in-process, libtrk.so's
    generic launchers as its host half (trk_spec_register_module with n_points > 0)."""
    meta: dict = {}
    source = codegen.generate_points_rollout_source(kin, pt, ident, meta=meta)
    code, lowered = _rtc_code_object(source, ident, meta["kernels"])
    L = _lib.lib()
    from . import _abi
    stamp = (C.c_int64 * 3)()
    L.trk_spec_layout_stamp(stamp)
    d = _abi.ModuleUnitDesc()
    d.spec_abi_version, d.sizeof_args, d.sizeof_cost_hdr = int(stamp[0]), int(stamp[1]), int(stamp[2])
    d.ident = ident.encode()
    d.model_hash = codegen.model_hash(kin)
    d.n_links, d.n_dofs = kin.n_links, kin.n_dofs
    obj = np.ascontiguousarray(pt.obj_cols, np.int32)
    pairs = np.ascontiguousarray(pt.self_pairs, np.int32).reshape(-1)
    i32p = C.POINTER(C.c_int32)
    d.n_obj_links, d.obj_link_idx = len(obj), obj.ctypes.data_as(i32p)
    d.n_self_pairs, d.self_pairs = len(pairs) // 2, pairs.ctypes.data_as(i32p)
    d.ee_link, d.ee2_link = int(pt.ee_link), int(pt.ee2_link)
    d.n_virtual = 0
    d.n_points, d.points_hash = len(pt.point_link), codegen.points_hash(pt.point_link, pt.point_offset)
    buf = C.create_string_buffer(code, len(code))
    d.code, d.code_size = C.cast(buf, C.c_void_p), len(code)
    names = (C.c_char_p * len(meta["kernels"]))(*[k.encode() for k in meta["kernels"]])
    lows = (C.c_char_p * len(lowered))(*[k.encode() for k in lowered])
    d.n_kernels, d.name_exprs, d.lowered_names = len(lowered), names, lows
    before = L.trk_spec_count()
    _lib.check(L.trk_spec_register_module(C.byref(d)), "trk_spec_register_module")
    if L.trk_spec_count() != before + 1:
        raise _lib.TrkError(f"spec_{ident}: libtrk.so refused the code-object unit")
    _rtc_keep.append((buf, names, lows, obj, pairs, d))
    return d


def _load_unit_rtc(kin: KinModel, tmpl: codegen.CollisionTemplate, ident: str) -> object:
    """generate + hipRTC-compile (or take from the cache) + register the code object with libtrk.so"""
    import json
    JIT_DIR.mkdir(parents=True, exist_ok=True)
    co, js = JIT_DIR / f"spec_{ident}.hsaco", JIT_DIR / f"spec_{ident}.rtc.json"
    meta: dict = {}
    source = codegen.generate_link_kernel_source(kin, tmpl, ident, meta=meta)
    if not meta:
        raise _lib.TrkError("this robot's link unit comes from the per-link pipeline generator, which the hipRTC fall-back does not serve")
    want = _rtc_stamp()
    code = lowered = None
    if co.exists() and js.exists():
        try:
            rec = json.loads(js.read_text())
            if rec.get("stamp") == want and rec.get("kernels") == meta["kernels"]:
                code, lowered = co.read_bytes(), rec["lowered"]
        except (OSError, ValueError):
            pass
    if code is None:
        code, lowered = _rtc_compile(source, ident, meta["kernels"])
        tag = f".tmp{os.getpid()}"
        (JIT_DIR / f"spec_{ident}{tag}.hsaco").write_bytes(code)
        os.replace(JIT_DIR / f"spec_{ident}{tag}.hsaco", co)
        (JIT_DIR / f"spec_{ident}{tag}.json").write_text(json.dumps({"stamp": want, "kernels": meta["kernels"], "lowered": lowered}))
        os.replace(JIT_DIR / f"spec_{ident}{tag}.json", js)
    L = _lib.lib()
    from . import _abi
    stamp = (C.c_int64 * 3)()
    L.trk_spec_layout_stamp(stamp)
    d = _abi.ModuleUnitDesc()
    d.spec_abi_version, d.sizeof_args, d.sizeof_cost_hdr = int(stamp[0]), int(stamp[1]), int(stamp[2])
    d.ident = ident.encode()
    d.model_hash = codegen.model_hash(kin)
    d.n_links, d.n_dofs = kin.n_links, kin.n_dofs
    obj = np.ascontiguousarray(tmpl.obj_links, np.int32)
    pairs = np.ascontiguousarray(tmpl.self_pairs, np.int32).reshape(-1)
    vsrc = np.ascontiguousarray([r[:2] for r in tmpl.virtual], np.int32).reshape(-1)
    vw = np.ascontiguousarray([r[2:] for r in tmpl.virtual], np.float32).reshape(-1)
    i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    d.n_obj_links, d.obj_link_idx = len(obj), obj.ctypes.data_as(i32p)
    d.n_self_pairs, d.self_pairs = len(pairs) // 2, pairs.ctypes.data_as(i32p)
    d.ee_link, d.ee2_link = int(tmpl.ee_link), int(tmpl.ee2_link)
    d.n_virtual, d.virtual_src, d.virtual_w = len(tmpl.virtual), vsrc.ctypes.data_as(i32p), vw.ctypes.data_as(f32p)
    for k in ("chunked", "fast_switch", "fkhbwd_ok", "fields_ok", "ik_ok", "ikgn_ok", "jac_ok", "jac_direct", "gp_ok"):
        setattr(d, k, int(bool(meta[k])))
    buf = C.create_string_buffer(code, len(code))
    d.code, d.code_size = C.cast(buf, C.c_void_p), len(code)
    names = (C.c_char_p * len(meta["kernels"]))(*[k.encode() for k in meta["kernels"]])
    lows = (C.c_char_p * len(lowered))(*[k.encode() for k in lowered])
    d.n_kernels, d.name_exprs, d.lowered_names = len(lowered), names, lows
    before = L.trk_spec_count()
    _lib.check(L.trk_spec_register_module(C.byref(d)), "trk_spec_register_module")
    if L.trk_spec_count() != before + 1:
        raise _lib.TrkError(f"spec_{ident}: libtrk.so refused the code-object unit")
    _rtc_keep.append((buf, names, lows, obj, pairs, vsrc, vw, d))
    return d


def unit_ident(kin: KinModel, tmpl: codegen.CollisionTemplate, pipeline: bool = False) -> str:
    return f"jit_{codegen.model_hash(kin):016x}_{template_hash(tmpl)}" + ("_p" if pipeline else "")


def build_unit(kin: KinModel, tmpl: codegen.CollisionTemplate, verbose: bool = False, pipeline: bool = False) -> Path:
    """Generate + compile (no GPU needed: hipcc cross-compiles); returns the path of the shared object.
    pipeline: force the per-link pipeline generator (codegen.generate_points_rollout_source in link mode)."""
    ident = unit_ident(kin, tmpl, pipeline)
    so, stamp = JIT_DIR / f"spec_{ident}.so", JIT_DIR / f"spec_{ident}.stamp"
    if so.exists() and stamp.exists() and stamp.read_text() == _generator_stamp():
        return so
    if pipeline:
        source = codegen.generate_points_rollout_source(kin, codegen.link_points_template(kin, tmpl), ident, link_mode=True)
    else:
        source = codegen.generate_link_kernel_source(kin, tmpl, ident)
    return _compile_unit(source, ident, verbose)


def specialize(kin: KinModel, obj_links: Sequence[int], self_pairs: Sequence[Tuple[int, int]] = (), ee_link: int = -1,
               verbose: bool = False, pipeline: bool = False, ee2_link: int = -1, virtual=()) -> str:
    """Make sure a generated fused kernel for (kin, collision template) is registered with libtrk.so.  Idempotent.
    Returns the unit's identifier.  Robots that already have an ahead-of-time unit with the same template need nothing."""
    tmpl = codegen.CollisionTemplate(obj_links=[int(i) for i in obj_links],
                                     self_pairs=[(int(a), int(b)) for a, b in self_pairs], ee_link=int(ee_link),
                                     ee2_link=int(ee2_link), virtual=[tuple(r) for r in virtual])
    ident = unit_ident(kin, tmpl, pipeline)
    if ident not in _loaded:
        if hipcc_available() or pipeline:
            _loaded[ident] = _load_unit(build_unit(kin, tmpl, verbose, pipeline), ident)
        else:                               # no compiler driver on this box: the in-process fall-back
            _loaded[ident] = _load_unit_rtc(kin, tmpl, ident)
        _loaded_templates[ident] = (codegen.model_hash(kin), tmpl)
    return ident


def _virtual_rows(spec):
    src = np.asarray(getattr(spec, "virtual_src", ()), np.int32).reshape(-1, 2)
    w = np.asarray(getattr(spec, "virtual_w", ()), np.float32).reshape(-1, 2)
    return [(int(a), int(b), float(wa), float(wb)) for (a, b), (wa, wb) in zip(src, w)]


def generatable(spec, points: bool = False) -> bool:
    """False for cost models only the table-driven kernels evaluate (the dispatcher's rule, trk_capi.hip: spec_matches): the
    single-link self distance (a degenerate pair) and -- for attached-point units -- interpolated (virtual) position columns."""
    if points and _virtual_rows(spec):
        return False
    sl = np.asarray(spec.self_link_idx, np.int32)
    return not any(int(sl[a]) == int(sl[b]) for a, b in np.asarray(spec.self_pairs, np.int32).reshape(-1, 2))


def _template_of(kin: KinModel, spec) -> Optional[codegen.CollisionTemplate]:
    if spec.n_links_in != kin.n_links or not generatable(spec):
        return None
    sl = np.asarray(spec.self_link_idx, np.int32)
    pairs = [(int(sl[a]), int(sl[b])) for a, b in np.asarray(spec.self_pairs, np.int32).reshape(-1, 2)]
    return codegen.CollisionTemplate(obj_links=[int(i) for i in spec.obj_link_idx], self_pairs=pairs,
                                     ee_link=int(spec.ee_link), ee2_link=int(spec.ee2_link), virtual=_virtual_rows(spec))


def _serves(unit: codegen.CollisionTemplate, want: codegen.CollisionTemplate) -> bool:
    """the dispatcher's rule (trk_capi.hip: spec_matches) for a caller that may use every weight"""
    if list(unit.obj_links) != list(want.obj_links) or [tuple(p) for p in unit.self_pairs] != [tuple(p) for p in want.self_pairs]:
        return False
    if [tuple(np.float32(v) for v in r) for r in unit.virtual] != [tuple(np.float32(v) for v in r) for r in want.virtual]:
        return False
    return want.ee_link < 0 or (unit.ee_link == want.ee_link and unit.ee2_link == want.ee2_link)


_loaded_templates: Dict[str, Tuple[int, codegen.CollisionTemplate]] = {}


def has_matching_unit(kin: KinModel, spec) -> bool:
    """True if an ahead-of-time unit (codegen.SPEC_ROBOTS) or an already loaded run-time unit serves this robot and
    collision model."""
    want = _template_of(kin, spec)
    if want is None:
        return False
    h = codegen.model_hash(kin)
    for mh, tm in _loaded_templates.values():
        if mh == h and _serves(tm, want):
            return True
    for _ident, mh, t2 in codegen.aot_units():
        if mh == h and _serves(t2, want):
            return True
    return False


def specialize_for_cost_spec(kin: KinModel, spec, verbose: bool = False) -> Optional[str]:
    """Template from a CostModelSpec whose columns are the links (no attached points)."""
    if spec.n_links_in != kin.n_links or not generatable(spec):
        return None
    sl = np.asarray(spec.self_link_idx, np.int32)
    pairs = [(int(sl[a]), int(sl[b])) for a, b in np.asarray(spec.self_pairs, np.int32).reshape(-1, 2)]
    return specialize(kin, [int(i) for i in spec.obj_link_idx], pairs, int(spec.ee_link), verbose, ee2_link=int(spec.ee2_link),
                      virtual=_virtual_rows(spec))


# ----------------------------------------------------------------------------------------------------------------------
# attached-point units (link spheres, grasped-object points): same mechanism, codegen.generate_points_rollout_source
# ----------------------------------------------------------------------------------------------------------------------
def _points_template_of(kin: KinModel, point_link, point_offset, spec) -> Optional[codegen.PointsTemplate]:
    pl = np.ascontiguousarray(point_link, np.int32).reshape(-1)
    po = np.ascontiguousarray(point_offset, np.float32).reshape(-1, 3)
    if spec.n_links_in != len(pl) or not generatable(spec, points=True):
        return None
    pos_of = {int(kin.order[p]): p for p in range(kin.n_links)}
    rank = [pos_of[int(i)] for i in pl]
    obj = [int(c) for c in spec.obj_link_idx]
    if any(rank[k] > rank[k + 1] for k in range(len(rank) - 1)) or sorted(obj) != obj:
        return None                                # the generator needs walk-ordered columns / increasing collision columns
    sl = np.asarray(spec.self_link_idx, np.int32)
    pairs = [(int(sl[a]), int(sl[b])) for a, b in np.asarray(spec.self_pairs, np.int32).reshape(-1, 2)]
    return codegen.PointsTemplate(point_link=pl, point_offset=po, obj_cols=obj, self_pairs=pairs, ee_link=int(spec.ee_link),
                                  ee2_link=int(spec.ee2_link))


def _points_template_hash(pt: codegen.PointsTemplate) -> str:
    h = hashlib.sha1()
    h.update(np.asarray(pt.obj_cols, np.int32).tobytes())
    h.update(np.asarray(pt.self_pairs, np.int32).reshape(-1).tobytes())
    h.update(np.asarray([pt.ee_link, pt.ee2_link], np.int32).tobytes())
    return h.hexdigest()[:8]


def _points_serves(unit: codegen.PointsTemplate, want: codegen.PointsTemplate) -> bool:
    if codegen.points_hash(unit.point_link, unit.point_offset) != codegen.points_hash(want.point_link, want.point_offset):
        return False
    if list(unit.obj_cols) != list(want.obj_cols) or [tuple(p) for p in unit.self_pairs] != [tuple(p) for p in want.self_pairs]:
        return False
    return want.ee_link < 0 or (unit.ee_link == want.ee_link and unit.ee2_link == want.ee2_link)


_loaded_point_templates: Dict[str, Tuple[int, codegen.PointsTemplate]] = {}


def has_matching_points_unit(kin: KinModel, point_link, point_offset, spec) -> bool:
    want = _points_template_of(kin, point_link, point_offset, spec)
    if want is None:
        return False
    h = codegen.model_hash(kin)
    for mh, pt in _loaded_point_templates.values():
        if mh == h and _points_serves(pt, want):
            return True
    from .kinematics import URDF_DIR
    for ident, (urdf, fn) in codegen.SPEC_POINT_ROBOTS.items():
        k2 = KinModel.from_urdf(str(URDF_DIR / urdf))
        if codegen.model_hash(k2) == h and _points_serves(fn(k2), want):
            return True
    return False


def specialize_points(kin: KinModel, point_link, point_offset, spec, verbose: bool = False) -> Optional[str]:
    """Generated fused kernel for a robot whose collision columns are attached points (RobotPanda with another sphere
    table, another grasped object ...).  Returns None when the column layout is not one the generator handles."""
    pt = _points_template_of(kin, point_link, point_offset, spec)
    if pt is None:
        return None
    ident = (f"jitp_{codegen.model_hash(kin):016x}_{codegen.points_hash(pt.point_link, pt.point_offset):016x}_"
             f"{_points_template_hash(pt)}")
    if ident in _loaded:
        return ident
    if hipcc_available():
        so, stamp = JIT_DIR / f"spec_{ident}.so", JIT_DIR / f"spec_{ident}.stamp"
        if not (so.exists() and stamp.exists() and stamp.read_text() == _generator_stamp()):
            so = _compile_unit(codegen.generate_points_rollout_source(kin, pt, ident), ident, verbose)
        _loaded[ident] = _load_unit(so, ident)
    else:           # no compiler driver on this box: the unit's device half through hipRTC (round 5: attached-point units too)
        _loaded[ident] = _load_points_unit_rtc(kin, pt, ident)
    _loaded_point_templates[ident] = (codegen.model_hash(kin), pt)
    return ident
