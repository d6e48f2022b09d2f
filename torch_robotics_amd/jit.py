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
        _loaded[ident] = _load_unit(build_unit(kin, tmpl, verbose, pipeline), ident)
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
    so, stamp = JIT_DIR / f"spec_{ident}.so", JIT_DIR / f"spec_{ident}.stamp"
    if not (so.exists() and stamp.exists() and stamp.read_text() == _generator_stamp()):
        so = _compile_unit(codegen.generate_points_rollout_source(kin, pt, ident), ident, verbose)
    _loaded[ident] = _load_unit(so, ident)
    _loaded_point_templates[ident] = (codegen.model_hash(kin), pt)
    return ident
