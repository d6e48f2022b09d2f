"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/trk.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

from helpers import model
from torch_robotics_amd import _abi, _lib

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def trk():
    if not _lib.LIB_PATH.exists():
        _lib.build()
    return _lib.lib()


def test_header_symbols_all_exported(trk):
    header = (ROOT / "include" / "trk.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t|void|const char\*)\s+(trk_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed from trk.h"
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(trk, name), f"libtrk.so does not export {name}"
    assert trk.trk_abi_version() == _abi.TRK_ABI_VERSION


def test_struct_layouts_match_header():
    # sizes the C compiler produced for the same structs (sizeof is stable across gcc/hipcc on x86-64)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "trk.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",sizeof(TrkKinModelDesc),' \
          'sizeof(TrkPrimitive),sizeof(TrkObject),sizeof(TrkGridDesc),sizeof(TrkCostModelDesc),sizeof(TrkRolloutWeights));}'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(Path(d) / "s.c"), "-o", str(Path(d) / "s")], check=True)
        out = subprocess.run([str(Path(d) / "s")], capture_output=True, text=True, check=True).stdout.split()
    sizes = [C.sizeof(t) for t in (_abi.KinModelDesc, _abi.Primitive, _abi.Object, _abi.GridDesc,
                                   _abi.CostModelDesc, _abi.RolloutWeights)]
    assert [int(v) for v in out] == sizes


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly(trk):
    from torch_robotics_amd import ops
    with pytest.raises(_lib.TrkError, match="no HIP device"):
        ops.ModelHandle(model("panda_arm_no_gripper"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops._dev_f32(torch.zeros(2, 7), "q")


def test_argument_validation_without_gpu(trk):
    # invalid descriptors are rejected before any device work
    m = model("panda_arm_no_gripper")
    desc, keep = _abi.kin_desc(m)
    h = C.c_void_p()
    desc.abi_version = 99
    assert trk.trk_model_create(C.byref(desc), C.byref(h)) == _abi.TRK_ERR_INVALID_ARG
    desc.abi_version = _abi.TRK_ABI_VERSION
    desc.n_links = 1000
    assert trk.trk_model_create(C.byref(desc), C.byref(h)) == _abi.TRK_ERR_UNSUPPORTED
    assert b"n_links" in trk.trk_last_error()
    assert trk.trk_model_create(None, C.byref(h)) == _abi.TRK_ERR_INVALID_ARG


def test_every_entry_point_is_documented_for_the_reference_side():
    """INTEGRATION.md maps each function of include/trk.h to the reference call site it replaces (or names it as a helper)."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    declared = sorted(set(re.findall(r"\b(trk_[a-z0-9_]+)\s*\(", (root / "include" / "trk.h").read_text())))
    text = (root / "INTEGRATION.md").read_text()
    assert [n for n in declared if n not in text] == []
