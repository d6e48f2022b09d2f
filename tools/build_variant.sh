#!/bin/bash
# Build a tick-schedule variant of libtrk.so next to the in-tree one (for tools/ab_many.sh):
#   bash tools/build_variant.sh <first burst chunks> <objective tick slots> <out.so>
# Regenerates csrc/generated with the experiment knobs, compiles with the matching -DTRK_OBJ_TICK_SLOTS, copies the library,
# then restores the default sources and library.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
gen() { TRK_EXP_FIRST_BURST=$1 TRK_EXP_OBJ_SLOTS=$2 python - <<PY
from pathlib import Path
from torch_robotics_amd import codegen
codegen.generate_all(Path("torch_robotics_amd/csrc/generated"))
PY
}
gen $1 $2
( cd torch_robotics_amd/csrc && touch generated/*.hip && make -j6 GENFLAGS="-mllvm -amdgpu-sched-strategy=max-ilp -DTRK_OBJ_TICK_SLOTS=$2" 2>&1 | grep -E "error" | head -3; cp libtrk.so $R/$3 )
gen 0 5
( cd torch_robotics_amd/csrc && touch generated/*.hip && make -j6 2>&1 | grep -E "error" | head -3 ) || true
