#!/bin/bash
# VGPR / SGPR / spill / LDS figures of the kernels in a generated unit's object file (the code object's metadata notes).
#   tools/kernel_resources.sh torch_robotics_amd/csrc/generated/spec_dual_panda.o [kernel-name-substring]
set -e
OBJ=$(readlink -f "$1"); PAT=${2:-k_rollout}
T=$(mktemp -d)
cp "$OBJ" $T/u.o
( cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading u.o > /dev/null 2>&1 )
CO=$(ls $T/u.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | awk -v pat="$PAT" '
  /\.group_segment_fixed_size:/ { lds=$2 }
  /\.name:/ { name=$2 }
  /\.sgpr_count:/ { sg=$2 }
  /\.sgpr_spill_count:/ { ss=$2 }
  /\.vgpr_count:/ { vg=$2 }
  /\.agpr_count:/ { ag=$2 }
  /\.vgpr_spill_count:/ { vs=$2; if (name ~ pat) printf "vgpr %3d  spill %3d  sgpr %3d  sspill %3d  lds %6d  %s\n", vg, vs, sg, ss, lds, name }' | c++filt | sed 's/(SpecArgs)//' 
rm -rf $T
