#!/bin/bash
# Static instruction counts per kernel of a generated unit's object file (straight-line kernels: static ~ dynamic per wavefront).
#   tools/isa_valu_count.sh torch_robotics_amd/csrc/generated/spec_panda_grasp.o [kernel-name-substring]
set -e
OBJ=$1; PAT=${2:-k_rollout}
T=$(mktemp -d)
cp "$OBJ" $T/u.o
( cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading u.o > /dev/null 2>&1 )
CO=$(ls $T/u.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$CO" | awk -v pat="$PAT" '
  /^[0-9a-f]+ <.*>:$/ { name=$2; next }
  name ~ pat { if ($1 ~ /^v_/) v[name]++; else if ($1 ~ /^s_/) s[name]++; else if ($1 ~ /^ds_/) d[name]++; else if ($1 ~ /^(global|buffer|flat|scratch)_/) g[name]++;
               if ($1 ~ /^v_(rsq|sqrt|rcp|sin|cos|exp|log)/) t[name]++ }
  END { for (n in v) printf "%-90s VALU %6d  SALU %5d  LDS %4d  VMEM %4d  transc %4d\n", n, v[n], s[n], d[n], g[n], t[n] }' | sort
rm -rf $T
