#!/bin/bash
# Register / LDS / spill figures of every kernel of one generated unit (or any .hip of csrc/), compiled device-only with the
# flags of csrc/Makefile.  No GPU needed.   usage: tools/kernel_resources.sh generated/spec_panda.hip [extra hipcc flags]
# columns: LDS bytes, SGPR spills, VGPRs, VGPR spills, scratch bytes, kernel
set -e
cd "$(dirname "$0")/../torch_robotics_amd/csrc"
SRC=$1; shift || true
TMP=$(mktemp -d)
GEN=""
case "$SRC" in generated/*|jit/*) GEN="-Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee -mllvm -amdgpu-sched-strategy=max-ilp";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -ffp-contract=off \
    -Xarch_device -fno-slp-vectorize -I. $GEN "$@" --offload-device-only -c "$SRC" -o "$TMP/dev.o"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/dev.o" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    print("%7s %4s %4s %4s %6s  %s" % (g("group_segment_fixed_size"), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"),
                                     g("private_segment_fixed_size"), g("name")))
' | c++filt
rm -rf "$TMP"
