#!/bin/bash
# Static instruction mix (VALU / SALU / LDS / VMEM, half-rate and transcendental VALU, s_waitcnt) of the kernels of one generated
# unit whose mangled name matches a pattern.   usage: tools/isa_counts.sh generated/spec_dual_panda.hip 'rollout.*DF16_Lb1'
set -e
cd "$(dirname "$0")/../torch_robotics_amd/csrc"
SRC=$1; PAT=${2:-.}
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed -ffp-contract=off \
    -Xarch_device -fno-slp-vectorize -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee -mllvm -amdgpu-sched-strategy=max-ilp \
    --offload-device-only -c "$SRC" -o "$TMP/dev.o"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/dev.o" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/dev.co"
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$TMP/dev.co" > "$TMP/dev.s"
python3 - "$TMP/dev.s" "$PAT" <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for f in re.split(r'\n(?=[0-9a-f]+ <[^>]+>:)', txt):
    m = re.match(r'[0-9a-f]+ <([^>]+)>:', f)
    if not m or not pat.search(m.group(1)):
        continue
    ins = [l.split()[0] for l in f.splitlines()[1:] if re.match(r'\s+[a-z_0-9]+', l)]
    c = collections.Counter('valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'lds' if i.startswith('ds_') else
                            'vmem' if re.match(r'(global|flat|buffer)_', i) else 'other' for i in ins)
    half = sum(1 for i in ins if re.match(r'v_(min|max|med3|cmp|cndmask|and_or|bfi|lshl|bfe|add3|cvt|pk_|mad_u|mul_lo|readfirst|perm|lshr|ashr)', i))
    trans = sum(1 for i in ins if re.match(r'v_(rsq|rcp|sqrt|sin|cos|exp|log)', i))
    print(f"{m.group(1)[:70]:70s} {dict(c)} half-rate {half} trans {trans} waitcnt {sum(1 for i in ins if i == 's_waitcnt')}")
PY
rm -rf "$TMP"
