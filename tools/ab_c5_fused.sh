# same-box A/B of compile-time variants of the dual-Panda unit on the fused config-5 launch:
#   bash tools/ab_c5_fused.sh "base:" "aligned:-DTRK_EXP_SEG_ALIGN"
cd $GRAFT_REPO_ROOT/torch_robotics_amd/csrc
U=dual_panda
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off -Xarch_device -fno-slp-vectorize -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee -mllvm -amdgpu-sched-strategy=max-ilp"
cp generated/spec_$U.o /tmp/spec_$U.o.orig
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  $CXX $flags -c generated/spec_$U.hip -o generated/spec_$U.o 2>/tmp/ab_err.txt || { echo "$name: BUILD FAILED"; tail -5 /tmp/ab_err.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
  echo "== $name"; (cd $GRAFT_REPO_ROOT && python tools/exp_c5_fused.py 2>/dev/null | head -2)
done
cp /tmp/spec_$U.o.orig generated/spec_$U.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
