# Same-box A/B of LLVM scheduling strategies on one generated unit: UNIT=dual_panda bash tools/ab_sched_robot.sh
cd $GRAFT_REPO_ROOT/torch_robotics_amd/csrc
U=${UNIT:-dual_panda}
BASE="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee"
cp generated/spec_$U.o /tmp/spec_$U.o.orig
for v in "max-ilp:-mllvm -amdgpu-sched-strategy=max-ilp" "default:" "max-memory-clause:-mllvm -amdgpu-sched-strategy=max-memory-clause" "iterative-ilp:-mllvm -amdgpu-sched-strategy=iterative-ilp" "iterative-minreg:-mllvm -amdgpu-sched-strategy=iterative-minreg"; do
  name=${v%%:*}; flags=${v#*:}
  $BASE $flags -c generated/spec_$U.hip -o generated/spec_$U.o 2>/tmp/ab_err.txt || { echo "$name: BUILD FAILED"; tail -3 /tmp/ab_err.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
  echo "== $name"; (cd $GRAFT_REPO_ROOT && python tools/ablate_robot.py $U 2>/dev/null | sed -n 2,3p)
done
cp /tmp/spec_$U.o.orig generated/spec_$U.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
