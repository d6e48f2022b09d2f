# like ab_defines.sh, for another generated unit: UNIT=ur10_allegro bash tools/ab_defines_robot.sh "base:" "x:-DFOO"
cd $GRAFT_REPO_ROOT/torch_robotics_amd/csrc
U=${UNIT:-ur10_allegro}
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee -mllvm -amdgpu-sched-strategy=max-ilp"
cp generated/spec_$U.o /tmp/spec_$U.o.orig
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  $CXX $flags -c generated/spec_$U.hip -o generated/spec_$U.o 2>/tmp/ab_err.txt || { echo "$name: BUILD FAILED"; tail -5 /tmp/ab_err.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
  echo "== $name"; (cd $GRAFT_REPO_ROOT && python tools/ablate_robot.py $U 2>/dev/null | head -4)
done
cp /tmp/spec_$U.o.orig generated/spec_$U.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
