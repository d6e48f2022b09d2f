# Same-box A/B of the attached-point units under compile-time switches: for each "name:flags" argument rebuild the three point units
# (generated/spec_panda_{spheres,grasp,spheres_grasp}.o) with the extra flags, relink libtrk.so and run tools/bench_points.py.
# usage (on the GPU box): bash tools/ab_defines_points.sh "base:" "exp1:-DTRK_EXP_FOO=1" ...
cd $GRAFT_REPO_ROOT/torch_robotics_amd/csrc
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off -Xarch_device -fno-slp-vectorize -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee -mllvm -amdgpu-sched-strategy=max-ilp"
U="spec_panda_spheres spec_panda_grasp spec_panda_spheres_grasp"
for u in $U; do cp generated/$u.o /tmp/$u.o.orig; done
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ok=1
  for u in $U; do $CXX $flags -c generated/$u.hip -o generated/$u.o 2>/tmp/ab_err_$u.txt & done; wait
  for u in $U; do [ -s generated/$u.o ] || ok=0; done
  [ $ok = 1 ] || { echo "$name: BUILD FAILED"; tail -5 /tmp/ab_err_*.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o trk_exchange.o generated/*.o
  echo "== $name ($flags)"
  for i in 1 2; do (cd $GRAFT_REPO_ROOT && python tools/bench_points.py 2>/dev/null | grep "fused rollout"); done
done
for u in $U; do cp /tmp/$u.o.orig generated/$u.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o trk_exchange.o generated/*.o
