# Same-box A/B of the headline kernel under compile-time switches: for each "name:flags" argument rebuild ONLY
# generated/spec_panda.o with the extra flags, relink libtrk.so and run the default bench three times.
# usage (on the GPU box): bash tools/ab_defines.sh "base:" "exp1:-DTRK_EXP_FOO" ...
cd $GRAFT_REPO_ROOT/torch_robotics_amd/csrc
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -ffp-contract=off -Xarch_device -fno-slp-vectorize -I. -Xarch_device -fno-honor-nans -Xarch_device -mno-amdgpu-ieee"
ILP="-mllvm -amdgpu-sched-strategy=max-ilp"      # the Makefile's GENFLAGS; a variant whose flags contain NOILP is compiled without it
cp generated/spec_panda.o /tmp/spec_panda.o.orig
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  case "$flags" in *NOILP*) flags=${flags/NOILP/};; *) flags="$ILP $flags";; esac
  $CXX $flags -c generated/spec_panda.hip -o generated/spec_panda.o 2>/tmp/ab_err.txt || { echo "$name: BUILD FAILED"; tail -5 /tmp/ab_err.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
  printf "%-28s" "$name"
  for i in 1 2 3; do
    (cd $GRAFT_REPO_ROOT && python bench.py --cpu-seconds 0 --steps 3000 --no-out-of-cache $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' %6.2f' % d['roofline']['launch_us'], end='')")
  done
  echo
done
cp /tmp/spec_panda.o.orig generated/spec_panda.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libtrk.so trk_capi.o trk_kernels.o generated/*.o
