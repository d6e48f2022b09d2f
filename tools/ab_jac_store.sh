# Round 5 experiment: cache policy of the geometric Jacobian's read-out stores (contiguous 16-byte vectors) inside config 4's step
# (fused rollout 141 MB + Jacobian 169 MB per step: together beyond the 256 MB Infinity Cache).  Output: gpurun_out/r05j/ab_jac_store.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
H=torch_robotics_amd/csrc/trk_device.h
cp $H /tmp/trk_device.orig
b() { python bench.py --cpu-seconds 0 --config c4 --steps 500 --warmup 50 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('   c4 step %8.2f us   rollout alone %7.2f us' % (d['ms_per_step']*1e3, d['roofline']['launch_us']))"; python tools/bench_jacobian_kernel.py ur10_allegro 2>/dev/null | tail -2; }
{
for rep in 1 2; do
for mod in "sc1" "nt" ""; do
  sed -e "s/off sc1\" :: \"v\"(lo + k0)/off $mod\" :: \"v\"(lo + k0)/; s/off sc1\" :: \"v\"(ao + k0)/off $mod\" :: \"v\"(ao + k0)/" /tmp/trk_device.orig > $H
  make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/make.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/make.log; }
  echo "jacobian read-out stores: [$mod]  (rep $rep)"
  b
done
done
} 2>&1 | tee $O/ab_jac_store.txt
cp /tmp/trk_device.orig $H
