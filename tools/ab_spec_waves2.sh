# Round 5 experiment: wavefronts per workgroup of the generated kernels (SPEC_WAVES) for the TWO-wavefront-per-SIMD kernels (config 5,
# attached points, config 4): smaller workgroups finish staggered -- does the hole between the two generations shrink?
# Two trees on one box, alternated: A = 4 wavefronts per workgroup (committed), B = 2.  Output: gpurun_out/r05w2/ab_spec_waves.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05w2; mkdir -p $O
W=${1:-2}
B=/tmp/treeB
rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out $B/torch_robotics_amd/csrc/jit
sed -i "s/^#define SPEC_WAVES 4/#define SPEC_WAVES $W/" $B/torch_robotics_amd/csrc/trk_spec_common.h
( cd $B && make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/makeB.log 2>&1 ) || { echo "BUILD B FAILED"; tail -5 /tmp/makeB.log; }
b() { ( cd $1 && python bench.py --cpu-seconds 0 --no-out-of-cache "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('   %-40s step %8.3f us  kernel %8.3f us' % (' '.join(sys.argv[1:]), d['ms_per_step']*1e3, d['roofline']['launch_us']))" "${@:2}" ); }
{
for rep in 1 2 3; do
  for t in A B; do
    if [ $t = A ]; then D=$R; else D=$B; fi
    echo "tree $t: SPEC_WAVES = $([ $t = A ] && echo 4 || echo $W)  (rep $rep)"
    b $D --config c5 --steps 1000 --warmup 100
    b $D --config c4 --steps 500 --warmup 50
    b $D --steps 2000 --warmup 200
    ( cd $D && python tools/bench_points.py 2>/dev/null | grep "fused rollout" )
  done
done
} 2>&1 | tee $O/ab_spec_waves_$W.txt
