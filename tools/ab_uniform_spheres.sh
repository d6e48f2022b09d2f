# Round 5 experiment: the constant-link evaluation (scene_min_sdf_uniform_point) also in the SPHERES-ONLY instantiation -- the headline
# kernel.  Two trees on one box, alternated: A = the tree as committed, B = generated with TRK_EXP_UNIFORM_SPHERES=1.
# Output: gpurun_out/r05v/ab_uniform_spheres.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05v; mkdir -p $O
B=/tmp/treeB
rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out
( cd $B && TRK_EXP_UNIFORM_SPHERES=1 python -c "from torch_robotics_amd import codegen; codegen.generate_all('torch_robotics_amd/csrc/generated')" && TRK_EXP_UNIFORM_SPHERES=1 make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/makeB.log 2>&1 ) || { echo "BUILD B FAILED"; tail -5 /tmp/makeB.log; }
grep -c "if (!A.C.has_grid) {" $B/torch_robotics_amd/csrc/generated/spec_panda.hip
b() { ( cd $1 && python bench.py --cpu-seconds 0 --no-out-of-cache "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('   %-34s %8.3f us  frac %.3f' % (' '.join(sys.argv[1:]), d['roofline']['launch_us'], d['roofline']['frac']))" "${@:2}" ); }
{
for rep in 1 2 3 4; do
  for t in A B; do
    if [ $t = A ]; then D=$R; else D=$B; fi
    echo "tree $t (rep $rep)"
    b $D --steps 3000 --warmup 300
    b $D --config c3 --steps 3000 --warmup 300
  done
done
} 2>&1 | tee $O/ab_uniform_spheres.txt
