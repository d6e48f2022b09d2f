# Same-box A/B of a bench.py configuration under GENERATION-time knobs (environment variables read by codegen.py): one tree per setting, built
# on the box, alternated.   usage (gpurun): bash tools/ab_env_bench.sh "--config c5 --steps 1000 --warmup 100" "TRK_EXP_GP_PRIOR=lds" "TRK_EXP_GP_PRIOR=dpp"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05envb; mkdir -p $O
ARGS=$1; shift
{
k=0
for KV in "$@"; do
  k=$((k+1)); B=/tmp/tree_envb_$k
  rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out $B/torch_robotics_amd/csrc/jit
  ( cd $B && env $KV make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/make_envb_$k.log 2>&1 ) || { echo "BUILD $KV FAILED"; tail -5 /tmp/make_envb_$k.log; }
done
for rep in 1 2 3; do
  k=0
  for KV in "$@"; do
    k=$((k+1)); B=/tmp/tree_envb_$k
    ( cd $B && env $KV python bench.py --cpu-seconds 0 --no-out-of-cache $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('%-32s rep %s  step %8.3f us  kernel %8.3f us' % (sys.argv[1], sys.argv[2], d['ms_per_step']*1e3, d['roofline']['launch_us']))" "$KV" $rep )
  done
done
} 2>&1 | tee $O/ab.txt
