# Same-box A/B of the attached-point units under GENERATION-time knobs (environment variables read by codegen.py): one tree per setting,
# built on the box, tools/bench_points.py in each.   usage (gpurun): bash tools/ab_env_points.sh "TRK_EXP_OBJ_GROUP=6" "TRK_EXP_OBJ_GROUP=8" ...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05env; mkdir -p $O
{
k=0
for KV in "$@"; do
  k=$((k+1)); B=/tmp/tree_env_$k
  rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out $B/torch_robotics_amd/csrc/jit
  ( cd $B && env $KV make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/make_env_$k.log 2>&1 ) || { echo "BUILD $KV FAILED"; tail -5 /tmp/make_env_$k.log; continue; }
  echo "== $KV"
  for i in 1 2; do ( cd $B && env $KV python tools/bench_points.py 2>/dev/null | grep "fused rollout" ); done
done
} 2>&1 | tee $O/ab.txt
