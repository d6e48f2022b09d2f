# Round 5 experiment: floats per chunk of the attached-point rollout's position staging (TRK_EXP_CHUNK_FLOATS at generation time).
# 36 (12 columns, 144 B per sample and chunk -- the committed value) against 32 / 64 (whole 32-byte sectors when the row is a multiple of
# 32 bytes: the 45-sphere model's 672-byte rows) and 72; N:1 = the widest vector that divides the chunk even where the row length leaves
# it only 4- / 8-byte aligned (TRK_EXP_CHUNK_WIDE).  One tree per value, built on the box.   Output: gpurun_out/r05chunk/ab.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05chunk; mkdir -p $O
{
for NW in ${1:-36:0 64:0 64:1}; do
  N=$(echo $NW | cut -d: -f1); WIDE=$(echo $NW | cut -d: -f2); BWD=$(echo $NW | cut -d: -f3); BWD=${BWD:-36}
  export TRK_EXP_CHUNK_WIDE=$WIDE TRK_EXP_BWD_CHUNK_FLOATS=$BWD
  B=/tmp/tree_${N}_${WIDE}_$BWD
  rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out $B/torch_robotics_amd/csrc/jit
  ( cd $B && TRK_EXP_CHUNK_FLOATS=$N make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/make_$N.log 2>&1 ) || { echo "BUILD $N FAILED"; tail -5 /tmp/make_$N.log; continue; }
  echo "chunk floats = $N, wide stores at 4- / 8-byte alignment = $WIDE, reverse mode chunk floats = $BWD"
  ( cd $B && TRK_EXP_CHUNK_FLOATS=$N python tools/bench_points.py 2>/dev/null )
done
} 2>&1 | tee $O/ab.txt
