# Same-box comparison of several builds of libtrk.so: bash tools/ab_many.sh lib1.so lib2.so ...   ("" = the in-tree build)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in "$@"; do
    t=$(TRK_LIBTRK=$lib python bench.py --cpu-seconds 0 --steps 3000 $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['roofline']['launch_us'])")
    echo "${lib:-current}: $t us"
  done
done
