# Same-box A/B of two builds of libtrk.so (same C ABI): usage  bash tools/ab_libs.sh <base.so> [bench args...]
# Alternates base / current three times; prints the launch time of each run.
cd $GRAFT_REPO_ROOT
base=$1; shift
for i in 1 2 3; do
  for lib in "$base" ""; do
    t=$(TRK_LIBTRK=$lib python bench.py --cpu-seconds 0 --steps 3000 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['roofline']['launch_us'])")
    echo "${lib:-current}: $t us"
  done
done
