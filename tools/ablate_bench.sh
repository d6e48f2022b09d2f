# Run-time ablation of a bench configuration: weights zeroed one at a time, with / without the position output (bench.py --weights / --no-pos).
#   (gpurun) bash tools/ablate_bench.sh "--config c5"      -> gpurun_out/r05abl/<tag>.txt
R=$GRAFT_REPO_ROOT
ARGS=${1:-"--config c5"}
TAG=$(echo $ARGS | tr -d ' -')
O=$R/gpurun_out/r05abl; mkdir -p $O
cd $R
b() { python bench.py --cpu-seconds 0 --no-out-of-cache --steps 1000 --warmup 100 $ARGS "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('%8.2f us' % (d['roofline']['launch_us']), end='')"; }
{
echo "ablation of: bench.py $ARGS   (kernel time by events; weights = self, obj, ws, ee)"
for w in "default" "0,1,1,1" "1,0,1,1" "1,1,0,1" "1,1,1,0" "0,1,0,0" "0,0,0,1" "0,0,0,0"; do
  if [ "$w" = default ]; then WA=""; else WA="--weights $w"; fi
  printf "   weights %-10s  with positions " "$w"; b $WA; printf "   without "; b $WA --no-pos; echo
done
} 2>&1 | tee $O/$TAG.txt
