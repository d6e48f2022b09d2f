# same-box ablation of the headline kernel by run-time switches (no rebuild): which part of the 11 us is what
cd $GRAFT_REPO_ROOT
for a in "" "--no-pos" "--weights 0,0,0,0" "--weights 0,0,0,0 --no-pos" "--weights 0,1,0,0" "--weights 0,0,0,1" "--weights 1,1,1,1" "--batch 8192" "--batch 2048"; do
  python bench.py --steps 2000 --warmup 200 --cpu-seconds 0 $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s %7.2f us  frac %.3f' % (sys.argv[1] or 'default', d['ms_per_step'] * 1e3, d['roofline']['frac']))" "$a"
done
