# same-box A/B of builds of the tree: every directory under _ab/ (tools/ab_build.sh) plus, with AB_WITH_TREE=1, the working tree itself.
# usage (inside one gpurun call): bash tools/ab_run.sh > gpurun_out/ab.txt
R=$GRAFT_REPO_ROOT
SIDES=$(ls $R/_ab)
[ -n "$AB_WITH_TREE" ] && SIDES="$SIDES tree"
for rep in 1 2; do
  for side in $SIDES; do
    D=$R/_ab/$side; [ $side = tree ] && D=$R
    cd $D
    echo "== $side (rep $rep)"
    for a in "c2:--steps 2000 --warmup 200" "c3:--steps 2000 --warmup 200 --config c3" "shelf:--steps 1000 --warmup 100 --scene shelf" "grid:--steps 1000 --warmup 100 --scene grid"; do
      python bench.py ${a#*:} --cpu-seconds 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${a%%:*} us/step', round(d['ms_per_step'] * 1e3, 3))"
    done
    python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
s = sys.stdin.read(); d = json.loads(s[s.find('{'):])
for k, v in d.items():
    for o in v['ops']: print(k, o['op'][:50], o['us'])"
    python tools/bench_points.py 2>/dev/null | grep rollout
  done
done
