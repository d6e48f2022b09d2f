# same-box A/B of two builds of the tree: _ab/base (a built `git archive` of the commit to compare with) against the working tree.
# usage (inside one gpurun call): bash tools/ab_run.sh > gpurun_out/ab.txt
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for side in base new; do
    D=$R; [ $side = base ] && D=$R/_ab/base
    cd $D
    echo "== $side (rep $rep)"
    python bench.py --steps 2000 --warmup 200 --cpu-seconds 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 us/step', round(d['ms_per_step'] * 1e3, 3))"
    python bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --config c3 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 us/step', round(d['ms_per_step'] * 1e3, 3))"
    python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene shelf 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shelf us/step', round(d['ms_per_step'] * 1e3, 3))"
    python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene grid 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid us/step', round(d['ms_per_step'] * 1e3, 3))"
    python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
s = sys.stdin.read(); d = json.loads(s[s.find('{'):])
for k, v in d.items():
    for o in v['ops']: print(k, o['op'][:50], o['us'])"
    python tools/bench_points.py 2>/dev/null | grep rollout
  done
done
