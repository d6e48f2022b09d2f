# same-box A/B of tools/bench_configs.py only (5 alternations): every directory under _ab/ plus the working tree
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  for side in $(ls $R/_ab) tree; do
    D=$R/_ab/$side; [ $side = tree ] && D=$R
    cd $D
    echo "== $side (rep $rep)"
    python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
s = sys.stdin.read(); d = json.loads(s[s.find('{'):])
for k, v in d.items():
    for o in v['ops']: print(k, o['op'][:50], o['us'])"
  done
done
