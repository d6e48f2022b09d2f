# same-box A/B of the headline bench lines only (5 alternations, c2 + c3): every directory under _ab/
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for side in $(ls $R/_ab); do
    cd $R/_ab/$side
    echo "== $side (rep $rep)"
    for a in "c2:--steps 2000 --warmup 200" "c3:--steps 2000 --warmup 200 --config c3"; do
      python bench.py ${a#*:} --cpu-seconds 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${a%%:*} us/step', round(d['ms_per_step'] * 1e3, 3))"
    done
  done
done
