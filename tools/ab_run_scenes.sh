# same-box A/B of the bench's scene variants (spheres c2 / c3, shelf, maze, grid; 4 alternations): every directory under _ab/ plus the working tree
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for side in $(ls $R/_ab) tree; do
    D=$R/_ab/$side; [ $side = tree ] && D=$R
    cd $D
    echo "== $side (rep $rep)"
    for a in "c2:--steps 2000 --warmup 200" "c3:--steps 2000 --warmup 200 --config c3" "shelf:--steps 1000 --warmup 100 --scene shelf" "maze:--steps 1000 --warmup 100 --scene maze" "grid:--steps 1000 --warmup 100 --scene grid"; do
      python bench.py ${a#*:} --cpu-seconds 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${a%%:*} us/step', round(d['ms_per_step'] * 1e3, 3))"
    done
  done
done
