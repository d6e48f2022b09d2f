# same-box A/B of every bench scene (3 alternations): each directory under _ab/ that understands the flags (SIDES overrides the list)
R=$GRAFT_REPO_ROOT
SIDES=${SIDES:-$(ls $R/_ab)}
for rep in 1 2 3; do
  for side in $SIDES; do
    cd $R/_ab/$side
    echo "== $side (rep $rep)"
    for a in "c2:" "c3:--config c3" "grid_iid:--scene grid" "grid_smooth:--scene grid --q smooth" "shelf_iid:--scene shelf" "shelf_smooth:--scene shelf --q smooth" \
             "maze_iid:--scene maze" "maze_smooth:--scene maze --q smooth" "c5:--config c5"; do
      python bench.py --steps 1000 --warmup 100 ${a#*:} --cpu-seconds 0 --no-out-of-cache 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${a%%:*} us/step', round(d['ms_per_step'] * 1e3, 3))"
    done
  done
done
