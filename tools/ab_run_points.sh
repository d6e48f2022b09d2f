# same-box A/B of the attached-point benches only (5 alternations): every directory under _ab/ plus the working tree
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  for side in $(ls $R/_ab) tree; do
    D=$R/_ab/$side; [ $side = tree ] && D=$R
    cd $D
    echo "== $side (rep $rep)"
    python tools/bench_points.py 2>/dev/null | grep rollout
  done
done
