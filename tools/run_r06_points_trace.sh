# Round 6: the attached-point kernels under rocprofv3 with ONE row per run, so that the fused rollout and the positions-only launch of the
# same kernel are not averaged.   (gpurun) bash tools/run_r06_points_trace.sh -> gpurun_out/r06pts/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pts
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_points.py > $O/bench_points.txt 2>/dev/null
for k in plan positions backward; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$k -o p -- python3 $R/tools/bench_points.py --only $k > /dev/null 2>> $O/err.txt
  f=$(find $O/prof_$k -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_points_$k.csv
  echo "== $k"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "spec_panda" in r["Name"]:
        print("  %-80s calls %5s avg %8.2f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cat $O/bench_points.txt
