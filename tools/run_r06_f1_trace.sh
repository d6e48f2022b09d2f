# Round 6: kernel-level trace of the trajectory-validation path (F1): k_coll_bi (boolean exit, plain and via-point mode), k_traj_flags,
# k_traj_partition, k_traj_gather.   (gpurun) bash tools/run_r06_f1_trace.sh  -> gpurun_out/r06f1/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06f1
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_task_api.py > $O/bench_task_api.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/bench_task_api.py > $O/bench_task_api_under_rocprof.txt 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(O + "/kernel_stats_f1.csv", "w") as out:
        w = csv.writer(out); w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalDurationNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"]])
            if any(k in r["Name"] for k in ("k_coll", "k_traj", "k_interp")):
                print("%-90s calls %6s avg %9.2f us  min %8.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
grep -n "get_trajs\|rollout_collision\|traj_validate\|compute_collision(q)" $O/bench_task_api.txt
