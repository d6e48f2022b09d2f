# Round 6: kernel-level trace of the IK paths: the Adam loop (k_ik: trk_ik_steps) and the Gauss-Newton loop (k_ikgn: trk_ik_gn_steps,
# and its two-launch form k_jac + k_jtj).   (gpurun) bash tools/run_r06_ik_trace.sh  -> gpurun_out/r06ik/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06ik
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_ik.py > $O/bench_ik.txt 2>/dev/null
python3 $R/tools/bench_ik_gn.py > $O/bench_ik_gn.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_adam -o t -- python3 $R/tools/bench_ik.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gn -o t -- python3 $R/tools/bench_ik_gn.py > /dev/null 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
with open(O + "/kernel_stats_ik.csv", "w") as out:
    w = csv.writer(out); w.writerow(["Run", "Name", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalDurationNs"])
    for run in ("adam", "gn"):
        for f in glob.glob(O + f"/trace_{run}/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if any(k in r["Name"] for k in ("k_ik", "k_jac", "k_jtj", "k_fk")):
                    w.writerow([run, r["Name"], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"]])
                    print("%-5s %-80s calls %6s avg %9.2f us  min %8.2f" % (run, r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
cat $O/bench_ik.txt | tail -12; cat $O/bench_ik_gn.txt | tail -8
