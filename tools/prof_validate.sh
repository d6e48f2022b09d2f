# kernel durations of one trajectory validation (rocprofv3 --kernel-trace --stats of tools/bench_task_api.py)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_validate -o v -- python3 $R/tools/bench_task_api.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_validate/**/v_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
