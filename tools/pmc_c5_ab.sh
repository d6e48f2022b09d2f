# Round 6: SQ counters + kernel trace of config 5's fused launch with one ROBOT per lane (k_rollout_gpt, TRK_GP_ARM_LANES=0) and one ARM
# per lane (k_rollout_gpa, =1).  Separate --pmc passes, --kernel-trace only.
#   (gpurun) bash tools/pmc_c5_ab.sh       -> gpurun_out/r06c5/summary.txt, gpurun_out/r06c5/stats_{robot,arm}/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c5
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
 "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS"
)
CMD="$R/bench.py --config c5 --steps 30 --warmup 5 --cpu-seconds 0 --no-out-of-cache"
for w in robot arm; do
  if [ $w = robot ]; then export TRK_GP_ARM_LANES=0; else export TRK_GP_ARM_LANES=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -o s -- python3 $CMD > $O/bench_$w.json 2>> $O/err_$w.txt
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${w}_p$i -o p -- python3 $CMD > /dev/null 2>> $O/err_$w.txt
  done
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*_p*/**/*counter_collection.csv", recursive=True):
    w = os.path.relpath(f, O).split("_p")[0]
    for r in csv.DictReader(open(f)):
        if "k_rollout_gp" not in r["Kernel_Name"]:
            continue
        res[w][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for s in res.values() for c in s})
print("config 5, fused launch: per WAVEFRONT (counter / SQ_WAVES) and per LAUNCH")
print("%-26s %14s %14s %16s %16s" % ("counter", "robot / wave", "arm / wave", "robot / launch", "arm / launch"))
for c in names:
    row = []
    for w in ("robot", "arm"):
        v = res[w][c]; wv = res[w]["SQ_WAVES"]
        nw = sum(wv) / max(1, len(wv))
        row.append((sum(v) / max(1, len(v))) if v else float("nan"))
        row.append(nw)
    print("%-26s %14.1f %14.1f %16.0f %16.0f" % (c, row[0] / max(1.0, row[1]), row[2] / max(1.0, row[3]), row[0], row[2]))
for w in ("robot", "arm"):
    for f in glob.glob(O + f"/stats_{w}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout_gp" in r["Name"]:
                print(w, r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
