R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcpts
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o p -- python3 $R/tools/bench_points.py > /dev/null 2>> $O/err.txt
done
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spec_" in r["Kernel_Name"]:
            res[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(res):
    w = sum(res[k]["SQ_WAVES"]) / max(1, len(res[k]["SQ_WAVES"])) or 1
    print(k, "waves", int(w), {c: round(sum(v) / len(v) / w, 1) for c, v in sorted(res[k].items()) if c != "SQ_WAVES"})
PY
