# SQ counters of one bench scene for every tree under _ab/ (separate --pmc passes, --kernel-trace only): where do the cycles of two
# variants of a kernel differ?   usage (gpurun): bash tools/pmc_ab.sh "--scene shelf"      -> gpurun_out/pmcab/<side>_p<i>/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcab
rm -rf $O; mkdir -p $O
ARGS=${1:-"--scene shelf"}
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
P4="SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_LDS"
P5="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P6="SQ_INSTS_WAVE32_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM"
for side in $(ls $R/_ab); do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${side}_p$i -o p -- python3 $R/_ab/$side/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-out-of-cache $ARGS > /dev/null 2>> $O/err.txt
  done
done
python3 - $O <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    side = os.path.relpath(f, O).split("_p")[0]
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]:
            res[side][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for s in res.values() for c in s})
print("%-28s" % "counter (per wave)", *["%12s" % s for s in sorted(res)])
for c in names:
    row = []
    for s in sorted(res):
        w = sum(res[s]["SQ_WAVES"]) / max(1, len(res[s]["SQ_WAVES"])) or 1
        v = res[s][c]
        row.append("%12.1f" % (sum(v) / max(1, len(v)) / w))
    print("%-28s" % c, *row)
PY
