# SQ / SQC counters of the two-wavefront-per-SIMD kernels (config 5, config 4, grasped box): what does a wavefront that issues one
# instruction per ~10 cycles wait for?  Instruction fetch is the suspect not yet measured for the BIG kernels (10 000 static instructions
# ~ 60 KB of code against a 64 KB instruction cache shared by two CUs).  Separate --pmc passes, --kernel-trace only.
#   (gpurun) bash tools/pmc_sq_big.sh       -> gpurun_out/r05sq/summary.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05sq
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -o "SQC_[A-Z0-9_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $O/counters_available.txt | sort -u > $O/counter_names.txt
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_WAVES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ"
 "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT"
)
declare -A CMD
CMD[c5]="$R/bench.py --config c5 --steps 30 --warmup 5 --cpu-seconds 0 --no-out-of-cache"
CMD[c4]="$R/bench.py --config c4 --steps 30 --warmup 5 --cpu-seconds 0 --no-out-of-cache"
CMD[c2]="$R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-out-of-cache"
CMD[points]="$R/tools/bench_points.py"
for w in c5 c4 c2 points; do
  i=0
  for P in "${PASSES[@]}"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/${w}_p$i -o p -- python3 ${CMD[$w]} > /dev/null 2>> $O/err_$w.txt
  done
done
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections, os
O = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    w = os.path.relpath(f, O).split("_p")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_rollout" not in k:
            continue
        short = k.split("::")[-1].split("(")[0][:40] if w == "points" else "k_rollout*"
        if w == "points":
            short = k.split("::")[0].replace("void ", "")[-28:] + "::" + short
        res[(w, short)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for s in res.values() for c in s})
keys = sorted(res)
print("per wavefront (counter / SQ_WAVES of the same pass where present)")
print("%-30s" % "counter", *["%24s" % ("%s %s" % k)[:24] for k in keys])
for c in names:
    row = []
    for k in keys:
        v = res[k][c]
        wv = res[k]["SQ_WAVES"]
        w = (sum(wv) / len(wv)) if wv else 1.0
        row.append("%24.1f" % ((sum(v) / max(1, len(v))) / w) if v else "%24s" % "-")
    print("%-30s" % c, *row)
PY
