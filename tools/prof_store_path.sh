# (The TCP_* / TCC_* _sum counters did not finish within minutes on this pool: SQ passes only.  PMC serialises the dispatches, ~0.15 s each: 2 configurations x 23 launches per pass.)
# Where do the output stores of a generated rollout kernel wait?  SQ / TCP / TCC counters of tools/ablate_robot.py <robot> (its
# configurations with and without positions are different kernel instantiations), separate --pmc passes.  -> gpurun_out/storepath/
R=$GRAFT_REPO_ROOT
ROBOT=${1:-ur10_allegro}
O=$R/gpurun_out/storepath_$ROBOT
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR" \
         "SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  TRK_ABLATE_LAUNCHES=20 timeout 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o p -- python3 $R/tools/ablate_robot.py $ROBOT two > /dev/null 2>> $O/err.txt
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:44s} {sum(v)/len(v):16.1f}   ({len(v)} dispatches)")
PY
