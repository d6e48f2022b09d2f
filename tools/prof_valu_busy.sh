# VALU pipe occupancy of the hot kernels from the SQ counters (separate --pmc passes, --kernel-trace only), next to the same counters
# on the issue-rate probe (tools/microbench/valu_table) as calibration.  Output: gpurun_out/valu/<target>_<pass>/...counter_collection.csv;
# tools/valu_busy_collect.py folds them into profiles/r03_valu_busy.json.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/valu
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
P4="SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"
P5="GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/bench_p$i -o p -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 > /dev/null 2>> $O/err.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/shelf_p$i -o p -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --scene shelf > /dev/null 2>> $O/err.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/points_p$i -o p -- python3 $R/tools/bench_points.py > /dev/null 2>> $O/err.txt
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/configs_p$i -o p -- python3 $R/tools/bench_configs.py > /dev/null 2>> $O/err.txt
  VALU_ONLY=",0,7,11,19,44,47,60,62,64,65," VALU_W=4 timeout 120 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/calib_p$i -o p -- $R/tools/microbench/valu_table > $O/calib_p$i.txt 2>> $O/err.txt
done
find $O -name "*counter_collection.csv" | wc -l
