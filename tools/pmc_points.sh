cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_points/$tag -o p -- python3 $R/tools/pmc_points.py $1 > /dev/null 2>> $R/gpurun_out/pmc_points.err
done
find $R/gpurun_out/pmc_points -name "*counter_collection.csv" | head
