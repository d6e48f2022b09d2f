# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of one generated robot's rollout kernel.
# usage (on the GPU box): UNIT=ur10_allegro bash tools/robot_traffic.sh
R=$GRAFT_REPO_ROOT; U=${UNIT:-ur10_allegro}; O=$R/gpurun_out/traffic_$U; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -o p -- python3 $R/tools/ablate_robot.py $U first > /dev/null 2>> $O/err.txt
done
python3 $R/tools/pmc_summary.py $O k_rollout
