# re-runs of the round-4 set that needed fixing: the grid scene with smooth trajectories WITHOUT the out-of-cache side measurement (its
# 8 x launches were averaged into the per-launch counters), and the N > 1 lines after the cadence change
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc/gridsmooth_* $O/prof_grid_smooth
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/gridsmooth_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --scene grid --q smooth --no-out-of-cache > /dev/null 2>> $O/pmc.err
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_grid_smooth -o r04 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene grid --q smooth --no-out-of-cache > /dev/null 2>> $O/prof.err
cd $R
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --steps 2000 --warmup 200 > $O/bench_2rank_gloo_one_gpu.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --config c5 --steps 1000 --warmup 100 > $O/bench_c5_2rank_gloo_one_gpu.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_rccl_one_rank_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --steps 2000 --warmup 200 --cpu-seconds 0 > $O/bench_rccl_one_rank.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --config c5 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c5_rccl_one_rank_driver_settings.json 2>> $O/bench.err
