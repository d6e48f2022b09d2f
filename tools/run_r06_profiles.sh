# Round-6 measurement set (the round-5 set on the round-6 library; c4 = the one-launch form, its two-launch form next to it) (one gpurun call): bench lines of every BASELINE config (c2 spheres / grid / grid with smooth trajectories /
# shelf / maze, c3, c4, c5 fused and two-launch), the driver's settings, 2-rank debug runs (c2, c5), one-rank RCCL, rocprofv3 kernel
# stats, PMC passes (HBM traffic + SQ instruction counts; separate passes, --kernel-trace only; calibration on a 50 331 648-byte
# elementwise copy KERNEL in the same run), IK / task-API / ops benches.  Outputs under gpurun_out/r06/; judged copies -> profiles/
# (tools/r04_collect.py).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd $R
B="timeout 400 python bench.py"
$B --steps 2000 --warmup 200 > $O/bench_c2.json 2> $O/bench.err
$B --steps 2000 --warmup 200 --config c3 --cpu-seconds 6 > $O/bench_c3.json 2>> $O/bench.err
$B --steps 1000 --warmup 100 --config c4 --cpu-seconds 6 > $O/bench_c4.json 2>> $O/bench.err
$B --steps 1000 --warmup 100 --config c5 --cpu-seconds 6 > $O/bench_c5.json 2>> $O/bench.err
$B --steps 1000 --warmup 100 --config c5 --two-launch --cpu-seconds 0 > $O/bench_c5_two_launch.json 2>> $O/bench.err
$B --steps 1000 --warmup 100 --config c4 --two-launch --cpu-seconds 0 > $O/bench_c4_two_launch.json 2>> $O/bench.err
TRK_GP_ARM_LANES=1 $B --steps 1000 --warmup 100 --config c5 --cpu-seconds 0 > $O/bench_c5_arm_lanes.json 2>> $O/bench.err
for s in grid shelf maze; do $B --steps 1000 --warmup 100 --scene $s --cpu-seconds 6 > $O/bench_c2_$s.json 2>> $O/bench.err; done
$B --steps 1000 --warmup 100 --scene grid --q smooth --cpu-seconds 0 > $O/bench_c2_grid_smooth.json 2>> $O/bench.err
$B --steps 1000 --warmup 100 --q smooth --cpu-seconds 0 > $O/bench_c2_smooth.json 2>> $O/bench.err
$B --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c2_driver_settings.json 2>> $O/bench.err
$B --steps 20 --warmup 5 --cpu-seconds 0 --config c5 > $O/bench_c5_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --steps 2000 --warmup 200 > $O/bench_2rank_gloo_one_gpu.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --config c5 --steps 1000 --warmup 100 > $O/bench_c5_2rank_gloo_one_gpu.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_one_rank_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --exchange rccl --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_one_rank_rccl_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --steps 2000 --warmup 200 --cpu-seconds 0 > $O/bench_one_rank.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --config c5 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c5_one_rank_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --force-dist --config c3 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c3_one_rank_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --exchange p2p --graph 20 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_2rank_p2p_one_gpu.json 2>> $O/bench.err
# the 8-GPU run (and config 5's 4-GPU split) rehearsed on one GPU: N processes share cuda:0, world-N mailbox, the SCALE line's fields
timeout 900 python bench.py --gpus 8 --dist-backend gloo --single-device --exchange p2p --graph 20 --steps 20 --warmup 5 --batch 512 --cpu-seconds 3 > $O/bench_8rank_p2p_one_gpu.json 2>> $O/bench.err
timeout 900 python bench.py --config c5 --gpus 4 --dist-backend gloo --single-device --exchange p2p --graph 20 --steps 20 --warmup 5 --batch 256 --cpu-seconds 3 > $O/bench_c5_4rank_p2p_one_gpu.json 2>> $O/bench.err
for s in shelf maze; do $B --steps 1000 --warmup 100 --scene $s --q smooth --cpu-seconds 0 --no-out-of-cache > $O/bench_c2_${s}_smooth.json 2>> $O/bench.err; done
for b in 2048 8192 16384 32768 49152; do timeout 300 python bench.py --steps 300 --warmup 30 --batch $b --cpu-seconds 0 --no-out-of-cache >> $O/bench_batches.jsonl 2>> $O/bench.err; done
cd /tmp; export TMPDIR=/tmp
P="timeout 400 rocprofv3 --kernel-trace --stats --output-format csv"
$P -d $O/prof_c2 -o r06 -- python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --no-out-of-cache > $O/bench_c2_under_rocprof.json 2> $O/prof.err
$P -d $O/prof_c3 -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --config c3 > /dev/null 2>> $O/prof.err
$P -d $O/prof_c4 -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --config c4 > $O/bench_c4_under_rocprof.json 2>> $O/prof.err
$P -d $O/prof_c5 -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --config c5 > $O/bench_c5_under_rocprof.json 2>> $O/prof.err
$P -d $O/prof_c4_two_launch -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --config c4 --two-launch > /dev/null 2>> $O/prof.err
$P -d $O/prof_points -o r06 -- python3 $R/tools/bench_points.py > /dev/null 2>> $O/prof.err
$P -d $O/prof_f1 -o r06 -- python3 $R/tools/bench_task_api.py > /dev/null 2>> $O/prof.err
for s in grid shelf maze; do
  $P -d $O/prof_$s -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene $s --no-out-of-cache > /dev/null 2>> $O/prof.err
done
$P -d $O/prof_grid_smooth -o r06 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene grid --q smooth --no-out-of-cache > /dev/null 2>> $O/prof.err
cat > /tmp/calib.py <<'PY'
import torch
x = torch.rand(50331648 // 4, device="cuda"); y = torch.empty_like(x)
for _ in range(20): torch.mul(x, 1.0, out=y)        # an elementwise KERNEL (copy_ of equal dtypes is a DMA copy: no kernel record)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  Q="timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv"
  for s in spheres grid shelf maze; do
    $Q -d $O/pmc/${s}_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --scene $s --no-out-of-cache > /dev/null 2>> $O/pmc.err
  done
  $Q -d $O/pmc/gridsmooth_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --scene grid --q smooth --no-out-of-cache > /dev/null 2>> $O/pmc.err
  for cfg in c3 c4 c5; do
    $Q -d $O/pmc/${cfg}_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --config $cfg > /dev/null 2>> $O/pmc.err
  done
  if [ "$n" != "SQ_INSTS_VALU" ]; then $Q -d $O/pmc/calib_$n -o p -- python3 /tmp/calib.py > /dev/null 2>> $O/pmc.err; fi
done
cd $R
timeout 300 python tools/bench_task_api.py 2>/dev/null | grep -v "Warn\|as_tensor\|Python builtin\|third-party\|warn_once" > $O/bench_task_api.txt
timeout 300 python tools/bench_ik_gn.py 2>/dev/null > $O/bench_ik_gn.txt
timeout 300 python tools/bench_ops.py > $O/bench_ops.txt 2>/dev/null
timeout 300 python tools/bench_points.py > $O/bench_points.txt 2>/dev/null
# the exchange under the profiler: kernel stats + trace of the one-rank runs
cd /tmp
for cfg in c2 c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_exchange_$cfg -o r06 -- python3 $R/bench.py --force-dist --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > /dev/null 2>> $O/prof.err
done
cd $R
find $O -name "*.csv" | wc -l
