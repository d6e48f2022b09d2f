# Round-3 measurement set (one gpurun call): bench lines (spheres / grid / shelf / maze, c2 / c3), the 2-rank debug run,
# rocprofv3 kernel stats of the default bench command and of the scene variants, PMC passes (HBM traffic + SQ instruction counts,
# separate passes, --kernel-trace only), task-API / jtj / points benches.  Outputs under gpurun_out/r03/; judged copies -> profiles/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 2000 --warmup 200 > $O/bench_c2.json 2> $O/bench.err
timeout 300 python bench.py --steps 2000 --warmup 200 --config c3 --cpu-seconds 0 > $O/bench_c3.json 2>> $O/bench.err
for s in grid shelf maze; do timeout 300 python bench.py --steps 1000 --warmup 100 --scene $s --cpu-seconds 6 > $O/bench_c2_$s.json 2>> $O/bench.err; done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c2_driver_settings.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --steps 2000 --warmup 200 > $O/bench_2rank_gloo_one_gpu.json 2>> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o r03 -- python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $O/bench_c2_under_rocprof.json 2> $O/prof.err
for s in grid shelf maze; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$s -o r03 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene $s > /dev/null 2>> $O/prof.err
done
cat > /tmp/calib.py <<'PY'
import torch
x = torch.empty(50331648 // 4, device="cuda"); y = torch.empty_like(x)
for _ in range(20): y.copy_(x)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  for s in spheres grid shelf maze; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/${s}_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --scene $s > /dev/null 2>> $O/pmc.err
  done
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/c3_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --config c3 > /dev/null 2>> $O/pmc.err
  if [ "$n" != "SQ_INSTS_VALU" ]; then timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/calib_$n -o p -- python3 /tmp/calib.py > /dev/null 2>> $O/pmc.err; fi
done
cd $R
timeout 300 python tools/bench_task_api.py > $O/bench_task_api.txt 2>/dev/null
timeout 300 python tools/bench_jtj.py > $O/bench_jtj.txt 2>/dev/null
timeout 300 python tools/bench_points.py > $O/bench_points.txt 2>/dev/null
timeout 300 python tools/bench_configs.py > $O/bench_c4_c5.json 2>/dev/null
timeout 300 python tools/bench_ops.py > $O/bench_ops.txt 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_jtj -o r03 -- python3 $R/tools/bench_jtj.py > /dev/null 2>> $O/prof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_task_api -o r03 -- python3 $R/tools/bench_task_api.py > /dev/null 2>> $O/prof.err
find $O -name "*.csv" | wc -l
