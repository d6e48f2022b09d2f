set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 2000 --warmup 200 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
python bench.py --steps 2000 --warmup 200 --config c3 --cpu-seconds 0 > gpurun_out/bench_c3.json 2>> gpurun_out/bench_c2.err
python tools/bench_configs.py > gpurun_out/bench_c4_c5.json 2>/dev/null
python tools/bench_ops.py > gpurun_out/bench_ops.txt 2>/dev/null
python tools/bench_robots.py > gpurun_out/bench_robots.txt 2>/dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01b -- python3 $R/bench.py --steps 500 --warmup 50 --cpu-seconds 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
ls -R $R/gpurun_out/prof | head -20
