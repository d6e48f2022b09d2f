# rocprofv3 kernel-trace stats of the default bench command (the summary committed under profiles/)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $R/gpurun_out/bench_plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r01b -- python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $R/gpurun_out/bench_plain2.json 2>/dev/null
