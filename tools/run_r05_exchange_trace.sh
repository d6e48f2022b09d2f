# Round 5, verdict item 1a: rocprofv3 kernel trace of the one-rank RCCL bench at the driver's settings (c2, c5): the pack kernel, RCCL's
# kernels and the gaps on the launch stream.  Outputs: gpurun_out/r05x/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05x
rm -rf $O; mkdir -p $O
cd $R
for cfg in c2 c5; do
  timeout 300 python bench.py --force-dist --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_${cfg}_force_dist.json 2>> $O/bench.err
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 --no-out-of-cache > $O/bench_${cfg}_plain.json 2>> $O/bench.err
done
cd /tmp; export TMPDIR=/tmp
for cfg in c2 c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -o x -- python3 $R/bench.py --force-dist --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_${cfg}_under_rocprof.json 2>> $O/prof.err
done
find $O -name "*.csv" | xargs ls -la
