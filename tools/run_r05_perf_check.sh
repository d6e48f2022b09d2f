# Round 5: factorised self-pair phase of the attached-point kernels + the stream-store (nt) instantiation: parity tests, attached-point
# bench, batch sweep around the Infinity Cache with both store policies, in-cache gather rates.  Output: gpurun_out/r05p/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_custom_ops.py -q -k "stream_store or points or grasp or sphere or interpolated or native_rollout or graphed or host_tensors or full_size" > $O/pytest_perf.txt 2>&1
tail -12 $O/pytest_perf.txt
timeout 300 python tools/bench_points.py > $O/bench_points.txt 2>/dev/null; cat $O/bench_points.txt
b() { python bench.py --cpu-seconds 0 --no-out-of-cache "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   %-28s %8.2f us  frac %.3f' % (' '.join(sys.argv[1:]), d['roofline']['launch_us'], d['roofline']['frac']))" "$@"; }
{
for rep in 1 2; do
for mode in 0 1; do
  export TRK_STREAM_STORES=$mode
  echo "TRK_STREAM_STORES=$mode (0 = write-through sc1, 1 = non-temporal nt)  rep $rep"
  for B in 16384 20480 24576 28672 32768 49152; do b --batch $B --steps 200 --warmup 20; done
done
done
unset TRK_STREAM_STORES
echo "launch's own choice (threshold 256 MiB)"
for B in 16384 20480 24576 32768; do b --batch $B --steps 200 --warmup 20; done
} 2>&1 | tee $O/stream_store_sweep.txt
timeout 300 python bench.py --steps 2000 --warmup 200 > $O/bench_c2.json 2>> $O/bench.err
timeout 300 python bench.py --config c3 --batch 32768 --steps 100 --warmup 10 --cpu-seconds 0 > $O/bench_c3_32768.json 2>> $O/bench.err
for mib in 32 64 128 512 4096; do timeout 120 $R/tools/gather_calib.bin $mib 8 20 >> $O/gather_rates.txt 2>&1; done
cat $O/gather_rates.txt
