# Round 5: the rebuilt pack kernel + peer-to-peer mailbox on hardware: tests, the driver's settings with one rank (RCCL process group) for c2 / c5,
# two ranks on one GPU, and a kernel trace of the one-rank run.  Outputs: gpurun_out/r05y/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05y
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -k "mailbox or packed or reassigned or no_bundled or build_defined or dual_panda_fp16" > $O/pytest_exchange.txt 2>&1
tail -15 $O/pytest_exchange.txt
for cfg in c2 c5; do
  timeout 300 python bench.py --force-dist --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_${cfg}_force_dist.json 2>> $O/bench.err
  timeout 300 python bench.py --force-dist --exchange rccl --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_${cfg}_force_dist_rccl.json 2>> $O/bench.err
done
timeout 300 python bench.py --force-dist --steps 2000 --warmup 200 --cpu-seconds 0 > $O/bench_c2_force_dist_2000.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --dist-backend gloo --single-device --exchange p2p --graph 20 --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_c2_2rank_p2p.json 2>> $O/bench.err
timeout 900 python -m pytest tests/test_bench_launch.py -q > $O/pytest_bench_launch.txt 2>&1
tail -15 $O/pytest_bench_launch.txt
cd /tmp; export TMPDIR=/tmp
for cfg in c2 c5; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -o x -- python3 $R/bench.py --force-dist --config $cfg --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_${cfg}_under_rocprof.json 2>> $O/prof.err
done
# FETCH_SIZE calibration on 16-byte / 8-byte random gathers + a streaming read, and the DRAM's gather rate
timeout 300 $R/tools/gather_calib.bin 4096 8 20 > $O/gather_calib.txt 2>&1
cat $O/gather_calib.txt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gather_fetch -o g -- $R/tools/gather_calib.bin 4096 8 4 > /dev/null 2>> $O/prof.err
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $O/gather_raw -o g -- $R/tools/gather_calib.bin 4096 8 4 > /dev/null 2>> $O/prof.err
ls -la $O/gather_fetch $O/gather_raw
grep -v "amdgpu.ids\|socket.cpp" $O/bench.err | tail -30
