# Round-2 measurement set (one gpurun call): GPU tests, bench c2 / c3, rocprofv3 kernel stats of the default bench command,
# PMC HBM traffic (separate passes), unfused ops.  Outputs under gpurun_out/r02/ ; the judged copies go to profiles/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python bench.py --steps 2000 --warmup 200 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 300 python bench.py --steps 2000 --warmup 200 --config c3 --cpu-seconds 0 > $O/bench_c3.json 2>> $O/bench_c2.err
for b in 2048 8192 16384; do timeout 200 python bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --batch $b; done > $O/bench_batches.jsonl 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $O/prof_bench.json 2> $O/prof.err
timeout 300 python3 $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 > $O/bench_after_prof.json 2>/dev/null
cat > /tmp/calib.py <<'PY'
import torch
x = torch.empty(50331648 // 4, device="cuda"); y = torch.empty_like(x)
for _ in range(20): y.copy_(x)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  for cfg in c2 c3; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hbm/${cfg}_$c -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --config $cfg > /dev/null 2>> $O/pmc_hbm.err
  done
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hbm/calib_$c -o p -- python3 /tmp/calib.py > /dev/null 2>> $O/pmc_hbm.err
done
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_sq/$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 > /dev/null 2>> $O/pmc_sq.err
done
cd $R
timeout 300 python tools/bench_ops.py > $O/bench_ops.txt 2>/dev/null
find $O -name "*.csv" | head -30
