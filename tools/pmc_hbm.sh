# HBM traffic of the headline kernel: separate --pmc passes (kernel-trace only), plus a calibration copy of known size.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/calib.py <<'PY'
import torch
x = torch.empty(50331648 // 4, device="cuda"); y = torch.empty_like(x)
for _ in range(20): y.copy_(x)
torch.cuda.synchronize()
PY
for c in FETCH_SIZE WRITE_SIZE; do
  for cfg in c2 c3; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_hbm/${cfg}_$c -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --config $cfg > /dev/null 2>> $R/gpurun_out/pmc_hbm.err
  done
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_hbm/calib_$c -o p -- python3 /tmp/calib.py > /dev/null 2>> $R/gpurun_out/pmc_hbm.err
done
find $R/gpurun_out/pmc_hbm -name "*counter_collection.csv" | wc -l
