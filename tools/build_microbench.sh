#!/bin/bash
# Build the standalone HIP microbenchmarks of tools/ for gfx950 (cross-compiles without a GPU); run them with gpurun.
cd "$(dirname "$0")"
for f in gather_calib clock_calib io_skeleton dispatch_ramp regen_gap valu_microbench3 valu_microbench2 valu_microbench sin_accuracy; do
  [ -f $f.hip ] && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $f.hip -o $f.bin 2>/dev/null && echo "built $f.bin"
done
