# Round 5: the box scenes again after the constant-link change (bench lines, kernel stats, PMC passes), merged into gpurun_out/r05/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O/pmc
cd $R
B="timeout 400 python bench.py"
for s in shelf maze; do $B --steps 1000 --warmup 100 --scene $s --cpu-seconds 6 > $O/bench_c2_$s.json 2>> $O/bench.err; done
for s in shelf maze; do $B --steps 1000 --warmup 100 --scene $s --q smooth --cpu-seconds 0 --no-out-of-cache > $O/bench_c2_${s}_smooth.json 2>> $O/bench.err; done
cd /tmp; export TMPDIR=/tmp
P="timeout 400 rocprofv3 --kernel-trace --stats --output-format csv"
for s in shelf maze; do
  rm -rf $O/prof_$s
  $P -d $O/prof_$s -o r05 -- python3 $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --scene $s --no-out-of-cache > /dev/null 2>> $O/prof.err
done
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  Q="timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv"
  for s in shelf maze; do
    rm -rf $O/pmc/${s}_$n
    $Q -d $O/pmc/${s}_$n -o p -- python3 $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --scene $s --no-out-of-cache > /dev/null 2>> $O/pmc.err
  done
done
find $O/pmc -name "*counter_collection.csv" | wc -l
