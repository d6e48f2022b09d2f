cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ops -o ops -- python3 $R/tools/bench_ops.py > /dev/null 2> $R/gpurun_out/prof_ops.err
ls -R $R/gpurun_out/prof_ops | head
