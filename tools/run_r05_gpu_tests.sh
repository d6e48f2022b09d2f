# Round 5: the whole GPU suite + smoke on a fresh box (what the driver runs at round end).  Output: gpurun_out/r05t/
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05t
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu -x --durations=15 > $O/pytest_gpu.txt 2>&1
tail -40 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 300 python tools/bench_task_api.py 2>/dev/null | grep -v "Warn\|as_tensor\|Python builtin\|third-party\|warn_once" > $O/bench_task_api.txt; cat $O/bench_task_api.txt
# the run-time compiled units the suite left behind: a cache that travels with the tree (git-ignored, keyed by a location-independent stamp
# of the generator, the headers and the compile command) saves the next fresh box their compile time
( cd $R/torch_robotics_amd/csrc && tar czf $O/jit_cache.tgz jit ) && ls -la $O/jit_cache.tgz
