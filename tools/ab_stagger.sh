# Round 5 experiment: de-phase the two wavefronts a SIMD holds in the two-wavefront-per-SIMD kernels (config 5, config 4, attached points).
# "phases do not overlap, because with two wavefronts per SIMD, both at the same point of the same program, nothing covers a
# wavefront's own latencies".  Tree B: in the FIRST generation of workgroups (workgroup id < 512 = 2 per CU) the wavefront in an odd wave
# slot of its SIMD (HW_ID.WAVE_ID & 1) sleeps N x 64 cycles before its first load; afterwards the two slots stay out of phase by themselves.
#   tools/ab_stagger.sh "32 64 127"        Output: gpurun_out/r05stag/ab_stagger.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05stag; mkdir -p $O
b() { ( cd $1 && python bench.py --cpu-seconds 0 --no-out-of-cache "${@:2}" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('   %-40s step %8.3f us  kernel %8.3f us' % (' '.join(sys.argv[1:]), d['ms_per_step']*1e3, d['roofline']['launch_us']))" "${@:2}" ); }
run() { echo "tree $2"; if [ -z "$POINTS_ONLY" ]; then b $1 --config c5 --steps 1000 --warmup 100; b $1 --config c4 --steps 500 --warmup 50; fi; ( cd $1 && python tools/bench_points.py 2>/dev/null | grep "fused rollout\|fk_map_collision  " ); }
{
for N in ${1:-64}; do
  B=/tmp/treeB_$N
  rm -rf $B; cp -r $R $B; rm -rf $B/gpurun_out $B/torch_robotics_amd/csrc/jit
  python3 - $B/torch_robotics_amd/codegen.py $N <<'PY'
import sys
p, n = sys.argv[1], int(sys.argv[2])
s = open(p).read()
anchor = 'E.raw("    const int rows = (int)max((int64_t)0, min((int64_t)TRK_WAVE, A.n - base));")'
sleeps = " ".join("__builtin_amdgcn_s_sleep(%d);" % min(127, n - k) for k in range(0, n, 127))
stag = ('E.raw("    { unsigned hwid_; asm volatile(\\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\\" : \\"=s\\"(hwid_)); '
        'if ((hwid_ & 1u) && __builtin_amdgcn_workgroup_id_x() < 512u) { ' + sleeps + ' } }")')
parts = s.split(anchor)
assert len(parts) == 15, len(parts)
# only the fused rollouts: the first occurrence (k_rollout / k_rollout_gpt of the link units) and the attached-point rollout (13th)
out = ""
for k, part in enumerate(parts[:-1]):
    out += part + anchor + ("\n        " + stag if k in (0, 12) else "")
s = out + parts[-1]
open(p, "w").write(s)
print("patched", s.count(stag), "generators: sleep", n, "x 64 cycles")
PY
  ( cd $B && make -C torch_robotics_amd/csrc -j 64 libtrk.so > /tmp/makeB_$N.log 2>&1 ) || { echo "BUILD B FAILED"; tail -5 /tmp/makeB_$N.log; continue; }
  for rep in 1 2; do
    run $R "A (committed)  rep $rep"
    run $B "B (odd wave slots of the first generation sleep $N x 64 cycles)  rep $rep"
  done
done
} 2>&1 | tee $O/ab_stagger.txt
