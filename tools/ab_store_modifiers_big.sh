# Round 5, verdict item 9: the output stores' cache policy for launches whose working set exceeds the 256 MB Infinity Cache
# (32768 x 64: 403 MB) next to the in-cache headline (4096 x 64: 50 MB), same box, alternated.  Rebuilds only the Panda unit per variant.
# Output: gpurun_out/r05s/ab_store_modifiers.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
H=torch_robotics_amd/csrc/trk_spec_common.h
cp $H /tmp/spec_common.orig
b() { python bench.py --cpu-seconds 0 --no-out-of-cache "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   %-28s %8.2f us  frac %.3f' % (' '.join(sys.argv[1:]), d['roofline']['launch_us'], d['roofline']['frac']))" "$@"; }
{
for rep in 1 2; do
for mod in "sc1" "nt" "sc1 nt" "sc0 sc1" ""; do
  sed "s/ sc1\\\\n/ $mod\\\\n/g" /tmp/spec_common.orig > $H
  touch torch_robotics_amd/csrc/generated/spec_panda.hip
  make -C torch_robotics_amd/csrc -j 32 libtrk.so > /tmp/make.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/make.log; }
  echo "stores: [$mod]  (rep $rep)"
  b --batch 32768 --steps 200 --warmup 20
  b --batch 16384 --steps 300 --warmup 30
  b --steps 2000 --warmup 200
done
done
} 2>&1 | tee $O/ab_store_modifiers.txt
cp /tmp/spec_common.orig $H
make -C torch_robotics_amd/csrc -j 32 libtrk.so > /dev/null 2>&1
