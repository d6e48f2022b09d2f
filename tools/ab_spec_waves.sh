# A/B of the workgroup size of the generated kernels (SPEC_WAVES wavefronts per workgroup); rebuilds on the GPU box.
b() { for i in 1 2 3; do python bench.py --cpu-seconds 0 --steps 3000 | python3 -c "import json,sys; d=json.load(sys.stdin); print(\"  \", d[\"roofline\"][\"launch_us\"])"; done; }
cp torch_robotics_amd/csrc/trk_spec_common.h /tmp/spec_common.orig
for w in 4 2 8 1; do
  sed "s/#define SPEC_WAVES 4/#define SPEC_WAVES $w/" /tmp/spec_common.orig > torch_robotics_amd/csrc/trk_spec_common.h
  touch torch_robotics_amd/csrc/generated/spec_panda.hip
  make -C torch_robotics_amd/csrc > /dev/null 2>&1 || echo BUILD FAILED
  echo "SPEC_WAVES=$w"; b
done
cp /tmp/spec_common.orig torch_robotics_amd/csrc/trk_spec_common.h
