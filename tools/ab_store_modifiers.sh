b() { for i in 1 2 3; do python bench.py --cpu-seconds 0 --steps 3000 | python3 -c "import json,sys; d=json.load(sys.stdin); print(\"  \", d[\"roofline\"][\"launch_us\"])"; done; }
cp torch_robotics_amd/csrc/trk_spec_common.h /tmp/spec_common.orig
for mod in "sc1" "nt" "sc1 nt" "sc0 sc1" "sc0 sc1 nt" ""; do
  sed "s/off sc1\\\\n/off $mod\\\\n/g" /tmp/spec_common.orig > torch_robotics_amd/csrc/trk_spec_common.h
  touch torch_robotics_amd/csrc/generated/spec_panda.hip
  make -C torch_robotics_amd/csrc > /dev/null 2>&1 || echo BUILD FAILED
  echo "stores: [$mod]"; b
done
cp /tmp/spec_common.orig torch_robotics_amd/csrc/trk_spec_common.h
