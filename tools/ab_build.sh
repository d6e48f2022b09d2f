# builds a variant copy of the working tree under _ab/<name> with other GENFLAGS for the generated units (same-box A/B, see ab_run.sh)
# usage: bash tools/ab_build.sh <name> <GENFLAGS...>     (HEAD's committed tree: pass --head as the first flag)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $R/_ab/$name; mkdir -p $R/_ab/$name
if [ "$1" = "--head" ]; then shift; (cd $R && git archive HEAD) | tar -x -C $R/_ab/$name
else (cd $R && tar -c --exclude=./_ab --exclude=./.git --exclude=./gpurun_out --exclude='*.o' --exclude='libtrk.so' --exclude='./torch_robotics_amd/csrc/jit' .) | tar -x -C $R/_ab/$name; fi
cd $R/_ab/$name/torch_robotics_amd/csrc
python -c "import sys; sys.path.insert(0, '$R/_ab/$name'); from torch_robotics_amd import codegen; codegen.generate_all('generated')"
if [ $# -gt 0 ]; then make -j8 GENFLAGS="$*" > /dev/null; else make -j8 > /dev/null; fi
ls -la libtrk.so | awk '{print $5, $9}'
