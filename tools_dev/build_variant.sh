#!/bin/bash
# Development build of the library with extra flags for ONE source (A/B on the GPU box through ATVS_LIB=<the .so>):
#   bash tools_dev/build_variant.sh <name> <source stem, e.g. conv_xw> "<extra hipcc flags>"
# -> tools_dev/_dbg/lib_<name>.so (git-ignored; travels with gpurun)
name=$1; stem=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
src=$root/a-tvsnet_amd/csrc
mkdir -p $root/tools_dev/_dbg
# the product's own per-file flags (incl. -fno-slp-vectorize where _lib.flags_for sets it), then the extra ones
flags=$(cd $root && python3 -c "import atvsnet_amd; from atvsnet_amd import _lib; print(' '.join(_lib.flags_for('$stem.hip')))") || exit 1
/opt/rocm/bin/hipcc $flags "$@" \
  -c $src/$stem.hip -o $root/tools_dev/_dbg/${stem}_$name.o || exit 1
objs=$(ls $src/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools_dev/_dbg/lib_$name.so $objs $root/tools_dev/_dbg/${stem}_$name.o && echo built lib_$name.so
