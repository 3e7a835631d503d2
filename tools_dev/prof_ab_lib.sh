#!/bin/bash
# kernel-trace statistics of bench.py with two builds of the library (same box):  bash tools_dev/prof_ab_lib.sh <lib A | -> <lib B | -> tag
# ("-" = the product's library; a variant: tools_dev/build_variant.sh -> tools_dev/_dbg/lib_<name>.so)
la=$1; lb=$2; tag=${3:-ablib}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for l in $la $lb; do
  if [ "$l" = "-" ]; then unset ATVS_LIB; else export ATVS_LIB=$root/$l; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$i -o bench -- python3 $root/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench$i.log 2>&1
  cp $(find $out/prof$i -name "*kernel_stats.csv" | head -1) $out/stats$i.csv
  rm -rf $out/prof$i
  i=$((i+1))
done
