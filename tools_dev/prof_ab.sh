#!/bin/bash
# kernel-trace statistics of bench.py under two settings of one environment switch (same box):  bash tools_dev/prof_ab.sh VAR tag
var=$1; tag=${2:-ab}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export $var=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$v -o bench -- python3 $root/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench$v.log 2>&1
  cp $(find $out/prof$v -name "*kernel_stats.csv" | head -1) $out/stats$v.csv
  rm -rf $out/prof$v
done
