#!/bin/bash
# kernel-trace statistics of one bench.py run:  bash tools_dev/prof_one.sh tag  -> gpurun_out/<tag>/stats.csv
tag=${1:-one}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $root/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench.log 2>&1
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/stats.csv
rm -rf $out/prof
tail -1 $out/bench.log | cut -c1-200
