#!/bin/bash
# kernel-trace statistics of bench.py for one workload:  bash tools_dev/prof_wl.sh cfg5 tag
wl=${1:-cfg5}; tag=${2:-wl}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $root/bench.py --workload $wl --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench.log 2>&1
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/stats.csv
rm -rf $out/prof
