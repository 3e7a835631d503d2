# kernel-trace summary of the default bench command (per-map value = one map at a time): run on the GPU box from the repo root
#   bash tools_dev/prof_bench3.sh <tag>
tag=${1:-r3}
root=$PWD
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag/prof -o bench -- python3 $root/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-split-bf16 --no-power > $root/gpurun_out/$tag/bench_prof.log 2>&1
tail -1 $root/gpurun_out/$tag/bench_prof.log | cut -c1-400
f=$(find $root/gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
cp $f $root/gpurun_out/$tag/kernel_stats.csv
head -32 $f | cut -c1-180
