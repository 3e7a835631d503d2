#!/bin/bash
# The GPU-box half of tools_dev/final_run.sh (run from the repository root of the snapshot):  bash tools_dev/final_box.sh <tag>
tag=${1:-r6final}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cp profiles/.measured_head $out/head 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $out/smoke.log
python bench.py > $out/bench_default.log 2>&1
for w in cfg2 cfg4 cfg5; do python bench.py --workload $w --no-cpu-baseline --no-fp32-path --no-power > $out/bench_$w.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 $root/bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench_prof.log 2>&1
cd $root
bash tools_dev/pmc_bench.sh $tag > $out/pmc.log 2>&1
cat $out/gputests.log $out/smoke.log
for f in default cfg2 cfg4 cfg5 prof; do tail -1 $out/bench_$f.log | cut -c1-220; done
tail -3 $out/pmc.log | cut -c1-200
