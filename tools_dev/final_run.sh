# Round-end validation on the GPU box: full -m gpu suite, smoke, default bench (+ other workloads), kernel trace, PMC passes.
#   bash tools_dev/final_run.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/$tag/gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/$tag/smoke.log
python bench.py > gpurun_out/$tag/bench_default.log 2>&1
for w in cfg2 cfg4 cfg5; do python bench.py --workload $w --no-cpu-baseline --no-fp32-path > gpurun_out/$tag/bench_$w.log 2>&1; done
bash tools_dev/prof_bench3.sh $tag > gpurun_out/$tag/prof.log 2>&1
bash tools_dev/pmc_bench.sh $tag > gpurun_out/$tag/pmc.log 2>&1
cat gpurun_out/$tag/gputests.log gpurun_out/$tag/smoke.log; for f in default cfg2 cfg4 cfg5; do tail -1 gpurun_out/$tag/bench_$f.log | cut -c1-200; done
