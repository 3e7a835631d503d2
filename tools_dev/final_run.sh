mkdir -p gpurun_out/r3p
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5) > gpurun_out/r3p/gputests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3p/smoke.log 2>&1
python bench.py > gpurun_out/r3p/bench_default.log 2>&1
for w in cfg2 cfg4 cfg5; do python bench.py --workload $w --no-cpu-baseline --no-split-bf16 > gpurun_out/r3p/bench_$w.log 2>&1; done
bash tools_dev/prof_bench3.sh r3p > gpurun_out/r3p/prof.log 2>&1
bash tools_dev/pmc_bench.sh r3p > gpurun_out/r3p/pmc.log 2>&1
tail -3 gpurun_out/r3p/gputests.log; cat gpurun_out/r3p/smoke.log | tail -2; for f in default cfg2 cfg4 cfg5; do tail -1 gpurun_out/r3p/bench_$f.log | cut -c1-330; done
