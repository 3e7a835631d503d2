cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2l/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2l/bench.log 2>&1
tail -1 gpurun_out/r2l/bench.log | cut -c1-600
find gpurun_out/r2l/prof -name "*kernel_stats.csv" | head -1 | xargs head -25 | cut -c1-200
