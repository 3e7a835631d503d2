cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2m/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --inflight 1 > gpurun_out/r2m/bench.log 2>&1
tail -1 gpurun_out/r2m/bench.log | cut -c1-200
