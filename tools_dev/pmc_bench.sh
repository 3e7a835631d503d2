#!/bin/bash
# PMC passes over the eager form of bench.py (the very launches the timed graph replays), one rocprofv3 pass per counter
# group (--pmc with --kernel-trace only).  Run on the GPU box from the repo root:
#   bash tools_dev/pmc_bench.sh <tag> [bench.py arguments...]
# Writes gpurun_out/pmc_<tag>/summary.json: per kernel name, per-launch averages of every counter + the duration.
tag=$1; shift
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 $root/bench.py --eager --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-power "$@" > $out/p$i.log 2>&1
  tail -c 300 $out/p$i.log | head -c 300; echo
done
python3 $root/tools_dev/pmc_summary.py $out
