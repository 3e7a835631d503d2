#!/bin/bash
# PMC passes over one bench_conv.py configuration; run on the GPU box from the repo root:
#   bash tools_dev/pmc_conv.sh <tag> <D> <H> <W> <cin> <cout> <reps> [pb|sib]     (ATVS_LIB selects a development build)
tag=$1; shift
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 $root/tools_dev/bench_conv.py "$@" > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if 'conv_tiled' not in row['Kernel_Name'] and 'conv_xp' not in row['Kernel_Name']:
            continue
        a = acc[row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items():
        print('%-32s %16.0f  (per launch, %d launches)' % (k, v / n, n))
PY
