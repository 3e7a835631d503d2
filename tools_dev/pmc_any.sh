#!/bin/bash
# PMC passes over any python development script (its path relative to the repo root, then its arguments); run on the GPU box from the repo root:
#   bash tools_dev/pmc_layer.sh <tag> <kernel name substring> <script.py> <arguments...>
# One rocprofv3 pass per counter group (--pmc with --kernel-trace only); prints per-launch averages of the matching kernel.
tag=$1; shift
kern=$1; shift
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/p$i -- python3 $root/"$@" > $out/p$i.log 2>&1
done
tail -1 $out/p1.log
python3 - <<PY
import csv, glob, collections, json
res = {}
dur = []
for f in sorted(glob.glob('$out/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if '$kern' not in row['Kernel_Name']:
            continue
        a = acc[row['Counter_Name']]
        a[0] += float(row['Counter_Value']); a[1] += 1
    for k, (v, n) in acc.items():
        res[k] = v / n
        print('%-32s %16.0f  (per launch, %d launches)' % (k, v / n, n))
for f in sorted(glob.glob('$out/p3/**/*kernel_trace.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if '$kern' in row['Kernel_Name']:
            dur.append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
if dur:
    res['duration_ns_under_profiler'] = sum(dur) / len(dur)
    print('duration_ns_under_profiler %.0f' % res['duration_ns_under_profiler'])
json.dump(res, open('$out/summary.json', 'w'), indent=1)
PY
