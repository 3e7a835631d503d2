#!/bin/bash
# per-dispatch kernel trace of a short bench run (the last depth map's launches in order):  bash tools_dev/prof_trace.sh tag
tag=${1:-trace}
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o bench -- python3 $root/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-fp32-path --no-power > $out/bench.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$out/last_map.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last graph replay: the launches after the last upsample_softargmin but one
idx = [i for i, r in enumerate(rows) if 'upsample_softargmin' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
with open(sys.argv[2], 'w') as f:
    for r in rows[lo:hi]:
        f.write('%s,%d,%s,%s\n' % (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:90].replace(',', ';'), int(r['End_Timestamp']) - int(r['Start_Timestamp']),
                                  r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
print(hi - lo, 'launches in the last map')
P
rm -rf $out/prof
