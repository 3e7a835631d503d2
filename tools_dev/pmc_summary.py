"""Summarise the rocprofv3 --pmc passes of tools_dev/pmc_bench.sh: per kernel name, per-launch averages."""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', name).strip()


res = collections.defaultdict(lambda: {'counters': {}, 'launches': 0})
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        a = acc[(short(row['Kernel_Name']), row['Counter_Name'])]
        a[0] += float(row['Counter_Value'])
        a[1] += 1
    for (k, c), (v, n) in acc.items():
        res[k]['counters'][c] = v / n
        res[k]['launches'] = n
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/p3/**/*kernel_trace.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        dur[short(row['Kernel_Name'])].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
for k, v in dur.items():
    res[k]['duration_ns_under_profiler'] = sum(v) / len(v)
    res[k]['total_ms_under_profiler'] = sum(v) / 1e6
rows = sorted(res.items(), key=lambda kv: -kv[1].get('total_ms_under_profiler', 0))
json.dump({k: v for k, v in rows}, open(out + '/summary.json', 'w'), indent=1)
for k, v in rows[:40]:
    c = v['counters']
    d = v.get('duration_ns_under_profiler', 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0)
    busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui / 8 * 1024) if gui else 0
    vpm = (c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0)) / c['SQ_INSTS_MFMA'] if c.get('SQ_INSTS_MFMA') else 0
    print('%-58s n=%3d %8.1f us  fetch %8.1f MB  write %8.1f MB  mfma_busy %.2f  valu/mfma %.2f  ldsconf %.0f' % (
        k[:58], v['launches'], d / 1e3, c.get('FETCH_SIZE', 0) / 1024, c.get('WRITE_SIZE', 0) / 1024, busy, vpm,
        c.get('SQ_LDS_BANK_CONFLICT', 0)))
