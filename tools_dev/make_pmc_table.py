"""profiles/<name>.json from the summary of tools_dev/pmc_bench.sh:  python tools_dev/make_pmc_table.py gpurun_out/pmc_<tag> [profiles/name.json]"""
import json
import os
import sys

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(src, 'summary.json')))
out = {'note': 'rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only; tools_dev/pmc_bench.sh) over `bench.py --eager '
               '--inflight 1` on MI355X: per-launch averages for the kernels of one depth map (config 3), '
               'sorted by total time.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); valu_per_mfma = '
               '(SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA; clock_GHz = GRBM_GUI_ACTIVE / 8 / duration; fetch / write = FETCH_SIZE / '
               'WRITE_SIZE (KB at the memory side of L2, Infinity-Cache hits included; wide coalesced reads count at half their bytes '
               'on gfx950, MI355X_MICROARCH.md) as MB per launch, RAW.  Durations are under the profiler.', 'kernels': {}}
for k, v in list(d.items())[:40]:
    c = v['counters']
    dur = v.get('duration_ns_under_profiler', 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0)
    e = {'launches_profiled': v['launches'], 'duration_us': round(dur / 1e3, 1), 'total_ms': round(v.get('total_ms_under_profiler', 0), 3),
         'fetch_MB_raw': round(c.get('FETCH_SIZE', 0) / 1024, 1), 'write_MB_raw': round(c.get('WRITE_SIZE', 0) / 1024, 1)}
    if gui and dur:
        e['clock_GHz'] = round(gui / 8 / dur, 2)
    if c.get('SQ_INSTS_MFMA'):
        e['mfma_busy'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui / 8 * 1024), 3)
        e['valu_per_mfma'] = round((c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'], 2)
    if dur:
        e['memory_TBps_raw'] = round((c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / dur / 1e3, 2)
    e['lds_bank_conflict_cycles'] = int(c.get('SQ_LDS_BANK_CONFLICT', 0))
    out['kernels'][k] = e
json.dump(out, open(os.path.join(root, sys.argv[2] if len(sys.argv) > 2 else 'profiles/round4_pmc_kernels.json'), 'w'), indent=1)
for k in list(out['kernels'])[:10]:
    print(k, out['kernels'][k])
