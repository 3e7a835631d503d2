"""profiles/<name>.json from the summary of tools_dev/pmc_bench.sh:  python tools_dev/make_pmc_table.py gpurun_out/pmc_<tag> [profiles/name.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_derive as PD                                      # noqa: E402

src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(src, 'summary.json')))
out = {'note': 'rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only; tools_dev/pmc_bench.sh) over `bench.py --eager '
               '--inflight 1` on MI355X: per-launch averages for the kernels of one depth map (config 3), '
               'sorted by total time.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs), cycles and clock_GHz as tools_dev/pmc_derive.py '
               'derives them (never a clock above the 2.4 GHz spec: short kernels use the reference clock of the long ones); valu_per_mfma = '
               '(SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA; fetch / write = FETCH_SIZE / '
               'WRITE_SIZE (KB at the memory side of L2, Infinity-Cache hits included; wide coalesced reads count at half their bytes '
               'on gfx950, MI355X_MICROARCH.md) as MB per launch, RAW.  Durations are under the profiler.', 'kernels': {}}
ref = PD.reference_clock((v['counters'], v.get('duration_ns_under_profiler', 0)) for v in d.values())
out['reference_clock_GHz'] = round(ref, 3) if ref else None
for k, v in list(d.items())[:48]:
    c = v['counters']
    dur = v.get('duration_ns_under_profiler', 0)
    e = {'launches_profiled': v['launches'], 'duration_us': round(dur / 1e3, 1), 'total_ms': round(v.get('total_ms_under_profiler', 0), 3),
         'fetch_MB_raw': round(c.get('FETCH_SIZE', 0) / 1024, 1), 'write_MB_raw': round(c.get('WRITE_SIZE', 0) / 1024, 1)}
    e.update({n: x for n, x in PD.derive(c, dur, ref).items() if x is not None and n != 'cycles'})
    if dur:
        e['memory_TBps_raw'] = round((c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) * 1024 / dur / 1e3, 2)
    e['lds_bank_conflict_cycles'] = int(c.get('SQ_LDS_BANK_CONFLICT', 0))
    out['kernels'][k] = e
busy, ms = PD.conv3d_time_weighted_busy(out['kernels'])
out['conv3d_mfma_busy_time_weighted'] = {'value': busy, 'over_ms_of_conv3d_kernels': ms, 'kernels': list(PD.CONV3D),
                                          'note': 'sum(mfma_busy x time) / sum(time) over the 3-D convolution kernels of one profiled pass '
                                                  '(the busy counter includes the three products of the operand split)'}
json.dump(out, open(os.path.join(root, sys.argv[2] if len(sys.argv) > 2 else 'profiles/round6_pmc_kernels.json'), 'w'), indent=1)
for k in list(out['kernels'])[:10]:
    print(k, out['kernels'][k])
