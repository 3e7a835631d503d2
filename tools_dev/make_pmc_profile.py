"""gpurun_out/pmc_<tag>/summary.json (tools_dev/pmc_bench.sh) -> profiles/<name>.json: the x-pair (conv_xw) launches exactly as
bench.py issues them, counters per launch + derived figures.   usage: make_pmc_profile.py <tag> <profiles/name.json>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_derive as PD                                      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
S = json.load(open(os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag, 'summary.json')))
V, V2 = 192 * 128 * 160, 96 * 64 * 80
KERN = sys.argv[3] if len(sys.argv) > 3 else 'conv_xb_kernel'
FORMS = {   # kernel name -> (description, volumes per launch, algorithmic FLOP, algorithmic read bytes, write bytes)
    KERN + '<true, 0, true, true>': ('conv_b0_0_1 | conv_b0_1_0: 32 warped channels -> 8 | 16 (stride 2), plane biases', 8,
                                2.0 * 27 * 32 * 8 * V + 2.0 * 27 * 32 * 16 * V2, 4.0 * 32 * V, 4.0 * (8 * V + 16 * V2)),
    KERN + '<true, 2, false, false>': ('conv_b{1,2}_0_1 | 1_0: two 8-channel sources, batch norm + ReLU + add on load -> 8 | 16', 8,
                                2.0 * 27 * 8 * 8 * V + 2.0 * 27 * 8 * 16 * V2, 4.0 * 16 * V, 4.0 * (8 * V + 16 * V2)),
    KERN + '<true, 1, true, false>': ('global_refine_3dconv0_1 | 1_0: 32-channel concat, batch norm + ReLU on load -> 8 | 16', 4,
                                2.0 * 27 * 32 * 8 * V + 2.0 * 27 * 32 * 16 * V2, 4.0 * 32 * V, 4.0 * (8 * V + 16 * V2)),
    KERN + '<false, 0, true, true>': ('photo stem: 16 D-varying channels -> 8, plane bias', 4, 2.0 * 27 * 16 * 8 * V, 4.0 * 16 * V, 4.0 * 8 * V),
}
res = {'note': 'rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over `bench.py --eager --inflight 1` on '
               'MI355X (tools_dev/pmc_bench.sh): the launches of the timed graph, per-launch averages.  mfma_busy = '
               'SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); cycles / clock as tools_dev/pmc_derive.py derives them; '
               'FETCH_SIZE / WRITE_SIZE are KB at the L2 memory side (Infinity-Cache hits included; wide coalesced reads count at '
               'half their bytes on gfx950, other widths uncalibrated: MI355X_MICROARCH.md).  Durations are under the profiler.',
       'kernels': {}}
REF = PD.reference_clock((v['counters'], v.get('duration_ns_under_profiler', 0)) for v in S.values())
for name, (desc, vols, flop, rd, wr) in FORMS.items():
    k = S.get(name)
    if not k:
        continue
    c = k['counters']
    dur = k['duration_ns_under_profiler'] * 1e-9
    mf = c.get('SQ_INSTS_MFMA', 0.0)
    dv = PD.derive(c, k['duration_ns_under_profiler'], REF)
    e = {'layer': desc, 'volumes_per_launch': vols, 'launches_profiled': k['launches'],
         'duration_ns_under_profiler': round(k['duration_ns_under_profiler']),
         'counters_per_launch': {n: round(v) for n, v in sorted(c.items())},
         'algorithmic_flops': flop * vols, 'algorithmic_read_bytes': rd * vols, 'algorithmic_write_bytes': wr * vols,
         'achieved_TFLOPs_algorithmic': round(flop * vols / dur / 1e12, 1),
         'fraction_of_fp32_mfma_peak_157.3': round(flop * vols / dur / 1e12 / 157.3, 3),
         'clock_GHz': dv.get('clock_GHz'), 'clock_source': dv.get('clock_source'),
         'mfma_busy_fraction_of_simd_cycles': dv.get('mfma_busy'),
         'lds_active_fraction_of_cu_cycles': dv.get('lds_active'),
         'issued_mfma': round(mf), 'valu_per_mfma': dv.get('valu_per_mfma'),
         'lds_bank_conflict_cycles': round(c.get('SQ_LDS_BANK_CONFLICT', 0.0)),
         'write_bytes': round(c.get('WRITE_SIZE', 0.0) * 1024), 'fetch_bytes_raw': round(c.get('FETCH_SIZE', 0.0) * 1024)}
    alg = (rd + wr) * vols
    e['traffic_ratio_raw'] = round((e['write_bytes'] + e['fetch_bytes_raw']) / alg, 3)
    e['traffic_ratio_fetch_x2'] = round((e['write_bytes'] + 2 * e['fetch_bytes_raw']) / alg, 3)
    res['kernels'][name] = e
res['dominant'] = dict(res['kernels'][KERN + '<true, 0, true, true>'], kernel=KERN + '<true, 0, true, true>')
others = {}
for name, k in S.items():
    if name in FORMS or 'total_ms_under_profiler' not in k or k['total_ms_under_profiler'] < 0.5:
        continue
    c = k['counters']
    dv = PD.derive(c, k['duration_ns_under_profiler'], REF)
    others[name] = {'launches_profiled': k['launches'], 'avg_us': round(k['duration_ns_under_profiler'] / 1e3, 1),
                    'fetch_MB_raw': round(c.get('FETCH_SIZE', 0.0) / 1024, 1), 'write_MB': round(c.get('WRITE_SIZE', 0.0) / 1024, 1),
                    'mfma_busy': dv.get('mfma_busy'), 'clock_GHz': dv.get('clock_GHz'),
                    'valu_per_mfma': dv.get('valu_per_mfma'),
                    'lds_bank_conflict_cycles': round(c.get('SQ_LDS_BANK_CONFLICT', 0.0))}
res['other_kernels_of_the_step'] = others
json.dump(res, open(os.path.join(ROOT, out), 'w'), indent=1)
d = res['dominant']
print('dominant: %.2f ms, %.1f TF/s algorithmic (%.3f of peak), MFMA busy %.2f at %.2f GHz, %.2f VALU/MFMA, traffic x%.2f (raw) / x%.2f (fetch doubled)'
      % (d['duration_ns_under_profiler'] / 1e6, d['achieved_TFLOPs_algorithmic'], d['fraction_of_fp32_mfma_peak_157.3'],
         d['mfma_busy_fraction_of_simd_cycles'], d['clock_GHz'], d['valu_per_mfma'], d['traffic_ratio_raw'], d['traffic_ratio_fetch_x2']))
