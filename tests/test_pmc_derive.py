"""CPU: the arithmetic behind profiles/round6_pmc_*.json (tools_dev/pmc_derive.py).  Round 5 derived a kernel's clock as
GRBM_GUI_ACTIVE / 8 / duration and printed 3-6 GHz for short kernels (the counter spans more than the kernel), which made every
mfma_busy of a sub-100-us kernel too low by that factor.  Pinned here: no clock above the 2.4 GHz spec comes out, short
kernels fall back to SQ_BUSY_CYCLES or the long kernels' clock, and the time-weighted conv3d figure is what its name says."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools_dev'))
import pmc_derive as PD          # noqa: E402


def test_long_kernel_uses_its_own_counters():
    c = {'GRBM_GUI_ACTIVE': 8 * 2.0e6, 'SQ_BUSY_CYCLES': 32 * 1.9e6, 'SQ_VALU_MFMA_BUSY_CYCLES': 0.5 * 1.9e6 * 1024,
         'SQ_INSTS_MFMA': 1000.0, 'SQ_INSTS_VALU': 3000.0}
    d = PD.derive(c, 1.0e6, None)                                   # 1 ms
    assert d['cycles'] == 1900000 and d['clock_GHz'] == 1.9
    assert abs(d['mfma_busy'] - 0.5) < 1e-9 and d['valu_per_mfma'] == 2.0


def test_short_kernel_never_reports_a_clock_above_the_spec():
    # GRBM_GUI_ACTIVE spans dispatch: 6 GHz by the old formula; SQ_BUSY_CYCLES is per kernel
    c = {'GRBM_GUI_ACTIVE': 8 * 30000.0, 'SQ_BUSY_CYCLES': 32 * 9500.0, 'SQ_VALU_MFMA_BUSY_CYCLES': 0.25 * 9500.0 * 1024, 'SQ_INSTS_MFMA': 10.0,
         'SQ_INSTS_VALU': 30.0}
    d = PD.derive(c, 5000.0, 2.0)                                   # 5 us
    assert d['clock_GHz'] == 1.9 and abs(d['mfma_busy'] - 0.25) < 1e-9
    # both counters span too much: the reference clock of the long kernels
    c2 = {'GRBM_GUI_ACTIVE': 8 * 30000.0, 'SQ_BUSY_CYCLES': 32 * 20000.0, 'SQ_VALU_MFMA_BUSY_CYCLES': 1024 * 1000.0, 'SQ_INSTS_MFMA': 1.0, 'SQ_INSTS_VALU': 2.0}
    d2 = PD.derive(c2, 5000.0, 2.0)
    assert d2['clock_GHz'] == 2.0 and d2['cycles'] == 10000 and 'reference clock' in d2['clock_source']
    assert abs(d2['mfma_busy'] - 0.1) < 1e-9


def test_reference_clock_and_time_weighting():
    ks = [({'GRBM_GUI_ACTIVE': 8 * 2.0e6, 'SQ_BUSY_CYCLES': 32 * 2.0e6}, 1.0e6),      # 2.0 GHz, long
          ({'GRBM_GUI_ACTIVE': 8 * 5.4e5, 'SQ_BUSY_CYCLES': 32 * 5.4e5}, 3.0e5),      # 1.8 GHz, long
          ({'GRBM_GUI_ACTIVE': 8 * 3.0e4}, 5.0e3)]                                   # short: ignored
    ref = PD.reference_clock(ks)
    assert abs(ref - (2.0e6 + 5.4e5) / (1.0e6 + 3.0e5)) < 1e-9
    busy, ms = PD.conv3d_time_weighted_busy({'conv_xb_kernel<true>': {'mfma_busy': 0.5, 'total_ms': 3.0},
                                             'deconv_up_b_kernel<8>': {'mfma_busy': 0.1, 'total_ms': 1.0},
                                             'bn_add_kernel': {'mfma_busy': None, 'total_ms': 9.0},
                                             'conv2d_b_kernel<2>': {'mfma_busy': 0.9, 'total_ms': 5.0}})      # a 2-D kernel: not counted
    assert ms == 4.0 and busy == 0.4


def test_the_committed_tables_obey_it():
    for name in ('round6_pmc_kernels.json', 'round6_pmc_xpair.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        for k, v in d['kernels'].items():
            if v.get('clock_GHz') is not None:
                assert v['clock_GHz'] <= PD.SPEC_GHZ, (name, k, v['clock_GHz'])
            b = v.get('mfma_busy', v.get('mfma_busy_fraction_of_simd_cycles'))
            assert b is None or 0.0 <= b <= 1.0, (name, k, b)
    tw = json.load(open(os.path.join(ROOT, 'profiles', 'round6_pmc_kernels.json')))['conv3d_mfma_busy_time_weighted']
    assert 0.0 < tw['value'] < 1.0 and tw['over_ms_of_conv3d_kernels'] > 0
