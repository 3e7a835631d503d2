"""Derived figures of a kernel's rocprofv3 PMC counters (shared by make_pmc_table.py / make_pmc_profile.py).

Cycles of a kernel.  GRBM_GUI_ACTIVE (summed over the 8 XCDs) spans MORE than a short kernel -- dispatch, the drain of the previous
kernel -- so GRBM_GUI_ACTIVE / 8 / duration printed 3-6 GHz for sub-100-us kernels in round 5 and every mfma_busy derived from it was
too low by that factor.  SQ_BUSY_CYCLES (summed over the 32 shader engines) counts only cycles in which the kernel's waves are
resident.  cycles = the smaller of GRBM_GUI_ACTIVE / 8 and SQ_BUSY_CYCLES / 32; if that still implies a clock above the 2.4 GHz
spec, the kernel's cycles are duration x the REFERENCE clock: the duration-weighted mean clock of the step's kernels longer than
200 us (where both counters agree).  `clock_source` says which was used; a clock above 2.4 GHz is never printed."""
SPEC_GHZ = 2.4
LONG_NS = 200e3
CONV3D = ('conv_xb_kernel', 'conv_c16b_kernel', 'conv3d_b_kernel', 'conv3d_s2b_kernel', 'deconv_up_b_kernel', 'deconv_up_b_sum_kernel', 'aanet_b_kernel')


def raw_cycles(c):
    cand = []
    if c.get('GRBM_GUI_ACTIVE'):
        cand.append(c['GRBM_GUI_ACTIVE'] / 8.0)
    if c.get('SQ_BUSY_CYCLES'):
        cand.append(c['SQ_BUSY_CYCLES'] / 32.0)
    return min(cand) if cand else None


def reference_clock(kernels):
    """kernels: iterable of (counters, duration_ns).  GHz held over the long kernels of the step (duration-weighted)."""
    cyc = dur = 0.0
    for c, d in kernels:
        rc = raw_cycles(c)
        if rc and d >= LONG_NS and rc / d <= SPEC_GHZ:
            cyc += rc
            dur += d
    return cyc / dur if dur else None


def derive(c, dur_ns, ref_ghz):
    """-> dict(cycles, clock_GHz, clock_source, mfma_busy, valu_per_mfma, lds_active) for one kernel (per-launch averages)."""
    out = {}
    rc = raw_cycles(c)
    if rc and dur_ns and rc / dur_ns <= SPEC_GHZ:
        cycles, src = rc, 'min(GRBM_GUI_ACTIVE / 8, SQ_BUSY_CYCLES / 32)'
    elif ref_ghz and dur_ns:
        cycles, src = dur_ns * ref_ghz, 'duration x the reference clock of the long kernels (own counters span more than the kernel)'
    else:
        cycles, src = None, None
    out['cycles'] = round(cycles) if cycles else None
    out['clock_GHz'] = round(cycles / dur_ns, 3) if cycles and dur_ns else None
    out['clock_source'] = src
    mf = c.get('SQ_INSTS_MFMA', 0.0)
    if cycles and mf:
        out['mfma_busy'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cycles * 1024), 3)
        out['valu_per_mfma'] = round((c.get('SQ_INSTS_VALU', 0.0) - mf) / mf, 2)
    if cycles and c.get('SQ_LDS_IDX_ACTIVE'):
        out['lds_active'] = round(c['SQ_LDS_IDX_ACTIVE'] / (cycles * 256), 3)
    return out


def conv3d_time_weighted_busy(entries):
    """entries: {kernel name: dict with 'mfma_busy' and 'total_ms'} -> the time-weighted MFMA-busy fraction over the 3-D convolution
    kernels of the step (north_star: 'conv3d >= 40 % MFMA utilisation'), with the time it covers."""
    t = b = 0.0
    for name, e in entries.items():
        if any(k in name for k in CONV3D) and e.get('mfma_busy') is not None and e.get('total_ms'):
            t += e['total_ms']
            b += e['total_ms'] * e['mfma_busy']
    return (round(b / t, 3), round(t, 3)) if t else (None, 0.0)
