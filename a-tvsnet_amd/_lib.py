"""Build and load ``libatvsnet_hip.so`` (the C-ABI of include/atvsnet_hip.h).

The library is built IN-TREE with hipcc for gfx950 (cross-compiles without a
GPU) and loaded with ctypes.  There is no fallback: if it is missing or fails
to load, every op raises.
"""
import ctypes
import glob
import hashlib
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('ATVS_LIB') or os.path.join(_HERE, 'libatvsnet_hip.so')   # ATVS_LIB: A/B a development build
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'atvsnet_hip.h')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -fno-slp-vectorize: no COMPILER-FORMED packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in kernels whose
# wavefronts can share a SIMD with another kernel's: beside wavefronts issuing 16x16x32 MFMAs (two depth maps in flight) such
# kernels produced wrong lane quarters (tests/test_gpu_pipeline.py, tests/test_packed_fp32_census.py).
# The one-workgroup-per-CU split-operand kernels listed below fill their SIMDs' register file (nothing runs beside them) and
# keep the vectoriser: their operand split is a fifth of a stage and runs 1.9 instead of 1.5 VALU instructions per MFMA without.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function', '-Wno-unused-result']
# conv_xb (round 4, second half) is not in the list although it owns its SIMDs: its STAGING wavefronts run beside its own MFMA
# wavefronts on every SIMD -- exactly the constellation of the fault -- so its vector arithmetic is kept scalar too.
# deconv_up_b runs TWO workgroups per CU (round 4): wavefronts of the same kernel share SIMDs -- scalar arithmetic as well.
OWNS_ITS_SIMD = ('conv_c16b', 'conv3d_b', 'conv3d_s2b')


def flags_for(src):
    stem = os.path.splitext(os.path.basename(src))[0]
    return FLAGS + ([] if stem in OWNS_ITS_SIMD else ['-fno-slp-vectorize'])


def _flags_stamp(src):
    """What an object file was built WITH: a flags-only change (e.g. a new mitigation flag) must rebuild it although no
    source is newer.  Kept next to the object as <stem>.flags."""
    return hashlib.sha1(' '.join([HIPCC] + flags_for(src)).encode()).hexdigest()


def _stamp_ok(src):
    try:
        with open(src[:-4] + '.flags') as f:
            return f.read().strip() == _flags_stamp(src)
    except OSError:
        return False


_lib = None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [HEADER]
    return any(os.path.getmtime(p) > t for p in deps) or not all(_stamp_ok(s) for s in sources())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 every csrc/*.hip into one shared library."""
    if not force and not _stale():
        return LIB_PATH
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + '.o'
        objs.append(obj)
        if not force and os.path.exists(obj) and _stamp_ok(src) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src), os.path.getmtime(HEADER)] +
                [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h'))]):
            continue
        cmd = [HIPCC] + flags_for(src) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), out.decode('utf-8', 'replace')))
        with open(src[:-4] + '.flags', 'w') as f:
            f.write(_flags_stamp(src) + '\n')
        if verbose and out:
            print(out.decode('utf-8', 'replace'))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        raise RuntimeError('link failed: %s\n%s' % (' '.join(cmd), p.stdout.decode('utf-8', 'replace')))
    return LIB_PATH


def header_abi_version():
    """ATVS_ABI_VERSION of include/atvsnet_hip.h."""
    with open(HEADER) as f:
        m = re.search(r'#define\s+ATVS_ABI_VERSION\s+(\d+)', f.read())
    if not m:
        raise RuntimeError('%s defines no ATVS_ABI_VERSION' % HEADER)
    return int(m.group(1))


def declared_symbols():
    """Every function name declared in include/atvsnet_hip.h."""
    with open(HEADER) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(atvs_[a-z0-9_]+)\s*\(', text)))


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libatvsnet_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"`; '
                'there is no CPU fallback for the HIP path.' % LIB_PATH)
        # torch first: its wheel carries its own HIP runtime, and the streams / device pointers handed to this
        # library belong to THAT runtime.  Loaded after torch, the library's hip* symbols bind to the runtime
        # already in the process; loaded before it, they would bind to /opt/rocm's copy and every launch on a
        # torch stream would fail.
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        for name in declared_symbols():
            fn = getattr(_lib, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = ctypes.c_int
        have, want = _lib.atvs_abi_version(), header_abi_version()
        if have != want:          # a stale prebuilt library next to newer sources
            _lib = None
            raise RuntimeError('%s has ABI version %d, include/atvsnet_hip.h declares %d: rebuild it '
                               '(`python -c "import __graft_entry__ as g; g.build()"`)' % (LIB_PATH, have, want))
        _lib.atvs_target_arch.restype = ctypes.c_char_p
        _lib.atvs_conv_num_blocks.restype = ctypes.c_long
        _lib.atvs_channel_stats_num_blocks.restype = ctypes.c_long
        _lib.atvs_avg_pool_ws_floats.restype = ctypes.c_long
        _lib.atvs_conv_tiled_num_blocks.restype = ctypes.c_long
        _lib.atvs_conv_tiled_grid.restype = ctypes.c_long
        _lib.atvs_conv_xpair_grid.restype = ctypes.c_long
        _lib.atvs_conv2d_lds_rows.restype = ctypes.c_long
        _lib.atvs_conv_stem_rows.restype = ctypes.c_long
        _lib.atvs_conv1x1_rows.restype = ctypes.c_long
        _lib.atvs_conv1x1_b_rows.restype = ctypes.c_long
        _lib.atvs_bottleneck_b_rows.restype = ctypes.c_long
        _lib.atvs_conv3d_s2b_grid.restype = ctypes.c_long
        _lib.atvs_deconv_up_grid.restype = ctypes.c_long
        _lib.atvs_conv_c16_grid.restype = ctypes.c_long
    return _lib
