"""Which kernels of the library contain PACKED fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)?  (no GPU)

Why it matters (profiles/round4_coresident_matrix.md): on this pool's MI355X a wavefront executing packed
fp32 instructions on VGPR-pair operands computes wrong values in lanes 48-63 while ANOTHER kernel's wavefront issues
v_mfma_f32_16x16x32_{bf16,f16} on the same SIMD; scalar fp32 / integer / copy kernels never did (0 of 20 in every cell).  The
product therefore (1) runs one depth map at a time by default (no foreign wavefront anywhere), (2) lets the one-workgroup
split-operand kernels conv_c16b / conv3d_b / conv3d_s2b reserve their SIMD's whole register file, (3) builds every other source
with -fno-slp-vectorize.  NOT defended as aggressors (their 16x16x32 MFMA wavefronts can share a SIMD with another kernel's when
co-residency is switched on): conv2d_b, conv1x1_b, bottleneck_b (two workgroups per CU), deconv_up_b for 16 channels (two per CU; the
8-channel forms -- the plain one, one workgroup per CU, and the two-role summing decoder -- reserve their SIMDs' register file) -- which
is why the exposed victims below keep co-residency opt-in.  This census disassembles
the built library and pins the list of kernels that still contain packed fp32 instructions, so that a new one cannot slip in
unnoticed: such a kernel is only safe with two depth maps in flight if it owns its SIMD."""
import os
import re
import shutil
import subprocess

import pytest

from atvsnet_amd import _lib

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
# kernels that reserve the whole register file of their SIMD (asm volatile("" ::: "v255", "a255")): nothing runs beside them
OWNS_ITS_SIMD = ('conv_c16b_kernel', 'conv3d_b_kernel', 'conv3d_s2b_kernel')
# fp32-MFMA predecessors (A/B path, ops.configure(split16=False)): one workgroup per CU with 450-512 registers; packed fp32 by design
FP32_ONE_WORKGROUP = ('conv_xw_kernel', 'deconv_up_kernel', 'conv_c16_kernel')
# FMA-bound kernels with hand-written packed FMAs that share their SIMDs: the reason PipelinedInference(co_resident=True) is opt-in
KNOWN_EXPOSED = ('conv3d_8to1_kernel', 'refine_stems_kernel', 'probability_map_kernel')


def _census(tmp_path):
    lib = os.path.join(str(tmp_path), 'lib.so')
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, '--offloading', lib], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    counts = {}
    for f in sorted(os.listdir(str(tmp_path))):
        if 'gfx950' not in f:
            continue
        out = subprocess.run([OBJDUMP, '-d', os.path.join(str(tmp_path), f)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        name = None
        for ln in out.decode('utf-8', 'replace').splitlines():
            m = re.match(r'^[0-9a-f]+ <(.*)>:', ln)
            if m:
                name = m.group(1)
            elif name and re.search(r'\bv_pk_(mul|add|fma)_f32\b', ln):
                counts[name] = counts.get(name, 0) + 1
    return counts


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='no llvm-objdump')
def test_packed_fp32_only_where_it_is_accounted_for(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built')
    counts = _census(tmp_path)
    assert counts, 'the disassembly found no packed fp32 instruction at all: the census itself is broken'
    allowed = OWNS_ITS_SIMD + FP32_ONE_WORKGROUP + KNOWN_EXPOSED
    strangers = sorted(k for k in counts if not any(a in k for a in allowed))
    assert not strangers, ('kernels with packed fp32 arithmetic that neither own their SIMD nor are listed as exposed: %s -- build '
                           'their source with -fno-slp-vectorize / scalar arithmetic, or reserve the register file' % strangers)
    # the split-fp16 tower kernels run two workgroups per CU (they cannot own a SIMD): they must stay free of packed fp32
    for k in counts:
        assert 'conv2d_b_kernel' not in k and 'conv1x1_b_kernel' not in k and 'bottleneck_b_kernel' not in k, k
    # conv_xb's staging wavefronts compute (batch norm, operand split) beside its own MFMA wavefronts on the same SIMD: scalar fp32 only
    assert not any('conv_xb_kernel' in k for k in counts), [k for k in counts if 'conv_xb_kernel' in k]
    # aanet_b (round 6): the same constellation -- staging / softmax wavefronts beside its own MFMA wavefronts
    assert not any('aanet_b_kernel' in k for k in counts), [k for k in counts if 'aanet_b_kernel' in k]
    # deconv_up_b: two workgroups per CU, wavefronts of the same kernel share SIMDs
    assert not any('deconv_up_b_kernel' in k for k in counts), [k for k in counts if 'deconv_up_b_kernel' in k]
    # ... and the summing decoder (round 6): staging wavefronts beside its own MFMA wavefronts, as in conv_xb / aanet_b
    assert not any('deconv_up_b_sum_kernel' in k for k in counts), [k for k in counts if 'deconv_up_b_sum_kernel' in k]
    # the geometry / soft-argmin / norm kernels (the round-3 victims) are scalar
    for src in ('warp_planes', 'homographies_kernel', 'bn_add_kernel', 'bn_apply_kernel', 'softargmin_kernel', 'aanet_combine'):
        assert not any(src in k for k in counts), src
