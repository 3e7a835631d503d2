"""No matrix instruction of the built library reads a register that inline assembly wrote just in front of it.  (no GPU)

The compiler's hazard recogniser keeps wait states between a vector-ALU write and a v_mfma read of the same register, but not for
instructions inside asm statements; tools_dev/micro/mfma_asm_hazard.hip shows on the MI355X that a multiply directly behind such a
write reads the stale register (profiles/round6_mfma_asm_hazard.txt), and round 6 hit exactly that in csrc/bottleneck_b.hip.  The
scan (tools_dev/asm_mfma_scan.py) disassembles the library and must find nothing."""
import os
import sys

import pytest

from atvsnet_amd import _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools_dev'))
import asm_mfma_scan  # noqa: E402

HAZARD = '''
0000000000001000 <kernel_a>:
	v_cvt_pk_f16_f32 v26, v40, v41                             // 000000001000: 00000000
	v_cvt_pk_f16_f32 v29, v117, v122                           // 000000001008: 00000000
	s_mov_b32 s2, 0                                            // 000000001010: 00000000
	v_mfma_f32_16x16x32_f16 v[90:93], v[110:113], v[26:29], 0  // 000000001018: 00000000
	v_fma_mix_f32 v110, v26, -1.0, v40 op_sel_hi:[1,0,0]       // 000000001020: 00000000
	v_fma_mix_f32 v111, v29, -1.0, v117 op_sel_hi:[1,0,0]      // 000000001028: 00000000
0000000000002000 <kernel_b>:
	v_fma_mixhi_f16 v17, v30, s2, 0 op_sel_hi:[0,0,0]          // 000000002000: 00000000
	s_nop 3                                                    // 000000002008: 00000000
	v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[14:17], v[20:23]  // 000000002010: 00000000
0000000000003000 <kernel_c>:
	v_cvt_pk_f16_f32 v29, v117, v122                           // 000000003000: 00000000
	v_cvt_f32_f16_e32 v30, v29                                 // 000000003008: 00000000
	s_nop 1                                                    // 000000003010: 00000000
	v_mfma_f32_16x16x32_f16 v[90:93], v[110:113], v[26:29], 0  // 000000003018: 00000000
'''


def test_the_scan_sees_an_assembly_write_in_front_of_a_multiply():
    found = asm_mfma_scan.scan_text(HAZARD)
    kernels = sorted({f[0] for f in found})
    assert kernels == ['kernel_a'], found            # b: four wait states between; c: the compiler's own conversion
    assert any('v29' in f[2] and f[3] == 1 for f in found), found


@pytest.mark.skipif(not os.path.exists(asm_mfma_scan.OBJDUMP), reason='no llvm-objdump')
def test_no_multiply_reads_what_inline_assembly_just_wrote():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built')
    found = asm_mfma_scan.scan(_lib.LIB_PATH)
    assert not found, 'inline assembly feeds a matrix instruction without wait states:\n' + '\n'.join(
        '%s: %s <- %s (%d between)' % f for f in found[:10])
