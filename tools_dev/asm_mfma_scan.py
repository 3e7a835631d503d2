"""Does any matrix instruction of the built library read a register that INLINE ASSEMBLY wrote just in front of it?

Why: gfx950 needs wait states between a vector-ALU write of a register and a v_mfma that reads it as A or B.  The compiler's hazard
recogniser inserts them for its own instructions (it keeps two: `s_nop 1` in 52 places of this library) but it does not look inside
asm statements.  atvs_split2_f16 (csrc/common.h) is inline assembly; in round 6 conv1 of csrc/bottleneck_b.hip -- the one place that
fed a multiply's B operand straight from the split, not through LDS -- computed garbage in the 64-channel unit as soon as the one-fma
batch norm removed the few instructions that used to sit between (fixed by giving that site the C form of the split).

tools_dev/micro/mfma_asm_hazard.hip measures the rule on the MI355X (profiles/round6_mfma_asm_hazard.txt): a multiply DIRECTLY behind
the write (v_mov / v_cvt_pk_f16_f32 / v_fma_mixhi_f16, A or B) reads the stale register in most lanes, idle SIMD or busy; any one
instruction between (s_nop 0, a scalar move, an s_waitcnt, another VALU instruction) is enough there; overwriting A / B / C right
BEHIND the multiply is harmless at every distance.  This scan asks for more than the probe needs: fewer than MIN_GAP = 4
instructions between an assembly write and a multiply reading it is a finding.

    python tools_dev/asm_mfma_scan.py [lib.so]      -> the findings; exit code 1 if there is one   (tests/test_asm_mfma_scan.py)

Assembly-only mnemonics in this library: v_fma_mixlo_f16 / v_fma_mixhi_f16 / v_fma_mix_f32; a v_cvt_pk_f16_f32 counts as the
assembly's when a v_fma_mix_f32 shortly behind reads its destination as first source (the compiler's own conversions are followed
by v_cvt_f32_f16 and carry the compiler's wait states)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MIN_GAP = 4            # instructions (s_nop N counts N + 1) between the assembly's write and the multiply that reads it
ASM_ONLY = ('v_fma_mixlo_f16', 'v_fma_mixhi_f16', 'v_fma_mix_f32')


def _regs(tok):
    """'v[26:29]' -> {26..29}, 'v7' -> {7}; anything else -> empty"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def disassemble(lib, tmp):
    dst = os.path.join(tmp, 'lib.so')
    shutil.copy(lib, dst)
    subprocess.run([OBJDUMP, '--offloading', dst], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    for f in sorted(os.listdir(tmp)):
        if 'gfx950' in f:
            out = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, f)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            yield out.decode('utf-8', 'replace')


BRANCHES = ('s_branch', 's_cbranch_scc0', 's_cbranch_scc1', 's_cbranch_vccz', 's_cbranch_vccnz', 's_cbranch_execz',
            's_cbranch_execnz', 's_setpc_b64', 's_endpgm', 's_barrier')


def _kernels(text):
    """-> {kernel: [(position, mnemonic, operands, text)]}; s_nop N advances the position by N + 1 and is dropped"""
    out, name, pos = {}, None, 0
    for ln in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:', ln)
        if m:
            name, pos = m.group(1), 0
            out[name] = []
            continue
        body = ln.split('//')[0].strip()
        if not body or name is None:
            continue
        parts = body.replace(',', ' ').split()
        if parts[0].startswith('s_nop'):
            pos += int(parts[1], 0) + 1
            continue
        pos += 1
        out[name].append((pos, parts[0], parts[1:], body))
    return out


def scan_text(text):
    """-> [(kernel, mfma line, asm line, instructions between)]: the assembly wrote a register the multiply reads as A / B fewer than
    MIN_GAP instructions later (s_nop N counts N + 1; the scan stops at branches and barriers).  A v_cvt_pk_f16_f32 counts as the assembly's when a v_fma_mix_f32 within the next 16 instructions
    reads its destination as first source (the compiler's own C form converts back with v_cvt_f32_f16 instead)."""
    found = []
    for name, ins in _kernels(text).items():
        asm_at = set()
        for i, (pos, op, args, body) in enumerate(ins):
            if op in ASM_ONLY:
                asm_at.add(i)
            elif op == 'v_cvt_pk_f16_f32':
                dst = args[0]
                for (_, op2, args2, _) in ins[i + 1:i + 17]:
                    if op2 == 'v_fma_mix_f32' and len(args2) > 1 and args2[1] == dst:
                        asm_at.add(i)
                        break
                    if args2 and args2[0] == dst:
                        break                                   # overwritten first
        for i, (pos, op, args, body) in enumerate(ins):
            if not op.startswith('v_mfma'):
                continue
            ab = _regs(args[1]) | _regs(args[2])
            j = i - 1
            while j >= 0 and pos - ins[j][0] - 1 < MIN_GAP and ins[j][1] not in BRANCHES:
                if j in asm_at and _regs(ins[j][2][0]) & ab:
                    found.append((name, body, ins[j][3], pos - ins[j][0] - 1))
                j -= 1
    return found


def scan(lib):
    tmp = tempfile.mkdtemp(prefix='asmscan')
    try:
        out = []
        for text in disassemble(lib, tmp):
            out += scan_text(text)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'a-tvsnet_amd',
                                                              'libatvsnet_hip.so')
    res = scan(lib)
    for (k, mf, wr, gap) in res:
        print('%s\n    %s\n    <- %s   (%d instructions between)' % (k, mf, wr, gap))
    print('%d finding(s)' % len(res))
    sys.exit(1 if res else 0)
