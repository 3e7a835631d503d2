"""conv_xb.hip issues its producers' vector-memory loads from inline assembly and waits for them by hand (s_waitcnt vmcnt(N) with
the loaded registers as "+v" operands).  That is only sound if the compiler never MOVES a loaded value between its load and its
wait: no copy, no spill, and no scratch traffic at all (scratch operations count in vmcnt too).  This script compiles the file
to assembly and checks, per kernel:
  * no scratch_ instruction and no spill / reload anywhere;
  * no register-to-register move (v_mov_b32, v_pk_mov_b32, v_accvgpr_write/read, v_swap) whose source or destination is a register
    that some inline-assembly load of the kernel writes.  (After its wait a loaded value is consumed by arithmetic -- conversions,
    subtractions -- or by ds_write directly; a plain move of such a register is the signature of the register allocator
    shuffling a live range, which is what must not happen while the load is in flight.)

    python tools_dev/check_xb_inflight.py        (exit 1 on a violation; run by tests/test_xb_inflight.py)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import _lib                           # noqa: E402


def regs_of(text):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    src = os.path.join(ROOT, 'a-tvsnet_amd', 'csrc', 'conv_xb.hip')
    flags = [f for f in _lib.flags_for('conv_xb.hip') if f != '-fPIC']
    asm = subprocess.run([_lib.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', '-', src], check=True,
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    bad = 0
    kernels = {}
    cur = None
    in_asm = False
    for ln, line in enumerate(asm.splitlines(), 1):
        t = line.strip()
        m = re.match(r'^(_ZN\S*conv_xb_kernel\S*):', t)
        if m:
            cur = kernels.setdefault(m.group(1), {'loaded': set(), 'plain': [], 'nload': 0, 'nwait': 0, 'first': None, 'last': 0})
            continue
        if cur is None or not t:
            continue
        if t.startswith('s_endpgm'):
            cur = None
            continue
        if 'ASMSTART' in t:
            in_asm = True
            continue
        if 'ASMEND' in t:
            in_asm = False
            continue
        if t.startswith(';'):
            continue
        if 'scratch_' in t or 'Spill' in t or 'Reload' in t:
            print('line %d: spill / scratch traffic: %s' % (ln, t))
            bad += 1
        if in_asm:
            m = re.match(r'(buffer|global)_load_dwordx4\s+(v\[\d+:\d+\])', t)
            if m:
                cur['loaded'] |= regs_of(m.group(2))
                cur['nload'] += 1
                cur['first'] = cur['first'] or ln
                cur['last'] = ln
            if t.startswith('s_waitcnt vmcnt'):
                cur['nwait'] += 1
            continue
        if re.match(r'(v_mov_b32|v_pk_mov_b32|v_accvgpr_write|v_accvgpr_read|v_swap_b32|v_mov_b64)', t):
            cur['plain'].append((ln, t))
    for name, k in kernels.items():
        # the producers' code = the text between the kernel's first and last inline load (the consumers' code, behind it, reuses
        # the same physical registers for its own values)
        hits = [(ln, t) for ln, t in k['plain'] if k['first'] and k['first'] <= ln <= k['last'] and regs_of(t.split(';')[0]) & k['loaded']]
        for ln, t in hits:
            print('%s: line %d moves a register that an inline load writes: %s' % (name, ln, t))
        bad += len(hits)
        print('%s: %d inline loads into %d registers, %d inline waits, %d moves of them' %
              (name[-28:], k['nload'], len(k['loaded']), k['nwait'], len(hits)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
