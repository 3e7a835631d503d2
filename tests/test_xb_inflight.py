"""conv_xb.hip's staging wavefronts issue their vector-memory loads from inline assembly and wait for them by hand (exact
s_waitcnt vmcnt counts: DESIGN.md 4.1).  That is sound only while the compiler never moves a loaded register between its load and
its wait -- no copy, no spill, no scratch traffic.  tools_dev/check_xb_inflight.py compiles the file to assembly and checks it
(no GPU: hipcc cross-compiles gfx950)."""
import importlib.util
import os

import pytest

from atvsnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(_lib.HIPCC), reason='no hipcc')
def test_no_loaded_register_is_moved_or_spilled_while_in_flight(capsys):
    spec = importlib.util.spec_from_file_location('check_xb_inflight', os.path.join(ROOT, 'tools_dev', 'check_xb_inflight.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rc = mod.main()
    out = capsys.readouterr().out
    assert rc == 0, out
    # every instantiation was seen; the forms that split in the kernel have inline loads and waits, the four that take their input
    # in pieces (last template argument true: LDS-DMA through builtins, nothing in registers) have none
    lines = [ln for ln in out.splitlines() if 'inline loads' in ln]
    assert len(lines) == 11, out
    for ln in lines:
        if 'ELb1EEEvNS' in ln:
            assert ': 0 inline loads' in ln, ln
        else:
            assert ': 0 inline loads' not in ln and ' 0 inline waits' not in ln, ln
