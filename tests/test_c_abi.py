"""The boundary as a C program sees it (no GPU): include/atvsnet_hip.h is valid C99 on its own, and a plain-C caller
(tests/c/abi_smoke.c) loads the library, agrees with it on the ABI version and gets the documented status codes."""
import os
import shutil
import subprocess

import pytest

from atvsnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CC = shutil.which('gcc') or shutil.which('cc')


@pytest.mark.skipif(CC is None, reason='no C compiler')
def test_header_is_plain_c99():
    r = subprocess.run([CC, '-std=c99', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-x', 'c', _lib.HEADER],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()


@pytest.mark.skipif(CC is None, reason='no C compiler')
def test_plain_c_caller_loads_the_library(tmp_path):
    _lib.build()
    exe = str(tmp_path / 'abi_smoke')
    r = subprocess.run([CC, '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', exe, '-ldl'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    r = subprocess.run([exe, _lib.LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout.decode()
    assert ('abi %d ok' % _lib.header_abi_version()) in r.stdout.decode()


def test_flags_only_change_marks_the_library_stale(monkeypatch):
    """An object file carries a stamp of the hipcc flags it was built with (<stem>.flags): a flags-only change -- e.g. a
    new mitigation flag in _lib.flags_for -- must rebuild it although no source is newer (round-3 advisor finding)."""
    from atvsnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built')
    src = _lib.sources()[0]
    assert _lib._stamp_ok(src) and not _lib._stale()
    monkeypatch.setattr(_lib, 'FLAGS', _lib.FLAGS + ['-DSOMETHING_NEW'])
    assert not _lib._stamp_ok(src) and _lib._stale()
