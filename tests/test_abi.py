"""The C-ABI library loads and exports every symbol include/silent_speech_hip.h declares (no compute calls);
the product path refuses to run without the gfx950 library / on CPU tensors."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
HIP_SO = os.path.join(ROOT, 'silent_speech_amd', 'lib', 'libsilent_speech_hip.so')


def _declared():
    src = open(os.path.join(ROOT, 'include', 'silent_speech_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ss_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def hip_lib():
    if not os.path.exists(HIP_SO):
        subprocess.check_call(['make', '-s', '-j8', '-C', os.path.join(ROOT, 'silent_speech_amd', 'csrc'), 'hip'])
    return ctypes.CDLL(HIP_SO)


def test_every_declared_symbol_is_exported(hip_lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), 'include/silent_speech_hip.h declares %s but the library does not export it' % n


def test_ctypes_table_covers_the_header():
    from silent_speech_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib._HOST_FUNCS) | set(_lib._RESTYPES)
    assert set(_declared()) == bound


def test_library_is_gfx950_and_reports_errors(hip_lib):
    hip_lib.ss_target_arch.restype = ctypes.c_char_p
    assert hip_lib.ss_target_arch() == b'gfx950'
    out = subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-readelf', '-S', HIP_SO]).decode() if os.path.exists('/opt/rocm/lib/llvm/bin/llvm-readelf') else '.hip_fatbin'
    assert '.hip_fatbin' in out          # carries device code (not a host-only stub)
    hip_lib.ss_last_error.restype = ctypes.c_char_p
    rc = hip_lib.ss_dtw_align(None, None, 1, 4, 4, None, None, None)     # argument validation happens before any launch
    assert rc != 0 and b'null pointer' in hip_lib.ss_last_error()


def test_product_path_has_no_cpu_fallback():
    from silent_speech_amd import _lib, align
    import numpy as np
    _lib.load()                       # the gfx950 library
    assert not _lib.is_emulator()
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(Exception):
        align.align_from_distances(np.ones((4, 4), dtype=np.float32), device=torch.device('cpu'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.ptr(torch.zeros(4))


def test_missing_library_fails_loudly(tmp_path):
    from silent_speech_amd import _lib
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load(str(tmp_path / 'libsilent_speech_hip.so'))
    _lib.load()


def test_emulator_library_is_refused_by_the_product_loader():
    """Only tests may run the host-emulator build (use_library_for_testing); load() -- and therefore SS_AMD_LIBRARY -- takes gfx950 builds only."""
    from silent_speech_amd import _lib
    from tests.backend import EMU, _ensure_emu
    _ensure_emu()
    with pytest.raises(RuntimeError, match='not gfx950'):
        _lib.load(EMU)
    _lib.use_library_for_testing(EMU)
    assert _lib.is_emulator()
    _lib.load()
    assert not _lib.is_emulator()


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'silent_speech_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.h'):
                txt = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, os.path.join(dp, f)


def test_abi_version_is_checked_at_load(hip_lib, monkeypatch):
    """A library whose ABI version differs from the binding's is refused at load time (struct layouts change between versions)."""
    from silent_speech_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'silent_speech_hip.h')).read()
    declared = int(re.search(r'#define\s+SS_ABI_VERSION\s+(\d+)', src).group(1))
    hip_lib.ss_abi_version.restype = ctypes.c_int
    assert hip_lib.ss_abi_version() == declared == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, 'ABI_VERSION', declared + 1)
    with pytest.raises(RuntimeError, match='ABI version'):
        _lib.load()
    monkeypatch.undo()
    _lib.load()
