"""Two backends for the kernel parity tests:
  'hip' -- the product: libsilent_speech_hip.so on a real MI355X (marked gpu);
  'emu' -- the SAME kernel sources compiled for the host SIMT emulator (tools/emu), CPU tensors,
           small sizes; validates index math / staging / MFMA fragment maps without a GPU.
"""
import os
import subprocess

import pytest
import torch

from silent_speech_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
EMU = os.path.join(ROOT, 'silent_speech_amd', 'lib', 'libsilent_speech_emu.so')
_built = False


def _ensure_emu():
    global _built
    if not _built:
        subprocess.check_call(['make', '-s', '-j8', '-C', os.path.join(ROOT, 'silent_speech_amd', 'csrc'), 'emu'])
        _built = True


BACKENDS = [pytest.param('emu'), pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    if request.param == 'emu':
        _ensure_emu()
        _lib.use_library_for_testing(EMU)
        return torch.device('cpu')
    _lib.load()
    assert not _lib.is_emulator()
    assert torch.cuda.is_available(), 'gpu-marked test needs an MI355X'
    return torch.device('cuda')


def is_emu(dev):
    return dev.type == 'cpu'
