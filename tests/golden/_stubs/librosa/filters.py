import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..', '..')))
from oracle.mel_ref import slaney_mel_basis
def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw):
    return slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)
