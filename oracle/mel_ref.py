"""Oracle: STFT / mel-filterbank target extraction.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows reference data_utils.py:39-62 (mel_spectrogram) and :29-30 (log-clamp).  The filterbank is
librosa.filters.mel (call site data_utils.py:47) -- a third-party dependency that is NOT in
/root/reference (environment.yml:17, unpinned) -> basis values are 'parity unpinned'; restated
here from the published Slaney (Auditory Toolbox) formulation that librosa documents:
  mel(f) = f / (200/3)                         for f < 1000 Hz
         = 15 + ln(f/1000) / (ln(6.4)/27)      for f >= 1000 Hz
triangular filters between n_mels+2 mel-equispaced edge frequencies evaluated on rfft bin
centres, each scaled by 2/(f[i+2]-f[i]) ("slaney" area normalisation), float32.
"""
import numpy as np

_F_SP = 200.0 / 3.0
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    log = _MIN_LOG_MEL + np.log(np.maximum(f, 1e-30) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * _F_SP
    log = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return np.where(m >= _MIN_LOG_MEL, log, lin)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """(n_mels, 1 + n_fft//2) float32 filterbank."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    fft_freqs = np.linspace(0.0, sr / 2.0, n_bins)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(edges)
    ramps = edges[:, None] - fft_freqs[None, :]
    w = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (edges[2:n_mels + 2] - edges[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def hann_periodic(n):
    """torch.hann_window(n) default periodic=True (data_utils.py:49)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def mel_spectrogram_ref(y, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256,
                        win_size=1024, fmin=0, fmax=8000, basis=None):
    """y: (B, L) float32 in [-1,1] -> (B, num_mels, F) float32, F = 1 + (L + 2*pad - n_fft)//hop.

    data_utils.py:51 reflect-pad (n_fft-hop)/2 each side; :54 STFT hann(periodic), center=False,
    onesided; :57 sqrt(re^2+im^2+1e-9); :59 mel @ spec; :60 log(clamp(.,1e-5)).
    Frame DFT computed in float64 and rounded once (the 'true' value both torch.stft and the HIP
    FFT approximate to ~1e-6).
    """
    y = np.asarray(y, dtype=np.float32)
    assert y.ndim == 2
    if basis is None:
        basis = slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax)
    pad = int((n_fft - hop_size) / 2)
    yp = np.pad(y, ((0, 0), (pad, pad)), mode='reflect')
    L = yp.shape[1]
    nfr = 1 + (L - n_fft) // hop_size
    win = np.zeros(n_fft, dtype=np.float32)
    off = (n_fft - win_size) // 2
    win[off:off + win_size] = hann_periodic(win_size)
    idx = np.arange(n_fft)[None, :] + hop_size * np.arange(nfr)[:, None]
    frames = yp[:, idx] * win[None, None, :]                       # (B, F, n_fft) f32
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)          # (B, F, 513)
    mag = np.sqrt((spec.real ** 2 + spec.imag ** 2) + 1e-9).astype(np.float32)
    mel = np.einsum('mk,bfk->bmf', basis.astype(np.float32), mag, dtype=np.float32, optimize=False)
    return np.log(np.maximum(mel, np.float32(1e-5))).astype(np.float32)
