"""Offline EMG conditioning on the MI355X -- drop-in for the signal-processing helpers of the reference's read_emg.py
("next" row N4 of SURVEY section 8f):

    remove_drift(signal, fs)                      read_emg.py:27-29   filtfilt(butter(3, 2, 'highpass', fs=fs))
    notch(signal, freq, sample_frequency)         read_emg.py:31-33   filtfilt(iirnotch(freq, 30, sample_frequency))
    notch_harmonics(signal, freq, sample_freq)    read_emg.py:35-38   7 harmonics
    subsample(signal, new_freq, old_freq)         read_emg.py:40-44   np.interp onto the new grid
    apply_to_all(function, signal_array, ...)     read_emg.py:46-50   per-channel application
    condition_raw_emg_recording(x)                read_emg.py:65-70   the whole chain of load_utterance on a (T, 8) recording

The filters are designed here (closed forms of scipy.signal.iirnotch / butter / lfilter_zi, f64) and run by csrc/filters.hip; all
channels of a recording go through ONE cascade launch, so `apply_to_all(notch_harmonics, x, 60, 1000)` costs the same as one channel.
Inputs may be numpy arrays or tensors; the arithmetic is f64 on the device and results come back in the input's container.
File I/O, text alignment, the 112-d hand-crafted features (unused by the model) and dataset bookkeeping stay out of scope.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------ filter design (f64, closed forms of the scipy routines)
def iirnotch_coeffs(w0, Q, fs):
    """scipy.signal.iirnotch(w0, Q, fs): second-order notch, -3 dB bandwidth w0 / Q."""
    w0 = 2.0 * float(w0) / float(fs)
    bw = w0 / float(Q) * math.pi
    w0 = w0 * math.pi
    gb = 1.0 / math.sqrt(2.0)
    beta = (math.sqrt(1.0 - gb ** 2.0) / gb) * math.tan(bw / 2.0)
    gain = 1.0 / (1.0 + beta)
    b = gain * np.array([1.0, -2.0 * math.cos(w0), 1.0])
    a = np.array([1.0, -2.0 * gain * math.cos(w0), 2.0 * gain - 1.0])
    return b, a


def butter_highpass_coeffs(order, cutoff, fs):
    """scipy.signal.butter(order, cutoff, 'highpass', fs=fs): analog prototype -> frequency pre-warping -> lp2hp -> bilinear -> tf."""
    N = int(order)
    wn = 2.0 * float(cutoff) / float(fs)
    m = np.arange(-N + 1, N, 2)
    p = -np.exp(1j * math.pi * m / (2 * N))               # buttap
    k = 1.0
    fs2 = 2.0
    warped = 2.0 * fs2 * math.tan(math.pi * wn / fs2)
    # lp2hp_zpk
    z_hp = np.zeros(N, dtype=complex)
    p_hp = warped / p
    k_hp = k * np.real(1.0 / np.prod(-p))
    # bilinear_zpk
    fs4 = 2.0 * fs2
    z_d = (fs4 + z_hp) / (fs4 - z_hp)
    p_d = (fs4 + p_hp) / (fs4 - p_hp)
    k_d = k_hp * np.real(np.prod(fs4 - z_hp) / np.prod(fs4 - p_hp))
    b = k_d * np.real(np.poly(z_d))
    a = np.real(np.poly(p_d))
    return b, a


def lfilter_zi(b, a):
    """scipy.signal.lfilter_zi: the state of the transposed direct form II in steady state for a unit step input,
    zi = (I - A)^-1 B with A the transposed companion matrix of a."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64)); a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if a[0] != 1.0:
        b, a = b / a[0], a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]; b = np.r_[b, np.zeros(n - len(b))]
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    comp[1:, :-1] = np.eye(n - 2)
    IminusA = np.eye(n - 1) - comp.T
    B = b[1:] - a[1:] * b[0]
    zi = np.linalg.solve(IminusA, B)                      # the LAPACK solve scipy uses: the drift filter's I - A is ill-conditioned (poles at
                                                          # 0.99), and its closed form differs from this in the 10th digit
    return zi


_PACKED = {}


def _pack_filters(filters):
    """(b, a, zi, order, padlen) rows of a cascade; a pure function of the coefficients, cached (the steady-state solve per section was
    0.8 ms of host time per batch in the device loader, for the same eight filters every time)."""
    key = tuple((tuple(np.asarray(b, dtype=np.float64).ravel().tolist()), tuple(np.asarray(a, dtype=np.float64).ravel().tolist())) for b, a in filters)
    hit = _PACKED.get(key)
    if hit is None:
        if len(_PACKED) > 64:
            _PACKED.clear()
        hit = _PACKED[key] = _pack_filters_uncached(filters)
    return hit


def _pack_filters_uncached(filters):
    rows = []
    for b, a in filters:
        b = np.asarray(b, dtype=np.float64); a = np.asarray(a, dtype=np.float64)
        b, a = b / a[0], a / a[0]
        n = max(len(a), len(b)) - 1
        if not 1 <= n <= 3:
            raise ValueError('filter order 1..3 supported, got %d' % n)
        zi = lfilter_zi(b, a)
        row = np.zeros(13)
        row[:len(b)] = b; row[4:4 + len(a)] = a; row[8:8 + n] = zi; row[11] = n; row[12] = 3 * max(len(a), len(b))
        rows.append(row)
    return np.ascontiguousarray(np.stack(rows, 0))


# ------------------------------------------------------------------ device entry points
def _to_device_2d(signal):
    """-> (tensor (T, C) f64 on the kernel device, restore function)"""
    is_np = not torch.is_tensor(signal)
    t = torch.from_numpy(np.ascontiguousarray(signal, dtype=np.float64)) if is_np else signal.to(torch.float64)
    one_d = t.dim() == 1
    if one_d:
        t = t.unsqueeze(1)
    if t.dim() != 2:
        raise ValueError('signal must be (T,) or (T, channels)')
    dev = _lib.kernel_device_for(t)
    t = t.to(dev).contiguous()

    def restore(y):
        y = y[:, 0] if one_d else y
        return y.cpu().numpy() if is_np else y
    return t, restore


def filtfilt_cascade(filters, signal):
    """scipy.signal.filtfilt(b, a, signal, axis=0) for every (b, a) of `filters` in turn, all channels at once."""
    x, restore = _to_device_2d(signal)
    T, C = x.shape
    coef = _pack_filters(filters)
    maxpad = int(coef[:, 12].max())
    if T <= maxpad:
        raise ValueError('The length of the input vector x must be greater than padlen, which is %d.' % maxpad)      # scipy's message
    L = _lib.lib()
    nbytes = int(L.ss_iir_filtfilt_workspace_bytes(T, C, maxpad))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    rc = L.ss_iir_filtfilt(_lib.ptr(x), _lib.ptr(y), T, C, coef.shape[0], coef.ctypes.data_as(ctypes.c_void_p), _lib.ptr(ws), nbytes, _lib.stream_of(x))
    _lib.check(rc, 'ss_iir_filtfilt')
    return restore(y)


def filtfilt_cascade_batch(filters, signals):
    """filtfilt_cascade for a RAGGED BATCH: `signals` = list of (T_u, C) f64 device tensors of different lengths -> list of filtered
    tensors (views of one packed buffer).  One launch sequence for all recordings (8 launches per filter + 1), instead of one per
    recording: the recurrences are serial in time, so the parallelism is (chunks x channels) and a batch supplies many more of them.
    `signals` may also be the pair (packed (sum T, C) f64 device tensor, lengths): the recordings back to back, as one upload left them."""
    if isinstance(signals, tuple):
        x, lens = signals[0], np.asarray([int(n) for n in signals[1]], dtype=np.int32)
        if x.dtype != torch.float64 or x.dim() != 2 or int(x.shape[0]) != int(lens.sum()) or not x.is_contiguous():
            raise ValueError('packed signals: a contiguous (sum(lengths), C) float64 tensor')
        signals = [None] * len(lens)
        dev, C = x.device, int(x.shape[1])
    else:
        dev = signals[0].device
        C = int(signals[0].shape[1])
        lens = np.asarray([int(t.shape[0]) for t in signals], dtype=np.int32)
        x = torch.cat([t.to(torch.float64) for t in signals], 0).contiguous() if len(signals) > 1 else signals[0].to(torch.float64).contiguous()
    coef = _pack_filters(filters)
    maxpad = int(coef[:, 12].max())
    if int(lens.min()) <= maxpad:
        raise ValueError('The length of the input vector x must be greater than padlen, which is %d.' % maxpad)
    L = _lib.lib()
    lp = lens.ctypes.data_as(ctypes.c_void_p)
    nbytes = int(L.ss_iir_filtfilt_batch_workspace_bytes(lp, len(signals), C, maxpad, coef.shape[0]))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc = L.ss_iir_filtfilt_batch(_lib.ptr(x), _lib.ptr(y), lp, len(signals), C, coef.shape[0], coef.ctypes.data_as(ctypes.c_void_p), _lib.ptr(ws), nbytes, _lib.stream_of(x))
    _lib.check(rc, 'ss_iir_filtfilt_batch')
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [y[offs[u]:offs[u + 1]] for u in range(len(signals))]


def resampled_length(T, new_freq, old_freq):
    """len(np.arange(0, (T - 1) / old_freq, 1 / new_freq)): the number of samples read_emg.py:40-44 produces."""
    return len(np.arange(0, (T - 1) / old_freq, 1 / new_freq))


def subsample_batch(signals, new_freq, old_freq):
    """subsample() for a list of (T_u, C) f64 device tensors in ONE launch -> list of (T_out_u, C) views of one packed buffer."""
    dev = signals[0].device
    C = int(signals[0].shape[1])
    lens = [int(t.shape[0]) for t in signals]
    outs = [resampled_length(T, new_freq, old_freq) for T in lens]
    x = torch.cat(signals, 0).contiguous() if len(signals) > 1 else signals[0].contiguous()
    io, oo = np.concatenate([[0], np.cumsum(lens)]), np.concatenate([[0], np.cumsum(outs)])
    table = torch.from_numpy(np.stack([io[:-1], lens, oo[:-1], outs], 1).astype(np.int64)).to(dev, non_blocking=True)
    y = torch.empty(int(oo[-1]), C, dtype=torch.float64, device=dev)
    rc = _lib.lib().ss_linear_resample_batch(_lib.ptr(x), _lib.ptr(y), _lib.ptr(table), len(signals), C, float(old_freq), float(new_freq), int(oo[-1]), _lib.stream_of(x))
    _lib.check(rc, 'ss_linear_resample_batch')
    return [y[oo[u]:oo[u + 1]] for u in range(len(signals))]


def remove_drift(signal, fs):
    return filtfilt_cascade([butter_highpass_coeffs(3, 2, fs)], signal)


def notch(signal, freq, sample_frequency):
    return filtfilt_cascade([iirnotch_coeffs(freq, 30, sample_frequency)], signal)


def notch_harmonics(signal, freq, sample_frequency):
    return filtfilt_cascade([iirnotch_coeffs(freq * harmonic, 30, sample_frequency) for harmonic in range(1, 8)], signal)


def subsample(signal, new_freq, old_freq):
    x, restore = _to_device_2d(signal)
    T, C = x.shape
    times_last = (T - 1) / old_freq
    T_out = len(np.arange(0, times_last, 1 / new_freq))
    y = torch.empty(T_out, C, dtype=torch.float64, device=x.device)
    rc = _lib.lib().ss_linear_resample(_lib.ptr(x), _lib.ptr(y), T, C, float(old_freq), float(new_freq), T_out, _lib.stream_of(x))
    _lib.check(rc, 'ss_linear_resample')
    return restore(y)


def apply_to_all(function, signal_array, *args, **kwargs):
    """read_emg.py:46-50.  The device functions above already process every column of a (T, C) array in one launch, so for them this
    is a single call; any other callable is applied column by column like the reference does."""
    if function in (remove_drift, notch, notch_harmonics, subsample):
        return function(signal_array, *args, **kwargs)
    results = []
    for i in range(signal_array.shape[1]):
        results.append(function(signal_array[:, i], *args, **kwargs))
    return np.stack(results, 1)


def condition_raw_emg_recording(raw_emg, raw_emg_before=None, raw_emg_after=None):
    """The signal chain of load_utterance (read_emg.py:52-70) on in-memory recordings: context concatenation, notch harmonics, drift
    removal, context removal, resampling to 689.06 Hz (model input, `emg_orig`) and 516.79 Hz (feature rate).  Returns (emg_orig, emg)."""
    raw_emg = np.asarray(raw_emg)
    before = np.zeros([0, raw_emg.shape[1]]) if raw_emg_before is None else np.asarray(raw_emg_before)
    after = np.zeros([0, raw_emg.shape[1]]) if raw_emg_after is None else np.asarray(raw_emg_after)
    x = np.concatenate([before, raw_emg, after], 0)
    filters = [iirnotch_coeffs(60 * h, 30, 1000) for h in range(1, 8)] + [butter_highpass_coeffs(3, 2, 1000)]
    x = filtfilt_cascade(filters, x)                               # notch_harmonics then remove_drift, ONE launch sequence for all 8 channels
    x = x[before.shape[0]:x.shape[0] - after.shape[0], :]
    emg_orig = subsample(x, 689.06, 1000)
    emg = subsample(x, 516.79, 1000)
    return emg_orig, emg
