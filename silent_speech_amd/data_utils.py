"""Drop-in for the hot-path parts of the reference's data_utils.py:

    mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)   (:39-62)
    combine_fixed_length(tensor_list, length) / decollate_tensor(tensor, lengths)                      (:158-178)
    FeatureNormalizer                                                                                   (:138-156)
    phoneme_inventory                                                                                   (:17)

mel_spectrogram runs on the MI355X as two exact-f32 MFMA GEMMs: frames x windowed-DFT matrix (the
hop-strided, overlapping frames of the reflect-padded signal are just a RowMap -- nothing is copied),
then |.| (HIP kernel) and the mel filterbank GEMM with the log-clamp fused in its epilogue.
A 1024-point DFT as a dense contraction costs 2.1 MFLOP/frame -- ~13 ns at the f32 MFMA rate -- and
has no butterfly data movement, which is why it beats an FFT on this machine for n_fft = 1024.
"""
import math

import numpy as np
import torch

from . import _lib, ops

phoneme_inventory = ['aa', 'ae', 'ah', 'ao', 'aw', 'ax', 'axr', 'ay', 'b', 'ch', 'd', 'dh', 'dx', 'eh', 'el', 'em', 'en', 'er', 'ey', 'f', 'g',
                     'hh', 'hv', 'ih', 'iy', 'jh', 'k', 'l', 'm', 'n', 'nx', 'ng', 'ow', 'oy', 'p', 'r', 's', 'sh', 't', 'th', 'uh', 'uw',
                     'v', 'w', 'y', 'z', 'zh', 'sil']


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """What librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) returns with its defaults (Slaney mel
    scale, 'slaney' area normalisation, float32) -- the call at data_utils.py:47.  librosa is a
    third-party dependency absent from the reference tree: basis values are 'parity unpinned'."""
    if fmax is None:
        fmax = sr / 2.0
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), m * f_sp)

    n_bins = 1 + n_fft // 2
    freqs = np.linspace(0.0, sr / 2.0, n_bins)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    fdiff = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, None]
    return w.astype(np.float32)


_dft_cache = {}
_mel_cache = {}


def _settle_cache(device):
    """A cached constant (DFT matrix, mel basis, FFT tables) is uploaded on whichever stream misses the cache first and read from every stream after that
    (pipeline.DeviceBatchBuilder runs the audio half of a batch on a side stream): the upload is followed by ONE device-wide synchronisation, so that no
    later reader on another stream can start before the bytes are there (round-5 advisor finding; a cache miss happens once per configuration)."""
    if getattr(device, 'type', str(device)) == 'cuda' and torch.cuda.is_available():
        torch.cuda.synchronize(device)


def _windowed_dft(n_fft, win_size, device):
    """[2*nb][n_fft] f32: rows 0..nb-1 = hann(n) cos(2 pi k n / N), rows nb..2nb-1 = hann(n) sin(.)   (nb = N/2+1)"""
    key = (n_fft, win_size, str(device))
    if key not in _dft_cache:
        nb = n_fft // 2 + 1
        n = np.arange(n_fft, dtype=np.float64)
        win = np.zeros(n_fft, dtype=np.float64)
        off = (n_fft - win_size) // 2
        win[off:off + win_size] = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_size) / win_size)).astype(np.float32)   # torch.hann_window (periodic), f32
        ang = 2.0 * np.pi * np.outer(np.arange(nb), n) / n_fft
        mat = np.concatenate([np.cos(ang) * win[None, :], np.sin(ang) * win[None, :]], 0).astype(np.float32)
        _dft_cache[key] = torch.from_numpy(mat).to(device)
        _settle_cache(device)
    return _dft_cache[key]


def _mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax, device, ld):
    key = (sampling_rate, n_fft, num_mels, fmin, fmax, str(device), ld)
    if key not in _mel_cache:
        b = slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        pad = np.zeros((num_mels, ld), dtype=np.float32)
        pad[:, :b.shape[1]] = b
        _mel_cache[key] = torch.from_numpy(pad).to(device)
        _settle_cache(device)
    return _mel_cache[key]


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """y: (B, L) float32 in [-1, 1] on the GPU -> (B, num_mels, F) float32 log-mel, F = 1 + (L + n_fft - hop - n_fft)//hop.
    (The reference's min/max range warnings (:40-43) would force a device sync and are omitted.)  Dispatches through
    torch.ops.silent_speech.stft_logmel (torch_ops.py)."""
    if y.dim() != 2 or y.dtype != torch.float32:
        raise ValueError('y must be a float32 (B, L) tensor')
    if hop_size % 4 or n_fft % 4:
        raise ValueError('hop_size and n_fft must be multiples of 4 (16-byte f32 rows)')
    from . import torch_ops  # noqa: F401
    return torch.ops.silent_speech.stft_logmel(y, int(n_fft), int(num_mels), int(sampling_rate), int(hop_size), int(win_size), int(fmin), int(fmax), bool(center))


def _mel_spectrogram_impl(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center):
    y = y.contiguous()
    B, L = y.shape
    pad = int((n_fft - hop_size) / 2)
    Lp = L + 2 * pad
    dev = y.device
    if not center and y.dtype == torch.float32 and L > pad and Lp >= n_fft:
        # n_fft = 1024: one kernel straight from y (reflection is index arithmetic in the frames that touch the ends): no padded copy
        F = 1 + (Lp - n_fft) // hop_size
        out = torch.empty(B, num_mels, F, dtype=torch.float32, device=dev)
        if _logmel_fft(y, None, None, L, B, F, pad, False, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major=False):
            return out
    ldp = (Lp + 3) // 4 * 4
    ypad = torch.zeros(B, ldp, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ss_reflect_pad(_lib.ptr(y), _lib.ptr(ypad), B, L, pad, ldp, _lib.stream_of(y)), 'ss_reflect_pad')
    if center:          # torch.stft(center=True, pad_mode='reflect') (data_utils.py:54): n_fft//2 more samples, reflected off the PADDED signal
        pad2 = n_fft // 2
        if pad2 >= Lp:
            raise ValueError('signal too short for centred frames')
        Lp2 = Lp + 2 * pad2
        ldp2 = (Lp2 + 3) // 4 * 4
        ypad2 = torch.zeros(B, ldp2, dtype=torch.float32, device=dev)
        src = ypad if ldp == Lp else ypad[:, :Lp].contiguous()
        _lib.check(_lib.lib().ss_reflect_pad(_lib.ptr(src), _lib.ptr(ypad2), B, Lp, pad2, ldp2, _lib.stream_of(y)), 'ss_reflect_pad')
        ypad, Lp, ldp = ypad2, Lp2, ldp2
    if Lp < n_fft:
        raise ValueError('signal too short for one frame')
    F = 1 + (Lp - n_fft) // hop_size
    out = torch.empty(B, num_mels, F, dtype=torch.float32, device=dev)
    if not _logmel_fft(ypad, None, None, ldp, B, F, 0, False, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major=False):    # (center=True: rows already padded twice)
        _stft_logmel(ypad, B, F, ldp, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major=False)
    return out


_fft_cache = {}


def _fft_tables(n_fft, win_size, num_mels, sampling_rate, fmin, fmax, device):
    """What ss_stft_logmel_fft reads besides the signal: the f32 hann window (torch.hann_window, periodic, centred in n_fft like torch.stft does) and
    the mel filterbank in sparse form -- per band the first non-zero bin, the length of the run up to the last non-zero bin (padded to a multiple of 4
    with zero weights) and the packed weights --, and the deal of the bands to the 64 lanes of a wave (longest band first onto the least loaded lane,
    two bands per lane at most).  None when the filterbank does not fit the kernel's tables."""
    key = (n_fft, win_size, num_mels, sampling_rate, fmin, fmax, str(device))
    if key not in _fft_cache:
        win = np.zeros(n_fft, dtype=np.float32)
        off = (n_fft - win_size) // 2
        win[off:off + win_size] = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_size) / win_size)).astype(np.float32)
        basis = slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        nb = n_fft // 2 + 1
        lo, cnt, offs, ws = [], [], [], []
        for m in range(num_mels):
            nz = np.nonzero(basis[m])[0]
            a, n = (int(nz[0]), int(nz[-1]) - int(nz[0]) + 1) if len(nz) else (0, 0)
            n4 = (n + 3) // 4 * 4
            run = np.zeros(n4, dtype=np.float32)
            run[:n] = basis[m, a:a + n]
            lo.append(a); cnt.append(n4); offs.append(sum(len(w) for w in ws)); ws.append(run)
        n_w = int(sum(cnt))
        load, lanes = [0] * 64, [[] for _ in range(64)]
        for m in sorted(range(num_mels), key=lambda m: -cnt[m]):
            l = min((l for l in range(64) if len(lanes[l]) < 2), key=lambda l: load[l], default=None)
            if l is None:
                break
            lanes[l].append(m); load[l] += cnt[m]
        ok = num_mels <= 128 and n_w <= 4096 and sum(len(x) for x in lanes) == num_mels and all(lo[m] + cnt[m] <= nb + 7 for m in range(num_mels))
        if not ok:
            _fft_cache[key] = None
        else:
            lb = np.full((2, 64), -1, dtype=np.int32)
            for l in range(64):
                for it, m in enumerate(lanes[l]):
                    lb[it, l] = m
            w = np.concatenate(ws) if n_w else np.zeros(1, dtype=np.float32)
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dt))).to(device)
            _fft_cache[key] = (t(win, np.float32), t(lo, np.int32), t(cnt, np.int32), t(offs, np.int32), t(w, np.float32), t(lb, np.int32), n_w)
            _settle_cache(device)
    return _fft_cache[key]


def _logmel_fft(y, offs, lens, uniform_len, B, F, pad, clip, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major):
    """One ss_stft_logmel_fft launch (csrc/mel.hip) on the caller's signals; False when this configuration needs the GEMM formulation."""
    if n_fft != 1024 or win_size > n_fft:
        return False
    tabs = _fft_tables(n_fft, win_size, num_mels, sampling_rate, fmin, fmax, y.device)
    if tabs is None:
        return False
    win, lo, cnt, boff, w, lb, n_w = tabs
    sb, sf, sm = (F * num_mels, num_mels, 1) if frame_major else (num_mels * F, 1, F)
    _lib.check(_lib.lib().ss_stft_logmel_fft(_lib.ptr(y), _lib.ptr(offs) if offs is not None else None, _lib.ptr(lens) if lens is not None else None, int(uniform_len), B, F,
                                             pad, int(bool(clip)), n_fft, hop_size, _lib.ptr(win), _lib.ptr(lo), _lib.ptr(cnt), _lib.ptr(boff), _lib.ptr(w), _lib.ptr(lb),
                                             num_mels, n_w, 1e-5, _lib.ptr(out), sb, sf, sm, _lib.stream_of(y)), 'ss_stft_logmel_fft')
    return True


def _stft_logmel(ypad, B, F, ldp, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major):
    """padded signals [B][ldp] -> log-mel of frames 0..F-1 of every row: the GEMM formulation (any n_fft): two f32 GEMMs (windowed DFT matrix, mel basis)
    around the magnitude kernel.  frame_major: out is [B*F][num_mels] (frames of a signal are contiguous rows) instead of the reference's (B, num_mels, F)."""
    dev = ypad.device
    nb = n_fft // 2 + 1
    W = _windowed_dft(n_fft, win_size, dev)
    ld_spec = (2 * nb + 7) // 8 * 8
    spec = torch.empty(B * F, ld_spec, dtype=torch.float32, device=dev)
    ops.gemm(ypad, W, spec, B * F, 2 * nb, n_fft, ops.rowmap(hop_size, F, ldp), ops.rowmap(n_fft), ops.rowmap(ld_spec))
    ld_mag = (nb + 7) // 8 * 8
    mag = torch.empty(B * F, ld_mag, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ss_stft_magnitude(_lib.ptr(spec), ld_spec, nb, _lib.ptr(mag), ld_mag, B * F, _lib.stream_of(ypad)), 'ss_stft_magnitude')
    basis = _mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax, dev, ld_mag)
    if frame_major:
        ops.gemm_ex(mag, basis, out, B * F, num_mels, ld_mag, ops.rowmap(ld_mag), ops.rowmap(ld_mag), ops.rowmap(num_mels), log_clamp=1e-5)
    else:
        ops.gemm_ex(mag, basis, out, B * F, num_mels, ld_mag, ops.rowmap(ld_mag), ops.rowmap(ld_mag), ops.rowmap(1, F, num_mels * F),
                    col_perm=(num_mels, F, 0), log_clamp=1e-5)
    return out


def mel_spectrogram_batch(signals, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=8000, clip=True):
    """load_audio's `np.clip` + `mel_spectrogram(..., center=False)` (data_utils.py:76-78) for EVERY utterance of a batch in one
    launch sequence: one ragged clip + reflect-pad kernel, ONE hop-strided DFT GEMM over all frames of all utterances (rows of the
    padded buffer = utterances, RowMap batch stride), magnitude, one mel GEMM with the log-clamp epilogue.
    signals: list of 1-D f32 device tensors at `sampling_rate`, or the pair (flat f32 device tensor holding them back to back, lengths) -- what
    a loader that uploaded the whole batch in one copy has.  Returns (buf [B][F_max][num_mels] f32, frames per utterance):
    utterance b's `pytorch_mspec.squeeze(0).T` is buf[b, :frames[b]] -- a contiguous view, no per-utterance copy."""
    if hop_size % 4 or n_fft % 4:
        raise ValueError('hop_size and n_fft must be multiples of 4 (16-byte f32 rows)')
    packed = isinstance(signals, tuple)
    lens = [int(n) for n in signals[1]] if packed else [int(t.shape[0]) for t in signals]
    B = len(lens)
    pad = int((n_fft - hop_size) / 2)
    if min(lens) + 2 * pad < n_fft or min(lens) <= pad:
        raise ValueError('signal too short for one frame')
    dev = signals[0].device
    if packed:
        flat = signals[0].reshape(-1)
        if flat.dtype != torch.float32 or int(flat.shape[0]) != sum(lens):
            raise ValueError('packed signals: a flat float32 tensor of sum(lengths) samples')
    else:
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in signals]) if B > 1 else signals[0].reshape(-1).to(torch.float32).contiguous()
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    frames = [1 + (L + 2 * pad - n_fft) // hop_size for L in lens]
    F = max(frames)
    ldp = (max(lens) + 2 * pad + 3) // 4 * 4
    offs_d = torch.from_numpy(offs).to(dev, non_blocking=True)
    lens_d = torch.from_numpy(np.asarray(lens, dtype=np.int32)).to(dev, non_blocking=True)
    out = torch.empty(B, F, num_mels, dtype=torch.float32, device=dev)
    if _logmel_fft(flat, offs_d, lens_d, 0, B, F, pad, clip, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major=True):
        return out, frames
    ypad = torch.empty(B, ldp, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().ss_reflect_pad_ragged(_lib.ptr(flat), _lib.ptr(offs_d), _lib.ptr(lens_d), _lib.ptr(ypad), B, min(lens), pad, ldp, int(bool(clip)),
                                                _lib.stream_of(flat)), 'ss_reflect_pad_ragged')
    _stft_logmel(ypad, B, F, ldp, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, out, frame_major=True)
    return out, frames


class FeatureNormalizer(object):
    """data_utils.py:138-156 (normalize mutates its argument in place, like the reference)."""

    def __init__(self, feature_samples, share_scale=False):
        feature_samples = np.concatenate(feature_samples, axis=0)
        self.feature_means = feature_samples.mean(axis=0, keepdims=True)
        self.feature_stddevs = feature_samples.std() if share_scale else feature_samples.std(axis=0, keepdims=True)

    def normalize(self, sample):
        sample -= self.feature_means
        sample /= self.feature_stddevs
        return sample

    def inverse(self, sample):
        return sample * self.feature_stddevs + self.feature_means


class PackJob(object):
    """One combine_fixed_length (or plain concatenation, length=None) of device tensors as data: the [pointers | cumulative byte
    offsets] table the gather kernel (csrc/optim.hip `ss_concat_pad`) reads, and the launch over it.  Building the table is host
    arithmetic on ~40 (pointer, size) pairs; it travels to the device through staging.upload -- alone (combine_fixed_length) or
    together with every other table of the batch (transduction_model.prepare_batch: ONE copy per batch)."""

    def __init__(self, tensor_list, length=None):
        first = tensor_list[0]
        trailing = tuple(first.shape[1:])
        ts = [t if t.is_contiguous() else t.contiguous() for t in tensor_list]
        assert all(t.dtype == first.dtype and tuple(t.shape[1:]) == trailing for t in ts), 'utterances of one batch share dtype and feature shape'
        frames = sum(int(t.shape[0]) for t in ts)
        row_bytes = first.element_size()
        for d in trailing:
            row_bytes *= int(d)
        if length is None:
            self.out_shape = (frames,) + trailing
            total = frames * row_bytes
        else:
            rows = (frames + length - 1) // length
            self.out_shape = (rows, length) + trailing
            total = rows * length * row_bytes
        n = len(ts)
        table = np.empty(2 * n + 1, dtype=np.int64)
        table[:n] = np.fromiter((t.data_ptr() for t in ts), dtype=np.uint64, count=n).view(np.int64)
        sizes = np.fromiter((int(t.shape[0]) for t in ts), dtype=np.int64, count=n) * row_bytes
        table[n] = 0
        np.cumsum(sizes, out=table[n + 1:])
        gran = 16
        probe = int(np.bitwise_or.reduce(table[:n])) | int(np.bitwise_or.reduce(sizes)) | total
        while gran > 1 and probe % gran:
            gran //= 2
        self.gran = 1 if gran == 2 else gran
        self.table, self.n, self.total, self.keep = table, n, total, ts
        self.dtype, self.device = first.dtype, first.device

    def launch(self, table_dev):
        out = torch.empty(self.out_shape, dtype=self.dtype, device=self.device)
        _lib.check(_lib.lib().ss_concat_pad(_lib.ptr(table_dev), self.n, _lib.ptr(out), self.total, self.gran, _lib.stream_of(out)), 'ss_concat_pad')
        return out


def _host_pack(tensor_list, length, pin=False):
    first = tensor_list[0]
    trailing = tuple(first.shape[1:])
    frames = sum(int(t.shape[0]) for t in tensor_list)
    rows = frames if length is None else (frames + length - 1) // length * length
    out = torch.empty((rows,) + trailing, dtype=first.dtype, pin_memory=pin)
    if frames:
        torch.cat(list(tensor_list), 0, out=out[:frames])
    out[frames:].zero_()
    return out if length is None else out.view((rows // length, length) + trailing)


def combine_fixed_length(tensor_list, length):
    """data_utils.py:158-167: the utterances back to back along time, zero-padded to a whole number of rows of `length` frames,
    viewed as (rows, length, ...).  Device tensors are packed by ONE gather launch over an offset table (csrc/optim.hip
    `ss_concat_pad`; the table is rebuilt for every call -- a training loop hands over new tensors every step -- and uploaded
    from pinned memory).  Host tensors (staging code, tests) are packed with a host concatenation."""
    first = tensor_list[0]
    on_device = _lib.kernels_can_read(first)
    if not on_device:
        return _host_pack(tensor_list, length)
    from . import staging
    job = PackJob(tensor_list, length)
    table, = staging.upload([job.table], first.device)
    return job.launch(table)


def decollate_tensor(tensor, lengths):
    """data_utils.py:169-178."""
    b, s, d = tensor.size()
    tensor = tensor.reshape(b * s, d)
    results, idx = [], 0
    for length in lengths:
        assert idx + length <= b * s
        results.append(tensor[idx:idx + length])
        idx += length
    return results
