"""Device-side input pipeline ("next" row N3 of SURVEY section 8f): what the reference's EMGDataset.__getitem__ /
load_utterance / load_audio do on the CPU per utterance (and lru_cache), done on the MI355X per batch.

    condition_raw_emg(raw)                  read_emg.py:227-228   raw / 20, then 50 * tanh(. / 50)
    normalize_features(x, normalizer, 8.0)  read_emg.py:231-233   FeatureNormalizer.normalize (+ 8 * tanh(. / 8) for EMG features)
    mel_targets(audio, normalizer, ...)     data_utils.py:64-83 + read_emg.py:231   clip -> mel_spectrogram (HIP DFT/mel GEMMs) -> (T, 80) -> normalise
    ShardedSizeAwareSampler                 read_emg.py:115-140   greedy length-budget batches, dealt round-robin to the DP ranks

File decoding, resampling of 16 kHz audio and the offline EMG filters (read_emg.py:27-71) stay on the host (row N4 /
out of scope); everything here starts from arrays already in memory.  All functions keep the data on the device and
enqueue on the current stream; nothing synchronises.
"""
import random

import numpy as np
import torch

from . import _lib, staging
from .data_utils import mel_spectrogram

_L = _lib.lib
_p = _lib.ptr


def _soft_clip(x, out, C, mean, std, pre_div, limit):
    rc = _L().ss_soft_clip(_p(x), _p(out), x.numel(), int(C), _p(mean) if mean is not None else None, _p(std) if std is not None else None,
                           float(pre_div), float(limit), _lib.stream_of(x))
    _lib.check(rc, 'ss_soft_clip')
    return out


def condition_raw_emg(raw):
    """(T0, 8) f32 raw EMG at 689.06 Hz, as stored by load_utterance -> the model's x_raw (read_emg.py:227-228)."""
    raw = raw.contiguous().float()
    return _soft_clip(raw, torch.empty_like(raw), raw.shape[-1], None, None, 20.0, 50.0)


def _normalizer_tensors(normalizer, C, device):
    mean = torch.as_tensor(np.broadcast_to(np.asarray(normalizer.feature_means, dtype=np.float32).reshape(-1), (C,)).copy(), device=device)
    std = torch.as_tensor(np.broadcast_to(np.asarray(normalizer.feature_stddevs, dtype=np.float32).reshape(-1), (C,)).copy(), device=device)
    return mean, std


def normalize_features(x, normalizer, limit=0.0):
    """FeatureNormalizer.normalize on the device (data_utils.py:228-231), optionally followed by limit * tanh(. / limit)."""
    x = x.contiguous().float()
    mean, std = _normalizer_tensors(normalizer, x.shape[-1], x.device)
    return _soft_clip(x, torch.empty_like(x), x.shape[-1], mean, std, 1.0, limit)


def mel_targets(audio, normalizer=None, max_frames=None):
    """audio: 1-D f32 waveform at 22 050 Hz on the device (after the host-side decode / resample of load_audio,
    data_utils.py:64-76) -> (frames, 80) normalised log-mel targets (data_utils.py:77-83, read_emg.py:231)."""
    y = audio.float().clamp(-1, 1).unsqueeze(0)                                            # :76
    mspec = mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, 8000, center=False).squeeze(0).t().contiguous()   # :77-78
    if max_frames is not None and mspec.shape[0] > max_frames:
        mspec = mspec[:max_frames].contiguous()                                            # :79-80
    return normalize_features(mspec, normalizer) if normalizer is not None else mspec


def greedy_length_batches(order, length_of, max_len, yield_empty=False, on_oversize=None):
    """read_emg.py:124-140, the ONE implementation of the reference's batching rule: walk `order`, keep adding utterances while the
    running sum of their lengths (raw 1 kHz EMG samples) stays within `max_len`; an utterance that would overflow closes the batch
    and opens the next one; the incomplete last batch is dropped.  length_of(idx) -> int, or None for utterances the reference
    skips (:129-130).  yield_empty reproduces the reference's quirk of emitting an empty batch when the very first utterance of a
    batch is already over budget (:135-137)."""
    batch, batch_length = [], 0
    for idx in order:
        length = length_of(idx)
        if length is None:
            continue
        if length > max_len and on_oversize is not None:
            on_oversize(idx)
        if length + batch_length > max_len and (batch or yield_empty):
            yield batch
            batch, batch_length = [], 0
        batch.append(idx)
        batch_length += length
    # dropping last incomplete batch


class SizeAwareSampler(torch.utils.data.Sampler):
    """Drop-in for read_emg.py:115-140 -- same constructor (emg_dataset, max_len), same greedy packing under a budget of raw 1 kHz
    EMG samples, reshuffled on every __iter__ with Python's `random`, text-less utterances skipped (:129-130), the incomplete
    last batch dropped (:140).  Works on the reference's EMGDataset protocol (example_indices -> <idx>_info.json with 'text' and
    'chunks') and on datasets that answer example_length(idx) directly (synthetic.SyntheticEMGDataset; None = skip).
    rank / world / seed (keyword-only extras): data-parallel training deals batch i to rank i % world from a shuffle that every
    rank derives from the shared seed and epoch (equal step counts, no communication)."""

    def __init__(self, emg_dataset, max_len, *, rank=0, world=1, seed=None):
        self.dataset, self.max_len, self.rank, self.world, self.seed, self.epoch = emg_dataset, max_len, rank, world, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _length(self, idx):
        ds = self.dataset
        if hasattr(ds, 'example_length'):
            return ds.example_length(idx)
        import json
        import os
        import string
        directory_info, file_idx = ds.example_indices[idx]
        with open(os.path.join(directory_info.directory, f'{file_idx}_info.json')) as f:
            info = json.load(f)
        if not np.any([l in string.ascii_letters for l in info['text']]):
            return None
        return sum([emg_len for emg_len, _, _ in info['chunks']])

    def _batches(self):
        import logging
        indices = list(range(len(self.dataset)))
        if self.world > 1 or self.seed is not None:
            random.Random((self.seed or 0) * 1000003 + self.epoch).shuffle(indices)
        else:
            random.shuffle(indices)                         # read_emg.py:122
        return greedy_length_batches(indices, self._length, self.max_len, yield_empty=True,
                                     on_oversize=lambda idx: logging.warning(f'Warning: example {idx} cannot fit within desired batch length'))

    def __iter__(self):
        if self.world == 1:
            return self._batches()
        batches = [b for b in self._batches() if b]
        usable = len(batches) - len(batches) % self.world    # equal step counts on every rank
        return iter(batches[self.rank:usable:self.world])


class ShardedSizeAwareSampler(torch.utils.data.Sampler):
    """read_emg.py:115-140 for data-parallel training: the same greedy packing under a budget of 1 kHz samples (shuffled
    order, the incomplete last batch dropped), with batch i going to rank i % world.  Every rank builds the identical list
    from the shared seed + epoch, so no communication is needed and all ranks see the same number of batches."""

    def __init__(self, lengths, max_len, rank=0, world=1, seed=0, shuffle=True):
        self.lengths, self.max_len, self.rank, self.world, self.seed, self.shuffle = list(lengths), max_len, rank, world, seed, shuffle
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def all_batches(self):
        indices = list(range(len(self.lengths)))
        if self.shuffle:
            random.Random(self.seed * 1000003 + self.epoch).shuffle(indices)
        return list(greedy_length_batches(indices, self.lengths.__getitem__, self.max_len))

    def __iter__(self):
        batches = self.all_batches()
        usable = len(batches) - len(batches) % self.world    # equal step counts on every rank
        return iter(batches[self.rank:usable:self.world])

    def __len__(self):
        return len(self.all_batches()) // self.world


# ------------------------------------------------------------------ the per-batch device loader (N3 composed with N4)
class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _feature_frames(n_1k):
    """Frames of get_emg_features (data_utils.py:100: librosa.util.frame(frame_length=16, hop_length=6)) on the 516.79 Hz resample
    (read_emg.py:71) of a recording of n_1k raw 1 kHz samples -- the count load_utterance truncates everything to (:82-88)."""
    n516 = len(np.arange(0, (n_1k - 1) / 1000.0, 1 / 516.79))
    return 0 if n516 < 16 else 1 + (n516 - 16) // 6


class DeviceBatchBuilder(object):
    """`[dataset[i] for i in batch]` + `collate_raw` (read_emg.py:223-296) on the MI355X, from in-memory recordings: what
    load_utterance / load_audio / EMGDataset.__getitem__ compute per utterance on the host (and memoise with lru_cache) is computed
    here per BATCH on the device, and the result is the batch dict the training step consumes (lists of per-utterance device
    tensors), so `train_model` / `dtw_loss` / `bench.py` run from raw recordings.

    A recording is a dict:
        raw_emg (n, 8)              raw 1 kHz samples of `<idx>_emg.npy`          [+ raw_emg_before / raw_emg_after: the neighbours
                                    load_utterance concatenates as filter context, read_emg.py:55-66]
        audio (L,)                  waveform at 22 050 Hz, i.e. after load_audio's host-side decode / resample (data_utils.py:65-75)
        phonemes (frames,) int64    read_phonemes output (optional: 'sil' everywhere like read_emg.py:98)
        silent bool, text_int int64, session_index int
        parallel                    for silent recordings: the voiced twin (same dict shape); its audio features and phonemes become
                                    the targets (read_emg.py:242-256, collate_raw :268-271)
    Stages (each ONE launch sequence for the whole batch unless noted):
        EMG   : context concat -> 7 notch harmonics + 2 Hz high-pass, zero-phase, f64 (csrc/filters.hip, ragged batch: work items are
                (chunk, channel) pairs over all recordings) -> crop -> np.interp to 689.06 Hz -> rows 8..8+8n (:90) gathered by one
                launch -> /20, 50 tanh(./50) (:227-228) in one kernel
        audio : clip + reflect pad (ragged) -> one hop-strided DFT GEMM -> |.| -> one mel GEMM + log clamp -> FeatureNormalizer, all
                utterances at once (data_utils.mel_spectrogram_batch); truncation to n frames is a view
    The 112-d hand-crafted EMG features (`emg`, data_utils.py:92-136) are not inputs of the model (architecture.py:61 ignores
    x_feat) and are emitted as zeros of the right shape unless the recording brings `emg_features`."""

    def __init__(self, device, mfcc_norm=None, emg_norm=None, limit_length=False, sil_index=0, remove_channels=()):
        """remove_channels: the reference's FLAGS.remove_channels (read_emg.py:73-75): those raw-EMG columns are zeroed after the
        filtering / resampling and before the soft clip, exactly where load_utterance does it."""
        self.device, self.mfcc_norm, self.emg_norm, self.limit_length, self.sil_index = torch.device(device), mfcc_norm, emg_norm, limit_length, sil_index
        self.remove_channels = tuple(int(c) for c in remove_channels)

    # ---- host arrays of a batch in ONE pinned copy (round 5: the leg was host-bound -- 123 small uploads and ~100 glue launches per batch)
    @staticmethod
    def _host(x, dtype):
        """x as a contiguous numpy array of `dtype`, or None when it lives on the device already (then the per-recording path is taken)."""
        if torch.is_tensor(x):
            if x.is_cuda:
                return None
            x = x.detach().numpy()
        return np.ascontiguousarray(np.asarray(x), dtype=dtype)

    # ---- EMG: every recording of the batch through ONE filter / resample launch sequence
    def _filtered_689(self, recordings, packed=None):
        from .read_emg import butter_highpass_coeffs, filtfilt_cascade_batch, iirnotch_coeffs, subsample_batch
        dev = self.device
        sigs, cuts = [], []
        for rec in recordings:
            parts = [rec.get('raw_emg_before'), rec['raw_emg'], rec.get('raw_emg_after')]
            cuts.append((0 if parts[0] is None else len(parts[0]), 0 if parts[2] is None else len(parts[2])))
            if packed is None:
                ts = [torch.as_tensor(np.asarray(p) if not torch.is_tensor(p) else p).to(device=dev, dtype=torch.float64) for p in parts if p is not None and len(p)]
                sigs.append(torch.cat(ts, 0) if len(ts) > 1 else ts[0])                    # read_emg.py:66
        filters = getattr(DeviceBatchBuilder, '_filters', None)
        if filters is None:                                                                 # the same eight sections for every batch
            filters = DeviceBatchBuilder._filters = [iirnotch_coeffs(60 * h, 30, 1000) for h in range(1, 8)] + [butter_highpass_coeffs(3, 2, 1000)]
        ys = filtfilt_cascade_batch(filters, sigs if packed is None else packed)           # :67-68
        ys = [y[nb:y.shape[0] - na] for y, (nb, na) in zip(ys, cuts)]                      # :69
        return subsample_batch(ys, 689.06, 1000)                                           # :70

    def _frames(self, rec, mel_frames, limit):
        n = min(_feature_frames(len(rec['raw_emg'])), mel_frames)
        return min(n, 800) if limit else n

    def build(self, recordings):
        from .data_utils import mel_spectrogram_batch
        dev = self.device
        # audio of every recording whose mel frames are needed: own audio (lengths; targets when voiced) and the voiced twins
        twins = [r['parallel'] if r['silent'] else None for r in recordings]
        audio_src = list(recordings) + [t for t in twins if t is not None]
        # everything that arrives in host memory crosses PCIe in one pinned copy per kind (staging.upload).  The raw EMG goes first, on the main
        # stream, with its filter chain right behind it (~50 short, latency-bound launches: the critical path of a build); the audio is then
        # copied into its pinned buffer WHILE that chain runs, uploaded on a SIDE stream, and its half (clip, reflect pad, DFT / mel GEMMs,
        # normalise) runs there under the EMG half: the two are independent until the batch dict is assembled
        a_host = [self._host(r['audio'], np.float32) for r in audio_src]
        e_host = [[self._host(p, np.float64) for p in (r.get('raw_emg_before'), r['raw_emg'], r.get('raw_emg_after')) if p is not None and len(p)] for r in recordings]
        emg_packed = None
        if all(p is not None for ps in e_host for p in ps):
            e_rows = [sum(int(p.shape[0]) for p in ps) for ps in e_host]
            emg_packed = (staging.upload([[p.reshape(p.shape[0], -1) for ps in e_host for p in ps]], dev)[0], e_rows)
        e689 = self._filtered_689(recordings, emg_packed)
        side = None
        if dev.type == 'cuda':
            side = getattr(self, '_side', None)
            if side is None:
                side = self._side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
        with (torch.cuda.stream(side) if side is not None else _NullCtx()):
            if all(a is not None for a in a_host):
                sig = (staging.upload([[a.reshape(-1) for a in a_host]], dev)[0], [int(a.shape[0]) for a in a_host])
            else:
                sig = [torch.as_tensor(np.asarray(r['audio']) if not torch.is_tensor(r['audio']) else r['audio']).to(device=dev, dtype=torch.float32) for r in audio_src]
                if side is not None:
                    side.wait_stream(main)                                                  # device tensors handed in: produced on the caller's stream
                    for t in sig:
                        t.record_stream(side)
            mel, mframes = mel_spectrogram_batch(sig)
            if self.mfcc_norm is not None:
                mean, std = _normalizer_tensors(self.mfcc_norm, mel.shape[-1], dev)
                _soft_clip(mel, mel, mel.shape[-1], mean, std, 1.0, 0.0)                   # FeatureNormalizer.normalize (read_emg.py:231), in place
        if side is not None:
            mel_done = torch.cuda.Event()
            mel_done.record(side)
            mel.record_stream(main)
        n_own = [self._frames(r, mframes[i], self.limit_length) for i, r in enumerate(recordings)]
        # raw EMG (filtered above): rows 8 .. 8 + 8 n of every recording gathered into ONE buffer, soft-clipped in one launch
        # a resampled signal that ends before row 8 + 8 n: the reference's slice raw_emg[8:8+8n] (read_emg.py:90) silently comes out shorter;
        # here the frame count of that recording follows the samples that exist (whole frames), so every downstream length stays consistent
        n_own = [min(n, max(0, (int(e.shape[0]) - 8) // 8)) for e, n in zip(e689, n_own)]
        # rows 8 .. 8 + 8 n of every recording (read_emg.py:90), as f32 (:100), gathered into ONE buffer by one launch
        e32 = e689[0]._base.to(torch.float32) if e689[0]._base is not None else torch.cat(e689, 0).to(torch.float32)
        offs = np.concatenate([[0], np.cumsum([int(e.shape[0]) for e in e689])])
        from .data_utils import combine_fixed_length
        raw = combine_fixed_length([e32[offs[u] + 8:offs[u] + 8 + 8 * n] for u, n in enumerate(n_own)], 1).view(-1, 8)
        off, raw_views = 0, []
        for n in n_own:
            raw_views.append(raw[off:off + 8 * n])
            off += 8 * n
        for ch in self.remove_channels:                                                     # read_emg.py:73-75 (on the model-rate signal; zero stays zero through the clip)
            raw[:, ch].zero_()
        _soft_clip(raw, raw, 8, None, None, 20.0, 50.0)                                     # read_emg.py:227-228
        if side is not None:
            main.wait_event(mel_done)                                                        # the mel targets join the batch here
        out = {k: [] for k in ('audio_features', 'audio_feature_lengths', 'emg', 'raw_emg', 'parallel_voiced_emg', 'phonemes', 'session_ids',
                               'lengths', 'silent', 'text_int', 'text_int_lengths')}
        # pass 1 (host): target lengths and phoneme labels; the labels and session ids of the whole batch then travel in one copy, and the
        # (unused, zero) hand-crafted feature tensors are views of ONE zero buffer -- instead of three small launches per utterance
        ti = len(recordings)
        meta, ph_host, sess_host = [], [], []
        for i, (r, n) in enumerate(zip(recordings, n_own)):
            if r['silent']:
                t = r['parallel']
                nt = self._frames(t, mframes[ti], False)                                    # load_utterance(voiced ..., limit_length=False), :245
                feats, ph_src = mel[ti, :nt], t
                ti += 1
            else:
                nt, feats, ph_src = n, mel[i, :n], r
            ph = ph_src.get('phonemes')
            ph = np.full((nt,), self.sil_index, dtype=np.int64) if ph is None else (ph.detach().cpu().numpy() if torch.is_tensor(ph) else np.asarray(ph)).astype(np.int64)[:nt]
            if ph.shape[0] != nt:
                raise ValueError('phoneme labels shorter than the %d target frames' % nt)
            meta.append((nt, feats))
            ph_host.append(ph)
            sess_host.append(np.full((n,), int(r.get('session_index', 0)), dtype=np.int64))
        ph_all, sess_all = staging.upload([ph_host, sess_host], dev)
        zeros = torch.zeros(sum(n_own), 112, dtype=torch.float32, device=dev)
        po = so = 0
        for i, (r, n) in enumerate(zip(recordings, n_own)):
            nt, feats = meta[i]
            ef = r.get('emg_features')
            if ef is None:
                emg = zeros[so:so + n]
            else:
                emg = torch.as_tensor(ef, dtype=torch.float32)[:n].to(dev)
                emg = normalize_features(emg, self.emg_norm, 8.0) if self.emg_norm is not None else emg
            text = torch.as_tensor(r.get('text_int', np.zeros(0, dtype=np.int64)), dtype=torch.int64)
            out['audio_features'].append(feats); out['audio_feature_lengths'].append(nt)
            out['emg'].append(emg); out['raw_emg'].append(raw_views[i])
            # read_emg.py:250-257: for a silent example the voiced twin's (normalised, soft-clipped) EMG features; np.zeros(1) otherwise.  The
            # 112-d hand-crafted features are not computed here (not model inputs): the twin's are passed through when the recording brings them.
            pv = np.zeros(1)
            if r['silent'] and r['parallel'].get('emg_features') is not None:
                pv = torch.as_tensor(r['parallel']['emg_features'], dtype=torch.float32).to(dev)
                pv = normalize_features(pv, self.emg_norm, 8.0) if self.emg_norm is not None else pv
            out['parallel_voiced_emg'].append(pv)
            out['phonemes'].append(ph_all[po:po + nt]); out['session_ids'].append(sess_all[so:so + n])
            po += nt; so += n
            out['lengths'].append(n); out['silent'].append(bool(r['silent']))
            out['text_int'].append(text); out['text_int_lengths'].append(int(text.shape[0]))
        return out
