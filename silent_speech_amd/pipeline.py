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

from . import _lib
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
        batch, batch_length = [], 0
        for idx in indices:
            length = self._length(idx)
            if length is None:
                continue
            if length > self.max_len:
                logging.warning(f'Warning: example {idx} cannot fit within desired batch length')
            if length + batch_length > self.max_len:
                yield batch
                batch, batch_length = [], 0
            batch.append(idx)
            batch_length += length
        # dropping last incomplete batch

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
        batches, batch, batch_length = [], [], 0
        for idx in indices:
            length = self.lengths[idx]
            if length + batch_length > self.max_len and batch:
                batches.append(batch)
                batch, batch_length = [], 0
            batch.append(idx)
            batch_length += length
        return batches                                       # dropping last incomplete batch (:140)

    def __iter__(self):
        batches = self.all_batches()
        usable = len(batches) - len(batches) % self.world    # equal step counts on every rank
        return iter(batches[self.rank:usable:self.world])

    def __len__(self):
        return len(self.all_batches()) // self.world
