"""Synthetic stand-in for the reference's EMGDataset (read_emg.py:143-296): emits exactly the batch
dict the hot path consumes -- there is no network or dataset in the build/bench
environment.  Shapes, dtypes and value ranges follow read_emg.py:227-235 and collate_raw (:261-296)."""
import numpy as np
import torch

RAW_PER_FRAME = 8
HZ_RATIO = 0.68906            # 689.06 Hz model-rate raw EMG vs the 1 kHz samples the sampler budgets (read_emg.py:70,131)


def make_utterance(rng, T, silent, n_mel=80, n_phone=48, n_feat=112, session=0):
    z = rng.standard_normal((T * RAW_PER_FRAME, 8)).astype(np.float32) * 5.0
    raw = 50.0 * np.tanh(z / 50.0)                                              # read_emg.py:227-228
    emg = (8.0 * np.tanh(rng.standard_normal((T, n_feat)) / 8.0)).astype(np.float32)
    T2 = int(round(T * rng.uniform(0.8, 1.25))) if silent else T
    audio = (rng.standard_normal((T2, n_mel)) * 0.5).astype(np.float32)
    ph = np.zeros(T2, dtype=np.int64)
    i = 0
    while i < T2:                                                               # piecewise-constant phoneme runs of 3-15 frames
        run = int(rng.integers(3, 16))
        ph[i:i + run] = rng.integers(0, n_phone)
        i += run
    text_int = rng.integers(0, 37, max(1, T // 6)).astype(np.int64)
    return {'audio_features': torch.from_numpy(audio), 'emg': torch.from_numpy(emg), 'raw_emg': torch.from_numpy(raw.astype(np.float32)),
            'session_ids': torch.full((T,), session, dtype=torch.int64), 'phonemes': torch.from_numpy(ph), 'silent': bool(silent),
            'text_int': torch.from_numpy(text_int), 'length_1k': int(T * RAW_PER_FRAME / HZ_RATIO)}


class SyntheticEMGDataset(torch.utils.data.Dataset):
    num_features = 112
    num_speech_features = 80
    num_sessions = 1

    @property
    def text_transform(self):
        from .recognition_model import TextTransform
        return TextTransform()

    def __init__(self, n_utterances=64, seed=0, min_frames=200, max_frames=860, silent_fraction=0.25):
        rng = np.random.default_rng(seed)
        self.items = [make_utterance(rng, int(rng.integers(min_frames, max_frames + 1)), rng.random() < silent_fraction) for _ in range(n_utterances)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def example_length(self, i):
        """Raw 1 kHz samples of utterance i: what read_emg.py:131 sums from the chunk table (SizeAwareSampler protocol)."""
        return self.items[i]['length_1k']

    def subset(self, fraction):
        out = SyntheticEMGDataset(0)
        out.items = self.items[:max(1, int(fraction * len(self.items)))]
        return out

    def size_aware_sampler(self, max_len, shuffle_seed=None):
        """read_emg.py:115-140: greedy batches under a budget of raw 1 kHz samples; last partial batch dropped."""
        order = list(range(len(self.items)))
        if shuffle_seed is not None:
            np.random.default_rng(shuffle_seed).shuffle(order)
        from .pipeline import greedy_length_batches
        return list(greedy_length_batches(order, self.example_length, max_len))

    @staticmethod
    def collate_raw(batch):
        return {'audio_features': [ex['audio_features'] for ex in batch],
                'audio_feature_lengths': [ex['audio_features'].shape[0] for ex in batch],
                'emg': [ex['emg'] for ex in batch], 'raw_emg': [ex['raw_emg'] for ex in batch],
                'parallel_voiced_emg': [np.zeros(1) for _ in batch], 'phonemes': [ex['phonemes'] for ex in batch],
                'session_ids': [ex['session_ids'] for ex in batch], 'lengths': [ex['emg'].shape[0] for ex in batch],
                'silent': [ex['silent'] for ex in batch], 'text_int': [ex['text_int'] for ex in batch],
                'text_int_lengths': [ex['text_int'].shape[0] for ex in batch]}


def reference_size_batch(seed=0, budget=256000, device=None, silent_fraction=0.25):
    """One batch of the size the reference trains on: utterances U{200..860} frames until the 256 000-sample
    budget (transduction_model.py:166) is full -> ~40 utterances, ~22 k frames, ~110 packed rows."""
    rng = np.random.default_rng(seed)
    items, tot = [], 0
    while True:
        T = int(rng.integers(200, 861))
        n = int(T * RAW_PER_FRAME / HZ_RATIO)
        if tot + n > budget:
            break
        items.append(make_utterance(rng, T, rng.random() < silent_fraction))
        tot += n
    batch = SyntheticEMGDataset.collate_raw(items)
    if device is not None:
        for k in ('audio_features', 'emg', 'raw_emg', 'phonemes', 'session_ids'):
            batch[k] = [t.to(device) for t in batch[k]]
    return batch
