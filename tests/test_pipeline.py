"""Device-side input pipeline ("next" row N3): conditioning kernels vs the reference's numpy expressions, mel targets vs
the oracle, the sharded size-aware sampler vs the reference's packing rule."""
import os

import numpy as np
import pytest
import torch

from oracle import mel_ref
from silent_speech_amd import pipeline
from silent_speech_amd.data_utils import FeatureNormalizer
from tests.backend import dev, is_emu  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_condition_raw_emg_matches_reference_expression(dev):
    rng = np.random.default_rng(0)
    raw = (rng.standard_normal((123, 8)) * 400).astype(np.float32)
    want = raw / 20
    want = 50 * np.tanh(want / 50.)                                   # read_emg.py:227-228
    got = pipeline.condition_raw_emg(torch.from_numpy(raw).to(dev)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_normalize_features_with_the_shipped_normalizers(dev):
    z = np.load(os.path.join(GOLD, 'normalizers.npz'))
    rng = np.random.default_rng(1)
    norm = FeatureNormalizer([rng.standard_normal((4, 112)).astype(np.float32)])
    norm.feature_means, norm.feature_stddevs = z['emg_means'], z['emg_stds']
    x = (rng.standard_normal((57, 112)) * 30 + 5).astype(np.float32)
    want = 8 * np.tanh(norm.normalize(x.copy()) / 8.)                 # read_emg.py:231-233
    got = pipeline.normalize_features(torch.from_numpy(x).to(dev), norm, limit=8.0).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    mnorm = FeatureNormalizer([rng.standard_normal((4, 80)).astype(np.float32)], share_scale=True)
    mnorm.feature_means, mnorm.feature_stddevs = z['mfcc_means'], float(z['mfcc_std'])
    m = rng.standard_normal((31, 80)).astype(np.float32)
    got = pipeline.normalize_features(torch.from_numpy(m).to(dev), mnorm).cpu().numpy()
    np.testing.assert_allclose(got, mnorm.normalize(m.copy()), rtol=1e-5, atol=1e-5)


def test_mel_targets_match_the_reference_mel(dev):
    z = np.load(os.path.join(GOLD, 'mel.npz'))                          # reference mel_spectrogram output of 3 signals
    y = torch.from_numpy(z['y'][2][:256 * 13]).to(dev)                  # the chirp, 12 frames (small enough for the emulator)
    got = pipeline.mel_targets(y, None, max_frames=10).cpu().numpy()
    want = z['mel'][2].T[:10]
    assert got.shape == (10, 80)
    assert np.abs(got - want).mean() < 1e-4 and np.abs(got - want).max() < 1e-2


def test_sharded_sampler_partitions_the_reference_packing():
    rng = np.random.default_rng(2)
    lengths = [int(v) for v in rng.integers(2000, 12000, 200)]
    full = pipeline.ShardedSizeAwareSampler(lengths, 128000, 0, 1, seed=3)
    batches = full.all_batches()
    for b in batches:
        assert sum(lengths[i] for i in b) <= 128000
    flat = [i for b in batches for i in b]
    assert len(set(flat)) == len(flat)
    # greedy rule of read_emg.py:131-137: the next utterance would not have fitted
    order = list(range(len(lengths)))
    import random
    random.Random(3 * 1000003).shuffle(order)
    pos = {v: k for k, v in enumerate(order)}
    for b in batches:
        nxt = pos[b[-1]] + 1
        assert nxt < len(order) and sum(lengths[i] for i in b) + lengths[order[nxt]] > 128000
    shards = [list(pipeline.ShardedSizeAwareSampler(lengths, 128000, r, 4, seed=3)) for r in range(4)]
    assert len({len(s) for s in shards}) == 1                           # same number of steps on every rank
    dealt = [b for k in range(len(shards[0])) for s in shards for b in [s[k]]]
    assert dealt == batches[:len(dealt)]
    ep1 = pipeline.ShardedSizeAwareSampler(lengths, 128000, 0, 1, seed=3); ep1.set_epoch(1)
    assert ep1.all_batches() != batches


def test_size_aware_sampler_on_the_reference_dataset_protocol(tmp_path):
    """SizeAwareSampler(emg_dataset, max_len) exactly as transduction_model.py:166 constructs it, on an object that only offers
    what the reference's EMGDataset offers: example_indices -> (directory_info, file_idx) and <file_idx>_info.json on disk."""
    import json
    import random

    class _Dir(object):
        def __init__(self, d):
            self.directory = d
    rng = np.random.default_rng(5)
    d = str(tmp_path)
    lengths = {}
    for i in range(60):
        chunks = [[int(v), 0, 0] for v in rng.integers(500, 3000, 3)]
        text = '' if i % 13 == 5 else 'utterance %d' % i                  # text-less utterances are skipped (read_emg.py:129-130)
        with open(os.path.join(d, '%d_info.json' % i), 'w') as f:
            json.dump({'text': text, 'chunks': chunks}, f)
        lengths[i] = None if not text else sum(c[0] for c in chunks)

    class _DS(object):
        example_indices = [(_Dir(d), i) for i in range(60)]

        def __len__(self):
            return 60
    random.seed(11)
    batches = list(pipeline.SizeAwareSampler(_DS(), 20000))
    flat = [i for b in batches for i in b]
    assert len(set(flat)) == len(flat) and all(lengths[i] is not None for i in flat)
    assert all(sum(lengths[i] for i in b) <= 20000 for b in batches) and len(batches) >= 3
    random.seed(12)
    assert list(pipeline.SizeAwareSampler(_DS(), 20000)) != batches         # reshuffled on every __iter__ (read_emg.py:122)
    # data-parallel extras: every rank gets the same number of batches of one shared shuffle
    shards = [list(pipeline.SizeAwareSampler(_DS(), 20000, rank=r, world=2, seed=4)) for r in range(2)]
    assert len(shards[0]) == len(shards[1]) and not (set(map(tuple, shards[0])) & set(map(tuple, shards[1])))


def test_mel_spectrogram_center_true_matches_torch_stft(dev):
    """center=True is forwarded to torch.stft by the reference (data_utils.py:54): n_fft // 2 extra samples reflected off the padded signal."""
    from silent_speech_amd.data_utils import mel_spectrogram, slaney_mel_filterbank
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(2, 256 * 8, generator=g) * 2 - 1) * 0.5
    got = mel_spectrogram(y.to(dev), 1024, 80, 22050, 256, 1024, 0, 8000, center=True).cpu()
    pad = (1024 - 256) // 2
    yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode='reflect').squeeze(1)
    spec = torch.stft(yp, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True, pad_mode='reflect', normalized=False,
                      onesided=True, return_complex=True)
    want = torch.log(torch.clamp(torch.from_numpy(slaney_mel_filterbank(22050, 1024, 80, 0, 8000)) @ torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9), min=1e-5))
    assert got.shape == want.shape and float((got - want).abs().mean()) < 1e-4


def _recording(rng, n_1k, audio_frames, silent=False, context=True):
    t = np.arange(n_1k + 400) / 1000.0
    walk = np.cumsum(rng.standard_normal((n_1k + 400, 8)), 0) * 2.0                    # drift
    hum = 40.0 * np.sin(2 * np.pi * 60.0 * t)[:, None] * rng.uniform(0.5, 1.5, 8)[None]  # mains
    x = (walk + hum + rng.standard_normal((n_1k + 400, 8)) * 30.0)
    rec = {'raw_emg': x[200:200 + n_1k].copy(), 'silent': silent, 'session_index': 3,
           'audio': np.clip(rng.standard_normal(256 * audio_frames).astype(np.float32) * 0.4, -1.2, 1.2),      # some samples out of range: np.clip matters
           'text_int': rng.integers(0, 37, 5).astype(np.int64)}
    if context:
        rec['raw_emg_before'], rec['raw_emg_after'] = x[:200].copy(), x[200 + n_1k:].copy()
    rec['phonemes'] = rng.integers(0, 48, audio_frames).astype(np.int64)
    return rec


def _oracle_item(rec, mean, std, limit=False):
    """load_utterance + EMGDataset.__getitem__ (read_emg.py:52-100, 223-235) restated with the oracle's numpy pieces."""
    from oracle import filter_ref, mel_ref
    e689, e516 = filter_ref.condition(rec['raw_emg'], rec.get('raw_emg_before'), rec.get('raw_emg_after'))
    n_feat = 1 + (e516.shape[0] - 16) // 6                                               # librosa.util.frame(16, 6), data_utils.py:100
    mel = mel_ref.mel_spectrogram_ref(np.clip(rec['audio'], -1, 1)[None].astype(np.float32))[0].T
    n = min(n_feat, mel.shape[0], 800 if limit else 10 ** 9)
    raw = e689[8:8 + 8 * n].astype(np.float32)
    raw = (50 * np.tanh((raw / 20) / 50.)).astype(np.float32)
    return n, raw, ((mel[:n] - mean) / std).astype(np.float32), rec['phonemes'][:n]


def test_device_batch_builder_matches_the_per_utterance_chain(dev):
    """N3 composed: DeviceBatchBuilder.build(recordings) == [dataset[i] for i in batch] + collate_raw of the reference, restated per
    utterance with the oracle's filtfilt / np.interp / mel pieces: frame counts, conditioned raw EMG, normalised mel targets, phoneme
    targets (the voiced twin's for silent recordings)."""
    from silent_speech_amd.data_utils import FeatureNormalizer
    rng = np.random.default_rng(17)
    if is_emu(dev):
        recs = [_recording(rng, 260, 10), _recording(rng, 300, 14, silent=True, context=False)]
        recs[1]['parallel'] = _recording(rng, 280, 12)
    else:
        recs = [_recording(rng, int(n), int(f), silent=s, context=c) for n, f, s, c in
                [(3000, 300, False, True), (4200, 330, True, True), (5100, 500, False, False), (3600, 280, False, True),
                 (6000, 520, True, False), (3300, 400, False, True), (4800, 390, False, True), (2500, 190, False, True)]]
        recs[1]['parallel'] = _recording(rng, 4000, 350)
        recs[4]['parallel'] = _recording(rng, 5600, 470)
    norm = FeatureNormalizer([np.zeros((2, 80), dtype=np.float32)], share_scale=True)
    norm.feature_means = np.linspace(-6, -3, 80, dtype=np.float32)[None]
    norm.feature_stddevs = np.float32(2.5)
    b = pipeline.DeviceBatchBuilder(dev, mfcc_norm=norm).build(recs)
    assert set(b) == {'audio_features', 'audio_feature_lengths', 'emg', 'raw_emg', 'parallel_voiced_emg', 'phonemes', 'session_ids', 'lengths',
                      'silent', 'text_int', 'text_int_lengths'}
    mean, std = norm.feature_means.reshape(-1), float(norm.feature_stddevs)
    for i, r in enumerate(recs):
        n, raw, mel, ph = _oracle_item(r, mean, std)
        assert b['lengths'][i] == n and b['silent'][i] == r['silent']
        assert tuple(b['raw_emg'][i].shape) == (8 * n, 8) and tuple(b['emg'][i].shape) == (n, 112) and tuple(b['session_ids'][i].shape) == (n,)
        got = b['raw_emg'][i].cpu().numpy()
        assert float(np.abs(got - raw).max()) < 2e-4 * float(np.abs(raw).max()), (i, float(np.abs(got - raw).max()))
        if r['silent']:
            n, _, mel, ph = _oracle_item(r['parallel'], mean, std)
        assert b['audio_feature_lengths'][i] == n and tuple(b['audio_features'][i].shape) == (n, 80)
        d = b['audio_features'][i].cpu().numpy() - mel
        assert float(np.abs(d).mean()) < 1e-4 and float(np.abs(d).max()) < 2e-2, (i, float(np.abs(d).mean()))
        assert np.array_equal(b['phonemes'][i].cpu().numpy(), ph)
        assert int(b['session_ids'][i][0]) == 3 and b['text_int_lengths'][i] == 5
    # FLAGS.remove_channels (read_emg.py:73-75): those electrode columns are zero, the others unchanged; the voiced twin's EMG features ride along
    recs[1]['parallel']['emg_features'] = np.ones((b['audio_feature_lengths'][1], 112), dtype=np.float32)
    b2 = pipeline.DeviceBatchBuilder(dev, mfcc_norm=norm, remove_channels=(2, 5)).build(recs)
    for i in range(len(recs)):
        g0, g2 = b['raw_emg'][i].cpu(), b2['raw_emg'][i].cpu()
        assert float(g2[:, [2, 5]].abs().max()) == 0.0 and torch.equal(g2[:, [0, 1, 3, 4, 6, 7]], g0[:, [0, 1, 3, 4, 6, 7]])
    assert torch.is_tensor(b2['parallel_voiced_emg'][1]) and tuple(b2['parallel_voiced_emg'][1].shape) == (b['audio_feature_lengths'][1], 112)
    assert isinstance(b2['parallel_voiced_emg'][0], np.ndarray)


@pytest.mark.gpu
def test_training_step_runs_from_a_built_batch():
    """The dict the builder emits is what _pack_batch / Model / dtw_loss consume (silent + voiced utterances, device-resident)."""
    from silent_speech_amd import _lib
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.transduction_model import _pack_batch, dtw_loss
    _lib.load()
    dev = torch.device('cuda')
    rng = np.random.default_rng(2)
    recs = [_recording(rng, 3000, 300), _recording(rng, 3500, 320, silent=True), _recording(rng, 2800, 200)]
    recs[1]['parallel'] = _recording(rng, 3300, 290)
    batch = pipeline.DeviceBatchBuilder(dev).build(recs)
    torch.manual_seed(0)
    m = Model(112, 80, 48, model_size=64, num_layers=1, dropout=0.1).to(dev)
    m.train()
    X, X_raw, sess = _pack_batch(batch, dev)
    pred, aux = m(X, X_raw, sess)
    loss, _ = dtw_loss(pred, aux, batch, phoneme_loss_weight=0.5)
    loss.backward()
    assert torch.isfinite(loss).item() and float(m.w_out.weight.grad.abs().max()) > 0
