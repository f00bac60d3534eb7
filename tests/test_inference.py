"""Validation / inference path ("next" row N2): whole-utterance (un-chunked) eval forward, get_aligned_prediction
(transduction_model.py:75-96) and EnsembleModel (evaluate.py:22-34) vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import dtw_ref, model_ref
from silent_speech_amd import transduction_model as tm
from silent_speech_amd.architecture import Model
from silent_speech_amd.data_utils import FeatureNormalizer
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _model(dev, seed_shift=0.0):
    z = np.load(os.path.join(GOLD, 'model_d16_L1_train_r3_T40.npz'))
    sd = {k[3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith('sd/')}
    if seed_shift:
        g = torch.Generator().manual_seed(5)
        for k in sd:
            if sd[k].dtype == torch.float32 and 'running_var' not in k and 'relative_positional' not in k:
                sd[k] = sd[k] + seed_shift * torch.randn(sd[k].shape, generator=g)
    m = Model(112, 80, 48, model_size=16, num_layers=1, dropout=0.0, compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    return m.to(dev), sd


def _datapoint(T, T2, silent, seed):
    g = torch.Generator().manual_seed(seed)
    raw = 50.0 * torch.tanh(torch.randn(8 * T, 8, generator=g) * 5.0 / 50.0)
    d = dict(raw_emg=raw, emg=torch.zeros(T, 112), session_ids=torch.zeros(T, dtype=torch.long), silent=silent,
             audio_features=torch.randn(T if not silent else T2, 80, generator=g) * 0.5)
    if silent:
        d['parallel_voiced_audio_features'] = torch.randn(T2, 80, generator=g) * 0.5
    return d


@pytest.mark.parametrize('T', [53, 8])
def test_whole_utterance_forward_any_length(dev, T):
    """One utterance, batch of 1, a length that is not a multiple of anything (evaluate.py / save_output path)."""
    m, sd = _model(dev)
    dp = _datapoint(T, T, False, 1)
    want, want_aux = model_ref.model_forward(sd, dp['raw_emg'][None].clone(), training=False)
    got = tm.predict_utterance(m, dp, dev)
    assert got.shape == (T, 80) and m.training
    assert_close_robust(got, want[0], 2e-4, name='pred', max_outlier_frac=0)


def test_get_aligned_prediction_silent_and_voiced(dev):
    m, sd = _model(dev)
    norm = FeatureNormalizer([np.random.default_rng(0).standard_normal((50, 80)).astype(np.float32) * 2 + 1], share_scale=True)
    # voiced: identity alignment
    dp = _datapoint(37, 37, False, 2)
    out = tm.get_aligned_prediction(m, dp, dev, norm)
    want, _ = model_ref.model_forward(sd, dp['raw_emg'][None].clone(), training=False)
    assert_close_robust(out, norm.inverse(want[0]), 2e-4, name='voiced', max_outlier_frac=0)
    # silent: DTW against the parallel voiced features, plain Euclidean cost (torch.cdist in the reference, :87)
    dp = _datapoint(41, 57, True, 3)
    out = tm.get_aligned_prediction(m, dp, dev, norm)
    want, _ = model_ref.model_forward(sd, dp['raw_emg'][None].clone(), training=False)
    y = dp['parallel_voiced_audio_features']
    costs = (want[0][:, None, :] - y[None, :, :]).pow(2).sum(-1).sqrt()
    align = dtw_ref.align_from_distances_c(costs.T.contiguous().numpy())
    assert out.shape == (57, 80) and m.training
    assert_close_robust(out, norm.inverse(want[0][align]), 2e-4, name='silent', max_outlier_frac=0)


def test_confusion_matrix_on_the_device_equals_the_host_loop(dev):
    """transduction_model.py:130-137,147-152: confusion[pred][target] += 1 per target frame, through the DTW alignment for silent utterances.
    The device kernel (ss_phoneme_confusion; what test() accumulates into, one read-back per epoch) against the reference's per-utterance host
    loop on a mixed silent / voiced batch, and the numpy-matrix form of the dtw_loss argument."""
    from oracle import loss_ref
    from silent_speech_amd.synthetic import SyntheticEMGDataset, make_utterance
    m, sd = _model(dev)
    m.eval()
    rng = np.random.default_rng(11)
    batch = SyntheticEMGDataset.collate_raw([make_utterance(rng, 30, False), make_utterance(rng, 26, True), make_utterance(rng, 41, True),
                                             make_utterance(rng, 23, False)])
    batch = {k: ([t.to(dev) for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in batch.items()}
    dc = tm.DeviceConfusion(48, dev)
    host = np.zeros((48, 48))
    with torch.no_grad():
        for rep in range(2):                                # two batches accumulate into one device matrix
            X, X_raw, sess = tm._pack_batch(batch, dev, seq_len=40)
            pred, aux = m(X, X_raw, sess)
            loss, acc = tm.dtw_loss(pred, aux, batch, True, dc, phoneme_loss_weight=0.5)
            assert torch.is_tensor(acc)                     # no host synchronisation on this path
            X, X_raw, sess = tm._pack_batch(batch, dev, seq_len=40)
            pred, aux = m(X, X_raw, sess)
            loss2, acc2 = tm.dtw_loss(pred, aux, batch, True, host, phoneme_loss_weight=0.5)
            assert isinstance(acc2, float) and abs(acc2 - float(acc)) < 1e-6
        # the reference's loop on the oracle's predictions
        cpu = {k: ([t.cpu() for t in v] if isinstance(v, list) and len(v) and torch.is_tensor(v[0]) else v) for k, v in batch.items()}
        xr = loss_ref.combine_fixed_length(cpu['raw_emg'], 320)
        pr, ar = model_ref.model_forward(sd, xr, training=False)
        want = np.zeros((48, 48))
        _, _, aligns = loss_ref.dtw_loss_ref(pr, ar, cpu, lam=0.5, return_alignments=True)
        pps = loss_ref.decollate_tensor(ar, cpu['lengths'])
        for pp, yp, al in zip(pps, cpu['phonemes'], aligns):            # transduction_model.py:130-137 (silent: through the alignment), :147-152
            p = pp.argmax(-1).numpy()
            np.add.at(want, (p[al] if al is not None else p, yp.numpy()), 1)
    got = dc.numpy()
    assert got.sum() == 2 * sum(int(a.shape[0]) for a in cpu['audio_features'])
    assert np.array_equal(got, host.astype(np.int64))
    assert np.array_equal(got, 2 * want.astype(np.int64))
    # a label outside the inventory (the reference's numpy indexing raises IndexError, :134-137): counted on the device, raised at the read-back
    bad = dict(batch)
    bad['phonemes'] = [t.clone() for t in batch['phonemes']]
    bad['phonemes'][0][3] = 48
    with torch.no_grad():
        X, X_raw, sess = tm._pack_batch(bad, dev, seq_len=40)
        pred, aux = m(X, X_raw, sess)
        dc2 = tm.DeviceConfusion(48, dev)
        tm.dtw_loss(pred, aux, bad, True, dc2, phoneme_loss_weight=0.0)
    with pytest.raises(IndexError, match='1 frame'):
        dc2.numpy()


def test_ensemble_model_averages(dev):
    m1, sd1 = _model(dev)
    m2, sd2 = _model(dev, seed_shift=0.02)
    ens = tm.EnsembleModel([m1, m2]).eval()
    dp = _datapoint(24, 24, False, 4)
    with torch.no_grad():
        y, p = ens(dp['emg'][None].to(dev), dp['raw_emg'][None].to(dev), dp['session_ids'][None].to(dev))
    w1, a1 = model_ref.model_forward(sd1, dp['raw_emg'][None].clone(), training=False)
    w2, a2 = model_ref.model_forward(sd2, dp['raw_emg'][None].clone(), training=False)
    assert_close_robust(y, 0.5 * (w1 + w2), 2e-4, name='ens pred', max_outlier_frac=0)
    assert_close_robust(p, 0.5 * (a1 + a2), 2e-4, name='ens aux', max_outlier_frac=0)


@pytest.mark.gpu
@pytest.mark.parametrize('dt,tol', [(torch.float32, 3e-4), (torch.bfloat16, 8e-2)])
def test_long_utterance_gpu(dt, tol):
    """T ~ 1000 frames in one row: the attention band (+-99 frames) is a small part of the T x T square."""
    from silent_speech_amd import _lib
    _lib.load()
    sd = model_ref.init_state_dict(d_model=64, num_layers=2, seed=3)
    m = Model(112, 80, 48, model_size=64, num_layers=2, dropout=0.0, compute_dtype=dt)
    m.load_state_dict(sd, strict=True)
    m.to('cuda')
    dp = _datapoint(1003, 1003, False, 6)
    want, _ = model_ref.model_forward(sd, dp['raw_emg'][None].clone(), training=False)
    got = tm.predict_utterance(m, dp, 'cuda')
    assert_close_robust(got, want[0], tol, name='long', max_outlier_frac=0 if dt == torch.float32 else 1e-3)
