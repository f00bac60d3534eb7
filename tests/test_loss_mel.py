"""dtw_loss (fused HIP loss path) and mel_spectrogram vs golden vectors from the reference / the oracle."""
import os

import numpy as np
from silent_speech_amd import _lib as _lib_mod
import pytest
import torch

from oracle import loss_ref, mel_ref
from silent_speech_amd import data_utils, transduction_model as tm
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _example(z, n, dev):
    return dict(lengths=z['lengths'].tolist(), silent=z['silent'].tolist(),
                audio_features=[torch.from_numpy(z['audio/%d' % i]).to(dev) for i in range(n)],
                phonemes=[torch.from_numpy(z['phones/%d' % i]).to(dev) for i in range(n)])


def test_dtw_loss_mixed_golden(dev):
    """2 voiced + 2 silent utterances: loss, accuracy, gradients and eval-mode confusion matrix vs the reference."""
    z = np.load(os.path.join(GOLD, 'dtw_loss_mixed.npz'))
    pred = torch.from_numpy(z['pred']).to(dev).requires_grad_(True)
    aux = torch.from_numpy(z['aux']).to(dev).requires_grad_(True)
    ex = _example(z, 4, dev)
    loss, acc = tm.dtw_loss(pred, aux, ex, phoneme_loss_weight=0.5)
    assert abs(float(loss) - float(z['loss'])) < 2e-5 * abs(float(z['loss']))
    assert abs(float(acc) - float(z['acc_eval'])) < 1e-6
    loss.backward()
    assert_close_robust(pred.grad, z['dpred'], 1e-4, name='dpred', max_outlier_frac=0)
    assert_close_robust(aux.grad, z['daux'], 1e-4, name='daux', max_outlier_frac=0)
    conf = np.zeros((48, 48))
    with torch.no_grad():
        le, ae = tm.dtw_loss(pred.detach(), aux.detach(), ex, True, conf, phoneme_loss_weight=0.5)
    assert abs(float(le) - float(z['loss_eval'])) < 2e-5 * abs(float(z['loss_eval']))
    assert abs(ae - float(z['acc_eval'])) < 1e-9
    assert np.array_equal(conf, z['confusion'])


def test_dtw_loss_voiced_golden(dev):
    z = np.load(os.path.join(GOLD, 'dtw_loss_voiced.npz'))
    pred = torch.from_numpy(z['pred']).to(dev).requires_grad_(True)
    aux = torch.from_numpy(z['aux']).to(dev).requires_grad_(True)
    ex = dict(lengths=[200], silent=[False], audio_features=[torch.from_numpy(z['audio']).to(dev)], phonemes=[torch.from_numpy(z['phones']).to(dev)])
    loss, acc = tm.dtw_loss(pred, aux, ex, phoneme_loss_weight=0.5)
    assert abs(float(loss) - float(z['loss'])) < 2e-5 * abs(float(z['loss']))
    loss.backward()
    assert_close_robust(pred.grad, z['dpred'], 1e-4, name='dpred', max_outlier_frac=0)
    assert_close_robust(aux.grad, z['daux'], 1e-4, name='daux', max_outlier_frac=0)


def test_dtw_loss_voiced_length_mismatch_asserts(dev):
    pred = torch.zeros(1, 50, 80, device=dev); aux = torch.zeros(1, 50, 48, device=dev)
    ex = dict(lengths=[40], silent=[False], audio_features=[torch.zeros(39, 80)], phonemes=[torch.zeros(39, dtype=torch.long)])
    with pytest.raises(AssertionError):
        tm.dtw_loss(pred, aux, ex)


@pytest.mark.gpu
def test_dtw_loss_vs_oracle_realistic_sizes():
    """Mixed batch with ~600-frame silent utterances: the alignment is computed by the HIP DTW on device."""
    from silent_speech_amd import _lib
    _lib.load()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(11)
    lengths, silent = [420, 610, 333, 505], [True, False, True, False]
    t2 = [480, 610, 300, 505]
    rows = (sum(lengths) + 199) // 200
    pred = torch.randn(rows, 200, 80, generator=g); aux = torch.randn(rows, 200, 48, generator=g)
    audio = [torch.randn(n, 80, generator=g) * 0.7 for n in t2]
    phones = [torch.randint(0, 48, (n,), generator=g) for n in t2]
    ex = dict(lengths=lengths, silent=silent, audio_features=audio, phonemes=phones)
    pr, ar = pred.clone().requires_grad_(True), aux.clone().requires_grad_(True)
    lref, accref = loss_ref.dtw_loss_ref(pr, ar, ex)
    lref.backward()
    pd, ad = pred.to(dev).requires_grad_(True), aux.to(dev).requires_grad_(True)
    loss, acc = tm.dtw_loss(pd, ad, ex, phoneme_loss_weight=0.5)
    assert abs(float(loss) - float(lref)) < 2e-5 * abs(float(lref))
    assert abs(float(acc) - accref) < 1e-6
    loss.backward()
    assert_close_robust(pd.grad, pr.grad, 1e-4, name='dpred', max_outlier_frac=0)
    assert_close_robust(ad.grad, ar.grad, 1e-4, name='daux', max_outlier_frac=0)


def test_mel_spectrogram_golden(dev):
    """STFT-as-GEMM + mel + log vs the reference's torch.stft path (basis injected identically)."""
    z = np.load(os.path.join(GOLD, 'mel.npz'))
    y = torch.from_numpy(z['y'])
    if is_emu(dev):
        y = y[:1, :256 * 6]                       # emulator: 1 clip, 5 frames
        want = mel_ref.mel_spectrogram_ref(y.numpy(), basis=z['basis'])
    else:
        want = z['mel']
    got = data_utils.mel_spectrogram(y.to(dev), 1024, 80, 22050, 256, 1024, 0, 8000, center=False)
    assert tuple(got.shape) == tuple(want.shape)
    d = (got.cpu().numpy() - want)
    assert float(np.abs(d).mean()) < 1e-4, float(np.abs(d).mean())          # north_star: mel-L1 within 1e-4
    assert float(np.abs(d).max()) < 2e-2
    assert np.array_equal(data_utils.slaney_mel_filterbank(22050, 1024, 80, 0, 8000), z['basis'])


def test_mel_fft_kernel_vs_oracle_and_vs_the_gemm_formulation(dev, monkeypatch):
    """ss_stft_logmel_fft (csrc/mel.hip: one wave per frame, LDS radix-8 FFT, sparse filterbank) against the numpy oracle (oracle/mel_ref.py, numpy rfft) and against
    the dense-DFT GEMM formulation it replaces for n_fft = 1024, in both output layouts (the reference's (B, mels, F) and the loader's frame-major one), with
    frame counts that leave waves idle / give a wave several frames.  Tolerance: 1e-4 (north_star's mel-L1) on the mean, 2e-5 on every element -- the two
    device paths are both f32 and differ only in summation order."""
    rng = np.random.default_rng(5)
    for B, frames in ((1, 2), (3, 7)) if is_emu(dev) else ((1, 2), (3, 7), (5, 1030)):
        L = 256 * frames
        y = np.clip(0.2 * rng.standard_normal((B, L)), -1, 1).astype(np.float32)
        y[0, :min(L, 300)] = 0.0                                                # silence: the 1e-9 under the root and the 1e-5 clamp decide
        yd = torch.from_numpy(y).to(dev)
        got = data_utils.mel_spectrogram(yd, 1024, 80, 22050, 256, 1024, 0, 8000).cpu().numpy()
        want = mel_ref.mel_spectrogram_ref(y)
        assert got.shape == want.shape == (B, 80, frames)
        assert float(np.abs(got - want).mean()) < 1e-4 and float(np.abs(got - want).max()) < 2e-3
        buf, fr = data_utils.mel_spectrogram_batch([yd[b] for b in range(B)])
        assert fr == [frames] * B
        assert float(np.abs(buf.cpu().numpy() - want.transpose(0, 2, 1)).max()) < 2e-3
        if B >= 3:                                                              # ragged lengths (odd offsets), clipping, frames that run past a signal's padded end
            lens = [L - 131, L, 700]
            sig = [3.0 * yd[b, :lens[b]] for b in range(3)]
            buf, fr = data_utils.mel_spectrogram_batch(sig)
            for b in range(3):
                wb = mel_ref.mel_spectrogram_ref(np.clip(3.0 * y[b:b + 1, :lens[b]], -1, 1))[0].T
                assert fr[b] == wb.shape[0]
                assert float(np.abs(buf[b, :fr[b]].cpu().numpy() - wb).max()) < 2e-3, b
        if frames <= 7:                                                         # the GEMM formulation on the same input
            calls = []
            real = _lib_mod.lib().ss_stft_logmel_fft
            monkeypatch.setattr(data_utils, '_fft_tables', lambda *a, **k: calls.append(1))      # 'does not fit the kernel's tables' -> the GEMM formulation
            dense = data_utils.mel_spectrogram(yd, 1024, 80, 22050, 256, 1024, 0, 8000).cpu().numpy()
            monkeypatch.undo()
            assert calls and real is not None
            assert float(np.abs(dense - got).max()) < 2e-3 and float(np.abs(dense - got).mean()) < 2e-5


def test_pack_roundtrip_golden():
    z = np.load(os.path.join(GOLD, 'pack.npz'))
    ts = [torch.from_numpy(z['t/%d' % i]) for i in range(4)]
    packed = data_utils.combine_fixed_length(ts, 16)
    assert np.array_equal(packed.numpy(), z['packed'])
    for a, b in zip(ts, data_utils.decollate_tensor(packed, [t.shape[0] for t in ts])):
        assert torch.equal(a, b)


def test_pack_device_kernel_golden(dev):
    """The DEVICE branch of combine_fixed_length (the `ss_concat_pad` gather over a freshly uploaded pointer table) against the
    reference's own packed tensor (tests/golden/pack.npz), twice with different source tensors: nothing may be cached on addresses."""
    z = np.load(os.path.join(GOLD, 'pack.npz'))
    ts = [torch.from_numpy(z['t/%d' % i]).to(dev) for i in range(4)]
    packed = data_utils.combine_fixed_length(ts, 16)
    assert packed.device.type == dev.type and np.array_equal(packed.cpu().numpy(), z['packed'])
    for a, b in zip(ts, data_utils.decollate_tensor(packed, [t.shape[0] for t in ts])):
        assert torch.equal(a, b)
    ts[1].mul_(2.0)                                             # in-place edit of a source: same pointers, new content
    again = data_utils.combine_fixed_length(ts, 16)
    want = data_utils.combine_fixed_length([t.cpu() for t in ts], 16)
    if not is_emu(dev):
        assert want.device.type == 'cpu'
    assert torch.equal(again.cpu(), want.cpu())
    odd = [t[:, :3].contiguous()[1:] for t in ts]               # 12-byte rows at unaligned offsets: the 4-byte granule path
    assert torch.equal(data_utils.combine_fixed_length(odd, 5).cpu(), data_utils._host_pack([t.cpu() for t in odd], 5))


def test_prepare_batch_then_dtw_loss_equals_standalone(dev):
    """prepare_batch (one upload for the pack tables + the loss plan, before the forward) followed by dtw_loss on the same dict gives
    what the separate calls give; the prepared plan is consumed by that one call; targets edited in place afterwards are honoured."""
    z = np.load(os.path.join(GOLD, 'dtw_loss_mixed.npz'))
    ex = _example(z, 4, dev)
    g = torch.Generator().manual_seed(3)
    ex['emg'] = [torch.randn(n, 112, generator=g).to(dev) for n in ex['lengths']]
    ex['raw_emg'] = [torch.randn(8 * n, 8, generator=g).to(dev) for n in ex['lengths']]
    ex['session_ids'] = [torch.full((n,), 2, dtype=torch.int64).to(dev) for n in ex['lengths']]
    row = int(z['pred'].shape[1])
    X, X_raw, sess = tm.prepare_batch(ex, dev, seq_len=row)
    assert tm._Prepared.example is ex
    assert torch.equal(X.cpu(), data_utils._host_pack([t.cpu() for t in ex['emg']], row))
    assert torch.equal(X_raw.cpu(), data_utils._host_pack([t.cpu() for t in ex['raw_emg']], row * 8))
    assert torch.equal(sess.cpu(), data_utils._host_pack([t.cpu() for t in ex['session_ids']], row))
    pred, aux = torch.from_numpy(z['pred']).to(dev), torch.from_numpy(z['aux']).to(dev)
    l1, a1 = tm.dtw_loss(pred, aux, ex, True, None, phoneme_loss_weight=0.5)
    assert tm._Prepared.example is None                         # one-shot
    l2, a2 = tm.dtw_loss(pred, aux, ex, True, None, phoneme_loss_weight=0.5)
    assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1)) and a1 == a2        # f32 atomics: the block sums arrive in any order
    assert abs(float(l1) - float(z['loss_eval'])) < 2e-5 * abs(float(z['loss_eval']))
    ex['audio_features'][0].add_(0.25)                          # the advisor's case: a write that a (pointer, version) signature may miss
    l3, _ = tm.dtw_loss(pred, aux, ex, True, None, phoneme_loss_weight=0.5)
    cpu = dict(ex, audio_features=[a.cpu() for a in ex['audio_features']], phonemes=[p.cpu() for p in ex['phonemes']])
    want, _ = loss_ref.dtw_loss_ref(pred.cpu(), aux.cpu(), cpu, lam=0.5)
    assert abs(float(l3) - float(want)) < 2e-5 * abs(float(want)) and abs(float(l3) - float(l1)) > 1e-4 * abs(float(l1))
    # targets edited BETWEEN prepare_batch and dtw_loss (same dict, same lengths): the prepared plan snapshotted the old values; the
    # (address, version) signature of the targets sends this call to a fresh plan instead of silently using them
    tm.prepare_batch(ex, dev, seq_len=row)
    ex['audio_features'][1].mul_(0.5)
    l4, _ = tm.dtw_loss(pred, aux, ex, True, None, phoneme_loss_weight=0.5)
    cpu = dict(ex, audio_features=[a.cpu() for a in ex['audio_features']], phonemes=[p.cpu() for p in ex['phonemes']])
    want4, _ = loss_ref.dtw_loss_ref(pred.cpu(), aux.cpu(), cpu, lam=0.5)
    assert abs(float(l4) - float(want4)) < 2e-5 * abs(float(want4)) and abs(float(l4) - float(l3)) > 1e-4 * abs(float(l3))


def test_dtw_loss_all_silent_and_degenerate_utterances(dev):
    """Edge cases of the packed layout: every utterance silent, a 1-frame utterance (1 x M and N x 1 cost matrices: no
    interior DTW cell, align.py:24 leaves results at 0), ragged rows with padding frames, targets shorter and longer than
    the prediction."""
    g = torch.Generator().manual_seed(21)
    lengths = [1, 23, 9, 30]
    t2 = [4, 1, 14, 26]
    row = 16
    B = (sum(lengths) + row - 1) // row
    pred = torch.randn(B, row, 80, generator=g)
    aux = torch.randn(B, row, 48, generator=g)
    audio = [torch.randn(n, 80, generator=g) * 0.7 for n in t2]
    phones = [torch.randint(0, 48, (n,), generator=g) for n in t2]
    ex = dict(lengths=lengths, silent=[True] * 4, audio_features=audio, phonemes=phones)
    pr, ar = pred.clone().requires_grad_(True), aux.clone().requires_grad_(True)
    want, want_acc = loss_ref.dtw_loss_ref(pr, ar, ex, lam=0.5)
    want.backward()
    pd, ad = pred.to(dev).requires_grad_(True), aux.to(dev).requires_grad_(True)
    exd = dict(ex, audio_features=[a.to(dev) for a in audio], phonemes=[p.to(dev) for p in phones])
    got, acc = tm.dtw_loss(pd, ad, exd, True, None, phoneme_loss_weight=0.5)
    assert abs(float(got.detach()) - float(want.detach())) < 2e-5 * abs(float(want.detach()))
    assert abs(acc - want_acc) < 1e-9
    got.backward()
    assert_close_robust(pd.grad, pr.grad, 1e-4, name='dpred', max_outlier_frac=0)
    assert_close_robust(ad.grad, ar.grad, 1e-4, name='daux', max_outlier_frac=0)
    used = sum(lengths)
    assert not pd.grad.reshape(-1, 80)[used:].any() and not ad.grad.reshape(-1, 48)[used:].any()
