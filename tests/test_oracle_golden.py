"""Pin the oracle (oracle/) against golden vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from tests.util import assert_close_robust
from oracle import adamw_ref, dtw_ref, loss_ref, mel_ref, model_ref

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def group(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


# ---------------------------------------------------------------- DTW (bit-exact)
def test_dtw_small_bit_exact_numpy_and_c():
    z = load('dtw_small')
    for name, costs in group(z, 'costs/').items():
        want = z['align/' + name].tolist()
        assert dtw_ref.align_from_distances_numpy(costs) == want, name
        got_c, dtw_c = dtw_ref.align_from_distances_c(costs, return_dtw=True)
        assert got_c == want, name
        ref_dtw = z['dtw/' + name]
        # interior cells bit-exact (row 0 / col 0 are the inf/0 frame)
        assert np.array_equal(dtw_c[1:, 1:], ref_dtw[1:, 1:]), name
        assert np.array_equal(dtw_ref.time_warp_numpy(costs), ref_dtw), name


def test_dtw_strided_view():
    z = load('dtw_small')
    c = z['costs/walk_T_75x90']
    cT = np.ascontiguousarray(c.T).T          # non-contiguous view, as costs.T at transduction_model.py:126
    assert not cT.flags['C_CONTIGUOUS']
    assert dtw_ref.align_from_distances_c(cT) == z['align/walk_T_75x90'].tolist()


def test_dtw_big():
    z = load('dtw_big')
    big = np.random.default_rng(int(z['seed'])).random(tuple(z['shape']), dtype=np.float32)
    al, dtw = dtw_ref.align_from_distances_c(big, return_dtw=True)
    assert al == z['align'].tolist()
    assert np.array_equal(dtw[-1, 1:], z['dtw_last'][1:])
    assert float(dtw[1:, 1:].astype(np.float64).sum()) == float(z['dtw_sum64'])


# ---------------------------------------------------------------- packing
def test_pack_roundtrip():
    z = load('pack')
    ts = [torch.from_numpy(z['t/%d' % i]) for i in range(4)]
    packed = loss_ref.combine_fixed_length(ts, 16)
    assert np.array_equal(packed.numpy(), z['packed'])
    for a, b in zip(ts, loss_ref.decollate_tensor(packed, [t.shape[0] for t in ts])):
        assert torch.equal(a, b)


# ---------------------------------------------------------------- attention + relpos closed form
@pytest.mark.parametrize('T', [50, 100, 200, 250])
def test_mha_closed_form(T):
    z = load('mha_T%d' % T)
    x = torch.from_numpy(z['x']).transpose(0, 1).contiguous().requires_grad_(True)   # (B,T,d)
    sd = {'a.w_q': torch.from_numpy(z['w_q']).requires_grad_(True), 'a.w_k': torch.from_numpy(z['w_k']).requires_grad_(True),
          'a.w_v': torch.from_numpy(z['w_v']).requires_grad_(True), 'a.w_o': torch.from_numpy(z['w_o']).requires_grad_(True),
          'a.relative_positional.embeddings': torch.from_numpy(z['E'])}
    q = torch.einsum('btf,hfa->bhta', x, sd['a.w_q'])
    pos = model_ref.relpos_logits(q, sd['a.relative_positional.embeddings'])
    want_pos = torch.from_numpy(z['pos'])
    assert torch.allclose(pos[:1].detach(), want_pos, rtol=1e-5, atol=2e-5)
    if T > 100:   # out-of-band entries are exactly -1e8
        assert float(pos[0, 0, 0, T - 1].detach()) == -1e8 and float(want_pos[0, 0, 0, T - 1]) == -1e8
    out = model_ref.mha(x, sd, 'a')
    want = torch.from_numpy(z['out']).transpose(0, 1)
    assert torch.allclose(out, want, rtol=1e-4, atol=2e-5)
    (out * torch.from_numpy(z['w']).transpose(0, 1)).sum().backward()
    assert torch.allclose(x.grad, torch.from_numpy(z['dx']).transpose(0, 1), rtol=1e-4, atol=3e-5)
    for n in ('w_q', 'w_k', 'w_v', 'w_o'):
        assert torch.allclose(sd['a.' + n].grad, torch.from_numpy(z['d' + n]), rtol=1e-3, atol=1e-4), n


# ---------------------------------------------------------------- full model
@pytest.mark.parametrize('name', ['model_d8_L1_eval', 'model_d8_L1_train_r0', 'model_d16_L2_train_r3',
                                  'model_d16_L2_train_r7_T120'])
def test_model_forward_backward(name):
    z = load(name)
    training = bool(z['training'])
    sd = {k: torch.from_numpy(v) for k, v in group(z, 'sd/').items()}
    for k, v in sd.items():
        if v.dtype == torch.float32 and 'running' not in k:
            v.requires_grad_(True)
    x_raw = torch.from_numpy(z['x_raw']).clone()
    running = {}
    pred, aux = model_ref.model_forward(sd, x_raw, training=training, shift_r=int(z['r']), running_out=running)
    assert torch.allclose(pred, torch.from_numpy(z['pred']), rtol=1e-4, atol=1e-4)
    assert torch.allclose(aux, torch.from_numpy(z['aux']), rtol=1e-4, atol=1e-4)
    assert float((pred - torch.from_numpy(z['pred'])).abs().mean()) < 1e-5     # "mel-L1" of the oracle vs reference
    if not training:
        return
    assert np.array_equal(x_raw.numpy(), z['x_raw_after'])
    (pred * torch.from_numpy(z['wp'])).sum().add((aux * torch.from_numpy(z['wa'])).sum()).backward()
    for k, g in group(z, 'grad/').items():
        got = sd[k].grad
        assert got is not None, k
        # conv biases feed training-mode BatchNorm -> true gradient 0 (rounding noise only): skip them
        if k.endswith('.bias') and ('conv1' in k or 'conv2' in k or 'residual_path' in k):
            assert float(got.abs().max()) < 1e-3, k
            continue
        assert_close_robust(got, g, rtol=2e-3, name=k, min_outliers=40)  # 2e-3: one ReLU flip also perturbs everything upstream of it
    for k in group(z, 'nograd/'):
        assert 'relative_positional' in k           # the ONLY grad-less parameters
        assert sd[k].grad is None
    for k, v in group(z, 'after/').items():
        if k in running:
            assert torch.allclose(running[k], torch.from_numpy(v), rtol=1e-5, atol=1e-6), k


# ---------------------------------------------------------------- dtw_loss
def _example(z, n):
    return dict(lengths=z['lengths'].tolist(), silent=z['silent'].tolist(),
                audio_features=[torch.from_numpy(z['audio/%d' % i]) for i in range(n)],
                phonemes=[torch.from_numpy(z['phones/%d' % i]) for i in range(n)])


def test_dtw_loss_mixed():
    z = load('dtw_loss_mixed')
    pred = torch.from_numpy(z['pred']).requires_grad_(True)
    aux = torch.from_numpy(z['aux']).requires_grad_(True)
    loss, acc = loss_ref.dtw_loss_ref(pred, aux, _example(z, 4))
    assert abs(float(loss) - float(z['loss'])) < 1e-5 * abs(float(z['loss']))
    assert abs(acc - float(z['acc_eval'])) < 1e-12
    loss.backward()
    assert torch.allclose(pred.grad, torch.from_numpy(z['dpred']), rtol=1e-4, atol=1e-7)
    assert torch.allclose(aux.grad, torch.from_numpy(z['daux']), rtol=1e-4, atol=1e-7)


def test_dtw_loss_voiced():
    z = load('dtw_loss_voiced')
    pred = torch.from_numpy(z['pred']).requires_grad_(True)
    aux = torch.from_numpy(z['aux']).requires_grad_(True)
    ex = dict(lengths=[200], silent=[False], audio_features=[torch.from_numpy(z['audio'])], phonemes=[torch.from_numpy(z['phones'])])
    loss, acc = loss_ref.dtw_loss_ref(pred, aux, ex)
    assert abs(float(loss) - float(z['loss'])) < 1e-5 * abs(float(z['loss']))
    loss.backward()
    assert torch.allclose(pred.grad, torch.from_numpy(z['dpred']), rtol=1e-4, atol=1e-7)
    assert torch.allclose(aux.grad, torch.from_numpy(z['daux']), rtol=1e-4, atol=1e-7)


# ---------------------------------------------------------------- mel
def test_mel_vs_reference():
    z = load('mel')
    got = mel_ref.mel_spectrogram_ref(z['y'], basis=z['basis'])
    assert got.shape == z['mel'].shape
    # torch.stft (pocketfft f32) vs float64 DFT: log-mel agrees to ~1e-5; the 1e-4 bar of north_star
    assert float(np.abs(got - z['mel']).mean()) < 1e-4
    assert float(np.abs(got - z['mel']).max()) < 1e-2   # f32 FFT noise floor under the chirp peak, log-amplified


def test_mel_basis_self_checks():
    """Basis is 'parity unpinned' (librosa absent): structural self-checks from SURVEY 8c."""
    b = mel_ref.slaney_mel_basis(22050, 1024, 80, 0, 8000)
    assert b.shape == (80, 513) and b.dtype == np.float32
    assert int((b != 0).sum()) == 727
    freqs = np.linspace(0, 11025, 513)
    assert np.all(b[:, freqs > 8000] == 0)
    assert np.all(b >= 0) and np.all(b.sum(1) > 0)


# ---------------------------------------------------------------- AdamW
def test_adamw_three_steps():
    z = load('adamw')
    p = torch.from_numpy(z['p'][0]); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for it in range(3):
        lr = adamw_ref.warmup_lr(it)
        p, m, v = adamw_ref.adamw_step_ref(p, torch.from_numpy(z['g'][it]), m, v, it + 1, lr)
        assert torch.allclose(p, torch.from_numpy(z['p'][it + 1]), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('tag', ['ctc_mixed', 'ctc_repeats', 'ctc_infeasible'])
def test_ctc_oracle_matches_reference(tag):
    """oracle/ctc_ref.py against the reference's loss lines (recognition_model.py:96-101) run on torch CPU."""
    from oracle.ctc_ref import ctc_loss_packed
    z = np.load(os.path.join(GOLD, tag + '.npz'))
    lengths = [int(n) for n in z['lengths']]
    targets = [z['text/%d' % i] for i in range(len(lengths))]
    loss, d, nll = ctc_loss_packed(z['logits'], lengths, targets, int(z['blank']))
    fin = np.isfinite(z['nll'])
    assert np.array_equal(np.isfinite(nll), fin)
    np.testing.assert_allclose(nll[fin], z['nll'][fin], rtol=2e-6)
    if fin.all():
        np.testing.assert_allclose(loss, float(z['loss']), rtol=2e-6)
    else:
        assert not np.isfinite(loss) and not np.isfinite(float(z['loss']))
    nan = np.isnan(z['dlogits'])
    assert np.array_equal(np.isnan(d), nan)
    np.testing.assert_allclose(d[~nan], z['dlogits'][~nan], rtol=1e-4, atol=2e-7)
