"""CTC loss path of the recognition trainer ("next" row N1): the HIP kernels vs golden vectors from the reference's
loss lines (recognition_model.py:96-101 run on torch CPU) and vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle.ctc_ref import ctc_loss_packed
from silent_speech_amd import recognition_model as rm
from tests.backend import dev, is_emu  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _run(z, dev):
    lengths = [int(n) for n in z['lengths']]
    text = [torch.from_numpy(z['text/%d' % i]) for i in range(len(lengths))]
    logits = torch.from_numpy(z['logits']).to(dev).requires_grad_(True)
    loss, plan = rm.ctc_loss(logits, dict(lengths=lengths, text_int=text), blank=int(z['blank']), return_plan=True)
    loss.backward()
    return float(loss.detach()), logits.grad.cpu().numpy(), plan.nll.cpu().numpy()


@pytest.mark.parametrize('tag', ['ctc_mixed', 'ctc_repeats', 'ctc_long'])
def test_ctc_loss_golden(dev, tag):
    z = np.load(os.path.join(GOLD, tag + '.npz'))
    loss, d, nll = _run(z, dev)
    np.testing.assert_allclose(nll, z['nll'], rtol=1e-5)
    assert abs(loss - float(z['loss'])) < 1e-5 * abs(float(z['loss']))
    # f32 log-domain sums of ~1e3 carry ~1e-4 absolute error into exp(): the long case needs the looser floor
    np.testing.assert_allclose(d, z['dlogits'], rtol=5e-3, atol=4e-6 if tag == 'ctc_long' else 1e-6)
    used = int(z['lengths'].sum())
    assert not d.reshape(-1, d.shape[-1])[used:].any()                       # packed padding frames get no gradient


def test_ctc_infeasible_is_inf_and_nan_like_aten(dev):
    z = np.load(os.path.join(GOLD, 'ctc_infeasible.npz'))
    loss, d, nll = _run(z, dev)
    assert np.isinf(loss) and np.isinf(float(z['loss']))
    assert np.array_equal(np.isinf(nll), np.isinf(z['nll']))
    nan = np.isnan(z['dlogits'])
    assert np.array_equal(np.isnan(d), nan)
    np.testing.assert_allclose(d[~nan], z['dlogits'][~nan], rtol=2e-3, atol=1e-6)


def test_ctc_random_vs_oracle(dev):
    rng = np.random.default_rng(3)
    V, blank, row = 11, 4, 24                                                # blank in the middle of the alphabet
    lengths = [17, 1, 30, 23]
    tlens = [6, 1, 0, 11]
    rows = (sum(lengths) + row - 1) // row
    logits = (1.5 * rng.standard_normal((rows, row, V))).astype(np.float32)
    text = [rng.choice([c for c in range(V) if c != blank], n).astype(np.int64) for n in tlens]
    want_loss, want_d, want_nll = ctc_loss_packed(logits, lengths, text, blank)
    x = torch.from_numpy(logits).to(dev).requires_grad_(True)
    loss, plan = rm.ctc_loss(x, dict(lengths=lengths, text_int=[torch.from_numpy(t) for t in text]), blank=blank, return_plan=True)
    (2.0 * loss).backward()
    np.testing.assert_allclose(plan.nll.cpu().numpy(), want_nll, rtol=1e-5)
    assert abs(float(loss.detach()) - want_loss) < 1e-5 * abs(want_loss)
    np.testing.assert_allclose(x.grad.cpu().numpy(), 2.0 * want_d, rtol=2e-3, atol=1e-6)


def test_ctc_target_too_long_raises(dev):
    x = torch.zeros(1, 8, 5, device=dev)
    with pytest.raises(RuntimeError):
        rm.ctc_loss(x, dict(lengths=[8], text_int=[torch.zeros(600, dtype=torch.long)]), blank=4)


def test_greedy_decode_and_wer(dev):
    V, blank = 6, 5
    path = [5, 1, 1, 5, 1, 2, 2, 5, 5, 3, 0, 0, 4, 4, 5, 2]                  # utterances of 9 and 7 frames
    x = torch.full((2, 8, V), -3.0)
    for i, c in enumerate(path):
        x[i // 8, i % 8, c] = 2.0
    out = rm.greedy_decode(x.to(dev), [9, 7], blank)
    assert out == [[1, 1, 2], [3, 0, 4, 2]]
    assert rm.wer(['a b c', 'd e'], ['a x c', 'd e f']) == pytest.approx(2 / 5)
    tt = rm.TextTransform()
    assert tt.int_to_text(tt.text_to_int('Hi, 2 You!')) == 'hi 2 you'


# ---------------------------------------------------------------- the recognition trainer's inner loop
class _FixedShift(object):
    def __init__(self, r):
        self.r = r

    def randrange(self, n):
        return self.r


def _recog_steps(dev, dt):
    """recognition_model.py:86-108 on the golden's two batches: forward, CTC loss, accumulate, one AdamW step."""
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.data_utils import combine_fixed_length
    from silent_speech_amd.optim import FusedAdamW
    z = np.load(os.path.join(GOLD, 'recog_d16_L1_T40.npz'))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    m = Model(112, 38, model_size=16, num_layers=1, dropout=0.0, compute_dtype=dt)
    m.load_state_dict(sd, strict=True)
    m.to(dev).train()
    m.shift_rng = _FixedShift(int(z['r']))
    optim = FusedAdamW(m, weight_decay=0.0)
    optim.zero_grad()
    out = dict(pred=[], loss=[], grad0=None)
    for b in range(2):
        for pg in optim.param_groups:
            pg['lr'] = (b + 1) * 3e-4 / 1000
        lengths = [int(n) for n in z['lengths/%d' % b]]
        raw = [torch.from_numpy(z['raw/%d/%d' % (b, i)]).to(dev) for i in range(len(lengths))]
        text = [torch.from_numpy(z['text/%d/%d' % (b, i)]) for i in range(len(lengths))]
        X_raw = combine_fixed_length(raw, 40 * 8)
        rows = X_raw.shape[0]
        pred = m(torch.zeros(rows, 40, 112, device=dev), X_raw, torch.zeros(rows, 40, dtype=torch.long, device=dev))
        loss = rm.ctc_loss(pred, dict(lengths=lengths, text_int=text), blank=37)
        loss.backward()
        out['pred'].append(pred.detach().float().cpu().numpy())
        out['loss'].append(float(loss.detach()))
        if b == 0:
            out['grad0'] = {n: p.grad.detach().clone().cpu() for n, p in m.named_parameters() if p.grad is not None}
    out['grad'] = {n: p.grad.detach().clone().cpu() for n, p in m.named_parameters() if p.grad is not None}
    optim.step()
    out['after'] = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return z, out


def _is_bn_bias(n):
    return n.endswith('.bias') and ('conv1' in n or 'conv2' in n or 'residual_path' in n)


def test_recognition_two_batch_accumulation_and_step(dev):
    from tests.util import assert_close_robust
    z, out = _recog_steps(dev, torch.float32)
    for b in range(2):
        assert_close_robust(out['pred'][b], z['pred/%d' % b], 2e-4, name='pred%d' % b, max_outlier_frac=0)
        assert abs(out['loss'][b] - float(z['loss/%d' % b])) < 2e-4 * abs(float(z['loss/%d' % b]))
    for key in ('grad0', 'grad'):                                    # after one backward, and accumulated over two
        for n, g in out[key].items():
            if 'relative_positional' in n or _is_bn_bias(n):
                continue
            assert_close_robust(g, z[key + '/' + n], 3e-3, name=key + ':' + n, min_outliers=40, max_outlier_frac=2e-3)
    lr = float(z['lr'])
    for k, v in out['after'].items():
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(z['after/' + k])
        elif 'running' in k:
            assert_close_robust(v, z['after/' + k], 1e-4, name=k, max_outlier_frac=0)
        elif not _is_bn_bias(k):
            # AdamW's first step moves every weight by ~lr * sign(grad): compare the UPDATE, allowing sign flips where |grad| is noise
            want = z['after/' + k] - z['sd/' + k]
            got = v.numpy() - z['sd/' + k]
            bad = np.abs(got - want) > 0.05 * lr
            assert bad.mean() < 0.02, (k, bad.mean())


@pytest.mark.gpu
def test_recognition_bf16_tracks_reference():
    from silent_speech_amd import _lib
    _lib.load()
    z, out = _recog_steps(torch.device('cuda'), torch.bfloat16)
    for b in range(2):
        assert abs(out['loss'][b] - float(z['loss/%d' % b])) < 2e-2 * abs(float(z['loss/%d' % b]))


@pytest.mark.gpu
def test_recognition_step_at_full_size_vs_oracle():
    """BASELINE configs[4] at its real size: 768-d / 6-layer encoder with a 38-way output on one 128 000-sample batch (~55 rows of 200
    frames, ~20 utterances of up to 860 frames with ~T/6 labels each).  Oracle: oracle/model_ref.model_forward + the reference's own
    loss lines on torch CPU (recognition_model.py:96-101), two of the utterances re-checked with the f64 recursion of oracle/ctc_ref.py.
    f32 kernels: logits 2e-4, loss 1e-4, every gradient tensor 3e-3 relative L2 / cosine 0.99999; f32 storage with bf16 x 3 products: the same
    forward bars, gradients 1.5e-2 / 0.9999; bf16: loss within 2 %, recorded."""
    import json
    import torch.nn.functional as F
    from oracle import ctc_ref, loss_ref, model_ref
    from silent_speech_amd import _lib
    from silent_speech_amd.architecture import Model
    from silent_speech_amd.synthetic import reference_size_batch
    from silent_speech_amd.transduction_model import _pack_batch
    from tests.util import assert_close_robust, rel_l2_cos
    _lib.load()
    dev = torch.device('cuda')
    torch.manual_seed(4)
    m0 = Model(112, 38, model_size=768, num_layers=6, dropout=0.0, compute_dtype=torch.float32)
    sd = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    batch = reference_size_batch(seed=3, budget=128000)
    lens, text = batch['lengths'], batch['text_int']
    # ---- oracle
    ref = {k: v.clone() for k, v in sd.items()}
    for v in ref.values():
        if v.dtype == torch.float32:
            v.requires_grad_(True)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    xr = loss_ref.combine_fixed_length(batch['raw_emg'], 1600)
    logits_ref = model_ref.model_forward(ref, xr, training=True, shift_r=3, running_out={})
    lp = F.log_softmax(logits_ref, 2)
    lp = torch.nn.utils.rnn.pad_sequence(loss_ref.decollate_tensor(lp, lens), batch_first=False)
    y = torch.nn.utils.rnn.pad_sequence(text, batch_first=True)
    loss_ref_v = F.ctc_loss(lp, y, lens, [int(t.shape[0]) for t in text], blank=37)
    loss_ref_v.backward()
    flat = F.log_softmax(logits_ref.detach().double(), 2).reshape(-1, 38).numpy()
    off = 0
    for i, (T, t) in enumerate(zip(lens, text)):                  # torch's CTC == the f64 Graves recursion on the two shortest utterances
        if T <= sorted(lens)[1]:
            nll, _ = ctc_ref.ctc_utterance(flat[off:off + T], t.numpy(), 37)
            want = float(F.ctc_loss(torch.from_numpy(flat[off:off + T]).unsqueeze(1), t.unsqueeze(0), [T], [int(t.shape[0])], blank=37, reduction='sum'))
            assert abs(nll - want) < 1e-6 * abs(want)
        off += T
    rec = {}
    for name, dt, mm in (('fp32', torch.float32, 'exact'), ('fp32_bf16x3', torch.float32, 'bf16x3'), ('bf16', torch.bfloat16, 'exact')):
        m = Model(112, 38, model_size=768, num_layers=6, dropout=0.0, compute_dtype=dt, f32_matmul=mm)
        m.load_state_dict(sd, strict=True)
        m.to(dev).train()
        m.shift_rng = _FixedShift(3)
        X, X_raw, sess = _pack_batch(batch, dev)
        pred = m(X, X_raw, sess)
        loss = rm.ctc_loss(pred, batch, blank=37)
        loss.backward()
        torch.cuda.synchronize()
        figs = {n: rel_l2_cos(p.grad, ref[n].grad) for n, p in m.named_parameters() if not ('relative_positional' in n or _is_bn_bias(n))}
        worst = max(figs, key=lambda n: figs[n][0])
        rec[name] = {'loss': float(loss), 'loss_oracle': float(loss_ref_v), 'rows': int(pred.shape[0]), 'utterances': len(lens),
                     'logit_max_err_over_max': float((pred.detach().float().cpu() - logits_ref.detach()).abs().max() / logits_ref.detach().abs().max()),
                     'worst_grad': worst, 'worst_rel_l2': figs[worst][0], 'min_cos': min(c for _, c in figs.values())}
        if dt == torch.float32:
            assert_close_robust(pred, logits_ref.detach(), 2e-4, name='logits', max_outlier_frac=0)
            assert abs(float(loss) - float(loss_ref_v)) < 1e-4 * abs(float(loss_ref_v))
            g_l2, g_cos = (3e-3, 0.99999) if mm == 'exact' else (1.5e-2, 0.9999)          # bf16 x 3 products: the conv-stack tensors sit 3-4 x above the exact kernels
            for n, (rl2, cos) in figs.items():     # 3e-3: half as many frames as the transduction batch average the ReLU-kink flips of the conv stack less (measured 2.1e-3 on conv_blocks.0.conv1)
                assert rl2 <= g_l2 and cos >= g_cos, (n, rl2, cos)
        else:
            assert abs(float(loss) - float(loss_ref_v)) < 2e-2 * abs(float(loss_ref_v))
    out = os.path.join(os.path.dirname(__file__), '..', 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    json.dump(rec, open(os.path.join(out, 'ctc_fullsize_parity.json'), 'w'), indent=1)
    print('ctc fullsize parity:', json.dumps(rec))
