"""Dropout parity: the mode the benchmark runs (p = 0.2) against the oracle ON THE SAME MASKS.

The kernels' keep decisions are a pure function of (seed, stream, element); `oracle/dropout_ref.py` restates them in numpy.
Here: (1) every dropout site's mask, read back from the kernel, equals the restated one bit for bit; (2) a whole training step
of the model with p > 0 equals `oracle/model_ref.model_forward(layer_masks=...)` -- which applies the masks where the reference
applies nn.Dropout (transformer.py:33,38-39,109) -- in outputs and in every parameter gradient.  Emulator tier at toy sizes,
GPU tier at d_model 64..768 (tests/test_fullsize.py repeats it for the full benchmark batch)."""
import math

import numpy as np
import pytest
import torch

from oracle import dropout_ref, model_ref
from silent_speech_amd import _lib, ops
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust, rel_l2_cos


# ------------------------------------------------------------------ (1) the masks, site by site
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p', [0.2, 0.5])
def test_rowwise_mask_of_add_dropout_layernorm(dev, dt, p):
    rows, C = (9, 64) if is_emu(dev) else (333, 768)
    seed, stream = 0x1234567890ABCDEF, 5
    x = torch.zeros(rows, C, dtype=dt, device=dev)
    a = torch.ones(rows, C, dtype=dt, device=dev)
    y = torch.empty(rows, C, dtype=dt, device=dev)
    ops.add_dropout_layernorm(x, a, torch.ones(C, device=dev), torch.zeros(C, device=dev), y, rows, C, p=p, seed=seed, rng_stream=stream)
    keep = (a.float().cpu().numpy() != 0)                               # a now holds z = x + dropout(a)
    want = dropout_ref.rowwise_mask(seed, stream, rows, C, p)
    assert np.array_equal(keep, want)
    assert abs(keep.mean() - (1 - p)) < (0.08 if is_emu(dev) else 0.01)
    assert np.allclose(a.float().cpu().numpy()[keep], 1.0 / (1.0 - p), rtol=1e-2)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_epilogue_mask(dev, dt):
    """relu(A B^T + bias) with dropout in the epilogue (the FFN hidden layer, transformer.py:38): C != 0 <=> kept."""
    shapes = [(70, 48, 32)] if is_emu(dev) else [(70, 48, 32), (4098, 3072, 64)]          # the large one runs the 8-wave kernel
    seed, stream, p = 77, 14, 0.2
    for M, N, K in shapes:
        A = torch.ones(M, K, dtype=dt, device=dev); B = torch.ones(N, K, dtype=dt, device=dev) / K
        C = torch.zeros(M, N, dtype=dt, device=dev)
        ops.gemm(A, B, C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), relu=True, dropout_p=p, seed=seed, rng_stream=stream)
        keep = C.float().cpu().numpy() != 0
        want = dropout_ref.gemm_epilogue_mask(seed, stream, M, N, p)
        assert np.array_equal(keep, want), (M, N, K, _lib.lib().ss_gemm_last_kernel())
        assert np.allclose(C.float().cpu().numpy()[keep], 1.0 / (1.0 - p), rtol=1e-2)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_mask(dev, dt):
    """V = I: the output IS the dropped-out probability matrix.  f32 runs the per-tile kernels, bf16 the LDS-resident ones; they
    draw from different hashes (dropout_ref.attention_mask_tiled / _resident)."""
    B, H, T, dh, D, p = 2, 2, 32, 32, 9, 0.3
    dp, Tp = 32, 32
    seed, stream = 99, 4
    g = torch.Generator().manual_seed(3)
    q, k = [(torch.randn(B, H, T, dh, generator=g) * 0.3).to(dt) for _ in range(2)]
    v = torch.eye(T).expand(B, H, T, T).to(dt)
    pack = lambda x: x.permute(0, 2, 1, 3).reshape(B * T, H * dp)
    qkv = torch.cat([pack(q), pack(k), pack(v)], 1).contiguous()
    qkvT = qkv.view(B, T, 3 * H * dp).transpose(1, 2).contiguous()
    E = (torch.randn(H, 2 * D - 1, dp, generator=g) * 0.1).to(dt)
    out = torch.zeros(B * T, H * dp, dtype=dt, device=dev); lse = torch.zeros(B, H, T, device=dev)
    ops.relpos_attention_forward(qkv.to(dev), qkvT.to(dev), E.to(dev), out, lse, B, H, T, Tp, dp, D, 1 / math.sqrt(dh), p=p, seed=seed, rng_stream=stream)
    keep = (out.float().cpu().view(B, T, H, dp).permute(0, 2, 1, 3) != 0).numpy()
    family = _lib.lib().ss_relpos_attention_family(_lib.dtype_code(dt), T, dp, D)
    assert (family > 0) == (dt == torch.bfloat16)
    want = dropout_ref.attention_mask(family, seed, stream, B, H, T, p)
    band = np.abs(np.arange(T)[None, :] - np.arange(T)[:, None]) <= D - 1
    assert np.array_equal(keep[..., band], want[..., band])
    assert not keep[..., ~band].any()


@pytest.mark.parametrize('p', [0.05, 0.1, 0.2, 0.25, 0.5])
def test_attention_mask_transposed_drop_rate_per_key_position(p):
    """The four keys 4g .. 4g+3 of a query share one 64-bit product; every one of the four 16-bit draws must be uniform, i.e. the drop rate per key
    position mod 4 equals p (round 5 drew keys 4g+2, 4g+3 from the upper product word alone: 0.191 instead of 0.2 on every fourth key, 0.081 instead
    of 0.05).  The oracle function is what the kernel's masks are compared with bit for bit (test_attention_mask), so its statistics are the kernel's.
    480 000 draws per position: 5 sigma of a binomial plus the 2^-16 granularity of the threshold."""
    keep = dropout_ref.attention_mask_transposed(1234, 7, 6, 8, 200, p)
    n = keep[..., 0::4].size
    tol = 5.0 * math.sqrt(p * (1 - p) / n) + 2.0 ** -15
    for r in range(4):
        rate = 1.0 - keep[..., r::4].mean()
        assert abs(rate - p) < tol, (r, rate, p)
    # neighbouring keys of one product are not copies of each other
    a, b = keep[..., 2::4].ravel(), keep[..., 3::4].ravel()
    both = float((~a & ~b).mean())
    assert abs(both - p * p) < 5.0 * math.sqrt(p * p / a.size) + 1e-3, both


# ------------------------------------------------------------------ (2) a whole training step at p > 0
class _FixedShift(object):
    @staticmethod
    def randrange(n):
        return 5


def model_step_with_dropout(dev, dt, d, L, B, T, p, seed_base=0xD5, nthreads=None):
    """Runs one training forward + backward of Model(dropout=p) and of the oracle on the kernel's masks.
    Returns (model, pred, aux, oracle pred, oracle aux, oracle grads, bf16-storage yardstick grads or None)."""
    from silent_speech_amd.architecture import Model
    torch.manual_seed(d + L)
    m = Model(112, 80, 48, model_size=d, num_layers=L, dropout=p, compute_dtype=dt)
    with torch.no_grad():                                   # non-trivial norm parameters (init is 1 / 0)
        for n, q in m.named_parameters():
            if 'norm' in n or 'bn' in n:
                q.add_(torch.randn_like(q) * 0.1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(dev)
    m.shift_rng = _FixedShift
    m.set_seed(seed_base)
    m.train()
    g = torch.Generator().manual_seed(B * T)
    x_raw = torch.randn(B, 8 * T, 8, generator=g) * 3.0
    wp = torch.randn(B, T, 80, generator=g); wa = torch.randn(B, T, 48, generator=g)
    xd = x_raw.clone().to(dev)
    pred, aux = m(None, xd, None)
    ((pred * wp.to(dev)).sum() + (aux * wa.to(dev)).sum()).backward()
    seed = m.last_seed
    family = m.attention_mask_family(T)
    masks = dropout_ref.layer_masks(seed, L, B, T, d, 8, 3072, p, family)
    ref = {k: v.clone() for k, v in sd.items()}
    for v in ref.values():
        if v.dtype == torch.float32:
            v.requires_grad_(True)
    if nthreads:
        torch.set_num_threads(nthreads)
    pr, ar = model_ref.model_forward(ref, x_raw.clone(), training=True, shift_r=5, running_out={}, layer_masks=masks, dropout_p=p)
    ((pr * wp).sum() + (ar * wa).sum()).backward()
    grads = {k: v.grad for k, v in ref.items() if v.dtype == torch.float32 and v.grad is not None}
    yard = None
    if dt == torch.bfloat16:            # the yardstick: the SAME oracle with bf16 storage between its ops (model_ref.bf16_storage)
        for v in ref.values():
            v.grad = None
        with model_ref.bf16_storage():
            py, ay = model_ref.model_forward(ref, x_raw.clone(), training=True, shift_r=5, running_out={}, layer_masks=masks, dropout_p=p)
            ((py * wp).sum() + (ay * wa).sum()).backward()
        yard = {k: v.grad for k, v in ref.items() if v.dtype == torch.float32 and v.grad is not None}
        yard['pred'] = py.detach()
    return m, pred.detach().float().cpu(), aux.detach().float().cpu(), pr.detach(), ar.detach(), grads, yard


def _is_bn_fed_bias(n):
    return n.endswith('.bias') and ('conv1' in n or 'conv2' in n or 'residual_path' in n)


def yardstick_levels(yard, grads, names):
    """Relative-L2 level of the bf16-storage yardstick per tensor: its own figure, but not below the mean of its group (conv stack
    / encoder / heads) -- with few frames the number of flipped ReLU gates behind one tensor is a small, noisy count."""
    own = {n: rel_l2_cos(yard[n], grads[n])[0] for n in names}
    group = lambda n: n.split('.')[0]
    mean = {}
    for g in set(group(n) for n in names):
        v = [own[n] for n in names if group(n) == g]
        mean[g] = sum(v) / len(v)
    return {n: max(own[n], mean[group(n)]) for n in names}


def _check_step(dev, dt, d, L, B, T, p):
    """f32 kernels: every gradient tensor within 2e-3 relative L2 / cosine 0.99999 of the oracle.  bf16 kernels: judged against
    the yardstick -- the oracle itself with bf16 storage between its ops sits 5-20 % (relative L2) from the f32 oracle on the
    ReLU-gated tensors (gates flip where a pre-activation is within bf16 rounding of zero); our deviation must not exceed twice
    the yardstick's (+1 %), tensor by tensor."""
    m, pred, aux, pr, ar, grads, yard = model_step_with_dropout(dev, dt, d, L, B, T, p)
    f32 = dt == torch.float32
    assert_close_robust(pred, pr, 2e-4 if f32 else 6e-2, name='pred', max_outlier_frac=0 if f32 else 1e-3)
    assert_close_robust(aux, ar, 2e-4 if f32 else 6e-2, name='aux', max_outlier_frac=0 if f32 else 1e-3)
    if f32:
        assert float((pred - pr).abs().mean()) < 1e-4
    else:
        assert float((pred - pr).abs().mean()) <= 2.0 * float((yard['pred'] - pr).abs().mean()) + 1e-3
    names = [n for n, _ in m.named_parameters() if not ('relative_positional' in n or _is_bn_fed_bias(n))]
    ylev = yardstick_levels(yard, grads, names) if not f32 else None
    for n, q in m.named_parameters():
        if n not in names:
            continue
        rl2, cos = rel_l2_cos(q.grad, grads[n])
        if f32:
            assert rl2 <= 2e-3 and cos >= 0.99999, (n, rl2, cos)
        else:
            assert rl2 <= 2.0 * ylev[n] + 1e-2, (n, rl2, cos, 'yardstick', ylev[n])


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_model_step_with_dropout_tiny(dev, dt):
    _check_step(dev, dt, d=16, L=1, B=2, T=40, p=0.2)


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cfg', [(64, 2, 3, 200, 0.2), (256, 2, 4, 200, 0.1), (32, 1, 1, 264, 0.3)])
def test_model_step_with_dropout_gpu(dt, cfg):
    """T = 200: the training rows (bf16 -> resident attention, saved-probability backward); T = 264: longer than the resident
    limit (per-tile kernels in both dtypes)."""
    _lib.load()
    d, L, B, T, p = cfg
    _check_step(torch.device('cuda'), dt, d, L, B, T, p)


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_model_shape_with_dropout(dt):
    """d_qkv = 96, D = 100, T = 200, H = 8, p = 0.2: forward and both backward forms (saved probabilities / recomputation) vs the
    closed form with the restated mask."""
    from tests.test_attention import _run
    _lib.load()
    _run(torch.device('cuda'), dt, B=3, H=8, T=200, dh=96, D=100, seed=20, tol_f=3e-5 if dt == torch.float32 else 2e-2,
         tol_b=1e-4 if dt == torch.float32 else 3e-2, p=0.2)
