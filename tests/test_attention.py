"""Fused relative-position attention (forward + backward) vs the closed-form fp32 torch restatement
(oracle/model_ref.relpos_logits, itself pinned to the reference's pad/view skew by golden vectors)."""
import math

import pytest
import torch

from oracle import model_ref
from silent_speech_amd import ops
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust


def _reference(q, k, v, E, D, dh, drop=None):
    """q,k,v: (B,H,T,dh) f32 leaf tensors; E: (H,2D-1,dh)."""
    logits = torch.einsum('bhqa,bhka->bhqk', q, k) / math.sqrt(dh)
    logits = logits + model_ref.relpos_logits(q, E[..., None], max_rel=D)
    P = torch.softmax(logits, -1)
    if drop is not None:
        P = P * drop
    return torch.einsum('bhqk,bhka->bhqa', P, v), torch.logsumexp(logits, -1)


def _pack(x, dp):          # (B,H,T,dh) -> (B,T,H,dp) zero padded
    B, H, T, dh = x.shape
    o = torch.zeros(B, T, H, dp, dtype=x.dtype)
    o[..., :dh] = x.permute(0, 2, 1, 3)
    return o


def _run(dev, dt, B, H, T, dh, D, seed, tol_f, tol_b, p=0.0, f32_math='exact'):
    """p > 0: dropout on the probabilities with the kernels' own mask, restated by oracle/dropout_ref.py."""
    dp = (dh + 31) // 32 * 32
    Tp = (T + 7) // 8 * 8
    g = torch.Generator().manual_seed(seed)
    q, k, v = [(torch.randn(B, H, T, dh, generator=g) * 0.8).to(dt).float().requires_grad_(True) for _ in range(3)]
    E = (torch.randn(H, 2 * D - 1, dh, generator=g) * dh ** -0.5).to(dt).float()
    dO = torch.randn(B, H, T, dh, generator=g).to(dt).float()
    drop, kw = None, {}
    if f32_math != 'exact':
        kw['f32_math'] = f32_math
    if p > 0:
        from oracle import dropout_ref
        from silent_speech_amd import _lib
        family = _lib.lib().ss_relpos_attention_family(_lib.dtype_code(dt), T, dp, D) if f32_math == 'exact' else 0
        mask = dropout_ref.attention_mask(family, seed + 1000, 8, B, H, T, p)
        drop = torch.from_numpy(mask).float() / (1.0 - p)
        kw.update(p=p, seed=seed + 1000, rng_stream=8)
    O_ref, lse_ref = _reference(q, k, v, E, D, dh, drop=drop)
    O_ref.backward(dO)
    # device operands
    qkv = torch.cat([_pack(t.detach(), dp).reshape(B * T, H * dp) for t in (q, k, v)], 1).to(dt).contiguous()          # [B*T][3*H*dp]
    qkvT = torch.zeros(B, 3 * H * dp, Tp, dtype=dt)
    qkvT[:, :, :T] = qkv.view(B, T, 3 * H * dp).transpose(1, 2)
    Ed = torch.zeros(H, 2 * D - 1, dp, dtype=dt); Ed[..., :dh] = E.to(dt)
    MPt = (2 * D - 1 + 31) // 32 * 32
    ETd = torch.zeros(H, dp, MPt, dtype=dt); ETd[:, :dh, :2 * D - 1] = E.transpose(1, 2).to(dt)
    out = torch.zeros(B * T, H * dp, dtype=dt, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    qkv_d, qkvT_d, E_d, ET_d = qkv.to(dev), qkvT.to(dev), Ed.to(dev), ETd.to(dev)
    scale = 1.0 / math.sqrt(dh)
    nsaved = ops.relpos_attention_saved_bytes(dt, B, H, T, dp, D)
    saved = torch.empty(nsaved, dtype=torch.uint8, device=dev) if nsaved else None             # the probability image of the resident kernels
    ops.relpos_attention_forward(qkv_d, qkvT_d, E_d, out, lse, B, H, T, Tp, dp, D, scale, saved=saved, **kw)
    O = out.view(B, T, H, dp)[..., :dh].permute(0, 2, 1, 3)
    assert_close_robust(O, O_ref, tol_f, name='O', max_outlier_frac=0)
    assert_close_robust(lse, lse_ref, (1e-5 if f32_math == 'exact' else 1e-4) if dt == torch.float32 else 2e-2, name='lse', max_outlier_frac=0)
    if dp > dh:
        assert float(out.view(B, T, H, dp)[..., dh:].float().abs().max()) == 0.0          # padded head dims stay zero
    # backward
    dOd = _pack(dO, dp).reshape(B * T, H * dp).to(dt).contiguous()
    dOT = torch.zeros(B, H * dp, Tp, dtype=dt); dOT[:, :, :T] = dOd.view(B, T, H * dp).transpose(1, 2)
    dqkv = torch.full((B * T, 3 * H * dp), 7.0, dtype=dt, device=dev)
    dsc = torch.empty(B, H, T, device=dev)
    family = ops.relpos_attention_family(dt, T, dp, D) if f32_math == 'exact' else 0
    # backward from the saved probabilities, and recomputing them (the transposed-score kernels, family 2, work from the image only)
    for sv in ([saved] if family == 2 else [saved, None] if saved is not None else [None]):
        dqkv.fill_(7.0)
        ops.relpos_attention_backward(qkv_d, qkvT_d, E_d, ET_d, out, lse, dOd.to(dev), dOT.to(dev), dsc, dqkv, B, H, T, Tp, dp, D, scale, saved=sv, **kw)
        dq, dk, dv = [dqkv.view(B, T, 3, H, dp)[:, :, i, :, :dh].permute(0, 2, 1, 3) for i in range(3)]
        tag = ' (saved P)' if sv is not None else ''
        assert_close_robust(dv, v.grad, tol_b, name='dV' + tag, max_outlier_frac=0)
        assert_close_robust(dk, k.grad, tol_b, name='dK' + tag, max_outlier_frac=0)
        assert_close_robust(dq, q.grad, tol_b, name='dQ' + tag, max_outlier_frac=0)
        if dp > dh:
            assert float(dqkv.view(B, T, 3, H, dp)[..., dh:].float().abs().max()) == 0.0


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_small_band(dev, dt):
    """T > D: banded; ragged T (not a multiple of 16 or 8); padded head dim."""
    if is_emu(dev):
        _run(dev, dt, B=1, H=2, T=37, dh=8, D=9, seed=1, tol_f=2e-5 if dt == torch.float32 else 2e-2, tol_b=5e-5 if dt == torch.float32 else 3e-2)
    else:
        _run(dev, dt, B=2, H=3, T=77, dh=24, D=21, seed=1, tol_f=2e-5 if dt == torch.float32 else 2e-2, tol_b=5e-5 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_no_band(dev, dt):
    """T <= D: no masking at all (transformer.py:256 branch not taken)."""
    T = 20 if is_emu(dev) else 50
    _run(dev, dt, B=1, H=1, T=T, dh=32, D=100 if not is_emu(dev) else 24, seed=2, tol_f=2e-5 if dt == torch.float32 else 2e-2, tol_b=5e-5 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize('case', [(37, 8, 9), (70, 32, 100), (65, 32, 20), (97, 64, 33)])
@pytest.mark.parametrize('p', [0.0, 0.25])
def test_attention_transposed_score_kernels(dev, case, p):
    """Family 2 (csrc/attention_t.hip): one wave per 32-query tile, S^T = K Q^T on 32 x 32 x 16 MFMAs, the skew through LDS with a per-lane
    offset, the embedding table E / scale in fragment order, the backward from the saved probabilities (query-major: dQ and D; key-major: dK,
    dV).  Ragged T, several tiles, banded and unbanded, padded head dims, with and without dropout (the kernels' own mask, restated in
    oracle/dropout_ref.attention_mask_transposed)."""
    T, dh, D = case
    if not is_emu(dev):
        T, dh, D = {37: (200, 96, 100), 70: (224, 96, 100), 65: (209, 64, 40), 97: (131, 96, 17)}[T]
    dp = (dh + 31) // 32 * 32
    assert ops.relpos_attention_family(torch.bfloat16, T, dp, D) == 2
    _run(dev, torch.bfloat16, B=2, H=2 if is_emu(dev) else 8, T=T, dh=dh, D=D, seed=T + D, tol_f=2e-2, tol_b=3e-2, p=p)


def _need_res16(monkeypatch, T, dp, D):
    """The LDS-resident 16 x 16 family (rounds 1-4) lives only in A/B builds of the library (-DSS_ATTN_RES16, tools/measure_lib.sh) since round 6."""
    monkeypatch.setenv('SS_ATTN_T', '0')
    if ops.relpos_attention_family(torch.bfloat16, T, dp, D) != 1:
        pytest.skip('the 16 x 16 resident attention kernels are compiled only into A/B builds (-DSS_ATTN_RES16)')



@pytest.mark.parametrize('dt_old', [torch.bfloat16])
def test_attention_resident_16x16_kernels_still_agree(dev, monkeypatch, dt_old):
    """SS_ATTN_T=0 keeps the LDS-resident 16 x 16 kernels of rounds 1-4 reachable (A/B measurements): same function."""
    dev_T, dev_dp, dev_D = (37, 32, 9) if is_emu(dev) else (200, 96, 100)
    _need_res16(monkeypatch, dev_T, dev_dp, dev_D)
    if is_emu(dev):
        _run(dev, dt_old, B=1, H=2, T=37, dh=8, D=9, seed=1, tol_f=2e-2, tol_b=3e-2, p=0.25)
    else:
        _run(dev, dt_old, B=2, H=8, T=200, dh=96, D=100, seed=1, tol_f=2e-2, tol_b=3e-2, p=0.2)


@pytest.mark.parametrize('p', [0.0, 0.25])
def test_attention_f32_storage_bf16x3_arithmetic(dev, p):
    """SS_F32X3 (the plan's parity-grade fast mode): f32 tensors, every product on three bf16 MFMAs.  Bars 10 x the exact-f32 ones
    (measured ~1e-5) and 100 x tighter than bf16's: a silent plain-bf16 path fails them."""
    if is_emu(dev):
        _run(dev, torch.float32, B=1, H=2, T=37, dh=8, D=9, seed=11, tol_f=2e-4, tol_b=5e-4, p=p, f32_math='bf16x3')
    else:
        _run(dev, torch.float32, B=2, H=8, T=200, dh=96, D=100, seed=11, tol_f=2e-4, tol_b=5e-4, p=p, f32_math='bf16x3')


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('T', [200, 250, 100])
def test_attention_model_shape(dt, T):
    """The real configuration: d_qkv = 96, D = 100, T = 200 (training rows) / 250 / 100."""
    from silent_speech_amd import _lib
    _lib.load()
    _run(torch.device('cuda'), dt, B=2, H=8, T=T, dh=96, D=100, seed=T, tol_f=3e-5 if dt == torch.float32 else 2e-2, tol_b=1e-4 if dt == torch.float32 else 3e-2)


def test_attention_dropout_statistics(dev):
    """P~ = P*keep/(1-p): with V = 1 every output equals sum_k P~ -> mean 1, and backward uses the same mask."""
    B, H, T, dh, D, p = 1, 1, 32, 32, 8, 0.25
    dp, Tp = 32, 32
    qkv = torch.zeros(B * T, 3 * dp); qkv[:, 2 * dp:] = 1.0
    qkvT = qkv.view(B, T, 3 * dp).transpose(1, 2).contiguous()
    E = torch.zeros(H, 2 * D - 1, dp)
    out = torch.zeros(B * T, dp, device=dev); lse = torch.zeros(B, H, T, device=dev)
    ops.relpos_attention_forward(qkv.to(dev), qkvT.to(dev), E.to(dev), out, lse, B, H, T, Tp, dp, D, 1.0, p=p, seed=5, rng_stream=2)
    o = out.cpu()[:, 0]
    # band of 2D-1 = 15 uniform probabilities, each kept w.p. 0.75 and scaled by 1/0.75
    assert abs(o.mean().item() - 1.0) < 0.15
    assert o.std().item() > 0.01


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_attention_backward_uses_the_forward_dropout_mask(dev, dt):
    """Recover the kept set from a forward pass with V = I (O reveals P~), then check the kernel backward against
    autograd of the closed form with exactly that mask (all three kernels regenerate the same Philox stream)."""
    B, H, T, dh, D, p = 1, 2, 32, 32, 9, 0.3
    dp, Tp = 32, 32
    g = torch.Generator().manual_seed(21)
    bf = dt == torch.bfloat16                                                           # bf16 rows of <= 208 frames: the LDS-resident kernels
    tolP, tolG = (2e-2, 3e-2) if bf else (2e-5, 5e-5)
    q, k = [(torch.randn(B, H, T, dh, generator=g) * 0.5).to(dt).float().requires_grad_(True) for _ in range(2)]
    v = torch.eye(T).expand(B, H, T, T).clone().requires_grad_(True)                    # dh == T
    E = (torch.randn(H, 2 * D - 1, dh, generator=g) * dh ** -0.5).to(dt).float()
    scale = 1.0 / math.sqrt(dh)
    qkv = torch.cat([_pack(t.detach(), dp).reshape(B * T, H * dp) for t in (q, k, v)], 1).to(dt).contiguous()
    qkvT = qkv.view(B, T, 3 * H * dp).transpose(1, 2).contiguous()
    Ed = E.clone().to(dt); MPt = (2 * D - 1 + 31) // 32 * 32
    ETd = torch.zeros(H, dp, MPt, dtype=dt); ETd[:, :, :2 * D - 1] = E.transpose(1, 2).to(dt)
    out = torch.zeros(B * T, H * dp, dtype=dt, device=dev); lse = torch.zeros(B, H, T, device=dev)
    qkv_d, qkvT_d, E_d, ET_d = qkv.to(dev), qkvT.to(dev), Ed.to(dev), ETd.to(dev)
    nsaved = ops.relpos_attention_saved_bytes(dt, B, H, T, dp, D)
    saved = torch.empty(nsaved, dtype=torch.uint8, device=dev) if nsaved else None
    ops.relpos_attention_forward(qkv_d, qkvT_d, E_d, out, lse, B, H, T, Tp, dp, D, scale, p=p, seed=99, rng_stream=4, saved=saved)
    Pd = out.cpu().float().view(B, T, H, dp).permute(0, 2, 1, 3)                        # (B,H,q,k) = P~
    band = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).abs() <= D - 1
    keep = (Pd != 0).float()
    frac = keep[..., band].mean().item()
    assert abs(frac - (1 - p)) < 0.08, frac
    O_ref, _ = _reference(q, k, v, E, D, dh, drop=keep / (1 - p))
    assert_close_robust(Pd, O_ref, tolP, name='P~', max_outlier_frac=0)
    dO = torch.randn(B, H, T, dh, generator=g).to(dt).float()
    O_ref.backward(dO)
    dOd = _pack(dO, dp).reshape(B * T, H * dp).to(dt).contiguous()
    dOT = dOd.view(B, T, H * dp).transpose(1, 2).contiguous()
    dqkv = torch.zeros(B * T, 3 * H * dp, dtype=dt, device=dev); dsc = torch.empty(B, H, T, device=dev)
    assert (saved is not None) == bf
    for sv in ([saved] if ops.relpos_attention_family(dt, T, dp, D) == 2 else [saved, None] if saved is not None else [None]):
        dqkv.zero_()
        ops.relpos_attention_backward(qkv_d, qkvT_d, E_d, ET_d, out, lse, dOd.to(dev), dOT.to(dev), dsc, dqkv, B, H, T, Tp, dp, D, scale, p=p, seed=99, rng_stream=4, saved=sv)
        dq, dk, dv = [dqkv.float().view(B, T, 3, H, dp)[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
        assert_close_robust(dv, v.grad, tolG, name='dV', max_outlier_frac=0)
        assert_close_robust(dk, k.grad, tolG, name='dK', max_outlier_frac=0)
        assert_close_robust(dq, q.grad, tolG, name='dQ', max_outlier_frac=0)


@pytest.mark.parametrize('T,D,dh,p', [(40, 9, 32, 0.0), (72, 30, 64, 0.25), (200, 100, 96, 0.2)])
def test_resident_forward_generations_agree(dev, monkeypatch, T, D, dh, p):
    """The hand-scheduled resident forward (default) and the compiler-scheduled one (SS_ATTN_FWD2=0) implement the same function:
    same log-sum-exp, same dropped set, outputs equal up to the bf16 rounding of the probabilities (normalised before vs after P~V)."""
    if is_emu(dev) and T > 100:
        pytest.skip('full-size rows: gpu tier')
    B, H, dt = 2, 2, torch.bfloat16
    dp, Tp = (dh + 31) // 32 * 32, (T + 7) // 8 * 8
    _need_res16(monkeypatch, T, dp, D)                   # the 16 x 16 resident kernels (rounds 1-4), kept behind this switch in A/B builds
    g = torch.Generator().manual_seed(7)
    qkv = (torch.randn(B * T, 3 * H * dp, generator=g) * 0.7).to(dt)
    E = (torch.randn(H, 2 * D - 1, dp, generator=g) * dh ** -0.5).to(dt)
    res = {}
    for gen in ('0', '1'):
        monkeypatch.setenv('SS_ATTN_FWD2', gen)
        out = torch.zeros(B * T, H * dp, dtype=dt, device=dev); lse = torch.zeros(B, H, T, device=dev)
        ops.relpos_attention_forward(qkv.to(dev), None, E.to(dev), out, lse, B, H, T, Tp, dp, D, 1.0 / math.sqrt(dh), p=p, seed=31, rng_stream=6)
        res[gen] = (out.float().cpu(), lse.cpu())
    assert_close_robust(res['1'][1], res['0'][1], 1e-5, name='lse', max_outlier_frac=0)
    assert_close_robust(res['1'][0], res['0'][0], 2e-2, name='O', max_outlier_frac=0)


@pytest.mark.parametrize('p,BH', [(0.0, (5, 2)), (0.2, (5, 2)), (0.2, (6, 2)), (0.0, (5, 1)), (0.0, (7, 1))])
def test_persistent_per_head_schedule_equals_one_workgroup_per_pair(dev, monkeypatch, p, BH):
    """More (sequence, head) pairs than CUs: the forward and the query-major backward run ONE persistent workgroup per CU that keeps its
    head's embedding table in LDS and walks several sequences (whole ones, and a half of one of the last, partial round); SS_ATTN_PERSIST=0
    is the old one-workgroup-per-pair launch.  Same arithmetic per pair: output, lse, the probability image and dqkv must be bit-identical."""
    # emulator: 4 "CUs".  (5, 2): 2 workgroups per head, 2 whole sequences each + one split in halves; (6, 2): no partial round; (5, 1): 4 workgroups,
    # two of them without a half; (7, 1): 3 left over of 4 -> not split (a second whole round for three workgroups).  GPU: the benchmarked launch.
    if not is_emu(dev) and BH != (5, 2):
        pytest.skip('emulator-size schedules')
    B, H, T, dh, D = (BH[0], BH[1], 40, 32, 9) if is_emu(dev) else (110, 8, 200, 96, 100)
    dt, dp, Tp = torch.bfloat16, (dh + 31) // 32 * 32, (T + 7) // 8 * 8
    _need_res16(monkeypatch, T, dp, D)                   # a schedule of the 16 x 16 resident kernels
    g = torch.Generator().manual_seed(17)
    qkv = (torch.randn(B * T, 3 * H * dp, generator=g) * 0.7).to(dt).to(dev)
    E = (torch.randn(H, 2 * D - 1, dp, generator=g) * dh ** -0.5).to(dt)
    MPt = (2 * D - 1 + 31) // 32 * 32
    ET = torch.zeros(H, dp, MPt, dtype=dt); ET[:, :, :2 * D - 1] = E.transpose(1, 2)
    E, ET = E.to(dev), ET.to(dev)
    dO = torch.randn(B * T, H * dp, generator=g).to(dt).to(dev)
    nsaved = ops.relpos_attention_saved_bytes(dt, B, H, T, dp, D)
    assert nsaved > 0
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('SS_ATTN_PERSIST', mode)
        out = torch.zeros(B * T, H * dp, dtype=dt, device=dev); lse = torch.zeros(B, H, T, device=dev)
        saved = torch.zeros(nsaved, dtype=torch.uint8, device=dev)
        ops.relpos_attention_forward(qkv, None, E, out, lse, B, H, T, Tp, dp, D, 1.0 / math.sqrt(dh), p=p, seed=5, rng_stream=2, saved=saved)
        dqkv = torch.zeros(B * T, 3 * H * dp, dtype=dt, device=dev); dsc = torch.empty(B, H, T, device=dev)
        ops.relpos_attention_backward(qkv, None, E, ET, out, lse, dO, None, dsc, dqkv, B, H, T, Tp, dp, D, 1.0 / math.sqrt(dh), p=p, seed=5, rng_stream=2, saved=saved)
        res[mode] = [t.cpu() for t in (out.view(torch.int16), lse, saved, dqkv.view(torch.int16))]
    for a, b, name in zip(res['0'], res['1'], ('O', 'lse', 'image', 'dqkv')):
        assert torch.equal(a, b), name
    assert float(res['1'][0].float().abs().max()) > 0 and float(res['1'][3].float().abs().max()) > 0


def test_transposed_copies_only_needed_by_the_per_tile_kernels(dev):
    """ss_relpos_attention_needs_transposed: bf16 rows of <= 208 frames run the LDS-resident kernels (qkvT / dOT may be NULL);
    f32 and long sequences run the per-tile kernels, which refuse to start without them."""
    from silent_speech_amd import _lib
    L = _lib.lib()
    BF16, F32 = _lib.dtype_code(torch.bfloat16), _lib.dtype_code(torch.float32)
    assert L.ss_relpos_attention_needs_transposed(BF16, 200, 96, 100) == 0
    assert L.ss_relpos_attention_needs_transposed(BF16, 40, 32, 100) == 0
    assert L.ss_relpos_attention_needs_transposed(BF16, 224, 96, 100) == 0
    assert L.ss_relpos_attention_needs_transposed(BF16, 225, 96, 100) == 1
    assert L.ss_relpos_attention_family(BF16, 200, 96, 100) == 2 and L.ss_relpos_attention_family(BF16, 225, 96, 100) == 0
    assert L.ss_relpos_attention_family(F32, 40, 32, 100) == 0
    assert L.ss_relpos_attention_needs_transposed(F32, 40, 32, 100) == 1
    assert L.ss_relpos_attention_needs_transposed(BF16, 200, 128, 100) == 1          # operands do not fit the LDS
    B, H, T, dp, D = 1, 2, 24, 32, 9
    for dt, ok in ((torch.bfloat16, True), (torch.float32, False)):
        qkv = torch.randn(B * T, 3 * H * dp).to(dt).to(dev)
        E = torch.randn(H, 2 * D - 1, dp).to(dt).to(dev)
        out = torch.zeros(B * T, H * dp, dtype=dt, device=dev); lse = torch.zeros(B, H, T, device=dev)
        if ok:
            ops.relpos_attention_forward(qkv, None, E, out, lse, B, H, T, 24, dp, D, 0.25)
            assert torch.isfinite(out.float()).all() and float(out.float().abs().max()) > 0
        else:
            with pytest.raises(RuntimeError, match='transposed copy'):
                ops.relpos_attention_forward(qkv, None, E, out, lse, B, H, T, 24, dp, D, 0.25)


# ------------------------------------------------------------------ x3 on the transposed-score kernels (round 6): f32 operands as hi / lo bf16 planes
def _run_x3_planes(dev, B, H, T, dh, D, seed, p=0.0):
    """ss_relpos_attention_x3_forward / _backward against the f64 closed form: qkv / dO in as plane pairs, O / dqkv out as plane pairs.  The error
    must be that of the bf16 x 3 arithmetic (about 1e-5 relative; a plain-bf16 path would sit at 1e-2)."""
    from oracle import dropout_ref
    dp = (dh + 31) // 32 * 32
    g = torch.Generator().manual_seed(seed)
    q, k, v = [(torch.randn(B, H, T, dh, generator=g) * 0.8).double().requires_grad_(True) for _ in range(3)]
    E = (torch.randn(H, 2 * D - 1, dh, generator=g) * dh ** -0.5)
    dO = torch.randn(B, H, T, dh, generator=g)
    drop, kw = None, {}
    if p > 0:
        mask = dropout_ref.attention_mask(2, seed + 1000, 8, B, H, T, p)          # the masks of kernel family 2
        drop = torch.from_numpy(mask).double() / (1.0 - p)
        kw.update(p=p, seed=seed + 1000, rng_stream=8)
    O_ref, lse_ref = _reference(q, k, v, E.double(), D, dh, drop=drop)
    O_ref.backward(dO.double())
    assert ops.relpos_attention_x3_supported(T, dp, D)
    qkv = torch.cat([_pack(t.detach().float(), dp).reshape(B * T, H * dp) for t in (q, k, v)], 1).contiguous().to(dev)
    tab = ops.relpos_attention_x3_tables(E.to(dev), dp, 1.0 / math.sqrt(dh))
    qkv_p = ops.split_planes(qkv)
    out_p = [torch.full((B * T, H * dp), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    lse = torch.zeros(B, H, T, device=dev)
    saved = torch.zeros(ops.relpos_attention_x3_saved_bytes(B, H, T, dp, D), dtype=torch.uint8, device=dev)
    scale = 1.0 / math.sqrt(dh)
    ops.relpos_attention_x3_forward(qkv_p, tab, out_p, lse, B, H, T, dp, D, scale, saved=saved, **kw)
    out = out_p[0].float() + out_p[1].float()
    O = out.view(B, T, H, dp)[..., :dh].permute(0, 2, 1, 3)
    assert_close_robust(O, O_ref.float(), 1e-4, name='O (x3 planes)', max_outlier_frac=0)
    assert_close_robust(lse, lse_ref.float(), 1e-4, name='lse (x3 planes)', max_outlier_frac=0)
    if dp > dh:
        assert float(out.view(B, T, H, dp)[..., dh:].abs().max()) == 0.0
    dOd = _pack(dO, dp).reshape(B * T, H * dp).contiguous().to(dev)
    dqkv_p = [torch.full((B * T, 3 * H * dp), 7.0, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    dsc = torch.empty(B, H, T, device=dev)
    ops.relpos_attention_x3_backward(qkv_p, tab, out_p, ops.split_planes(dOd), dsc, dqkv_p, saved, B, H, T, dp, D, scale, **kw)
    dqkv = dqkv_p[0].float() + dqkv_p[1].float()
    dq, dk, dv = [dqkv.view(B, T, 3, H, dp)[:, :, i, :, :dh].permute(0, 2, 1, 3) for i in range(3)]
    assert_close_robust(dv, v.grad.float(), 3e-4, name='dV (x3 planes)', max_outlier_frac=0)
    assert_close_robust(dk, k.grad.float(), 3e-4, name='dK (x3 planes)', max_outlier_frac=0)
    assert_close_robust(dq, q.grad.float(), 3e-4, name='dQ (x3 planes)', max_outlier_frac=0)
    if dp > dh:
        assert float(dqkv.view(B, T, 3, H, dp)[..., dh:].abs().max()) == 0.0


@pytest.mark.parametrize('case', [(37, 8, 9), (70, 32, 100), (65, 32, 20), (97, 64, 33)])
@pytest.mark.parametrize('p', [0.0, 0.25])
def test_attention_x3_planes(dev, case, p):
    T, dh, D = case
    if not is_emu(dev):
        # (d_head 96: four K / V plane tables fit the LDS up to T = 201 -- the training rows are 200; longer rows of that width keep the per-tile SS_F32X3 kernels)
        T, dh, D = {37: (200, 96, 100), 70: (224, 64, 100), 65: (209, 64, 40), 97: (131, 96, 17)}[T]
        assert not ops.relpos_attention_x3_supported(224, 96, 100)
    _run_x3_planes(dev, B=2, H=2 if is_emu(dev) else 8, T=T, dh=dh, D=D, seed=T + D, p=p)
