"""ss_gemm parity: every operand-mode / dtype / epilogue combination against a torch fp32 matmul."""
import numpy as np
import pytest
import torch

from silent_speech_amd import ops
from silent_speech_amd._lib import OP_KC, OP_OC
from tests.backend import dev, is_emu as backend_is_emu, is_emu  # noqa: F401
from tests.util import assert_close_robust


def _tol(dt):
    return 2e-5 if dt == torch.float32 else 1.5e-2


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('modes', [(OP_KC, OP_KC), (OP_KC, OP_OC), (OP_OC, OP_OC), (OP_OC, OP_KC)])
def test_gemm_modes(dev, dt, modes):
    am, bm = modes
    big = not is_emu(dev)
    M, N, K = (520, 264, 328) if big else (136, 72, 104)   # ragged vs the 128x128x{32,64} tile
    g = torch.Generator().manual_seed(M + am * 2 + bm)
    a = torch.randn(M, K, generator=g).to(dt)
    b = torch.randn(N, K, generator=g).to(dt)     # asymmetric operands: a transposed C fails
    want = a.float() @ b.float().t()
    A = (a if am == OP_KC else a.t().contiguous()).to(dev)
    B = (b if bm == OP_KC else b.t().contiguous()).to(dev)
    out_dt = dt
    C = torch.full((M, N), 7.0, dtype=out_dt, device=dev)
    ops.gemm(A, B, C, M, N, K, ops.rowmap(K if am == OP_KC else M), ops.rowmap(K if bm == OP_KC else N), ops.rowmap(N),
             a_mode=am, b_mode=bm)
    assert_close_robust(C, want, rtol=_tol(dt), name='gemm', max_outlier_frac=0)


@pytest.mark.parametrize('modes', [(OP_KC, OP_KC), (OP_KC, OP_OC), (OP_OC, OP_OC), (OP_OC, OP_KC)])
def test_gemm_f32_operands_bf16x3_arithmetic(dev, modes):
    """SS_F32X3: f32 operands split hi + lo in registers, three bf16 MFMAs per product.  Measured against the f64 product: the error
    must sit between the exact-f32 kernel's and 2^-16 of sum |a||b| (an accidental plain-bf16 path would be ~100 x worse: asserted too)."""
    am, bm = modes
    big = not is_emu(dev)
    M, N, K = (520, 264, 328) if big else (136, 72, 104)
    g = torch.Generator().manual_seed(77 + am * 2 + bm)
    a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))       # row scales: the split must hold across magnitudes
    b = torch.randn(N, K, generator=g)
    want = a.double() @ b.double().t()
    scale = a.abs().double() @ b.abs().double().t()
    A = (a if am == OP_KC else a.t().contiguous()).to(dev)
    B = (b if bm == OP_KC else b.t().contiguous()).to(dev)
    errs = {}
    for math in ('exact', 'bf16x3'):
        C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
        ops.gemm(A, B, C, M, N, K, ops.rowmap(K if am == OP_KC else M), ops.rowmap(K if bm == OP_KC else N), ops.rowmap(N), a_mode=am, b_mode=bm, f32_math=math)
        errs[math] = float(((C.cpu().double() - want).abs() / scale).max())
    bf = float(((a.bfloat16().double() @ b.bfloat16().double().t() - want).abs() / scale).max())
    assert errs['exact'] < 1e-6
    assert errs['bf16x3'] < 2.0 ** -16, errs
    assert errs['bf16x3'] < bf / 30, (errs, bf)
    with pytest.raises(ValueError):
        ops.gemm(A.bfloat16(), B.bfloat16(), torch.empty(M, N, dtype=torch.bfloat16, device=dev), M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), f32_math='bf16x3')


@pytest.mark.parametrize('K', [32, 64, 160])
def test_gemm_bf16x3_copy_pipeline(dev, K):
    """The KC x KC bf16 x 3 kernel keeps TWO K tiles in flight on its two LDS stages (a stage is handed back once every wave holds its split
    fragments): 1, 2 and 5 K steps, more tiles than workgroup slots (the persistent loop's hand-over between items), ragged M / N, bias + ReLU
    through the LDS-staged epilogue.  Must run the global->LDS kernel (ss_gemm_last_kernel == 1)."""
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    M, N = (70000, 392) if big else (520, 72)
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g); b = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
    want = torch.relu(a.double() @ b.double().t() + bias.double())
    scale = a.abs().double() @ b.abs().double().t() + bias.abs().double()
    C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True, f32_math='bf16x3')
    assert _lib.lib().ss_gemm_last_kernel() == 1
    err = float(((C.cpu().double() - want).abs() / scale).max())
    assert err < 2.0 ** -16, err


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_epilogue_bias_relu_gate_accumulate(dev, dt):
    M, N, K = 70, 40, 64
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
    bias = torch.randn(N, generator=g)
    want = torch.relu(0.5 * (a.float() @ b.float().t()) + bias)
    C = torch.zeros(M, N, dtype=dt, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True, alpha=0.5)
    assert_close_robust(C, want, rtol=_tol(dt), name='bias_relu', max_outlier_frac=0)
    # gate (ReLU/dropout backward from the saved activation) + accumulate
    gate = (torch.randn(M, N, generator=g) > 0).to(dt)
    base = torch.randn(M, N, generator=g).to(dt)
    C2 = base.clone().to(dev)
    ops.gemm(a.to(dev), b.to(dev), C2, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), gate=gate.to(dev), gate_scale=1.25, mode=1)
    want2 = base.float() + (a.float() @ b.float().t()) * gate.float() * 1.25
    assert_close_robust(C2, want2, rtol=_tol(dt), name='gate_acc', max_outlier_frac=0)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_split_k_atomic(dev, dt):
    """dW = dY^T X : both operands outer-contiguous, reduction over rows, f32 atomic split-K."""
    big = not is_emu(dev)
    R, N, K = (1000, 136, 264) if big else (200, 24, 40)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(R, N, generator=g).to(dt); x = torch.randn(R, K, generator=g).to(dt)
    want = dy.float().t() @ x.float()
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ops.gemm(dy.to(dev), x.to(dev), dW, N, K, R, ops.rowmap(N), ops.rowmap(K), ops.rowmap(K), a_mode=OP_OC, b_mode=OP_OC,
             mode=2, split_k=3)
    assert_close_robust(dW, want, rtol=_tol(dt), name='splitk', max_outlier_frac=0)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('stride', [1, 2])
def test_conv_k3_as_implicit_gemm(dev, dt, stride):
    """k=3 conv over a zero-padded (B, T+2, C) buffer == GEMM with overlapping 3C-wide rows."""
    Bn, T, Ci, Co = 3, 24, 16, 40
    g = torch.Generator().manual_seed(7)
    x = torch.randn(Bn, T, Ci, generator=g).to(dt)
    w = (torch.randn(Co, Ci, 3, generator=g) * 0.3).to(dt)
    bias = torch.randn(Co, generator=g)
    want = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), bias, stride=stride, padding=1).transpose(1, 2)
    To = want.shape[1]
    xpad = torch.zeros(Bn, T + 2, Ci, dtype=dt); xpad[:, 1:-1] = x
    wg = w.permute(0, 2, 1).reshape(Co, 3 * Ci).contiguous()          # [o][kk*Ci + i]
    y = torch.zeros(Bn, To, Co, dtype=dt, device=dev)
    ops.gemm(xpad.to(dev), wg.to(dev), y, Bn * To, Co, 3 * Ci,
             ops.rowmap(stride * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci), ops.rowmap(3 * Ci), ops.rowmap(Co), bias=bias.to(dev))
    assert_close_robust(y, want, rtol=_tol(dt), name='conv', max_outlier_frac=0)


def test_gemm_dropout_statistics(dev):
    M, N, K = 64, 128, 32
    a = torch.ones(M, K); b = torch.ones(N, K) / K
    C = torch.zeros(M, N, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), dropout_p=0.2, seed=1234, rng_stream=3)
    c = C.cpu()
    kept = (c != 0)
    assert torch.allclose(c[kept], torch.full_like(c[kept], 1.25), atol=1e-5)
    frac = kept.float().mean().item()
    assert abs(frac - 0.8) < 0.02, frac
    C2 = torch.zeros(M, N, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C2, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), dropout_p=0.2, seed=1234, rng_stream=3)
    assert torch.equal(C2.cpu(), c)               # same seed/stream -> same mask


def test_gemm_rejects_bad_arguments(dev):
    a = torch.zeros(8, 6, device=dev); c = torch.zeros(8, 8, device=dev)
    with pytest.raises(RuntimeError, match='multiple of'):
        ops.gemm(a, a, c, 8, 8, 6, ops.rowmap(6), ops.rowmap(6), ops.rowmap(8))


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_direct_to_lds_path(dev, dt):
    """K a multiple of the K-tile -> global_load_lds staging kernel; several items per persistent block, ragged M/N
    (clamped rows), bias+ReLU, gate and accumulate, overlapping conv rows."""
    big = not is_emu(dev)
    M, N, K = (1000, 328, 512) if big else (300, 200, 192)
    g = torch.Generator().manual_seed(17)
    a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
    bias = torch.randn(N, generator=g)
    C = torch.zeros(M, N, dtype=dt, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True)
    assert_close_robust(C, torch.relu(a.float() @ b.float().t() + bias), _tol(dt), name='glds', max_outlier_frac=0)
    gate = (torch.randn(M, N, generator=g) > 0).to(dt); base = torch.randn(M, N, generator=g).to(dt)
    C2 = base.clone().to(dev)
    ops.gemm(a.to(dev), b.to(dev), C2, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), gate=gate.to(dev), gate_scale=1.25, mode=1)
    assert_close_robust(C2, base.float() + (a.float() @ b.float().t()) * gate.float() * 1.25, _tol(dt), name='glds gate/acc', max_outlier_frac=0)
    # k=3 conv with C_in = 64 (K = 192): overlapping rows through the direct-to-LDS path
    Bn, T, Ci, Co = 2, 40, 64, 48
    x = torch.randn(Bn, T, Ci, generator=g).to(dt); w = (torch.randn(Co, Ci, 3, generator=g) * 0.2).to(dt)
    want = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), None, stride=2, padding=1).transpose(1, 2)
    To = want.shape[1]
    xpad = torch.zeros(Bn, T + 2, Ci, dtype=dt); xpad[:, 1:-1] = x
    wg = w.permute(0, 2, 1).reshape(Co, 3 * Ci).contiguous()
    y = torch.zeros(Bn, To, Co, dtype=dt, device=dev)
    ops.gemm(xpad.to(dev), wg.to(dev), y, Bn * To, Co, 3 * Ci, ops.rowmap(2 * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci), ops.rowmap(3 * Ci), ops.rowmap(Co))
    assert_close_robust(y, want, _tol(dt), name='glds conv', max_outlier_frac=0)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_transposed_second_output(dev, dt):
    """The QKV / dO projections also emit a per-sequence transposed copy [b][col][t] (read by the attention kernels)."""
    Bn, T, N, K = (4, 40, 136, 128) if is_emu(dev) else (6, 200, 328, 256)
    M, Tp = Bn * T, T
    g = torch.Generator().manual_seed(23)
    a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
    want = a.float() @ b.float().t()
    C = torch.zeros(M, N, dtype=dt, device=dev); C2 = torch.zeros(Bn, N, Tp, dtype=dt, device=dev)
    ops.gemm_ex(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), c2=C2, cmap2=ops.rowmap(1, T, N * Tp), col_stride2=Tp)
    assert_close_robust(C, want, _tol(dt), name='C', max_outlier_frac=0)
    assert torch.equal(C2.cpu(), C.cpu().view(Bn, T, N).transpose(1, 2).contiguous())


# ------------------------------------------------------------------ 8-wave 256-column-tile kernels (csrc/gemm8.hip)
@pytest.fixture
def gemm_opts():
    """Restores the kernel-selection knobs after a test that forces a variant."""
    yield ops.gemm_set_option
    for what in range(7):
        ops.gemm_set_option(what, -1)


@pytest.mark.parametrize('ni', [8, 9])
@pytest.mark.parametrize('pin', [0, 1])
def test_gemm8_kc_forced(dev, gemm_opts, ni, pin):
    """bf16 KC x KC through gemm8_kc_kernel (tile 32*ni x 256): several items per persistent workgroup, ragged M / N
    (clamped rows), 1 / 2 / 3 K tiles, bias + ReLU, gate + accumulate, f32 output, overlapping conv rows."""
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    gemm_opts(ops.GEMM_OPT_G8, 2); gemm_opts(ops.GEMM_OPT_G8_NI, ni); gemm_opts(ops.GEMM_OPT_G8_PIN, pin)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(100 + ni)
    for (M, N, K) in ([(3000, 776, 768), (1000, 264, 64), (700, 512, 128)] if big else [(600, 264, 192), (330, 520, 64), (40, 72, 128)]):
        a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
        bias = torch.randn(N, generator=g)
        C = torch.zeros(M, N, dtype=dt, device=dev)
        ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True)
        assert _lib.lib().ss_gemm_last_kernel() == (4 if ni == 9 else 3)
        assert_close_robust(C, torch.relu(a.float() @ b.float().t() + bias), _tol(dt), name='gemm8 %s' % ((M, N, K),), max_outlier_frac=0)
    M, N, K = (1000, 328, 256) if big else (300, 136, 128)
    a = torch.randn(M, K, generator=g).to(dt); b = torch.randn(N, K, generator=g).to(dt)
    gate = (torch.randn(M, N, generator=g) > 0).to(dt); base = torch.randn(M, N, generator=g).to(dt)
    C2 = base.clone().to(dev)
    ops.gemm(a.to(dev), b.to(dev), C2, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), gate=gate.to(dev), gate_scale=1.25, mode=1)
    assert _lib.lib().ss_gemm_last_kernel() in (3, 4)
    assert_close_robust(C2, base.float() + (a.float() @ b.float().t()) * gate.float() * 1.25, _tol(dt), name='gemm8 gate/acc', max_outlier_frac=0)
    C3 = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), b.to(dev), C3, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))
    assert _lib.lib().ss_gemm_last_kernel() in (3, 4)
    assert_close_robust(C3, a.float() @ b.float().t(), 2e-3, name='gemm8 f32 out', max_outlier_frac=0)
    # k=3 conv with C_in = 64 (K = 192), stride 2, overlapping rows, output scattered into the odd rows of a (B, 2T, C) buffer
    Bn, T, Ci, Co = (3, 400, 64, 264) if big else (2, 40, 64, 48)
    x = torch.randn(Bn, T, Ci, generator=g).to(dt); w = (torch.randn(Co, Ci, 3, generator=g) * 0.2).to(dt)
    want = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), None, stride=2, padding=1).transpose(1, 2)
    To = want.shape[1]
    xpad = torch.zeros(Bn, T + 2, Ci, dtype=dt); xpad[:, 1:-1] = x
    wg = w.permute(0, 2, 1).reshape(Co, 3 * Ci).contiguous()
    y = torch.zeros(Bn, 2 * To, Co, dtype=dt, device=dev)
    ops.gemm(xpad.to(dev), wg.to(dev), y, Bn * To, Co, 3 * Ci, ops.rowmap(2 * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci), ops.rowmap(3 * Ci),
             ops.rowmap(2 * Co, To, 2 * To * Co, base=Co))
    assert _lib.lib().ss_gemm_last_kernel() in (3, 4)
    assert_close_robust(y[:, 1::2], want, _tol(dt), name='gemm8 conv', max_outlier_frac=0)
    assert float(y[:, 0::2].float().abs().max()) == 0.0


def test_gemm8_dropout_matches_other_kernels(dev, gemm_opts):
    """The dropout mask is a function of (seed, stream, row, column) only: every kernel variant draws the same one."""
    M, N, K = (600, 264, 128) if is_emu(dev) else (2000, 776, 256)
    g = torch.Generator().manual_seed(41)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev); b = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for g8 in (0, 2):
        gemm_opts(ops.GEMM_OPT_G8, g8)
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(a, b, C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), relu=True, dropout_p=0.2, seed=99, rng_stream=6)
        outs.append(C.cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('kt', [4, 2, 0])
@pytest.mark.parametrize('split', [0, 1, 3])
def test_gemm_dw_grouped(dev, split, kt):
    """Grouped dW = dY^T X: three problems in one launch (plain maps, a ragged last K tile, conv-style overlapping rows with
    batches of frames), accumulated onto a non-zero C; automatic / forced K split; kt = 4 / 2: the K-contiguous-tile kernel
    (8 x 8 register transposes on the way into LDS, 4 / 2 MFMA row tiles per phase), 0: the transposing-read kernel."""
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    jobs, wants, outs = [], [], []
    R = 1000 if big else 200
    for (N, K) in ([(520, 264), (256, 768)] if big else [(264, 40), (24, 264)]):
        dy = torch.randn(R, N, generator=g).to(dt); x = torch.randn(R, K, generator=g).to(dt)
        base = torch.randn(N, K, generator=g)
        dW = base.clone().to(dev)
        jobs.append((dy.to(dev), x.to(dev), dW, N, K, R, ops.rowmap(N), ops.rowmap(K), K))
        wants.append(base + dy.float().t() @ x.float()); outs.append(dW)
    # conv2-style: dY rows and the 3-tap windows of a zero-padded (B, T+2, C) input, batches of T frames
    Bn, T, Ci, Co = (4, 200, 64, 264) if big else (3, 72, 16, 40)            # batches of >= 64 frames, a multiple of 8: legal for every kernel
    dyc = torch.randn(Bn, T, Co, generator=g).to(dt)
    xc = torch.zeros(Bn, T + 2, Ci, dtype=dt); xc[:, 1:-1] = torch.randn(Bn, T, Ci, generator=g).to(dt)
    dWc = torch.zeros(Co, 3 * Ci, device=dev)
    jobs.append((dyc.to(dev), xc.to(dev), dWc, Co, 3 * Ci, Bn * T, ops.rowmap(Co, T, T * Co), ops.rowmap(Ci, T, (T + 2) * Ci), 3 * Ci))
    win = torch.stack([xc[:, k:k + T].float() for k in range(3)], 2).reshape(Bn * T, 3 * Ci)
    wants.append(dyc.float().reshape(Bn * T, Co).t() @ win); outs.append(dWc)
    old = _lib.lib().ss_gemm_dw_set_option(0, split)
    old_kt = _lib.lib().ss_gemm_dw_set_option(3, kt)
    try:
        ops.gemm_dw_grouped(jobs)
    finally:
        _lib.lib().ss_gemm_dw_set_option(0, old)
        _lib.lib().ss_gemm_dw_set_option(3, old_kt)
    for i, (o, w) in enumerate(zip(outs, wants)):
        assert_close_robust(o, w, 1.5e-2, name='dw job %d' % i, max_outlier_frac=0)


@pytest.mark.parametrize('ni', [8, 9])
@pytest.mark.parametrize('out_dt', [torch.bfloat16, torch.float32])
def test_gemm8_column_statistics(dev, gemm_opts, ni, out_dt):
    """Epilogue column statistics of the 8-wave kernel: col_sum += sum_m (C - shift), col_sumsq += sum_m (C - shift)^2 over the STORED
    values (BatchNorm batch statistics of a conv output / bias gradients), several tiles per column, ragged M and N, gate epilogue."""
    from silent_speech_amd import _lib
    gemm_opts(ops.GEMM_OPT_G8, 2); gemm_opts(ops.GEMM_OPT_G8_NI, ni)
    M, N, K = (700, 328, 128) if is_emu(dev) else (5000, 776, 256)
    g = torch.Generator().manual_seed(9 + ni)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16); b = (torch.randn(N, K, generator=g) * 0.2).to(torch.bfloat16)
    bias = torch.randn(N, generator=g); shift = torch.randn(N, generator=g) * 0.3
    gate = (torch.randn(M, N, generator=g) > 0).to(out_dt)
    for use_gate in (False, True):
        C = torch.zeros(M, N, dtype=out_dt, device=dev)
        cs = torch.full((N,), 2.0, device=dev); cq = torch.full((N,), 3.0, device=dev)          # accumulated into, not overwritten
        kw = dict(gate=gate.to(dev), gate_scale=1.5) if use_gate else dict(bias=bias.to(dev))
        ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), col_stats=(cs, cq, shift.to(dev)), **kw)
        assert _lib.lib().ss_gemm_last_kernel() == (4 if ni == 9 else 3)
        stored = C.float().cpu()
        want = a.float() @ b.float().t()
        want = want * gate.float() * 1.5 if use_gate else want + bias
        assert_close_robust(stored, want, 1.5e-2 if out_dt == torch.bfloat16 else 2e-3, name='C', max_outlier_frac=0)
        x = stored - shift
        assert_close_robust(cs.cpu() - 2.0, x.sum(0), 2e-5, name='col_sum', max_outlier_frac=0)
        assert_close_robust(cq.cpu() - 3.0, (x * x).sum(0), 2e-5, name='col_sumsq', max_outlier_frac=0)
    # a shape the cost model leaves to the 128-wide kernels refuses the request instead of silently skipping the statistics
    gemm_opts(ops.GEMM_OPT_G8, 0)
    assert _lib.lib().ss_gemm_last_kernel() in (3, 4)
    with pytest.raises(RuntimeError, match='8-wave kernel'):
        ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), col_stats=(cs, None, None))


@pytest.mark.parametrize('colsum', [False, True])
def test_gemm8_gate_is_a_sign_test_of_the_saved_activation(dev, gemm_opts, colsum):
    """The bf16 gate epilogue of the 8-wave kernel (out = gate > 0 ? v * gate_scale : 0, the ReLU / dropout backward of the FFN) with gates of
    every kind: positive, negative, +0, -0, tiny, huge; bias + ReLU in front and, optionally, the column sums of the stored values
    (linear1.bias.grad) -- the one gated launch of the training step."""
    from silent_speech_amd import _lib
    gemm_opts(ops.GEMM_OPT_G8, 2); gemm_opts(ops.GEMM_OPT_G8_NI, 9)
    M, N, K = (300, 264, 64) if is_emu(dev) else (5000, 776, 256)
    g = torch.Generator().manual_seed(21)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16); b = (torch.randn(N, K, generator=g) * 0.2).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    gate = torch.randn(M, N, generator=g)
    gate[::3, ::5] = 0.0; gate[1::3, ::7] = -0.0; gate[2::5, 1::4] = 1e-30; gate[::7, 2::9] = 3e38; gate[3::11] *= -1
    gate = gate.to(torch.bfloat16)
    C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
    cs = torch.full((N,), 2.0, device=dev)
    kw = dict(col_stats=(cs, None, None)) if colsum else {}
    ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True, alpha=0.5, gate=gate.to(dev), gate_scale=1.25, **kw)
    assert _lib.lib().ss_gemm_last_kernel() == 4
    want = torch.relu(0.5 * (a.float() @ b.float().t()) + bias) * 1.25 * (gate.float() > 0).float()
    stored = C.float().cpu()
    assert_close_robust(stored, want, 1.5e-2, name='gated', max_outlier_frac=0)
    assert torch.equal(stored == 0, (want == 0) | (stored == 0)) and bool(((stored != 0) <= (gate.float() > 0)).all())      # nothing leaks through a closed gate
    if colsum:
        assert_close_robust(cs.cpu() - 2.0, stored.sum(0), 2e-5, name='col_sum', max_outlier_frac=0)


@pytest.mark.parametrize('ni', [8, 9])
@pytest.mark.parametrize('side', ['accumulate', 'gate', 'gate+accumulate'])
def test_gemm8_side_inputs_on_a_halo_output_map(dev, gemm_opts, ni, side):
    """The register epilogue of the 8-wave kernel (direct_store8: results leave straight from the accumulators, the gate / the old C are read at
    the store addresses two row tiles ahead) on what the training step gives it: C += v and the ReLU gate into the (B, T + 2, C) buffer of a
    convolution -- rows mapped through a division, zero halo rows that must stay untouched -- with ragged last row / column tiles, several items
    per persistent workgroup and 1 / 3 K tiles (the next item's first K tile rides the ring only from 2 K steps on)."""
    from silent_speech_amd import _lib
    gemm_opts(ops.GEMM_OPT_G8, 2); gemm_opts(ops.GEMM_OPT_G8_NI, ni)
    big = not is_emu(dev)
    g = torch.Generator().manual_seed(300 + ni)
    for K in (64, 192):
        Bn, T, N = (5, 700, 520) if big else (3, 130, 264)
        M = Bn * T
        a = torch.randn(M, K, generator=g).to(torch.bfloat16); b = (torch.randn(N, K, generator=g) * 0.2).to(torch.bfloat16)
        base = torch.randn(Bn, T + 2, N, generator=g).to(torch.bfloat16)
        base[:, 0] = 0; base[:, -1] = 0
        gate = torch.randn(Bn, T + 2, N, generator=g).to(torch.bfloat16)
        C = base.clone().to(dev)
        cmap = ops.rowmap(N, T, (T + 2) * N, base=N)
        kw = {}
        if 'gate' in side:
            kw.update(gate=gate.to(dev), gate_scale=1.25)
        if 'accumulate' in side:
            kw.update(mode=1)
        ops.gemm(a.to(dev), b.to(dev), C, M, N, K, ops.rowmap(K), ops.rowmap(K), cmap, **kw)
        assert _lib.lib().ss_gemm_last_kernel() == (4 if ni == 9 else 3)
        v = (a.float() @ b.float().t()).view(Bn, T, N).to(torch.bfloat16).float()          # the epilogue rounds the product to bf16 first
        if 'gate' in side:
            v = v * 1.25 * (gate[:, 1:-1].float() > 0).float()
        if 'accumulate' in side:
            v = v + base[:, 1:-1].float()
        got = C.float().cpu()
        assert_close_robust(got[:, 1:-1], v, 1.2e-2, name='gemm8 %s K=%d' % (side, K), max_outlier_frac=0)
        assert float(got[:, 0].abs().max()) == 0.0 and float(got[:, -1].abs().max()) == 0.0


# ------------------------------------------------------------------ K <= 32: the LDS-free kernel of the first convolution (csrc/gemm_smallk.hip)
@pytest.mark.parametrize('ktaps', [3, 1])
@pytest.mark.parametrize('stats', [False, True])
def test_gemm_smallk_first_conv(dev, gemm_opts, ktaps, stats):
    """The model's first convolution (8 channels x 3 taps, stride 2, overlapping rows of the padded input) and its 1x1 residual projection
    (K = 8, centre tap), bias and BatchNorm column statistics in the epilogue, against conv1d -- and against the tiled kernel it replaces."""
    from silent_speech_amd import _lib
    Bn, T, Ci, Co = (2, 40, 8, 64) if backend_is_emu(dev) else (7, 800, 8, 768)
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(Bn, T, Ci, generator=g).to(dt)
    w = (torch.randn(Co, Ci, ktaps, generator=g) * 0.3).to(dt)
    bias = torch.randn(Co, generator=g); shift = torch.randn(Co, generator=g) * 0.1
    want = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.float(), bias, stride=2, padding=ktaps // 2).transpose(1, 2)
    To = want.shape[1]
    xpad = torch.zeros(Bn, T + 2, Ci, dtype=dt); xpad[:, 1:-1] = x
    wg = w.permute(0, 2, 1).reshape(Co, ktaps * Ci).contiguous()
    amap = ops.rowmap(2 * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci, base=(0 if ktaps == 3 else Ci))
    res = {}
    for on in (1, 0):
        gemm_opts(ops.GEMM_OPT_SMALLK, on)
        y = torch.zeros(Bn, To, Co, dtype=dt, device=dev)
        cs, cq = torch.full((Co,), 2.0, device=dev), torch.full((Co,), 3.0, device=dev)
        if stats and not on:
            continue                                                   # the tiled kernels have no statistics for this K
        kw = dict(col_stats=(cs, cq, shift.to(dev))) if stats else {}
        ops.gemm(xpad.to(dev), wg.to(dev), y, Bn * To, Co, ktaps * Ci, amap, ops.rowmap(ktaps * Ci), ops.rowmap(Co), bias=bias.to(dev), **kw)
        assert int(_lib.lib().ss_gemm_last_kernel()) == (5 if on else 0)
        res[on] = y.float().cpu()
        assert_close_robust(y, want, rtol=_tol(dt), name='conv K=%d' % (ktaps * Ci), max_outlier_frac=0)
        if stats:
            v = y.float().cpu().reshape(-1, Co) - shift
            assert_close_robust(cs, 2.0 + v.sum(0), 1e-4, name='col_sum', max_outlier_frac=0)
            assert_close_robust(cq, 3.0 + (v * v).sum(0), 1e-4, name='col_sumsq', max_outlier_frac=0)
    if 0 in res:
        assert torch.equal(res[1], res[0])                              # same products, same f32 accumulation order inside one MFMA


# ------------------------------------------------------------------ hi / lo bf16 planes (round 6: the parity-grade arithmetic on the 8-wave kernels)
def test_split_planes(dev):
    """hi = bf16(x), lo = bf16(x - hi), bit for bit against torch's round-to-nearest-even; x = hi + lo to 2^-16 relative; ragged length."""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3001, generator=g) * torch.exp(3 * torch.randn(3001, generator=g)))
    x[:4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-30])
    hi, lo = ops.split_planes(x.to(dev))
    want_hi = x.bfloat16()
    want_lo = (x - want_hi.float()).bfloat16()
    assert torch.equal(hi.cpu().view(torch.int16), want_hi.view(torch.int16))
    assert torch.equal(lo.cpu().view(torch.int16), want_lo.view(torch.int16))
    rel = ((hi.cpu().float() + lo.cpu().float() - x).abs() / x.abs().clamp_min(1e-30)).max()
    assert float(rel) < 2.0 ** -16


def _x3_err(C, want, scale):
    return float(((C.cpu().double() - want).abs() / scale).max())


@pytest.mark.parametrize('ni', [8, 9])
def test_gemm_planes_kc(dev, gemm_opts, ni):
    """ss_gemm_planes on gemm8_kc_kernel<float, ..., PL>: the three-segment K ring (lo.hi | hi.lo | hi.hi), several items per persistent
    workgroup (the next item's first tile is a LO-plane tile), 1 / 2 / 5 K tiles per segment, ragged M / N, bias + ReLU; the error against
    the f64 product must be that of the bf16 x 3 arithmetic (< 2^-16 of sum |a||b|, far below plain bf16)."""
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    gemm_opts(ops.GEMM_OPT_G8_NI, ni)
    g = torch.Generator().manual_seed(300 + ni)
    for (M, N, K) in ([(3000, 776, 768), (1000, 264, 64), (70000, 264, 128)] if big else [(600, 264, 128), (330, 520, 64), (40, 72, 320)]):
        a = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g)); b = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
        want = torch.relu(a.double() @ b.double().t() + bias.double())
        scale = a.abs().double() @ b.abs().double().t() + bias.abs().double()
        C = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
        assert ops.gemm_planes_supported(C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))
        ops.gemm_planes(ops.split_planes(a.to(dev)), ops.split_planes(b.to(dev)), C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), bias=bias.to(dev), relu=True)
        assert _lib.lib().ss_gemm_last_kernel() == (4 if ni == 9 else 3)
        err = _x3_err(C, want, scale)
        bf = _x3_err(torch.relu(a.bfloat16().double() @ b.bfloat16().double().t() + bias.double()), want, scale)
        assert err < 2.0 ** -16 and err < bf / 30, (M, N, K, err, bf)


def test_gemm_planes_epilogues_and_conv(dev, gemm_opts):
    """gate + C += v, dropout (the same mask as the bf16 kernels draw), column statistics, and a k = 3 stride-2 convolution over a zero-padded
    (B, T+2, C) buffer whose planes keep the buffer's geometry (overlapping rows, row-mapped output)."""
    big = not is_emu(dev)
    g = torch.Generator().manual_seed(17)
    M, N, K = (1000, 328, 256) if big else (300, 136, 128)
    a = torch.randn(M, K, generator=g); b = torch.randn(N, K, generator=g)
    A, B = ops.split_planes(a.to(dev)), ops.split_planes(b.to(dev))
    prod = a.double() @ b.double().t()
    scale = a.abs().double() @ b.abs().double().t() + 1.0
    gate = (torch.randn(M, N, generator=g) > 0).float(); base = torch.randn(M, N, generator=g)
    C = base.clone().to(dev)
    ops.gemm_planes(A, B, C, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), gate=gate.to(dev), gate_scale=1.25, mode=1)
    assert _x3_err(C, base.double() + prod * gate.double() * 1.25, scale) < 2.0 ** -16
    # dropout: identical keep decisions to the bf16 kernel with the same (seed, stream)
    Cd = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ops.gemm_planes(A, B, Cd, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), relu=True, dropout_p=0.2, seed=99, rng_stream=6)
    Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    gemm_opts(ops.GEMM_OPT_G8, 2)
    ops.gemm(a.bfloat16().to(dev), b.bfloat16().to(dev), Cb, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), relu=True, dropout_p=0.2, seed=99, rng_stream=6)
    pos = torch.relu(prod) > 1e-2 * scale
    assert torch.equal((Cd.cpu() != 0) & pos, (Cb.cpu().float() != 0) & pos)
    assert _x3_err(torch.where(Cd.cpu() != 0, Cd.cpu().double(), torch.relu(prod) / 0.8), torch.relu(prod) / 0.8, scale) < 2.0 ** -15
    # column statistics of the stored result
    cs = torch.zeros(N, device=dev); cq = torch.zeros(N, device=dev); sh = torch.randn(N, generator=g).to(dev)
    Cs = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ops.gemm_planes(A, B, Cs, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), col_stats=(cs, cq, sh))
    d = Cs.cpu().double() - sh.cpu().double()
    assert_close_robust(cs, d.sum(0).float(), 1e-4, name='planes col_sum', max_outlier_frac=0)
    assert_close_robust(cq, (d * d).sum(0).float(), 1e-4, name='planes col_sumsq', max_outlier_frac=0)
    # convolution
    Bn, T, Ci, Co = (3, 400, 64, 264) if big else (2, 40, 64, 48)
    x = torch.randn(Bn, T, Ci, generator=g); w = torch.randn(Co, Ci, 3, generator=g) * 0.2
    want = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), None, stride=2, padding=1).transpose(1, 2)
    To = want.shape[1]
    xpad = torch.zeros(Bn, T + 2, Ci); xpad[:, 1:-1] = x
    wg = w.permute(0, 2, 1).reshape(Co, 3 * Ci).contiguous()
    y = torch.zeros(Bn, 2 * To, Co, dtype=torch.float32, device=dev)
    ops.gemm_planes(ops.split_planes(xpad.to(dev)), ops.split_planes(wg.to(dev)), y, Bn * To, Co, 3 * Ci,
                    ops.rowmap(2 * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci), ops.rowmap(3 * Ci), ops.rowmap(2 * Co, To, 2 * To * Co, base=Co))
    sc = torch.nn.functional.conv1d(x.abs().double().transpose(1, 2), w.abs().double(), None, stride=2, padding=1).transpose(1, 2)
    assert _x3_err(y[:, 1::2], want, sc) < 2.0 ** -16
    assert float(y[:, 0::2].abs().max()) == 0.0


def test_gemm_sign_bits_producer_and_gate_consumer(dev, gemm_opts):
    """ss_gemm_epilogue.sign_out / gate_bits (ABI 9; transformer.py:57 relu + dropout and its backward).  Producer: the register epilogue of the 8-wave kernel writes
    [stored value > 0] as one bit per element beside the bf16 result -- equal to packbits of the result's sign, with bias + ReLU + dropout, ragged M, both tile heights.
    Consumer: the column-sum epilogue gates from those bits -- bit-identical to gating from the tensor itself.  Kernels that do not carry the paths refuse."""
    import ctypes as ct
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    g = torch.Generator().manual_seed(31)
    for ni, (M, N, K) in ((9, (3000, 776, 128) if big else (600, 264, 128)), (8, (1000, 264, 64) if big else (330, 520, 64))):
        gemm_opts(ops.GEMM_OPT_G8_NI, ni); gemm_opts(ops.GEMM_OPT_G8, 2)
        a = torch.randn(M, K, generator=g).bfloat16().to(dev); b = torch.randn(N, K, generator=g).bfloat16().to(dev); bias = torch.randn(N, generator=g).to(dev)
        rm = (ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))
        e = ops.GemmEpilogue(); e.alpha = 1.0; e.gate_scale = 1.0; e.relu = 1; e.dropout_p = 0.2
        Cq = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        assert _lib.lib().ss_gemm_sign_bits_supported(_lib.SS_BF16, _lib.SS_BF16, 0, 0, ops._p(Cq), M, N, K, ct.byref(rm[0]), ct.byref(rm[1]), ct.byref(rm[2]), ct.byref(e), 1) == 1
        H = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        sign = torch.full((M, N // 8), 0xAA, dtype=torch.uint8, device=dev)
        ops.gemm_ex(a, b, H, M, N, K, *rm, bias=bias, relu=True, dropout_p=0.2, seed=7, rng_stream=2, sign_out=sign)
        want = np.packbits((H.cpu().float().numpy() > 0).reshape(M, N // 8, 8), axis=-1, bitorder='little').reshape(M, N // 8)
        assert np.array_equal(sign.cpu().numpy(), want)
        H2 = torch.zeros_like(H)
        ops.gemm(a, b, H2, M, N, K, *rm, bias=bias, relu=True, dropout_p=0.2, seed=7, rng_stream=2)
        assert torch.equal(H.view(torch.int16), H2.view(torch.int16))          # the result does not notice
        # consumer: dX-like GEMM of the same shape with gate + column sums
        dy = torch.randn(M, K, generator=g).bfloat16().to(dev); w = torch.randn(N, K, generator=g).bfloat16().to(dev)
        D1 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); D2 = torch.zeros_like(D1)
        s1 = torch.zeros(N, device=dev); s2 = torch.zeros(N, device=dev)
        ops.gemm(dy, w, D1, M, N, K, *rm, gate=H, gate_scale=1.25, col_stats=(s1, None, None))
        ops.gemm_ex(dy, w, D2, M, N, K, *rm, gate_scale=1.25, col_stats=(s2, None, None), gate_bits=sign)
        assert torch.equal(D1.view(torch.int16), D2.view(torch.int16))
        assert_close_robust(s2, s1, 1e-6, name='col_sum with bit gate', max_outlier_frac=0)
    # a kernel without the paths must refuse instead of ignoring the request
    x = torch.randn(64, 96, generator=g).bfloat16().to(dev); C = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev)
    sg = torch.zeros(64, 8, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match='sign_out / gate_bits need the 8-wave kernel'):
        ops.gemm_ex(x, x[:64], C, 64, 64, 96, ops.rowmap(96), ops.rowmap(96), ops.rowmap(64), sign_out=sg)      # K % 64 != 0: the 128 x 128 kernels would run


def test_gemm_planes_emits_its_result_as_planes(dev, gemm_opts):
    """ss_gemm_epilogue.planes_hi / planes_lo / planes_only (ABI 9): the value a plane GEMM stores also leaves as hi / lo bf16 planes addressed like C --
    bit for bit what ss_split_planes gives on the stored C --, with bias + ReLU, with gate + C += v + column sums, on both tile heights and ragged M / N;
    planes_only leaves C untouched.  (The training plan lets the qkv / FFN-hidden / dO GEMMs of the parity-grade mode feed their consumers this way.)"""
    big = not is_emu(dev)
    g = torch.Generator().manual_seed(23)
    for ni, (M, N, K) in ((9, (3000, 776, 128) if big else (600, 264, 128)), (8, (1000, 264, 64) if big else (330, 520, 64))):
        gemm_opts(ops.GEMM_OPT_G8_NI, ni)
        a = torch.randn(M, K, generator=g); b = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g)
        A, B = ops.split_planes(a.to(dev)), ops.split_planes(b.to(dev))
        rm = (ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))
        C = torch.zeros(M, N, dtype=torch.float32, device=dev)
        hi = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=dev); lo = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=dev)
        ops.gemm_planes(A, B, C, M, N, K, *rm, bias=bias.to(dev), relu=True, planes_out=(hi, lo))
        want_hi, want_lo = ops.split_planes(C)
        assert torch.equal(hi.view(torch.int16), want_hi.view(torch.int16).view(M, N)) and torch.equal(lo.view(torch.int16), want_lo.view(torch.int16).view(M, N))
        ref = torch.zeros(M, N, dtype=torch.float32, device=dev)
        ops.gemm_planes(A, B, ref, M, N, K, *rm, bias=bias.to(dev), relu=True)
        assert torch.equal(C, ref)                                            # the f32 result does not notice
        # planes only: C keeps what it held
        C2 = torch.full((M, N), 7.0, dtype=torch.float32, device=dev)
        hi2 = torch.zeros_like(hi); lo2 = torch.zeros_like(lo)
        ops.gemm_planes(A, B, C2, M, N, K, *rm, bias=bias.to(dev), relu=True, planes_out=(hi2, lo2), planes_only=True)
        assert float((C2 - 7.0).abs().max()) == 0.0
        assert torch.equal(hi2.view(torch.int16), hi.view(torch.int16)) and torch.equal(lo2.view(torch.int16), lo.view(torch.int16))
        # gate + accumulate + column sums: the planes follow the value that is stored
        gate = (torch.randn(M, N, generator=g) > 0).float().to(dev); base = torch.randn(M, N, generator=g).to(dev)
        cs = torch.zeros(N, device=dev)
        C3 = base.clone()
        ops.gemm_planes(A, B, C3, M, N, K, *rm, gate=gate, gate_scale=1.25, mode=1, col_stats=(cs, None, None), planes_out=(hi, lo))
        w_hi, w_lo = ops.split_planes(C3)
        assert torch.equal(hi.view(torch.int16), w_hi.view(torch.int16).view(M, N)) and torch.equal(lo.view(torch.int16), w_lo.view(torch.int16).view(M, N))
        assert_close_robust(cs, C3.double().sum(0).float(), 1e-4, name='col_sum beside planes', max_outlier_frac=0)
    C = torch.zeros(64, 64, dtype=torch.float32, device=dev)
    with pytest.raises(RuntimeError, match='ss_gemm_planes only'):           # the epilogue of ss_gemm has no plane output
        e = ops.GemmEpilogue(); e.alpha = 1.0; e.gate_scale = 1.0
        h = torch.zeros(64, 64, dtype=torch.bfloat16, device=dev)
        e.planes_hi = e.planes_lo = ops._p(h).value
        import ctypes as _ct
        from silent_speech_amd import _lib as _l
        x = torch.zeros(64, 64, dtype=torch.float32, device=dev)
        m = ops.rowmap(64)
        _l.check(_l.lib().ss_gemm(_l.SS_F32, _l.SS_F32, 0, 0, ops._p(x), ops._p(x), ops._p(C), 64, 64, 64, _ct.byref(m), _ct.byref(m), _ct.byref(m), _ct.byref(e), 1, ops._s(C)), 'ss_gemm')


def test_gemm_planes_refuses_what_the_8_wave_kernel_cannot_run(dev):
    C = torch.zeros(64, 64, dtype=torch.float32, device=dev)
    assert not ops.gemm_planes_supported(C, 64, 64, 96, ops.rowmap(96), ops.rowmap(96), ops.rowmap(64))      # K % 64
    a = ops.split_planes(torch.zeros(64, 96).to(dev))
    with pytest.raises(RuntimeError):
        ops.gemm_planes(a, a, C, 64, 64, 96, ops.rowmap(96), ops.rowmap(96), ops.rowmap(64))


@pytest.mark.parametrize('split', [0, 2])
def test_gemm_dw_grouped_planes(dev, split):
    """Weight gradients of the parity-grade mode: (hi, lo) plane pairs -> three jobs per gradient accumulating into the same dW (atomics even
    at split 1); more tiles than the (emulator's / chip's) workgroup slots exercises the rounds-based K split."""
    from silent_speech_amd import _lib
    big = not is_emu(dev)
    g = torch.Generator().manual_seed(78)
    jobs, wants, outs, scales = [], [], [], []
    R = 4000 if big else 200
    for (N, K) in ([(520, 264), (256, 768), (768, 3072)] if big else [(264, 40), (24, 264)]):
        dy = torch.randn(R, N, generator=g) * torch.exp(torch.randn(1, N, generator=g)); x = torch.randn(R, K, generator=g)
        base = torch.randn(N, K, generator=g)
        dW = base.clone().to(dev)
        jobs.append((ops.split_planes(dy.to(dev)), ops.split_planes(x.to(dev)), dW, N, K, R, ops.rowmap(N), ops.rowmap(K), K))
        wants.append(base.double() + dy.double().t() @ x.double()); outs.append(dW); scales.append(dy.abs().double().t() @ x.abs().double() + base.abs().double())
    old = _lib.lib().ss_gemm_dw_set_option(0, split)
    try:
        ops.gemm_dw_grouped(jobs)
    finally:
        _lib.lib().ss_gemm_dw_set_option(0, old)
    for i, (o, w, sc) in enumerate(zip(outs, wants, scales)):
        assert _x3_err(o, w, sc) < 2.0 ** -16, i
