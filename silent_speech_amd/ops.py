"""Thin Python wrappers over the C ABI (one function per entry point; tensors in, tensors out).
No math happens here: every wrapper validates shapes/dtypes and forwards device pointers."""
import ctypes

import torch

from . import _lib
from ._lib import OP_KC, OP_OC, RowMap, GemmEpilogue


def rowmap(row_stride, rows_per_batch=0, batch_stride=0, base=0):
    return RowMap(int(base), int(batch_stride), int(row_stride), int(rows_per_batch))


PROFILER = None     # bench.py installs a LaunchProfiler here: per-launch HIP-event timing of the kernels that are launched from Python


class LaunchProfiler(object):
    """Brackets C-ABI launches made from Python (loss / DTW / AdamW / weight re-layout / stand-alone GEMMs) with HIP events on
    the stream they are launched on (torch's current stream).  The kernels of the model's forward / backward are timed inside the
    native plan instead (ss_plan_profile)."""

    def __init__(self):
        self.records = []
        self.enabled = True

    def run(self, name, flops, nbytes, fn):
        if not self.enabled or not torch.cuda.is_available():
            return fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.records.append((name, float(flops), float(nbytes), s, e))
        return r

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, s, e in self.records:
            a = agg.setdefault(name, dict(calls=0, flops=0.0, bytes=0.0, seconds=0.0))
            a['calls'] += 1
            a['flops'] += flops
            a['bytes'] += nbytes
            a['seconds'] += s.elapsed_time(e) * 1e-3
        self.records = []
        return agg


def timed(name, flops, nbytes, fn):
    if PROFILER is not None:
        return PROFILER.run(name, flops, nbytes, fn)
    return fn()


def gemm(A, B, C, M, N, K, amap, bmap, cmap, **kw):
    """C[m][n] (+)= alpha * sum_k A(m,k)*B(n,k); see include/silent_speech_hip.h:ss_gemm."""
    return _gemm_full(A, B, C, M, N, K, amap, bmap, cmap, {}, **kw)


def permute3d(inp, out, dims, strides, valid1=None, valid2=None, scale=1.0, accumulate=False):
    d0, d1, d2 = dims
    assert out.numel() >= d0 * d1 * d2 and out.is_contiguous()
    rc = _lib.lib().ss_permute3d(_lib.ptr(inp), _lib.dtype_code(inp.dtype), _lib.ptr(out), _lib.dtype_code(out.dtype),
                                 d0, d1, d2, strides[0], strides[1], strides[2],
                                 d1 if valid1 is None else valid1, d2 if valid2 is None else valid2,
                                 float(scale), int(accumulate), _lib.stream_of(out))
    _lib.check(rc, 'ss_permute3d')
    return out


# ------------------------------------------------------------------ helpers
def _L():
    return _lib.lib()


def _p(t):
    return _lib.ptr(t)


def _s(t):
    return _lib.stream_of(t)


def _dt(t):
    return _lib.dtype_code(t.dtype)


_epi_defaults = dict(log_clamp=0.0, c2=None, cmap2=None, col_stride2=0, sign_out=None, gate_bits=None)


def gemm_ex(A, B, C, M, N, K, amap, bmap, cmap, **kw):
    """gemm() plus the rarely used epilogue extras: log_clamp, c2/cmap2/col_stride2, sign_out / gate_bits (uint8 [M][N / 8] tensors: the sign of the
    stored result as one bit per element, and a gate read from such bits instead of a tensor)."""
    extras = {k: kw.pop(k) for k in list(kw) if k in _epi_defaults}
    if not extras:
        return gemm(A, B, C, M, N, K, amap, bmap, cmap, **kw)
    return _gemm_full(A, B, C, M, N, K, amap, bmap, cmap, extras, **kw)


def _gemm_full(A, B, C, M, N, K, amap, bmap, cmap, extras, a_mode=OP_KC, b_mode=OP_KC, bias=None, relu=False, gate=None,
               gate_scale=1.0, alpha=1.0, dropout_p=0.0, seed=0, rng_stream=0, mode=0, split_k=1, col_perm=None, col_stats=None, f32_math='exact'):
    """f32_math (f32 operands only): 'exact' = f32 MFMA, 'bf16x3' = SS_F32X3 (three bf16 MFMAs per product).
    col_stats = (col_sum, col_sumsq or None, col_shift or None): f32 [N] tensors accumulated by the 8-wave kernel's epilogue
    (RuntimeError when another kernel would run: ask ss_gemm_fuses_column_stats first)."""
    epi = GemmEpilogue()
    if col_stats is not None:
        cs, cq, sh = col_stats
        epi.col_sum, epi.col_sumsq, epi.col_shift = _p(cs).value, (_p(cq).value if cq is not None else None), (_p(sh).value if sh is not None else None)
    epi.bias = _p(bias).value if bias is not None else None
    epi.gate = _p(gate).value if gate is not None else None
    epi.gate_scale, epi.alpha, epi.relu, epi.dropout_p = gate_scale, alpha, int(bool(relu)), float(dropout_p)
    epi.seed, epi.rng_stream, epi.mode = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_stream), int(mode)
    if col_perm is not None:
        epi.col_mod, epi.col_mul, epi.col_div_mul = col_perm
    epi.log_clamp = float(extras.get('log_clamp', 0.0))
    if extras.get('sign_out') is not None:
        assert extras['sign_out'].dtype == torch.uint8 and extras['sign_out'].numel() >= M * (N // 8)
        epi.sign_out, epi.sign_pitch = _p(extras['sign_out']).value, N // 8
    if extras.get('gate_bits') is not None:
        assert extras['gate_bits'].dtype == torch.uint8 and extras['gate_bits'].numel() >= M * (N // 8)
        epi.gate_bits, epi.gate_bits_pitch = _p(extras['gate_bits']).value, N // 8
    c2 = extras.get('c2')
    if c2 is not None:
        assert c2.dtype == C.dtype
        epi.c2 = _p(c2).value
        epi.cmap2 = extras['cmap2']
        epi.col_stride2 = int(extras['col_stride2'])
    assert A.dtype == B.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    if gate is not None:
        assert gate.dtype == C.dtype

    dt_in = _dt(A)
    if f32_math != 'exact':
        if f32_math != 'bf16x3' or A.dtype != torch.float32:
            raise ValueError("f32_math is 'exact' or 'bf16x3', and 'bf16x3' needs float32 operands")
        dt_in = _lib.SS_F32X3

    def launch():
        rc = _L().ss_gemm(dt_in, _dt(C), a_mode, b_mode, _p(A), _p(B), _p(C), M, N, K, ctypes.byref(amap), ctypes.byref(bmap),
                          ctypes.byref(cmap), ctypes.byref(epi), split_k, _s(C))
        _lib.check(rc, 'ss_gemm')
    if PROFILER is not None and C.is_cuda:
        PROFILER.run('ss_gemm (from Python)', 2.0 * M * N * K, (M * K + N * K) * A.element_size() + M * N * C.element_size(), launch)
    else:
        launch()
    return C


GEMM_OPT_W2, GEMM_OPT_W2_BM, GEMM_OPT_G8, GEMM_OPT_G8_NI, GEMM_OPT_G8_PIN, GEMM_OPT_DEBUG, GEMM_OPT_SMALLK = range(7)


def gemm_set_option(what, value):
    """Kernel-selection knobs of ss_gemm (see include/silent_speech_hip.h); returns the previous value."""
    return _L().ss_gemm_set_option(int(what), int(value))


def gemm_dw_grouped(jobs):
    """jobs: list of (dY, X, dW, M, N, K, amap, bmap, ldc): dW[m][n] += sum_k dY(k, m) X(k, n) for every job, ONE launch
    (include/silent_speech_hip.h: ss_gemm_dw_grouped).  Groups of more than 8 jobs are cut into several launches.
    bf16 operands, or -- the parity-grade f32 form -- (hi, lo) plane pairs from split_planes(): three jobs per gradient
    (lo.hi, hi.lo, hi.hi) that accumulate into the same dW."""
    flat = []
    for (dy, x, dw, M, N, K, amap, bmap, ldc) in jobs:
        if isinstance(dy, tuple):
            (dh, dl), (xh, xl) = dy, x
            flat += [(dl, xh, dw, M, N, K, amap, bmap, ldc, 1), (dh, xl, dw, M, N, K, amap, bmap, ldc, 1), (dh, xh, dw, M, N, K, amap, bmap, ldc, 1)]
        else:
            flat.append((dy, x, dw, M, N, K, amap, bmap, ldc, 0))
    cap = 24 if any(f[9] for f in flat) else 8
    for g0 in range(0, len(flat), cap):
        grp = flat[g0:g0 + cap]
        arr = (_lib.DwJob * len(grp))()
        for j, (dy, x, dw, M, N, K, amap, bmap, ldc, flags) in zip(arr, grp):
            assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
            j.A, j.B, j.C = _p(dy).value, _p(x).value, _p(dw).value
            j.amap, j.bmap, j.ldc, j.M, j.N, j.K, j.flags = amap, bmap, int(ldc), int(M), int(N), int(K), flags
        _lib.check(_L().ss_gemm_dw_grouped(len(grp), arr, _s(grp[0][2])), 'ss_gemm_dw_grouped')


def split_planes(x):
    """f32 tensor -> (hi, lo) bf16 tensors of the same shape: hi = bf16(x), lo = bf16(x - hi) (ss_split_planes)."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_L().ss_split_planes(_p(x), _p(hi), _p(lo), x.numel(), _s(x)), 'ss_split_planes')
    return hi, lo


def gemm_planes_supported(C, M, N, K, amap, bmap, cmap, epi=None):
    return bool(_L().ss_gemm_planes_supported(_lib.SS_F32, _p(C), M, N, K, ctypes.byref(amap), ctypes.byref(bmap), ctypes.byref(cmap), ctypes.byref(epi) if epi is not None else None))


def gemm_planes(A, B, C, M, N, K, amap, bmap, cmap, bias=None, relu=False, gate=None, gate_scale=1.0, alpha=1.0, dropout_p=0.0, seed=0, rng_stream=0,
                mode=0, col_stats=None, planes_out=None, planes_only=False):
    """C = epilogue(A . B^T) with A = (A_hi, A_lo), B = (B_hi, B_lo) plane pairs (split_planes), f32 C: the bf16 x 3 arithmetic of
    gemm(f32_math='bf16x3') as ONE bf16 contraction of length 3 K on the 8-wave kernel (ss_gemm_planes).
    planes_out = (hi, lo): bf16 tensors laid out like C that receive the stored value as planes (what split_planes(C) would give);
    planes_only: C itself is not written."""
    epi = GemmEpilogue()
    if planes_out is not None:
        assert all(t.dtype == torch.bfloat16 for t in planes_out)
        epi.planes_hi, epi.planes_lo, epi.planes_only = _p(planes_out[0]).value, _p(planes_out[1]).value, int(bool(planes_only))
    if col_stats is not None:
        cs, cq, sh = col_stats
        epi.col_sum, epi.col_sumsq, epi.col_shift = _p(cs).value, (_p(cq).value if cq is not None else None), (_p(sh).value if sh is not None else None)
    epi.bias = _p(bias).value if bias is not None else None
    epi.gate = _p(gate).value if gate is not None else None
    epi.gate_scale, epi.alpha, epi.relu, epi.dropout_p = gate_scale, alpha, int(bool(relu)), float(dropout_p)
    epi.seed, epi.rng_stream, epi.mode = int(seed) & 0xFFFFFFFFFFFFFFFF, int(rng_stream), int(mode)
    assert C.dtype == torch.float32 and all(t.dtype == torch.bfloat16 for t in (A[0], A[1], B[0], B[1]))
    rc = _L().ss_gemm_planes(_lib.SS_F32, _p(A[0]), _p(A[1]), _p(B[0]), _p(B[1]), _p(C), M, N, K, ctypes.byref(amap), ctypes.byref(bmap), ctypes.byref(cmap),
                             ctypes.byref(epi), _s(C))
    _lib.check(rc, 'ss_gemm_planes')
    return C


# ------------------------------------------------------------------ BatchNorm / LayerNorm / misc
def bn_scratch(B, T, C, device):
    return torch.empty(int(_L().ss_bn_scratch_floats(B, T, C)), dtype=torch.float32, device=device)


def bn_stats(x, B, T, C, pad, scratch, running_mean, running_var, momentum=0.1, eps=1e-5, training=True, shift=None,
             reduce_fn=None):
    """Training: batch statistics (+ running-stat update); eval: from the running statistics.
    reduce_fn(sums, n) -> n_total: optional hook that all-reduces the [3][C] sums across data-parallel ranks
    (then `shift` must be a vector shared by all ranks)."""
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    n_total, sums = float(B * T), None
    if training:
        sums = torch.empty(3 * C, dtype=torch.float32, device=x.device)
        _lib.check(_L().ss_bn_stats_sums(_dt(x), _p(x), B, T, C, pad, _p(scratch), _p(shift), _p(sums), _s(x)), 'ss_bn_stats_sums')
        if reduce_fn is not None:
            n_total = reduce_fn(sums[:2 * C], n_total)
    rc = _L().ss_bn_finalize(_p(sums), n_total, C, _p(mean), _p(invstd), _p(running_mean), _p(running_var), momentum, eps, int(training), _s(x))
    _lib.check(rc, 'ss_bn_finalize')
    return mean, invstd


def bn_apply(xa, sa, pad_xa, y, pad_y, B, T, C, relu, xb=None, sb=None, pad_xb=0):
    """sa/sb = (mean, invstd, gamma, beta)."""
    n4 = [None] * 4
    b4 = sb if sb is not None else n4
    rc = _L().ss_bn_apply(_dt(xa), _p(xa), _p(sa[0]), _p(sa[1]), _p(sa[2]), _p(sa[3]), pad_xa,
                          _p(xb), _p(b4[0]), _p(b4[1]), _p(b4[2]), _p(b4[3]), pad_xb, _p(y), pad_y, B, T, C, int(relu), _s(xa))
    _lib.check(rc, 'ss_bn_apply')
    return y


def bn_backward(dy, pad_dy, y, pad_y, xa, pad_xa, sa, dxa, pad_dxa, dgamma_a, dbeta_a, scratch, B, T, C, relu,
                xb=None, pad_xb=0, sb=None, dxb=None, pad_dxb=0, dgamma_b=None, dbeta_b=None, reduce_fn=None, beta_a=None, beta_b=None):
    """sa/sb = (mean, invstd, gamma).  reduce_fn as in bn_stats (all-reduces the [3][C] gradient sums).  beta_a [, beta_b]: recompute
    the ReLU gate from xa / xb and the affine parameters instead of reading the saved output y (which may then be None)."""
    b3 = sb if sb is not None else [None] * 3
    gate = (sa[2], beta_a, b3[2], beta_b) if beta_a is not None else (None, None, None, None)
    sums = torch.empty(3 * C, dtype=torch.float32, device=dy.device)
    rc = _L().ss_bn_backward_sums(_dt(dy), _p(dy), pad_dy, _p(y), pad_y, _p(xa), pad_xa, _p(sa[0]), _p(sa[1]),
                                  _p(xb), pad_xb, _p(b3[0]), _p(b3[1]), _p(dgamma_a), _p(dbeta_a), _p(dgamma_b), _p(dbeta_b),
                                  _p(scratch), _p(sums), B, T, C, int(relu), _p(gate[0]), _p(gate[1]), _p(gate[2]), _p(gate[3]), _s(dy))
    _lib.check(rc, 'ss_bn_backward_sums')
    n_total = float(B * T)
    if reduce_fn is not None:
        n_total = reduce_fn(sums, n_total)
    rc = _L().ss_bn_backward_apply(_dt(dy), _p(dy), pad_dy, _p(y), pad_y, _p(xa), pad_xa, _p(sa[0]), _p(sa[1]), _p(sa[2]),
                                   _p(xb), pad_xb, _p(b3[0]), _p(b3[1]), _p(b3[2]), _p(sums), n_total, _p(dxa), pad_dxa, _p(dxb), pad_dxb,
                                   B, T, C, int(relu), _p(gate[1]), _p(gate[3]), _s(dy))
    _lib.check(rc, 'ss_bn_backward_apply')


def colsum(x, rows, C, ld, out_accum):
    scratch = torch.empty(int(_L().ss_colsum_scratch_floats(rows, C)), dtype=torch.float32, device=x.device)
    _lib.check(_L().ss_colsum(_dt(x), _p(x), rows, C, ld, _p(scratch), _p(out_accum), _s(x)), 'ss_colsum')


def add_dropout_layernorm(x, branch_inout, gamma, beta, y, rows, C, eps=1e-5, p=0.0, seed=0, rng_stream=0, planes=None):
    """planes = (hi, lo): bf16 tensors shaped like y (f32 data only) that also receive y as hi / lo planes (what split_planes(y) gives)."""
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = _L().ss_add_dropout_layernorm_forward_planes(_dt(x), _p(x), _p(branch_inout), _p(gamma), _p(beta), _p(y), _p(planes[0]) if planes else None, _p(planes[1]) if planes else None,
                                                      _p(mean), _p(rstd), rows, C, eps, p, int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(x))
    _lib.check(rc, 'ss_add_dropout_layernorm_forward')
    return mean, rstd


_ln_scratch = {}


def layernorm_backward(dy, z, mean, rstd, gamma, dres, dbranch, dgamma, dbeta, rows, C, p=0.0, seed=0, rng_stream=0, dbranch_colsum=None, planes=None):
    """dres / dbranch / (+=) dgamma, dbeta [, dbranch_colsum]; the per-workgroup column sums go through a cached scratch buffer when the
    width has the 16-wave form (ss_layernorm_backward_scratch_floats > 0), through atomics otherwise."""
    n = int(_L().ss_layernorm_backward_scratch_floats(rows, C))
    scratch = None
    if n:
        key = (str(dy.device), n, int(_s(dy).value or 0))       # per stream: two streams must not share partial sums
        scratch = _ln_scratch.get(key)
        if scratch is None:
            scratch = _ln_scratch[key] = torch.empty(n, dtype=torch.float32, device=dy.device)
    rc = _L().ss_layernorm_backward_ws_planes(_dt(dy), _p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dbranch), _p(planes[0]) if planes else None,
                                              _p(planes[1]) if planes else None, _p(dgamma), _p(dbeta),
                                              _p(dbranch_colsum), _p(scratch), n, rows, C, p, int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(dy))
    _lib.check(rc, 'ss_layernorm_backward')


def emg_prepare(x_raw, out_padded, shifted_copy, B, T0, Cin, shift):
    rc = _L().ss_emg_prepare(_dt(out_padded), _p(x_raw), _p(out_padded), _p(shifted_copy), B, T0, Cin, shift, _s(x_raw))
    _lib.check(rc, 'ss_emg_prepare')


def adamw_step(p, g, m, v, n, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    timed('adamw_kernel', 0, 28.0 * n,
          lambda: _lib.check(_L().ss_adamw_step(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, _s(p)), 'ss_adamw_step'))


def cast_f32(src, dst, n):
    _lib.check(_L().ss_cast_f32(_p(src), _p(dst), _dt(dst), n, _s(src)), 'ss_cast_f32')


# ------------------------------------------------------------------ attention
def relpos_attention_saved_bytes(dtype, B, H, T, dp, D):
    """bytes of the probability image the forward can save for the backward (0: this shape recomputes)"""
    return int(_L().ss_relpos_attention_saved_bytes(dtype if isinstance(dtype, int) else _lib.dtype_code(dtype), B, H, T, dp, D))


def _attn_dt(qkv, f32_math):
    if f32_math == 'exact':
        return _dt(qkv)
    if f32_math != 'bf16x3' or qkv.dtype != torch.float32:
        raise ValueError("f32_math is 'exact' or 'bf16x3', and 'bf16x3' needs float32 tensors")
    return _lib.SS_F32X3


def relpos_attention_family(dtype, T, dp, D):
    """0 = per-tile kernels (need qkvT / dOT), 1 = LDS-resident 16 x 16 tiles, 2 = transposed 32 x 32 score tiles (need the prepared tables)"""
    return int(_L().ss_relpos_attention_family(dtype if isinstance(dtype, int) else _lib.dtype_code(dtype), T, dp, D))


def relpos_attention_tables(emb, dp, scale, out=None):
    """The embedding table of the transposed-score kernels (family 2): E / scale in MFMA-fragment order (ss_relpos_attention_prepare_tables).
    emb: [H][2D-1][dh] (any float dtype; the f32 parameter of transformer.py:172-176, or a compute-dtype copy with dh <= dp columns)."""
    emb = emb.detach().reshape(emb.shape[0], emb.shape[1], -1).float().contiguous()
    H, NE, dh = emb.shape
    D = (NE + 1) // 2
    nbytes = int(_L().ss_relpos_attention_table_bytes(H, dp, D))
    if nbytes <= 0:
        raise ValueError('no transposed-score attention tables for H=%d dp=%d' % (H, dp))
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=emb.device)
    _lib.check(_L().ss_relpos_attention_prepare_tables(_p(emb), _p(out), H, D, dh, dp, scale, _s(emb)), 'ss_relpos_attention_prepare_tables')
    return out


def _attn_tab(qkv, E, T, dp, D, scale, f32_math, tab):
    if tab is None and f32_math == 'exact' and relpos_attention_family(_dt(qkv), T, dp, D) == 2:
        tab = relpos_attention_tables(E, dp, scale)           # convenience of the per-kernel callers (tests, eager sub-modules); the plan binds a prepared table
    return tab


def relpos_attention_forward(qkv, qkvT, E, out, lse, B, H, T, Tp, dp, D, scale, p=0.0, seed=0, rng_stream=0, saved=None, f32_math='exact', tab=None):
    tab = _attn_tab(qkv, E, T, dp, D, scale, f32_math, tab)
    rc = _L().ss_relpos_attention_forward_p(_attn_dt(qkv, f32_math), _p(qkv), _p(qkvT), _p(E), _p(tab), _p(out), _p(lse), _p(saved), B, H, T, Tp, dp, D, scale, p,
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(qkv))
    _lib.check(rc, 'ss_relpos_attention_forward')


def relpos_attention_backward(qkv, qkvT, E, ET, out, lse, dO, dOT, dscratch, dqkv, B, H, T, Tp, dp, D, scale, p=0.0, seed=0, rng_stream=0, saved=None,
                              f32_math='exact', tab=None):
    tab = _attn_tab(qkv, E, T, dp, D, scale, f32_math, tab)
    rc = _L().ss_relpos_attention_backward_p(_attn_dt(qkv, f32_math), _p(qkv), _p(qkvT), _p(E), _p(ET), _p(tab), _p(out), _p(lse), _p(dO), _p(dOT), _p(dscratch),
                                             _p(dqkv), _p(saved), B, H, T, Tp, dp, D, scale, p, int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(qkv))
    _lib.check(rc, 'ss_relpos_attention_backward')


# ---- the parity-grade attention on hi / lo planes (attention_t.hip x3 kernels)
def relpos_attention_x3_supported(T, dp, D):
    return bool(_L().ss_relpos_attention_x3_supported(T, dp, D))


def relpos_attention_x3_saved_bytes(B, H, T, dp, D):
    return int(_L().ss_relpos_attention_x3_saved_bytes(B, H, T, dp, D))


def relpos_attention_x3_tables(emb, dp, scale, out=None):
    """E / scale in MFMA-fragment order as [hi | lo] bf16 planes (ss_relpos_attention_x3_prepare_tables); emb: [H][2D-1][dh] f32."""
    emb = emb.detach().reshape(emb.shape[0], emb.shape[1], -1).float().contiguous()
    H, NE, dh = emb.shape
    D = (NE + 1) // 2
    nbytes = int(_L().ss_relpos_attention_x3_table_bytes(H, dp, D))
    if nbytes <= 0:
        raise ValueError('no transposed-score attention tables for H=%d dp=%d' % (H, dp))
    if out is None:
        out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=emb.device)
    _lib.check(_L().ss_relpos_attention_x3_prepare_tables(_p(emb), _p(out), H, D, dh, dp, scale, _s(emb)), 'ss_relpos_attention_x3_prepare_tables')
    return out


def relpos_attention_x3_forward(qkv, tab, out, lse, B, H, T, dp, D, scale, p=0.0, seed=0, rng_stream=0, saved=None):
    """qkv, out: (hi, lo) bf16 plane pairs (split_planes); saved: uint8 buffer of relpos_attention_x3_saved_bytes or None."""
    rc = _L().ss_relpos_attention_x3_forward(_p(qkv[0]), _p(qkv[1]), _p(tab), _p(out[0]), _p(out[1]), _p(lse), _p(saved), B, H, T, dp, D, scale, p,
                                             int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(lse))
    _lib.check(rc, 'ss_relpos_attention_x3_forward')


def relpos_attention_x3_backward(qkv, tab, out, dO, dscratch, dqkv, saved, B, H, T, dp, D, scale, p=0.0, seed=0, rng_stream=0):
    rc = _L().ss_relpos_attention_x3_backward(_p(qkv[0]), _p(qkv[1]), _p(tab), _p(out[0]), _p(out[1]), _p(dO[0]), _p(dO[1]), _p(dscratch), _p(dqkv[0]), _p(dqkv[1]),
                                              _p(saved), B, H, T, dp, D, scale, p, int(seed) & 0xFFFFFFFFFFFFFFFF, rng_stream, _s(dscratch))
    _lib.check(rc, 'ss_relpos_attention_x3_backward')


class PermuteBatch(object):
    """A table of ss_permute3d jobs executed by ONE kernel launch.  Tensors are referenced by address: they must stay
    alive and in place while the batch is in use (parameter arenas / persistent prepared buffers do)."""

    def __init__(self):
        self.jobs, self.keep, self._dev = [], [], None

    def add(self, inp, out, dims, strides, valid1=None, valid2=None, scale=1.0, accumulate=False, out_strides=None):
        d0, d1, d2 = dims
        if out_strides is None:
            assert out.is_contiguous() and out.numel() >= d0 * d1 * d2
            out_strides = (d1 * d2, d2)
        j = _lib.PermuteJob()
        j.inp, j.out = _p(inp).value, _p(out).value
        j.s0, j.s1, j.s2 = [int(x) for x in strides]
        j.o0, j.o1 = int(out_strides[0]), int(out_strides[1])
        j.d0, j.d1, j.d2 = d0, d1, d2
        j.valid1 = d1 if valid1 is None else valid1
        j.valid2 = d2 if valid2 is None else valid2
        j.in_dtype, j.out_dtype, j.accumulate, j.scale = _dt(inp), _dt(out), int(accumulate), float(scale)
        total = d0 * d1 * d2
        j.nblocks = max(1, min(2048, (total + 4095) // 4096))       # one 4096-element tile (16 elements per thread) per block and pass
        self.jobs.append(j)
        self.keep.append((inp, out))
        self._dev = None
        return out

    def _finalize(self, device):
        import numpy as np
        first, job_of_block = 0, []
        for i, j in enumerate(self.jobs):
            j.first_block = first
            first += j.nblocks
            job_of_block += [i] * j.nblocks
        arr = (_lib.PermuteJob * len(self.jobs))(*self.jobs)
        raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8).clone()
        self._dev = (raw.to(device), torch.tensor(job_of_block, dtype=torch.int32).to(device), first)

    def device_tables(self, device):
        """(jobs, job_of_block, total_blocks) on the device: what ss_permute3d_batch / a native plan slot takes."""
        if self._dev is None or self._dev[0].device != device:
            self._finalize(device)
        return self._dev

    def run(self, device):
        if not self.jobs:
            return
        if self._dev is None or self._dev[0].device != device:
            self._finalize(device)
        jobs, jb, total = self._dev
        all_f32 = int(all(j.in_dtype == 0 and j.out_dtype == 0 for j in self.jobs))     # f32 -> f32 batches (gradient un-layout) have their own kernel
        nbytes = sum(j.d0 * j.d1 * j.d2 * ((4 if j.in_dtype == 0 else 2) + (4 if j.out_dtype == 0 else 2)) for j in self.jobs)
        timed('permute3d_batch (weight re-layout)', 0, nbytes,
              lambda: _lib.check(_L().ss_permute3d_batch(_p(jobs), _p(jb), total, all_f32, _s(jobs)), 'ss_permute3d_batch'))
