"""Thin Python wrappers over the C ABI (one function per entry point; tensors in, tensors out).
No math happens here: every wrapper validates shapes/dtypes and forwards device pointers."""
import ctypes

import torch

from . import _lib
from ._lib import OP_KC, OP_OC, RowMap, GemmEpilogue


def rowmap(row_stride, rows_per_batch=0, batch_stride=0, base=0):
    return RowMap(int(base), int(batch_stride), int(row_stride), int(rows_per_batch))


def gemm(A, B, C, M, N, K, amap, bmap, cmap, a_mode=OP_KC, b_mode=OP_KC, bias=None, relu=False, gate=None,
         gate_scale=1.0, alpha=1.0, dropout_p=0.0, seed=0, rng_stream=0, mode=0, split_k=1, col_perm=None):
    """C[m][n] (+)= alpha * sum_k A(m,k)*B(n,k); see include/silent_speech_hip.h:ss_gemm."""
    assert A.dtype == B.dtype
    epi = GemmEpilogue()
    epi.bias = _lib.ptr(bias).value if bias is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    epi.gate = _lib.ptr(gate).value if gate is not None else None
    if gate is not None:
        assert gate.dtype == C.dtype
    epi.gate_scale = gate_scale
    epi.alpha = alpha
    epi.relu = int(bool(relu))
    epi.dropout_p = float(dropout_p)
    epi.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    epi.rng_stream = int(rng_stream)
    epi.mode = int(mode)
    if col_perm is not None:
        epi.col_mod, epi.col_mul, epi.col_div_mul = col_perm
    rc = _lib.lib().ss_gemm(_lib.dtype_code(A.dtype), _lib.dtype_code(C.dtype), a_mode, b_mode, _lib.ptr(A), _lib.ptr(B),
                            _lib.ptr(C), M, N, K, ctypes.byref(amap), ctypes.byref(bmap), ctypes.byref(cmap),
                            ctypes.byref(epi), split_k, _lib.stream_of(C))
    _lib.check(rc, 'ss_gemm')
    return C


def permute3d(inp, out, dims, strides, valid1=None, valid2=None, scale=1.0, accumulate=False):
    d0, d1, d2 = dims
    assert out.numel() >= d0 * d1 * d2 and out.is_contiguous()
    rc = _lib.lib().ss_permute3d(_lib.ptr(inp), _lib.dtype_code(inp.dtype), _lib.ptr(out), _lib.dtype_code(out.dtype),
                                 d0, d1, d2, strides[0], strides[1], strides[2],
                                 d1 if valid1 is None else valid1, d2 if valid2 is None else valid2,
                                 float(scale), int(accumulate), _lib.stream_of(out))
    _lib.check(rc, 'ss_permute3d')
    return out
