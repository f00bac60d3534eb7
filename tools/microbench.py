#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the MI355X (timed with HIP events on torch's current stream, which
is the stream every kernel is launched on).  Usage: python tools/microbench.py [gemm] [dtw] ..."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from silent_speech_amd import ops, align, _lib  # noqa: E402
from silent_speech_amd._lib import OP_KC, OP_OC  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def bench_gemm():
    dev = torch.device('cuda')
    out = []
    for dt in (torch.bfloat16, torch.float32):
        for (M, N, K, am, bm, tag) in [(22000, 768, 768, OP_KC, OP_KC, 'linear'), (22000, 3072, 768, OP_KC, OP_KC, 'ffn1'),
                                       (22000, 768, 3072, OP_KC, OP_KC, 'ffn2'), (22000, 768, 2304, OP_KC, OP_KC, 'conv@200'),
                                       (88000, 768, 2304, OP_KC, OP_KC, 'conv@800'), (22000, 768, 3072, OP_KC, OP_OC, 'dX ffn1'),
                                       (3072, 768, 22000, OP_OC, OP_OC, 'dW ffn1')]:
            if dt == torch.float32 and M > 30000:
                continue
            a = torch.randn((M, K) if am == OP_KC else (K, M), device=dev).to(dt)
            b = torch.randn((N, K) if bm == OP_KC else (K, N), device=dev).to(dt)
            split = 1
            c = torch.zeros(M, N, device=dev, dtype=dt)
            mode = 0
            if tag.startswith('dW'):
                split, mode = 16, 2
                c = torch.zeros(M, N, device=dev, dtype=torch.float32)
            f = lambda: ops.gemm(a, b, c, M, N, K, ops.rowmap(K if am == OP_KC else M), ops.rowmap(K if bm == OP_KC else N), ops.rowmap(N),
                                 a_mode=am, b_mode=bm, mode=mode, split_k=split)
            t = timeit(f)
            out.append(dict(kernel='gemm', dtype=str(dt), tag=tag, M=M, N=N, K=K, ms=t * 1e3, tflops=2.0 * M * N * K / t / 1e12))
            print(out[-1], flush=True)
    return out


def bench_dtw():
    dev = torch.device('cuda')
    out = []
    rng = np.random.default_rng(0)
    for nb, n in [(1, 1000), (64, 1000), (256, 1000), (512, 1000)]:
        flat = torch.from_numpy(rng.random((nb, n, n), dtype=np.float32)).to(dev)
        shapes = [(n, n)] * nb
        offs = [i * n * n for i in range(nb)]
        f = lambda: align.dtw_align_batch(flat.view(-1), shapes, offs, [(n, 1)] * nb)
        t = timeit(f, iters=5, warmup=2)
        out.append(dict(kernel='dtw(skew+fwd+backtrace+host desc)', batch=nb, n=n, ms=t * 1e3, matrices_per_s=nb / t,
                        algorithmic_GBps=8.0 * n * n * nb / t / 1e9))
        print(out[-1], flush=True)
    return out


def bench_attn():
    import math
    dev = torch.device('cuda')
    B, H, T, dh, D = 110, 8, 200, 96, 100
    dp, Tp = 96, 200
    out = []
    for dt in (torch.bfloat16,):
        qkv = (torch.randn(B * T, 3 * H * dp, device=dev) * 0.5).to(dt)
        qkvT = qkv.view(B, T, 3 * H * dp).transpose(1, 2).contiguous()
        E = (torch.randn(H, 2 * D - 1, dp, device=dev) * 0.1).to(dt)
        ET = torch.zeros(H, dp, 224, device=dev, dtype=dt); ET[:, :, :199] = E.transpose(1, 2)
        o = torch.empty(B * T, H * dp, device=dev, dtype=dt); lse = torch.empty(B, H, T, device=dev)
        dO = torch.randn(B * T, H * dp, device=dev).to(dt); dOT = dO.view(B, T, H * dp).transpose(1, 2).contiguous()
        dqkv = torch.empty_like(qkv); dsc = torch.empty(B, H, T, device=dev)
        sc = 1 / math.sqrt(dh)
        for p in (0.0, 0.2):
            tf = timeit(lambda: ops.relpos_attention_forward(qkv, qkvT, E, o, lse, B, H, T, Tp, dp, D, sc, p=p, seed=1, rng_stream=0))
            tb = timeit(lambda: ops.relpos_attention_backward(qkv, qkvT, E, ET, o, lse, dO, dOT, dsc, dqkv, B, H, T, Tp, dp, D, sc, p=p, seed=1, rng_stream=0))
            useful = B * H * (2.0 * T * T * dh * 2 + 2.0 * T * 199 * dh)      # QK^T + PV + positional term, forward
            out.append(dict(kernel='attention', dtype=str(dt), p=p, fwd_us=tf * 1e6, bwd_us=tb * 1e6, fwd_useful_tflops=useful / tf / 1e12))
            print(out[-1], flush=True)
    return out


def bench_ln():
    dev = torch.device('cuda')
    rows, C = 22000, 768
    out = []
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16); a = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    for p in (0.0, 0.2):
        t = timeit(lambda: ops.add_dropout_layernorm(x, a, g, b, y, rows, C, p=p, seed=3, rng_stream=1))
        mean, rstd = ops.add_dropout_layernorm(x, a, g, b, y, rows, C, p=p, seed=3, rng_stream=1)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev); dbr = torch.empty_like(x); dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        tb = timeit(lambda: ops.layernorm_backward(dy, a, mean, rstd, g, dy, dbr, dg, db, rows, C, p=p, seed=3, rng_stream=1))
        out.append(dict(kernel='layernorm', p=p, fwd_us=t * 1e6, fwd_GBps=rows * C * 2 * 4 / t / 1e9, bwd_us=tb * 1e6, bwd_GBps=rows * C * 2 * 4 / tb / 1e9))
        print(out[-1], flush=True)
    return out


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'dtw']
    res = []
    if 'gemm' in which:
        res += bench_gemm()
    if 'dtw' in which:
        res += bench_dtw()
    if 'attn' in which:
        res += bench_attn()
    if 'ln' in which:
        res += bench_ln()
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/microbench_%d.json' % int(time.time()), 'w') as f:
        json.dump(res, f, indent=1)
