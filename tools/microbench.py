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


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'dtw']
    res = []
    if 'gemm' in which:
        res += bench_gemm()
    if 'dtw' in which:
        res += bench_dtw()
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/microbench_%d.json' % int(time.time()), 'w') as f:
        json.dump(res, f, indent=1)
