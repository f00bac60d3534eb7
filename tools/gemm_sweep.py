#!/usr/bin/env python3
"""GEMM shape sweep on the MI355X: separates fixed per-tile cost from per-K-step cost and shows tile-count quantisation."""
import sys
import torch
sys.path.insert(0, '.')
from silent_speech_amd import ops, _lib
from silent_speech_amd._lib import OP_KC, OP_OC
import ctypes, os
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    _lib.load(sys.argv[1])

def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

dev = torch.device('cuda')
dt = torch.bfloat16
def run(M, N, K, am=OP_KC, bm=OP_KC, split=1, out_dt=None, tag=''):
    a = torch.randn((M, K) if am == OP_KC else (K, M), device=dev).to(dt)
    b = torch.randn((N, K) if bm == OP_KC else (K, N), device=dev).to(dt)
    mode = 2 if split > 1 else 0
    c = torch.zeros(M, N, device=dev, dtype=torch.float32 if split > 1 else (out_dt or dt))
    f = lambda: ops.gemm(a, b, c, M, N, K, ops.rowmap(K if am == OP_KC else M), ops.rowmap(K if bm == OP_KC else N), ops.rowmap(N), a_mode=am, b_mode=bm, mode=mode, split_k=split)
    t = timeit(f)
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * split
    print('%-10s M=%6d N=%5d K=%6d split=%2d tiles=%5d  %8.1f us  %7.1f TF' % (tag, M, N, K, split, tiles, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)

if os.environ.get('SS_GEMM_DEBUG') is not None or os.environ.get('SHORT'):
    print('debug', os.environ.get('SS_GEMM_DEBUG'))
    for K in (64, 768):
        run(22016, 768, K, tag='Ksweep')
    run(88064, 768, 24, tag='conv0')
    sys.exit(0)
print('--- K sweep (M=22016, N=768)')
for K in (64, 256, 768, 1536, 3072, 6144):
    run(22016, 768, K, tag='Ksweep')
print('--- M sweep (N=768, K=768): tile-count quantisation; 512 slots = 2 blocks x 256 CUs')
for mt in (42, 85, 86, 128, 170, 171, 172, 256, 344, 512, 688):
    run(mt * 128, 768, 768, tag='Msweep')
print('--- N sweep (M=22016,K=768)')
for N in (128, 768, 2304, 3072):
    run(22016, N, 768, tag='Nsweep')
print('--- dX form (KC,OC) and dW form (OC,OC)')
run(22016, 768, 3072, OP_KC, OP_OC, tag='dX ffn1')
run(22016, 3072, 768, OP_KC, OP_OC, tag='dX ffn2')
for s in (1, 4, 8, 16, 32):
    run(3072, 768, 22016, OP_OC, OP_OC, split=s, tag='dW ffn1')
run(768, 768, 22016, OP_OC, OP_OC, split=16, tag='dW lin')
run(768, 2304, 88064, OP_OC, OP_OC, split=16, tag='dW conv')
