"""dW = dY^T X (both operands outer-contiguous, split-K atomics) at several reduction lengths: separates the main loop from the
per-item fixed cost (first-tile load + atomic epilogue)."""
import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
from silent_speech_amd.engine import _split_k
dev = torch.device('cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (M, N) in [(3072, 768), (768, 768), (2304, 768)]:
    for K in (5504, 11008, 22016, 44032):
        a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
        c = torch.zeros(M, N, device=dev)
        for split in (_split_k(M, N, K), max(1, _split_k(M, N, K) // 2)):
            t = timeit(lambda: ops.gemm(a, b, c, M, N, K, ops.rowmap(M), ops.rowmap(N), ops.rowmap(N), a_mode=1, b_mode=1, mode=2, split_k=split))
            print('M=%d N=%d K=%d split=%d  %.1f us  %.0f TF' % (M, N, K, split, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
