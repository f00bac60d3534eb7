import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
dev = torch.device('cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (M, N, K, splits) in [(3072, 768, 22016, (1, 2, 3, 4, 5, 7)), (768, 3072, 22016, (2, 3, 4, 7)), (768, 768, 22016, (4, 7, 10, 14, 20)), (2304, 768, 22016, (2, 3, 4, 5, 9)),
                          (768, 2304, 22016, (3, 4, 5, 9)), (768, 2304, 44032, (3, 4, 5, 9)), (768, 2304, 88064, (4, 5, 9, 14))]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    out = []
    for split in splits:
        t = timeit(lambda: ops.gemm(a, b, c, M, N, K, ops.rowmap(M), ops.rowmap(N), ops.rowmap(N), a_mode=1, b_mode=1, mode=2, split_k=split))
        out.append('s%d: %.0fus %.0fTF' % (split, t * 1e6, 2.0 * M * N * K / t / 1e12))
    print('M=%d N=%d K=%d  ' % (M, N, K) + ' | '.join(out), flush=True)
