"""Where does the f32-storage / bf16 x 3 GEMM spend its time?  Times ss_gemm(f32_math='bf16x3') on the step's shapes with the ablation
mask of SS_GEMM_DEBUG (4 = no MMA at all: copies + barriers only) next to the exact-f32 and the bf16 kernels.  Tuning aid (GPU only)."""
import sys
import torch
from silent_speech_amd import ops
from silent_speech_amd._lib import OP_KC, OP_OC

dev = torch.device('cuda')


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((22000, 3072, 768), (22000, 768, 3072), (22000, 2304, 768), (22000, 768, 2304)):
    a32 = torch.randn(M, K, device=dev); b32 = torch.randn(N, K, device=dev); c32 = torch.empty(M, N, device=dev)
    a16, b16, c16 = a32.bfloat16(), b32.bfloat16(), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    row = []
    for name, fn in (('bf16', lambda: ops.gemm(a16, b16, c16, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))),
                     ('f32 exact', lambda: ops.gemm(a32, b32, c32, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N))),
                     ('x3', lambda: ops.gemm(a32, b32, c32, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), f32_math='bf16x3'))):
        us = timeit(fn)
        row.append('%s %.0f us (%.0f TF)' % (name, us, fl / us / 1e6))
    ops.gemm_set_option(5, 4)
    us = timeit(lambda: ops.gemm(a32, b32, c32, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N), f32_math='bf16x3'))
    ops.gemm_set_option(5, 0)
    row.append('x3 without MMA (copies + barriers) %.0f us' % us)
    # dW shape: reduction over the M rows
    dw = torch.zeros(N, K, device=dev)
    us = timeit(lambda: ops.gemm(c32, a32, dw, N, K, M, ops.rowmap(N), ops.rowmap(K), ops.rowmap(K), a_mode=OP_OC, b_mode=OP_OC, mode=2, split_k=8))
    row.append('dW exact %.0f us' % us)
    us = timeit(lambda: ops.gemm(c32, a32, dw, N, K, M, ops.rowmap(N), ops.rowmap(K), ops.rowmap(K), a_mode=OP_OC, b_mode=OP_OC, mode=2, split_k=8, f32_math='bf16x3'))
    row.append('dW x3 %.0f us (%.0f TF)' % (us, fl / us / 1e6))
    # round 6: the same arithmetic on hi / lo bf16 planes through the 8-wave kernels (ss_gemm_planes, three dW jobs per gradient)
    ap, bp, cp = ops.split_planes(a32), ops.split_planes(b32), ops.split_planes(c32)
    us = timeit(lambda: ops.gemm_planes(ap, bp, c32, M, N, K, ops.rowmap(K), ops.rowmap(K), ops.rowmap(N)))
    row.append('x3 planes %.0f us (%.0f TF)' % (us, fl / us / 1e6))
    us = timeit(lambda: ops.split_planes(a32))
    row.append('split A %.0f us (%.2f TB/s)' % (us, a32.numel() * 8 / us / 1e6))
    us = timeit(lambda: ops.gemm_dw_grouped([(cp, ap, dw, N, K, M, ops.rowmap(N), ops.rowmap(K), K)]))
    row.append('dW x3 planes %.0f us (%.0f TF)' % (us, fl / us / 1e6))
    print('%d x %d x %d: %s' % (M, N, K, ' | '.join(row)))
