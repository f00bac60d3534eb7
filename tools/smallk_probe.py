"""timing of the K <= 32 GEMM (first conv shape) with / without column statistics, against the tiled kernel"""
import sys, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
dev = torch.device('cuda')
Bn, T, Ci, Co = 110, 1600, 8, 768
To = T // 2
xpad = torch.randn(Bn, T + 2, Ci, device=dev).to(torch.bfloat16)
for ktaps in (3, 1):
    wg = (torch.randn(Co, ktaps * Ci, device=dev) * 0.3).to(torch.bfloat16)
    bias = torch.randn(Co, device=dev); shift = torch.zeros(Co, device=dev)
    amap = ops.rowmap(2 * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci, base=(0 if ktaps == 3 else Ci))
    y = torch.zeros(Bn, To, Co, dtype=torch.bfloat16, device=dev)
    cs, cq = torch.zeros(Co, device=dev), torch.zeros(Co, device=dev)
    for on, stats in ((1, False), (1, True), (0, False)):
        ops.gemm_set_option(ops.GEMM_OPT_SMALLK, on)
        kw = dict(col_stats=(cs, cq, shift)) if stats else {}
        f = lambda: ops.gemm(xpad, wg, y, Bn * To, Co, ktaps * Ci, amap, ops.rowmap(ktaps * Ci), ops.rowmap(Co), bias=bias, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print('K=%2d smallk=%d stats=%d : %.1f us  (%.0f GB/s of C)' % (ktaps * Ci, on, stats, us, Bn * To * Co * 2 / us / 1e3), flush=True)
