"""Developer tool: ss_bn_backward_sums / ss_bn_backward_apply alone at the training step's shapes (two branches, ReLU gate recomputed), HIP-event times
and the fraction of the 8 TB/s HBM peak of their algorithmic bytes.  Env knobs of csrc/norm.hip: SS_BN_CHUNKS, SS_BN_RED_THREADS, SS_BN_GRID."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd import _lib, ops
dev = torch.device('cuda:0')
L = _lib.lib()
C = 768
for (B, T) in ((110, 800), (110, 400), (110, 200)):
    g = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(B, T + 2, C, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy, xa, xb, dxa, dxb = mk(), mk(), mk(), mk(), mk()
    stat = lambda: [torch.randn(C, generator=g).to(dev) * 0.1, (torch.rand(C, generator=g) + 0.5).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)]
    sa, sb = stat(), stat()
    beta_a, beta_b = torch.randn(C, generator=g).to(dev) * 0.1, torch.randn(C, generator=g).to(dev) * 0.1
    dga, dba, dgb, dbb = (torch.zeros(C, device=dev) for _ in range(4))
    scratch = torch.empty(int(L.ss_bn_scratch_floats(B, T, C)), device=dev)
    def run():
        ops.bn_backward(dy, 1, None, 1, xa, 1, sa, dxa, 1, dga, dba, scratch, B, T, C, True, xb=xb, pad_xb=1, sb=sb, dxb=dxb, pad_dxb=1, dgamma_b=dgb, dbeta_b=dbb, beta_a=beta_a, beta_b=beta_b)
    for _ in range(3): run()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): run()
        torch.cuda.synchronize()
    agg = {}
    for e in prof.events():
        if e.device_type is not None and str(e.device_type).endswith('CUDA'):
            a = agg.setdefault(e.name[:40], [0, 0.0]); a[0] += 1; a[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
    rows = B * T
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:3]:
        us = t / n
        nbytes = rows * C * 2 * (3 if 'partial' in k else 5)
        print('rows %6d  %-40s %7.1f us  %5.2f TB/s (%2.0f %%)' % (rows, k, us, nbytes / us / 1e6, nbytes / us / 1e6 / 8 * 100))
