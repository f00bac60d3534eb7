import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd.architecture import Model
from silent_speech_amd import engine, ops
dev = torch.device('cuda')
m = Model(112, 80, 48).to(dev)
for p in m.parameters():
    if p.grad is None: p.grad = torch.zeros_like(p)
m.flat_arenas()
gu = engine.grad_unpack(m, dev)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print('encoder batch', len(gu.encoder_batch.jobs), 'jobs', sum(j.nblocks for j in gu.encoder_batch.jobs), 'blocks', '%.1f us' % timeit(lambda: gu.encoder_batch.run(dev)))
for i, b in enumerate(gu.conv_batches):
    print('conv batch', i, '%.1f us' % timeit(lambda: b.run(dev)))
# single jobs
H, dh, d, dp = 8, 96, 768, 96
src = torch.randn(H * dp * d, device=dev); dst = torch.zeros(H, d, dh, device=dev)
for name, dims, st in [('wq (H,d,dh)', (H, d, dh), (dp * d, 1, d)), ('wo (H,dh,d)', (H, dh, d), (dp, 1, H * dp))]:
    for acc in (False, True):
        b = ops.PermuteBatch(); b.add(src, dst.view(-1)[:dims[0] * dims[1] * dims[2]].view(*dims), dims, st, accumulate=acc)
        print(name, 'acc' if acc else 'store', b.jobs[0].nblocks, 'blocks', '%.1f us' % timeit(lambda: b.run(dev)))
