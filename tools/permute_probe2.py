import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd import ops
dev = torch.device('cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        f(); e[i + 1].record()
    torch.cuda.synchronize()
    return min(e[i].elapsed_time(e[i + 1]) for i in range(n)) * 1e3
for (n, k, dt) in [(3072, 768, torch.bfloat16), (768, 3072, torch.bfloat16), (3072, 768, torch.float32)]:
    src = torch.randn(n, k, device=dev).to(dt); dst = torch.empty(k, n, device=dev, dtype=torch.bfloat16)
    b = ops.PermuteBatch(); b.add(src, dst, (k, 1, n), (1, 0, k))
    t = timeit(lambda: b.run(dev)); print('transpose', n, k, dt, 'blocks', b.jobs[0].nblocks, '%.1f us' % t, '%.0f GB/s' % (n * k * (src.element_size() + 2) / t / 1e3))
    assert torch.equal(dst, src.t().to(torch.bfloat16))
w = torch.randn(768, 768, 3, device=dev); o = torch.empty(768, 2304, device=dev, dtype=torch.bfloat16)
b = ops.PermuteBatch(); b.add(w, o, (768, 3, 768), (2304, 1, 3))
t = timeit(lambda: b.run(dev)); print('conv (O,3,I)', '%.1f us' % t)
w = torch.randn(3072, 768, device=dev); o = torch.empty(3072, 768, device=dev, dtype=torch.bfloat16)
b = ops.PermuteBatch(); b.add(w, o, (1, 3072, 768), (0, 768, 1))
t = timeit(lambda: b.run(dev)); print('cast rows', '%.1f us' % t, '%.0f GB/s' % (3072 * 768 * 6 / t / 1e3))
