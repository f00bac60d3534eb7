"""Times the per-step weight re-layout batches (Prepared.b1 / b2) and the gradient un-layout batch on the full model."""
import sys, time, torch
sys.path.insert(0, '.')
from silent_speech_amd.architecture import Model
from silent_speech_amd import engine
dev = torch.device('cuda')
m = Model(112, 80, 48).to(dev)
pr = engine.prepared(m)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print('b1 jobs', len(pr.b1.jobs), 'us', timeit(lambda: pr.b1.run(dev)))
print('b2 jobs', len(pr.b2.jobs), 'us', timeit(lambda: pr.b2.run(dev)))
from collections import Counter
def classify(j):
    fx1 = j.s1 == 1 and j.d1 > 1; fx0 = (not fx1) and j.s0 == 1 and j.d0 > 1
    tot = j.d0 * j.d1 * j.d2
    if j.s2 > 1 and (fx0 or fx1) and j.s0 >= 0 and j.s1 >= 0 and tot >= 4096: return 'transpose'
    if j.s2 == 1 and j.d2 % 4 == 0 and j.s0 % 4 == 0 and j.s1 % 4 == 0 and j.o0 % 4 == 0 and j.o1 % 4 == 0 and j.valid2 % 4 == 0: return 'rows'
    return 'generic'
for name, b in (('b1', pr.b1), ('b2', pr.b2)):
    c = Counter(); e = Counter()
    for j in b.jobs:
        k = classify(j); c[k] += 1; e[k] += j.d0 * j.d1 * j.d2
    print(name, dict(c), {k: v / 1e6 for k, v in e.items()})
    for j in b.jobs:
        if classify(j) == 'generic' and j.d0 * j.d1 * j.d2 > 100000:
            print('   generic big:', (j.d0, j.d1, j.d2), (j.s0, j.s1, j.s2), (j.o0, j.o1), j.valid1, j.valid2)
