"""Developer tool: where the time of the device loader leg goes (bench.py pipeline_leg workload, torch profiler kernel table)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd.pipeline import DeviceBatchBuilder
dev = torch.device('cuda:0')
rng = np.random.default_rng(4)
recs = []
for _ in range(24):
    T = int(rng.integers(200, 861)); n = int(T * 8 / 0.68906) + 40
    x = np.cumsum(rng.standard_normal((n + 400, 8)), 0) + 40.0 * np.sin(2 * np.pi * 60.0 * np.arange(n + 400) / 1000.0)[:, None] + rng.standard_normal((n + 400, 8)) * 30.0
    recs.append({'raw_emg': x[200:200 + n], 'raw_emg_before': x[:200], 'raw_emg_after': x[200 + n:], 'silent': False,
                 'audio': np.clip(0.1 * rng.standard_normal(256 * (T + 2)), -1, 1).astype(np.float32), 'text_int': np.zeros(3, dtype=np.int64)})
for _ in range(2):
    b = DeviceBatchBuilder(dev).build(recs)
torch.cuda.synchronize()
t0 = time.perf_counter(); b = DeviceBatchBuilder(dev).build(recs); torch.cuda.synchronize(); print('build: %.2f ms' % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); b = DeviceBatchBuilder(dev).build(recs); t1 = time.perf_counter(); torch.cuda.synchronize(); print('host enqueue: %.2f ms, total %.2f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    b = DeviceBatchBuilder(dev).build(recs); torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith('CUDA'):
        agg[e.name[:80]][0] += 1; agg[e.name[:80]][1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
tot = sum(v[1] for v in agg.values())
print('device time total %.2f ms' % (tot / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-82s %4d %9.1f us' % (k, n, t))
import cProfile, pstats
bb = DeviceBatchBuilder(dev)
for _ in range(4): b = bb.build(recs)
torch.cuda.synchronize()
import time as _t
for _ in range(3):
    t0 = _t.perf_counter(); b = bb.build(recs); t1 = _t.perf_counter(); torch.cuda.synchronize(); print("steady build: host %.2f ms total %.2f ms" % ((t1 - t0) * 1e3, (_t.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); b = bb.build(recs); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    b = DeviceBatchBuilder(dev).build(recs)
pr.disable(); torch.cuda.synchronize()
print('host profile of 5 builds:')
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
