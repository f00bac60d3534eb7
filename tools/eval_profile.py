"""Developer tool: host vs device time of the evaluation legs (bench.py eval_leg): test() over a synthetic dev set, predict_utterance."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd.architecture import Model
from silent_speech_amd.synthetic import SyntheticEMGDataset
from silent_speech_amd.transduction_model import predict_utterance, test
dev = torch.device('cuda:0')
torch.manual_seed(2)
model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.bfloat16).to(dev)
ds = SyntheticEMGDataset(64, seed=5, min_frames=200, max_frames=860)
for _ in range(2):
    test(model, ds, dev)
torch.cuda.synchronize()
t0 = time.perf_counter(); test(model, ds, dev); t1 = time.perf_counter(); torch.cuda.synchronize(); print('test(): host %.2f ms total %.2f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); test(model, ds, dev); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    test(model, ds, dev); torch.cuda.synchronize()
tot = sum((e.device_time if hasattr(e, 'device_time') else e.cuda_time) for e in prof.events() if e.device_type is not None and str(e.device_type).endswith('CUDA'))
print('test(): device time %.2f ms' % (tot / 1e3))
long_ds = SyntheticEMGDataset(16, seed=6, min_frames=600, max_frames=1000, silent_fraction=0.0)
items = [long_ds[i] for i in range(16)]
for i in range(2):
    predict_utterance(model, items[i], dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in items:
    out = predict_utterance(model, it, dev)
t1 = time.perf_counter(); torch.cuda.synchronize(); print('predict_utterance x16: host %.2f ms total %.2f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for it in items:
        out = predict_utterance(model, it, dev)
    torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith('CUDA'):
        agg[e.name[:70]][0] += 1; agg[e.name[:70]][1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
print('predict_utterance x16: device time %.2f ms' % (sum(v[1] for v in agg.values()) / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print('%-72s %4d %9.1f us' % (k, n, t))
