#!/usr/bin/env python3
"""Where the HOST time of a training step goes (cProfile over rotated steps of the bench loop): the two native plan calls vs the
Python around them.  usage: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from silent_speech_amd.architecture import Model              # noqa: E402
from silent_speech_amd.optim import FusedAdamW                # noqa: E402
from silent_speech_amd.synthetic import reference_size_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = Model(112, 80, 48, model_size=768, num_layers=6, dropout=0.2, compute_dtype=torch.bfloat16).to(dev)
model.train()
optim = FusedAdamW(model, weight_decay=1e-7)
batches = [reference_size_batch(seed=100 * j, device=dev) for j in range(4)]
step = bench.make_step(model, optim, batches, None, [0])
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    step()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.3f ms/step, incl. drain %.3f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
