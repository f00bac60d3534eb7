"""Timing probe for the DTW kernels (developer tool): the ten cost matrices of the bench batch and BASELINE configs[2] (64 x 1000^2).
SS_DTW_DEBUG=1 drops the backtrace, 2 the sweep, 4 forces the generic sweep path (results are then wrong: timing only)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from silent_speech_amd import align

def timed(job, flat, iters=10):
    job.run(flat)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        job.run(flat)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
CASES = (('bench batch', [(521, 489), (459, 415), (913, 843), (174, 213), (331, 331), (317, 386), (556, 549), (425, 415), (215, 210), (206, 256)]),
                     ('one 913x843', [(913, 843)]), ('one 1000x1000', [(1000, 1000)]), ('64 x 1000x1000', [(1000, 1000)] * 64), ('one 1100x900 (2 strips)', [(1100, 900)]))
sel = [int(a) for a in sys.argv[1:]] or range(len(CASES))
for name, shapes in [CASES[i] for i in sel]:
    offs, o = [], 0
    for n, m in shapes:
        offs.append(o); o += n * m
    flat = torch.from_numpy(rng.random(o, dtype=np.float32)).to(dev)
    job = align.DtwBatch(shapes, offs, [(m, 1) for n, m in shapes], dev)
    print('%-28s %8.1f us  (SS_DTW_DEBUG=%s)' % (name, timed(job, flat), os.environ.get('SS_DTW_DEBUG', '0')))
