"""Host -> device staging of the small per-batch tables (pointer / offset tables of combine_fixed_length, the per-utterance
descriptors of dtw_loss).  A training step sees NEW tensors every iteration (reference transduction_model.py:196-212 iterates a
DataLoader), so these tables cannot be cached: what can be done is to make them cheap -- everything a batch needs is laid out back
to back in ONE pinned host buffer and crosses PCIe in ONE asynchronous copy, issued before the forward pass is enqueued.  A copy
from pageable memory (`torch.from_numpy(x).to(device, non_blocking=True)`) is staged by the runtime inside the call and orders
itself behind everything already queued on the stream: several of those per step in the middle of the enqueue are what this
module replaces.

Pinned buffers come from a small per-device ring; a slot is reused only after the event recorded behind its last copy has
completed (in steady state it completed many steps ago, so the wait never blocks).
"""
import numpy as np
import torch

from . import _lib

import time

_SLOTS = 8
_ALIGN = 16
_rings = {}
WAIT_SECONDS = [0.0]          # host time spent blocked on a slot's event (back-pressure of a host that runs > 8 steps ahead), for bench.py's enqueue figure


class _Ring(object):
    def __init__(self, device):
        self.device = device
        self.bufs = [None] * _SLOTS
        self.events = [None] * _SLOTS
        self.at = 0

    def slot(self, nbytes):
        i = self.at
        self.at = (i + 1) % _SLOTS
        ev = self.events[i]
        if ev is not None and not ev.query():
            t0 = time.perf_counter()
            ev.synchronize()
            WAIT_SECONDS[0] += time.perf_counter() - t0
        buf = self.bufs[i]
        if buf is None or buf.numel() < nbytes:
            cap = 1 << max(16, int(nbytes - 1).bit_length())
            buf = self.bufs[i] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        return buf, ev


def upload(arrays, device):
    """arrays: list of numpy arrays.  Returns one device tensor per array (same dtype and shape), all living in ONE device
    buffer filled by ONE host-to-device copy on the current stream.  Emulator backend (tests): plain CPU tensors.
    An entry may also be a LIST of arrays to be concatenated along axis 0 (same dtype and trailing shape): the pieces are copied
    straight into the pinned buffer, without a host-side np.concatenate first (the device loader's 10 MB of raw recordings)."""
    parts = [[np.ascontiguousarray(p) for p in a] if isinstance(a, (list, tuple)) else None for a in arrays]
    arrays = [_Joined(ps) if ps is not None else np.ascontiguousarray(a) for a, ps in zip(arrays, parts)]
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    total = max(total, _ALIGN)
    device = torch.device(device)
    if device.type != 'cuda':
        if not _lib.is_emulator():
            raise RuntimeError('silent_speech_amd: staging to %s; the HIP kernels need an AMD GPU (no CPU fallback exists)' % device)
        return [torch.from_numpy(np.concatenate(a.parts, 0) if isinstance(a, _Joined) else a.copy()) for a in arrays]
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    ring = _rings.get(device)
    if ring is None:
        ring = _rings[device] = _Ring(device)
    host, ev = ring.slot(total)
    hv = host.numpy()
    jobs = []                                                  # (offset in the pinned buffer, source bytes)
    for a, o in zip(arrays, offs):
        if isinstance(a, _Joined):
            for q in a.parts:
                if q.nbytes:
                    jobs.append((o, q.view(np.uint8).reshape(-1)))
                    o += q.nbytes
        elif a.nbytes:
            jobs.append((o, a.view(np.uint8).reshape(-1)))
    _fill(hv, jobs, total)
    dev = torch.empty(total, dtype=torch.uint8, device=device)
    dev.copy_(host[:total], non_blocking=True)
    ev.record(torch.cuda.current_stream(device))
    out = []
    for a, o in zip(arrays, offs):
        t = dev[o:o + a.nbytes].view(_TORCH_DTYPE[a.dtype.type])
        out.append(t.view(a.shape) if a.ndim != 1 else t)
    return out


_POOL = [None]
_PAR_BYTES = 4 << 20          # below this one thread is as fast as the hand-over to a pool
_PAR_THREADS = max(1, int(__import__('os').environ.get('SS_STAGING_THREADS', '4')))


def _fill(hv, jobs, total):
    """The memcpys into the pinned buffer.  A device-loader batch is ~27 MB of raw recordings and audio: one thread copies that in ~1 ms,
    a quarter of the loader's host time per batch; numpy releases the GIL inside a large copy, so four threads share the jobs by bytes."""
    if total < _PAR_BYTES or len(jobs) < 2 or _PAR_THREADS < 2:
        for o, q in jobs:
            hv[o:o + q.nbytes] = q
        return
    if _POOL[0] is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL[0] = ThreadPoolExecutor(max_workers=_PAR_THREADS, thread_name_prefix='ss-staging')
    shares, acc, cur = [], 0, []
    per = (sum(q.nbytes for _, q in jobs) + _PAR_THREADS - 1) // _PAR_THREADS
    for o, q in jobs:                                          # contiguous runs of jobs of about `per` bytes; a large job is cut at that size
        at = 0
        while at < q.nbytes:
            n = min(q.nbytes - at, per - acc)
            cur.append((o + at, q[at:at + n]))
            at += n; acc += n
            if acc >= per:
                shares.append(cur); cur, acc = [], 0
    if cur:
        shares.append(cur)

    def run(share):
        for o, q in share:
            hv[o:o + q.nbytes] = q
    futs = [_POOL[0].submit(run, sh) for sh in shares[1:]]
    run(shares[0])
    for f in futs:
        f.result()


class _Joined(object):
    """Several arrays that become one along axis 0 in the staging buffer: what upload needs to know of the result."""

    def __init__(self, parts):
        self.parts = parts
        self.dtype = parts[0].dtype
        self.shape = (sum(int(q.shape[0]) for q in parts),) + tuple(parts[0].shape[1:])
        self.ndim = len(self.shape)
        self.nbytes = sum(q.nbytes for q in parts)


_TORCH_DTYPE = {np.int64: torch.int64, np.int32: torch.int32, np.uint8: torch.uint8, np.float32: torch.float32, np.float64: torch.float64}
