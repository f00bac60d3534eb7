"""silent_speech_amd/staging.py: the pinned staging buffer is filled by several threads for large batches (the device loader's raw recordings)."""
import numpy as np

from silent_speech_amd import staging


def _jobs(rng, sizes):
    parts = [rng.integers(0, 255, (n,), dtype=np.uint8) for n in sizes]
    jobs, o = [], 0
    for p in parts:
        jobs.append((o, p))
        o += p.nbytes + (-p.nbytes) % 16
    return parts, jobs, o


def test_threaded_fill_equals_the_serial_copy():
    rng = np.random.default_rng(5)
    for sizes in ([1 << 20] * 9, [7, 5 << 20, 13, 3 << 20, 1], [9 << 20], [100] * 50 + [6 << 20]):
        parts, jobs, total = _jobs(rng, sizes)
        want = np.zeros(total, dtype=np.uint8)
        for o, q in jobs:
            want[o:o + q.nbytes] = q
        for threads in (1, 2, 4, 7):
            old = staging._PAR_THREADS
            staging._PAR_THREADS, staging._POOL[0] = threads, None
            try:
                got = np.zeros(total, dtype=np.uint8)
                staging._fill(got, jobs, total)
            finally:
                staging._PAR_THREADS, staging._POOL[0] = old, None
            assert np.array_equal(got, want), (sizes[:3], threads)


def test_small_uploads_stay_on_the_calling_thread():
    rng = np.random.default_rng(6)
    parts, jobs, total = _jobs(rng, [1000, 2000, 3000])
    staging._POOL[0] = None
    got = np.zeros(total, dtype=np.uint8)
    staging._fill(got, jobs, total)
    assert staging._POOL[0] is None
    for (o, q) in jobs:
        assert np.array_equal(got[o:o + q.nbytes], q)
