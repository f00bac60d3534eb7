"""DTW HIP kernel vs golden vectors from the reference and vs the oracle (bit-exact indices)."""
import os

import numpy as np
import pytest
import torch

from oracle import dtw_ref
from silent_speech_amd import align
from tests.backend import dev, is_emu  # noqa: F401

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_dtw_golden_small(dev):
    z = np.load(os.path.join(GOLD, 'dtw_small.npz'))
    names = [k[6:] for k in z.files if k.startswith('costs/')]
    if is_emu(dev):
        names = [n for n in names if z['costs/' + n].size <= 2600]      # emulator: keep the CPU tier quick
    for name in names:
        got = align.align_from_distances(z['costs/' + name], device=dev)
        assert got == z['align/' + name].tolist(), name


def test_dtw_strided_transposed_view(dev):
    """costs.T as passed at transduction_model.py:126 (non-contiguous view)."""
    rng = np.random.default_rng(3)
    c = rng.random((30, 41), dtype=np.float32)
    want = dtw_ref.align_from_distances_c(c.T)
    assert align.align_from_distances(c.T, device=dev) == want
    assert align.align_from_distances(torch.from_numpy(c).t(), device=dev) == want
    # neither axis unit-stride (a sliced view), and views a few hundred rows / columns large (several tiles of the skew pass)
    big = rng.random((91, 130), dtype=np.float32)
    v = big[::2, ::3]
    assert align.align_from_distances(torch.from_numpy(big)[::2, ::3], device=dev) == dtw_ref.align_from_distances_c(np.ascontiguousarray(v))
    wide = rng.random((300, 70), dtype=np.float32)
    assert align.align_from_distances(torch.from_numpy(wide).t(), device=dev) == dtw_ref.align_from_distances_c(np.ascontiguousarray(wide.T))
    assert align.align_from_distances(torch.from_numpy(wide), device=dev) == dtw_ref.align_from_distances_c(wide)


def test_dtw_degenerate(dev):
    rng = np.random.default_rng(4)
    assert align.align_from_distances(rng.random((1, 7), dtype=np.float32), device=dev) == [0]
    assert align.align_from_distances(rng.random((7, 1), dtype=np.float32), device=dev) == [0] * 7
    assert align.align_from_distances(rng.random((1, 1), dtype=np.float32), device=dev) == [0]
    assert align.align_from_distances(np.ones((6, 5), dtype=np.float32), device=dev) == [0, 1, 2, 3, 4, 4]
    assert align.align_from_distances(np.ones((5, 6), dtype=np.float32), device=dev) == [0, 1, 2, 3, 4]
    with pytest.raises((IndexError, ValueError)):
        align.align_from_distances(np.zeros((0, 4), dtype=np.float32), device=dev)


def test_dtw_multiwave_rows(dev):
    """> 256 rows: crosses the wave boundary (LDS ring hand-off between waves)."""
    rng = np.random.default_rng(5)
    n, m = (300, 40) if is_emu(dev) else (700, 333)
    c = rng.integers(0, 4, (n, m)).astype(np.float32)       # many exact ties
    assert align.align_from_distances(c, device=dev) == dtw_ref.align_from_distances_c(c)


def test_dtw_batch_ragged(dev):
    rng = np.random.default_rng(6)
    shapes = [(17, 23), (64, 64), (5, 90), (1, 4)] if is_emu(dev) else [(517, 623), (1000, 1000), (64, 64), (5, 900), (1, 4), (1300, 700)]
    mats = [rng.random(s, dtype=np.float32) for s in shapes]
    flat = torch.from_numpy(np.concatenate([m.ravel() for m in mats])).to(dev)
    offs = np.cumsum([0] + [m.size for m in mats[:-1]]).tolist()
    res, roffs = align.dtw_align_batch(flat, shapes, offs, [(s[1], 1) for s in shapes])
    res = res.cpu().numpy()
    for m, s, ro in zip(mats, shapes, roffs):
        assert res[ro:ro + s[0]].tolist() == dtw_ref.align_from_distances_c(m), s


def test_dtw_sweep_paths_and_windows(dev):
    """several fast super-steps + a generic tail, two row strips (N > 1025), a backtrace longer than the 2048-step LDS window, and
    horizontal runs of hundreds / thousands of cells inside ONE lane column (register-cache refills, window exhausted mid-column)"""
    rng = np.random.default_rng(12)
    cases = [rng.random((150, 200), dtype=np.float32), rng.integers(0, 3, (1100, 70)).astype(np.float32),
             rng.random((40, 2), dtype=np.float32), rng.random((2, 40), dtype=np.float32), rng.integers(0, 2, (70, 66)).astype(np.float32),   # 64 / 128 steps exactly: only fast super-steps
             (rng.standard_normal((40, 2300)) ** 2).astype(np.float32)]
    for n, m in ((20, 300), (9, 2300), (270, 2200)):
        c = rng.random((n, m), dtype=np.float32) + 5.0
        c[n - 1, :] = 0.0                                   # the path runs along the last row, then climbs column 1
        c[:, 1] = 0.0
        cases.append(c)
    c = rng.random((300, 90), dtype=np.float32) + 5.0
    c[:, 88] = 0.0                                          # a vertical run through every lane column, then along row 1
    c[1, :] = 0.0
    cases.append(c)
    for c in cases:
        assert align.align_from_distances(c, device=dev) == dtw_ref.align_from_distances_c(c), c.shape


def _inplace_cases(rng, emu):
    """Matrices the sweep reads IN PLACE (one unit-stride axis of <= 1025 cells dealt to the lanes): lane counts that end inside a lane's
    4 cells (a straddling lane), exactly full lanes, the single-strip maximum, enough steps for clamp-free super-steps, exact ties."""
    shapes = [(5, 5), (6, 9), (9, 6), (70, 200), (200, 70), (131, 300), (258, 77), (66, 129)]
    shapes += [(1025, 90), (90, 1025)] if emu else [(1025, 700), (700, 1025), (1024, 1000), (999, 1023), (1000, 1000), (517, 1021)]
    shapes += [(2300, 9)] if emu else [(2500, 300), (4500, 37)]          # more steps than the 2048-step backtrace window (row-major: the transposed walker re-stages)
    for n, m in shapes:
        yield rng.random((n, m), dtype=np.float32)
        yield rng.integers(0, 3, (n, m)).astype(np.float32)               # ties: first-minimum order (up, left, diag)
    n, m = (300, 150) if emu else (900, 800)
    for kind in range(4):                                                 # long runs: along the last row / the last column / row 1 / column 1
        c = rng.random((n, m), dtype=np.float32) + 5.0
        if kind == 0: c[n - 1, :] = 0.0; c[:, 1] = 0.0
        if kind == 1: c[:, m - 1] = 0.0; c[1, :] = 0.0
        if kind == 2: c[:, m // 2] = 0.0; c[n // 2, :] = 0.0
        if kind == 3: c[:] = 1.0                                          # all ties: the pure "up" staircase of the reference's min()
        yield c


def test_dtw_in_place_sources(dev):
    """ss_dtw_align reads row-major matrices (lanes own columns: the transposed sweep + its own backtrace) and column-major views
    (lanes own rows) straight from the caller's memory; SS_DTW_DEBUG=8 would force the skewed-strip copy instead.  Bit-exact either way."""
    from silent_speech_amd import _lib
    rng = np.random.default_rng(21)
    src = _lib.lib().ss_dtw_source
    assert (src(1000, 1000, 1000, 1), src(1000, 1000, 1, 1000), src(300, 2000, 2000, 1), src(300, 2000, 1, 300), src(50, 60, 120, 2), src(4, 4, 4, 1)) == (2, 1, 0, 1, 0, 0)
    for c in _inplace_cases(rng, is_emu(dev)):
        assert src(c.shape[0], c.shape[1], c.shape[1], 1) == 2 and src(c.shape[0], c.shape[1], 1, c.shape[0]) == (1 if c.shape[0] <= 1025 else 0)
        want = dtw_ref.align_from_distances_c(c)
        assert align.align_from_distances(c, device=dev) == want, ('row-major', c.shape)
        ct = torch.from_numpy(np.ascontiguousarray(c.T)).to(dev).t()      # same matrix, column-major
        assert align.align_from_distances(ct, device=dev) == want, ('column-major', c.shape)
    # padded rows (stride > width) and a batch mixing both orientations with a strip-path matrix
    big = rng.random((120, 260), dtype=np.float32)
    v = big[:, 3:203]
    assert align.align_from_distances(torch.from_numpy(big)[:, 3:203], device=dev) == dtw_ref.align_from_distances_c(np.ascontiguousarray(v))
    mats = [rng.random((60, 45), dtype=np.float32), rng.random((33, 80), dtype=np.float32), rng.random((20, 70), dtype=np.float32)]
    flat = torch.from_numpy(np.concatenate([mats[0].ravel(), np.ascontiguousarray(mats[1].T).ravel(), np.repeat(mats[2].ravel(), 2)])).to(dev)
    offs = [0, mats[0].size, mats[0].size + mats[1].size]
    res, roffs = align.dtw_align_batch(flat, [m.shape for m in mats], offs, [(45, 1), (1, 33), (140, 2)])
    res = res.cpu().numpy()
    for m, ro in zip(mats, roffs):
        assert res[ro:ro + m.shape[0]].tolist() == dtw_ref.align_from_distances_c(m), m.shape


@pytest.mark.gpu
def test_dtw_big_golden_and_strips():
    """BASELINE cfg3 size (1000x1000, golden from the reference) and a >1024-row matrix (2 strips)."""
    from silent_speech_amd import _lib
    _lib.load()
    d = torch.device('cuda')
    z = np.load(os.path.join(GOLD, 'dtw_big.npz'))
    big = np.random.default_rng(int(z['seed'])).random(tuple(z['shape']), dtype=np.float32)
    assert align.align_from_distances(big, device=d) == z['align'].tolist()
    rng = np.random.default_rng(8)
    tall = rng.integers(0, 3, (2500, 300)).astype(np.float32)
    assert align.align_from_distances(tall, device=d) == dtw_ref.align_from_distances_c(tall)
    wide = (rng.standard_normal((300, 2500)) ** 2).astype(np.float32)
    assert align.align_from_distances(wide, device=d) == dtw_ref.align_from_distances_c(wide)


def test_time_warp_matches_reference_matrices(dev):
    """align.py:5-14: the cumulative matrix itself, bit for bit against the matrices the reference's time_warp produced."""
    z = np.load(os.path.join(GOLD, 'dtw_small.npz'))
    names = [k[6:] for k in z.files if k.startswith('costs/')]
    if is_emu(dev):
        names = [n for n in names if z['costs/' + n].size <= 2600]
    for name in names:
        got = align.time_warp(z['costs/' + name], device=dev)
        want = z['dtw/' + name]
        assert got.dtype == want.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want), name
    # strided view in, tensor in -> tensor out
    c = np.random.default_rng(5).random((23, 31), dtype=np.float32)
    t = align.time_warp(torch.from_numpy(c).t(), device=dev)
    assert isinstance(t, torch.Tensor) and np.array_equal(t.cpu().numpy(), dtw_ref.time_warp_numpy(np.ascontiguousarray(c.T)))
    one = align.time_warp(np.ones((1, 4), dtype=np.float32), device=dev)
    assert one[0, 0] == 0 and np.isinf(one[0, 1:]).all()


def test_float64_input_keeps_float64_arithmetic(dev):
    """align.py:6 `zeros_like(costs)`: a float64 matrix is accumulated in float64.  The matrix below ties in float32 (the small terms
    are below half an ulp of 1) but not in float64: rounding the input to float32 changes the alignment ([0,1,2,3,4] vs [0,1,3,4,5])."""
    k = np.array([[3, 2, 2, 1, 1, 0], [0, 0, 0, 3, 2, 3], [2, 2, 3, 2, 2, 2], [2, 3, 1, 3, 2, 0], [1, 3, 2, 0, 3, 2]], dtype=np.float64)
    c = 1.0 + k * 1e-9                                         # all ones in float32
    want64 = dtw_ref.backtrace_numpy(dtw_ref.time_warp_numpy(c))
    want32 = dtw_ref.backtrace_numpy(dtw_ref.time_warp_numpy(c.astype(np.float32)))
    assert want64 != want32, 'the test matrix must separate the two precisions'
    assert align.align_from_distances(c, device=dev) == want64
    assert align.align_from_distances(c.astype(np.float32), device=dev) == want32
    d64 = align.time_warp(c, device=dev)
    assert d64.dtype == np.float64 and np.array_equal(d64, dtw_ref.time_warp_numpy(c))
    rng = np.random.default_rng(11)
    r = rng.random((40, 57))
    assert align.align_from_distances(r, device=dev) == dtw_ref.backtrace_numpy(dtw_ref.time_warp_numpy(r))


@pytest.mark.gpu
def test_time_warp_big_golden():
    """The 1000 x 1000 reference matrix: last row and f64 checksum of the cumulative matrix the reference computed."""
    from silent_speech_amd import _lib
    _lib.load()
    z = np.load(os.path.join(GOLD, 'dtw_big.npz'))
    n, m = [int(v) for v in z['shape']]
    big = np.random.default_rng(int(z['seed'])).random((n, m), dtype=np.float32)
    d = align.time_warp(big, device=torch.device('cuda'))
    assert np.array_equal(d[-1], z['dtw_last'])
    assert float(d[1:, 1:].astype(np.float64).sum()) == float(z['dtw_sum64'])
