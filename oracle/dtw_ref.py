"""Oracle: DTW cumulative cost + backtrace.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference align.py:5-14 (time_warp) and align.py:16-34 (align_from_distances):
  dtw = zeros_like(costs); dtw[0,1:] = dtw[1:,0] = +inf            (align.py:7-9)
  dtw[i,j] = costs[i,j] + min(dtw[i-1,j], dtw[i,j-1], dtw[i-1,j-1]) for i,j >= 1   (:11-13)
  backtrace from (N-1,M-1) while i>0 and j>0: results[i]=j, move to the FIRST minimum of
  [(i-1,j),(i,j-1),(i-1,j-1)] (Python min => ties prefer up, then left, then diag)  (:21-26)
Arithmetic stays in the input dtype (float32 from torch), one add per cell, no reassociation.
The compiled C twin (dtw_ref.c) is used for large matrices and as the timed CPU baseline.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_clib = None


def time_warp_numpy(costs):
    costs = np.asarray(costs)
    n, m = costs.shape
    dtw = np.zeros_like(costs)
    dtw[0, 1:] = np.inf
    dtw[1:, 0] = np.inf
    for i in range(1, n):
        row_prev = dtw[i - 1]
        row = dtw[i]
        c = costs[i]
        for j in range(1, m):
            a = row_prev[j]
            b = row[j - 1]
            d = row_prev[j - 1]
            best = a if a <= b else b          # min(a,b)   (value only; ties irrelevant for value)
            best = best if best <= d else d
            row[j] = c[j] + best
    return dtw


def backtrace_numpy(dtw):
    n, m = dtw.shape
    i, j = n - 1, m - 1
    res = [0] * n
    while i > 0 and j > 0:
        res[i] = j
        up, left, diag = dtw[i - 1, j], dtw[i, j - 1], dtw[i - 1, j - 1]
        # first-wins order: up, left, diag
        if up <= left and up <= diag:
            i -= 1
        elif left <= diag:
            j -= 1
        else:
            i -= 1
            j -= 1
    return res


def align_from_distances_numpy(costs):
    return backtrace_numpy(time_warp_numpy(np.ascontiguousarray(costs)))


def _load_c():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, '_build', 'liboracle_dtw.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle C library not built: run `make -C oracle` (or __graft_entry__.build())')
        _clib = ctypes.CDLL(path)
        _clib.oracle_dtw_align_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                               ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
        _clib.oracle_dtw_align_f32.restype = ctypes.c_int
    return _clib


def align_from_distances_c(costs, return_dtw=False):
    """costs: (N,M) float32, any strides.  Returns list[int] (and the f32 dtw matrix)."""
    costs = np.asarray(costs)
    assert costs.dtype == np.float32 and costs.ndim == 2
    n, m = costs.shape
    lib = _load_c()
    res = np.zeros(n, dtype=np.int32)
    dtw = np.empty((n, m), dtype=np.float32)
    rc = lib.oracle_dtw_align_f32(costs.ctypes.data, n, m, costs.strides[0] // 4, costs.strides[1] // 4,
                                  dtw.ctypes.data, res.ctypes.data)
    assert rc == 0
    out = [int(v) for v in res]
    return (out, dtw) if return_dtw else out
