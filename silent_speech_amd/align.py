"""DTW alignment on the MI355X -- drop-in for the reference's align.py.

  time_warp(costs: np.ndarray) -> np.ndarray (cumulative cost matrix, dtype of the input)  (align.py:5)
  align_from_distances(distance_matrix: np.ndarray, debug=False) -> list[int]               (align.py:16)

plus the batched, device-resident form used by dtw_loss (no D2H of the cost matrix, no per-utterance
sync): `dtw_align_batch`.  The HIP kernel (csrc/dtw.hip) is bit-exact against the reference's f32
recurrence and first-minimum (up, left, diag) backtrace.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_DESC = 10


def _workspace_layout(shapes):
    """Per-matrix byte offsets (sk, dirs, bnd) into one workspace; returns (rows, total_bytes)."""
    rows, off = [], 0
    sk, dr, bd = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    for n, m in shapes:
        _lib.lib().ss_dtw_workspace_bytes(int(n), int(m), ctypes.byref(sk), ctypes.byref(dr), ctypes.byref(bd))
        rows.append((off, off + sk.value, off + sk.value + dr.value))
        off += sk.value + dr.value + bd.value
    return rows, off


class DtwBatch(object):
    """Device-side descriptors, workspace and result buffer of one batch of DTW problems (shapes / offsets / strides as for
    dtw_align_batch).  Building it is host work (descriptor table, one H2D copy); run() only enqueues the two kernels, so a
    caller that aligns batches of the same shapes repeatedly (benchmarks, evaluation loops) pays the host part once."""

    def __init__(self, shapes, offsets, strides, device):
        n = len(shapes)
        for N, M in shapes:
            if N < 1 or M < 1:
                raise ValueError('DTW cost matrix must be non-empty, got %dx%d' % (N, M))
        layout, ws_bytes = _workspace_layout(shapes)
        self.res_offs, tot = [], 0
        desc = np.zeros((n, _DESC), dtype=np.int64)
        for b, ((N, M), off, st, (sk, dr, bd)) in enumerate(zip(shapes, offsets, strides, layout)):
            desc[b] = [N, M, off, st[0], st[1], sk, dr, bd, tot, 0]
            self.res_offs.append(tot)
            tot += N
        self.n, self.max_n, self.max_m = n, max(s[0] for s in shapes), max(s[1] for s in shapes)
        self.desc = torch.from_numpy(desc).to(device, non_blocking=True)
        self.ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=device)
        self.results = torch.empty(max(tot, 1), dtype=torch.int32, device=device)
        self.cells = float(sum(N * M for N, M in shapes))

    def run(self, costs):
        assert costs.dtype == torch.float32
        rc = _lib.lib().ss_dtw_align(_lib.ptr(costs), _lib.ptr(self.desc), self.n, self.max_n, self.max_m,
                                     _lib.ptr(self.ws), _lib.ptr(self.results), _lib.stream_of(costs))
        _lib.check(rc, 'ss_dtw_align')
        return self.results, self.res_offs


def dtw_align_batch(costs, shapes, offsets, strides):
    """costs: one f32 device tensor holding every matrix; matrix b has logical shape shapes[b] = (N, M),
    starts at element offsets[b] and element (i, j) sits at offsets[b] + i*strides[b][0] + j*strides[b][1].
    Returns (results int32 device tensor, list of per-matrix offsets into it).  No host sync."""
    return DtwBatch(shapes, offsets, strides, costs.device).run(costs)


def _as_device_matrix(x, device):
    t = x.detach() if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))
    if t.dim() != 2:
        raise ValueError('cost matrix must be 2-D')
    if t.shape[0] < 1 or t.shape[1] < 1:
        raise IndexError('empty cost matrix')                         # the reference raises at shape[0]-1 indexing too
    if device is None:
        device = _lib.kernel_device_for(t)
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float32)
    return t.to(device)


def _cumulative(t, want_alignment):
    N, M = t.shape
    out = torch.empty(N, M, dtype=t.dtype, device=t.device)
    res = torch.empty(N, dtype=torch.int32, device=t.device) if want_alignment else None
    rc = _lib.lib().ss_dtw_cumulative(_lib.dtype_code(t.dtype), _lib.ptr(t), t.stride(0), t.stride(1), N, M, _lib.ptr(out),
                                      _lib.ptr(res) if res is not None else None, _lib.stream_of(t))
    _lib.check(rc, 'ss_dtw_cumulative')
    return out, res


def time_warp(costs, device=None):
    """align.py:5-14: the cumulative cost matrix itself (`dtw[-1, -1]` is the alignment cost), in the dtype of the input like the
    reference's `zeros_like(costs)` (float32 or float64; anything else is converted to float32).  numpy in -> numpy out, tensor
    in -> tensor on the compute device.  Any strided view is read in place."""
    t = _as_device_matrix(costs, device)
    out, _ = _cumulative(t, False)
    return out if isinstance(costs, torch.Tensor) else out.cpu().numpy()


def align_from_distances(distance_matrix, debug=False, device=None):
    """Reference-compatible entry point (align.py:16): numpy (N, M) matrix in, list[int] of length N out.
    The matrix may be any strided view (the reference passes costs.T).  float32 (the reference's torch->numpy path) runs the
    batched strip kernel; float64 input keeps float64 arithmetic like the reference (align.py:6 `zeros_like`): dense cumulative
    matrix + backtrace in one launch of ss_dtw_cumulative."""
    t = _as_device_matrix(distance_matrix, device)
    N, M = t.shape
    if t.dtype == torch.float64:
        _, res = _cumulative(t, True)
        return res.cpu().tolist()
    # keep the caller's strides: no transposed copy is materialised, the kernel reads the view in place (torch_ops.py: silent_speech::dtw_align)
    from . import torch_ops  # noqa: F401
    return torch.ops.silent_speech.dtw_align(t).cpu().tolist()
