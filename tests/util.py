"""Shared comparison helpers for parity tests."""
import numpy as np
import torch


def to_np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().float().cpu().numpy()
    return np.asarray(x)


def assert_close_robust(got, want, rtol, atol_frac=1e-5, name='', max_outlier_frac=2e-3, min_outliers=0):
    """|got-want| <= rtol*max|want| + atol_frac*max|want| elementwise, except for a tiny fraction of
    outliers (a ReLU pre-activation within rounding of 0 flips one hidden unit at one frame and
    perturbs that unit's gradient row -- observed between the reference and ANY re-implementation)."""
    got, want = to_np(got), to_np(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = float(np.abs(want).max()) + 1e-12
    err = np.abs(got - want)
    assert np.isfinite(got).all(), name + ': non-finite values'
    bad = err > (rtol + atol_frac) * scale + 1e-7
    nbad = int(bad.sum())
    allowed = max(min_outliers, int(max_outlier_frac * got.size))
    assert nbad <= allowed, '%s: %d/%d elements off (max err %.3e, scale %.3e, rtol %.1e)' % (
        name, nbad, got.size, float(err.max()), scale, rtol)
    return float(err.max()) / scale


def rel_l2_cos(got, want):
    """Per-tensor parity figures that no outlier allowance can hide behind: ||got - want||_2 / ||want||_2 and the cosine
    between the two tensors (f64 accumulation)."""
    g, w = to_np(got).astype(np.float64).ravel(), to_np(want).astype(np.float64).ravel()
    assert g.shape == w.shape and np.isfinite(g).all()
    nw, ng = np.linalg.norm(w), np.linalg.norm(g)
    if nw == 0.0:
        return (0.0, 1.0) if ng == 0.0 else (float('inf'), 0.0)
    return float(np.linalg.norm(g - w) / nw), float(np.dot(g, w) / (ng * nw + 1e-300))
