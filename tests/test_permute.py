"""ss_permute3d_batch (weight re-layout / gradient un-layout) against an index gather on every code path of the kernel:
linear casts, vector rows, vector and scalar LDS-tiled transposes (short X, short c, negative / near-contiguous strides), generic gather."""
import numpy as np
import pytest
import torch

from silent_speech_amd import ops
from tests.backend import dev  # noqa: F401

CASES = [
    # dims, input strides, input offset, valid1, valid2, accumulate, in dtype, out dtype
    ((1, 96, 64), (0, 64, 1), 0, None, None, False, torch.float32, torch.bfloat16),          # rows (cast)
    ((40, 1, 136), (1, 0, 40), 0, None, None, False, torch.bfloat16, torch.bfloat16),        # [n][k] -> [k][n]
    ((24, 3, 200), (600, 1, 3), 0, None, None, False, torch.float32, torch.bfloat16),        # conv (O,I,3) -> (O,3,I): short X
    ((24, 200, 3), (600, 1, 200), 0, None, None, True, torch.float32, torch.float32),        # gradient un-layout: short c, accumulate
    ((72, 3, 80), (3, -1, 216), 2, None, None, False, torch.float32, torch.bfloat16),        # flipped taps: negative stride, stride-3 X
    ((4, 32, 72), (72 * 20, 1, 20), 0, 20, None, False, torch.float32, torch.bfloat16),      # per-head projection, zero-padded head dim
    ((5, 7, 11), (77, 11, 1), 0, None, 9, False, torch.float32, torch.float32),              # small generic job with column padding
    ((2, 130, 70), (130 * 70, 1, 130), 0, None, None, True, torch.bfloat16, torch.float32),  # ragged tiles, bf16 in, accumulate
    ((1, 96, 160), (0, 160, 1), 0, None, None, False, torch.float32, torch.float32),         # dense copy: linear cast path, f32 out
    ((3, 40, 64), (2560, 64, 1), 0, None, None, False, torch.bfloat16, torch.bfloat16),      # dense 3-d copy (cast path)
    ((136, 1, 200), (1, 0, 136), 0, None, None, False, torch.bfloat16, torch.bfloat16),      # [n][k] -> [k][n], vector transpose path, ragged tiles
    ((3, 72, 128), (128 * 88, 1, 88), 0, 64, None, False, torch.float32, torch.bfloat16),    # per-head projection (vector transpose, f32 in, head dim padded 64 -> 72)
    ((192, 2, 80), (1, 80 * 192, 192), 0, None, 72, False, torch.float32, torch.bfloat16),   # W_o form: X = dim 0, columns beyond 72 zero
    ((3, 68, 132), (132 * 72, 1, 72), 0, None, None, True, torch.float32, torch.float32),    # gradient un-layout of a per-head projection: f32 vector transpose, accumulate, ragged tiles
    ((100, 1, 36), (1, 0, 100), 0, None, None, True, torch.float32, torch.float32),          # [n][k] -> [k][n] f32 accumulate
]


@pytest.mark.parametrize('case', range(len(CASES)))
def test_permute_batch_paths(dev, case):
    dims, st, off, v1, v2, acc, ti, to = CASES[case]
    g = torch.Generator().manual_seed(case)
    d0, d1, d2 = dims
    lo = off + sum(min(0, s * (d - 1)) for s, d in zip(st, dims))
    hi = off + sum(max(0, s * (d - 1)) for s, d in zip(st, dims))
    assert lo >= 0
    src = torch.randn(hi + 1 + 8, generator=g).to(ti)
    idx = off + st[0] * torch.arange(d0)[:, None, None] + st[1] * torch.arange(d1)[None, :, None] + st[2] * torch.arange(d2)[None, None, :]
    view = src.float()[idx]
    if v1 is not None:
        view[:, v1:, :] = 0
    if v2 is not None:
        view[:, :, v2:] = 0
    base = torch.randn(d0, d1, d2, generator=g).to(to)
    want = (base.float() + 0.5 * view if acc else 0.5 * view).to(to)
    src_d, out_d = src.to(dev), base.clone().to(dev)
    b = ops.PermuteBatch()
    b.add(src_d[off:], out_d, dims, st, valid1=v1, valid2=v2, scale=0.5, accumulate=acc)
    b.run(dev)
    got = out_d.cpu()
    if to == torch.float32 and ti == torch.float32:
        assert torch.equal(got, want)
    else:
        assert torch.allclose(got.float(), want.float(), rtol=1e-2, atol=1e-6)
