"""BatchNorm / LayerNorm / AdamW / EMG-prepare kernels vs torch fp32 autograd and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import adamw_ref
from silent_speech_amd import _lib, ops
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust

DTS = [torch.float32, torch.bfloat16]


def _tol(dt, f32=2e-5, bf=2e-2):
    return f32 if dt == torch.float32 else bf


def _pad(x):           # (B,T,C) -> (B,T+2,C) with zero halo
    B, T, C = x.shape
    o = torch.zeros(B, T + 2, C, dtype=x.dtype)
    o[:, 1:-1] = x
    return o


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('shape', ['fixed', 'generic', 'wide'])
def test_batchnorm_resblock_tail_fwd_bwd(dev, dt, shape):
    """y = relu(bn2(xa) + res_norm(xb)) with padded input/outputs; forward, running stats, full backward.  'fixed': the grid stride is a
    multiple of C/8 (a thread keeps its column chunk, constants in registers); 'generic': C = 24, where it cannot be (per-chunk form);
    'wide': the model's 768 channels."""
    big = not is_emu(dev)
    B, T, C = {'fixed': ((5, 130, 96), (3, 20, 16)), 'generic': ((3, 50, 24), (2, 9, 24)), 'wide': ((2, 33, 768), (1, 6, 768))}[shape][0 if big else 1]
    g = torch.Generator().manual_seed(1)
    xa = (torch.randn(B, T, C, generator=g) * 2 + 0.5).to(dt); xb = torch.randn(B, T, C, generator=g).to(dt)
    ga, ba = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gb, bb = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dy = torch.randn(B, T, C, generator=g).to(dt)
    # torch reference (fp32 math on the same rounded inputs)
    xa_r = xa.float().requires_grad_(True); xb_r = xb.float().requires_grad_(True)
    ga_r, ba_r, gb_r, bb_r = [t.clone().requires_grad_(True) for t in (ga, ba, gb, bb)]
    rm_r, rv_r = rm.clone(), rv.clone()
    ya = F.batch_norm(xa_r.transpose(1, 2), rm_r, rv_r, ga_r, ba_r, True, 0.1, 1e-5)
    yb = F.batch_norm(xb_r.transpose(1, 2), torch.zeros(C), torch.ones(C), gb_r, bb_r, True, 0.1, 1e-5)
    y_ref = torch.relu(ya + yb).transpose(1, 2)
    y_ref.backward(dy.float())
    # HIP
    scratch = ops.bn_scratch(B, T, C, dev)
    xa_d, xb_d = _pad(xa).to(dev), xb.to(dev)                  # xa padded, xb not
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    ma, ia = ops.bn_stats(xa_d, B, T, C, 1, scratch, rm_d, rv_d)
    mb, ib = ops.bn_stats(xb_d, B, T, C, 0, scratch, torch.zeros(C, device=dev), torch.ones(C, device=dev))
    assert_close_robust(rm_d, rm_r, 1e-5, name='running_mean', max_outlier_frac=0)
    assert_close_robust(rv_d, rv_r, 2e-5 if dt == torch.float32 else 1e-4, name='running_var', max_outlier_frac=0)
    y = torch.full((B, T + 2, C), 9.0, dtype=dt, device=dev)
    ops.bn_apply(xa_d, (ma, ia, ga.to(dev), ba.to(dev)), 1, y, 1, B, T, C, True, xb=xb_d, sb=(mb, ib, gb.to(dev), bb.to(dev)), pad_xb=0)
    assert float(y[:, 0].abs().max()) == 0 and float(y[:, -1].abs().max()) == 0      # halo rows zeroed
    assert_close_robust(y[:, 1:-1], y_ref, _tol(dt), name='y', max_outlier_frac=0)
    # backward: dy unpadded, y padded; dxa padded (feeds the conv dX GEMM), dxb unpadded
    dxa = torch.full((B, T + 2, C), 5.0, dtype=dt, device=dev); dxb = torch.empty(B, T, C, dtype=dt, device=dev)
    dga, dba, dgb, dbb = [torch.zeros(C, device=dev) for _ in range(4)]
    ops.bn_backward(dy.to(dev), 0, y, 1, xa_d, 1, (ma, ia, ga.to(dev)), dxa, 1, dga, dba, scratch, B, T, C, True,
                    xb=xb_d, pad_xb=0, sb=(mb, ib, gb.to(dev)), dxb=dxb, pad_dxb=0, dgamma_b=dgb, dbeta_b=dbb)
    assert float(dxa[:, 0].abs().max()) == 0 and float(dxa[:, -1].abs().max()) == 0
    # the ReLU mask comes from the dt-rounded y: compare only where |y_ref| is clear of 0 for bf16
    assert_close_robust(dxa[:, 1:-1], xa_r.grad, _tol(dt, 5e-5, 3e-2), name='dxa', max_outlier_frac=0 if dt == torch.float32 else 5e-3)
    assert_close_robust(dxb, xb_r.grad, _tol(dt, 5e-5, 3e-2), name='dxb', max_outlier_frac=0 if dt == torch.float32 else 5e-3)
    for got, want, n in ((dga, ga_r.grad, 'dgamma_a'), (dba, ba_r.grad, 'dbeta_a'), (dgb, gb_r.grad, 'dgamma_b'), (dbb, bb_r.grad, 'dbeta_b')):
        assert_close_robust(got, want, _tol(dt, 5e-5, 2e-2), name=n, max_outlier_frac=0)
    # the same backward with the ReLU gate RECOMPUTED from xa / xb and the affine parameters (what the plan runs): the saved output is not read
    dxa2 = torch.full((B, T + 2, C), 5.0, dtype=dt, device=dev); dxb2 = torch.empty(B, T, C, dtype=dt, device=dev)
    g2 = [torch.zeros(C, device=dev) for _ in range(4)]
    ops.bn_backward(dy.to(dev), 0, None, 1, xa_d, 1, (ma, ia, ga.to(dev)), dxa2, 1, g2[0], g2[1], scratch, B, T, C, True,
                    xb=xb_d, pad_xb=0, sb=(mb, ib, gb.to(dev)), dxb=dxb2, pad_dxb=0, dgamma_b=g2[2], dbeta_b=g2[3], beta_a=ba.to(dev), beta_b=bb.to(dev))
    assert_close_robust(dxa2[:, 1:-1], xa_r.grad, _tol(dt, 5e-5, 3e-2), name='dxa (gate recomputed)', max_outlier_frac=0 if dt == torch.float32 else 5e-3)
    assert_close_robust(dxb2, xb_r.grad, _tol(dt, 5e-5, 3e-2), name='dxb (gate recomputed)', max_outlier_frac=0 if dt == torch.float32 else 5e-3)
    for got, want, n in zip(g2, (ga_r.grad, ba_r.grad, gb_r.grad, bb_r.grad), ('dgamma_a', 'dbeta_a', 'dgamma_b', 'dbeta_b')):
        assert_close_robust(got, want, _tol(dt, 5e-5, 2e-2), name=n + ' (gate recomputed)', max_outlier_frac=0)
    flips = int(((dxa2[:, 1:-1].float() == 0) != (dxa[:, 1:-1].float() == 0)).sum())       # gates that differ between the two forms (pre-activations within rounding of 0)
    assert flips <= (3 if dt == torch.float32 else B * T * C // 500), flips          # f32: only pre-activations within an ulp of 0 can differ


def test_batchnorm_eval_mode(dev):
    B, T, C = 2, 10, 16
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, C, generator=g); rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    ga, ba = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    want = torch.relu(F.batch_norm(x.transpose(1, 2), rm, rv, ga, ba, False, 0.1, 1e-5)).transpose(1, 2)
    m, i = ops.bn_stats(x.to(dev), B, T, C, 0, None, rm.to(dev), rv.to(dev), training=False)
    y = torch.empty(B, T, C, device=dev)
    ops.bn_apply(x.to(dev), (m, i, ga.to(dev), ba.to(dev)), 0, y, 0, B, T, C, True)
    assert_close_robust(y, want, 1e-5, name='bn_eval', max_outlier_frac=0)


def test_layernorm_plane_outputs(dev):
    """ss_add_dropout_layernorm_forward_planes / ss_layernorm_backward_ws_planes (the parity-grade mode, transformer.py:55-60): y and the branch gradient also
    leave as hi / lo bf16 planes -- bit for bit ss_split_planes of the stored f32 tensors --, with dropout, and the f32 results are those of the plain entry points."""
    rows, C = (9, 768) if is_emu(dev) else (1003, 768)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(rows, C, generator=g); a = torch.randn(rows, C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    xd, ad = x.to(dev), a.clone().to(dev)
    y = torch.empty(rows, C, device=dev); hi = torch.zeros(rows, C, dtype=torch.bfloat16, device=dev); lo = torch.zeros_like(hi)
    mean, rstd = ops.add_dropout_layernorm(xd, ad, gamma.to(dev), beta.to(dev), y, rows, C, p=0.2, seed=5, rng_stream=3, planes=(hi, lo))
    a2 = a.clone().to(dev); y2 = torch.empty_like(y)
    ops.add_dropout_layernorm(xd, a2, gamma.to(dev), beta.to(dev), y2, rows, C, p=0.2, seed=5, rng_stream=3)
    assert torch.equal(y, y2) and torch.equal(ad, a2)
    wh, wl = ops.split_planes(y)
    assert torch.equal(hi.view(torch.int16), wh.view(torch.int16).view(rows, C)) and torch.equal(lo.view(torch.int16), wl.view(torch.int16).view(rows, C))
    dres = dy.clone().to(dev); dbr = torch.empty(rows, C, device=dev); bh = torch.zeros_like(hi); bl = torch.zeros_like(hi)
    dg, db, cs = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_backward(dres, ad, mean, rstd, gamma.to(dev), dres, dbr, dg, db, rows, C, p=0.2, seed=5, rng_stream=3, dbranch_colsum=cs, planes=(bh, bl))
    dres2 = dy.clone().to(dev); dbr2 = torch.empty_like(dbr)
    dg2, db2, cs2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_backward(dres2, ad, mean, rstd, gamma.to(dev), dres2, dbr2, dg2, db2, rows, C, p=0.2, seed=5, rng_stream=3, dbranch_colsum=cs2)
    assert torch.equal(dres, dres2) and torch.equal(dbr, dbr2)
    wh, wl = ops.split_planes(dbr)
    assert torch.equal(bh.view(torch.int16), wh.view(torch.int16).view(rows, C)) and torch.equal(bl.view(torch.int16), wl.view(torch.int16).view(rows, C))
    with pytest.raises(RuntimeError, match='planes'):                       # bf16 data has no plane output
        ops.add_dropout_layernorm(xd.bfloat16(), ad.bfloat16(), gamma.to(dev), beta.to(dev), y.bfloat16(), rows, C, planes=(hi, lo))


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('C', [64, 768])
def test_add_dropout_layernorm_fwd_bwd(dev, dt, C):
    rows = 37 if is_emu(dev) else 1000
    if is_emu(dev) and C == 768:
        rows = 9
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, C, generator=g).to(dt); a = torch.randn(rows, C, generator=g).to(dt)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g).to(dt)
    xr, ar = x.float().requires_grad_(True), a.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr + ar, (C,), gr, br, 1e-5)
    y_ref.backward(dy.float())
    xd, ad = x.to(dev), a.clone().to(dev)
    y = torch.empty(rows, C, dtype=dt, device=dev)
    mean, rstd = ops.add_dropout_layernorm(xd, ad, gamma.to(dev), beta.to(dev), y, rows, C)
    assert_close_robust(y, y_ref, _tol(dt), name='ln_y', max_outlier_frac=0)
    assert_close_robust(ad, (x.float() + a.float()), _tol(dt, 1e-6, 1e-2), name='z', max_outlier_frac=0)
    dyd = dy.to(dev)
    dbranch = torch.empty(rows, C, dtype=dt, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_backward(dyd, ad, mean, rstd, gamma.to(dev), dyd, dbranch, dg, db, rows, C)       # in place: dres aliases dy
    assert_close_robust(dyd, xr.grad, _tol(dt, 5e-5, 3e-2), name='dres', max_outlier_frac=0)
    assert torch.equal(dyd, dbranch)                                                               # p = 0
    assert_close_robust(dg, gr.grad, _tol(dt, 5e-5, 2e-2), name='dgamma', max_outlier_frac=0)
    assert_close_robust(db, br.grad, _tol(dt, 5e-5, 2e-2), name='dbeta', max_outlier_frac=0)


def test_layernorm_dropout_mask_consistency(dev):
    """forward and backward regenerate the same Philox mask; keep fraction ~ 1-p; scale 1/(1-p)."""
    rows, C, p = 24, 64, 0.2
    x = torch.zeros(rows, C); a = torch.ones(rows, C)
    ad = a.clone().to(dev)
    y = torch.empty(rows, C, device=dev)
    mean, rstd = ops.add_dropout_layernorm(x.to(dev), ad, torch.ones(C, device=dev), torch.zeros(C, device=dev), y, rows, C, p=p, seed=77, rng_stream=5)
    z = ad.cpu()
    kept = z != 0
    assert torch.allclose(z[kept], torch.full_like(z[kept], 1.25))
    assert abs(kept.float().mean().item() - 0.8) < 0.05
    dy = torch.randn(rows, C)
    dres = torch.empty(rows, C, device=dev); dbr = torch.empty(rows, C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_backward(dy.to(dev), ad, mean, rstd, torch.ones(C, device=dev), dres, dbr, dg, db, rows, C, p=p, seed=77, rng_stream=5)
    assert torch.allclose(dbr.cpu(), dres.cpu() * kept.float() * 1.25, atol=1e-6)


@pytest.mark.parametrize('dt', DTS)
@pytest.mark.parametrize('C', [256, 768])
def test_layernorm_backward_forms_agree(dev, dt, C, monkeypatch):
    """The 16-wave scratch form (C = 256 / 512 / 768) and the atomic form (SS_LN_BWD2=0) of the LayerNorm backward: same dres / dbranch
    (same dropout mask) and dgamma / dbeta / branch column sums up to summation order; the sums ACCUMULATE into their destinations."""
    rows, p = (41 if is_emu(dev) else 3001), 0.25
    g = torch.Generator().manual_seed(8)
    z = torch.randn(rows, C, generator=g).to(dt); dy = torch.randn(rows, C, generator=g).to(dt)
    gamma = torch.rand(C, generator=g) + 0.5
    zf = z.float(); mean = zf.mean(1); rstd = (zf.var(1, unbiased=False) + 1e-5).rsqrt()
    assert int(_lib.lib().ss_layernorm_backward_scratch_floats(rows, C)) > 0
    res = {}
    for form in ('1', '0'):
        monkeypatch.setenv('SS_LN_BWD2', form)
        dres = torch.empty(rows, C, dtype=dt, device=dev); dbr = torch.empty(rows, C, dtype=dt, device=dev)
        dg, db, dbs = [torch.full((C,), 3.0, device=dev) for _ in range(3)]
        ops.layernorm_backward(dy.to(dev), z.to(dev), mean.to(dev), rstd.to(dev), gamma.to(dev), dres, dbr, dg, db, rows, C, p=p, seed=5, rng_stream=9, dbranch_colsum=dbs)
        res[form] = [t.float().cpu() for t in (dres, dbr, dg, db, dbs)]
    tol = 2e-6 if dt == torch.float32 else 1e-2                              # the row sums are taken in a different lane order
    assert_close_robust(res['1'][0], res['0'][0], tol, name='dres', max_outlier_frac=0)
    assert_close_robust(res['1'][1], res['0'][1], tol, name='dbranch', max_outlier_frac=0)
    assert torch.equal(res['1'][1] == 0, res['0'][1] == 0)                     # the same dropped set
    for i, name in ((2, 'dgamma'), (3, 'dbeta'), (4, 'dbranch_colsum')):
        assert_close_robust(res['1'][i], res['0'][i], 2e-5 if dt == torch.float32 else 1e-3, name=name, max_outlier_frac=0)   # bf16: a dbranch entry may round the other way
    assert_close_robust(res['1'][4], 3.0 + res['1'][1].sum(0), 1e-4, name='colsum of dbranch', max_outlier_frac=0)


@pytest.mark.parametrize('dt', DTS)
def test_colsum(dev, dt):
    rows, C = (300, 64) if is_emu(dev) else (5000, 768)
    x = torch.randn(rows, C, generator=torch.Generator().manual_seed(4)).to(dt)
    out = torch.ones(C, device=dev)
    ops.colsum(x.to(dev), rows, C, C, out)
    assert_close_robust(out, 1 + x.float().sum(0), 1e-4, name='colsum', max_outlier_frac=0)


@pytest.mark.parametrize('r', [0, 3, 7])
def test_emg_prepare_shift(dev, r):
    """architecture.py:64-68 shift-left-by-r with zero tail, + halo rows + cast."""
    B, T0 = 2, 40
    x = torch.randn(B, T0, 8, generator=torch.Generator().manual_seed(5))
    want = x.clone()
    if r > 0:
        want[:, :-r] = x[:, r:]
        want[:, -r:] = 0
    out = torch.full((B, T0 + 2, 8), 3.0, dtype=torch.bfloat16, device=dev)
    sh = torch.empty(B, T0, 8, device=dev)
    ops.emg_prepare(x.to(dev), out, sh, B, T0, 8, r)
    assert torch.equal(sh.cpu(), want)
    assert torch.equal(out[:, 1:-1].cpu(), want.to(torch.bfloat16))
    assert float(out[:, 0].float().abs().max()) == 0 and float(out[:, -1].float().abs().max()) == 0


def test_adamw_golden_and_oracle(dev, golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, 'adamw.npz'))
    n = z['p'].shape[1]
    p = torch.from_numpy(z['p'][0]).clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for it in range(3):
        lr = adamw_ref.warmup_lr(it)
        ops.adamw_step(p, torch.from_numpy(z['g'][it]).clone().to(dev), m, v, n, lr, it + 1, weight_decay=1e-7)
        assert_close_robust(p, z['p'][it + 1], 2e-6, name='adamw step %d' % it, max_outlier_frac=0)   # vs torch.optim.AdamW
