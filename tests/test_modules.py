"""The reference's sub-modules called on their own (eager.py: one C-ABI call per kernel) against the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model_ref
from silent_speech_amd.architecture import Model, ResBlock
from silent_speech_amd.transformer import MultiHeadAttention, TransformerEncoderLayer
from tests.backend import dev, is_emu  # noqa: F401
from tests.util import assert_close_robust

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('T', [50, 100, 200, 250])
def test_mha_forward_matches_reference_golden(dev, T):
    """MultiHeadAttention.forward (transformer.py:87-112) against outputs of the reference's own module (tests/golden/mha_T*.npz)."""
    if is_emu(dev) and T > 100:
        pytest.skip('emulator: small sequence lengths only')
    z = np.load(os.path.join(GOLD, 'mha_T%d.npz' % T))
    H, d, dh = z['w_q'].shape
    m = MultiHeadAttention(d, H, dropout=0.1, relative_positional=True, relative_positional_distance=100)
    with torch.no_grad():
        for n in ('w_q', 'w_k', 'w_v', 'w_o'):
            getattr(m, n).copy_(torch.from_numpy(z[n]))
        m.relative_positional.embeddings.copy_(torch.from_numpy(z['E']))
    m = m.to(dev).eval()
    x = torch.from_numpy(z['x']).to(dev)
    out = m(x)
    assert tuple(out.shape) == tuple(z['out'].shape)
    assert_close_robust(out, z['out'], 5e-5, name='mha out', max_outlier_frac=0)
    outb = m(x.to(torch.bfloat16))                                    # the bf16 MFMA kernels (LDS-resident attention for T <= 208)
    assert outb.dtype == torch.bfloat16
    assert_close_robust(outb.float(), z['out'], 4e-2, name='mha out bf16', max_outlier_frac=0)


def test_encoder_layer_forward_matches_oracle(dev):
    T, B, d, H, ff = (24, 2, 16, 2, 32) if is_emu(dev) else (200, 3, 64, 8, 128)
    torch.manual_seed(5)
    layer = TransformerEncoderLayer(d, H, dim_feedforward=ff, dropout=0.2).eval()
    with torch.no_grad():
        layer.norm1.weight.uniform_(0.5, 1.5); layer.norm1.bias.normal_(); layer.norm2.weight.uniform_(0.5, 1.5); layer.norm2.bias.normal_()
    sd = {'transformer.layers.0.' + k: v.detach().clone() for k, v in layer.state_dict().items()}
    x = torch.randn(T, B, d)
    want = model_ref.encoder_layer(x.transpose(0, 1), sd, 'transformer.layers.0').transpose(0, 1).detach()      # the oracle works on (B, T, d)
    got = layer.to(dev)(x.to(dev))
    assert_close_robust(got, want, 1e-4, name='encoder layer', max_outlier_frac=0)
    got2 = layer(x.to(dev), None, None)                              # mask arguments accepted and ignored (transformer.py:43)
    assert_close_robust(got2, want, 1e-4, name='encoder layer (masks)', max_outlier_frac=0)


@pytest.mark.parametrize('cfg', [(8, 16, 2), (16, 16, 1), (16, 24, 1)])
@pytest.mark.parametrize('training', [False, True])
def test_resblock_forward_matches_torch(dev, cfg, training):
    """ResBlock.forward (architecture.py:29-40): strided with projection, identity residual, projection without stride; eval mode uses
    the running statistics, train mode batch statistics and updates them like nn.BatchNorm1d."""
    Ci, Co, s = cfg
    B, T = (2, 24) if is_emu(dev) else (3, 200)
    torch.manual_seed(11)
    blk = ResBlock(Ci, Co, s)
    with torch.no_grad():
        for bn in [blk.bn1, blk.bn2] + ([blk.res_norm] if blk.residual_path is not None else []):
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    blk.train(training)
    x = torch.randn(B, Ci, T)

    def bn(m, v, rm, rv):
        return F.batch_norm(v, rm, rv, m.weight, m.bias, training, m.momentum, m.eps)
    rms = {k: v.clone() for k, v in blk.state_dict().items() if 'running' in k}
    h = F.relu(bn(blk.bn1, F.conv1d(x, blk.conv1.weight, blk.conv1.bias, stride=s, padding=1), rms['bn1.running_mean'], rms['bn1.running_var']))
    h = bn(blk.bn2, F.conv1d(h, blk.conv2.weight, blk.conv2.bias, padding=1), rms['bn2.running_mean'], rms['bn2.running_var'])
    if blk.residual_path is not None:
        r = bn(blk.res_norm, F.conv1d(x, blk.residual_path.weight, blk.residual_path.bias, stride=s), rms['res_norm.running_mean'], rms['res_norm.running_var'])
    else:
        r = x
    want = F.relu(h + r).detach()
    blk = blk.to(dev)
    got = blk(x.to(dev))
    assert tuple(got.shape) == tuple(want.shape)
    assert_close_robust(got, want, 1e-4, name='resblock', max_outlier_frac=0)
    if training:
        for k, v in rms.items():
            assert_close_robust(blk.state_dict()[k], v, 1e-4, name=k, max_outlier_frac=0)


def test_model_submodules_are_callable(dev):
    """`model.conv_blocks(x)` and `model.transformer(x)` run (nn.Sequential / TransformerEncoder of the callable sub-modules) and compose
    to the eval-mode Model.forward."""
    torch.manual_seed(3)
    model = Model(112, 80, 48, model_size=16, num_layers=1, dropout=0.0, compute_dtype=torch.float32).to(dev).eval()
    B, T = 2, 16
    x_raw = torch.randn(B, 8 * T, 8).to(dev)
    with torch.no_grad():
        pred, aux = model(None, x_raw, None)
        h = model.conv_blocks(x_raw.transpose(1, 2))                 # architecture.py:70-72
        assert tuple(h.shape) == (B, 16, T)
        h = torch.nn.functional.linear(h.transpose(1, 2).cpu(), model.w_raw_in.weight.cpu(), model.w_raw_in.bias.cpu()).to(dev)
        h = model.transformer(h.transpose(0, 1).contiguous()).transpose(0, 1)
        want = torch.nn.functional.linear(h.cpu(), model.w_out.weight.cpu(), model.w_out.bias.cpu())
    assert_close_robust(pred, want, 2e-4, name='composed forward', max_outlier_frac=0)
