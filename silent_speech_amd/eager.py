"""Stand-alone forward passes of the reference's sub-modules, composed from the per-kernel C-ABI calls:

    ResBlock.forward(x)                       architecture.py:29-40      x (B, C_in, T)  -> (B, C_out, T / stride)
    MultiHeadAttention.forward(x)             transformer.py:87-112      x (T, B, d)     -> (T, B, d)
    TransformerEncoderLayer.forward(src, ..)  transformer.py:43-60       src (T, B, d)   -> (T, B, d)
    TransformerEncoder.forward(src)           nn.TransformerEncoder      layers in sequence, no final norm

`Model` never calls these: its whole forward / backward is ONE native plan (engine.py, csrc/plan.hip) with fused epilogues,
saved activations and a hand-derived backward.  They exist so that the reference's module API is callable piece by piece --
`model.conv_blocks(x)`, `layer.self_attn(x)`, probing an encoder layer of a loaded checkpoint -- on the same HIP kernels.
They are INFERENCE forwards: no autograd graph is recorded (training goes through Model), BatchNorm uses the running statistics
in eval() mode and batch statistics (with the running-stat update) in train() mode, dropout is applied in train() mode with the
kernels' counter-based draws.  Layout changes of weights (`ss_permute3d`) happen on every call: this is not a hot path.
The math runs in the dtype of the input (float32 -> exact-f32 MFMA kernels, bfloat16 -> bf16 MFMA); like everything else in the
package there is no CPU / eager-PyTorch fallback.
"""
import math

import torch

from . import _lib, ops

RM = ops.rowmap
_seed = [0x5EED0000]


def _round_up(x, m):
    return (x + m - 1) // m * m


def _check(x):
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('float32 or bfloat16 input expected, got %s' % x.dtype)
    if not _lib.kernels_can_read(x):
        raise RuntimeError('silent_speech_amd: tensor is on %s; the HIP kernels need an AMD GPU tensor (no CPU fallback exists)' % x.device)


def _next_seed():
    _seed[0] += 1
    return _seed[0]


def _linear(x2d, weight, bias, relu=False, dropout_p=0.0, seed=0, rng_stream=0):
    """y = act(x W^T + b) on the GEMM kernels (weight in nn.Linear layout [N][K], cast to the activation dtype)."""
    M, K = x2d.shape
    N = weight.shape[0]
    w = weight.detach()
    if w.dtype != x2d.dtype:
        wc = torch.empty(N, K, dtype=x2d.dtype, device=x2d.device)
        ops.permute3d(w, wc, (1, N, K), (0, K, 1))
        w = wc
    y = torch.empty(M, N, dtype=x2d.dtype, device=x2d.device)
    ops.gemm(x2d, w, y, M, N, K, RM(K), RM(K), RM(N), bias=None if bias is None else bias.detach().float(), relu=relu,
             dropout_p=dropout_p, seed=seed, rng_stream=rng_stream)
    return y


# ---------------------------------------------------------------------------------------------- attention
def mha_forward(m, x):
    """transformer.py:87-112 on the banded relative-position attention kernel (csrc/attention.hip)."""
    _check(x)
    T, B, d = x.shape
    dt, dev = x.dtype, x.device
    H, dh = m.n_head, m.d_qkv
    if d != m.d_model:
        raise ValueError('expected d_model = %d, got %d' % (m.d_model, d))
    D = m.relative_positional.max_relative_pos
    dp = _round_up(dh, 32)
    M = B * T
    xb = x.transpose(0, 1).contiguous().view(M, d)                               # (B, T, d): the kernels' frame-major layout
    wqkv = torch.zeros(3, H, dp, d, dtype=dt, device=dev)
    for i, w in enumerate((m.w_q, m.w_k, m.w_v)):                                 # (H, d, dh) -> [h][a (padded to dp)][f]
        ops.permute3d(w.detach(), wqkv[i], (H, dp, d), (d * dh, 1, dh), valid1=dh)
    qkv = torch.empty(M, 3 * H * dp, dtype=dt, device=dev)
    ops.gemm(xb, wqkv.view(3 * H * dp, d), qkv, M, 3 * H * dp, d, RM(d), RM(d), RM(3 * H * dp))
    Tp = _round_up(T, 8)
    qkvT = None
    if _lib.lib().ss_relpos_attention_needs_transposed(_lib.dtype_code(dt), T, dp, D):
        qkvT = torch.zeros(B, 3 * H * dp, Tp, dtype=dt, device=dev)               # the per-tile kernels (f32, long sequences) read K / V time-contiguous
        qkvT[:, :, :T] = qkv.view(B, T, 3 * H * dp).transpose(1, 2)
    emb = m.relative_positional.embeddings.detach()                              # (H, 2D-1, dh, 1)
    E = torch.zeros(H, 2 * D - 1, dp, dtype=dt, device=dev)
    ops.permute3d(emb, E, (H, 2 * D - 1, dp), ((2 * D - 1) * dh, dh, 1), valid2=dh)
    out = torch.empty(M, H * dp, dtype=dt, device=dev)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=dev)
    p = float(m.dropout.p) if m.training else 0.0
    tab = None
    if ops.relpos_attention_family(dt, T, dp, D) == 2:                           # the transposed-score kernels read E / scale in fragment order, from the f32 parameter
        tab = ops.relpos_attention_tables(emb, dp, 1.0 / math.sqrt(dh))
    ops.relpos_attention_forward(qkv, qkvT, E, out, lse, B, H, T, Tp, dp, D, 1.0 / math.sqrt(dh), p=p, seed=_next_seed(), rng_stream=1, tab=tab)
    wo = torch.zeros(d, H * dp, dtype=dt, device=dev)                             # (H, dh, d) -> [f][h][a (padded)]
    ops.permute3d(m.w_o.detach(), wo, (d, H, dp), (1, dh * d, d), valid2=dh)
    y = torch.empty(M, d, dtype=dt, device=dev)
    ops.gemm(out, wo, y, M, d, H * dp, RM(H * dp), RM(H * dp), RM(d))
    return y.view(B, T, d).transpose(0, 1)


# ---------------------------------------------------------------------------------------------- encoder layer
def encoder_layer_forward(m, src, src_mask=None, src_key_padding_mask=None, is_causal=False):
    """transformer.py:43-60 (post-norm; the mask arguments are accepted and ignored, like the reference)."""
    _check(src)
    T, B, d = src.shape
    M = B * T
    dt, dev = src.dtype, src.device
    p = float(m.dropout.p) if m.training else 0.0
    seed = _next_seed()
    x = src.transpose(0, 1).contiguous().view(M, d)
    a = mha_forward(m.self_attn, src).transpose(0, 1).contiguous().view(M, d)
    y1 = torch.empty(M, d, dtype=dt, device=dev)
    ops.add_dropout_layernorm(x, a, m.norm1.weight.detach().float(), m.norm1.bias.detach().float(), y1, M, d, eps=m.norm1.eps,
                              p=float(m.dropout1.p) if m.training else 0.0, seed=seed, rng_stream=2)
    h = _linear(y1, m.linear1.weight, m.linear1.bias, relu=True, dropout_p=p, seed=seed, rng_stream=3)
    f = _linear(h, m.linear2.weight, m.linear2.bias)
    y2 = torch.empty(M, d, dtype=dt, device=dev)
    ops.add_dropout_layernorm(y1, f, m.norm2.weight.detach().float(), m.norm2.bias.detach().float(), y2, M, d, eps=m.norm2.eps,
                              p=float(m.dropout2.p) if m.training else 0.0, seed=seed, rng_stream=4)
    return y2.view(B, T, d).transpose(0, 1)


def encoder_forward(m, src, mask=None, src_key_padding_mask=None):
    for layer in m.layers:
        src = encoder_layer_forward(layer, src)
    return src


# ---------------------------------------------------------------------------------------------- ResBlock
def _bn(bn, x, B, T, C, pad):
    """(mean, invstd, gamma, beta) of one BatchNorm1d over a (B, T (+2 pad), C) buffer: batch statistics + running-stat update in
    train() mode, the running statistics in eval() mode (nn.BatchNorm1d semantics, architecture.py:19,21,25)."""
    scratch = ops.bn_scratch(B, T, C, x.device) if bn.training else None
    mean, invstd = ops.bn_stats(x, B, T, C, pad, scratch, bn.running_mean, bn.running_var, momentum=bn.momentum, eps=bn.eps, training=bn.training)
    if bn.training:
        bn.num_batches_tracked += 1
    return mean, invstd, bn.weight.detach().float(), bn.bias.detach().float()


def resblock_forward(m, x):
    """architecture.py:29-40.  x (B, C_in, T) like nn.Conv1d; convolutions are GEMMs over the overlapping 3C-wide rows of a
    zero-padded (B, T + 2, C) buffer (csrc/gemm.hip RowMap), BatchNorm + ReLU (+ the residual branch) one fused kernel."""
    _check(x)
    B, Ci, T = x.shape
    dt, dev = x.dtype, x.device
    Co, s = m.conv1.out_channels, m.stride
    if Ci != m.conv1.in_channels:
        raise ValueError('expected %d input channels, got %d' % (m.conv1.in_channels, Ci))
    To = (T - 1) // s + 1                                                         # Conv1d(k = 3, padding = 1, stride = s)
    xp = torch.zeros(B, T + 2, Ci, dtype=dt, device=dev)
    xp[:, 1:-1] = x.transpose(1, 2)
    # conv1 (stride s): [o][tap * Ci + i] <- (o, i, tap)
    w1 = torch.empty(Co, 3 * Ci, dtype=dt, device=dev)
    ops.permute3d(m.conv1.weight.detach(), w1, (Co, 3, Ci), (3 * Ci, 1, 3))
    y1 = torch.empty(B, To, Co, dtype=dt, device=dev)
    ops.gemm(xp, w1, y1, B * To, Co, 3 * Ci, RM(s * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci), RM(3 * Ci), RM(Co), bias=m.conv1.bias.detach().float())
    a1 = torch.empty(B, To + 2, Co, dtype=dt, device=dev)                          # relu(bn1(.)) into a padded buffer: the input of conv2
    ops.bn_apply(y1, _bn(m.bn1, y1, B, To, Co, 0), 0, a1, 1, B, To, Co, True)
    w2 = torch.empty(Co, 3 * Co, dtype=dt, device=dev)
    ops.permute3d(m.conv2.weight.detach(), w2, (Co, 3, Co), (3 * Co, 1, 3))
    y2 = torch.empty(B, To, Co, dtype=dt, device=dev)
    ops.gemm(a1, w2, y2, B * To, Co, 3 * Co, RM(Co, rows_per_batch=To, batch_stride=(To + 2) * Co), RM(3 * Co), RM(Co), bias=m.conv2.bias.detach().float())
    if m.residual_path is not None:
        wr = torch.empty(Co, Ci, dtype=dt, device=dev)
        ops.permute3d(m.residual_path.weight.detach(), wr, (1, Co, Ci), (0, Ci, 1))
        r = torch.empty(B, To, Co, dtype=dt, device=dev)
        # the 1x1 stride-s convolution reads the CENTRE tap of every window: base offset = one padded row
        ops.gemm(xp, wr, r, B * To, Co, Ci, RM(s * Ci, rows_per_batch=To, batch_stride=(T + 2) * Ci, base=Ci), RM(Ci), RM(Co), bias=m.residual_path.bias.detach().float())
        sb = _bn(m.res_norm, r, B, To, Co, 0)
        pad_r = 0
    else:
        r, pad_r = xp, 1                                                          # identity branch: the (padded) input itself, through a unit affine map
        one, zero = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        sb = (zero, one, one, zero)
    out = torch.empty(B, To, Co, dtype=dt, device=dev)
    ops.bn_apply(y2, _bn(m.bn2, y2, B, To, Co, 0), 0, out, 0, B, To, Co, True, xb=r, sb=sb, pad_xb=pad_r)
    return out.transpose(1, 2)
