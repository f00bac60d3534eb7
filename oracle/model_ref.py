"""Oracle: EMG->mel transduction model forward (fp32, torch CPU).  TEST INFRASTRUCTURE ONLY.

Functional restatement (parameters come in as a reference-format state_dict) of
  architecture.py:14-40  ResBlock           -> resblock()
  architecture.py:61-84  Model.forward      -> model_forward()
  transformer.py:43-60   encoder layer      -> encoder_layer()   (post-norm)
  transformer.py:87-112  MultiHeadAttention -> mha()
  transformer.py:162-297 LearnedRelativePositionalEmbedding (unmasked, per-head, keys only)
                                             -> relpos_logits()  in CLOSED FORM:
      pos[b,h,q,k] = sum_a Q[b,h,q,a] * E[h, k-q+D-1, a]   if |k-q| <= D-1   (D = 100)
                   = -1e8                                  otherwise
  (the reference builds this with zero-padding under no_grad, an einsum over 2T-1 relative
  positions, in-place `-= 1e8` on the padded columns and a pad/view skew; the closed form is what
  those steps compute, and E receives no gradient because the padding is done under no_grad
  (transformer.py:214-218)).
Autograd through this module is the gradient oracle.
"""
import math
import torch
import torch.nn.functional as F

MAX_REL = 100


class _RoundBf16(torch.autograd.Function):
    """x -> bf16 -> f32 in the forward AND on the incoming gradient: an activation that is stored in bf16 between two kernels."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


_STORAGE = [None]


class bf16_storage(object):
    """Context manager: inside it the oracle rounds every activation that crosses a kernel boundary in the HIP plan (conv / linear
    outputs, BatchNorm / LayerNorm outputs, q k v, the attention probabilities and outputs) and every weight operand to bf16 --
    arithmetic stays f32.  This is NOT the parity oracle: it is the yardstick that says how far ANY bf16-storage implementation
    of the reference sits from the f32 reference (ReLU gates flip where a pre-activation is within bf16 rounding of 0), against
    which the bf16 kernels' deviation is judged (tests/test_fullsize.py)."""

    def __enter__(self):
        _STORAGE[0] = 'bf16'

    def __exit__(self, *a):
        _STORAGE[0] = None


def _st(x):
    return _RoundBf16.apply(x) if _STORAGE[0] == 'bf16' else x


def shift_left_(x_raw, r):
    """architecture.py:64-68 (in place on the input, same r for the whole batch)."""
    if r > 0:
        x_raw[:, :-r, :] = x_raw[:, r:, :].clone()
        x_raw[:, -r:, :] = 0
    return x_raw


def _bn(x, sd, prefix, training, running_out):
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    if training:
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        if running_out is not None:
            n = x.shape[0] * x.shape[2]
            rm = sd[prefix + '.running_mean'] * 0.9 + 0.1 * mean.detach()
            rv = sd[prefix + '.running_var'] * 0.9 + 0.1 * var.detach() * (n / max(n - 1, 1))
            running_out[prefix + '.running_mean'] = rm
            running_out[prefix + '.running_var'] = rv
    else:
        mean, var = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    xh = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5)
    return xh * w[None, :, None] + b[None, :, None]


def resblock(x, sd, p, stride, training, running_out=None):
    """x: (B, C_in, T).  architecture.py:29-40."""
    h = _st(F.conv1d(x, _st(sd[p + '.conv1.weight']), sd[p + '.conv1.bias'], stride=stride, padding=1))
    h = _st(F.relu(_bn(h, sd, p + '.bn1', training, running_out)))
    h = _st(F.conv1d(h, _st(sd[p + '.conv2.weight']), sd[p + '.conv2.bias'], stride=1, padding=1))
    h = _bn(h, sd, p + '.bn2', training, running_out)
    if (p + '.residual_path.weight') in sd:
        r = _st(F.conv1d(x, _st(sd[p + '.residual_path.weight']), sd[p + '.residual_path.bias'], stride=stride))
        r = _bn(r, sd, p + '.res_norm', training, running_out)
    else:
        r = x
    return _st(F.relu(h + r))


def relpos_logits(q, E, max_rel=MAX_REL):
    """q: (B,H,T,dh) UNSCALED queries, E: (H, 2*max_rel-1, dh, 1) -> (B,H,T,T)."""
    B, H, T, dh = q.shape
    Em = E[..., 0].detach()                                   # no grad reaches E (transformer.py:214)
    rel = torch.einsum('bhqa,hma->bhqm', q, Em)               # (B,H,T,2D-1)
    kq = torch.arange(T)[None, :] - torch.arange(T)[:, None]  # k - q
    idx = (kq + (max_rel - 1)).clamp(0, 2 * max_rel - 2)
    inband = (kq.abs() <= max_rel - 1)
    pos = torch.gather(rel, 3, idx[None, None].expand(B, H, T, T))
    return torch.where(inband[None, None], pos, torch.full_like(pos, -1e8))


def mha(x, sd, p, dropout_p=0.0, drop_mask=None):
    """x: (B,T,d) -> (B,T,d).  transformer.py:87-112 (layout (T,B,d) there; math is per row)."""
    wq, wk, wv, wo = sd[p + '.w_q'], sd[p + '.w_k'], sd[p + '.w_v'], sd[p + '.w_o']
    dh = wq.shape[2]
    q = _st(torch.einsum('btf,hfa->bhta', x, _st(wq)))
    k = _st(torch.einsum('btf,hfa->bhta', x, _st(wk)))
    v = _st(torch.einsum('btf,hfa->bhta', x, _st(wv)))
    logits = torch.einsum('bhqa,bhka->bhqk', q, k) / (dh ** 0.5)
    logits = logits + relpos_logits(q, sd[p + '.relative_positional.embeddings'])
    probs = _st(F.softmax(logits, dim=-1))
    if drop_mask is not None:
        probs = probs * drop_mask / (1.0 - dropout_p)
    o = _st(torch.einsum('bhqk,bhka->bhqa', probs, v))
    return _st(torch.einsum('bhta,haf->btf', o, _st(wo)))


def encoder_layer(x, sd, p, masks=None, dropout_p=0.0):
    """Post-norm layer, transformer.py:54-59.  masks: optional dict of keep-masks
    {'attn','res1','ffn','res2'} (1 = keep) to replay a given dropout pattern."""
    m = masks or {}
    sc = 1.0 / (1.0 - dropout_p) if masks else 1.0
    a = mha(x, sd, p + '.self_attn', dropout_p, m.get('attn'))
    if 'res1' in m:
        a = a * m['res1'] * sc
    x = _st(F.layer_norm(_st(x + a), (x.shape[-1],), sd[p + '.norm1.weight'], sd[p + '.norm1.bias'], 1e-5))
    h = F.relu(F.linear(x, _st(sd[p + '.linear1.weight']), sd[p + '.linear1.bias']))
    if 'ffn' in m:
        h = h * m['ffn'] * sc
    f = _st(F.linear(_st(h), _st(sd[p + '.linear2.weight']), sd[p + '.linear2.bias']))
    if 'res2' in m:
        f = f * m['res2'] * sc
    return _st(F.layer_norm(_st(x + f), (x.shape[-1],), sd[p + '.norm2.weight'], sd[p + '.norm2.bias'], 1e-5))


def num_layers_of(sd):
    n = 0
    while ('transformer.layers.%d.linear1.weight' % n) in sd:
        n += 1
    return n


def model_forward(sd, x_raw, training=False, shift_r=0, running_out=None, layer_masks=None, dropout_p=0.0):
    """x_raw: (B, 8*T, 8) f32 -> pred (B,T,num_outs)[, aux (B,T,num_aux)].  architecture.py:61-84.
    NOTE mutates x_raw when training and shift_r>0, as the reference does."""
    if training:
        shift_left_(x_raw, shift_r)
    x = x_raw.transpose(1, 2)
    for i in range(3):
        x = resblock(x, sd, 'conv_blocks.%d' % i, 2, training, running_out)
    x = x.transpose(1, 2)
    x = _st(F.linear(x, _st(sd['w_raw_in.weight']), sd['w_raw_in.bias']))
    for l in range(num_layers_of(sd)):
        x = encoder_layer(x, sd, 'transformer.layers.%d' % l,
                          None if layer_masks is None else layer_masks[l], dropout_p)
    pred = F.linear(x, _st(sd['w_out.weight']), sd['w_out.bias'])
    if 'w_aux.weight' in sd:
        return pred, F.linear(x, _st(sd['w_aux.weight']), sd['w_aux.bias'])
    return pred


def init_state_dict(d_model=768, num_layers=6, num_outs=80, num_aux=48, nhead=8, ff=3072, seed=0):
    """Random-init reference-format state_dict (shapes per SURVEY 8b; init distributions follow
    torch defaults / transformer.py:75-78,158-160; all encoder layers start identical as
    nn.TransformerEncoder deep-copies one layer, architecture.py:53-54)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    cin = 8
    for i in range(3):
        p = 'conv_blocks.%d' % i
        sd[p + '.conv1.weight'] = uni((d_model, cin, 3), cin * 3); sd[p + '.conv1.bias'] = uni((d_model,), cin * 3)
        sd[p + '.conv2.weight'] = uni((d_model, d_model, 3), d_model * 3); sd[p + '.conv2.bias'] = uni((d_model,), d_model * 3)
        sd[p + '.residual_path.weight'] = uni((d_model, cin, 1), cin); sd[p + '.residual_path.bias'] = uni((d_model,), cin)
        for bn in ('bn1', 'bn2', 'res_norm'):
            sd[p + '.%s.weight' % bn] = torch.ones(d_model); sd[p + '.%s.bias' % bn] = torch.zeros(d_model)
            sd[p + '.%s.running_mean' % bn] = torch.zeros(d_model); sd[p + '.%s.running_var' % bn] = torch.ones(d_model)
            sd[p + '.%s.num_batches_tracked' % bn] = torch.zeros((), dtype=torch.long)
        cin = d_model
    sd['w_raw_in.weight'] = uni((d_model, d_model), d_model); sd['w_raw_in.bias'] = uni((d_model,), d_model)
    dh = d_model // nhead
    layer = {}
    std_qkv = math.sqrt(2.0 / (d_model * dh + nhead * dh))   # xavier_normal on (H,d,dh): fan_in=d*dh, fan_out=H*dh
    std_o = math.sqrt(2.0 / (dh * d_model + nhead * d_model))
    for n_ in ('w_q', 'w_k', 'w_v'):
        layer['self_attn.' + n_] = torch.randn((nhead, d_model, dh), generator=g) * std_qkv
    layer['self_attn.w_o'] = torch.randn((nhead, dh, d_model), generator=g) * std_o
    layer['self_attn.relative_positional.embeddings'] = torch.randn((nhead, 2 * MAX_REL - 1, dh, 1), generator=g) * dh ** -0.5
    layer['linear1.weight'] = uni((ff, d_model), d_model); layer['linear1.bias'] = uni((ff,), d_model)
    layer['linear2.weight'] = uni((d_model, ff), ff); layer['linear2.bias'] = uni((d_model,), ff)
    for n_ in ('norm1', 'norm2'):
        layer[n_ + '.weight'] = torch.ones(d_model); layer[n_ + '.bias'] = torch.zeros(d_model)
    for l in range(num_layers):
        for k_, v_ in layer.items():
            sd['transformer.layers.%d.%s' % (l, k_)] = v_.clone()
    sd['w_out.weight'] = uni((num_outs, d_model), d_model); sd['w_out.bias'] = uni((num_outs,), d_model)
    if num_aux:
        sd['w_aux.weight'] = uni((num_aux, d_model), d_model); sd['w_aux.bias'] = uni((num_aux,), d_model)
    return sd
