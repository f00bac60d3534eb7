"""Execution plan of the transduction model on the MI355X: which C-ABI kernel runs when, on which
buffers.  Pure orchestration -- every FLOP happens in libsilent_speech_hip.so (csrc/*.hip); torch is
used for device memory, streams and autograd bookkeeping only.

Forward follows reference architecture.py:61-84 (Model.forward), :29-40 (ResBlock.forward) and
transformer.py:43-60,87-112; backward is the hand-derived reverse pass that
loss.backward() (transduction_model.py:209) triggers in the reference through autograd.

Canonical activation layout: (B, T, C) row-major == a flat [B*T][C] matrix (the reference's
(B,C,T) / (T,B,C) transposes, architecture.py:70,72,77,79, are layout-only).  Convolution inputs live
in (B, T+2, C) buffers with a zero halo row at both ends of every sequence, so a k=3 window is one
contiguous 3C-wide row and conv == GEMM with an overlapping-row RowMap (csrc/gemm.hip).
"""
import math
import os

import torch

from . import _lib, ops
from ._lib import OP_KC, OP_OC

RM = ops.rowmap


def _round_up(x, m):
    return (x + m - 1) // m * m


def _split_k(M, N, K):
    """Split-K factor of a weight-gradient GEMM: the largest one that keeps tiles x split within ~400 work items, i.e. inside
    ONE round of the 512 persistent workgroup slots (2 per CU) with room left for the main-stream kernels these GEMMs run
    beside.  Measured alone (tools/dw_split_probe.py, K = 22 016): 144 tiles x 3 = 432 items 201 us, x 4 = 576 items 292 us
    (a second, nearly empty round), x 7 = 1008 items 206 us (two full rounds, twice the f32 atomic epilogues); 36 tiles:
    x 14 -> 72 us, x 28 -> 103 us.  Measured in the step: budgets of 340-420 items give 20.8-21.0 ms, 512 gives 21.4, 1024 gave 22.6."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    overlapped = SIDE_STREAM_ENABLED and os.environ.get('SS_AMD_SIDE_STREAM', '1') != '0'
    s = max(1, min(64, (400 if overlapped else 512) // max(tiles, 1)))     # alone on the GPU a full round (<= 512 items) is best
    return max(1, min(s, K // 512 if K >= 512 else 1))


class Prepared(object):
    """GEMM-ready copies of the parameters in the compute dtype (bf16 or f32): layout changes for the conv / attention
    tensors, casts and transposed copies (so that dX = dY.W is K-contiguous as well) for the nn.Linear ones.
    The buffers are persistent; after every optimiser step they are refreshed by TWO batched launches
    (ss_permute3d_batch): stage 1 reads the parameter arena, stage 2 derives transposes from stage-1 outputs."""

    def __init__(self, model):
        self.build(model)

    @staticmethod
    def layout_signature(model):
        return (tuple(p.data_ptr() for p in model.parameters()), model.compute_dtype, str(model.w_out.weight.device))

    @staticmethod
    def version_signature(model):
        return (model._weights_version, tuple(p._version for p in model.parameters()))

    def build(self, model):
        dt, dev = model.compute_dtype, model.w_out.weight.device
        d = model.d_model
        b1, b2 = ops.PermuteBatch(), ops.PermuteBatch()
        self.b1, self.b2, self.dev = b1, b2, dev

        def new(*shape, dtype=dt, zero=False):
            return (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=dev)

        def cast(w):                 # plain [N][K] weight in the compute dtype
            w = w.detach()
            if dt == torch.float32:
                return w
            return b1.add(w, new(*w.shape), (1, w.shape[0], w.shape[1]), (0, w.shape[1], 1))

        def transposed(src, n, k):   # [n][k] -> [k][n]
            return b2.add(src, new(k, n), (k, 1, n), (1, 0, k))

        self.blocks = []
        for blk in model.conv_blocks:
            O, I, _ = blk.conv1.weight.shape
            e = {}
            w1, w2 = blk.conv1.weight.detach(), blk.conv2.weight.detach()
            e['w1f'] = b1.add(w1, new(O, 3 * I), (O, 3, I), (3 * I, 1, 3))
            e['w2f'] = b1.add(w2, new(O, 3 * O), (O, 3, O), (3 * O, 1, 3))
            e['wr'] = b1.add(blk.residual_path.weight.detach(), new(O, I), (O, 1, I), (I, 0, 1))
            e['wrT'] = transposed(e['wr'], O, I)
            # input-gradient forms (flipped taps):  conv2 (stride 1)  Wb[i][j*O+o] = W[o][i][2-j]
            e['w2b'] = b1.add(w2.view(-1)[2:], new(O, 3 * O), (O, 3, O), (3, -1, 3 * O))
            if I % 8 == 0:   # stride-2 conv1: even rows use tap 1, odd rows taps (2, 0)
                e['w1b_even'] = b1.add(w1.view(-1)[1:], new(I, O), (I, 1, O), (3, 0, 3 * I))
                e['w1b_odd'] = b1.add(w1.view(-1)[2:], new(I, 2 * O), (I, 2, O), (3, -2, 3 * I))
            self.blocks.append(e)

        self.w_raw_in = cast(model.w_raw_in.weight)
        self.w_raw_in_T = transposed(self.w_raw_in, d, d)
        H, dh, dp, D = model.n_head, model.d_qkv, model.dp, model.max_rel
        MPt = _round_up(2 * D - 1, 32)
        self.layers = []
        for layer in model.transformer.layers:
            a = layer.self_attn
            e = {}
            wqkv = new(3, H, dp, d)
            for i, w in enumerate((a.w_q, a.w_k, a.w_v)):       # (H, d, dh) -> [h][a (padded)][f]
                b1.add(w.detach(), wqkv[i], (H, dp, d), (d * dh, 1, dh), valid1=dh)
            e['wqkv'] = wqkv.view(3 * H * dp, d)
            e['wqkvT'] = transposed(e['wqkv'], 3 * H * dp, d)
            e['wo'] = b1.add(a.w_o.detach(), new(d, H * dp), (d, H, dp), (1, dh * d, d), valid2=dh)
            e['woT'] = transposed(e['wo'], d, H * dp)
            emb = a.relative_positional.embeddings.detach()    # (H, 2D-1, dh, 1)
            e['E'] = b1.add(emb, new(H, 2 * D - 1, dp), (H, 2 * D - 1, dp), ((2 * D - 1) * dh, dh, 1), valid2=dh)
            e['ET'] = b1.add(emb, new(H, dp, MPt), (H, dp, MPt), ((2 * D - 1) * dh, 1, dh), valid1=dh, valid2=2 * D - 1)
            e['w1'] = cast(layer.linear1.weight)
            e['w2'] = cast(layer.linear2.weight)
            e['w1T'] = transposed(e['w1'], layer.linear1.weight.shape[0], d)
            e['w2T'] = transposed(e['w2'], d, layer.linear2.weight.shape[1])
            self.layers.append(e)
        n_out = model.w_out.weight.shape[0]
        n_aux = model.w_aux.weight.shape[0] if model.has_aux_out else 0
        nh = _round_up(n_out + n_aux, 8)
        wh = new(nh, d, zero=True)
        bh = new(nh, dtype=torch.float32, zero=True)
        b1.add(model.w_out.weight.detach(), wh[:n_out], (1, n_out, d), (0, d, 1))
        b1.add(model.w_out.bias.detach(), bh[:n_out], (1, 1, n_out), (0, 0, 1))
        if n_aux:
            b1.add(model.w_aux.weight.detach(), wh[n_out:n_out + n_aux], (1, n_aux, d), (0, d, 1))
            b1.add(model.w_aux.bias.detach(), bh[n_out:n_out + n_aux], (1, 1, n_aux), (0, 0, 1))
        self.w_head, self.b_head, self.n_head_cols = wh, bh, nh
        self.w_head_T = transposed(wh, nh, d)
        self.layout_sig = self.layout_signature(model)
        self.refresh(model)

    def refresh(self, model):
        self.b1.run(self.dev)
        self.b2.run(self.dev)
        self.version_sig = self.version_signature(model)


def prepared(model):
    pr = getattr(model, '_prepared', None)
    if pr is None or pr.layout_sig != Prepared.layout_signature(model):
        pr = Prepared(model)
        model._prepared = pr
    elif pr.version_sig != Prepared.version_signature(model):
        pr.refresh(model)
    return pr


class Ctx(object):
    pass


def _bn(mod):
    return mod.weight.detach(), mod.bias.detach()


def forward(model, x_raw, training, shift_r, seed):
    """x_raw (B, 8T, 8) f32 on the GPU -> head [B*T][n_head_cols] f32 and the saved context."""
    pr = prepared(model)
    dt, dev = model.compute_dtype, x_raw.device
    B, T0, Cin0 = x_raw.shape
    if T0 % 8 != 0:
        raise ValueError('raw EMG length %d must be a multiple of 8 (three stride-2 convolutions)' % T0)
    d = model.d_model
    ctx = Ctx()
    ctx.B, ctx.T0, ctx.training, ctx.seed = B, T0, training, seed
    p_drop = model.dropout_p if training else 0.0
    ctx.p_drop = p_drop
    bn_reduce = model._bn_reduce_fn if training else None

    xin = torch.empty(B, T0 + 2, Cin0, dtype=dt, device=dev)
    shifted = torch.empty_like(x_raw) if (training and shift_r > 0) else None
    ops.emg_prepare(x_raw, xin, shifted, B, T0, Cin0, shift_r if training else 0)
    if shifted is not None:
        x_raw.copy_(shifted)            # the reference mutates its input in place (architecture.py:67-68)

    ctx.blocks = []
    Tin, Cin = T0, Cin0
    nblk = len(model.conv_blocks)
    for i, (blk, w) in enumerate(zip(model.conv_blocks, pr.blocks)):
        O = blk.conv1.weight.shape[0]
        Tout = Tin // 2
        rows = B * Tout
        s = Ctx()
        s.xin, s.Tin, s.Cin, s.Tout, s.O = xin, Tin, Cin, Tout, O
        scratch = ops.bn_scratch(B, Tout, O, dev)
        s.scratch = scratch
        in_bs = (Tin + 2) * Cin
        c1 = torch.empty(rows, O, dtype=dt, device=dev)
        ops.gemm(xin, w['w1f'], c1, rows, O, 3 * Cin, RM(2 * Cin, Tout, in_bs), RM(3 * Cin), RM(O), bias=blk.conv1.bias.detach())
        cr = torch.empty(rows, O, dtype=dt, device=dev)
        ops.gemm(xin, w['wr'], cr, rows, O, Cin, RM(2 * Cin, Tout, in_bs, base=Cin), RM(Cin), RM(O), bias=blk.residual_path.bias.detach())
        g1, b1 = _bn(blk.bn1)
        sh1 = blk.bn1.running_mean if bn_reduce is not None else None
        m1, i1 = ops.bn_stats(c1, B, Tout, O, 0, scratch, blk.bn1.running_mean, blk.bn1.running_var, training=training, shift=sh1, reduce_fn=bn_reduce)
        h1 = torch.empty(B, Tout + 2, O, dtype=dt, device=dev)
        ops.bn_apply(c1, (m1, i1, g1, b1), 0, h1, 1, B, Tout, O, True)
        c2 = torch.empty(rows, O, dtype=dt, device=dev)
        ops.gemm(h1, w['w2f'], c2, rows, O, 3 * O, RM(O, Tout, (Tout + 2) * O), RM(3 * O), RM(O), bias=blk.conv2.bias.detach())
        g2, b2 = _bn(blk.bn2)
        gr, br = _bn(blk.res_norm)
        sh2 = blk.bn2.running_mean if bn_reduce is not None else None
        shr = blk.res_norm.running_mean if bn_reduce is not None else None
        m2, i2 = ops.bn_stats(c2, B, Tout, O, 0, scratch, blk.bn2.running_mean, blk.bn2.running_var, training=training, shift=sh2, reduce_fn=bn_reduce)
        mr, ir = ops.bn_stats(cr, B, Tout, O, 0, scratch, blk.res_norm.running_mean, blk.res_norm.running_var, training=training, shift=shr, reduce_fn=bn_reduce)
        last = i == nblk - 1
        pad_y = 0 if last else 1
        y = torch.empty(B, Tout + 2 * pad_y, O, dtype=dt, device=dev)
        ops.bn_apply(c2, (m2, i2, g2, b2), 0, y, pad_y, B, Tout, O, True, xb=cr, sb=(mr, ir, gr, br), pad_xb=0)
        if training:
            for bnm in (blk.bn1, blk.bn2, blk.res_norm):
                bnm.num_batches_tracked += 1
        s.c1, s.cr, s.h1, s.c2, s.y, s.pad_y = c1, cr, h1, c2, y, pad_y
        s.st1, s.st2, s.str_ = (m1, i1, g1), (m2, i2, g2), (mr, ir, gr)
        ctx.blocks.append(s)
        xin, Tin, Cin = y, Tout, O

    T = Tin
    M = B * T
    ctx.T, ctx.M = T, M
    conv_out = xin.view(M, d)
    ctx.conv_out = conv_out
    x = torch.empty(M, d, dtype=dt, device=dev)
    ops.gemm(conv_out, pr.w_raw_in, x, M, d, d, RM(d), RM(d), RM(d), bias=model.w_raw_in.bias.detach())

    H, dp, D = model.n_head, model.dp, model.max_rel
    Tp = _round_up(T, 8)
    scale = 1.0 / math.sqrt(model.d_qkv)
    ctx.Tp, ctx.scale = Tp, scale
    ctx.need_T = bool(_lib.lib().ss_relpos_attention_needs_transposed(_lib.dtype_code(dt), T, dp, D))
    ctx.layers = []
    for l, (layer, w) in enumerate(zip(model.transformer.layers, pr.layers)):
        s = Ctx()
        s.x = x
        qkv = torch.empty(M, 3 * H * dp, dtype=dt, device=dev)
        if ctx.need_T:      # the per-tile attention kernels read a per-sequence transposed copy, written by the same GEMM epilogue
            qkvT = torch.empty(B, 3 * H * dp, Tp, dtype=dt, device=dev)
            ops.gemm_ex(x, w['wqkv'], qkv, M, 3 * H * dp, d, RM(d), RM(d), RM(3 * H * dp),
                        c2=qkvT, cmap2=RM(1, T, 3 * H * dp * Tp), col_stride2=Tp)
        else:               # LDS-resident attention (bf16 rows of <= 208 frames): row-major operands only
            qkvT = None
            ops.gemm(x, w['wqkv'], qkv, M, 3 * H * dp, d, RM(d), RM(d), RM(3 * H * dp))
        o = torch.empty(M, H * dp, dtype=dt, device=dev)
        lse = torch.empty(B, H, T, dtype=torch.float32, device=dev)
        ops.relpos_attention_forward(qkv, qkvT, w['E'], o, lse, B, H, T, Tp, dp, D, scale, p=p_drop, seed=seed, rng_stream=4 * l)
        a = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(o, w['wo'], a, M, d, H * dp, RM(H * dp), RM(H * dp), RM(d))
        y1 = torch.empty(M, d, dtype=dt, device=dev)
        mean1, rstd1 = ops.add_dropout_layernorm(x, a, layer.norm1.weight.detach(), layer.norm1.bias.detach(), y1, M, d,
                                                 eps=layer.norm1.eps, p=p_drop, seed=seed, rng_stream=4 * l + 1)
        ff = layer.linear1.weight.shape[0]
        hid = torch.empty(M, ff, dtype=dt, device=dev)
        ops.gemm(y1, w['w1'], hid, M, ff, d, RM(d), RM(d), RM(ff), bias=layer.linear1.bias.detach(), relu=True,
                 dropout_p=p_drop, seed=seed, rng_stream=4 * l + 2)
        f = torch.empty(M, d, dtype=dt, device=dev)
        ops.gemm(hid, w['w2'], f, M, d, ff, RM(ff), RM(ff), RM(d), bias=layer.linear2.bias.detach())
        y2 = torch.empty(M, d, dtype=dt, device=dev)
        mean2, rstd2 = ops.add_dropout_layernorm(y1, f, layer.norm2.weight.detach(), layer.norm2.bias.detach(), y2, M, d,
                                                 eps=layer.norm2.eps, p=p_drop, seed=seed, rng_stream=4 * l + 3)
        s.qkv, s.qkvT, s.o, s.lse, s.z1, s.mean1, s.rstd1, s.y1 = qkv, qkvT, o, lse, a, mean1, rstd1, y1
        s.hid, s.z2, s.mean2, s.rstd2 = hid, f, mean2, rstd2
        ctx.layers.append(s)
        x = y2
    ctx.x_final = x
    nh = pr.n_head_cols
    head = torch.empty(M, nh, dtype=torch.float32, device=dev)
    ops.gemm(x, pr.w_head, head, M, nh, d, RM(d), RM(d), RM(nh), bias=pr.b_head)
    if not training:
        ctx = None
    return head, ctx


def _grad(p):
    """The (flat-arena backed) gradient buffer of a parameter; accumulated into, as autograd would."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class GradUnpack(object):
    """f32 staging buffers for the weight gradients whose GEMM layout differs from the parameter layout (conv (O,I,k),
    per-head attention projections, fused heads) + ONE batched launch that accumulates them into the .grad arena."""

    def __init__(self, model, dev):
        d, H, dh, dp = model.d_model, model.n_head, model.d_qkv, model.dp
        pr = prepared(model)
        nh = pr.n_head_cols
        sizes = [('head_w', nh * d), ('head_b', nh)]
        for l in range(len(model.transformer.layers)):
            sizes += [('wo%d' % l, d * H * dp), ('wqkv%d' % l, 3 * H * dp * d)]
        for i, blk in enumerate(model.conv_blocks):
            O, I, _ = blk.conv1.weight.shape
            sizes += [('c2_%d' % i, O * 3 * O), ('c1_%d' % i, O * 3 * I)]
        total = sum((n + 3) // 4 * 4 for _, n in sizes)
        self.arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buf, off = {}, 0
        for k, n in sizes:
            self.buf[k] = self.arena[off:off + n]
            off += (n + 3) // 4 * 4
        # One batched launch per group, issued on the side stream as soon as the group's dW GEMMs are queued: the encoder
        # group overlaps the conv backward; only conv block 0 (the last to finish) is left for the end of the step.
        ub = ops.PermuteBatch()
        n_out = model.w_out.weight.shape[0]
        n_aux = model.w_aux.weight.shape[0] if model.has_aux_out else 0
        hw, hb = self.buf['head_w'].view(nh, d), self.buf['head_b']
        ub.add(hw[:n_out], model.w_out.weight.grad, (1, n_out, d), (0, d, 1), accumulate=True)
        ub.add(hb[:n_out], model.w_out.bias.grad, (1, 1, n_out), (0, 0, 1), accumulate=True)
        if n_aux:
            ub.add(hw[n_out:n_out + n_aux], model.w_aux.weight.grad, (1, n_aux, d), (0, d, 1), accumulate=True)
            ub.add(hb[n_out:n_out + n_aux], model.w_aux.bias.grad, (1, 1, n_aux), (0, 0, 1), accumulate=True)
        for l, layer in enumerate(model.transformer.layers):
            a = layer.self_attn
            ub.add(self.buf['wo%d' % l], a.w_o.grad, (H, dh, d), (dp, 1, H * dp), accumulate=True)
            for i, wp in enumerate((a.w_q, a.w_k, a.w_v)):
                ub.add(self.buf['wqkv%d' % l][i * H * dp * d:], wp.grad, (H, d, dh), (dp * d, 1, d), accumulate=True)
        self.encoder_batch = ub
        self.conv_batches = []
        for i, blk in enumerate(model.conv_blocks):
            O, I, _ = blk.conv1.weight.shape
            cb = ops.PermuteBatch()
            cb.add(self.buf['c2_%d' % i], blk.conv2.weight.grad, (O, O, 3), (3 * O, 1, O), accumulate=True)
            cb.add(self.buf['c1_%d' % i], blk.conv1.weight.grad, (O, I, 3), (3 * I, 1, I), accumulate=True)
            self.conv_batches.append(cb)
        self.sig = self.signature(model)

    @staticmethod
    def signature(model):
        return (model._gflat.data_ptr() if getattr(model, '_gflat', None) is not None else 0,
                tuple(p.grad.data_ptr() for p in model.optimized_parameters()), str(model.w_out.weight.device))


def grad_unpack(model, dev):
    gu = getattr(model, '_grad_unpack', None)
    if gu is None or gu.sig != GradUnpack.signature(model):
        gu = GradUnpack(model, dev)
        model._grad_unpack = gu
    return gu


SIDE_STREAM_ENABLED = True      # bench.py clears this on its event-timed steps so that per-launch durations are exclusive


class _SideStream(object):
    """Weight-gradient GEMMs (dW = dY^T X), bias column sums and gradient re-layouts do not feed the backward chain,
    so they run on a second HIP stream: their workgroups fill the CUs that the dependent chain (dX GEMMs, attention,
    norm kernels) leaves idle in its tail rounds (e.g. a 22 k x 768 GEMM is 1032 tiles = 2.02 rounds of 512 slots).
    Inputs are kept alive (and never written again on the main stream) until join()."""

    def __init__(self, model, dev):
        self.enabled = dev.type == 'cuda' and SIDE_STREAM_ENABLED and os.environ.get('SS_AMD_SIDE_STREAM', '1') != '0'
        self.keep = []
        self.blocks_per_cu = int(os.environ.get('SS_AMD_SIDE_BLOCKS', '2'))
        if self.enabled:
            st = getattr(model, '_side_stream', None)
            if st is None or st.device != dev:
                st = torch.cuda.Stream(device=dev)
                model._side_stream = st
            self.stream = st

    def run(self, fn, *keep):
        if not self.enabled:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        from . import _lib
        old = _lib.lib().ss_gemm_set_blocks_per_cu(self.blocks_per_cu)     # leave room on every CU for the main-stream kernels
        try:
            with torch.cuda.stream(self.stream):
                fn()
        finally:
            _lib.lib().ss_gemm_set_blocks_per_cu(old)
        self.keep.extend(keep)

    def join(self):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
        self.keep = []


def _dw_direct(dy, x, grad, N, K, rows, amap, bmap):
    """grad[N][K] += dy^T x  (both operands outer-contiguous, split-K with f32 atomics)."""
    ops.gemm(dy, x, grad, N, K, rows, amap, bmap, RM(K), a_mode=OP_OC, b_mode=OP_OC, mode=2, split_k=_split_k(N, K, rows))


class _DwGroup(object):
    """Weight-gradient GEMMs that become ready at about the same time (the four of an encoder layer, the three of a ResBlock)
    are launched as ONE grouped kernel (ss_gemm_dw_grouped): the K split that fills the 256 CUs is chosen for the group, which
    divides the number of f32 atomic accumulations by ~3.5 and lets every workgroup own a 256 x 256 tile.  bf16 only; the exact
    f32 mode keeps the per-GEMM kernels."""

    def __init__(self, grouped):
        self.grouped, self.jobs = grouped, []

    def add(self, dy, x, grad, N, K, rows, amap, bmap):
        if self.grouped:
            self.jobs.append((dy, x, grad, N, K, rows, amap, bmap, K))
        else:
            _dw_direct(dy, x, grad, N, K, rows, amap, bmap)

    def launch(self):
        if self.jobs:
            ops.gemm_dw_grouped(self.jobs)
            self.jobs = []


def backward(model, ctx, dhead):
    """Accumulates into .grad of every parameter (except the relative-position embeddings, which get
    no gradient in the reference either, transformer.py:214-218).  dhead: [M][n_head_cols] f32."""
    pr = prepared(model)
    dt, dev = model.compute_dtype, dhead.device
    B, T, M, d = ctx.B, ctx.T, ctx.M, model.d_model
    H, dp, D, dh = model.n_head, model.dp, model.max_rel, model.d_qkv
    Tp, p_drop, seed = ctx.Tp, ctx.p_drop, ctx.seed
    keep_scale = 1.0 / (1.0 - p_drop)
    bn_reduce = model._bn_reduce_fn
    nh = pr.n_head_cols

    # ---- heads (architecture.py:82)
    dh_t = dhead if dt == torch.float32 else torch.empty(M, nh, dtype=dt, device=dev)
    if dt != torch.float32:
        ops.cast_f32(dhead, dh_t, M * nh)
    n_out = model.w_out.weight.shape[0]
    n_aux = model.w_aux.weight.shape[0] if model.has_aux_out else 0
    side = _SideStream(model, dev)
    for p_ in model.optimized_parameters():
        _grad(p_)
    gu = grad_unpack(model, dev)
    gu.arena.zero_()                 # staging buffers of the re-laid-out weight gradients (one memset, main stream)

    grouped = dt == torch.bfloat16 and os.environ.get('SS_AMD_DW_GROUPED', '1') != '0'
    grp = _DwGroup(grouped)                      # head + last encoder layer travel together
    grp.add(dh_t, ctx.x_final, gu.buf['head_w'], nh, d, M, RM(nh), RM(d))
    side.run(lambda: ops.colsum(dh_t, M, nh, nh, gu.buf['head_b']), dh_t, dhead)
    G = torch.empty(M, d, dtype=dt, device=dev)
    ops.gemm(dh_t, pr.w_head_T, G, M, d, nh, RM(nh), RM(nh), RM(d))

    keep_last = ()
    # ---- encoder layers, last to first (transformer.py:54-59)
    for l in range(len(ctx.layers) - 1, -1, -1):
        layer, w, s = model.transformer.layers[l], pr.layers[l], ctx.layers[l]
        a = layer.self_attn
        ff = layer.linear1.weight.shape[0]
        dF = torch.empty(M, d, dtype=dt, device=dev)
        ops.layernorm_backward(G, s.z2, s.mean2, s.rstd2, layer.norm2.weight.detach(), G, dF, _grad(layer.norm2.weight), _grad(layer.norm2.bias),
                               M, d, p=p_drop, seed=seed, rng_stream=4 * l + 3)
        grp.add(dF, s.hid, layer.linear2.weight.grad, d, ff, M, RM(d), RM(ff))
        side.run(lambda dF=dF, layer=layer: ops.colsum(dF, M, d, d, layer.linear2.bias.grad), dF)
        dHid = torch.empty(M, ff, dtype=dt, device=dev)
        ops.gemm(dF, w['w2T'], dHid, M, ff, d, RM(d), RM(d), RM(ff), gate=s.hid, gate_scale=keep_scale)

        grp.add(dHid, s.y1, layer.linear1.weight.grad, ff, d, M, RM(ff), RM(d))
        side.run(lambda dHid=dHid, layer=layer: ops.colsum(dHid, M, ff, ff, layer.linear1.bias.grad), dHid)
        ops.gemm(dHid, w['w1T'], G, M, d, ff, RM(ff), RM(ff), RM(d), mode=1)
        dA = torch.empty(M, d, dtype=dt, device=dev)
        ops.layernorm_backward(G, s.z1, s.mean1, s.rstd1, layer.norm1.weight.detach(), G, dA, _grad(layer.norm1.weight), _grad(layer.norm1.bias),
                               M, d, p=p_drop, seed=seed, rng_stream=4 * l + 1)
        # output projection  out[t,b,f] = sum_{h,a} o[b,h,t,a] w_o[h,a,f]   (transformer.py:111)
        grp.add(dA, s.o, gu.buf['wo%d' % l], d, H * dp, M, RM(d), RM(H * dp))
        dO = torch.empty(M, H * dp, dtype=dt, device=dev)
        if ctx.need_T:
            dOT = torch.empty(B, H * dp, Tp, dtype=dt, device=dev)
            ops.gemm_ex(dA, w['woT'], dO, M, H * dp, d, RM(d), RM(d), RM(H * dp),
                        c2=dOT, cmap2=RM(1, T, H * dp * Tp), col_stride2=Tp)
        else:
            dOT = None
            ops.gemm(dA, w['woT'], dO, M, H * dp, d, RM(d), RM(d), RM(H * dp))
        dqkv = torch.empty(M, 3 * H * dp, dtype=dt, device=dev)
        dsc = torch.empty(B, H, T, dtype=torch.float32, device=dev)
        ops.relpos_attention_backward(s.qkv, s.qkvT, w['E'], w['ET'], s.o, s.lse, dO, dOT, dsc, dqkv, B, H, T, Tp, dp, D, ctx.scale,
                                      p=p_drop, seed=seed, rng_stream=4 * l)
        grp.add(dqkv, s.x, gu.buf['wqkv%d' % l], 3 * H * dp, d, M, RM(3 * H * dp), RM(d))
        ops.gemm(dqkv, w['wqkvT'], G, M, d, 3 * H * dp, RM(3 * H * dp), RM(3 * H * dp), RM(d), mode=1)
        if l > 0:                               # layer 0's group waits for w_raw_in's gradient
            side.run(grp.launch, dF, dHid, dA, dqkv, dh_t)
            grp = _DwGroup(grouped)
        keep_last = (dF, dHid, dA, dqkv)
        del dqkv, dO, dOT, dA, dF, dHid

    # ---- w_raw_in (architecture.py:73)
    grp.add(G, ctx.conv_out, model.w_raw_in.weight.grad, d, d, M, RM(d), RM(d))
    side.run(grp.launch, G, dh_t, *keep_last)
    del keep_last
    side.run(lambda G=G: ops.colsum(G, M, d, d, model.w_raw_in.bias.grad), G)
    side.run(lambda: gu.encoder_batch.run(dev))                # heads + encoder layers: re-laid-out gradients -> .grad arena, under the conv backward
    dy = torch.empty(M, d, dtype=dt, device=dev)
    ops.gemm(G, pr.w_raw_in_T, dy, M, d, d, RM(d), RM(d), RM(d))
    del G

    # ---- ResBlocks, last to first (architecture.py:29-40)
    for i in range(len(ctx.blocks) - 1, -1, -1):
        blk, w, s = model.conv_blocks[i], pr.blocks[i], ctx.blocks[i]
        O, Cin, Tin, Tout = s.O, s.Cin, s.Tin, s.Tout
        rows = B * Tout
        pbs = (Tout + 2) * O                    # batch stride of a padded (B, Tout+2, O) buffer
        dc2 = torch.empty(B, Tout + 2, O, dtype=dt, device=dev)
        dcr = torch.empty(rows, O, dtype=dt, device=dev)
        ops.bn_backward(dy, 0, s.y, s.pad_y, s.c2, 0, s.st2, dc2, 1, _grad(blk.bn2.weight), _grad(blk.bn2.bias), s.scratch, B, Tout, O, True,
                        xb=s.cr, pad_xb=0, sb=s.str_, dxb=dcr, pad_dxb=0, dgamma_b=_grad(blk.res_norm.weight), dbeta_b=_grad(blk.res_norm.bias),
                        reduce_fn=bn_reduce)
        # conv2 (k3, stride 1): weight, bias, input gradients
        cgrp = _DwGroup(grouped)
        cgrp.add(dc2, s.h1, gu.buf['c2_%d' % i], O, 3 * O, rows, RM(O, Tout, pbs, base=O), RM(O, Tout, pbs))
        # d/d(bias) of a conv feeding training-mode BatchNorm is identically 0 (BN removes the mean): nothing to add
        dh1 = torch.empty(rows, O, dtype=dt, device=dev)
        ops.gemm(dc2, w['w2b'], dh1, rows, O, 3 * O, RM(O, Tout, pbs), RM(3 * O), RM(O))
        dc1 = torch.empty(B, Tout + 2, O, dtype=dt, device=dev)
        ops.bn_backward(dh1, 0, s.h1, 1, s.c1, 0, s.st1, dc1, 1, _grad(blk.bn1.weight), _grad(blk.bn1.bias), s.scratch, B, Tout, O, True,
                        reduce_fn=bn_reduce)
        del dh1
        # conv1 (k3, stride 2) and the 1x1 stride-2 residual path
        in_bs = (Tin + 2) * Cin
        cgrp.add(dc1, s.xin, gu.buf['c1_%d' % i], O, 3 * Cin, rows, RM(O, Tout, pbs, base=O), RM(2 * Cin, Tout, in_bs))
        cgrp.add(dcr, s.xin, blk.residual_path.weight.grad, O, Cin, rows, RM(O, Tout, Tout * O), RM(2 * Cin, Tout, in_bs, base=Cin))
        side.run(cgrp.launch, dc2, dc1, dcr)
        del dc2
        side.run(lambda i=i: gu.conv_batches[i].run(dev))          # this block's conv gradients -> parameter layout
        if i > 0:
            dx = torch.empty(B * Tin, Cin, dtype=dt, device=dev)
            out_even = RM(2 * Cin, Tout, Tin * Cin)
            out_odd = RM(2 * Cin, Tout, Tin * Cin, base=Cin)
            ops.gemm(dc1, w['w1b_even'], dx, rows, Cin, O, RM(O, Tout, pbs, base=O), RM(O), out_even)
            ops.gemm(dcr, w['wrT'], dx, rows, Cin, O, RM(O), RM(O), out_even, mode=1)
            ops.gemm(dc1, w['w1b_odd'], dx, rows, Cin, 2 * O, RM(O, Tout, pbs, base=O), RM(2 * O), out_odd)
            dy = dx
        del dc1, dcr
    side.join()
