"""Host side of the transduction model's execution plan on the MI355X.  The plan itself -- which kernel runs
when, on which buffers, for the whole forward pass (reference architecture.py:61-84, :29-40, transformer.py:43-60,87-112)
and the hand-derived backward pass (what loss.backward(), transduction_model.py:209, triggers through autograd) -- is
native code behind TWO C entry points (csrc/plan.hip: ss_plan_forward / ss_plan_backward), so a training step costs two
host calls instead of ~450 ctypes round trips.  This module keeps what is host bookkeeping by nature: the GEMM-ready
copies of the parameters (Prepared), the staging buffers / job tables of the gradient un-layout (GradUnpack), binding
their device pointers to the plan's named slots, and the per-call workspace.  torch is used for device memory, streams
and autograd bookkeeping only; every FLOP happens in libsilent_speech_hip.so.

Canonical activation layout: (B, T, C) row-major == a flat [B*T][C] matrix (the reference's
(B,C,T) / (T,B,C) transposes, architecture.py:70,72,77,79, are layout-only).  Convolution inputs live
in (B, T+2, C) buffers with a zero halo row at both ends of every sequence, so a k=3 window is one
contiguous 3C-wide row and conv == GEMM with an overlapping-row RowMap (csrc/gemm.hip).
"""
import ctypes
import math
import os

import torch

from . import _lib, ops
from ._lib import OP_KC, OP_OC

RM = ops.rowmap


def _round_up(x, m):
    return (x + m - 1) // m * m


class Prepared(object):
    """GEMM-ready copies of the parameters in the compute dtype (bf16 or f32): layout changes for the conv / attention
    tensors, casts and transposed copies (so that dX = dY.W is K-contiguous as well) for the nn.Linear ones.
    The buffers are persistent; after every optimiser step they are refreshed by TWO batched launches
    (ss_permute3d_batch): stage 1 reads the parameter arena, stage 2 derives transposes from stage-1 outputs."""

    def __init__(self, model):
        self.build(model)

    @staticmethod
    def layout_signature(model):
        return (tuple(p.data_ptr() for p in model.all_parameters()), model.compute_dtype, str(model.w_out.weight.device))

    @staticmethod
    def version_signature(model):
        return (model._weights_version, tuple(p._version for p in model.all_parameters()))

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
            # transposed-score attention kernels (bf16 rows of <= 224 frames): E / scale in MFMA-fragment order (ss_relpos_attention_prepare_tables);
            # the embeddings are never trained (transformer.py:214-218), so the table is rebuilt only when their version counter moves
            # f32 plans: the [hi | lo] tables of the plane kernels (the parity-grade mode, f32_matmul='bf16x3'; unused by the exact-f32 mode)
            nb = int(_lib.lib().ss_relpos_attention_table_bytes(H, dp, D)) if dt == torch.bfloat16 else int(_lib.lib().ss_relpos_attention_x3_table_bytes(H, dp, D))
            e['EF.x3'] = dt != torch.bfloat16 and nb > 0
            e['EF'] = new(max(nb // 2, 8), dtype=torch.bfloat16)
            e['EF.src'], e['EF.version'], e['EF.scale'] = a.relative_positional.embeddings, None, 1.0 / math.sqrt(dh)
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
        for e in self.layers:
            src = e['EF.src']
            if e['EF'].numel() > 8 and e['EF.version'] != (src._version, src.data_ptr()):
                (ops.relpos_attention_x3_tables if e['EF.x3'] else ops.relpos_attention_tables)(src, model.dp, e['EF.scale'], out=e['EF'])
                e['EF.version'] = (src._version, src.data_ptr())
        self.version_sig = self.version_signature(model)


def prepared(model):
    pr = getattr(model, '_prepared', None)
    if pr is None or pr.layout_sig != Prepared.layout_signature(model):
        pr = Prepared(model)
        model._prepared = pr
    elif pr.version_sig != Prepared.version_signature(model):
        pr.refresh(model)
    return pr


class ModelDims(ctypes.Structure):
    _fields_ = [('d_model', ctypes.c_int32), ('n_layers', ctypes.c_int32), ('n_head', ctypes.c_int32), ('d_qkv', ctypes.c_int32),
                ('dp', ctypes.c_int32), ('max_rel', ctypes.c_int32), ('ff', ctypes.c_int32), ('n_head_cols', ctypes.c_int32),
                ('dtype', ctypes.c_int32), ('ln_eps', ctypes.c_float)]


class GradUnpack(object):
    """f32 staging buffers for the weight gradients whose GEMM layout differs from the parameter layout (conv (O,I,k),
    per-head attention projections, fused heads) + the batched launches that accumulate them into the .grad arena."""

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
        self.head_batch = ub
        ua = ops.PermuteBatch()                     # heads + every layer in one table: the launch of a run without per-layer listeners
        ua.add(hw[:n_out], model.w_out.weight.grad, (1, n_out, d), (0, d, 1), accumulate=True)
        ua.add(hb[:n_out], model.w_out.bias.grad, (1, 1, n_out), (0, 0, 1), accumulate=True)
        if n_aux:
            ua.add(hw[n_out:n_out + n_aux], model.w_aux.weight.grad, (1, n_aux, d), (0, d, 1), accumulate=True)
            ua.add(hb[n_out:n_out + n_aux], model.w_aux.bias.grad, (1, 1, n_aux), (0, 0, 1), accumulate=True)
        self.all_batch = ua
        # one batch per encoder layer, launched right behind the layer's weight-gradient GEMMs: the layer's gradients are then final
        # (and, under data parallelism, on their way through the all-reduce) while the backward of the earlier layers still runs
        self.layer_batches = []
        for l, layer in enumerate(model.transformer.layers):
            a = layer.self_attn
            lb = ops.PermuteBatch()
            lb.add(self.buf['wo%d' % l], a.w_o.grad, (H, dh, d), (dp, 1, H * dp), accumulate=True)
            for i, wp in enumerate((a.w_q, a.w_k, a.w_v)):
                lb.add(self.buf['wqkv%d' % l][i * H * dp * d:], wp.grad, (H, d, dh), (dp * d, 1, d), accumulate=True)
            self.layer_batches.append(lb)
            ua.add(self.buf['wo%d' % l], a.w_o.grad, (H, dh, d), (dp, 1, H * dp), accumulate=True)
            for i, wp in enumerate((a.w_q, a.w_k, a.w_v)):
                ua.add(self.buf['wqkv%d' % l][i * H * dp * d:], wp.grad, (H, d, dh), (dp * d, 1, d), accumulate=True)
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


SIDE_STREAM_ENABLED = True      # tools clear this to measure the serial schedule


def _split_k(M, N, K):
    """Split-K factor of a weight-gradient GEMM on the 128-wide transposing-read kernel (exact-f32 mode, tests): the largest one
    that keeps tiles x split within one round of the 512 persistent workgroup slots (same rule as csrc/plan.hip)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    overlapped = SIDE_STREAM_ENABLED and os.environ.get('SS_AMD_SIDE_STREAM', '1') != '0'
    s = max(1, min(64, (400 if overlapped else 512) // max(tiles, 1)))
    return max(1, min(s, K // 512 if K >= 512 else 1))


_REDUCE_HOOK = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p)
_EVENT_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)


class PlanBinding(object):
    """The native plan (ss_plan) of one model + the binding of its named pointer slots."""

    def __init__(self, model):
        L = _lib.lib()
        pr = prepared(model)
        layer0 = model.transformer.layers[0] if len(model.transformer.layers) else None
        dims = ModelDims(model.d_model, len(model.transformer.layers), model.n_head, model.d_qkv, model.dp, model.max_rel,
                         layer0.linear1.weight.shape[0] if layer0 is not None else 0, pr.n_head_cols, _lib.dtype_code(model.compute_dtype),
                         float(layer0.norm1.eps) if layer0 is not None else 1e-5)
        self.handle = L.ss_plan_create(ctypes.byref(dims))
        if not self.handle:
            raise RuntimeError('ss_plan_create failed: %s' % L.ss_last_error().decode())
        self.lib = L
        self.names = [L.ss_plan_slot_name(self.handle, i).decode() for i in range(L.ss_plan_slot_count(self.handle))]
        self.sig = None
        self.layout = (model.d_model, len(model.transformer.layers), model.compute_dtype)
        self.ctx_bytes = int(L.ss_plan_ctx_bytes())
        self.keep = None
        self.ws = None                               # workspace of the call in flight (the data-parallel hook maps pointers into it)
        self._hook = self._hook_fn = None
        self._event = self._event_fn = None
        self.callback_error = None

    def __del__(self):
        try:
            if self.handle:
                self.lib.ss_plan_destroy(self.handle)
        except Exception:
            pass

    # ---- slot table: name -> tensor (device pointer) or int
    @staticmethod
    def _table(model, pr, gu, dev):
        t = {}

        def bn(prefix, m, grads):
            t[prefix + '.weight'], t[prefix + '.bias'] = m.weight, m.bias
            t[prefix + '.running_mean'], t[prefix + '.running_var'], t[prefix + '.num_batches_tracked'] = m.running_mean, m.running_var, m.num_batches_tracked
            if grads:
                t[prefix + '.weight.grad'], t[prefix + '.bias.grad'] = m.weight.grad, m.bias.grad

        def batch(prefix, pb):
            jobs, jb, total = pb.device_tables(dev)
            t[prefix + '.jobs'], t[prefix + '.blocks'], t[prefix + '.total'], t[prefix + '.all_f32'] = jobs, jb, int(total), 1

        g = gu is not None
        for i, (blk, w) in enumerate(zip(model.conv_blocks, pr.blocks)):
            p = 'conv_blocks.%d.' % i
            for k in ('w1f', 'w2f', 'wr', 'wrT', 'w2b', 'w1b_even', 'w1b_odd'):
                if k in w:
                    t[p + k] = w[k]
            t[p + 'conv1.bias'], t[p + 'residual_path.bias'], t[p + 'conv2.bias'] = blk.conv1.bias, blk.residual_path.bias, blk.conv2.bias
            bn(p + 'bn1', blk.bn1, g); bn(p + 'bn2', blk.bn2, g); bn(p + 'res_norm', blk.res_norm, g)
            if g:
                t[p + 'conv2.weight.stage'], t[p + 'conv1.weight.stage'] = gu.buf['c2_%d' % i], gu.buf['c1_%d' % i]
                t[p + 'residual_path.weight.grad'] = blk.residual_path.weight.grad
                batch(p + 'unpack', gu.conv_batches[i])
        t['w_raw_in'], t['w_raw_in_T'], t['w_raw_in.bias'] = pr.w_raw_in, pr.w_raw_in_T, model.w_raw_in.bias
        if g:
            t['w_raw_in.weight.grad'], t['w_raw_in.bias.grad'] = model.w_raw_in.weight.grad, model.w_raw_in.bias.grad
        for l, (layer, w) in enumerate(zip(model.transformer.layers, pr.layers)):
            p = 'transformer.layers.%d.' % l
            for k in ('wqkv', 'wqkvT', 'wo', 'woT', 'E', 'ET', 'EF', 'w1', 'w2', 'w1T', 'w2T'):
                t[p + k] = w[k]
            t[p + 'linear1.bias'], t[p + 'linear2.bias'] = layer.linear1.bias, layer.linear2.bias
            t[p + 'norm1.weight'], t[p + 'norm1.bias'], t[p + 'norm2.weight'], t[p + 'norm2.bias'] = layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias
            if g:
                for k in ('norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias', 'linear1.weight', 'linear1.bias', 'linear2.weight', 'linear2.bias'):
                    mod, attr = k.split('.')
                    t[p + k + '.grad'] = getattr(getattr(layer, mod), attr).grad
                t[p + 'w_o.stage'], t[p + 'w_qkv.stage'] = gu.buf['wo%d' % l], gu.buf['wqkv%d' % l]
                batch(p + 'unpack', gu.layer_batches[l])
        t['w_head'], t['w_head_T'], t['b_head'] = pr.w_head, pr.w_head_T, pr.b_head
        if g:
            t['head_w.stage'], t['head_b.stage'] = gu.buf['head_w'], gu.buf['head_b']
            t['stage_arena'], t['stage_arena.bytes'] = gu.arena, int(gu.arena.numel() * 4)
            batch('unpack_encoder', gu.head_batch)
            batch('unpack_encoder_all', gu.all_batch)
        return t

    def ensure_bound(self, model, pr, gu, dev):
        sig = (pr.layout_sig, gu.sig if gu is not None else None, tuple(b.data_ptr() for b in model.all_buffers()))
        if sig == self.sig:
            return
        table = self._table(model, pr, gu, dev)
        for i, n in enumerate(self.names):
            v = table.get(n)
            if v is None:
                val = None
            elif isinstance(v, int):
                val = ctypes.c_void_p(v)
            else:
                val = _lib.ptr(v.detach() if v.requires_grad else v)
            _lib.check(self.lib.ss_plan_bind(self.handle, i, val), 'ss_plan_bind')
        self.keep, self.sig = table, sig

    def set_reduce_hook(self, fn):
        """fn(sums_tensor, n_local) -> n_total (all-reduces the per-channel BatchNorm sums across data-parallel ranks) or None."""
        if fn is self._hook_fn:
            return
        self._hook_fn = fn
        if fn is None:
            self._hook = None
            self.lib.ss_plan_set_reduce_hook(self.handle, ctypes.cast(None, _REDUCE_HOOK), None)
            return

        def trampoline(user, sums_ptr, n_floats, n_local, stream):
            try:                                  # an exception cannot cross the C frames of the plan: park it, re-raise after the call
                off = int(sums_ptr) - self.ws.data_ptr()
                sums = self.ws[off:off + 4 * n_floats].view(torch.float32)
                return float(fn(sums, n_local))
            except BaseException as e:            # noqa: BLE001
                self.callback_error = self.callback_error or e
                return float(n_local)
        self._hook = _REDUCE_HOOK(trampoline)
        self.lib.ss_plan_set_reduce_hook(self.handle, self._hook, None)

    def set_event_hook(self, fn):
        """fn(what, stream) is called when a group of parameter gradients is final on `stream` (the raw hipStream_t the plan
        produced them on: the side stream, or the main stream when the side stream is off) -- bucketed all-reduce."""
        if fn is self._event_fn:
            return
        self._event_fn = fn
        if fn is None:
            self._event = None
            self.lib.ss_plan_set_event_hook(self.handle, ctypes.cast(None, _EVENT_HOOK), None)
            return
        def trampoline(user, what, stream):
            try:
                fn(int(what), int(stream or 0))
            except BaseException as e:            # noqa: BLE001
                self.callback_error = self.callback_error or e
        self._event = _EVENT_HOOK(trampoline)
        self.lib.ss_plan_set_event_hook(self.handle, self._event, None)

    def raise_callback_error(self):
        """Re-raises the first exception a reduce / event hook threw inside the last native call (ctypes would only print it)."""
        e, self.callback_error = self.callback_error, None
        if e is not None:
            raise e


def plan_binding(model):
    pb = getattr(model, '_plan', None)
    if pb is None or pb.layout != (model.d_model, len(model.transformer.layers), model.compute_dtype):
        pb = PlanBinding(model)
        model._plan = pb
    return pb


def _side_stream(model, dev):
    if dev.type != 'cuda' or not SIDE_STREAM_ENABLED or os.environ.get('SS_AMD_SIDE_STREAM', '1') == '0':
        return None
    st = getattr(model, '_side_stream', None)
    if st is None or st.device != dev:
        st = torch.cuda.Stream(device=dev)
        model._side_stream = st
    return st


class Ctx(object):
    """What backward needs from forward: the plan's host context struct and the workspace its pointers live in."""

    def __init__(self, buf, ws, M):
        self.buf, self.ws, self.M = buf, ws, M


def forward(model, x_raw, training, shift_r, seed):
    """x_raw (B, 8T, 8) f32 on the GPU -> head [B*T][n_head_cols] f32, the saved context (None in eval mode) and the time-shifted input
    (None without a shift).  x_raw itself is NOT modified (plan option 6): the caller mirrors the reference's in-place shift."""
    pr = prepared(model)
    dev = x_raw.device
    B, T0, Cin0 = x_raw.shape
    if T0 % 8 != 0:
        raise ValueError('raw EMG length %d must be a multiple of 8 (three stride-2 convolutions)' % T0)
    pb = plan_binding(model)
    gu = None
    if training:
        model.flat_arenas()                          # every .grad is (again) a view of the flat gradient arena
        gu = grad_unpack(model, dev)
    pb.ensure_bound(model, pr, gu, dev)
    pb.set_reduce_hook(model._bn_reduce_fn if training else None)
    L = pb.lib
    L.ss_plan_set_option(pb.handle, 5, int(model.f32_matmul == 'bf16x3'))         # (before the sizing pass: the plane form of that mode allocates operand planes)
    L.ss_plan_set_option(pb.handle, 7, int(os.environ.get('SS_AMD_X3_PLANES', '1') != '0'))
    L.ss_plan_set_option(pb.handle, 9, int(os.environ.get('SS_AMD_X3_EMIT', '1') != '0'))           # plane GEMMs write their consumers' planes from the epilogue (A/B: 0 = split passes)
    L.ss_plan_set_option(pb.handle, 10, int(os.environ.get('SS_AMD_SIGN_GATE', '1') != '0'))        # FFN ReLU / dropout gate as sign bits (A/B: 0 = the saved activation)
    L.ss_plan_set_option(pb.handle, 8, int(bool(pr.layers) and all(e['EF.x3'] for e in pr.layers) and os.environ.get('SS_AMD_X3_ATTENTION', '1') != '0'))
    L.ss_plan_set_option(pb.handle, 1, int(os.environ.get('SS_AMD_DW_GROUPED', '1') != '0'))
    nbytes = int(L.ss_plan_workspace_bytes(pb.handle, B, T0, int(training)))
    if nbytes < 0:
        raise RuntimeError('ss_plan_workspace_bytes failed: %s' % L.ss_last_error().decode())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    head = torch.empty(B * (T0 // 8), pr.n_head_cols, dtype=torch.float32, device=dev)
    shifted = torch.empty_like(x_raw) if (training and shift_r > 0) else None
    buf = ctypes.create_string_buffer(pb.ctx_bytes)
    pb.ws = ws
    L.ss_plan_set_option(pb.handle, 6, 1)
    p_drop = model.dropout_p if training else 0.0
    rc = L.ss_plan_forward(pb.handle, _lib.ptr(x_raw), _lib.ptr(shifted), _lib.ptr(ws), nbytes, B, T0, int(training), int(shift_r if training else 0),
                           float(p_drop), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(head), buf, _lib.stream_of(x_raw))
    pb.raise_callback_error()
    _lib.check(rc, 'ss_plan_forward')
    return head, (Ctx(buf, ws, B * (T0 // 8)) if training else None), shifted


def backward(model, ctx, dhead):
    """Accumulates into .grad of every parameter (except the relative-position embeddings, which get
    no gradient in the reference either, transformer.py:214-218).  dhead: [M][n_head_cols] f32."""
    pr = prepared(model)
    dev = dhead.device
    pb = plan_binding(model)
    model.flat_arenas()                              # a zero_grad(set_to_none=True) between forward and backward lands here
    gu = grad_unpack(model, dev)
    pb.ensure_bound(model, pr, gu, dev)
    pb.set_reduce_hook(model._bn_reduce_fn)
    pb.set_event_hook(getattr(model, '_grad_ready_fn', None))
    pb.ws = ctx.ws
    side = _side_stream(model, dev)
    L = pb.lib
    L.ss_plan_set_option(pb.handle, 1, int(os.environ.get('SS_AMD_DW_GROUPED', '1') != '0'))
    L.ss_plan_set_option(pb.handle, 2, int(os.environ.get('SS_AMD_SIDE_BLOCKS', '2')))
    L.ss_plan_set_option(pb.handle, 4, int(os.environ.get('SS_AMD_BN_REGATE', '1') != '0'))
    L.ss_plan_set_option(pb.handle, 5, int(model.f32_matmul == 'bf16x3'))
    L.ss_plan_set_option(pb.handle, 6, 1)
    L.ss_plan_set_option(pb.handle, 7, int(os.environ.get('SS_AMD_X3_PLANES', '1') != '0'))
    L.ss_plan_set_option(pb.handle, 8, int(bool(pr.layers) and all(e['EF.x3'] for e in pr.layers) and os.environ.get('SS_AMD_X3_ATTENTION', '1') != '0'))
    rc = L.ss_plan_backward(pb.handle, ctx.buf, _lib.ptr(dhead), _lib.stream_of(dhead), ctypes.c_void_p(side.cuda_stream) if side is not None else None)
    pb.raise_callback_error()
    _lib.check(rc, 'ss_plan_backward')
