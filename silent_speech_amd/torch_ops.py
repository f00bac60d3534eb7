"""PyTorch dispatcher registration of the hot path: `torch.ops.silent_speech.*` (torch.library custom ops over the C ABI).

    model_forward / model_backward   Model.forward (architecture.py:61-84) and what loss.backward() triggers: one native plan each
    dtw_loss                         transduction_model.py:98-157 (cost matrices in strip layout, DTW + backtrace, loss and d loss / d head)
    dtw_align                        align.py:16-34 on one device matrix
    ctc_loss                         recognition_model.py:96-101
    stft_logmel                      data_utils.py:39-62
    fused_adamw                      torch.optim.AdamW over the flat parameter arena (transduction_model.py:178,210)

The package's own entry points (Model.forward, dtw_loss, ctc_loss, FusedAdamW.step, mel_spectrogram, align_from_distances) call these
ops, so the drop-in surface reaches the kernels THROUGH the dispatcher; every op has a fake (meta) implementation for shape inference
and, where the reference differentiates through it, a registered autograd formula whose backward is itself one of the ops.  Ops take
tensors and plain scalars only; per-model state (bound plan, weight copies) is looked up through an integer handle.
Emulator builds (tests) run the same registrations on CPU tensors.
"""
import weakref
from typing import Tuple

import torch
from torch import Tensor

from . import _lib, ops

_L = _lib.lib
_p = _lib.ptr

# ------------------------------------------------------------------------------------------------ model handles
_models = {}
_next_id = [1]
_KEEP = 2          # saved forward contexts per model (a recognition step runs forward / backward twice before its optimiser step)


def model_handle(model):
    h = getattr(model, '_op_handle', None)
    if h is None:
        h = model._op_handle = _next_id[0]
        _next_id[0] += 1
        _models[h] = weakref.ref(model)
        model._saved_ctx = {}
    return h


def _model(handle):
    ref = _models.get(int(handle))
    m = ref() if ref is not None else None
    if m is None:
        raise RuntimeError('silent_speech::model_*: unknown or freed model handle %d' % handle)
    return m


@torch.library.custom_op('silent_speech::model_forward', mutates_args=())
def model_forward(x_raw: Tensor, anchor: Tensor, handle: int, training: bool, shift_r: int, seed: int) -> Tuple[Tensor, Tensor]:
    """x_raw (B, 8 T, 8) f32 -> head [B T][n_head_cols] f32 = [mel prediction | phoneme logits | pad] per frame.  `anchor` is a
    1-element tensor that requires grad: the parameters are updated in place by model_backward (their .grad live in the flat
    arena), so autograd needs one differentiable input to call the backward formula at all.  The op is FUNCTIONAL: x_raw is not touched; the
    time-shifted signal of a training-mode forward is the second output (empty without a shift) and Model.forward copies it back into its
    argument, mirroring the reference's in-place shift (architecture.py:67-68) outside the operator.  The forward context is kept for the
    backward only when the anchor asks for a gradient (Model.forward hands over a detached anchor under no_grad / in eval mode): a forward
    that will never be back-propagated does not pin its multi-GB workspace."""
    from . import engine
    m = _model(handle)
    head, saved, shifted = engine.forward(m, x_raw, training, shift_r, seed)
    if saved is not None and anchor.requires_grad:
        ctxs = m._saved_ctx
        while len(ctxs) >= _KEEP:                      # a training-mode forward that is never back-propagated must not pin its 5 GB workspace
            ctxs.pop(next(iter(ctxs)))
        ctxs[int(seed)] = saved
    return head, (shifted if shifted is not None else x_raw.new_empty(0))


@model_forward.register_fake
def _(x_raw, anchor, handle, training, shift_r, seed):
    from . import engine
    m = _model(handle)
    return (x_raw.new_empty((x_raw.shape[0] * (x_raw.shape[1] // 8), engine.prepared(m).n_head_cols), dtype=torch.float32),
            torch.empty_like(x_raw) if (training and shift_r > 0) else x_raw.new_empty(0))


@torch.library.custom_op('silent_speech::model_backward', mutates_args=())
def model_backward(dhead: Tensor, handle: int, seed: int) -> None:
    """Accumulates into the .grad of every parameter of the model (flat gradient arena) from d loss / d head."""
    from . import engine
    m = _model(handle)
    saved = m._saved_ctx.pop(int(seed), None)
    if saved is None:
        raise RuntimeError('backward through a forward pass that ran in eval mode, ran twice, or was displaced by %d later forward passes' % _KEEP)
    engine.backward(m, saved, dhead)


@model_backward.register_fake
def _(dhead, handle, seed):
    return None


def _model_setup(ctx, inputs, output):
    ctx.handle, ctx.seed, ctx.training = inputs[2], inputs[5], inputs[3]


def _model_bwd(ctx, dhead, g_shifted):
    if not ctx.training:
        raise RuntimeError('backward through a forward pass that ran in eval mode')
    torch.ops.silent_speech.model_backward(dhead.contiguous(), ctx.handle, ctx.seed)
    return None, None, None, None, None, None


model_forward.register_autograd(_model_bwd, setup_context=_model_setup)


# ------------------------------------------------------------------------------------------------ dtw_loss
@torch.library.custom_op('silent_speech::dtw_loss', mutates_args=())
def dtw_loss(head: Tensor, Y: Tensor, phones: Tensor, idx: Tensor, desc: Tensor, n_mel: int, n_ph: int, lam: float, inv_total: float,
             n_voiced: int, n_silent_frames: int, n_silent: int, ws_bytes: int, res_total: int, max_n: int, max_m: int,
             cells: float, perimeter: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """head [M][ld] f32; Y / phones: all utterances' targets back to back; idx: the five per-frame int32 tables of ss_loss_index_tables
    back to back (vo_pred, vo_tgt | si_tgt, si_base, si_res; sizes max(n_voiced, 1) resp. max(n_silent_frames, 1)); desc: the DTW
    descriptors of the silent utterances.  Returns (loss [1], correct [1] int32, d loss / d head, alignment results, per-frame arg-max)."""
    dev = head.device
    M, ld = head.shape
    st = _lib.stream_of(head)
    nv, ns = max(n_voiced, 1), max(n_silent_frames, 1)
    vo_pred, vo_tgt = idx[:nv], idx[nv:2 * nv]
    si_tgt, si_base, si_res = idx[2 * nv:2 * nv + ns], idx[2 * nv + ns:2 * nv + 2 * ns], idx[2 * nv + 2 * ns:]
    lse = torch.empty(M, dtype=torch.float32, device=dev)
    amax = torch.empty(M, dtype=torch.int32, device=dev)
    dhead = torch.zeros_like(head)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    correct = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_L().ss_frame_lse(_p(head), ld, n_mel, n_ph, M, _p(lse), _p(amax), st), 'ss_frame_lse')
    if n_voiced:
        _lib.check(_L().ss_voiced_loss(_p(head), ld, n_mel, n_ph, _p(lse), _p(amax), _p(Y), _p(phones), _p(vo_pred), _p(vo_tgt),
                                       n_voiced, lam, inv_total, _p(dhead), _p(loss), _p(correct), st), 'ss_voiced_loss')
    results = torch.empty(max(res_total, 1), dtype=torch.int32, device=dev)
    if n_silent:
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        ops.timed('silent_cost_skewed_kernel', 0, 4.0 * cells + 4.0 * (n_mel + n_ph) * perimeter,
                  lambda: _lib.check(_L().ss_silent_cost_skewed(_p(head), ld, n_mel, _p(lse), _p(Y), _p(phones), _p(desc), n_silent, max_n, max_m,
                                                                lam, _p(ws), _p(results), st), 'ss_silent_cost_skewed'))
        ops.timed('dtw_kernel', 0, 8.0 * cells,         # 8 N M bytes per matrix (f32 cost in + f32 cumulative out)
                  lambda: _lib.check(_L().ss_dtw_align_skewed(_p(desc), n_silent, _p(ws), _p(results), st), 'ss_dtw_align_skewed'))
        _lib.check(_L().ss_silent_loss(_p(head), ld, n_mel, n_ph, _p(lse), _p(amax), _p(Y), _p(phones), _p(results), _p(si_tgt),
                                       _p(si_base), _p(si_res), n_silent_frames, lam, inv_total, _p(dhead), _p(loss), _p(correct), st), 'ss_silent_loss')
    return loss, correct, dhead, results, amax


@dtw_loss.register_fake
def _(head, Y, phones, idx, desc, n_mel, n_ph, lam, inv_total, n_voiced, n_silent_frames, n_silent, ws_bytes, res_total, max_n, max_m, cells, perimeter):
    M = head.shape[0]
    return (head.new_empty(1), head.new_empty(1, dtype=torch.int32), torch.empty_like(head), head.new_empty(max(res_total, 1), dtype=torch.int32),
            head.new_empty(M, dtype=torch.int32))


def _dtw_setup(ctx, inputs, output):
    ctx.save_for_backward(output[2])           # (an op output kept as a plain attribute would form a tensor -> grad_fn -> ctx -> tensor cycle)


def _dtw_bwd(ctx, g_loss, g_correct, g_dhead, g_results, g_amax):
    return (ctx.saved_tensors[0] * g_loss,) + (None,) * 17


dtw_loss.register_autograd(_dtw_bwd, setup_context=_dtw_setup)


# ------------------------------------------------------------------------------------------------ dtw_align
@torch.library.custom_op('silent_speech::dtw_align', mutates_args=())
def dtw_align(costs: Tensor) -> Tensor:
    """align.py:16-34 on one (N, M) float32 device matrix (any strides: `costs.T` is read in place): int32 [N], results[i] = the column
    the optimal monotone path visits last in row i (first-minimum tie order up, left, diagonal, bit-exact with the reference)."""
    from .align import dtw_align_batch
    if costs.dim() != 2 or costs.dtype != torch.float32:
        raise ValueError('dtw_align: a 2-D float32 matrix is expected')
    N, M = costs.shape
    res, _ = dtw_align_batch(costs, [(N, M)], [0], [costs.stride()])
    return res[:N].clone()


@dtw_align.register_fake
def _(costs):
    return costs.new_empty(costs.shape[0], dtype=torch.int32)


# ------------------------------------------------------------------------------------------------ ctc_loss
@torch.library.custom_op('silent_speech::ctc_loss', mutates_args=())
def ctc_loss(logits: Tensor, desc: Tensor, targets: Tensor, n: int, max_s: int, ws_floats: int, V: int, blank: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """logits [M][V] f32 raw model outputs of the packed rows; desc [n][5] int64 = (first frame, frames, first label, labels, workspace
    offset) per utterance; targets int32.  Returns (mean loss [1], d loss / d logits, per-utterance nll, per-frame arg-max)."""
    dev = logits.device
    M, ld = logits.shape
    st = _lib.stream_of(logits)
    lse = torch.empty(M, dtype=torch.float32, device=dev)
    amax = torch.empty(M, dtype=torch.int32, device=dev)
    _lib.check(_L().ss_frame_lse(_p(logits), ld, 0, V, M, _p(lse), _p(amax), st), 'ss_frame_lse')
    ws = torch.empty(2 * max(ws_floats, 1), dtype=torch.float32, device=dev)
    nll = torch.empty(max(n, 1), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    _lib.check(_L().ss_ctc_loss(_p(logits), ld, V, blank, _p(lse), _p(desc) if n else None, n, max_s, M, _p(targets),
                                _p(ws), _p(ws[max(ws_floats, 1):]), _p(nll), _p(dlogits), _p(loss), st), 'ss_ctc_loss')
    return loss, dlogits, nll, amax


@ctc_loss.register_fake
def _(logits, desc, targets, n, max_s, ws_floats, V, blank):
    return logits.new_empty(1), torch.empty_like(logits), logits.new_empty(max(n, 1)), logits.new_empty(logits.shape[0], dtype=torch.int32)


def _ctc_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _ctc_bwd(ctx, g_loss, g_dl, g_nll, g_amax):
    return (ctx.saved_tensors[0] * g_loss,) + (None,) * 7


ctc_loss.register_autograd(_ctc_bwd, setup_context=_ctc_setup)


# ------------------------------------------------------------------------------------------------ stft_logmel
@torch.library.custom_op('silent_speech::stft_logmel', mutates_args=())
def stft_logmel(y: Tensor, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int, fmin: int, fmax: int, center: bool) -> Tensor:
    """data_utils.py:39-62: (B, L) float32 -> (B, num_mels, F) log-mel."""
    from .data_utils import _mel_spectrogram_impl
    return _mel_spectrogram_impl(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center)


@stft_logmel.register_fake
def _(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center):
    Lp = y.shape[1] + 2 * int((n_fft - hop_size) / 2) + (2 * (n_fft // 2) if center else 0)
    return y.new_empty(y.shape[0], num_mels, 1 + (Lp - n_fft) // hop_size)


# ------------------------------------------------------------------------------------------------ fused_adamw
@torch.library.custom_op('silent_speech::fused_adamw', mutates_args=('p', 'm', 'v'))
def fused_adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, n: int, lr: float, step: int, beta1: float, beta2: float, eps: float,
                weight_decay: float, grad_scale: float) -> None:
    """One decoupled-weight-decay Adam step over the first n floats of the flat arenas (torch.optim.AdamW semantics, bias correction from `step`)."""
    ops.adamw_step(p, g, m, v, n, lr, step, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale)


@fused_adamw.register_fake
def _(p, g, m, v, n, lr, step, beta1, beta2, eps, weight_decay, grad_scale):
    return None


OPS = ('model_forward', 'model_backward', 'dtw_loss', 'dtw_align', 'ctc_loss', 'stft_logmel', 'fused_adamw')
