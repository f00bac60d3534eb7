"""EMG -> mel transduction model on the MI355X -- drop-in for the reference's architecture.py.

    Model(num_features, num_outs, num_aux_outs=None)          architecture.py:43
    .forward(x_feat, x_raw, session_ids) -> pred | (pred, aux)  architecture.py:61-84

Same parameter names / shapes / initialisation (so `state_dict()` round-trips with reference
checkpoints), same train()/eval() semantics (random 0-7 sample shift augmentation drawn from Python's
`random`, dropout, BatchNorm batch statistics + running-stat updates), `x_feat` and `session_ids`
accepted and ignored exactly like the reference.  All compute runs in hand-written HIP kernels
(engine.py sequences them); there is no eager / CPU fallback.

Extras (keyword-only, defaults reproduce the reference): model_size / num_layers / dropout override the
absl FLAGS the reference reads globally (architecture.py:47-54); compute_dtype selects bf16 MFMA
(default, BASELINE config 2) or f32 tensors, whose matmuls run either on exact-f32 MFMAs (f32_matmul='exact') or on
three bf16 MFMAs per product over operands split hi + lo in registers (f32_matmul='bf16x3': ~1e-5 of the exact result
at 1.8 x the speed; the fastest mode inside north_star's 1e-4 mel-L1).
"""
import random

import torch
from torch import nn

from . import engine, torch_ops
from .flags import FLAGS
from .transformer import TransformerEncoder, TransformerEncoderLayer


class ResBlock(nn.Module):
    """architecture.py:14-40.  Inside `Model` the block is part of the native plan (engine.py); `forward` on its own runs the same
    kernels call by call (eager.py: inference forward, no autograd)."""

    def __init__(self, num_ins, num_outs, stride=1):
        super().__init__()
        self.conv1 = nn.Conv1d(num_ins, num_outs, 3, padding=1, stride=stride)
        self.bn1 = nn.BatchNorm1d(num_outs)
        self.conv2 = nn.Conv1d(num_outs, num_outs, 3, padding=1)
        self.bn2 = nn.BatchNorm1d(num_outs)
        if stride != 1 or num_ins != num_outs:
            self.residual_path = nn.Conv1d(num_ins, num_outs, 1, stride=stride)
            self.res_norm = nn.BatchNorm1d(num_outs)
        else:
            self.residual_path = None
        self.stride = stride

    def forward(self, x):
        """architecture.py:29-40: x (batch, channels, time) -> (batch, num_outs, time / stride)."""
        from . import eager
        return eager.resblock_forward(self, x)


class Model(nn.Module):
    def __init__(self, num_features, num_outs, num_aux_outs=None, *, model_size=None, num_layers=None, dropout=None,
                 compute_dtype=torch.bfloat16, f32_matmul='exact'):
        super().__init__()
        model_size = FLAGS.model_size if model_size is None else model_size
        num_layers = FLAGS.num_layers if num_layers is None else num_layers
        dropout = FLAGS.dropout if dropout is None else dropout
        if model_size % 8 != 0:
            raise ValueError('model_size must be a multiple of 8 (nhead=8, 16-byte channel vectors)')
        self.conv_blocks = nn.Sequential(ResBlock(8, model_size, 2), ResBlock(model_size, model_size, 2), ResBlock(model_size, model_size, 2))
        self.w_raw_in = nn.Linear(model_size, model_size)
        encoder_layer = TransformerEncoderLayer(d_model=model_size, nhead=8, relative_positional=True, relative_positional_distance=100,
                                                dim_feedforward=3072, dropout=dropout)
        self.transformer = TransformerEncoder(encoder_layer, num_layers)
        self.w_out = nn.Linear(model_size, num_outs)
        self.has_aux_out = num_aux_outs is not None
        if self.has_aux_out:
            self.w_aux = nn.Linear(model_size, num_aux_outs)
        # ---- MI355X execution state (not part of the state_dict)
        self.d_model, self.n_head, self.max_rel = model_size, 8, 100
        self.d_qkv = model_size // 8
        self.dp = (self.d_qkv + 31) // 32 * 32            # head dim zero-padded to the MFMA K granularity
        self.dropout_p = float(dropout)
        self.num_outs, self.num_aux_outs = num_outs, num_aux_outs
        if compute_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError('compute_dtype must be torch.bfloat16 or torch.float32')
        if f32_matmul not in ('exact', 'bf16x3'):
            raise ValueError("f32_matmul must be 'exact' (f32 MFMA) or 'bf16x3' (f32 storage, three bf16 MFMAs per product)")
        self.compute_dtype = compute_dtype
        self.f32_matmul = f32_matmul                      # read by the plan when compute_dtype is float32 (engine.py)
        self._weights_version = 0
        self._bn_reduce_fn = None                         # set by the data-parallel wrapper (SyncBN-equivalent statistics)
        self._seed_base, self._step = 0x5EED, 0
        self.shift_rng = random                           # architecture.py:65 draws r = random.randrange(8)
        self._anchor = None
        self._flat = None
        self._cache = {}

    # ------------------------------------------------------------------ flat parameter / gradient arenas
    def optimized_parameters(self):
        """Every parameter except the relative-position embeddings, which never receive a gradient in the
        reference (padded under no_grad, transformer.py:214-218) and are therefore never updated by AdamW."""
        c = self._cache.get('opt')
        if c is None:
            c = self._cache['opt'] = [p for n, p in self.named_parameters() if 'relative_positional' not in n]
        return c

    def all_parameters(self):
        """list(self.parameters()), cached: the per-step signature checks (engine.py) walk the parameters several times, and
        nn.Module.parameters() re-traverses the module tree on every call (1.5 ms of host time per step before the cache).
        The module tree is fixed after construction; Module._apply (.to / .cuda / .float) drops the cache."""
        c = self._cache.get('all')
        if c is None:
            c = self._cache['all'] = list(self.parameters())
        return c

    def all_buffers(self):
        c = self._cache.get('buf')
        if c is None:
            c = self._cache['buf'] = list(self.buffers())
        return c

    def _apply(self, fn, *a, **k):
        self._cache.clear()
        return super()._apply(fn, *a, **k)

    def flatten_parameters(self):
        """Re-home all optimised parameters (and their .grad) in two contiguous f32 arenas: one fused AdamW
        launch and one gradient all-reduce cover the whole model.  Views keep names/shapes intact."""
        ps = self.optimized_parameters()
        if not ps:
            return
        dev = ps[0].device
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4                 # 16-byte aligned slots
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        gflat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(ps, offs):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1).to(torch.float32))
                p.data = flat[o:o + p.numel()].view(p.shape)
                p.grad = gflat[o:o + p.numel()].view(p.shape)
        self._flat, self._gflat, self._flat_n = flat, gflat, n
        self._weights_version += 1

    def flat_arenas(self):
        if self._flat is None or self._flat.device != self.w_out.weight.device or self.w_out.weight.data_ptr() < self._flat.data_ptr() \
                or self.w_out.weight.data_ptr() >= self._flat.data_ptr() + 4 * self._flat_n:
            self.flatten_parameters()
        # zero_grad(set_to_none=True) detaches the gradients, and anything that assigned p.grad (a fresh zeros_like, a checkpoint
        # loader) points them outside the arena: the fused optimiser and the gradient all-reduce would then read a stale arena.
        lo, hi = self._gflat.data_ptr(), self._gflat.data_ptr() + 4 * self._flat_n
        for p in self.optimized_parameters():
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                self._reattach_grads()
                break
        return self._flat, self._gflat, self._flat_n

    def _reattach_grads(self):
        """Gradients that live outside the arena are carried over (None counts as zero), then every .grad is a view of it again."""
        off = 0
        lo, hi = self._gflat.data_ptr(), self._gflat.data_ptr() + 4 * self._flat_n
        with torch.no_grad():
            for p in self.optimized_parameters():
                view = self._gflat[off:off + p.numel()].view(p.shape)
                g = p.grad
                if g is None:
                    view.zero_()
                elif not (lo <= g.data_ptr() < hi):
                    view.copy_(g.to(torch.float32))
                p.grad = view
                off += (p.numel() + 3) // 4 * 4

    def arena_ranges(self):
        """Parameter name -> (first, one-past-last) float offsets in the flat arenas, in arena order."""
        out, off = [], 0
        for n, p in self.named_parameters():
            if 'relative_positional' in n:
                continue
            out.append((n, off, off + p.numel()))
            off += (p.numel() + 3) // 4 * 4
        return out

    def mark_weights_updated(self):
        """Called by optimisers that update the arena through the C ABI (no torch version bump)."""
        self._weights_version += 1

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._weights_version += 1
        return r

    def attention_mask_family(self, T):
        """Which dropout-mask function the attention of a training forward on rows of T frames draws from (oracle/dropout_ref.attention_mask):
        0 per-tile kernels, 1 LDS-resident 16 x 16 (A/B builds), 2 transposed-score kernels -- bf16 plans ask the library; an f32 plan runs the
        per-tile kernels unless it is the parity-grade mode on hi / lo planes (f32_matmul='bf16x3'), whose attention is the transposed-score one."""
        import os
        from . import _lib
        L = _lib.lib()
        if self.compute_dtype == torch.bfloat16:
            return int(L.ss_relpos_attention_family(_lib.dtype_code(self.compute_dtype), T, self.dp, self.max_rel))
        planes = self.f32_matmul == 'bf16x3' and all(os.environ.get(k, '1') != '0' for k in ('SS_AMD_X3_PLANES', 'SS_AMD_X3_ATTENTION', 'SS_AMD_DW_GROUPED'))
        return 2 if planes and L.ss_relpos_attention_x3_supported(T, self.dp, self.max_rel) else 0

    def set_seed(self, seed):
        self._seed_base = int(seed)

    # ------------------------------------------------------------------ forward
    def forward(self, x_feat, x_raw, session_ids):
        # x shape is (batch, time, electrode); x_feat and session_ids are unused, as in the reference
        if x_raw.dim() != 3 or x_raw.shape[2] != 8:
            raise ValueError('x_raw must be (batch, time, 8)')
        if x_raw.dtype != torch.float32:
            raise TypeError('x_raw must be float32')
        r = 0
        if self.training:
            r = self.shift_rng.randrange(8)               # architecture.py:65
            self.flat_arenas()
        xr = x_raw if x_raw.is_contiguous() else x_raw.contiguous()
        self._step += 1
        seed = (self._seed_base * 0x9E3779B1 + self._step) & 0x7FFFFFFFFFFFFFFF      # an int64 operator argument
        self.last_seed = seed                             # dropout draws are a pure function of (seed, stream, element): tests replay them
        if self._anchor is None or self._anchor.device != xr.device:
            self._anchor = torch.zeros(1, device=xr.device, requires_grad=True)
        # ONE dispatcher op for the whole network (torch_ops.py): forward = the native plan; when gradients are wanted its registered
        # autograd formula runs silent_speech::model_backward, which accumulates straight into the flat .grad arena.
        track = self.training and torch.is_grad_enabled()
        head, shifted = torch.ops.silent_speech.model_forward(xr, self._anchor if track else self._anchor.detach(), torch_ops.model_handle(self),
                                                              bool(self.training), int(r), int(seed))
        if self.training and r > 0:
            with torch.no_grad():
                x_raw.copy_(shifted)                       # x[:, :-r] = x[:, r:]; x[:, -r:] = 0 on the caller's tensor (architecture.py:67-68)
        B, T = x_raw.shape[0], x_raw.shape[1] // 8
        n_out = self.num_outs
        pred = head[:, :n_out].view(B, T, n_out)
        if self.has_aux_out:
            return pred, head[:, n_out:n_out + self.num_aux_outs].view(B, T, self.num_aux_outs)
        return pred
