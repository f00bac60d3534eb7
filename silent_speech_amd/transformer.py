"""Parameter containers of the relative-position Transformer encoder -- same class names, constructor
arguments, parameter names/shapes and initialisation as the reference's transformer.py, so reference
checkpoints load with load_state_dict().  Inside `Model` the math is sequenced by the native plan (engine.py, csrc/plan.hip) and these
classes only hold parameters; called on their own, their `forward` runs the same HIP kernels one C-ABI call at a time (eager.py:
inference forwards, no autograd -- training goes through Model)."""
import copy

import torch
from torch import nn


class LearnedRelativePositionalEmbedding(nn.Module):
    """transformer.py:114-160.  Only the configuration the reference uses is supported:
    unmasked=True, heads_share_embeddings=False, add_to_values=False (transformer.py:83)."""

    def __init__(self, max_relative_pos, num_heads, embedding_dim, unmasked=False, heads_share_embeddings=False, add_to_values=False):
        super().__init__()
        if not unmasked or heads_share_embeddings or add_to_values:
            raise NotImplementedError('only the encoder configuration used by the reference (unmasked, per-head, keys only) is implemented')
        self.max_relative_pos, self.num_heads, self.embedding_dim = max_relative_pos, num_heads, embedding_dim
        self.unmasked, self.heads_share_embeddings, self.add_to_values = unmasked, heads_share_embeddings, add_to_values
        self.embeddings = nn.Parameter(torch.zeros(num_heads, 2 * max_relative_pos - 1, embedding_dim, 1))
        nn.init.normal_(self.embeddings, mean=0.0, std=embedding_dim ** (-0.5))      # transformer.py:158-160


class MultiHeadAttention(nn.Module):
    """transformer.py:62-85: bias-free per-head projections w_q/w_k/w_v (H, d, d_qkv), w_o (H, d_qkv, d)."""

    def __init__(self, d_model=256, n_head=4, dropout=0.1, relative_positional=True, relative_positional_distance=100):
        super().__init__()
        self.d_model, self.n_head = d_model, n_head
        d_qkv = d_model // n_head
        assert d_qkv * n_head == d_model, 'd_model must be divisible by n_head'        # transformer.py:68
        self.d_qkv = d_qkv
        self.w_q = nn.Parameter(torch.Tensor(n_head, d_model, d_qkv))
        self.w_k = nn.Parameter(torch.Tensor(n_head, d_model, d_qkv))
        self.w_v = nn.Parameter(torch.Tensor(n_head, d_model, d_qkv))
        self.w_o = nn.Parameter(torch.Tensor(n_head, d_qkv, d_model))
        for w in (self.w_q, self.w_k, self.w_v, self.w_o):
            nn.init.xavier_normal_(w)
        self.dropout = nn.Dropout(dropout)
        if not relative_positional:
            raise NotImplementedError('the HIP attention kernel implements the relative-positional variant the reference trains')
        self.relative_positional = LearnedRelativePositionalEmbedding(relative_positional_distance, n_head, d_qkv, True)

    def forward(self, x):
        """transformer.py:87-112: x (length, batch, d_model) -> (length, batch, d_model)."""
        from . import eager
        return eager.mha_forward(self, x)


class TransformerEncoderLayer(nn.Module):
    """transformer.py:7-41 (post-norm; masks accepted and ignored by the reference, :43,54)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, relative_positional=True, relative_positional_distance=100):
        super().__init__()
        self.self_attn = MultiHeadAttention(d_model, nhead, dropout=dropout, relative_positional=relative_positional,
                                            relative_positional_distance=relative_positional_distance)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = nn.ReLU()

    def forward(self, src, src_mask=None, src_key_padding_mask=None, is_causal=False):
        """transformer.py:43-60 (the mask arguments are accepted and ignored, as in the reference)."""
        from . import eager
        return eager.encoder_layer_forward(self, src, src_mask, src_key_padding_mask, is_causal)


class TransformerEncoder(nn.Module):
    """Same state_dict keys as nn.TransformerEncoder(encoder_layer, num_layers) (architecture.py:54):
    `layers.{i}.*`, every layer a deep copy of the first (identical initial weights), no final norm."""

    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    def forward(self, src, mask=None, src_key_padding_mask=None):
        from . import eager
        return eager.encoder_forward(self, src, mask, src_key_padding_mask)
