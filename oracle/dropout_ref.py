"""TEST INFRASTRUCTURE ONLY (never imported by silent_speech_amd/): numpy restatement of the dropout draws of the HIP kernels.

The reference draws its dropout masks from torch's global generator (`transformer.py:33,38-39,109`: nn.Dropout on the attention
probabilities, on both residual branches and inside the FFN), which no other implementation can replay.  What CAN be checked is
that our kernels compute the reference's function for THE SAME mask: the kernels' keep decisions are a pure function of
(seed, stream, element index) -- `silent_speech_amd/csrc/common.h` (`mix32`, `dropout_keep4`) and
`silent_speech_amd/csrc/attention.hip` (`res_drop_key`, `res_drop_words`, `res_drop_keep4`) -- restated here in uint32 numpy, and
the oracle model (`oracle/model_ref.model_forward(layer_masks=..., dropout_p=...)`) applies the masks exactly where the reference
applies `nn.Dropout`.  Stream numbering of encoder layer l (`csrc/plan.hip`): 4l attention probabilities, 4l+1 residual branch 1,
4l+2 FFN hidden, 4l+3 residual branch 2.
"""
import numpy as np

U32 = np.uint32
_M32 = 0xFFFFFFFF


def _u32(x):
    return np.asarray(x, dtype=np.uint64).astype(U32) if not (isinstance(x, np.ndarray) and x.dtype == U32) else x


def mix32(x):
    """common.h: mix32 (multiply-xorshift finaliser), element-wise on uint32."""
    x = _u32(x).copy()
    x ^= x >> U32(16); x *= U32(0x7feb352d); x ^= x >> U32(15); x *= U32(0x846ca68b); x ^= x >> U32(16)
    return x


def dropout_threshold(p):
    """common.h: dropout_threshold."""
    t = float(np.float32(p)) * 4294967296.0
    if t <= 0:
        return 0
    if t >= 4294967295.0:
        return 4294967295
    return int(t)


def _seed_fold(seed, stream):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return ((seed & _M32) ^ (((seed >> 32) * 0x9E3779B9) & _M32) ^ ((int(stream) * 0x85EBCA6B) & _M32)) & _M32


def keep4(seed, stream, g4, p):
    """common.h: dropout_keep4.  g4: integer array of group numbers (< 2^63) -> bool array g4.shape + (4,)."""
    with np.errstate(over='ignore'):
        g4 = np.asarray(g4, dtype=np.uint64)
        lo, hi = (g4 & np.uint64(_M32)).astype(U32), (g4 >> np.uint64(32)).astype(U32)
        key = U32(_seed_fold(seed, stream)) ^ (hi * U32(0xC2B2AE35))
        a = mix32(lo * U32(2) + key)
        b = mix32(lo * U32(2) + U32(1) + (key ^ U32(0x68E31DA4)))
        t16 = U32(dropout_threshold(p) >> 16)
        return np.stack([(a & U32(0xffff)) >= t16, (a >> U32(16)) >= t16, (b & U32(0xffff)) >= t16, (b >> U32(16)) >= t16], -1)


def rowwise_mask(seed, stream, rows, C, p):
    """Keep mask (rows, C) of `add_dropout_layernorm` (csrc/norm.hip): element r*C + c <-> group (r*C + c) >> 2, slot c & 3."""
    assert C % 4 == 0
    g4 = np.arange(rows * C // 4, dtype=np.uint64)
    return keep4(seed, stream, g4, p).reshape(rows, C)


def gemm_epilogue_mask(seed, stream, M, N, p):
    """Keep mask (M, N) of the GEMM epilogue dropout (csrc/common.h dropout_keep1 / the epilogues of gemm_common.h and gemm8.hip): element (row, col) is
    element e = row * N + col of the stream: group e >> 2, slot e & 3 (round 6: row-major like add_dropout_layernorm's mask; rounds 1-5 grouped four ROWS of a column)."""
    n = M * N
    g4 = np.arange((n + 3) // 4, dtype=np.uint64)
    return keep4(seed, stream, g4, p).reshape(-1)[:n].reshape(M, N)


def attention_mask_tiled(seed, stream, B, H, T, p):
    """Per-tile attention kernels (f32, or rows longer than the LDS-resident limit; csrc/attention.hip `attn_fwd_kernel`):
    probability (q, k) of pair bh <-> group ((bh*T + q/4)*T + k), slot q & 3.  -> bool (B, H, T, T)."""
    Tq = (T + 3) // 4 * 4
    bh = np.arange(B * H, dtype=np.uint64)[:, None, None]
    qg = np.arange(Tq // 4, dtype=np.uint64)[None, :, None]
    k = np.arange(T, dtype=np.uint64)[None, None, :]
    kp = keep4(seed, stream, (bh * np.uint64(T) + qg) * np.uint64(T) + k, p)          # (BH, Tq/4, T, 4)
    return np.ascontiguousarray(kp.transpose(0, 1, 3, 2)).reshape(B, H, Tq, T)[:, :, :T]


def attention_mask_resident(seed, stream, B, H, T, p):
    """LDS-resident attention kernels (bf16 rows of <= 208 frames; csrc/attention.hip `res_drop_key` / `res_drop_words` /
    `res_drop_keep4`): the 4 query rows 4g..4g+3 of key column k draw 15-bit values from two words hashed from (pair, row group,
    k); an entry is dropped iff its draw is below t15 = threshold >> 17.  -> bool (B, H, T, T)."""
    with np.errstate(over='ignore'):
        Tq = (T + 3) // 4 * 4
        sd = U32(_seed_fold(seed, stream))
        row = (np.arange(B * H, dtype=np.uint64)[:, None] * np.uint64(T) + np.arange(Tq // 4, dtype=np.uint64)[None, :]).astype(U32)
        key = mix32((row * U32(0x9E3779B1)) ^ sd) + sd                                         # (BH, Tq/4)
        k = np.arange(T, dtype=np.uint64).astype(U32)
        a = (key[:, :, None] + U32(2) * k[None, None, :]) * U32(0x7feb352d)
        a ^= a >> U32(15); a *= U32(0x846ca68b); a ^= a >> U32(16)
        b = (a ^ U32(0x68E31DA4)) * U32(0x9E3779B1); b ^= b >> U32(15)
        t15 = U32(dropout_threshold(p) >> 17)
        kp = np.stack([(a & U32(0x7fff)) >= t15, ((a >> U32(16)) & U32(0x7fff)) >= t15,
                       (b & U32(0x7fff)) >= t15, ((b >> U32(16)) & U32(0x7fff)) >= t15], 2)   # (BH, Tq/4, 4, T)
        return kp.reshape(B, H, Tq, T)[:, :, :T]


def attention_mask_transposed(seed, stream, B, H, T, p):
    """Transposed-score attention kernels (bf16 rows of <= 224 frames; csrc/attention_t.hip `drop_key` / `drop_signs`): the 4 consecutive
    keys 4g..4g+3 of query q draw 16 bits each from ONE 32 x 32 -> 64-bit product of x = key(pair, q) + g * 0x632BE5AB with 0x9E3779B1
    (keys 4g, 4g+1: the halves of lo ^ hi; 4g+2, 4g+3: the halves of hi * 0x85EBCA6B + lo); an entry is dropped iff its draw, read as a SIGNED 16-bit
    number, is below t16 - 32768 (saturating subtract, sign bit), t16 = threshold >> 16.  -> bool (B, H, T, T)."""
    with np.errstate(over='ignore'):
        Tk = (T + 3) // 4 * 4
        sd = U32(_seed_fold(seed, stream))
        row = (np.arange(B * H, dtype=np.uint64)[:, None] * np.uint64(T) + np.arange(T, dtype=np.uint64)[None, :]).astype(U32)
        key = mix32((row * U32(0x9E3779B1)) ^ sd) + sd                                         # (BH, T)
        g = np.arange(Tk // 4, dtype=np.uint64).astype(U32)
        x = (key[:, :, None] + g[None, None, :] * U32(0x632BE5AB)).astype(np.uint64)
        prod = x * np.uint64(0x9E3779B1)
        lo, hi = (prod & np.uint64(_M32)).astype(U32), (prod >> np.uint64(32)).astype(U32)
        a = lo ^ hi
        c = hi * U32(0x85EBCA6B) + lo
        draws = np.stack([a & U32(0xffff), a >> U32(16), c & U32(0xffff), c >> U32(16)], -1).astype(np.uint16).view(np.int16).astype(np.int32)   # (BH, T, Tk/4, 4)
        ts = int(dropout_threshold(p) >> 16) - 32768
        keep = draws >= ts
        return keep.reshape(B, H, T, Tk)[..., :T]


def attention_mask(family, seed, stream, B, H, T, p):
    """Keep mask of the attention probabilities for the kernel family ss_relpos_attention_family reports (0 per-tile, 1 LDS-resident
    16 x 16, 2 transposed 32 x 32)."""
    return (attention_mask_tiled, attention_mask_resident, attention_mask_transposed)[family](seed, stream, B, H, T, p)


def layer_masks(seed, num_layers, B, T, d_model, n_head, ff, p, resident_attention):
    """Keep masks of one training forward as torch float tensors in the shapes `model_ref.encoder_layer` applies them:
    [{'attn': (B,H,T,T), 'res1': (B,T,d), 'ffn': (B,T,ff), 'res2': (B,T,d)} for each layer]."""
    import torch
    # resident_attention: bool of the earlier rounds (False = per-tile, True = LDS-resident 16 x 16) or the family number 0 / 1 / 2
    att = (attention_mask_tiled, attention_mask_resident, attention_mask_transposed)[int(resident_attention)]
    out = []
    for l in range(num_layers):
        out.append({
            'attn': torch.from_numpy(att(seed, 4 * l, B, n_head, T, p)).float(),
            'res1': torch.from_numpy(rowwise_mask(seed, 4 * l + 1, B * T, d_model, p)).float().view(B, T, d_model),
            'ffn': torch.from_numpy(gemm_epilogue_mask(seed, 4 * l + 2, B * T, ff, p)).float().view(B, T, ff),
            'res2': torch.from_numpy(rowwise_mask(seed, 4 * l + 3, B * T, d_model, p)).float().view(B, T, d_model),
        })
    return out
