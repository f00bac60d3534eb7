// attention.hip -- fused multi-head self-attention with learned relative-position logits (gfx950).
// Reference: transformer.py:87-112 MultiHeadAttention.forward and :162-297
// LearnedRelativePositionalEmbedding (unmasked / per-head / keys-only path), in closed form:
//     logits[b,h,q,k] = Q.K / sqrt(d_qkv) + ( |k-q| <= D-1 ?  Q[q] . E[h, k-q+D-1]  :  -1e8 )      (Q unscaled in the 2nd term)
//     P = softmax_k(logits);  P~ = dropout(P);  O = P~ V
// The -1e8 makes every out-of-band probability exactly 0 in f32 (as in the reference), so the attention is
// BANDED: a 16-row query tile only ever touches <= 16 + 2(D-1) keys, whatever the sequence length, and the
// whole logit row fits in registers (no online-softmax rescaling, one pass).  E receives no gradient
// (transformer.py:214-218 pads it under no_grad), so backward produces dQ, dK, dV only.
//
// Layouts (compute dtype T = bf16 or f32, head dim zero-padded to dp = 32*DPK):
//   qkv  [B*T][3*H*dp]  rows = frames, columns (q|k|v, head, d)      -- written by the fused QKV GEMM
//   qkvT [B][3*H*dp][Tp] the same values transposed per sequence      -- 2nd output of that GEMM's epilogue
//   E    [H][2D-1][dp],  ET [H][dp][MPt] (m contiguous, zero padded)  -- ss_permute3d of the embeddings
// so every MFMA operand fragment (8 consecutive contraction elements of one row) is ONE aligned 16-byte
// global load: nothing is transposed through LDS except the probability / dS tiles (accumulator layout ->
// A-operand layout).  One wave owns one 16-row tile; 4 independent waves per workgroup.
//   forward : S = Q K^T and R = Q E^T on MFMA (R blocks slide along the band: 3+3 MFMA per 16x16 logits),
//             relative->absolute "skew" = a rotation inside each 16-lane group (2 ds_bpermute per value),
//             softmax by 16-lane xor-shuffles, P -> LDS -> A fragments, O = P V^T(qkvT) on MFMA.
//   backward: query-major kernel (dQ, incl. the positional term via the un-skewed dS tile times ET) and
//             key-major kernel (dK, dV); both recompute P from the saved log-sum-exp.
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>

namespace {

// XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with a private L2).
// The tiles of one (batch, head) re-read the same Q/K/V/dO rows, so consecutive logical ids are mapped onto ONE XCD
// (bijective chunked remap) and their re-reads hit that XCD's L2 instead of HBM.
__device__ __forceinline__ void attn_block_coord(int gx, int H, int& bx, int& h, int& b) {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    bx = id % gx; const int t = id / gx; h = t % H; b = t / H;
}

struct AttnP {
    const void* qkv; const void* qkvT; const void* E; const void* ET;
    void* out; float* lse;
    const void* dO; const void* dOT; const float* Dv; void* dqkv;
    int B, H, T, Tp, dp, D, MPt, gx;
    float scale;
    unsigned drop_thresh; float drop_scale; unsigned long long seed; unsigned stream;
};

constexpr int NB_MAX = 16;      // 16-key blocks per query tile: ceil((31 + 16 + 2*99)/16)
constexpr int PT_LD = 256 + 8;  // probability tile row length (elements)

template <class T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 lo, hi; };

__device__ __forceinline__ void frag_zero(Frag<bf16_t>& f) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.v = z; }
__device__ __forceinline__ void frag_zero(Frag<float>& f) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; f.lo = z; f.hi = z; }
__device__ __forceinline__ void frag_load(Frag<bf16_t>& f, const bf16_t* p) { f.v = *(const bf16x8*)p; }
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) { f.lo = *(const f32x4*)p; f.hi = *(const f32x4*)(p + 4); }
// keep only the first n (0..8) elements
__device__ __forceinline__ void frag_keep(Frag<bf16_t>& f, int n) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (e >= n) f.v[e] = 0;
}
__device__ __forceinline__ void frag_keep(Frag<float>& f, int n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (e >= n) f.lo[e] = 0.f; if (e + 4 >= n) f.hi[e] = 0.f; }
}
__device__ __forceinline__ f32x4 mma32(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4 c) { return mfma_bf16_16x16x32(a.v, b.v, c); }
__device__ __forceinline__ f32x4 mma32(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = mfma_f32_16x16x4(a.lo[e], b.lo[e], c);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = mfma_f32_16x16x4(a.hi[e], b.hi[e], c);
    return c;
}

// fragments of one matrix row (8 consecutive d per lane per 32-deep step); zero if !valid
template <class T, int DPK>
__device__ __forceinline__ void row_frags(Frag<T> (&f)[DPK], const T* rowp, bool valid, int lane) {
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) { if (valid) frag_load(f[kk], rowp + kk * 32 + (lane >> 4) * 8); else frag_zero(f[kk]); }
}
__device__ __forceinline__ void frag_select(Frag<bf16_t>& f, bool keep) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.v = keep ? f.v : z; }
__device__ __forceinline__ void frag_select(Frag<float>& f, bool keep) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; f.lo = keep ? f.lo : z; f.hi = keep ? f.hi : z; }
// branch-free variant: always loads (row clamped into [0, nrows)), zeroes by select -> no control flow, so the
// compiler can hoist the loads of later key blocks above the MFMAs of earlier ones (memory-level parallelism)
template <class T, int DPK>
__device__ __forceinline__ void row_frags_nb(Frag<T> (&f)[DPK], const T* base, long long stride, int row, int nrows, int lane) {
    const bool ok = row >= 0 && row < nrows;
    const int r = row < 0 ? 0 : (row >= nrows ? nrows - 1 : row);
    const T* rowp = base + (long long)r * stride + (lane >> 4) * 8;
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) { frag_load(f[kk], rowp + kk * 32); frag_select(f[kk], ok); }
}
// 8 consecutive time steps t0..t0+7 of one row of a [..][Tp] transposed copy, zero beyond T
// branch-free variant (t0 is a multiple of 8, Tp a multiple of 8 and >= Tlen)
template <class T>
__device__ __forceinline__ void time_frag_nb(Frag<T>& f, const T* rowp, int t0, int Tlen, int Tp) {
    const int tc = t0 > Tp - 8 ? Tp - 8 : t0;
    frag_load(f, rowp + tc);
    int n = Tlen - t0; n = n < 0 ? 0 : (n > 8 ? 8 : n);
    frag_keep(f, tc == t0 ? n : 0);
}
template <class T>
__device__ __forceinline__ void time_frag(Frag<T>& f, const T* rowp, int t0, int Tlen) {
    if (t0 >= Tlen || t0 < 0) { frag_zero(f); return; }
    frag_load(f, rowp + t0);
    if (t0 + 8 > Tlen) frag_keep(f, Tlen - t0);
}

template <class T, int DPK>
__device__ __forceinline__ f32x4 dot_frags(const Frag<T> (&a)[DPK], const Frag<T> (&b)[DPK]) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) c = mma32(a[kk], b[kk], c);
    return c;
}

// relative -> absolute: this lane (column c, row group g) needs R[row][c - row + 15] of the 32-wide window (lo | hi)
__device__ __forceinline__ void skew_gather(const f32x4& lo, const f32x4& hi, int lane, float (&pos)[4]) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int cc = c - (g * 4 + reg) + 15;
        const int src = (cc & 15) + 16 * g;
        const float a = __shfl(lo[reg], src), b = __shfl(hi[reg], src);
        pos[reg] = cc < 16 ? a : b;
    }
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// logits of one 16(q) x 16(k) block in accumulator layout (row = (lane>>4)*4+reg, col = lane&15)
__device__ __forceinline__ void finish_logits(const f32x4& s, const float (&pos)[4], int q0, int k0, int lane, int Tlen, int D, float scale, float (&out)[4]) {
    const int c = lane & 15, g = lane >> 4, k = k0 + c;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int q = q0 + g * 4 + reg;
        int dlt = k - q; dlt = dlt < 0 ? -dlt : dlt;
        const float pl = dlt <= D - 1 ? pos[reg] : -1e8f;       // transformer.py:256-261
        out[reg] = k < Tlen ? s[reg] * scale + pl : -INFINITY;
    }
}

}  // namespace

// =========================================================================== forward
template <class T, int DPK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p)
{
    __shared__ __attribute__((aligned(16))) T ptile[4][16][PT_LD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    int bxi, h, b; attn_block_coord(p.gx, p.H, bxi, h, b);
    const int q0 = (bxi * 4 + w) * 16;
    const int Tn = p.T, D = p.D, dp = p.dp, H = p.H;
    const long long ldq = 3LL * H * dp;
    const T* Q = (const T*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const T* K = Q + H * dp;
    const T* VT = (const T*)p.qkvT + ((long long)b * 3 * H * dp + 2 * H * dp + h * dp) * p.Tp;
    const T* E = (const T*)p.E + (long long)h * (2 * D - 1) * dp;
    const bool tile_ok = q0 < Tn;

    int kstart = q0 - (D - 1); kstart = kstart < 0 ? 0 : kstart; kstart &= ~31;
    int kend = q0 + 16 + (D - 1); kend = kend > Tn ? Tn : kend;
    const int nb = tile_ok ? (kend - kstart + 15) / 16 : 0;
    const int m_org = kstart - q0 - 15 + (D - 1);

    Frag<T> qf[DPK];
    { int qr = q0 + c; qr = qr < Tn ? qr : Tn - 1; row_frags<T, DPK>(qf, Q + (long long)qr * ldq, tile_ok, lane); }

    float lg[NB_MAX][4];
    f32x4 rprev;
    { Frag<T> ef[DPK]; row_frags_nb<T, DPK>(ef, E, dp, m_org + c, 2 * D - 1, lane); rprev = dot_frags<T, DPK>(qf, ef); }
    // All NB_MAX key blocks are computed unconditionally and branch-free (blocks beyond the band are masked to -inf):
    // with no control flow between them the scheduler overlaps the fragment loads of later blocks with the MFMAs,
    // shuffles and softmax prologue of earlier ones.
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        const int k0 = kstart + 16 * j;
        Frag<T> kf[DPK], ef[DPK];
        row_frags_nb<T, DPK>(kf, K, ldq, k0 + c, Tn, lane);
        row_frags_nb<T, DPK>(ef, E, dp, m_org + 16 * (j + 1) + c, 2 * D - 1, lane);
        const f32x4 s = dot_frags<T, DPK>(qf, kf);
        const f32x4 rn = dot_frags<T, DPK>(qf, ef);
        float pos[4];
        skew_gather(rprev, rn, lane, pos);
        finish_logits(s, pos, q0, k0, lane, j < nb ? Tn : 0, D, p.scale, lg[j]);
        rprev = rn;
    }
    // ---- softmax over the band (row = g*4+reg lives on the 16 lanes of group g)
    float mx[4], sm[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NB_MAX; ++j) m = fmaxf(m, lg[j][reg]);
        mx[reg] = group16_max(m);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NB_MAX; ++j) { const float e = lg[j][reg] == -INFINITY ? 0.f : expf(lg[j][reg] - mx[reg]); lg[j][reg] = e; s += e; }
        sm[reg] = group16_sum(s);
    }
    if (tile_ok && c == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { const int q = q0 + g * 4 + reg; if (q < Tn) p.lse[((long long)b * H + h) * Tn + q] = mx[reg] + logf(sm[reg]); }
    }
    // ---- P~ (normalised, dropped-out) -> LDS in A-operand order (all NB_MAX blocks: zeros beyond the band)
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        bool kp[4] = {true, true, true, true};
        if (p.drop_thresh)   // probability (q, k) <-> Philox block ((bh*T + q/4)*T + k), word q & 3
            dropout_keep4(p.seed, p.stream, (((unsigned long long)b * H + h) * Tn + ((q0 >> 2) + g)) * Tn + (kstart + 16 * j + c), p.drop_thresh, kp);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float pv = sm[reg] > 0.f ? lg[j][reg] / sm[reg] : 0.f;
            if (p.drop_thresh) pv = kp[reg] ? pv * p.drop_scale : 0.f;
            stf(&ptile[w][g * 4 + reg][16 * j + c], pv);
        }
    }
    wave_lds_sync();
    // ---- O = P~ V  (B operand straight from the transposed copy of V); unconditional over the whole tile width
    f32x4 o[2 * DPK];
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; o[n] = z; }
#pragma unroll
    for (int kc = 0; kc < NB_MAX / 2; ++kc) {
        Frag<T> pa; frag_load(pa, &ptile[w][c][kc * 32 + g * 8]);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) {
            Frag<T> vb; time_frag_nb(vb, VT + (long long)(n * 16 + c) * p.Tp, kstart + kc * 32 + g * 8, Tn, p.Tp);
            o[n] = mma32(pa, vb, o[n]);
        }
    }
    if (tile_ok) {
        T* O = (T*)p.out + (long long)b * Tn * (H * dp) + h * dp;
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { const int q = q0 + g * 4 + reg; if (q < Tn) stf(O + (long long)q * (H * dp) + n * 16 + c, o[n][reg]); }
    }
}

// =========================================================================== backward helpers
// D[b,h,q] = sum_d dO . O : one wave per frame row (all heads), 16-byte chunks, per-head segment sums through LDS
template <class T>
__global__ __launch_bounds__(256) void attn_dsum_kernel(const T* __restrict__ dO, const T* __restrict__ O, float* __restrict__ Dv, int B, int H, int Tn, int dp)
{
    __shared__ float part[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int cph = dp >> 3, nchunk = H * cph;            // chunks per head, per row (<= 128: H*dp <= 1024)
    const long long rows = (long long)B * Tn;
    for (long long r = (long long)blockIdx.x * wpb + w; r < rows; r += (long long)gridDim.x * wpb) {
        for (int c = lane; c < nchunk; c += 64) {
            float a[8], o[8]; Vec8<T>::load(dO + r * (H * dp) + c * 8, a); Vec8<T>::load(O + r * (H * dp) + c * 8, o);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += a[e] * o[e];
            part[w][c] = s;
        }
        wave_lds_sync();
        if (lane < H) {
            float s = 0.f;
            for (int c = 0; c < cph; ++c) s += part[w][lane * cph + c];
            const int b = (int)(r / Tn), q = (int)(r - (long long)b * Tn);
            Dv[((long long)b * H + lane) * Tn + q] = s;
        }
        wave_lds_sync();
    }
}

// probability and dS of one block from recomputed logits:  p = exp(l - lse);  dS = p * (keep ? dP/(1-pd) : 0  -  D)
__device__ __forceinline__ void prob_ds(const float (&lgt)[4], const f32x4& dpv, const float (&lse)[4], const float (&dv)[4], const bool (&rowok)[4],
                                        const AttnP& p, int b, int h, int q0, int k0, int lane, float (&pd)[4], float (&ds)[4])
{
    const int c = lane & 15, g = lane >> 4;
    bool kp[4] = {true, true, true, true};
    if (p.drop_thresh) dropout_keep4(p.seed, p.stream, (((unsigned long long)b * p.H + h) * p.T + ((q0 >> 2) + g)) * p.T + (k0 + c), p.drop_thresh, kp);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float pr = (rowok[reg] && lgt[reg] != -INFINITY) ? expf(lgt[reg] - lse[reg]) : 0.f;
        float keep = 1.f;
        if (p.drop_thresh) keep = kp[reg] ? p.drop_scale : 0.f;
        pd[reg] = pr * keep;
        ds[reg] = pr * (dpv[reg] * keep - dv[reg]);
    }
}

// =========================================================================== backward: query-major (dQ)
template <class T, int DPK>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnP p)
{
    SS_DYN_SMEM(smem_raw);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    int bxi, h, b; attn_block_coord(p.gx, p.H, bxi, h, b);
    const int q0 = (bxi * (blockDim.x >> 6) + w) * 16;
    const int Tn = p.T, D = p.D, dp = p.dp, H = p.H, MPt = p.MPt;
    const int ldb = MPt + 8;
    T* tileA = (T*)smem_raw + (long long)w * 16 * (PT_LD + ldb);      // [16][PT_LD]  dS by key
    T* tileB = tileA + 16 * PT_LD;                                    // [16][ldb]    dS by relative position m
    const long long ldq = 3LL * H * dp;
    const T* Q = (const T*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const T* K = Q + H * dp;
    const T* V = Q + 2 * H * dp;
    const T* KT = (const T*)p.qkvT + ((long long)b * 3 * H * dp + H * dp + h * dp) * p.Tp;
    const T* E = (const T*)p.E + (long long)h * (2 * D - 1) * dp;
    const T* ET = (const T*)p.ET + (long long)h * dp * MPt;
    const T* dO = (const T*)p.dO + (long long)b * Tn * (H * dp) + h * dp;
    const bool tile_ok = q0 < Tn;

    int kstart = q0 - (D - 1); kstart = kstart < 0 ? 0 : kstart; kstart &= ~31;
    int kend = q0 + 16 + (D - 1); kend = kend > Tn ? Tn : kend;
    const int nb = tile_ok ? (kend - kstart + 15) / 16 : 0;
    const int nchunk = (nb + 1) / 2;
    const int m_org = kstart - q0 - 15 + (D - 1);

    // zero the relative-position tile
    for (int i = lane; i < 16 * ldb; i += 64) stf(tileB + i, 0.f);

    Frag<T> qf[DPK], dof[DPK];
    { int qr = q0 + c; qr = qr < Tn ? qr : Tn - 1;
      row_frags<T, DPK>(qf, Q + (long long)qr * ldq, tile_ok, lane);
      row_frags<T, DPK>(dof, dO + (long long)qr * (H * dp), tile_ok, lane); }
    float lse[4], dv[4]; bool rowok[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int q = q0 + g * 4 + reg; rowok[reg] = tile_ok && q < Tn;
        const long long si = ((long long)b * H + h) * Tn + (rowok[reg] ? q : 0);
        lse[reg] = p.lse[si]; dv[reg] = p.Dv[si];
    }
    wave_lds_sync();
    f32x4 rprev;
    { Frag<T> ef[DPK]; row_frags_nb<T, DPK>(ef, E, dp, m_org + c, 2 * D - 1, lane); rprev = dot_frags<T, DPK>(qf, ef); }
    // all NB_MAX key blocks, unconditional and branch-free (see attn_fwd_kernel); dS of blocks beyond the band is 0
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        const int k0 = kstart + 16 * j;
        float ds[4], pd[4], pos[4], lgt[4];
        Frag<T> kf[DPK], ef[DPK], vf[DPK];
        row_frags_nb<T, DPK>(kf, K, ldq, k0 + c, Tn, lane);
        row_frags_nb<T, DPK>(vf, V, ldq, k0 + c, Tn, lane);
        row_frags_nb<T, DPK>(ef, E, dp, m_org + 16 * (j + 1) + c, 2 * D - 1, lane);
        const f32x4 s = dot_frags<T, DPK>(qf, kf);
        const f32x4 rn = dot_frags<T, DPK>(qf, ef);
        const f32x4 dpv = dot_frags<T, DPK>(dof, vf);
        skew_gather(rprev, rn, lane, pos);
        finish_logits(s, pos, q0, k0, lane, j < nb ? Tn : 0, D, p.scale, lgt);
        prob_ds(lgt, dpv, lse, dv, rowok, p, b, h, q0, k0, lane, pd, ds);
        rprev = rn;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ql = g * 4 + reg;
            stf(tileA + ql * PT_LD + 16 * j + c, ds[reg]);
            const int m = k0 + c - (q0 + ql) + (D - 1);
            if (j < nb && m >= 0 && m <= 2 * D - 2) stf(tileB + ql * ldb + m, ds[reg]);
        }
    }
    wave_lds_sync();
    f32x4 acc[2 * DPK];
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[n] = z; }
#pragma unroll
    for (int kc = 0; kc < NB_MAX / 2; ++kc) {                               // content term: dS . K
        Frag<T> a; frag_load(a, tileA + c * PT_LD + kc * 32 + g * 8);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { Frag<T> kb; time_frag_nb(kb, KT + (long long)(n * 16 + c) * p.Tp, kstart + kc * 32 + g * 8, Tn, p.Tp); acc[n] = mma32(a, kb, acc[n]); }
    }
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) acc[n] = acc[n] * p.scale;
#pragma unroll
    for (int mc = 0; mc < 7; ++mc) {                                        // positional term: dR . E (unscaled Q); MPt <= 224
        const bool on = mc * 32 < MPt;
        const int mo = on ? mc * 32 : 0;
        Frag<T> a; frag_load(a, tileB + c * ldb + mo + g * 8); frag_select(a, on);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { Frag<T> eb; frag_load(eb, ET + (long long)(n * 16 + c) * MPt + mo + g * 8); acc[n] = mma32(a, eb, acc[n]); }
    }
    if (tile_ok) {
        T* dQ = (T*)p.dqkv + (long long)b * Tn * ldq + h * dp;
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { const int q = q0 + g * 4 + reg; if (q < Tn) stf(dQ + (long long)q * ldq + n * 16 + c, acc[n][reg]); }
    }
}

// =========================================================================== backward: key-major (dK, dV)
template <class T, int DPK>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnP p)
{
    __shared__ __attribute__((aligned(16))) T tP[4][16][40];     // [key][32 queries (+pad)]  P~^T
    __shared__ __attribute__((aligned(16))) T tS[4][16][40];     //                           dS^T
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    int bxi, h, b; attn_block_coord(p.gx, p.H, bxi, h, b);
    const int k0 = (bxi * 4 + w) * 16;
    const int Tn = p.T, D = p.D, dp = p.dp, H = p.H;
    const long long ldq = 3LL * H * dp;
    const T* Q = (const T*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const T* K = Q + H * dp;
    const T* V = Q + 2 * H * dp;
    const T* QT = (const T*)p.qkvT + ((long long)b * 3 * H * dp + h * dp) * p.Tp;
    const T* E = (const T*)p.E + (long long)h * (2 * D - 1) * dp;
    const T* dO = (const T*)p.dO + (long long)b * Tn * (H * dp) + h * dp;
    const T* dOT = (const T*)p.dOT + ((long long)b * H * dp + h * dp) * p.Tp;
    const bool tile_ok = k0 < Tn;

    int qstart = k0 - (D - 1); qstart = qstart < 0 ? 0 : qstart; qstart &= ~31;
    int qend = k0 + 16 + (D - 1); qend = qend > Tn ? Tn : qend;
    const int nqb = tile_ok ? (qend - qstart + 15) / 16 : 0;
    const int npair = (nqb + 1) / 2;

    Frag<T> kf[DPK], vf[DPK];
    { int kr = k0 + c; const bool ok = tile_ok && kr < Tn; kr = kr < Tn ? kr : Tn - 1;
      row_frags<T, DPK>(kf, K + (long long)kr * ldq, ok, lane);
      row_frags<T, DPK>(vf, V + (long long)kr * ldq, ok, lane); }
    f32x4 dk[2 * DPK], dvv[2 * DPK];
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dk[n] = z; dvv[n] = z; }

    // Each wave owns its P~^T / dS^T tiles, so the hand-off is wave-local (no workgroup barrier, waves run independently).
    // Both 16-query halves of a 32-query step are branch-free: loads clamped, contributions of out-of-band / out-of-range
    // queries are exact zeros, so the scheduler can overlap the second half's loads with the first half's MFMAs.
    for (int pr = 0; pr < npair; ++pr) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int jq = 2 * pr + half, qb0 = qstart + 16 * jq;
            float pd[4], ds[4], pos[4], lgt[4], lse[4], dv[4]; bool rowok[4];
            Frag<T> qf[DPK], dof[DPK], e0[DPK], e1[DPK];
            row_frags_nb<T, DPK>(qf, Q, ldq, qb0 + c, Tn, lane);
            row_frags_nb<T, DPK>(dof, dO, (long long)H * dp, qb0 + c, Tn, lane);
            const int m0 = k0 - qb0 - 15 + (D - 1);
            row_frags_nb<T, DPK>(e0, E, dp, m0 + c, 2 * D - 1, lane);
            row_frags_nb<T, DPK>(e1, E, dp, m0 + 16 + c, 2 * D - 1, lane);
            const f32x4 s = dot_frags<T, DPK>(qf, kf);
            const f32x4 rlo = dot_frags<T, DPK>(qf, e0), rhi = dot_frags<T, DPK>(qf, e1);
            const f32x4 dpv = dot_frags<T, DPK>(dof, vf);
            skew_gather(rlo, rhi, lane, pos);
            finish_logits(s, pos, qb0, k0, lane, Tn, D, p.scale, lgt);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int q = qb0 + g * 4 + reg; rowok[reg] = q < Tn && jq < nqb;
                const long long si = ((long long)b * H + h) * Tn + (q < Tn ? q : Tn - 1);
                lse[reg] = p.lse[si]; dv[reg] = p.Dv[si];
            }
            prob_ds(lgt, dpv, lse, dv, rowok, p, b, h, qb0, k0, lane, pd, ds);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const bool ok = rowok[reg];
                stf(&tP[w][c][half * 16 + g * 4 + reg], ok ? pd[reg] : 0.f); stf(&tS[w][c][half * 16 + g * 4 + reg], ok ? ds[reg] : 0.f);
            }
        }
        wave_lds_sync();
        {
            Frag<T> pa, sa; frag_load(pa, &tP[w][c][g * 8]); frag_load(sa, &tS[w][c][g * 8]);
            const int t0 = qstart + 32 * pr + g * 8;
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) {
                Frag<T> db, qb;
                time_frag_nb(db, dOT + (long long)(n * 16 + c) * p.Tp, t0, Tn, p.Tp);
                time_frag_nb(qb, QT + (long long)(n * 16 + c) * p.Tp, t0, Tn, p.Tp);
                dvv[n] = mma32(pa, db, dvv[n]);
                dk[n] = mma32(sa, qb, dk[n]);
            }
        }
        wave_lds_sync();
    }
    if (tile_ok) {
        T* dK = (T*)p.dqkv + (long long)b * Tn * ldq + H * dp + h * dp;
        T* dV = dK + H * dp;
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int k = k0 + g * 4 + reg;
                if (k < Tn) { stf(dK + (long long)k * ldq + n * 16 + c, dk[n][reg] * p.scale); stf(dV + (long long)k * ldq + n * 16 + c, dvv[n][reg]); }
            }
    }
}

// =========================================================================== host side
static int attn_check(const char* what, int dtype, int B, int H, int T, int Tp, int dp, int D, float dropout_p)
{
    SS_CHECK(dtype == SS_F32 || dtype == SS_BF16, "%s: bad dtype", what);
    SS_CHECK(B > 0 && H > 0 && T > 0, "%s: empty problem", what);
    SS_CHECK(dp % 32 == 0 && dp >= 32 && dp <= 128, "%s: padded head dim %d must be 32, 64, 96 or 128", what, dp);
    SS_CHECK(H <= 64 && H * dp <= 1024, "%s: H=%d heads x padded dim %d exceeds 1024 columns", what, H, dp);
    SS_CHECK(D >= 1 && D <= 100, "%s: relative_positional_distance %d not in [1,100]", what, D);
    SS_CHECK(Tp >= T && Tp % 8 == 0, "%s: Tp=%d must be a multiple of 8 and >= T", what, Tp);
    SS_CHECK(dropout_p >= 0.f && dropout_p < 1.f, "%s: dropout p out of range", what);
    return 0;
}

static void attn_fill(AttnP& p, int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream)
{
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.T = T; p.Tp = Tp; p.dp = dp; p.D = D; p.MPt = (2 * D - 1 + 31) / 32 * 32; p.scale = scale;
    if (dropout_p > 0.f) { p.drop_thresh = dropout_threshold(dropout_p); p.drop_scale = 1.f / (1.f - dropout_p); } else { p.drop_scale = 1.f; }
    p.seed = seed; p.stream = rng_stream;
}

#define SS_ATTN_DISPATCH(KERNEL, grid, BLK, smem)                                                              \
    do {                                                                                                        \
        const int dpk = dp / 32;                                                                                \
        if (dtype == SS_BF16) {                                                                                 \
            if (dpk == 1) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 1>), grid, dim3(BLK), smem, stream, p);            \
            else if (dpk == 2) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 2>), grid, dim3(BLK), smem, stream, p);       \
            else if (dpk == 3) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 3>), grid, dim3(BLK), smem, stream, p);       \
            else SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 4>), grid, dim3(BLK), smem, stream, p);                     \
        } else {                                                                                                \
            if (dpk == 1) SS_LAUNCH(SS_KERNEL(KERNEL<float, 1>), grid, dim3(BLK), smem, stream, p);             \
            else if (dpk == 2) SS_LAUNCH(SS_KERNEL(KERNEL<float, 2>), grid, dim3(BLK), smem, stream, p);        \
            else if (dpk == 3) SS_LAUNCH(SS_KERNEL(KERNEL<float, 3>), grid, dim3(BLK), smem, stream, p);        \
            else SS_LAUNCH(SS_KERNEL(KERNEL<float, 4>), grid, dim3(BLK), smem, stream, p);                      \
        }                                                                                                       \
    } while (0)

extern "C" int ss_relpos_attention_forward(int dtype, const void* qkv, const void* qkvT, const void* E, void* out, float* lse,
                                           int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_forward", dtype, B, H, T, Tp, dp, D, dropout_p)) return 1;
    SS_CHECK(qkv && qkvT && E && out && lse, "ss_relpos_attention_forward: null pointer");
    AttnP p; attn_fill(p, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream);
    p.qkv = qkv; p.qkvT = qkvT; p.E = E; p.out = out; p.lse = lse;
    p.gx = ((T + 15) / 16 + 3) / 4;
    dim3 grid(p.gx * H * B);
    SS_ATTN_DISPATCH(attn_fwd_kernel, grid, 256, 0);
    SS_LAUNCH_CHECK("ss_relpos_attention_forward");
    return 0;
}

extern "C" int ss_relpos_attention_backward(int dtype, const void* qkv, const void* qkvT, const void* E, const void* ET, const void* out, const float* lse,
                                            const void* dO, const void* dOT, float* Dscratch, void* dqkv,
                                            int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_backward", dtype, B, H, T, Tp, dp, D, dropout_p)) return 1;
    SS_CHECK(qkv && qkvT && E && ET && out && lse && dO && dOT && Dscratch && dqkv, "ss_relpos_attention_backward: null pointer");
    AttnP p; attn_fill(p, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream);
    p.qkv = qkv; p.qkvT = qkvT; p.E = E; p.ET = ET; p.out = (void*)out; p.lse = (float*)lse; p.dO = dO; p.dOT = dOT; p.Dv = Dscratch; p.dqkv = dqkv;
    {
        long long blocks = ((long long)B * T + 3) / 4; if (blocks > 8192) blocks = 8192;
        if (dtype == SS_BF16) SS_LAUNCH(attn_dsum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)dO, (const bf16_t*)out, Dscratch, B, H, T, dp);
        else SS_LAUNCH(attn_dsum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)dO, (const float*)out, Dscratch, B, H, T, dp);
    }
    const size_t esz = dtype == SS_BF16 ? 2 : 4;
    const int nwq = dtype == SS_BF16 ? 4 : 2;                         // keep the dynamic LDS request under 64 KiB
    const size_t smem_q = (size_t)nwq * 16 * (size_t)(PT_LD + p.MPt + 8) * esz;
    p.gx = ((T + 15) / 16 + nwq - 1) / nwq;
    dim3 gridq(p.gx * H * B);
    SS_ATTN_DISPATCH(attn_bwd_q_kernel, gridq, nwq * 64, smem_q);
    p.gx = ((T + 15) / 16 + 3) / 4;
    dim3 grid(p.gx * H * B);
    SS_ATTN_DISPATCH(attn_bwd_kv_kernel, grid, 256, 0);
    SS_LAUNCH_CHECK("ss_relpos_attention_backward");
    return 0;
}
