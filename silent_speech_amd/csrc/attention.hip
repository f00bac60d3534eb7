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
#include "attention_t.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

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
    void* pimg;     // saved probabilities of the resident kernels (see the P image below), or null
    int B, H, T, Tp, dp, D, MPt, gx;
    float scale;
    unsigned drop_thresh; float drop_scale; unsigned long long seed; unsigned stream;
    int h1;         // hand-scheduled kernels: half-workgroups dispatched FIRST (see res2_block)
    int persist;    // > 0: persistent schedule of the kernels that stage E -- this many workgroups per head, each walks several sequences of ITS head (see res2_item)
    int tail;       // hand-scheduled kernels: bytes behind the last table (chunk buffers / room for reads that run past the E table)
    int debug;      // SS_ATTN_DEBUG (measurement only): bit 0 skips the operand staging, bit 1 the tile loop of the hand-scheduled forward
};

constexpr int NB_MAX = 16;      // 16-key blocks per query tile: ceil((31 + 16 + 2*99)/16)
constexpr int PT_LD = 256 + 8;  // probability tile row length (elements)

template <class T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 lo, hi; };
// f32 storage, bf16 x 3 arithmetic (dtype SS_F32X3, per-tile kernels): a fragment is split into hi = bf16(x) and lo = bf16(x - hi) when
// it is loaded; a product is a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on three bf16 MFMAs (f32 accumulate), see gemm.hip: split_bf16x3.
struct x3_t;
template <> struct Frag<x3_t> { bf16x8 hi, lo; };
template <class MT> struct StorageOf { typedef MT type; };
template <> struct StorageOf<x3_t> { typedef float type; };

__device__ __forceinline__ void frag_zero(Frag<bf16_t>& f) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.v = z; }
__device__ __forceinline__ void frag_zero(Frag<float>& f) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; f.lo = z; f.hi = z; }
__device__ __forceinline__ void frag_load(Frag<bf16_t>& f, const bf16_t* p) { f.v = *(const bf16x8*)p; }
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) { f.lo = *(const f32x4*)p; f.hi = *(const f32x4*)(p + 4); }
__device__ __forceinline__ void frag_zero(Frag<x3_t>& f) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.hi = z; f.lo = z; }
__device__ __forceinline__ void frag_load(Frag<x3_t>& f, const float* p) {
    const f32x4 x0 = *(const f32x4*)p, x1 = *(const f32x4*)(p + 4);
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned hp = pack_bf16(v[2 * e], v[2 * e + 1]);
        h[e] = hp;
        l[e] = pack_bf16(v[2 * e] - __uint_as_float(hp << 16), v[2 * e + 1] - __uint_as_float(hp & 0xffff0000u));
    }
    f.hi = __builtin_bit_cast(bf16x8, h); f.lo = __builtin_bit_cast(bf16x8, l);
}
// keep only the first n (0..8) elements
__device__ __forceinline__ void frag_keep(Frag<bf16_t>& f, int n) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (e >= n) f.v[e] = 0;
}
__device__ __forceinline__ void frag_keep(Frag<float>& f, int n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (e >= n) f.lo[e] = 0.f; if (e + 4 >= n) f.hi[e] = 0.f; }
}
__device__ __forceinline__ void frag_keep(Frag<x3_t>& f, int n) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (e >= n) { f.hi[e] = 0; f.lo[e] = 0; }
}
__device__ __forceinline__ f32x4 mma32(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4 c) { return mfma_bf16_16x16x32(a.v, b.v, c); }
__device__ __forceinline__ f32x4 mma32(const Frag<x3_t>& a, const Frag<x3_t>& b, f32x4 c) {
    c = mfma_bf16_16x16x32(a.lo, b.hi, c);
    c = mfma_bf16_16x16x32(a.hi, b.lo, c);
    return mfma_bf16_16x16x32(a.hi, b.hi, c);
}
__device__ __forceinline__ f32x4 mma32(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = mfma_f32_16x16x4(a.lo[e], b.lo[e], c);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = mfma_f32_16x16x4(a.hi[e], b.hi[e], c);
    return c;
}

// fragments of one matrix row (8 consecutive d per lane per 32-deep step); zero if !valid
template <class T, int DPK, class F>
__device__ __forceinline__ void row_frags(F (&f)[DPK], const T* rowp, bool valid, int lane) {
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) { if (valid) frag_load(f[kk], rowp + kk * 32 + (lane >> 4) * 8); else frag_zero(f[kk]); }
}
__device__ __forceinline__ void frag_select(Frag<bf16_t>& f, bool keep) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.v = keep ? f.v : z; }
__device__ __forceinline__ void frag_select(Frag<float>& f, bool keep) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; f.lo = keep ? f.lo : z; f.hi = keep ? f.hi : z; }
__device__ __forceinline__ void frag_select(Frag<x3_t>& f, bool keep) { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; f.hi = keep ? f.hi : z; f.lo = keep ? f.lo : z; }
// branch-free variant: always loads (row clamped into [0, nrows)), zeroes by select -> no control flow, so the
// compiler can hoist the loads of later key blocks above the MFMAs of earlier ones (memory-level parallelism)
template <class T, int DPK, class F>
__device__ __forceinline__ void row_frags_nb(F (&f)[DPK], const T* base, long long stride, int row, int nrows, int lane) {
    const bool ok = row >= 0 && row < nrows;
    const int r = row < 0 ? 0 : (row >= nrows ? nrows - 1 : row);
    const T* rowp = base + (long long)r * stride + (lane >> 4) * 8;
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) { frag_load(f[kk], rowp + kk * 32); frag_select(f[kk], ok); }
}
// 8 consecutive time steps t0..t0+7 of one row of a [..][Tp] transposed copy, zero beyond T
// branch-free variant (t0 is a multiple of 8, Tp a multiple of 8 and >= Tlen)
template <class F, class T>
__device__ __forceinline__ void time_frag_nb(F& f, const T* rowp, int t0, int Tlen, int Tp) {
    const int tc = t0 > Tp - 8 ? Tp - 8 : t0;
    frag_load(f, rowp + tc);
    int n = Tlen - t0; n = n < 0 ? 0 : (n > 8 ? 8 : n);
    frag_keep(f, tc == t0 ? n : 0);
}
template <class F, class T>
__device__ __forceinline__ void time_frag(F& f, const T* rowp, int t0, int Tlen) {
    if (t0 >= Tlen || t0 < 0) { frag_zero(f); return; }
    frag_load(f, rowp + t0);
    if (t0 + 8 > Tlen) frag_keep(f, Tlen - t0);
}

template <class T, int DPK, class F>
__device__ __forceinline__ f32x4 dot_frags(const F (&a)[DPK], const F (&b)[DPK]) {
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
template <class MT, int DPK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p)
{
    typedef typename StorageOf<MT>::type T;
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

    Frag<MT> qf[DPK];
    { int qr = q0 + c; qr = qr < Tn ? qr : Tn - 1; row_frags<T, DPK>(qf, Q + (long long)qr * ldq, tile_ok, lane); }

    float lg[NB_MAX][4];
    f32x4 rprev;
    { Frag<MT> ef[DPK]; row_frags_nb<T, DPK>(ef, E, dp, m_org + c, 2 * D - 1, lane); rprev = dot_frags<T, DPK>(qf, ef); }
    // All NB_MAX key blocks are computed unconditionally and branch-free (blocks beyond the band are masked to -inf):
    // with no control flow between them the scheduler overlaps the fragment loads of later blocks with the MFMAs,
    // shuffles and softmax prologue of earlier ones.
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        const int k0 = kstart + 16 * j;
        Frag<MT> kf[DPK], ef[DPK];
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
        Frag<MT> pa; frag_load(pa, &ptile[w][c][kc * 32 + g * 8]);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) {
            Frag<MT> vb; time_frag_nb(vb, VT + (long long)(n * 16 + c) * p.Tp, kstart + kc * 32 + g * 8, Tn, p.Tp);
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
template <class MT, int DPK>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnP p)
{
    typedef typename StorageOf<MT>::type T;
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

    Frag<MT> qf[DPK], dof[DPK];
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
    { Frag<MT> ef[DPK]; row_frags_nb<T, DPK>(ef, E, dp, m_org + c, 2 * D - 1, lane); rprev = dot_frags<T, DPK>(qf, ef); }
    // all NB_MAX key blocks, unconditional and branch-free (see attn_fwd_kernel); dS of blocks beyond the band is 0
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
        const int k0 = kstart + 16 * j;
        float ds[4], pd[4], pos[4], lgt[4];
        Frag<MT> kf[DPK], ef[DPK], vf[DPK];
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
        Frag<MT> a; frag_load(a, tileA + c * PT_LD + kc * 32 + g * 8);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { Frag<MT> kb; time_frag_nb(kb, KT + (long long)(n * 16 + c) * p.Tp, kstart + kc * 32 + g * 8, Tn, p.Tp); acc[n] = mma32(a, kb, acc[n]); }
    }
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) acc[n] = acc[n] * p.scale;
#pragma unroll
    for (int mc = 0; mc < 7; ++mc) {                                        // positional term: dR . E (unscaled Q); MPt <= 224
        const bool on = mc * 32 < MPt;
        const int mo = on ? mc * 32 : 0;
        Frag<MT> a; frag_load(a, tileB + c * ldb + mo + g * 8); frag_select(a, on);
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { Frag<MT> eb; frag_load(eb, ET + (long long)(n * 16 + c) * MPt + mo + g * 8); acc[n] = mma32(a, eb, acc[n]); }
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
template <class MT, int DPK>
__global__ __launch_bounds__(256, DPK <= 3 ? 2 : 1) void attn_bwd_kv_kernel(AttnP p)      // <= 256 registers up to d_head 96: two waves per SIMD (the bf16 x 3 fragments would otherwise take 296)
{
    typedef typename StorageOf<MT>::type T;
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

    Frag<MT> kf[DPK], vf[DPK];
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
            Frag<MT> qf[DPK], dof[DPK], e0[DPK], e1[DPK];
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
            Frag<MT> pa, sa; frag_load(pa, &tP[w][c][g * 8]); frag_load(sa, &tS[w][c][g * 8]);
            const int t0 = qstart + 32 * pr + g * 8;
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) {
                Frag<MT> db, qb;
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

// The LDS-resident 16 x 16 family of rounds 1-4 below is compiled only into A/B builds (-DSS_ATTN_RES16: tools/measure_lib.sh): every shape it
// ran (bf16, T <= 208, d_head <= 96) runs the transposed-score kernels of attention_t.hip since round 5, so the product library carries no
// kernel that only an environment switch (SS_ATTN_T=0) could reach.
#if defined(SS_ATTN_RES16)
// =========================================================================== resident (whole sequence in LDS) kernels, bf16
// The training rows are T = 200 frames: the K/V (or Q/dO) rows of one (sequence, head) and the head's 2D-1 embedding rows
// fit in the 160 KB LDS of a CU.  The per-tile kernels above re-fetch those operands from L2 for every 16-row tile
// (1.7 GB of L2->CU traffic per forward launch at the reference batch: they are L2-bandwidth bound); here ONE workgroup of
// 8 waves owns a (sequence, head), stages the operands once (coalesced 16-byte copies, rows padded by 16 B so that every
// fragment read is bank-conflict free) and its waves pull 16-row tiles from an LDS counter, heaviest (band-centre) first.
// Contractions over the time axis take their B operand straight from the row-major tiles with ds_read_b64_tr_b16, so no
// transposed copies are staged; only key blocks that intersect the +-(D-1) band are computed.
namespace {
constexpr int RES_NB = 13, RT_LD = 40;                  // <= 13 key blocks (T <= 208), chunk-tile row length
constexpr int RES_W_FWD = 8, RES_W_BQ = 8, RES_W_BKV = 8;      // waves per workgroup (12 = 3 per SIMD was measured slower: the 168-register cap spills)
typedef bf16_t RT;

__device__ __forceinline__ bf16x8 lds16(const unsigned char* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ bf16x8 bzero8() { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
// A operand in the k-order of the transposing reads: lane group g holds contraction elements {4g..4g+3, 16+4g..16+4g+3}
__device__ __forceinline__ bf16x8 lds_a_tr(const RT* row, int g) {
    const s16x4 lo = *(const s16x4*)(row + 4 * g), hi = *(const s16x4*)(row + 16 + 4 * g);
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
// B operand B[k][n] = tile[row0 + k][col0 + n] (k = 32 contraction rows, n = 16 columns) of a row-major LDS tile
__device__ __forceinline__ bf16x8 lds_b_tr(const unsigned char* tile, int pitch, int row0, int colbyte0, int c, int g) {
    const unsigned char* a0 = tile + (row0 + g * 4 + (c >> 2)) * pitch + colbyte0 + (c & 3) * 8;
    const s16x4 lo = lds_read_tr16(a0), hi = lds_read_tr16(a0 + 16 * pitch);
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
template <int DPK>
__device__ __forceinline__ void lds_row_frags(bf16x8 (&f)[DPK], const unsigned char* tile, int pitch, int row, int g) {
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) f[kk] = lds16(tile + row * pitch + kk * 64 + g * 16);
}
// rows of the embedding table with per-lane clamp + select (standard, non-transposing reads)
template <int DPK>
__device__ __forceinline__ void lds_e_frags(bf16x8 (&f)[DPK], const unsigned char* Es, int pitch, int m, int nrows, int g) {
    const bool ok = m >= 0 && m < nrows;
    const int r = m < 0 ? 0 : (m >= nrows ? nrows - 1 : m);
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) { const bf16x8 v = lds16(Es + r * pitch + kk * 64 + g * 16); f[kk] = ok ? v : bzero8(); }
}
template <int DPK>
__device__ __forceinline__ f32x4 dot8(const bf16x8 (&a)[DPK], const bf16x8 (&b)[DPK]) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) c = mfma_bf16_16x16x32(a[kk], b[kk], c);
    return c;
}
template <int DPK>
__device__ __forceinline__ void glb_row_frags(bf16x8 (&f)[DPK], const RT* rowp, bool valid, int g) {
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) f[kk] = valid ? *(const bf16x8*)(rowp + kk * 32 + g * 8) : bzero8();
}
// cooperative copy of `rows` rows of dp elements (row stride ld) into an LDS tile; rows >= valid are zero-filled.
// Loads are issued 8 deep per thread before the first LDS store so that the copy is bandwidth- not latency-bound.
template <int DPK>
__device__ __forceinline__ void stage_rows(unsigned char* dst, int pitch, const RT* src, long long ld, int valid, int rows, int tid, int nthr) {
    constexpr int CPR = DPK * 4, U = 8;
    const int total = rows * CPR;
    for (int base = tid; base < total; base += nthr * U) {
        u32x4 v[U]; int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * nthr, r = i / CPR, ch = i - r * CPR;
            u32x4 z = {0u, 0u, 0u, 0u};
            v[u] = z; off[u] = i < total ? r * pitch + ch * 16 : -1;
            if (i < total && r < valid) v[u] = *(const u32x4*)(src + (long long)r * ld + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (off[u] >= 0) *(u32x4*)(dst + off[u]) = v[u];
    }
}
// accumulator tile (16 rows x DPK*32 columns, f32, scaled) -> global rows of `ld` elements through a 16 x 32 LDS slab:
// 3 wave-wide 16-byte stores per tile instead of 96 two-byte ones (which also kept vmcnt busy ahead of the next tile's loads)
template <int DPK>
__device__ __forceinline__ void store_tile_rows(RT* tile, const f32x4 (&acc)[2 * DPK], float scale, RT* dst, long long ld, int row0, int nrows, int lane) {
    const int c = lane & 15, g = lane >> 4, r = lane >> 2, ch = lane & 3;
#pragma unroll
    for (int sl = 0; sl < DPK; ++sl) {
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) tile[(g * 4 + reg) * RT_LD + nn * 16 + c] = f2bf(acc[2 * sl + nn][reg] * scale);
        wave_lds_sync();
        const u32x4 v = *(const u32x4*)(tile + r * RT_LD + ch * 8);
        if (row0 + r < nrows) *(u32x4*)(dst + (long long)(row0 + r) * ld + sl * 32 + ch * 8) = v;
        wave_lds_sync();
    }
}
// Dropout of the resident kernels: probability (q, k) of pair bh draws 16 bits of a hash of (row group q / 4, k); slot q & 3.
// The per-tile part (a full mix of the row group) is hoisted; a block costs one add, a short multiply-xorshift and one derived word.
__device__ __forceinline__ unsigned res_drop_key(const AttnP& p, int bh, int q0, int g) {
    const unsigned row = (unsigned)bh * (unsigned)p.T + (unsigned)((q0 >> 2) + g);
    const unsigned sd = (unsigned)p.seed ^ ((unsigned)(p.seed >> 32) * 0x9E3779B9u) ^ (p.stream * 0x85EBCA6Bu);
    return mix32(row * 0x9E3779B1u ^ sd) + sd;
}
// two 32-bit words of 16-bit draws for the 4 rows of a lane: word 0 = rows (1, 0), word 1 = rows (3, 2)
__device__ __forceinline__ void res_drop_words(unsigned key, int k, unsigned& a, unsigned& b) {
    a = (key + 2u * (unsigned)k) * 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    b = (a ^ 0x68E31DA4u) * 0x9E3779B1u; b ^= b >> 15;
}
// an entry is dropped iff the low 15 bits of its draw are below t15 = round(p * 2^15)  (p exact to 2^-15)
__device__ __forceinline__ void res_drop_keep4(unsigned key, int k, unsigned t15, bool (&keep)[4]) {
    unsigned a, b; res_drop_words(key, k, a, b);
    keep[0] = (a & 0x7fffu) >= t15; keep[1] = ((a >> 16) & 0x7fffu) >= t15; keep[2] = (b & 0x7fffu) >= t15; keep[3] = ((b >> 16) & 0x7fffu) >= t15;
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
// packed form: bit 15 of each half of the result is set iff that entry is dropped
__device__ __forceinline__ unsigned res_drop_sign2(unsigned draws, unsigned t15x2) {
    const s16x2 d = __builtin_bit_cast(s16x2, draws & 0x7fff7fffu) - __builtin_bit_cast(s16x2, t15x2);
    return __builtin_bit_cast(unsigned, d);
}
__device__ __forceinline__ unsigned res_sign_fill2(unsigned signs) {            // 0xffff in every half whose bit 15 is set
    const s16x2 m = __builtin_bit_cast(s16x2, signs) >> 15;
    return __builtin_bit_cast(unsigned, m);
}

// ---- the P image: what the resident forward leaves for the backward kernels.
// For every (pair, 16-query tile, 16-key block) the 64 lanes store the 4 probabilities they own (rows 4g..4g+3 of column c, the
// MFMA accumulator layout all three kernels compute in) as 4 bf16: normalised, BEFORE dropout, with bit 15 (the sign: P >= 0)
// set iff dropout removed the entry.  512 B per block, written and read as one 8-byte word per lane.  With it the backward
// kernels skip the recomputation of both logit products, the skew, the exponentials and the dropout draws (2/3 of their VALU
// work and 6 resp. 9 of their 15 resp. 18 MFMAs per block).  Slots per tile: nb + 5 (the fixed-size tile bodies of the forward
// also store the all-zero blocks just past the band).
__device__ __host__ __forceinline__ int pimg_slots(int nb) { return nb + 5; }
__device__ __forceinline__ u32x2* pimg_block(void* base, int pair, int nb, int tile, int block, int lane) {
    return (u32x2*)base + (((long long)pair * nb + tile) * pimg_slots(nb) + block) * 64 + lane;
}
// tile index of the i-th work item: centre of the sequence (full band, most key blocks) first
__device__ __forceinline__ int res_tile_of(int i, int nb) { const int mid = nb >> 1; return (i & 1) ? mid - ((i + 1) >> 1) : mid + (i >> 1); }
__device__ __forceinline__ int res_split_index(int i, int half) { return half < 0 ? i : 2 * i + half; }
__device__ __forceinline__ int res_next(int* ctr, int lane) {
    int i = 0; if (lane == 0) i = atomicAdd(ctr, 1);
    return wave_first(i);                            // scalar: the tile index drives uniform branches and LDS base addresses
}
}  // namespace

// ---- branch-free band bookkeeping.  Key blocks are indexed RELATIVE to the first in-band block jlo of the tile
// (block i <-> keys 16*(jlo+i) ..), a tile body handles a compile-time count NBLK of them (blocks i >= nblk are masked),
// so one tile is one basic block and the scheduler overlaps the LDS reads / MFMAs / shuffles of neighbouring blocks.
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f, MASKED2 = -1e8f * 1.4426950408889634f;
constexpr int RES_PL = 32;                                  // zero rows below / above the staged embedding table

// log2-domain logits of block (q0, k0): (s*scale + pos)*log2(e) inside the band and the sequence, -1e8*log2(e) elsewhere
// (transformer.py:256-261; keys >= T get the same treatment: exp2 makes both exactly 0 against any real logit)
__device__ __forceinline__ void res_logits(const f32x4& s, const float (&pos)[4], const int (&bandA)[4], int off, bool key_ok, unsigned span, float scale2, float (&out)[4]) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const bool in = (unsigned)(off + bandA[reg]) <= span && key_ok;        // 0 <= (k - q) + D-1 <= 2D-2
        out[reg] = in ? fmaf(s[reg], scale2, pos[reg] * LOG2E) : MASKED2;      // select AFTER the fma: masked blocks may have read LDS garbage
    }
}
template <int DPK>
__device__ __forceinline__ void lds_e_pad(bf16x8 (&f)[DPK], const unsigned char* Es, int pitch, int row, int last, int g) {
    row = row > last ? last : row;                                             // rows past the table are zero pad
#pragma unroll
    for (int kk = 0; kk < DPK; ++kk) f[kk] = lds16(Es + row * pitch + kk * 64 + g * 16);
}

template <int DPK, int NBLK, bool DROP>
__device__ __forceinline__ void fwd_res_tile(const AttnP& p, const unsigned char* Ks, const unsigned char* Es, const unsigned char* Vs, RT* Pt,
                                             const bf16x8 (&qf)[DPK], int b, int h, int q0, int jlo, int nblk, int lane, int ER, f32x4 (&o)[2 * DPK])
{
    constexpr int PK = DPK * 64 + 16, PTL = 20;
    const int c = lane & 15, g = lane >> 4, Tn = p.T, D = p.D, H = p.H;
    const float scale2 = p.scale * LOG2E;
    const int m_org = -q0 - 15 + (D - 1) + RES_PL;
    int bandA[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) bandA[reg] = c - (g * 4 + reg) + (D - 1);
    float lg[NBLK][4];
    f32x4 rprev;
    { bf16x8 ef[DPK]; lds_e_pad<DPK>(ef, Es, PK, m_org + 16 * jlo + c, ER - 1, g); rprev = dot8<DPK>(qf, ef); }
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int k0 = 16 * (jlo + i);
        bf16x8 kf[DPK], ef[DPK];
        lds_row_frags<DPK>(kf, Ks, PK, k0 + c, g);
        lds_e_pad<DPK>(ef, Es, PK, m_org + k0 + 16 + c, ER - 1, g);
        const f32x4 s = dot8<DPK>(qf, kf);
        const f32x4 rn = dot8<DPK>(qf, ef);
        float pos[4];
        skew_gather(rprev, rn, lane, pos);
        res_logits(s, pos, bandA, k0 - q0, i < nblk && k0 + c < Tn, 2u * (unsigned)(D - 1), scale2, lg[i]);
        rprev = rn;
    }
    float inv[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float m = lg[0][reg];
#pragma unroll
        for (int i = 1; i < NBLK; ++i) m = fmaxf(m, lg[i][reg]);
        m = group16_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NBLK; ++i) { const float e = fast_exp2(lg[i][reg] - m); lg[i][reg] = e; sum += e; }
        sum = group16_sum(sum);
        inv[reg] = fast_rcp(sum) * (DROP ? p.drop_scale : 1.f);
        if (c == 0) { const int q = q0 + g * 4 + reg; if (q < Tn) p.lse[((long long)b * H + h) * Tn + q] = m * LN2 + logf(sum); }
    }
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; o[n] = z; }
    // O = P~ V per 32-key chunk: P~^T goes to LDS as [key][query] (one 8-byte store per block: this lane's 4 rows are adjacent),
    // both operands come back through transposing reads (V stays row-major: no transposed copy of qkv is needed)
#pragma unroll
    for (int kc = 0; kc < (NBLK + 1) / 2; ++kc) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int i = 2 * kc + half;
            float pv[4] = {0.f, 0.f, 0.f, 0.f};
            if (i < NBLK) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) pv[reg] = lg[i < NBLK ? i : 0][reg] * inv[reg];
                if (DROP) {
                    bool kp[4];
                    res_drop_keep4(res_drop_key(p, b * H + h, q0, g), 16 * (jlo + i) + c, p.drop_thresh >> 17, kp);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) pv[reg] = kp[reg] ? pv[reg] : 0.f;
                }
            }
            u32x2 pk = {pack_bf16(pv[0], pv[1]), pack_bf16(pv[2], pv[3])};
            *(u32x2*)(Pt + (16 * half + c) * PTL + g * 4) = pk;
        }
        wave_lds_sync();
        const bf16x8 pa = lds_b_tr((const unsigned char*)Pt, PTL * 2, 0, 0, c, g);
        int vrow = 16 * jlo + kc * 32; vrow = 2 * kc < nblk ? vrow : 0;          // chunks past the band hold P = 0: keep their reads on staged (finite) rows
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) o[n] = mfma_bf16_16x16x32(pa, lds_b_tr(Vs, PK, vrow, n * 32, c, g), o[n]);
        wave_lds_sync();
    }
}

template <int DPK, bool DROP>
__global__ __launch_bounds__(RES_W_FWD * 64) void attn_fwd_res_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    // workgroups >= p.gx are HALVES of the (sequence, head) pairs that would otherwise form a short last round: each stages the
    // operands itself and takes every second tile of the heaviest-first order (res_split_index)
    const int bid = blockIdx.x, pair = bid < p.gx ? bid : p.gx + ((bid - p.gx) >> 1), half = bid < p.gx ? -1 : ((bid - p.gx) & 1);
    const int H = p.H, h = pair % H, b = pair / H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4, Tr = nb * 16, NE = 2 * D - 1, ER = NE + 2 * RES_PL;
    unsigned char* Ks = (unsigned char*)smem;
    unsigned char* Vs = Ks + Tr * PK;                                   // a chunk starting at an odd block reads up to 16 rows past Tr: they
    unsigned char* Es = Vs + Tr * PK;                                   // land in the (finite) embedding rows and meet P = 0
    RT* Pt = (RT*)(Es + ER * PK) + w * 16 * RT_LD;                      // row m + RES_PL of Es holds embedding m
    int* ctr = (int*)(Es + ER * PK + RES_W_FWD * 16 * RT_LD * 2);
    const long long ldq = 3LL * H * dp;
    const RT* Q = (const RT*)p.qkv + (long long)b * Tn * ldq + h * dp;
    {
        stage_rows<DPK>(Ks, PK, Q + H * dp, ldq, Tn, Tr, tid, RES_W_FWD * 64);
        stage_rows<DPK>(Vs, PK, Q + 2 * H * dp, ldq, Tn, Tr, tid, RES_W_FWD * 64);
        stage_rows<DPK>(Es, PK, (const RT*)p.E, dp, 0, RES_PL, tid, RES_W_FWD * 64);
        stage_rows<DPK>(Es + RES_PL * PK, PK, (const RT*)p.E + (long long)h * NE * dp, dp, NE, NE + RES_PL, tid, RES_W_FWD * 64);
        if (tid == 0) *ctr = 0;
    }
    __syncthreads();
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 qf[DPK], qn[DPK];
    if (it < nb) { int qr = res_tile_of(it, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(qf, Q + (long long)qr * ldq, true, g); }
    while (it < nb) {
        const int q0 = res_tile_of(it, nb) * 16;
        int jlo = q0 - (D - 1); jlo = jlo < 0 ? 0 : jlo >> 4;
        int jhi = (q0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int nblk = jhi - jlo + 1;
        // next tile's Q rows are requested now: loads issued before this tile's stores never wait for them
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int qr = res_tile_of(itn, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(qn, Q + (long long)qr * ldq, true, g); }
        f32x4 o[2 * DPK];
        if (nblk <= 4) fwd_res_tile<DPK, 4, DROP>(p, Ks, Es, Vs, Pt, qf, b, h, q0, jlo, nblk, lane, ER, o);
        else if (nblk <= 10) fwd_res_tile<DPK, 10, DROP>(p, Ks, Es, Vs, Pt, qf, b, h, q0, jlo, nblk, lane, ER, o);
        else fwd_res_tile<DPK, RES_NB, DROP>(p, Ks, Es, Vs, Pt, qf, b, h, q0, jlo, nblk, lane, ER, o);
        store_tile_rows<DPK>(Pt, o, 1.f, (RT*)p.out + (long long)b * Tn * (H * dp) + h * dp, (long long)H * dp, q0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) qf[kk] = qn[kk];
    }
}

// =========================================================================== resident forward, hand-scheduled
// Same data flow as attn_fwd_res_kernel (one workgroup per (sequence, head), K / V / E rows resident in LDS, one wave per 16-row
// tile).  What changes is WHO schedules the tile: hipcc placed every fragment read directly in front of its MFMA and waited for it
// there (87 s_waitcnt in the logits block of one tile, each exposing the LDS latency to a wave that has only one partner on its
// SIMD), which left the kernel at 9 % of the MFMA rate.  Here
//   * every LDS access of the tile body is issued from inline asm with compile-time offsets off four per-tile base addresses
//     (no address arithmetic per block), one block AHEAD of its use, and ONE s_waitcnt per block closes them;
//   * the blocks form a static software pipeline: fragment reads of block i+1 | relative->absolute skew of block i-1 |
//     MFMAs of block i | logit arithmetic of block i-2, then for the P~V product: P~ of chunk c+2 | reads of chunk c+1 | MFMAs of c;
//   * the skew takes ONE ds_bpermute per value (the SOURCE lane knows which half of the 32-wide window its reader wants);
//     band and sequence limits are one add + compare against per-row constants; row maxima / sums are DPP reductions;
//   * dropout is packed 16-bit arithmetic on the bf16 pairs (sign of draw - threshold -> and-mask), its scale moves to the outputs;
//   * the normalised probabilities are also stored for the backward pass (the P image, above).
// LDS: [V rows | K rows | E rows | per-wave P~ chunk buffers].  Reads that run past a table (key blocks beyond the band, relative
// positions outside [0, 2D-2]) land in the NEXT region: for K and E they only feed logits that the band test replaces, and V rows
// past the sequence are met by P~ = 0 and hold finite K values.
namespace {
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ void afence() { sched_fence(); }
template <int OFF, class V16>
__device__ __forceinline__ void ard128(V16& d, const unsigned char* lds, unsigned a) {
    static_assert(sizeof(V16) == 16, "128-bit destination");
#if defined(SS_EMU)
    d = *(const V16*)(lds + a + OFF);
#else
    (void)lds;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void ard64tr(s16x4& d, const unsigned char* lds, unsigned a) {
#if defined(SS_EMU)
    d = lds_read_tr16(lds + a + OFF);
#else
    (void)lds;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void ard64(s16x4& d, const unsigned char* lds, unsigned a) {
#if defined(SS_EMU)
    d = *(const s16x4*)(lds + a + OFF);
#else
    (void)lds;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void awr64(unsigned char* lds, unsigned a, u32x2 v) {
#if defined(SS_EMU)
    *(u32x2*)(lds + a + OFF) = v;
#else
    (void)lds;
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(v), "n"(OFF) : "memory");
#endif
}
__device__ __forceinline__ void abperm(float& d, unsigned src_byte, float v) {
#if defined(SS_EMU)
    d = __shfl(v, (int)(src_byte >> 2));
#else
    asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(d) : "v"(src_byte), "v"(v));
#endif
}
// every LDS operation this wave has issued is complete (and, on the emulator, visible to its other lanes)
__device__ __forceinline__ void await0() {
#if defined(SS_EMU)
    hipemu::sync_wave();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// all but the N most recently issued LDS operations are complete
template <int N> __device__ __forceinline__ void await_but() {
#if defined(SS_EMU)
    hipemu::sync_wave();
#else
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
#endif
}
// ties later uses of x to this point of the asm stream (registers written by the asm reads above are not read before the wait)
template <class T> __device__ __forceinline__ void apin(T& x) { pin_vgpr(x); }
__device__ __forceinline__ bf16x8 join8(const s16x4& lo, const s16x4& hi) { bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; return f; }

#if defined(SS_EMU)
__device__ __forceinline__ float row16_max(float v) { return group16_max(v); }
__device__ __forceinline__ float row16_sum(float v) { return group16_sum(v); }
#else
#define SS_DPP_F(v, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, SS_DPP_F(v, 0xB1)); v = fmaxf(v, SS_DPP_F(v, 0x4E)); v = fmaxf(v, SS_DPP_F(v, 0x141)); v = fmaxf(v, SS_DPP_F(v, 0x140));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += SS_DPP_F(v, 0xB1); v += SS_DPP_F(v, 0x4E); v += SS_DPP_F(v, 0x141); v += SS_DPP_F(v, 0x140);
    return v;
}
#endif

// Row pitch of the resident tables of the hand-scheduled kernels: dp*2 + 32 bytes (224 at dp = 96).  The transposing reads of a 32-lane
// group touch 8 rows x 32 bytes -- conflict-free iff the pitch is an odd multiple of 32 bytes -- and the 16-byte fragment reads of a
// 16-lane service group stay conflict-free as well (even slots for its g = 0 lanes, odd for g = 1).  With the 208-byte pitch of the
// compiler-scheduled kernels 42 % of the LDS cycles of the key-major kernel were bank conflicts (SQ_LDS_BANK_CONFLICT 7.0e6 -> 1.9e6).
#define RES2_PAD 32
// Workgroup -> (pair, half).  One workgroup occupies a CU and all of them stage their operands at the same moment: a bandwidth-bound
// burst (115 KB x 256 CUs, 4.9 TB/s) during which nothing computes, followed by a compute phase during which HBM idles, round after round.
// The pairs beyond the last full round are split into two half-workgroups anyway; dispatching p.h1 of those halves FIRST makes half of
// the CUs finish their first item after ~0.6 of a round, and from then on the two populations stage in each other's compute phases.
__device__ __forceinline__ void res2_block(const AttnP& p, int& pair, int& half) {
    const int bid = blockIdx.x;
    if (bid < p.h1) { pair = p.gx + (bid >> 1); half = bid & 1; }
    else if (bid < p.h1 + p.gx) { pair = bid - p.h1; half = -1; }
    else { const int hb = bid - p.gx; pair = p.gx + (hb >> 1); half = hb & 1; }
}
// The launch parameters re-read from the kernel-argument segment (AttnP is these kernels' only argument).  What the item loop needs to find its next
// sequence (B, H, the schedule, the operand base pointers) would otherwise stay in scalar registers across the hand-scheduled tile loop, which has none
// to spare: 48-62 scalars spilled to VGPR lanes and 26 more VGPRs to scratch, + 4 / + 17 us per launch.  An s_load per field and item costs nothing.
// an opaque copy of a per-thread value: what is computed from it inside a loop stays inside (staging addresses hoisted out of the item loop
// would live through the tile loop, which has no registers to spare)
__device__ __forceinline__ int opaque_v(int x) { pin_vgpr(x); return x; }
#if defined(SS_EMU)
typedef const AttnP* AttnArgs;
__device__ __forceinline__ AttnArgs fresh_args(const AttnP& p) { return &p; }
#else
typedef const __attribute__((address_space(4))) AttnP* AttnArgs;
__device__ __forceinline__ AttnArgs fresh_args(const AttnP&) {
    AttnArgs k = (AttnArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));                     // opaque: the loads below cannot be hoisted out of the item loop and kept alive
    return k;
}
#endif
// Persistent schedule (forward, query-major backward: the kernels that keep the embedding table E of a HEAD in LDS).  The grid is H x S workgroups,
// one per CU: workgroup (h, u) stages E once and walks the sequences b = u, u + S, ... of head h, restaging only K and V -- E is 46 of the 115 KB a
// (sequence, head) pair needs, and with H = 8 all workgroups of a head sit on one XCD (blocks are dealt round-robin), so its E stays in that L2.
// The B % S sequences of the last, partial round are split into half-pairs as before (two workgroups share one: each stages K / V and takes every
// second tile); the first S / 2 workgroups of a head run their half FIRST, the others last, which keeps the two populations half a round apart
// (they stage in each other's compute phases, see res2_block).  k = 0, 1, ...: the k-th item of this workgroup; false when it has none left.
template <class A>
__device__ __forceinline__ bool res2_item(A a, int k, int& pair, int& half) {
    const int S = a->persist;
    if (S <= 0) {
        if (k > 0) return false;
        const int bid = blockIdx.x, h1 = a->h1, gx = a->gx;                     // res2_block
        if (bid < h1) { pair = gx + (bid >> 1); half = bid & 1; }
        else if (bid < h1 + gx) { pair = bid - h1; half = -1; }
        else { const int hb = bid - gx; pair = gx + (hb >> 1); half = hb & 1; }
        return true;
    }
    const int H = a->H, B = a->B, h = (int)blockIdx.x % H, u = (int)blockIdx.x / H;
    const int r = B % S, nsplit = (B > S && 2 * r <= S) ? r : 0;                // sequences of the partial round that are split in halves
    const int Bfull = B - nsplit;                                               // sequences taken whole: b = u, u + S, ... < Bfull
    const int nfull = u < Bfull ? (Bfull - u + S - 1) / S : 0;
    const bool has_half = u < 2 * nsplit, half_first = has_half && u < S / 2;
    int kk = k;
    if (half_first) { if (kk == 0) { pair = (Bfull + (u >> 1)) * H + h; half = u & 1; return true; } --kk; }
    if (kk < nfull) { pair = (u + kk * S) * H + h; half = -1; return true; }
    if (has_half && !half_first && kk == nfull) { pair = (Bfull + (u >> 1)) * H + h; half = u & 1; return true; }
    return false;
}
constexpr int F2_PTB = 32 * 20 * 2;       // bytes of one P~ chunk buffer: [32 keys][16 queries + 4] bf16
constexpr float MASKED_NAT = -1e8f;       // transformer.py:256-261

template <int DPK, int NBLK, bool DROP>
__device__ __forceinline__ void fwd2_tile(const AttnP& p, unsigned char* lds, unsigned kaddr, unsigned eaddr, unsigned vaddr, unsigned ptw, unsigned ptr_,
                                          const bf16x8 (&qf)[DPK], const unsigned (&srcb)[4], const bool (&sel)[4],
                                          int bh, int q0, int jlo, int lane, f32x4 (&o)[2 * DPK], float (&lse)[4])
{
    constexpr int PK = DPK * 64 + RES2_PAD, BLK = 16 * PK, NCH = (NBLK + 1) / 2;
    const int c = lane & 15, g = lane >> 4, Tn = p.T, D = p.D;
    const float scale = p.scale;
    int t0[4]; unsigned lim[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int r = g * 4 + reg;
        int l = Tn - 1 - (q0 + r) + (D - 1); l = l < 2 * (D - 1) ? l : 2 * (D - 1);
        // relative position (+ D-1) of this lane's column in block 0; rows past the sequence fail every test (their P is 0 in the image)
        t0[reg] = q0 + r < Tn ? 16 * jlo - q0 + c - r + (D - 1) : (1 << 30);
        lim[reg] = (unsigned)(l < 0 ? 0 : l);
    }
    float lg[NBLK][4], mrun[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    bf16x8 F[2][2 * DPK];
    f32x4 sacc[3], racc[3];
    float bp[2][4];

    auto reads = [&](auto ic) {                                               // K rows of block i, E rows of window block i+1
        constexpr int i = ic;
        sfor<0, DPK>([&](auto kk) { ard128<i * BLK + kk * 64>(F[i & 1][kk], lds, kaddr); ard128<(i + 1) * BLK + kk * 64>(F[i & 1][DPK + kk], lds, eaddr); });
    };
    auto mfmas = [&](auto ic) {
        constexpr int i = ic;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) { s = mfma_bf16_16x16x32(qf[kk], F[i & 1][kk], s); r = mfma_bf16_16x16x32(qf[kk], F[i & 1][DPK + kk], r); }
        sacc[i % 3] = s; racc[(i + 1) % 3] = r;
    };
    auto skew_issue = [&](auto jc) {
        constexpr int j = jc;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) abperm(bp[j & 1][reg], srcb[reg], sel[reg] ? racc[j % 3][reg] : racc[(j + 1) % 3][reg]);
    };
    auto logits = [&](auto jc) {
        constexpr int j = jc;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float l = fmaf(sacc[j % 3][reg], scale, bp[j & 1][reg]);
            const bool in = (unsigned)(t0[reg] + 16 * j) <= lim[reg];
            const float v = in ? l : MASKED_NAT;                              // select AFTER the fma: masked entries may come from rows of another table
            lg[j][reg] = v; mrun[reg] = fmaxf(mrun[reg], v);
        }
    };
    // ---- logits
    sfor<0, DPK>([&](auto kk) { ard128<kk * 64>(F[1][DPK + kk], lds, eaddr); });
    reads(std::integral_constant<int, 0>{});
    await0();
    sfor<0, 2 * DPK>([&](auto x) { apin(F[0][x]); });
    sfor<0, DPK>([&](auto kk) { apin(F[1][DPK + kk]); });
    {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) r = mfma_bf16_16x16x32(qf[kk], F[1][DPK + kk], r);
        racc[0] = r;
    }
    afence();
    sfor<0, NBLK + 2>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (i + 1 < NBLK) reads(std::integral_constant<int, i + 1>{});
        if constexpr (i >= 1 && i - 1 < NBLK) skew_issue(std::integral_constant<int, i - 1>{});
        afence();
        if constexpr (i < NBLK) mfmas(ic);
        if constexpr (i >= 2) logits(std::integral_constant<int, i - 2>{});
        afence();
        await0();
        if constexpr (i + 1 < NBLK) sfor<0, 2 * DPK>([&](auto x) { apin(F[(i + 1) & 1][x]); });
        if constexpr (i >= 1 && i - 1 < NBLK) sfor<0, 4>([&](auto x) { apin(bp[(i - 1) & 1][x]); });
        afence();
    });
    // ---- probabilities (normalised, f32 in place), their image for the backward pass, and O = P~ V
    float nm2[4], sum[4] = {0.f, 0.f, 0.f, 0.f}, inv[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) { mrun[reg] = row16_max(mrun[reg]); nm2[reg] = -mrun[reg] * LOG2E; }
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { const float e = fast_exp2(fmaf(lg[j][reg], LOG2E, nm2[reg])); lg[j][reg] = e; sum[reg] += e; }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {                                        // rows past the sequence: P = 0 (their image entries are read by the key-major backward)
        const float sm = row16_sum(sum[reg]); inv[reg] = q0 + g * 4 + reg < Tn ? fast_rcp(sm) : 0.f; lse[reg] = mrun[reg] + logf(sm);
    }
    unsigned dkey = 0;
    if (DROP) dkey = res_drop_key(p, bh, q0, g);
    const unsigned t15 = p.drop_thresh >> 17, t15x2 = t15 | (t15 << 16);
    u32x2* img = p.pimg ? pimg_block(p.pimg, bh, (Tn + 15) >> 4, q0 >> 4, jlo, lane) : (u32x2*)0;
    s16x4 palo[2], pahi[2], vlo[2][2 * DPK], vhi[2][2 * DPK];
    auto pchunk = [&](auto kcc) {
        constexpr int kc = kcc;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int j = 2 * kc + half;
            u32x2 pk = {0u, 0u};
            if (j < NBLK) {
                const int jj = j < NBLK ? j : 0;
                pk[0] = pack_bf16(lg[jj][0] * inv[0], lg[jj][1] * inv[1]); pk[1] = pack_bf16(lg[jj][2] * inv[2], lg[jj][3] * inv[3]);
                u32x2 im = pk;
                if (DROP) {
                    unsigned a, b; res_drop_words(dkey, 16 * (jlo + j) + c, a, b);
                    const unsigned d0 = res_drop_sign2(a, t15x2), d1 = res_drop_sign2(b, t15x2);
                    im[0] = (d0 & 0x80008000u) | pk[0]; im[1] = (d1 & 0x80008000u) | pk[1];
                    pk[0] &= ~res_sign_fill2(d0); pk[1] &= ~res_sign_fill2(d1);
                }
                if (img) img[j * 64] = im;
            }
            if (half == 0) awr64<(kc & 1) * F2_PTB>(lds, ptw, pk); else awr64<(kc & 1) * F2_PTB + 16 * 40>(lds, ptw, pk);
        }
    };
    auto chunk_reads = [&](auto kcc) {
        constexpr int kc = kcc;
        ard64tr<(kc & 1) * F2_PTB>(palo[kc & 1], lds, ptr_); ard64tr<(kc & 1) * F2_PTB + 16 * 40>(pahi[kc & 1], lds, ptr_);
        sfor<0, 2 * DPK>([&](auto n) { ard64tr<kc * 32 * PK + n * 32>(vlo[kc & 1][n], lds, vaddr); ard64tr<kc * 32 * PK + 16 * PK + n * 32>(vhi[kc & 1][n], lds, vaddr); });
    };
    auto chunk_pin = [&](auto kcc) {
        constexpr int kc = kcc;
        apin(palo[kc & 1]); apin(pahi[kc & 1]);
        sfor<0, 2 * DPK>([&](auto n) { apin(vlo[kc & 1][n]); apin(vhi[kc & 1][n]); });
    };
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; o[n] = z; }
    pchunk(std::integral_constant<int, 0>{});
    if constexpr (NCH > 1) pchunk(std::integral_constant<int, 1>{});
    afence();
    await0();                                                                  // emulator: the chunk-0 stores are visible to the other lanes
    chunk_reads(std::integral_constant<int, 0>{});
    await0();
    chunk_pin(std::integral_constant<int, 0>{});
    afence();
    sfor<0, NCH>([&](auto kcc) {
        constexpr int kc = kcc;
        if constexpr (kc + 1 < NCH) chunk_reads(std::integral_constant<int, kc + 1>{});
        afence();
        {
            const bf16x8 pa = join8(palo[kc & 1], pahi[kc & 1]);
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) o[n] = mfma_bf16_16x16x32(pa, join8(vlo[kc & 1][n], vhi[kc & 1][n]), o[n]);
        }
        if constexpr (kc + 2 < NCH) pchunk(std::integral_constant<int, kc + 2>{});       // into the buffer chunk kc was read from (those reads completed last step)
        afence();
        await0();
        if constexpr (kc + 1 < NCH) chunk_pin(std::integral_constant<int, kc + 1>{});
        afence();
    });
    if (DROP) {
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) o[n] = o[n] * p.drop_scale;
    }
}
}  // namespace

template <int DPK, bool DROP>
__global__ __launch_bounds__(RES_W_FWD * 64) void attn_fwd_res2_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + RES2_PAD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int H = p.H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4, Tr = nb * 16, NE = 2 * D - 1;
    unsigned char* lds = (unsigned char*)smem;
    const unsigned VS = 0, KS = VS + Tr * PK, ES = KS + Tr * PK, PT = ES + NE * PK, CT = PT + (unsigned)p.tail;      // tail: chunk buffers, and room for reads past the E table
    int* ctr = (int*)(lds + CT);
    const long long ldq = 3LL * H * dp;
    const unsigned lbase = lds_byte_address(lds);
    int pair, half;
    for (int item = 0;; ++item) {                                           // one (sequence, head) pair, or the sequences of this workgroup's head (res2_item)
    const AttnArgs a = fresh_args(p);
    if (!res2_item(a, item, pair, half)) break;
    const int h = pair % H, b = pair / H;
    const RT* Q = (const RT*)a->qkv + (long long)b * Tn * ldq + h * dp;
    if (item > 0) __syncthreads();                                          // every wave has left the tile loop of the previous sequence: K, V and the counter are free
    if (!(p.debug & 1)) {
        const int ts = opaque_v(tid);
        stage_rows<DPK>(lds + KS, PK, Q + H * dp, ldq, Tn, Tr, ts, RES_W_FWD * 64);
        stage_rows<DPK>(lds + VS, PK, Q + 2 * H * dp, ldq, Tn, Tr, ts, RES_W_FWD * 64);
        if (item == 0) stage_rows<DPK>(lds + ES, PK, (const RT*)a->E + (long long)h * NE * dp, dp, NE, NE, ts, RES_W_FWD * 64);      // the head's table: once per workgroup
    }
    if (tid == 0) *ctr = (p.debug & 2) ? nb : 0;
    __syncthreads();
    const int lane = opaque_v(tid) & 63, c = lane & 15, g = lane >> 4;       // the tile loop's lane constants are formed per item: nothing of them lives through the staging
    unsigned srcb[4]; bool sel[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) { const int r = g * 4 + reg; srcb[reg] = (unsigned)((((c - r + 15) & 15) + 16 * g) * 4); sel[reg] = c + r >= 15; }
    const unsigned ptw = lbase + PT + w * 2 * F2_PTB + (c * 20 + g * 4) * 2, ptr_ = lbase + PT + w * 2 * F2_PTB + (g * 4 + (c >> 2)) * 40 + (c & 3) * 8;
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 qf[DPK], qn[DPK];
    if (it < nb) { int qr = res_tile_of(it, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(qf, Q + (long long)qr * ldq, true, g); }
    while (it < nb) {
        const int q0 = res_tile_of(it, nb) * 16;
        int jlo = q0 - (D - 1); jlo = jlo < 0 ? 0 : jlo >> 4;
        int jhi = (q0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int nblk = jhi - jlo + 1;
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int qr = res_tile_of(itn, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(qn, Q + (long long)qr * ldq, true, g); }
        const unsigned kaddr = lbase + KS + (16 * jlo + c) * PK + g * 16;
        const unsigned eaddr = lbase + ES + (unsigned)((16 * jlo - q0 - 15 + (D - 1) + c) * PK) + g * 16;        // rows below the table: the last K rows (band test)
        const unsigned vaddr = lbase + VS + (16 * jlo + g * 4 + (c >> 2)) * PK + (c & 3) * 8;
        f32x4 o[2 * DPK]; float lse[4];
        if (nblk <= 4) fwd2_tile<DPK, 4, DROP>(p, lds, kaddr, eaddr, vaddr, ptw, ptr_, qf, srcb, sel, b * H + h, q0, jlo, lane, o, lse);
        else if (nblk <= 10) fwd2_tile<DPK, 10, DROP>(p, lds, kaddr, eaddr, vaddr, ptw, ptr_, qf, srcb, sel, b * H + h, q0, jlo, lane, o, lse);
        else fwd2_tile<DPK, RES_NB, DROP>(p, lds, kaddr, eaddr, vaddr, ptw, ptr_, qf, srcb, sel, b * H + h, q0, jlo, lane, o, lse);
        if (c == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { const int q = q0 + g * 4 + reg; if (q < Tn) p.lse[((long long)b * H + h) * Tn + q] = lse[reg]; }
        }
        store_tile_rows<DPK>((RT*)(lds + PT + w * 2 * F2_PTB), o, 1.f, (RT*)p.out + (long long)b * Tn * (H * dp) + h * dp, (long long)H * dp, q0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) qf[kk] = qn[kk];
    }
    }
}

// =========================================================================== resident backward on the saved probabilities
// Both kernels read the P image of the forward instead of recomputing the logits.  With pf = the stored value as a float (negative
// iff dropout removed the entry), s = 1 / (1 - p_drop) and D' = D / s the scaled-down score gradient is
//     dS' = max(pf, 0) * dP - |pf| * D'          (dS = s * dS',  P~ = s * max(pf, 0),  dP = dO . V)
// i.e. 1.5 unpack + 3 arithmetic instructions per entry; the factor s is applied to the 16 x dp results.
namespace {
// the 4 stored probabilities of a lane: (signed) f32
__device__ __forceinline__ void pimg_unpack(const u32x2& w, float (&pf)[4]) {
    pf[0] = __uint_as_float(w[0] << 16); pf[1] = __uint_as_float(w[0] & 0xffff0000u);
    pf[2] = __uint_as_float(w[1] << 16); pf[3] = __uint_as_float(w[1] & 0xffff0000u);
}

// acc[n] += A B with A[q][k] handed over as packed bf16 column pairs (src(j) = this lane's 4 rows of 16-key block j, zero for
// blocks the caller does not have) and B = 32-row chunks of a row-major LDS table starting at baddr (per-lane transposing-read
// address): blocks go through the per-wave chunk buffers ([32 keys][16 queries + 4]) two chunks ahead of their MFMAs.
template <int DPK, int NC, class SRC>
__device__ __forceinline__ void chunk_mma(unsigned char* lds, unsigned ptw, unsigned ptr_, unsigned baddr, SRC&& src, f32x4 (&acc)[2 * DPK])
{
    constexpr int PK = DPK * 64 + RES2_PAD;
    s16x4 alo[2], ahi[2], blo[2][2 * DPK], bhi[2][2 * DPK];
    auto put = [&](auto kcc) {
        constexpr int kc = kcc;
        awr64<(kc & 1) * F2_PTB>(lds, ptw, src(std::integral_constant<int, 2 * kc>{}));
        awr64<(kc & 1) * F2_PTB + 16 * 40>(lds, ptw, src(std::integral_constant<int, 2 * kc + 1>{}));
    };
    auto get = [&](auto kcc) {
        constexpr int kc = kcc;
        ard64tr<(kc & 1) * F2_PTB>(alo[kc & 1], lds, ptr_); ard64tr<(kc & 1) * F2_PTB + 16 * 40>(ahi[kc & 1], lds, ptr_);
        sfor<0, 2 * DPK>([&](auto n) { ard64tr<kc * 32 * PK + n * 32>(blo[kc & 1][n], lds, baddr); ard64tr<kc * 32 * PK + 16 * PK + n * 32>(bhi[kc & 1][n], lds, baddr); });
    };
    auto pin = [&](auto kcc) {
        constexpr int kc = kcc;
        apin(alo[kc & 1]); apin(ahi[kc & 1]);
        sfor<0, 2 * DPK>([&](auto n) { apin(blo[kc & 1][n]); apin(bhi[kc & 1][n]); });
    };
    put(std::integral_constant<int, 0>{});
    if constexpr (NC > 1) put(std::integral_constant<int, 1>{});
    afence();
    await0();
    get(std::integral_constant<int, 0>{});
    await0();
    pin(std::integral_constant<int, 0>{});
    afence();
    sfor<0, NC>([&](auto kcc) {
        constexpr int kc = kcc;
        if constexpr (kc + 1 < NC) get(std::integral_constant<int, kc + 1>{});
        afence();
        {
            const bf16x8 a = join8(alo[kc & 1], ahi[kc & 1]);
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) acc[n] = mfma_bf16_16x16x32(a, join8(blo[kc & 1][n], bhi[kc & 1][n]), acc[n]);
        }
        if constexpr (kc + 2 < NC) put(std::integral_constant<int, kc + 2>{});
        afence();
        await0();
        if constexpr (kc + 1 < NC) pin(std::integral_constant<int, kc + 1>{});
        afence();
    });
}

// one query tile of the query-major backward: dQ = s * (scale * dS' K + dR' E)
template <int DPK, int NBLK>
__device__ __forceinline__ void bq2_tile(const AttnP& p, unsigned char* lds, unsigned vfaddr, unsigned kaddr_tr, unsigned eaddr_tr, unsigned ptw, unsigned ptr_,
                                         const bf16x8 (&dof)[DPK], const float (&dprime)[4], const u32x2* img, const unsigned (&srcb)[4], const bool (&sel)[4],
                                         f32x4 (&acc)[2 * DPK])
{
    constexpr int PK = DPK * 64 + RES2_PAD, BLK = 16 * PK, NCH = (NBLK + 1) / 2, NCHM = (NBLK + 2) / 2;
    u32x2 im[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) im[j] = img[j * 64];
    float ds[NBLK][4];
    bf16x8 F[2][DPK];
    f32x4 dpa[2];
    auto reads = [&](auto ic) { constexpr int i = ic; sfor<0, DPK>([&](auto kk) { ard128<i * BLK + kk * 64>(F[i & 1][kk], lds, vfaddr); }); };
    auto mfmas = [&](auto ic) {
        constexpr int i = ic;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) d = mfma_bf16_16x16x32(dof[kk], F[i & 1][kk], d);
        dpa[i & 1] = d;
    };
    auto grads = [&](auto jc) {
        constexpr int j = jc;
        float pf[4]; pimg_unpack(im[j], pf);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ds[j][reg] = fmaf(fmaxf(pf[reg], 0.f), dpa[j & 1][reg], -(fabsf(pf[reg]) * dprime[reg]));
    };
    reads(std::integral_constant<int, 0>{});
    await0();
    sfor<0, DPK>([&](auto x) { apin(F[0][x]); });
    afence();
    sfor<0, NBLK + 1>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (i + 1 < NBLK) reads(std::integral_constant<int, i + 1>{});
        afence();
        if constexpr (i < NBLK) mfmas(ic);
        if constexpr (i >= 1) grads(std::integral_constant<int, i - 1>{});
        afence();
        await0();
        if constexpr (i + 1 < NBLK) sfor<0, DPK>([&](auto x) { apin(F[(i + 1) & 1][x]); });
        afence();
    });
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[n] = z; }
    // content term
    chunk_mma<DPK, NCH>(lds, ptw, ptr_, kaddr_tr, [&](auto jc) {
        constexpr int j = jc;
        u32x2 pk = {0u, 0u};
        if constexpr (j < NBLK) { pk[0] = pack_bf16(ds[j][0], ds[j][1]); pk[1] = pack_bf16(ds[j][2], ds[j][3]); }
        return pk;
    }, acc);
#pragma unroll
    for (int n = 0; n < 2 * DPK; ++n) acc[n] = acc[n] * p.scale;
    // positional term: window block u of dR holds, for row r, column (col + r - 15) of key block u (col + r >= 15) or u - 1; the SOURCE lane
    // (column s) is read for block u iff s <= r, so it selects before the one ds_bpermute
    float dr[NBLK + 1][4];
    sfor<0, NBLK + 1>([&](auto uc) {
        constexpr int u = uc;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float hi = u < NBLK ? ds[u < NBLK ? u : 0][reg] : 0.f, lo = u >= 1 ? ds[u >= 1 ? u - 1 : 0][reg] : 0.f;
            abperm(dr[u][reg], srcb[reg], sel[reg] ? hi : lo);
        }
    });
    await0();
    sfor<0, NBLK + 1>([&](auto uc) { constexpr int u = uc; sfor<0, 4>([&](auto x) { apin(dr[u][x]); }); });
    afence();
    chunk_mma<DPK, NCHM>(lds, ptw, ptr_, eaddr_tr, [&](auto uc) {
        constexpr int u = uc;
        u32x2 pk = {0u, 0u};
        if constexpr (u < NBLK + 1) { pk[0] = pack_bf16(dr[u][0], dr[u][1]); pk[1] = pack_bf16(dr[u][2], dr[u][3]); }
        return pk;
    }, acc);
}
}  // namespace

template <int DPK>
__global__ __launch_bounds__(RES_W_BQ * 64) void attn_bwd_q2_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + RES2_PAD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int H = p.H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4, Tr = nb * 16, NE = 2 * D - 1;
    unsigned char* lds = (unsigned char*)smem;
    // [K | V | E | chunk buffers]: V rows past the band continue into E, E rows outside the table into V resp. the (zeroed) buffers: always finite, always met by dS' = 0
    const unsigned KS = 0, VS = KS + Tr * PK, ES = VS + Tr * PK, PT = ES + NE * PK, CT = PT + (unsigned)p.tail;
    int* ctr = (int*)(lds + CT);
    const long long ldq = 3LL * H * dp;
    const unsigned lbase = lds_byte_address(lds);
    const float sdrop = p.drop_scale, inv_s = 1.f / sdrop;
    int pair, half;
    for (int item = 0;; ++item) {                                           // one (sequence, head) pair, or the sequences of this workgroup's head (res2_item)
    const AttnArgs a = fresh_args(p);
    if (!res2_item(a, item, pair, half)) break;
    const int h = pair % H, b = pair / H;
    const RT* Q = (const RT*)a->qkv + (long long)b * Tn * ldq + h * dp;
    const RT* dO = (const RT*)a->dO + (long long)b * Tn * (H * dp) + h * dp;
    if (item > 0) __syncthreads();                                          // every wave has left the tile loop of the previous sequence
    {
        const int ts = opaque_v(tid);
        stage_rows<DPK>(lds + KS, PK, Q + H * dp, ldq, Tn, Tr, ts, RES_W_BQ * 64);
        stage_rows<DPK>(lds + VS, PK, Q + 2 * H * dp, ldq, Tn, Tr, ts, RES_W_BQ * 64);
        if (item == 0) {                                                    // the head's table and the zeroed buffers: once per workgroup
            stage_rows<DPK>(lds + ES, PK, (const RT*)a->E + (long long)h * NE * dp, dp, NE, NE, ts, RES_W_BQ * 64);
            for (unsigned i = PT + ts * 16; i < CT; i += RES_W_BQ * 64 * 16) { u32x4 z = {0u, 0u, 0u, 0u}; *(u32x4*)(lds + i) = z; }
        }
        if (tid == 0) *ctr = 0;
    }
    __syncthreads();
    const int lane = opaque_v(tid) & 63, c = lane & 15, g = lane >> 4;       // the tile loop's lane constants are formed per item: nothing of them lives through the staging
    unsigned srcb[4]; bool sel[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) { const int r = g * 4 + reg; srcb[reg] = (unsigned)((((c + r + 1) & 15) + 16 * g) * 4); sel[reg] = c <= r; }
    const unsigned ptw = lbase + PT + w * 2 * F2_PTB + (c * 20 + g * 4) * 2, ptr_ = lbase + PT + w * 2 * F2_PTB + (g * 4 + (c >> 2)) * 40 + (c & 3) * 8;
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 dof[DPK], don[DPK];
    if (it < nb) { int qr = res_tile_of(it, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(dof, dO + (long long)qr * (H * dp), true, g); }
    while (it < nb) {
        const int q0 = res_tile_of(it, nb) * 16;
        int jlo = q0 - (D - 1); jlo = jlo < 0 ? 0 : jlo >> 4;
        int jhi = (q0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int nblk = jhi - jlo + 1;
        float dprime[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { const int q = q0 + g * 4 + reg; dprime[reg] = p.Dv[((long long)b * H + h) * Tn + (q < Tn ? q : Tn - 1)] * inv_s; }
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int qr = res_tile_of(itn, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1; glb_row_frags<DPK>(don, dO + (long long)qr * (H * dp), true, g); }
        const unsigned vfaddr = lbase + VS + (16 * jlo + c) * PK + g * 16;
        const unsigned kaddr_tr = lbase + KS + (16 * jlo + g * 4 + (c >> 2)) * PK + (c & 3) * 8;
        const unsigned eaddr_tr = lbase + ES + (unsigned)((16 * jlo - q0 - 15 + (D - 1) + g * 4 + (c >> 2)) * PK) + (c & 3) * 8;
        const u32x2* img = pimg_block(p.pimg, pair, nb, q0 >> 4, jlo, lane);
        f32x4 acc[2 * DPK];
        if (nblk <= 4) bq2_tile<DPK, 4>(p, lds, vfaddr, kaddr_tr, eaddr_tr, ptw, ptr_, dof, dprime, img, srcb, sel, acc);
        else if (nblk <= 10) bq2_tile<DPK, 10>(p, lds, vfaddr, kaddr_tr, eaddr_tr, ptw, ptr_, dof, dprime, img, srcb, sel, acc);
        else bq2_tile<DPK, RES_NB>(p, lds, vfaddr, kaddr_tr, eaddr_tr, ptw, ptr_, dof, dprime, img, srcb, sel, acc);
        store_tile_rows<DPK>((RT*)(lds + PT + w * 2 * F2_PTB), acc, sdrop, (RT*)p.dqkv + (long long)b * Tn * ldq + h * dp, ldq, q0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) dof[kk] = don[kk];
    }
    }
}

// one key tile of the key-major backward: dV = s P~'^T dO, dK = s scale dS'^T Q over NCK 32-query chunks starting at chunk c0
namespace {
constexpr int KV_TILE = 16 * RT_LD * 2;            // bytes of one [16 keys][32 queries + 8] bf16 hand-over tile
template <int DPK, int NCK>
__device__ __forceinline__ void bkv2_tile(unsigned char* lds, unsigned dofaddr, unsigned dvaddr, unsigned qtr, unsigned dotr, unsigned tpa,
                                          const bf16x8 (&vf)[DPK], const u32x2* img_tile0, long long img_stride, int c0, int ilo, int ihi,
                                          f32x4 (&dk)[2 * DPK], f32x4 (&dvv)[2 * DPK])
{
    constexpr int PK = DPK * 64 + RES2_PAD, BLK = 16 * PK;
    // this lane's image words of the 2 NCK (tile, key tile) blocks; tiles outside [ilo, ihi] (never written by the forward) read a valid neighbour and are zeroed
    u32x2 im[2 * NCK];
#pragma unroll
    for (int x = 0; x < 2 * NCK; ++x) {
        const int i = 2 * c0 + x, ic = i < ilo ? ilo : (i > ihi ? ihi : i);
        const unsigned vm = (i >= ilo && i <= ihi) ? 0xffffffffu : 0u;
        u32x2 w = img_tile0[(long long)(ic - 2 * c0) * img_stride];
        w[0] &= vm; w[1] &= vm; im[x] = w;
    }
    bf16x8 dof[2][DPK];                             // dO rows (A operand of dP) of the two 16-query blocks of one chunk
    f32x4 dpr[2][2];                                // D' of those rows: ring of 2 chunks
    f32x4 dpa[2];
    s16x4 pal, pah, sal, sah, blo[2][2 * DPK], bhi[2][2 * DPK];          // [0] = dO columns (for dV), [1] = Q columns (for dK)
    auto rows_issue = [&](auto tc) {
        constexpr int t = tc;
        sfor<0, 2>([&](auto hc) { constexpr int hh = hc; sfor<0, DPK>([&](auto kk) { ard128<(2 * t + hh) * BLK + kk * 64>(dof[hh][kk], lds, dofaddr); }); });
        ard128<(2 * t) * 64>(dpr[t & 1][0], lds, dvaddr); ard128<(2 * t + 1) * 64>(dpr[t & 1][1], lds, dvaddr);
    };
    auto rows_pin = [&](auto tc) {
        constexpr int t = tc;
        sfor<0, 2>([&](auto hc) { constexpr int hh = hc; sfor<0, DPK>([&](auto kk) { apin(dof[hh][kk]); }); });
        apin(dpr[t & 1][0]); apin(dpr[t & 1][1]);
    };
    auto dp_mfma = [&]() {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DPK; ++kk) d = mfma_bf16_16x16x32(dof[hh][kk], vf[kk], d);
            dpa[hh] = d;
        }
    };
    auto grads_put = [&](auto tc) {                 // dS', P~' of chunk t -> hand-over tiles of buffer t & 1
        constexpr int t = tc;
        sfor<0, 2>([&](auto hc) {
            constexpr int hh = hc;
            float pf[4], us[4], ds[4]; pimg_unpack(im[2 * t + hh], pf);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { us[reg] = fmaxf(pf[reg], 0.f); ds[reg] = fmaf(us[reg], dpa[hh][reg], -(fabsf(pf[reg]) * dpr[t & 1][hh][reg])); }
            const u32x2 pp = {pack_bf16(us[0], us[1]), pack_bf16(us[2], us[3])}, ss = {pack_bf16(ds[0], ds[1]), pack_bf16(ds[2], ds[3])};
            awr64<(t & 1) * 2 * KV_TILE + hh * 32>(lds, tpa, pp);
            awr64<(t & 1) * 2 * KV_TILE + KV_TILE + hh * 32>(lds, tpa, ss);
        });
    };
    // a wave holds at most 15 LDS operations in flight (4-bit lgkmcnt): the operands of one chunk are requested in two batches of 2 + 4 DPK
    auto issue_v = [&](auto tc) {
        constexpr int t = tc;
        ard64<(t & 1) * 2 * KV_TILE>(pal, lds, tpa); ard64<(t & 1) * 2 * KV_TILE + 32>(pah, lds, tpa);
        sfor<0, 2 * DPK>([&](auto n) { ard64tr<t * 32 * PK + n * 32>(blo[0][n], lds, dotr); ard64tr<t * 32 * PK + 16 * PK + n * 32>(bhi[0][n], lds, dotr); });
    };
    auto issue_k = [&](auto tc) {
        constexpr int t = tc;
        ard64<(t & 1) * 2 * KV_TILE + KV_TILE>(sal, lds, tpa); ard64<(t & 1) * 2 * KV_TILE + KV_TILE + 32>(sah, lds, tpa);
        sfor<0, 2 * DPK>([&](auto n) { ard64tr<t * 32 * PK + n * 32>(blo[1][n], lds, qtr); ard64tr<t * 32 * PK + 16 * PK + n * 32>(bhi[1][n], lds, qtr); });
    };
    auto pin_v = [&]() { apin(pal); apin(pah); sfor<0, 2 * DPK>([&](auto n) { apin(blo[0][n]); apin(bhi[0][n]); }); };
    auto pin_k = [&]() { apin(sal); apin(sah); sfor<0, 2 * DPK>([&](auto n) { apin(blo[1][n]); apin(bhi[1][n]); }); };
    // prologue: rows of chunk 0 -> dP(0) -> rows of chunk 1 in flight -> tiles(0)
    rows_issue(std::integral_constant<int, 0>{});
    await0();
    rows_pin(std::integral_constant<int, 0>{});
    afence();
    dp_mfma();
    afence();
    if constexpr (NCK > 1) rows_issue(std::integral_constant<int, 1>{});
    grads_put(std::integral_constant<int, 0>{});
    afence();
    await0();
    if constexpr (NCK > 1) rows_pin(std::integral_constant<int, 1>{});
    afence();
    sfor<0, NCK>([&](auto tc) {
        constexpr int t = tc;
        issue_v(tc);                                                         // P~' tile + dO columns of chunk t
        afence();
        if constexpr (t + 1 < NCK) dp_mfma();                                // dP of chunk t+1 (its rows landed during the previous step)
        afence();
        await0();
        pin_v();
        afence();
        issue_k(tc);                                                         // dS' tile + Q columns
        afence();
        {
            const bf16x8 pa = join8(pal, pah);
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) dvv[n] = mfma_bf16_16x16x32(pa, join8(blo[0][n], bhi[0][n]), dvv[n]);
        }
        afence();
        await0();
        pin_k();
        afence();
        if constexpr (t + 2 < NCK) rows_issue(std::integral_constant<int, t + 2>{});
        afence();
        {
            const bf16x8 sa = join8(sal, sah);
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) dk[n] = mfma_bf16_16x16x32(sa, join8(blo[1][n], bhi[1][n]), dk[n]);
        }
        if constexpr (t + 1 < NCK) grads_put(std::integral_constant<int, t + 1>{});
        afence();
        await0();
        if constexpr (t + 2 < NCK) rows_pin(std::integral_constant<int, t + 2>{});
        afence();
    });
}
}  // namespace

template <int DPK>
__global__ __launch_bounds__(RES_W_BKV * 64) void attn_bwd_kv2_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + RES2_PAD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    int pair, half; res2_block(p, pair, half);
    const int H = p.H, h = pair % H, b = pair / H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4;
    const int TQ = ((nb + 1) >> 1) * 32;                               // whole 32-query chunks
    unsigned char* lds = (unsigned char*)smem;
    // [Q rows | dO rows | hand-over tiles (zeroed) | D' (zeroed past the sequence)]: chunk slots past the band read on into the next region, always finite, always met by P = 0
    const unsigned QS = 0, DOS = QS + TQ * PK, TP = DOS + TQ * PK, DV = TP + RES_W_BKV * 4 * KV_TILE, CT = DV + (TQ + 7 * 32) * 4;
    int* ctr = (int*)(lds + CT);
    const long long ldq = 3LL * H * dp;
    const RT* Q = (const RT*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const RT* K = Q + H * dp;
    const RT* V = Q + 2 * H * dp;
    const float sdrop = p.drop_scale, inv_s = 1.f / sdrop;
    {
        stage_rows<DPK>(lds + QS, PK, Q, ldq, Tn, TQ, tid, RES_W_BKV * 64);
        stage_rows<DPK>(lds + DOS, PK, (const RT*)p.dO + (long long)b * Tn * (H * dp) + h * dp, (long long)H * dp, Tn, TQ, tid, RES_W_BKV * 64);
        for (unsigned i = TP + tid * 16; i < CT; i += RES_W_BKV * 64 * 16) { u32x4 z = {0u, 0u, 0u, 0u}; *(u32x4*)(lds + i) = z; }
    }
    __syncthreads();
    for (int i = tid; i < Tn; i += RES_W_BKV * 64) ((float*)(lds + DV))[i] = p.Dv[((long long)b * H + h) * Tn + i] * inv_s;
    if (tid == 0) *ctr = 0;
    __syncthreads();
    const unsigned lbase = lds_byte_address(lds);
    const unsigned tpa = lbase + TP + w * 4 * KV_TILE + (c * RT_LD + g * 4) * 2;
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 vf[DPK], vn[DPK];
    if (it < nb) { int kr = res_tile_of(it, nb) * 16 + c; const bool ok = kr < Tn; kr = ok ? kr : Tn - 1; glb_row_frags<DPK>(vf, V + (long long)kr * ldq, ok, g); }
    while (it < nb) {
        const int jt = res_tile_of(it, nb), k0 = jt * 16;
        int ilo = k0 - (D - 1); ilo = ilo < 0 ? 0 : ilo >> 4;
        int ihi = (k0 + 15 + D - 1) >> 4; ihi = ihi > nb - 1 ? nb - 1 : ihi;
        const int c0 = ilo >> 1, nck = (ihi >> 1) - c0 + 1;
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int kr = res_tile_of(itn, nb) * 16 + c; const bool ok = kr < Tn; kr = ok ? kr : Tn - 1; glb_row_frags<DPK>(vn, V + (long long)kr * ldq, ok, g); }
        const unsigned dofaddr = lbase + DOS + (32 * c0 + c) * PK + g * 16;
        const unsigned dvaddr = lbase + DV + (32 * c0 + g * 4) * 4;
        const unsigned qtr = lbase + QS + (32 * c0 + g * 4 + (c >> 2)) * PK + (c & 3) * 8, dotr = qtr + (DOS - QS);
        const u32x2* img0 = pimg_block(p.pimg, pair, nb, 2 * c0, jt, lane);
        const long long istride = (long long)pimg_slots(nb) * 64;
        f32x4 dk[2 * DPK], dvv[2 * DPK];
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dk[n] = z; dvv[n] = z; }
        if (nck <= 2) bkv2_tile<DPK, 2>(lds, dofaddr, dvaddr, qtr, dotr, tpa, vf, img0, istride, c0, ilo, ihi, dk, dvv);
        else if (nck <= 5) bkv2_tile<DPK, 5>(lds, dofaddr, dvaddr, qtr, dotr, tpa, vf, img0, istride, c0, ilo, ihi, dk, dvv);
        else bkv2_tile<DPK, 7>(lds, dofaddr, dvaddr, qtr, dotr, tpa, vf, img0, istride, c0, ilo, ihi, dk, dvv);
        RT* dK = (RT*)p.dqkv + (long long)b * Tn * ldq + H * dp + h * dp;
        RT* tl = (RT*)(lds + TP + w * 4 * KV_TILE);
        store_tile_rows<DPK>(tl, dk, p.scale * sdrop, dK, ldq, k0, Tn, lane);
        store_tile_rows<DPK>(tl + 16 * RT_LD, dvv, sdrop, dK + H * dp, ldq, k0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) vf[kk] = vn[kk];
    }
}

// probability and dS of one block from log2-domain logits:  p = 2^(l2 - lse2);  dS = p * (keep ? dP/(1-pd) : 0  -  D).
// Masked entries carry l2 = -1e8*log2(e), so p is exactly 0 there; rows past the sequence are cut by rowok.
template <bool DROP>
__device__ __forceinline__ void res_prob_ds(const float (&l2)[4], const f32x4& dpv, const float (&lse2)[4], const float (&dv)[4], const bool (&rowok)[4],
                                            const AttnP& p, int b, int h, int q0, int k0, int lane, float (&pd)[4], float (&ds)[4])
{
    const int c = lane & 15, g = lane >> 4;
    bool kp[4] = {true, true, true, true};
    if (DROP) res_drop_keep4(res_drop_key(p, b * p.H + h, q0, g), k0 + c, p.drop_thresh >> 17, kp);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const float pr = rowok[reg] ? fast_exp2(l2[reg] - lse2[reg]) : 0.f;
        const float keep = DROP ? (kp[reg] ? p.drop_scale : 0.f) : 1.f;
        pd[reg] = pr * keep;
        ds[reg] = pr * (dpv[reg] * keep - dv[reg]);
    }
}

// query-major backward (dQ) on resident K, V rows and a zero-padded embedding table (PL rows below, PH above)
template <int DPK, bool DROP>
__global__ __launch_bounds__(RES_W_BQ * 64) void attn_bwd_q_res_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + 16, PL = 48, PH = 48, PTL = 20;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    // workgroups >= p.gx are HALVES of the (sequence, head) pairs that would otherwise form a short last round: each stages the
    // operands itself and takes every second tile of the heaviest-first order (res_split_index)
    const int bid = blockIdx.x, pair = bid < p.gx ? bid : p.gx + ((bid - p.gx) >> 1), half = bid < p.gx ? -1 : ((bid - p.gx) & 1);
    const int H = p.H, h = pair % H, b = pair / H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4, Tr = nb * 16, NE = 2 * D - 1, ER = NE + PL + PH;
    unsigned char* Ks = (unsigned char*)smem;
    unsigned char* Vs = Ks + Tr * PK;
    unsigned char* Es = Vs + Tr * PK;                                  // row m + PL holds embedding m
    RT* tA = (RT*)(Es + ER * PK) + w * 16 * RT_LD;                     // dS^T / dR^T chunk: [32 contraction rows][16 queries (+4)]
    int* ctr = (int*)(Es + ER * PK + RES_W_BQ * 16 * RT_LD * 2);
    const long long ldq = 3LL * H * dp;
    const RT* Q = (const RT*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const RT* dO = (const RT*)p.dO + (long long)b * Tn * (H * dp) + h * dp;
    const float scale2 = p.scale * LOG2E;
    {
        stage_rows<DPK>(Ks, PK, Q + H * dp, ldq, Tn, Tr, tid, RES_W_BQ * 64);
        stage_rows<DPK>(Vs, PK, Q + 2 * H * dp, ldq, Tn, Tr, tid, RES_W_BQ * 64);
        stage_rows<DPK>(Es, PK, (const RT*)p.E, dp, 0, PL, tid, RES_W_BQ * 64);
        stage_rows<DPK>(Es + PL * PK, PK, (const RT*)p.E + (long long)h * NE * dp, dp, NE, NE + PH, tid, RES_W_BQ * 64);
        if (tid == 0) *ctr = 0;
    }
    __syncthreads();
    int bandA[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) bandA[reg] = c - (g * 4 + reg) + (D - 1);
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 qf[DPK], dof[DPK], qn[DPK], don[DPK];
    if (it < nb) { int qr = res_tile_of(it, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1;
                   glb_row_frags<DPK>(qf, Q + (long long)qr * ldq, true, g); glb_row_frags<DPK>(dof, dO + (long long)qr * (H * dp), true, g); }
    while (it < nb) {
        const int q0 = res_tile_of(it, nb) * 16;
        int jlo = q0 - (D - 1); jlo = jlo < 0 ? 0 : jlo >> 4;
        int jhi = (q0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int m_org = -q0 - 15 + (D - 1);
        float lse2[4], dv[4]; bool rowok[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int q = q0 + g * 4 + reg; rowok[reg] = q < Tn;
            const long long si = ((long long)b * H + h) * Tn + (rowok[reg] ? q : 0);
            lse2[reg] = p.lse[si] * LOG2E; dv[reg] = p.Dv[si];
        }
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int qr = res_tile_of(itn, nb) * 16 + c; qr = qr < Tn ? qr : Tn - 1;
                        glb_row_frags<DPK>(qn, Q + (long long)qr * ldq, true, g); glb_row_frags<DPK>(don, dO + (long long)qr * (H * dp), true, g); }
        float dsr[RES_NB][4];
        f32x4 rprev;
        { bf16x8 ef[DPK]; lds_row_frags<DPK>(ef, Es, PK, m_org + 16 * jlo + c + PL, g); rprev = dot8<DPK>(qf, ef); }
#pragma unroll
        for (int j = 0; j < RES_NB; ++j) {
            if (j >= jlo && j <= jhi) {
                bf16x8 kf[DPK], vf[DPK], ef[DPK];
                lds_row_frags<DPK>(kf, Ks, PK, 16 * j + c, g);
                lds_row_frags<DPK>(vf, Vs, PK, 16 * j + c, g);
                lds_row_frags<DPK>(ef, Es, PK, m_org + 16 * (j + 1) + c + PL, g);
                const f32x4 s = dot8<DPK>(qf, kf);
                const f32x4 rn = dot8<DPK>(qf, ef);
                const f32x4 dpv = dot8<DPK>(dof, vf);
                float pos[4], l2[4], pd[4];
                skew_gather(rprev, rn, lane, pos);
                res_logits(s, pos, bandA, 16 * j - q0, 16 * j + c < Tn, 2u * (unsigned)(D - 1), scale2, l2);
                res_prob_ds<DROP>(l2, dpv, lse2, dv, rowok, p, b, h, q0, 16 * j, lane, pd, dsr[j]);
                rprev = rn;
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dsr[j][reg] = 0.f;
            }
        }
        f32x4 acc[2 * DPK];
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[n] = z; }
        // content term: dS . K.  dS^T goes to LDS as [key][query] (this lane's 4 rows are adjacent: one 8-byte store per block),
        // both operands come back through transposing reads; key rows past Tr belong to the V tile and meet dS = 0.
#pragma unroll
        for (int kc = 0; kc < (RES_NB + 1) / 2; ++kc) {
            if (2 * kc > jhi || 2 * kc + 1 < jlo) continue;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int j = 2 * kc + half;
                u32x2 pk = {0u, 0u};
                if (j < RES_NB) { pk[0] = pack_bf16(dsr[j < RES_NB ? j : 0][0], dsr[j < RES_NB ? j : 0][1]); pk[1] = pack_bf16(dsr[j < RES_NB ? j : 0][2], dsr[j < RES_NB ? j : 0][3]); }
                *(u32x2*)(tA + (16 * half + c) * PTL + g * 4) = pk;
            }
            wave_lds_sync();
            const bf16x8 a = lds_b_tr((const unsigned char*)tA, PTL * 2, 0, 0, c, g);
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) acc[n] = mfma_bf16_16x16x32(a, lds_b_tr(Ks, PK, kc * 32, n * 32, c, g), acc[n]);
            wave_lds_sync();
        }
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) acc[n] = acc[n] * p.scale;
        // positional term: dR . E with dR[q][m] = dS[q][k = m - (D-1) + q]: blocks u of 16 relative positions on the grid
        // m = m_org + 16u + col hold, for row r, column (col + r - 15) of key block u (col + r >= 15) or u-1 (otherwise)
#pragma unroll
        for (int mc = 0; mc < (RES_NB + 2) / 2; ++mc) {
            if (2 * mc > jhi + 1 || 2 * mc + 1 < jlo) continue;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int u = 2 * mc + half;
                float val[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int ql = g * 4 + reg, src = ((c + ql + 1) & 15) + 16 * g;
                    const float lo = u >= 1 && u - 1 < RES_NB ? dsr[(u >= 1 && u - 1 < RES_NB) ? u - 1 : 0][reg] : 0.f;
                    const float hi = u < RES_NB ? dsr[u < RES_NB ? u : 0][reg] : 0.f;
                    const float a = __shfl(lo, src), bb = __shfl(hi, src);
                    val[reg] = c + ql >= 15 ? bb : a;
                }
                u32x2 pk = {pack_bf16(val[0], val[1]), pack_bf16(val[2], val[3])};
                *(u32x2*)(tA + (16 * half + c) * PTL + g * 4) = pk;
            }
            wave_lds_sync();
            const bf16x8 a = lds_b_tr((const unsigned char*)tA, PTL * 2, 0, 0, c, g);
            const int er0 = m_org + 32 * mc + PL;                       // >= 2 and + 31 < ER for every chunk that passes the test above
#pragma unroll
            for (int n = 0; n < 2 * DPK; ++n) acc[n] = mfma_bf16_16x16x32(a, lds_b_tr(Es, PK, er0, n * 32, c, g), acc[n]);
            wave_lds_sync();
        }
        store_tile_rows<DPK>(tA, acc, 1.f, (RT*)p.dqkv + (long long)b * Tn * ldq + h * dp, ldq, q0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) { qf[kk] = qn[kk]; dof[kk] = don[kk]; }
    }
}

// key-major backward (dK, dV) on resident Q, dO rows
template <int DPK, bool DROP>
__global__ __launch_bounds__(RES_W_BKV * 64) void attn_bwd_kv_res_kernel(AttnP p)
{
    SS_DYN_SMEM(smem);
    constexpr int dp = DPK * 32, PK = dp * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    // workgroups >= p.gx are HALVES of the (sequence, head) pairs that would otherwise form a short last round: each stages the
    // operands itself and takes every second tile of the heaviest-first order (res_split_index)
    const int bid = blockIdx.x, pair = bid < p.gx ? bid : p.gx + ((bid - p.gx) >> 1), half = bid < p.gx ? -1 : ((bid - p.gx) & 1);
    const int H = p.H, h = pair % H, b = pair / H;
    const int Tn = p.T, D = p.D, nb = (Tn + 15) >> 4, NE = 2 * D - 1;
    const int TQ = ((nb + 1) >> 1) * 32;                               // whole 32-query steps
    unsigned char* Qs = (unsigned char*)smem;
    unsigned char* dOs = Qs + TQ * PK;
    unsigned char* Es = dOs + TQ * PK;
    RT* tP = (RT*)(Es + NE * PK) + w * 2 * 16 * RT_LD;                 // P~^T, dS^T chunks: [16 keys][32 queries (+8)]
    RT* tS = tP + 16 * RT_LD;
    float* lse2s = (float*)(Es + NE * PK + RES_W_BKV * 2 * 16 * RT_LD * 2);
    float* dvs = lse2s + TQ;
    int* ctr = (int*)(dvs + TQ);
    const long long ldq = 3LL * H * dp;
    const RT* Q = (const RT*)p.qkv + (long long)b * Tn * ldq + h * dp;
    const RT* K = Q + H * dp;
    const RT* V = Q + 2 * H * dp;
    const float scale2 = p.scale * LOG2E;
    {
        stage_rows<DPK>(Qs, PK, Q, ldq, Tn, TQ, tid, RES_W_BKV * 64);
        stage_rows<DPK>(dOs, PK, (const RT*)p.dO + (long long)b * Tn * (H * dp) + h * dp, (long long)H * dp, Tn, TQ, tid, RES_W_BKV * 64);
        stage_rows<DPK>(Es, PK, (const RT*)p.E + (long long)h * NE * dp, dp, NE, NE, tid, RES_W_BKV * 64);
        for (int i = tid; i < TQ; i += RES_W_BKV * 64) {
            const long long si = ((long long)b * H + h) * Tn + (i < Tn ? i : Tn - 1);
            lse2s[i] = p.lse[si] * LOG2E; dvs[i] = p.Dv[si];
        }
        if (tid == 0) *ctr = 0;
    }
    __syncthreads();
    int it = res_split_index(res_next(ctr, lane), half);
    bf16x8 kf[DPK], vf[DPK], kn[DPK], vn[DPK];
    if (it < nb) { int kr = res_tile_of(it, nb) * 16 + c; const bool ok = kr < Tn; kr = ok ? kr : Tn - 1;
                   glb_row_frags<DPK>(kf, K + (long long)kr * ldq, ok, g); glb_row_frags<DPK>(vf, V + (long long)kr * ldq, ok, g); }
    while (it < nb) {
        const int k0 = res_tile_of(it, nb) * 16;
        int jlo = k0 - (D - 1); jlo = jlo < 0 ? 0 : jlo >> 4;
        int jhi = (k0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int itn = res_split_index(res_next(ctr, lane), half);
        if (itn < nb) { int kr = res_tile_of(itn, nb) * 16 + c; const bool ok = kr < Tn; kr = ok ? kr : Tn - 1;
                        glb_row_frags<DPK>(kn, K + (long long)kr * ldq, ok, g); glb_row_frags<DPK>(vn, V + (long long)kr * ldq, ok, g); }
        f32x4 dk[2 * DPK], dvv[2 * DPK];
#pragma unroll
        for (int n = 0; n < 2 * DPK; ++n) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; dk[n] = z; dvv[n] = z; }
        const bool key_ok = k0 + c < Tn;
        for (int pr = jlo >> 1; pr <= jhi >> 1; ++pr) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int jq = 2 * pr + half, qb0 = 16 * jq;
                float pd[4] = {0.f, 0.f, 0.f, 0.f}, ds[4] = {0.f, 0.f, 0.f, 0.f};
                if (jq >= jlo && jq <= jhi) {
                    bf16x8 qf[DPK], dof[DPK], e0[DPK], e1[DPK];
                    lds_row_frags<DPK>(qf, Qs, PK, qb0 + c, g);
                    lds_row_frags<DPK>(dof, dOs, PK, qb0 + c, g);
                    const int m0 = k0 - qb0 - 15 + (D - 1);
                    lds_e_frags<DPK>(e0, Es, PK, m0 + c, NE, g);
                    lds_e_frags<DPK>(e1, Es, PK, m0 + 16 + c, NE, g);
                    const f32x4 s = dot8<DPK>(qf, kf);
                    const f32x4 rlo = dot8<DPK>(qf, e0), rhi = dot8<DPK>(qf, e1);
                    const f32x4 dpv = dot8<DPK>(dof, vf);
                    float pos[4], l2[4], lse2[4], dv[4]; bool rowok[4]; int bandA[4];
                    skew_gather(rlo, rhi, lane, pos);
                    const f32x4 lv = *(const f32x4*)(lse2s + qb0 + g * 4), dq = *(const f32x4*)(dvs + qb0 + g * 4);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) { bandA[reg] = c - (g * 4 + reg) + (D - 1); rowok[reg] = qb0 + g * 4 + reg < Tn; lse2[reg] = lv[reg]; dv[reg] = dq[reg]; }
                    res_logits(s, pos, bandA, k0 - qb0, key_ok, 2u * (unsigned)(D - 1), scale2, l2);
                    res_prob_ds<DROP>(l2, dpv, lse2, dv, rowok, p, b, h, qb0, k0, lane, pd, ds);
                }
                u32x2 pp = {pack_bf16(pd[0], pd[1]), pack_bf16(pd[2], pd[3])}, ss = {pack_bf16(ds[0], ds[1]), pack_bf16(ds[2], ds[3])};
                *(u32x2*)(tP + c * RT_LD + half * 16 + g * 4) = pp;
                *(u32x2*)(tS + c * RT_LD + half * 16 + g * 4) = ss;
            }
            wave_lds_sync();
            {
                const bf16x8 pa = lds_a_tr(tP + c * RT_LD, g), sa = lds_a_tr(tS + c * RT_LD, g);
#pragma unroll
                for (int n = 0; n < 2 * DPK; ++n) {
                    dvv[n] = mfma_bf16_16x16x32(pa, lds_b_tr(dOs, PK, 32 * pr, n * 32, c, g), dvv[n]);
                    dk[n] = mfma_bf16_16x16x32(sa, lds_b_tr(Qs, PK, 32 * pr, n * 32, c, g), dk[n]);
                }
            }
            wave_lds_sync();
        }
        RT* dK = (RT*)p.dqkv + (long long)b * Tn * ldq + H * dp + h * dp;
        store_tile_rows<DPK>(tP, dk, p.scale, dK, ldq, k0, Tn, lane);
        store_tile_rows<DPK>(tS, dvv, 1.f, dK + H * dp, ldq, k0, Tn, lane);
        it = itn;
#pragma unroll
        for (int kk = 0; kk < DPK; ++kk) { kf[kk] = kn[kk]; vf[kk] = vn[kk]; }
    }
}

#endif  // SS_ATTN_RES16

// =========================================================================== host side
static int attn_check(const char* what, int dtype, int B, int H, int T, int Tp, int dp, int D, float dropout_p)
{
    SS_CHECK(dtype == SS_F32 || dtype == SS_BF16 || dtype == SS_F32X3, "%s: bad dtype", what);
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
    { const char* e = getenv("SS_ATTN_DEBUG"); p.debug = e ? atoi(e) : 0; }
}

#define SS_ATTN_DISPATCH(KERNEL, grid, BLK, smem)                                                              \
    do {                                                                                                        \
        const int dpk = dp / 32;                                                                                \
        if (dtype == SS_BF16) {                                                                                 \
            if (dpk == 1) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 1>), grid, dim3(BLK), smem, stream, p);            \
            else if (dpk == 2) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 2>), grid, dim3(BLK), smem, stream, p);       \
            else if (dpk == 3) SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 3>), grid, dim3(BLK), smem, stream, p);       \
            else SS_LAUNCH(SS_KERNEL(KERNEL<bf16_t, 4>), grid, dim3(BLK), smem, stream, p);                     \
        } else if (dtype == SS_F32X3) {                                                                         \
            if (dpk == 1) SS_LAUNCH(SS_KERNEL(KERNEL<x3_t, 1>), grid, dim3(BLK), smem, stream, p);              \
            else if (dpk == 2) SS_LAUNCH(SS_KERNEL(KERNEL<x3_t, 2>), grid, dim3(BLK), smem, stream, p);         \
            else if (dpk == 3) SS_LAUNCH(SS_KERNEL(KERNEL<x3_t, 3>), grid, dim3(BLK), smem, stream, p);         \
            else SS_LAUNCH(SS_KERNEL(KERNEL<x3_t, 4>), grid, dim3(BLK), smem, stream, p);                       \
        } else {                                                                                                \
            if (dpk == 1) SS_LAUNCH(SS_KERNEL(KERNEL<float, 1>), grid, dim3(BLK), smem, stream, p);             \
            else if (dpk == 2) SS_LAUNCH(SS_KERNEL(KERNEL<float, 2>), grid, dim3(BLK), smem, stream, p);        \
            else if (dpk == 3) SS_LAUNCH(SS_KERNEL(KERNEL<float, 3>), grid, dim3(BLK), smem, stream, p);        \
            else SS_LAUNCH(SS_KERNEL(KERNEL<float, 4>), grid, dim3(BLK), smem, stream, p);                      \
        }                                                                                                       \
    } while (0)

#if defined(SS_ATTN_RES16)
// ---- resident-path dispatch
#include <stdlib.h>
static const size_t RES_LDS_MAX = 160 * 1024;
// bytes behind the E table of the hand-scheduled forward (which = 3) / query-major backward (4): the per-wave chunk buffers, or the
// farthest read past the table over all tiles of this (T, D) if that is more (those reads only meet masked logits resp. dS = 0)
static size_t res2_tail(int which, int T, int dp, int D) {
    const int nb = (T + 15) / 16, NE = 2 * D - 1;
    int over = 0;
    for (int t = 0; t < nb; ++t) {
        const int q0 = 16 * t;
        const int jlo = q0 - (D - 1) < 0 ? 0 : (q0 - (D - 1)) >> 4;
        int jhi = (q0 + 15 + D - 1) >> 4; jhi = jhi > nb - 1 ? nb - 1 : jhi;
        const int nblk = jhi - jlo + 1, NBLK = nblk <= 4 ? 4 : (nblk <= 10 ? 10 : RES_NB);
        const int first = 16 * jlo - q0 - 15 + (D - 1);
        const int last = which == 3 ? first + 16 * (NBLK + 1) + 15 : first + 32 * ((NBLK + 2) / 2) - 1;
        over = last + 1 - NE > over ? last + 1 - NE : over;
    }
    const size_t bufs = (size_t)(which == 3 ? RES_W_FWD : RES_W_BQ) * 2 * F2_PTB, room = (size_t)over * ((size_t)dp * 2 + RES2_PAD);
    return ((bufs > room ? bufs : room) + 15) / 16 * 16;
}
static size_t res_smem(int which, int T, int dp, int D) {
    const size_t PK = (size_t)dp * 2 + 16, nb = (size_t)(T + 15) / 16, Tr = nb * 16, NE = 2 * (size_t)D - 1, TQ = (nb + 1) / 2 * 32, tile = 16 * RT_LD * 2;
    if (which == 0) return 2 * Tr * PK + (NE + 2 * RES_PL) * PK + RES_W_FWD * tile + 16;
    if (which == 1) return 2 * Tr * PK + (NE + 96) * PK + RES_W_BQ * tile + 16;
    if (which == 3 || which == 4) {     // hand-scheduled forward / query-major backward: [2 tables of Tr rows | E | tail]
        const size_t P2 = (size_t)dp * 2 + RES2_PAD;
        return 2 * Tr * P2 + NE * P2 + res2_tail(which, T, dp, D) + 16;
    }
    if (which == 5) return 2 * TQ * ((size_t)dp * 2 + RES2_PAD) + RES_W_BKV * 4 * (size_t)KV_TILE + (TQ + 7 * 32) * 4 + 16;                              // key-major backward on the P image
    return 2 * TQ * PK + NE * PK + RES_W_BKV * 2 * tile + 8 * TQ + 16;
}
static bool res_enabled(int dtype, int T) {
    if (dtype != SS_BF16 || (T + 15) / 16 > RES_NB) return false;
    const char* e = getenv("SS_ATTN_RESIDENT");           // "0" forces the per-tile kernels (A/B measurements, tests of both paths)
    return !(e && e[0] == '0');
}
static bool fwd2_enabled() {
    const char* e = getenv("SS_ATTN_FWD2");               // "0" keeps the compiler-scheduled resident forward (A/B measurements, tests of both)
    return !(e && e[0] == '0');
}
typedef void (*ResKernel)(AttnP);
static int res_launch(ResKernel k, int slot, int pairs, int waves, size_t smem, void* stream, AttnP p, bool stagger = false, bool per_head = false) {
    // one workgroup per CU at a time: pairs beyond the last full round of #CU are split in two halves when that shortens it
    const int cus = ss_cu_count(4);
    const int rem = pairs % cus;
    const bool split = pairs > cus && rem > 0 && 2 * rem <= cus;
    p.gx = split ? pairs - rem : pairs;
    int blocks = split ? pairs + rem : pairs;
    p.h1 = 0; p.persist = 0;
    if (per_head && pairs > cus && cus % p.H == 0 && cus / p.H >= 2) {          // more than one round: one persistent workgroup per CU, S = cus / H per head (res2_item)
        const char* e = getenv("SS_ATTN_PERSIST");                                // "0": one workgroup per pair (A/B measurements, tests of both)
        if (!(e && e[0] == '0')) { p.persist = cus / p.H; blocks = cus; }
    }
    if (stagger && split && !p.persist) { const char* e = getenv("SS_ATTN_STAGGER"); if (!(e && e[0] == '0')) { p.h1 = 2 * rem < cus / 2 ? 2 * rem : (cus / 2) & ~1; } }
    static size_t granted[48] = {0};
    if (granted[slot] < smem) {
        if (!ss_grant_lds((const void*)k, smem)) { ss_set_error("attention: cannot reserve %zu bytes of LDS", smem); return 1; }
        granted[slot] = smem;
    }
    SS_LAUNCH(k, dim3(blocks), dim3(waves * 64), smem, stream, p);
    return 0;
}
static ResKernel res_pick(int which, int dpk, bool drop = false) {
    static const ResKernel fwd_drop[3] = {attn_fwd_res_kernel<1, true>, attn_fwd_res_kernel<2, true>, attn_fwd_res_kernel<3, true>};
    static const ResKernel fwd2[2][3] = {{attn_fwd_res2_kernel<1, false>, attn_fwd_res2_kernel<2, false>, attn_fwd_res2_kernel<3, false>},
                                         {attn_fwd_res2_kernel<1, true>, attn_fwd_res2_kernel<2, true>, attn_fwd_res2_kernel<3, true>}};
    if (which == 3) return dpk >= 1 && dpk <= 3 ? fwd2[drop ? 1 : 0][dpk - 1] : (ResKernel)0;
    static const ResKernel bq2[3] = {attn_bwd_q2_kernel<1>, attn_bwd_q2_kernel<2>, attn_bwd_q2_kernel<3>}, bkv2[3] = {attn_bwd_kv2_kernel<1>, attn_bwd_kv2_kernel<2>, attn_bwd_kv2_kernel<3>};
    if (which == 4 || which == 5) return dpk >= 1 && dpk <= 3 ? (which == 4 ? bq2 : bkv2)[dpk - 1] : (ResKernel)0;
    if (which == 0 && drop && dpk >= 1 && dpk <= 3) return fwd_drop[dpk - 1];
    static const ResKernel tab[3][3] = {
        {attn_fwd_res_kernel<1, false>, attn_fwd_res_kernel<2, false>, attn_fwd_res_kernel<3, false>},
        {attn_bwd_q_res_kernel<1, false>, attn_bwd_q_res_kernel<2, false>, attn_bwd_q_res_kernel<3, false>},
        {attn_bwd_kv_res_kernel<1, false>, attn_bwd_kv_res_kernel<2, false>, attn_bwd_kv_res_kernel<3, false>}};
    static const ResKernel bq_drop[3] = {attn_bwd_q_res_kernel<1, true>, attn_bwd_q_res_kernel<2, true>, attn_bwd_q_res_kernel<3, true>};
    static const ResKernel bkv_drop[3] = {attn_bwd_kv_res_kernel<1, true>, attn_bwd_kv_res_kernel<2, true>, attn_bwd_kv_res_kernel<3, true>};
    if (drop && dpk >= 1 && dpk <= 3 && which == 1) return bq_drop[dpk - 1];
    if (drop && dpk >= 1 && dpk <= 3 && which == 2) return bkv_drop[dpk - 1];
    return dpk >= 1 && dpk <= 3 ? tab[which][dpk - 1] : (ResKernel)0;
}

#endif  // SS_ATTN_RES16

// Which kernels a problem runs: 0 = per-tile (any T, f32 / bf16 x 3; read the transposed copies qkvT / dOT), 1 = LDS-resident 16 x 16 tiles
// (rounds 1-4, attention.hip), 2 = transposed 32 x 32 score tiles (attention_t.hip; need the prepared embedding tables).
static bool family_t(int dtype, int T, int dp, int D) {
    if (dtype != SS_BF16 || !attn_t_supported(T, dp, D)) return false;
#if defined(SS_ATTN_RES16)
    const char* e = getenv("SS_ATTN_T");                  // A/B builds: "0" keeps the 16 x 16 resident kernels
    if (e && e[0] == '0') return false;
#endif
    const char* r = getenv("SS_ATTN_RESIDENT");           // "0": the per-tile kernels (tests of both paths)
    return !(r && r[0] == '0');
}
extern "C" int ss_relpos_attention_family(int dtype, int T, int dp, int D)
{
    if (family_t(dtype, T, dp, D)) return 2;
    return ss_relpos_attention_needs_transposed(dtype, T, dp, D) ? 0 : 1;
}
extern "C" int64_t ss_relpos_attention_table_bytes(int H, int dp, int D)
{
    (void)D;
    return (H > 0 && dp % 32 == 0 && dp >= 32 && dp <= 96) ? attn_t_table_bytes(H, dp) : 0;
}
extern "C" int ss_relpos_attention_prepare_tables(const float* emb, void* tab, int H, int D, int dh, int dp, float scale, void* stream)
{
    SS_CHECK(emb && tab, "ss_relpos_attention_prepare_tables: null pointer");
    SS_CHECK(H > 0 && D >= 1 && D <= 100 && dh >= 1 && dh <= dp && dp % 32 == 0 && dp <= 96 && scale > 0.f, "ss_relpos_attention_prepare_tables: bad shape");
    if (attn_t_prepare_tables(emb, H, D, dh, dp, scale, tab, stream)) return 1;
    SS_LAUNCH_CHECK("ss_relpos_attention_prepare_tables");
    return 0;
}

// 1 if this problem runs the per-tile kernels (which read the transposed copies qkvT / dOT), 0 if the LDS-resident ones do
extern "C" int ss_relpos_attention_needs_transposed(int dtype, int T, int dp, int D)
{
    if (family_t(dtype, T, dp, D)) return 0;
#if defined(SS_ATTN_RES16)
    const int dpk = dp / 32;
    const bool resident = res_enabled(dtype, T) && dp % 32 == 0 && dpk >= 1 && dpk <= 3 && res_smem(0, T, dp, D) <= RES_LDS_MAX && res_smem(1, T, dp, D) <= RES_LDS_MAX &&
                          res_smem(2, T, dp, D) <= RES_LDS_MAX;
    return resident ? 0 : 1;
#else
    return 1;
#endif
}

// bytes of the P image (see above) the forward can leave for the backward, 0 if this shape does not run the kernels that use one
extern "C" int64_t ss_relpos_attention_saved_bytes(int dtype, int B, int H, int T, int dp, int D)
{
    if (B <= 0 || H <= 0 || T <= 0 || dp % 32 != 0 || dp < 32 || dp > 96 || D < 1 || D > 100) return 0;
    if (family_t(dtype, T, dp, D)) return attn_t_saved_bytes(B, H, T);
#if !defined(SS_ATTN_RES16)
    return 0;
#else
    if (ss_relpos_attention_needs_transposed(dtype, T, dp, D) || !fwd2_enabled()) return 0;
    if (res_smem(3, T, dp, D) > RES_LDS_MAX || res_smem(4, T, dp, D) > RES_LDS_MAX || res_smem(5, T, dp, D) > RES_LDS_MAX) return 0;
    const char* e = getenv("SS_ATTN_SAVE_P");             // "0": backward recomputes the probabilities (A/B measurements, tests of both paths)
    if (e && e[0] == '0') return 0;
    const int64_t nb = (T + 15) / 16;
    return (int64_t)B * H * nb * pimg_slots((int)nb) * 512;
#endif
}

static void attn_t_args(AttnTArgs& a, const void* qkv, const void* tab, int B, int H, int T, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream)
{
    memset(&a, 0, sizeof(a));
    a.qkv = qkv; a.tab = tab; a.B = B; a.H = H; a.T = T; a.dp = dp; a.D = D; a.scale = scale; a.dropout_p = dropout_p; a.seed = seed; a.stream_id = rng_stream;
}

extern "C" int ss_relpos_attention_forward_p(int dtype, const void* qkv, const void* qkvT, const void* E, const void* tab, void* out, float* lse, void* pimg,
                                           int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_forward", dtype, B, H, T, Tp, dp, D, dropout_p)) return 1;
    if (family_t(dtype, T, dp, D)) {
        SS_CHECK(qkv && tab && out && lse, "ss_relpos_attention_forward: null pointer (this shape runs the transposed-score kernels, which read the prepared tables: ss_relpos_attention_prepare_tables)");
        AttnTArgs a; attn_t_args(a, qkv, tab, B, H, T, dp, D, scale, dropout_p, seed, rng_stream);
        a.out = out; a.lse = lse; a.pimg = pimg;
        if (attn_t_forward(a, stream)) return 1;
        SS_LAUNCH_CHECK("ss_relpos_attention_forward");
        return 0;
    }
    SS_CHECK(qkv && E && out && lse, "ss_relpos_attention_forward: null pointer");
    SS_CHECK(qkvT || !ss_relpos_attention_needs_transposed(dtype, T, dp, D), "ss_relpos_attention_forward: this shape runs the per-tile kernels, which need the transposed copy qkvT");
    AttnP p; attn_fill(p, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream);
    p.qkv = qkv; p.qkvT = qkvT; p.E = E; p.out = out; p.lse = lse;
    SS_CHECK(!pimg || ss_relpos_attention_saved_bytes(dtype, B, H, T, dp, D) > 0, "ss_relpos_attention_forward_p: this shape does not save probabilities (ss_relpos_attention_saved_bytes is 0)");
    p.pimg = pimg;
#if defined(SS_ATTN_RES16)
    if (!ss_relpos_attention_needs_transposed(dtype, T, dp, D)) {
        if (fwd2_enabled() && res_smem(3, T, dp, D) <= RES_LDS_MAX) {
            p.tail = (int)res2_tail(3, T, dp, D);
            if (res_launch(res_pick(3, dp / 32, p.drop_thresh != 0), 24 + dp / 32 + (p.drop_thresh ? 4 : 0), B * H, RES_W_FWD, res_smem(3, T, dp, D), stream, p, true, true)) return 1;
        } else if (res_launch(res_pick(0, dp / 32, p.drop_thresh != 0), dp / 32 + (p.drop_thresh ? 12 : 0), B * H, RES_W_FWD, res_smem(0, T, dp, D), stream, p)) return 1;
        SS_LAUNCH_CHECK("ss_relpos_attention_forward");
        return 0;
    }
#endif
    p.gx = ((T + 15) / 16 + 3) / 4;
    dim3 grid(p.gx * H * B);
    SS_ATTN_DISPATCH(attn_fwd_kernel, grid, 256, 0);
    SS_LAUNCH_CHECK("ss_relpos_attention_forward");
    return 0;
}

// ---- the parity-grade mode on the transposed-score kernels: f32 operands as hi / lo bf16 planes (attention_t.hip, "x3")
extern "C" int ss_relpos_attention_x3_supported(int T, int dp, int D) { return attn_t_x3_supported(T, dp, D) ? 1 : 0; }
extern "C" int64_t ss_relpos_attention_x3_saved_bytes(int B, int H, int T, int dp, int D)
{
    return (B > 0 && H > 0 && attn_t_x3_supported(T, dp, D)) ? 2 * attn_t_saved_bytes(B, H, T) : 0;
}
extern "C" int64_t ss_relpos_attention_x3_table_bytes(int H, int dp, int D)
{
    (void)D;
    return (H > 0 && dp % 32 == 0 && dp >= 32 && dp <= 96) ? 2 * attn_t_table_bytes(H, dp) : 0;
}
extern "C" int ss_relpos_attention_x3_prepare_tables(const float* emb, void* tab, int H, int D, int dh, int dp, float scale, void* stream)
{
    SS_CHECK(emb && tab, "ss_relpos_attention_x3_prepare_tables: null pointer");
    SS_CHECK(H > 0 && D >= 1 && D <= 100 && dh >= 1 && dh <= dp && dp % 32 == 0 && dp <= 96 && scale > 0.f, "ss_relpos_attention_x3_prepare_tables: bad shape");
    if (attn_t_prepare_tables_x3(emb, H, D, dh, dp, scale, tab, stream)) return 1;
    SS_LAUNCH_CHECK("ss_relpos_attention_x3_prepare_tables");
    return 0;
}
extern "C" int ss_relpos_attention_x3_forward(const void* qkv_hi, const void* qkv_lo, const void* tab, void* out_hi, void* out_lo, float* lse, void* pimg,
                                              int B, int H, int T, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_x3_forward", SS_BF16, B, H, T, (T + 7) / 8 * 8, dp, D, dropout_p)) return 1;
    SS_CHECK(attn_t_x3_supported(T, dp, D), "ss_relpos_attention_x3_forward: T=%d dp=%d D=%d does not run the plane kernels (ss_relpos_attention_x3_supported)", T, dp, D);
    SS_CHECK(qkv_hi && qkv_lo && tab && out_hi && out_lo && lse, "ss_relpos_attention_x3_forward: null pointer");
    AttnTArgs a; attn_t_args(a, qkv_hi, tab, B, H, T, dp, D, scale, dropout_p, seed, rng_stream);
    a.qkv_lo = qkv_lo; a.out = out_hi; a.out_lo = out_lo; a.lse = lse; a.pimg = pimg;
    if (attn_t_forward_x3(a, stream)) return 1;
    SS_LAUNCH_CHECK("ss_relpos_attention_x3_forward");
    return 0;
}
extern "C" int ss_relpos_attention_x3_backward(const void* qkv_hi, const void* qkv_lo, const void* tab, const void* out_hi, const void* out_lo, const void* dO_hi, const void* dO_lo,
                                               float* Dscratch, void* dqkv_hi, void* dqkv_lo, const void* pimg,
                                               int B, int H, int T, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_x3_backward", SS_BF16, B, H, T, (T + 7) / 8 * 8, dp, D, dropout_p)) return 1;
    SS_CHECK(attn_t_x3_supported(T, dp, D), "ss_relpos_attention_x3_backward: T=%d dp=%d D=%d does not run the plane kernels (ss_relpos_attention_x3_supported)", T, dp, D);
    SS_CHECK(qkv_hi && qkv_lo && tab && out_hi && out_lo && dO_hi && dO_lo && Dscratch && dqkv_hi && dqkv_lo && pimg, "ss_relpos_attention_x3_backward: null pointer");
    AttnTArgs a; attn_t_args(a, qkv_hi, tab, B, H, T, dp, D, scale, dropout_p, seed, rng_stream);
    a.qkv_lo = qkv_lo; a.O = out_hi; a.O_lo = out_lo; a.dO = dO_hi; a.dO_lo = dO_lo; a.Dv = Dscratch; a.dqkv = dqkv_hi; a.dqkv_lo = dqkv_lo; a.pimg = (void*)pimg;
    if (attn_t_backward_x3(a, stream)) return 1;
    SS_LAUNCH_CHECK("ss_relpos_attention_x3_backward");
    return 0;
}

extern "C" int ss_relpos_attention_forward(int dtype, const void* qkv, const void* qkvT, const void* E, const void* tab, void* out, float* lse,
                                           int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_relpos_attention_forward_p(dtype, qkv, qkvT, E, tab, out, lse, nullptr, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream, stream);
}

extern "C" int ss_relpos_attention_backward_p(int dtype, const void* qkv, const void* qkvT, const void* E, const void* ET, const void* tab, const void* out, const float* lse,
                                            const void* dO, const void* dOT, float* Dscratch, void* dqkv, const void* pimg,
                                            int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    if (attn_check("ss_relpos_attention_backward", dtype, B, H, T, Tp, dp, D, dropout_p)) return 1;
    if (family_t(dtype, T, dp, D)) {
        SS_CHECK(qkv && tab && out && dO && Dscratch && dqkv, "ss_relpos_attention_backward: null pointer (this shape runs the transposed-score kernels, which read the prepared tables)");
        SS_CHECK(pimg, "ss_relpos_attention_backward: the transposed-score kernels work from the saved probabilities of the forward (pimg, ss_relpos_attention_saved_bytes)");
        AttnTArgs a; attn_t_args(a, qkv, tab, B, H, T, dp, D, scale, dropout_p, seed, rng_stream);
        a.O = out; a.dO = dO; a.Dv = Dscratch; a.dqkv = dqkv; a.pimg = (void*)pimg;
        if (attn_t_backward(a, stream)) return 1;
        SS_LAUNCH_CHECK("ss_relpos_attention_backward");
        return 0;
    }
    SS_CHECK(qkv && E && ET && out && lse && dO && Dscratch && dqkv, "ss_relpos_attention_backward: null pointer");
    SS_CHECK((qkvT && dOT) || !ss_relpos_attention_needs_transposed(dtype, T, dp, D), "ss_relpos_attention_backward: this shape runs the per-tile kernels, which need qkvT and dOT");
    AttnP p; attn_fill(p, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream);
    p.qkv = qkv; p.qkvT = qkvT; p.E = E; p.ET = ET; p.out = (void*)out; p.lse = (float*)lse; p.dO = dO; p.dOT = dOT; p.Dv = Dscratch; p.dqkv = dqkv;
    // D = rowsum(dO * O) stays a pass of its own (14.5 us, HBM-bound).  Round 4 formed it inside the image-path query-major kernel instead (from the
    // dO fragments of a tile and the matching O rows, handed to the key-major kernel through Dscratch): bit-identical gradients, but that kernel is
    // instruction-issue-bound and the ~200 extra instructions per tile cost it 20 us -- backward 167 -> 173 us at 880 pairs; removed.
    {
        long long blocks = ((long long)B * T + 3) / 4; if (blocks > 8192) blocks = 8192;
        if (dtype == SS_BF16) SS_LAUNCH(attn_dsum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)dO, (const bf16_t*)out, Dscratch, B, H, T, dp);
        else SS_LAUNCH(attn_dsum_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)dO, (const float*)out, Dscratch, B, H, T, dp);
    }
#if defined(SS_ATTN_RES16)
    if (pimg) {
        SS_CHECK(ss_relpos_attention_saved_bytes(dtype, B, H, T, dp, D) > 0, "ss_relpos_attention_backward_p: this shape has no saved probabilities");
        p.pimg = (void*)pimg;
        p.tail = (int)res2_tail(4, T, dp, D);
        if (res_launch(res_pick(4, dp / 32), 32 + dp / 32, B * H, RES_W_BQ, res_smem(4, T, dp, D), stream, p, true, true)) return 1;
        if (res_launch(res_pick(5, dp / 32), 36 + dp / 32, B * H, RES_W_BKV, res_smem(5, T, dp, D), stream, p, true)) return 1;
        SS_LAUNCH_CHECK("ss_relpos_attention_backward_p");
        return 0;
    }
    if (!ss_relpos_attention_needs_transposed(dtype, T, dp, D)) {
        const bool drop = p.drop_thresh != 0;
        if (res_launch(res_pick(1, dp / 32, drop), (drop ? 16 : 4) + dp / 32, B * H, RES_W_BQ, res_smem(1, T, dp, D), stream, p)) return 1;
        if (res_launch(res_pick(2, dp / 32, drop), (drop ? 20 : 8) + dp / 32, B * H, RES_W_BKV, res_smem(2, T, dp, D), stream, p)) return 1;
        SS_LAUNCH_CHECK("ss_relpos_attention_backward");
        return 0;
    }
#else
    SS_CHECK(!pimg, "ss_relpos_attention_backward_p: this shape has no saved probabilities");
#endif
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

extern "C" int ss_relpos_attention_backward(int dtype, const void* qkv, const void* qkvT, const void* E, const void* ET, const void* tab, const void* out, const float* lse,
                                            const void* dO, const void* dOT, float* Dscratch, void* dqkv,
                                            int B, int H, int T, int Tp, int dp, int D, float scale, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_relpos_attention_backward_p(dtype, qkv, qkvT, E, ET, tab, out, lse, dO, dOT, Dscratch, dqkv, nullptr, B, H, T, Tp, dp, D, scale, dropout_p, seed, rng_stream, stream);
}
