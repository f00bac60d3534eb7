// gemm_common.h -- pieces shared by the MFMA GEMM kernels (gemm.hip: 128x128 tiles, gemm8.hip: 8-wave 256-wide tiles):
// epilogue descriptor, LDS swizzle, elementwise epilogue + LDS-staged 16-byte row stores.
#pragma once
#include "common.h"
#include <math.h>

enum { OP_KC = 0, OP_OC = 1 };
#ifndef SS_GEMM_PRIO
#define SS_GEMM_PRIO 0
#endif
constexpr int BM = 128, BN = 128, ROWB = 128;   // ROWB: bytes of K per tile row

struct GemmEpi {
    const float* bias;       // [N] or null
    const void* gate;        // TO-typed, addressed like C; out = gate>0 ? out*gate_scale : 0
    float gate_scale;
    float alpha;
    int relu;
    unsigned drop_thresh;    // 0 = none
    float drop_scale;
    unsigned long long seed;
    unsigned stream;
    int mode;                // 0 store, 1 accumulate (C += v), 2 atomicAdd (f32 out only)
    RowMap cmap;
    int col_mod, col_mul, col_div_mul;   // output column permutation: c -> (c % col_mod)*col_mul + (c / col_mod)*col_div_mul
    float log_clamp;         // > 0: v = log(max(v, log_clamp))   (data_utils.py:29-30)
    void* c2;                // optional second copy of the result at rowmap2(row) + col*col_stride2 (transposed layouts)
    RowMap cmap2;
    long long col_stride2;
    int fast;                // 1: LDS-staged, 16-byte coalesced output path (host decides)
    int c2_pack;             // 1: the 4 rows a lane holds are consecutive, aligned elements of c2 -> one packed store
    int general;             // 1: dropout or log-clamp in the epilogue
    int c2_lds;              // 1: the transposed copy can leave through LDS as 16-byte stores (bf16, 8-row aligned sequences)
    float* col_sum;          // optional column statistics of the stored result (8-wave kernel only): += sum_m (C - shift), += sum_m (C - shift)^2
    float* col_sumsq;
    const float* col_shift;
    int debug;               // tuning experiments only (SS_GEMM_DEBUG): 1 skip flush stores, 2 skip stage, 4 skip MFMA
    void* planes_hi;         // plane GEMMs (8-wave kernel, f32 out): the stored value also leaves as hi / lo bf16 planes addressed like C
    void* planes_lo;
    int planes_only;         // ... and C itself is not written
    unsigned char* sign_out; // 8-wave register epilogue (bf16 out): byte [row * sign_pitch + col / 8], bit col % 8 = [stored value > 0]
    long long sign_pitch;
    const unsigned char* gate_bits;      // 8-wave C-piece epilogue: the gate as such bits instead of the saved tensor
    long long gate_bits_pitch;
};

template <class T> struct Elem;
template <> struct Elem<float> { static constexpr int EPC = 4; static constexpr int BK = 32; };
template <> struct Elem<bf16_t> { static constexpr int EPC = 8; static constexpr int BK = 64; };

// 16-byte chunk c of tile row r lives at chunk position c ^ s(r), s(r) = (r & 7) ^ (2 * ((r >> 3) & 3)):
//  * MFMA fragment reads (16 consecutive rows, two adjacent chunks per 16-lane service group) stay conflict-free
//    (bit 0 of s equals bit 0 of r, so a q=0 lane and a q=1 lane can never meet on one slot);
//  * the transposing (OC) stores write rows 8k+o / 4k+o at fixed o: the (r >> 3) term spreads them over 4 / 8 chunk
//    positions instead of one (was a 16-way bank conflict).
__device__ __forceinline__ unsigned swz(int row, int chunk) { return (unsigned)row * ROWB + (unsigned)((chunk ^ (row & 7) ^ (((row >> 3) & 3) << 1)) << 4); }

__device__ __forceinline__ void out_add(float* p, float v, int mode) {
    if (mode == 2) atomicAdd(p, v); else if (mode == 1) *p += v; else *p = v;
}
__device__ __forceinline__ void out_add(bf16_t* p, float v, int mode) {
    if (mode == 1) *p = f2bf(bf2f(*p) + v); else *p = f2bf(v);
}

// one 16x16 accumulator tile: this lane holds rows row0..row0+3 of column col
template <class TO>
__device__ __forceinline__ void epilogue_tile(const f32x4& a, TO* __restrict__ C, const GemmEpi& epi, int row0, int col, int M, int N)
{
    if (col >= N) return;
    const float bias = epi.bias ? epi.bias[col] : 0.f;
    const int pcol = epi.col_mod ? (col % epi.col_mod) * epi.col_mul + (col / epi.col_mod) * epi.col_div_mul : col;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = row0 + reg;
        if (row < M) {
            const long long off = rowmap_off(epi.cmap, row) + pcol;
            float v = a[reg] * epi.alpha + bias;
            if (epi.relu) v = fmaxf(v, 0.f);
            if (epi.drop_thresh)     // element (row, col) <-> element row * N + col of the stream (common.h: dropout_keep1)
                v = dropout_keep1(epi.seed, epi.stream, (unsigned long long)row * (unsigned)N + col, epi.drop_thresh) ? v * epi.drop_scale : 0.f;
            if (epi.gate) v = ldf((const TO*)epi.gate + off) > 0.f ? v * epi.gate_scale : 0.f;
            if (epi.log_clamp > 0.f) v = logf(fmaxf(v, epi.log_clamp));
            out_add(C + off, v, epi.mode);
            if (epi.c2) stf((TO*)epi.c2 + rowmap_off(epi.cmap2, row) + (long long)col * epi.col_stride2, v);
        }
    }
}

// ---- fast epilogue, phase 1: elementwise part in registers, result into the LDS C tile (and the packed c2 copy)
// GENERAL = false: alpha/bias/ReLU only (the common case; the compiler would otherwise if-convert the uniform
// dropout / log-clamp branches into per-element selects and evaluate Philox and v_log for every element).
template <class TO, int GENERAL, bool C2L = false>     // GENERAL: 0 alpha/bias/ReLU, 1 + dropout, 2 + log-clamp
__device__ __forceinline__ void epilogue_stage(const f32x4& a, TO* __restrict__ ct, int ldc, int lrow0, int lcol, const GemmEpi& epi, int row0, int col, int M, int N,
                                               TO* __restrict__ tt = nullptr, int ldt = 0)
{
    float v[4];
    const float bias = (epi.bias && col < N) ? epi.bias[col] : 0.f;
    const float lo = epi.relu ? 0.f : -INFINITY;
    bool kp[4] = {true, true, true, true};
    if (GENERAL == 1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) kp[reg] = dropout_keep1(epi.seed, epi.stream, (unsigned long long)(row0 + reg) * (unsigned)N + col, epi.drop_thresh);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float x = fmaxf(a[reg] * epi.alpha + bias, lo);
        if (GENERAL == 1) x = kp[reg] ? x * epi.drop_scale : 0.f;
        if (GENERAL == 2) x = logf(fmaxf(x, epi.log_clamp));
        v[reg] = x;
        stf(ct + (lrow0 + reg) * ldc + lcol, x);
    }
    if (C2L) {     // transposed copy goes through LDS too: [col][row], 4 consecutive rows = one 8-byte store (bf16)
        u32x2 w; w[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); w[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *(u32x2*)(tt + lcol * ldt + lrow0) = w;
    } else if (epi.c2 && col < N) {
        TO* c2 = (TO*)epi.c2 + (long long)col * epi.col_stride2;
        if (epi.c2_pack && row0 + 3 < M) {
            TO* p = c2 + rowmap_off(epi.cmap2, row0);
            if (sizeof(TO) == 2) { u32x2 w; w[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); w[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16); *(u32x2*)p = w; }
            else { f32x4 w = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = w; }
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) if (row0 + reg < M) stf(c2 + rowmap_off(epi.cmap2, row0 + reg), v[reg]);
        }
    }
}

// ---- fast epilogue, phase 2: 16-byte row-contiguous stores from the LDS C tile
template <class TO> struct OutVec;
template <> struct OutVec<bf16_t> { static constexpr int N = 8; };
template <> struct OutVec<float> { static constexpr int N = 4; };
__device__ __forceinline__ void outvec_load(const bf16_t* p, float (&v)[8]) { Vec8<bf16_t>::load(p, v); }
__device__ __forceinline__ void outvec_store(bf16_t* p, const float (&v)[8]) { Vec8<bf16_t>::store(p, v); }
__device__ __forceinline__ void outvec_load(const float* p, float (&v)[4]) { f32x4 a = *(const f32x4*)p; v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
__device__ __forceinline__ void outvec_store(float* p, const float (&v)[4]) { f32x4 a = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = a; }

template <class TO, int NTHR = 256, int TBN = BN>
__device__ __forceinline__ void epilogue_flush(const TO* __restrict__ ct, int ldc, TO* __restrict__ C, const GemmEpi& epi, int row_base, int nrows, int n0, int M, int N, int tid)
{
    constexpr int EV = OutVec<TO>::N;
    constexpr int CPR = TBN / EV;                      // 16-byte chunks per tile row
    const int total = nrows * CPR;
    for (int idx = tid; idx < total; idx += NTHR) {
        const int r = idx / CPR, ch = idx - r * CPR;
        const int row = row_base + r, col = n0 + ch * EV;
        if (row < M && col < N) {
            float v[EV];
            outvec_load(ct + r * ldc + ch * EV, v);
            const long long off = rowmap_off(epi.cmap, row) + col;
            if (epi.gate) {
                float g[EV]; outvec_load((const TO*)epi.gate + off, g);
#pragma unroll
                for (int e = 0; e < EV; ++e) v[e] = g[e] > 0.f ? v[e] * epi.gate_scale : 0.f;
            }
            if (epi.mode == 1) {
                float o[EV]; outvec_load(C + off, o);
#pragma unroll
                for (int e = 0; e < EV; ++e) v[e] += o[e];
            }
            outvec_store(C + off, v);
        }
    }
}

// 8-wave 256-column-tile kernels (gemm8.hip); ni = 8 or 9 (256 / 288 tile rows), pin = scheduling fences on/off
template <class TO>
int gemm8_launch_kc(int ni, int pin, const void* A, const void* B, void* C, int M, int N, int K, const RowMap& am, const RowMap& bm, const GemmEpi& epi, void* stream,
                    bool planes = false, long long a_lo = 0, long long b_lo = 0);

// gemm_smallk.hip: K <= 32 (first convolution and its residual projection), no LDS
bool gemm_smallk_ok(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* A, const void* B, const void* C, int M, int N, int K,
                    const RowMap& am, const RowMap& bm, const GemmEpi& epi, int split_k);
int gemm_smallk_launch(const void* A, const void* B, void* C, int M, int N, int K, const RowMap& am, const RowMap& bm, const GemmEpi& epi, void* stream);

