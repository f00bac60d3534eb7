// gemm_smallk.hip -- C[m][n] = sum_k A(m,k) B(n,k) for K <= 32 (gfx950, bf16): the first convolution of the model (8 EMG channels x 3
// taps = K 24, architecture.py:18 in the first ResBlock) and its 1x1 residual projection (K = 8, architecture.py:27), 88 000 rows x 768
// columns each at the reference batch.  With so short a contraction the GEMM is an HBM-bound WRITE of C (135 MB per call): the tiled
// MFMA kernels spent 73 us on it (1.85 TB/s) between their first-tile latency and LDS-staged epilogues, plus a separate pass over C
// for the BatchNorm statistics.  Here the operands never touch LDS:
//   * one wave owns 64 output columns for a run of 16-row slabs; its 4 weight fragments (the MFMA A operands, K zero-padded to 32)
//     stay in registers, a slab costs ONE 16-byte load per lane (its row of A: the overlapping k=3 window is contiguous) and 4 MFMAs;
//   * the weights enter with their rows permuted (A-operand row r of MFMA j <- weight row n0 + 16 (r / 4) + 4 j + r % 4), so a lane
//     ends up with 16 CONSECUTIVE columns of one output row; the 4 lanes that share a row are 16 lanes apart, though, and a store
//     instruction that scatters 16-byte pieces over 16 rows ran at 1.3 TB/s -- so the 2 KB slab takes one trip through a per-wave
//     LDS tile (2 ds_write_b128 + 2 ds_read_b128 per lane) and leaves with every 4 ADJACENT lanes writing 64 contiguous bytes;
//   * bias / ReLU and the optional column statistics (sum and sum of squares of the stored values minus a shift: BatchNorm batch
//     statistics) ride along in registers.  The 16 waves of a workgroup work on the SAME 64 columns, so their sums meet in LDS and a
//     workgroup issues 128 atomics: with 4-wave workgroups on different columns the 0.5 M same-line atomics of a call took 130 us.
#include "common.h"
#include "gemm_common.h"
#include "silent_speech_hip.h"

namespace {
constexpr int SK_WAVES = 16;             // waves per workgroup, ALL on the same 64 columns (different slabs): one set of statistics atomics per workgroup
constexpr int SK_SLABS = 8;              // 16-row slabs per wave

__device__ __forceinline__ float sk_row16_sum(float v) {
#if defined(SS_EMU)
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
#else
#define SS_SK_DPP(CTRL) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true))
    SS_SK_DPP(0xB1); SS_SK_DPP(0x4E); SS_SK_DPP(0x141); SS_SK_DPP(0x140);
#undef SS_SK_DPP
    return v;
#endif
}

// NJ = MFMAs per slab = 16-column groups per wave: 4 (64 columns, 16 per lane) for the plain kernel; the statistics variant carries
// 3 more values per column in registers and runs NJ = 2 (32 columns per wave) to stay under the 128 registers of a 16-wave workgroup
template <bool STATS, int NJ>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_smallk_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K,
                                                          RowMap am, RowMap bm, GemmEpi epi)
{
    constexpr int NC = 4 * NJ, ROWB = NC * 4 * 2, PITCH = ROWB + 16;      // columns per lane, bytes per tile row, LDS pitch (conflict-free 16-byte rows)
    __shared__ __attribute__((aligned(16))) unsigned char tiles[SK_WAVES][16 * PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NJ);
    unsigned char* tw = tiles[w] + c * PITCH + g * (NC * 2);           // this lane's NC columns of row c on the way in
    const int orow = lane >> 2, och = lane & 3;                        // on the way out: row lane / 4, 16-byte chunks och (and 4 + och)
    const unsigned char* tr = tiles[w] + orow * PITCH + och * 16;
    const bool kok = 8 * g < K;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 wf[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int n = n0 + NC * (c >> 2) + 4 * jj + (c & 3);
        wf[jj] = kok ? *(const bf16x8*)(B + rowmap_off(bm, n) + 8 * g) : zero8;
    }
    // this lane's NC columns: bias / shift as 16-byte loads (element-wise conditional loads became 32 serialised round trips)
    float bias[NC], sh[NC], cs[NC], cq[NC];
    {
        f32x4 bv[NJ], sv[NJ];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NJ; ++i) { bv[i] = z4; sv[i] = z4; }
        if (epi.bias) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) bv[i] = *(const f32x4*)(epi.bias + n0 + NC * g + 4 * i);
        }
        if (STATS && epi.col_shift) {
#pragma unroll
            for (int i = 0; i < NJ; ++i) sv[i] = *(const f32x4*)(epi.col_shift + n0 + NC * g + 4 * i);
        }
#pragma unroll
        for (int e = 0; e < NC; ++e) { bias[e] = bv[e >> 2][e & 3]; sh[e] = sv[e >> 2][e & 3]; cs[e] = 0.f; cq[e] = 0.f; }
    }
    const float lo = epi.relu ? 0.f : -INFINITY, alpha = epi.alpha;
    const int nslabs = (M + 15) >> 4, s0 = (blockIdx.y * SK_WAVES + w) * SK_SLABS, s1 = s0 + SK_SLABS < nslabs ? s0 + SK_SLABS : nslabs;
    for (int s = s0; s < s1; s += 4) {
        bf16x8 xf[4]; int mrow[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                  // 4 slabs of loads in flight
            mrow[u] = (s + u) * 16 + c;
            const bool ok = s + u < s1 && mrow[u] < M && kok;
            xf[u] = ok ? *(const bf16x8*)(A + rowmap_off(am, mrow[u]) + 8 * g) : zero8;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (s + u >= s1) break;                                    // uniform
            unsigned pk[NC / 2];
            {
                float v[NC];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 d = mfma_bf16_16x16x32(wf[jj], xf[u], z);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) v[4 * jj + reg] = fmaxf(d[reg] * alpha + bias[4 * jj + reg], lo);
                }
#pragma unroll
                for (int e = 0; e < NC / 2; ++e) pk[e] = pack_bf16(v[2 * e], v[2 * e + 1]);
                if (STATS && mrow[u] < M) {                            // of the STORED (rounded) values
#pragma unroll
                    for (int e = 0; e < NC; ++e) {
                        const float x = __uint_as_float((e & 1) ? (pk[e >> 1] & 0xffff0000u) : (pk[e >> 1] << 16)) - sh[e];
                        cs[e] += x; cq[e] += x * x;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NJ / 2; ++i) { const u32x4 r = {pk[4 * i], pk[4 * i + 1], pk[4 * i + 2], pk[4 * i + 3]}; *(u32x4*)(tw + 16 * i) = r; }
            wave_lds_sync();
            {
                const int om = (s + u) * 16 + orow;
                u32x4 q[NJ / 2];
#pragma unroll
                for (int i = 0; i < NJ / 2; ++i) q[i] = *(const u32x4*)(tr + 64 * i);
                if (om < M) {
                    bf16_t* o = C + rowmap_off(epi.cmap, om) + n0 + och * 8;
#pragma unroll
                    for (int i = 0; i < NJ / 2; ++i) *(u32x4*)(o + 32 * i) = q[i];
                }
            }
            wave_lds_sync();
        }
    }
    if (STATS) {
        float* red = (float*)&tiles[0][0];                             // [2][SK_WAVES][16 NJ] floats, inside the hand-over tiles
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NC; ++e) {
            const float a = sk_row16_sum(cs[e]), q = sk_row16_sum(cq[e]);
            if (c == 0) { red[w * (16 * NJ) + NC * g + e] = a; red[(SK_WAVES + w) * (16 * NJ) + NC * g + e] = q; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * 16 * NJ) {
            const int which = threadIdx.x / (16 * NJ), col = threadIdx.x - which * (16 * NJ);
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < SK_WAVES; ++ww) t += red[(which * SK_WAVES + ww) * (16 * NJ) + col];
            if (which == 0) atomicAdd(epi.col_sum + n0 + col, t); else if (epi.col_sumsq) atomicAdd(epi.col_sumsq + n0 + col, t);
        }
    }
}
}  // namespace

// legality: bf16 in and out, both operands K-contiguous, K = 8 / 16 / 24 / 32, N a multiple of 64, 16-byte aligned rows everywhere, an
// epilogue of alpha / bias / ReLU (+ column statistics) that stores (no accumulate, gate, dropout, log, permutation or second copy)
bool gemm_smallk_ok(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* A, const void* B, const void* C, int M, int N, int K,
                    const RowMap& am, const RowMap& bm, const GemmEpi& epi, int split_k)
{
    if (dtype_in != SS_BF16 || dtype_out != SS_BF16 || a_mode != SS_OP_KC || b_mode != SS_OP_KC || split_k > 1) return false;
    if (K < 8 || K > 32 || (K & 7) || N < 64 || (N & 63) || M < 1) return false;
    if (epi.gate || epi.drop_thresh || epi.mode != 0 || epi.col_mod || epi.log_clamp > 0.f || epi.c2) return false;
    auto al = [](const RowMap& m) { return !(m.base & 7) && !(m.batch_stride & 7) && !(m.row_stride & 7); };
    if (!al(am) || !al(bm) || !al(epi.cmap)) return false;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15) || ((uintptr_t)epi.bias & 15) || ((uintptr_t)epi.col_shift & 15)) return false;
    return true;
}

int gemm_smallk_launch(const void* A, const void* B, void* C, int M, int N, int K, const RowMap& am, const RowMap& bm, const GemmEpi& epi, void* stream)
{
    const int gy = (((M + 15) >> 4) + SK_WAVES * SK_SLABS - 1) / (SK_WAVES * SK_SLABS);
    const dim3 block(SK_WAVES * 64);
    if (epi.col_sum) SS_LAUNCH(SS_KERNEL(gemm_smallk_kernel<true, 2>), dim3(N / 32, gy), block, 0, stream, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, am, bm, epi);
    else SS_LAUNCH(SS_KERNEL(gemm_smallk_kernel<false, 4>), dim3(N / 64, gy), block, 0, stream, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, am, bm, epi);
    return 0;
}
