// gemm8_dw.hip -- grouped weight-gradient GEMM (every dW = dY^T X of the training step) for gfx950, bf16 in / f32 out.
//
//   gemm8_dw_kernel  C[m][n] (+)= sum_k A(k,m) B(k,n), both operands outer-contiguous, reduction over the B*T frames, GROUPED: one
//       persistent launch walks the (tile x K-slice) items of up to 8 problems, so the 4 weight gradients of an encoder layer
//       (108 tiles) need a split of 2, not 7 per GEMM -> 3.5x fewer f32 atomics.
//   gemm8_dwk_kernel (round 4, the default): the tiles are transposed on their way INTO LDS -- every thread owns an 8 (frames) x 8
//       (outer columns) block per K tile, 8 x 8-transposes it in registers (v_perm_b32) and stores K-contiguous 16-byte chunks
//       ([256 outer][64 k], pitch 144 B) -- so the main loop reads its fragments with ds_read_b128 exactly like the KC kernel
//       (24 reads per 64 MFMAs and wave, issued from asm one phase ahead, one counted wait per phase).
//   gemm8_dw_kernel (rounds 2-3; kept for frame maps whose batches are not multiples of 8 frames): tiles copied untransposed
//       ([64 k][256 m], pitch 544 B) and transposed by ds_read_b64_tr_b16 -- 24 half-rate transposing reads per 32 MFMAs, which
//       pinned it at 85 % of the 971 TFLOP/s that traffic allows (profiles/r03_mfma_peak.txt).
#include "gemm8_common.h"

struct DwJob {
    const bf16_t* A; const bf16_t* B; float* C;
    RowMap amap, bmap;                  // frame k -> element offset of its row (equal rows_per_batch)
    long long ldc;
    int M, N, K;                        // C is M x N, reduction over K frames
    int tiles_n, ntiles, split, k_chunk, item0, nitem, atomic;
};
constexpr int DW_MAX_JOBS = 8;
struct DwJobs { DwJob job[DW_MAX_JOBS]; int n; };

namespace g8 {
constexpr int TRP = 544;                // LDS row pitch of the untransposed [64 k][256 m] tiles: 512 B + 32 B (rows 8 banks apart)
constexpr int TR_STAGE = 2 * 64 * TRP;  // A tile + B tile

// Per thread: 16-byte chunk c of k rows r0 + 16 i (i < 4) of every 64-row K tile, for A and for B.  The frame -> row maps of
// both operands cut the frames into batches of the same length, so ONE (frame-in-batch) counter per row serves both; the
// element offsets advance incrementally (no division, no 64-bit multiply in the loop).  Outer columns beyond the matrix are
// clamped to column 0 (their products land in C entries that are never stored); frames beyond k_end must read as zero in
// both operands and only occur in the last K tile of an item (predicated variant).
template <int NR>                                          // NR = rows per thread and K tile: 4 (512 threads) or 8 (256 threads)
struct StageTR8 {
    static constexpr int RS = 64 / NR;                     // row stride between a thread's rows
    const unsigned char* pa; const unsigned char* pb;      // operand bases (+ this thread's outer column), bytes
    unsigned offa[NR], offb[NR];                           // byte offsets of the NR rows
    int tt[NR];                                            // frame index inside its batch
    int c, r0, rpb, wraps;
    unsigned stepa, stepb, wrapa, wrapb;                   // byte advance per 64 frames / extra advance per batch wrap
    __device__ __forceinline__ void init(const DwJob& J, int m0, int n0, int k_begin, int tid) {
        c = tid & 31; r0 = tid >> 5;
        const int ca = m0 + c * 8 < J.M ? m0 + c * 8 : 0, cb = n0 + c * 8 < J.N ? n0 + c * 8 : 0;
        pa = (const unsigned char*)(J.A + J.amap.base + ca); pb = (const unsigned char*)(J.B + J.bmap.base + cb);
        rpb = J.amap.rows_per_batch;
        wraps = rpb >= BK8 ? 1 : (BK8 + rpb - 1) / rpb;
        stepa = (unsigned)(J.amap.row_stride * BK8 * 2); stepb = (unsigned)(J.bmap.row_stride * BK8 * 2);
        wrapa = (unsigned)((J.amap.batch_stride - (long long)rpb * J.amap.row_stride) * 2);
        wrapb = (unsigned)((J.bmap.batch_stride - (long long)rpb * J.bmap.row_stride) * 2);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int rr = k_begin + r0 + RS * i;
            int b_ = 0, t_ = rr;
            if (rpb != 0x7fffffff) { b_ = rr / rpb; t_ = rr - b_ * rpb; }
            tt[i] = t_;
            offa[i] = (unsigned)(((long long)b_ * J.amap.batch_stride + (long long)t_ * J.amap.row_stride) * 2);
            offb[i] = (unsigned)(((long long)b_ * J.bmap.batch_stride + (long long)t_ * J.bmap.row_stride) * 2);
        }
    }
    __device__ __forceinline__ void advance() {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            tt[i] += BK8; offa[i] += stepa; offb[i] += stepb;
            for (int w = 0; w < wraps; ++w) { const bool over = tt[i] >= rpb; tt[i] -= over ? rpb : 0; offa[i] += over ? wrapa : 0u; offb[i] += over ? wrapb : 0u; }
        }
    }
    template <bool PRED>
    __device__ __forceinline__ void load(int k0, int kend, u32x4 (&ra)[NR], u32x4 (&rb)[NR]) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (PRED) {
                u32x4 z = {0u, 0u, 0u, 0u};
                const bool v = k0 + r0 + RS * i < kend;
                ra[i] = v ? *(const u32x4*)(pa + offa[i]) : z;
                rb[i] = v ? *(const u32x4*)(pb + offb[i]) : z;
            } else {
                ra[i] = *(const u32x4*)(pa + offa[i]);
                rb[i] = *(const u32x4*)(pb + offb[i]);
            }
        }
    }
    __device__ __forceinline__ void store(unsigned char* stage, const u32x4 (&ra)[NR], const u32x4 (&rb)[NR]) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { *(u32x4*)(stage + (r0 + RS * i) * TRP + c * 16) = ra[i]; *(u32x4*)(stage + (64 + r0 + RS * i) * TRP + c * 16) = rb[i]; }
    }
};
__device__ __forceinline__ bf16x8 tr_frag8(const unsigned char* tile, int sub, int kk, int c, int q) {
    const unsigned char* a0 = tile + (kk * 32 + q * 4 + (c >> 2)) * TRP + sub * 32 + (c & 3) * 8;
    const s16x4 lo = lds_read_tr16(a0), hi = lds_read_tr16(a0 + 16 * TRP);
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
}  // namespace g8

// NW = 8: waves 2 (M) x 4 (N), 128 x 64 per wave, two waves per SIMD.  NW = 4: waves 2 x 2, 128 x 128 per wave (256 accumulator
// registers: one wave per SIMD on the unified 512-entry file): a third less LDS fragment traffic per MFMA (96 -> 64 transposing
// reads per 128 MFMAs).
template <int PIN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm8_dw_kernel(DwJobs jobs, int nitems, int xorder)
{
    constexpr int NWN = NW / 2, NJ = 16 / NWN, NR = 2048 / (NW * 64);
    using namespace g8;
    SS_DYN_SMEM(lds_raw);
    unsigned char* lds = (unsigned char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
#if defined(SS_EMU)
    const int wave = tid >> 6;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int wm = wave / NWN, wn = wave % NWN, G = gridDim.x;
    StageTR8<NR> st;
    u32x4 ra[NR], rb[NR];
    int it = blockIdx.x, ji = 0, m0, n0, k_begin, k_end, nsteps;

#define G8_SETUP()                                                                                                          \
    do {                                                                                                                     \
        /* the 8 XCDs have private L2s and workgroup b runs on XCD b % 8: hand every XCD a CONTIGUOUS range of this round's items \
           (same problem, same K slice, neighbouring tiles = shared 256-column operand blocks), so that the re-reads of a block  \
           by the tiles of a tile row / column hit that XCD's L2 (was: every item fetched both of its operands from memory,      \
           2.1 GB per launch for 0.54 GB of data) */                                                                           \
        const int ch0_ = it / G * G, pos_ = it - ch0_;                                                                       \
        int R_ = nitems - ch0_; R_ = R_ > G ? G : R_;                                                                        \
        const int xq_ = R_ >> 3, xr_ = R_ & 7, xcd_ = pos_ & 7;                                                             \
        const int lit = xorder ? ch0_ + (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (pos_ >> 3) : it; \
        ji = 0;                                                                                                              \
        while (ji + 1 < jobs.n && lit >= jobs.job[ji].item0 + jobs.job[ji].nitem) ++ji;                                     \
        const DwJob& J_ = jobs.job[ji];                                                                                      \
        const int local = lit - J_.item0, z = local / J_.ntiles, tile = local - z * J_.ntiles;                               \
        const int mt_ = tile / J_.tiles_n, nt_ = tile - mt_ * J_.tiles_n;                                                    \
        m0 = mt_ * 256; n0 = nt_ * 256;                                                                                      \
        k_begin = z * J_.k_chunk; k_end = min(J_.K, k_begin + J_.k_chunk);                                                   \
        nsteps = (k_end - k_begin + BK8 - 1) / BK8;                                                                          \
        st.init(J_, m0, n0, k_begin, tid);                                                                                   \
        if (nsteps == 1) st.template load<true>(k_begin, k_end, ra, rb); else if (nsteps > 1) st.template load<false>(k_begin, k_end, ra, rb); \
    } while (0)

    G8_SETUP();
    for (;;) {
        f32x4 acc[8][NJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        if (nsteps > 0) st.store(lds, ra, rb);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int cur = s & 1;
            const bool more = s + 1 < nsteps;
            if (more) {
                st.advance();
                const int k0 = k_begin + (s + 1) * BK8;
                if (s + 2 == nsteps) st.template load<true>(k0, k_end, ra, rb); else st.template load<false>(k0, k_end, ra, rb);     // only the last K tile can be ragged
            }
            const unsigned char* As = lds + cur * TR_STAGE;
            const unsigned char* Bs = As + 64 * TRP;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[8], b[NJ];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = tr_frag8(As, wm * 8 + i, kk, c, q);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = tr_frag8(Bs, wn * NJ + j, kk, c, q);
                if (PIN) sched_fence();
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
                if (PIN) sched_fence();
            }
            if (more) st.store(lds + (cur ^ 1) * TR_STAGE, ra, rb);
            __syncthreads();
        }
        // ---- next item: its first global loads fly during this item's epilogue
        const DwJob& J = jobs.job[ji];
        float* Cj = J.C; const long long ldc = J.ldc; const int Mj = J.M, Nj = J.N, atomic = J.atomic, cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) { it += G; G8_SETUP(); }
        // ---- epilogue: lane holds rows q*4+reg, column c of each 16x16 tile
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = cn0 + (wn * NJ + j) * 16 + c, row0 = cm0 + (wm * 8 + i) * 16 + q * 4;
                if (col < Nj) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (row0 + reg < Mj) {
                            float* dst = Cj + (long long)(row0 + reg) * ldc + col;
                            if (atomic) atomicAdd(dst, acc[i][j][reg]); else *dst += acc[i][j][reg];
                        }
                    }
                }
            }
        if (!has_next) break;
    }
#undef G8_SETUP
}

// ================================================================ K-contiguous LDS tiles (transpose at write time)
namespace g8 {
constexpr int KP = 144;                      // LDS row pitch: 128 B (64 frames of one outer row) + 16 B.  Fragment reads: 16 consecutive rows, one
                                             // chunk -> 16-byte slots (9 row + chunk) mod 16, all distinct.  Block writes: see StageKT.
constexpr int KT_STAGE = 512 * KP;           // A tile (256 outer rows) + B tile

__device__ __forceinline__ unsigned perm_lo(unsigned hi_src, unsigned lo_src) {         // {lo16(lo_src), lo16(hi_src)}
#if defined(SS_EMU)
    return (lo_src & 0xffffu) | (hi_src << 16);
#else
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u);
#endif
}
__device__ __forceinline__ unsigned perm_hi(unsigned hi_src, unsigned lo_src) {         // {hi16(lo_src), hi16(hi_src)}
#if defined(SS_EMU)
    return (lo_src >> 16) | (hi_src & 0xffff0000u);
#else
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
#endif
}

// Per thread and K tile: ONE 8 x 8 block of ONE operand -- frames 8 r .. 8 r + 7 (r = lane & 7) of outer columns 8 c .. 8 c + 7
// (c = 8 * (wave & 3) + (lane >> 3)); waves 0..3 copy A, waves 4..7 copy B (wave-uniform operand).  A load instruction of a wave
// therefore touches 8 frames x 128 contiguous bytes (whole cache lines), and a 16-lane group of a block WRITE (same outer row
// index inside the block, r = 0..7, two neighbouring c) lands on 16-byte slots (8 c + 9 row + r) mod 16: conflict-free.
// Requires batches of a multiple of 8 frames (a block never straddles a batch), of at least 64 frames (one wrap per K tile at most),
// and K slices that start at multiples of 64.
struct StageKT {
    const unsigned char* p;                  // operand base + this thread's outer column chunk (bytes)
    unsigned off, step, wrap, rs;            // block's first frame (byte offset) / advance per 64 frames / extra per batch wrap / row stride (bytes)
    int tt, rpb, kb;                         // frame inside its batch, frames per batch, block's first frame inside a K tile
    unsigned wr;                             // LDS byte offset inside a stage: (operand, outer row 8 c, chunk r)
    __device__ __forceinline__ void init(const DwJob& J, int opsel, int m0, int n0, int k_begin, int wave, int lane) {
        const int r = lane & 7, c = (wave & 3) * 8 + (lane >> 3);
        const RowMap& map = opsel ? J.bmap : J.amap;
        const int outer = opsel ? n0 : m0, lim = opsel ? J.N : J.M;
        const int col = outer + c * 8 < lim ? outer + c * 8 : 0;         // beyond the matrix: column 0 (products land in C entries that are never stored)
        p = (const unsigned char*)((opsel ? J.B : J.A) + map.base + col);
        rpb = map.rows_per_batch; kb = 8 * r;
        rs = (unsigned)(map.row_stride * 2); step = (unsigned)(map.row_stride * BK8 * 2);
        wrap = (unsigned)((map.batch_stride - (long long)rpb * map.row_stride) * 2);
        const int rr = k_begin + kb;
        int b_ = 0, t_ = rr;
        if (rpb != 0x7fffffff) { b_ = rr / rpb; t_ = rr - b_ * rpb; }
        tt = t_;
        off = (unsigned)(((long long)b_ * map.batch_stride + (long long)t_ * map.row_stride) * 2);
        wr = (unsigned)((opsel * 256 + c * 8) * KP + r * 16);
    }
    __device__ __forceinline__ void advance() {
        tt += BK8; off += step;                                          // batches hold >= 64 frames (host-side condition): at most one wrap, no loop
        const bool over = tt >= rpb; tt -= over ? rpb : 0; off += over ? wrap : 0u;
    }
    template <bool PRED>
    __device__ __forceinline__ void load(int k0, int kend, u32x4 (&L)[8]) {
        const unsigned char* q = p + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PRED) { const u32x4 z = {0u, 0u, 0u, 0u}; L[i] = k0 + kb + i < kend ? *(const u32x4*)(q + i * rs) : z; }
            else L[i] = *(const u32x4*)(q + i * rs);
        }
    }
    // outer rows 2 W, 2 W + 1 of the block: dword W of the 8 frames -> two K-contiguous 16-byte chunks
    template <int W>
    __device__ __forceinline__ void write2(unsigned char* stage, const u32x4 (&L)[8]) {
        u32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o0[j] = perm_lo(L[2 * j + 1][W], L[2 * j][W]); o1[j] = perm_hi(L[2 * j + 1][W], L[2 * j][W]); }
        *(u32x4*)(stage + wr + (2 * W) * KP) = o0;
        *(u32x4*)(stage + wr + (2 * W + 1) * KP) = o1;
    }
};
// wait for this wave's LDS operations only (the global loads of the tile after next stay in flight), then the workgroup barrier
__device__ __forceinline__ void barrier_lds() {
#if defined(SS_EMU)
    __syncthreads();
#else
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0), vmcnt / expcnt untouched -- as a builtin, so that the compiler's own wait insertion knows
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}
}  // namespace g8

// HR = 16-row MFMA tiles per phase (2 or 4).  Waves 2 (M) x 4 (N), 128 x 64 per wave; a K step (64 frames) is 8 / HR phases per K half;
// every phase requests the NEXT phase's fragments from asm, issues its own HR x 4 MFMAs and closes with one lgkmcnt(0).  The tile after
// the current one (in registers since the previous step) is transposed and written in the first four MFMA groups of a step, the loads of
// the tile after that leave right behind the last write; the step's single barrier sits in front of its last phase.
template <int HR>
__global__ __launch_bounds__(512) void gemm8_dwk_kernel(DwJobs jobs, int nitems, int xorder)
{
    using namespace g8;
    constexpr int NPK = 8 / HR, NPH = 2 * NPK;
    SS_DYN_SMEM(lds_raw);
    unsigned char* lds = (unsigned char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
#if defined(SS_EMU)
    const int wave = tid >> 6;
    const unsigned lbase = 0;
#else
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
#endif
    const int wm = wave >> 2, wn = wave & 3, opsel = wave >> 2, G = gridDim.x;
    const unsigned aoff = (unsigned)((wm * 128 + c) * KP + q * 16), boff = (unsigned)((256 + wn * 64 + c) * KP + q * 16);
    StageKT st;
    u32x4 L[8];
    int it = blockIdx.x, ji = 0, m0, n0, k_begin, k_end, nsteps, cur = 0;

#define G8K_SETUP()                                                                                                         \
    do {                                                                                                                     \
        const int ch0_ = it / G * G, pos_ = it - ch0_;                                                                       \
        int R_ = nitems - ch0_; R_ = R_ > G ? G : R_;                                                                        \
        const int xq_ = R_ >> 3, xr_ = R_ & 7, xcd_ = pos_ & 7;                                                             \
        const int lit = xorder ? ch0_ + (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (pos_ >> 3) : it; \
        ji = 0;                                                                                                              \
        while (ji + 1 < jobs.n && lit >= jobs.job[ji].item0 + jobs.job[ji].nitem) ++ji;                                     \
        const DwJob& J_ = jobs.job[ji];                                                                                      \
        const int local = lit - J_.item0, z = local / J_.ntiles, tile = local - z * J_.ntiles;                               \
        const int mt_ = tile / J_.tiles_n, nt_ = tile - mt_ * J_.tiles_n;                                                    \
        m0 = mt_ * 256; n0 = nt_ * 256;                                                                                      \
        k_begin = z * J_.k_chunk; k_end = min(J_.K, k_begin + J_.k_chunk);                                                   \
        nsteps = (k_end - k_begin + BK8 - 1) / BK8;                                                                          \
        st.init(J_, opsel, m0, n0, k_begin, wave, lane);                                                                     \
        if (nsteps == 1) st.template load<true>(k_begin, k_end, L); else if (nsteps > 1) st.template load<false>(k_begin, k_end, L); \
    } while (0)

    G8K_SETUP();
    for (;;) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        bf16x8 fa[2][HR], fb[2][4];
        auto read_a = [&](auto phc, auto xc, unsigned sb) {
            constexpr int ph = phc, x = xc, kk = ph / NPK, i = (ph % NPK) * HR + x;
            lds_read128_async<i * 16 * KP + kk * 64>(fa[ph & 1][x], lds, lbase + sb + aoff);
        };
        auto read_b = [&](auto phc, auto jc, unsigned sb) {
            constexpr int ph = phc, j = jc, kk = ph / NPK;
            lds_read128_async<j * 16 * KP + kk * 64>(fb[kk][j], lds, lbase + sb + boff);
        };
        auto wait_frags = [&](auto phc) {
            constexpr int ph = phc;
            lds_wait_pin(fa[ph & 1]);
            if constexpr (ph % NPK == 0) lds_wait_pin(fb[ph / NPK]);
        };
        auto mfma_row = [&](auto phc, auto xc) {
            constexpr int ph = phc, x = xc, kk = ph / NPK, i = (ph % NPK) * HR + x;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16_16x16x32(fa[ph & 1][x], fb[kk][j], acc[i][j]);
        };
        // MFMA group g = ph * HR + x of step s (16 groups of 4 MFMAs): groups 0..3 write column pair g of tile s + 1 into the other stage, group 4
        // requests tile s + 2.  STEADY (s + 3 < nsteps, compile time): every condition true, tile s + 2 is not the (possibly ragged) last one.
        auto phase = [&](auto phc, auto steady_c, const unsigned S, const unsigned O, const int s) {
            constexpr int ph = phc, nx = ph + 1 < NPH ? ph + 1 : 0;
            constexpr bool last = ph + 1 == NPH, STEADY = steady_c;
            bool rd = true; unsigned rst = S;
            if constexpr (last) {
                if (STEADY || s + 1 < nsteps) { barrier_lds(); rst = O; }       // every wave holds its last fragments of S; tile s + 1 is complete in O
                else rd = false;
            }
            static_for<0, HR>([&](auto xc) {
                constexpr int x = xc, g = ph * HR + x;
                if (rd) {
                    read_a(std::integral_constant<int, nx>{}, xc, rst);
                    if constexpr (nx % NPK == 0) static_for<x * 4 / HR, (x + 1) * 4 / HR>([&](auto j) { read_b(std::integral_constant<int, nx>{}, j, rst); });
                }
                if constexpr (g < 4) { if (STEADY || s + 1 < nsteps) st.template write2<g>(lds + O, L); }
                if constexpr (g == 4) {
                    if (STEADY) { st.advance(); st.template load<false>(0, 0, L); }
                    else if (s + 2 < nsteps) { st.advance(); st.template load<true>(k_begin + (s + 2) * BK8, k_end, L); }
                }
                sched_fence();
                mfma_row(phc, xc);
                sched_fence();
            });
            if (rd) wait_frags(std::integral_constant<int, nx>{});
            sched_fence();
        };
        // ---- tile 0 (in registers since the previous item's epilogue) -> stage cur, tile 1 requested, first fragments
        if (nsteps > 0) { st.template write2<0>(lds + cur * KT_STAGE, L); st.template write2<1>(lds + cur * KT_STAGE, L); st.template write2<2>(lds + cur * KT_STAGE, L); st.template write2<3>(lds + cur * KT_STAGE, L); }
        if (nsteps > 1) { st.advance(); if (nsteps == 2) st.template load<true>(k_begin + BK8, k_end, L); else st.template load<false>(0, 0, L); }
        barrier_lds();
        if (nsteps > 0) {
            static_for<0, HR>([&](auto x) { read_a(std::integral_constant<int, 0>{}, x, (unsigned)(cur * KT_STAGE)); });
            static_for<0, 4>([&](auto j) { read_b(std::integral_constant<int, 0>{}, j, (unsigned)(cur * KT_STAGE)); });
            wait_frags(std::integral_constant<int, 0>{});
        }
        sched_fence();
        {
            int s = 0;
            for (; s + 3 < nsteps; ++s) { const unsigned S = cur * KT_STAGE, O = (cur ^ 1) * KT_STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::true_type{}, S, O, s); }); cur ^= 1; }
            for (; s < nsteps; ++s) { const unsigned S = cur * KT_STAGE, O = (cur ^ 1) * KT_STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::false_type{}, S, O, s); }); cur ^= 1; }
        }
        // `cur` now names the stage the last K tile did NOT use: free since the barrier of the last-but-one step (or never used), so the
        // next item's tile 0 may be written there without another barrier
        const DwJob& J = jobs.job[ji];
        float* Cj = J.C; const long long ldc = J.ldc; const int Mj = J.M, Nj = J.N, atomic = J.atomic, cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) { it += G; G8K_SETUP(); }
        // ---- epilogue: lane holds rows q*4+reg, column c of each 16x16 tile
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cn0 + (wn * 4 + j) * 16 + c, row0 = cm0 + (wm * 8 + i) * 16 + q * 4;
                if (col < Nj) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (row0 + reg < Mj) {
                            float* dst = Cj + (long long)(row0 + reg) * ldc + col;
                            if (atomic) atomicAdd(dst, acc[i][j][reg]); else *dst += acc[i][j][reg];
                        }
                    }
                }
            }
        if (!has_next) break;
    }
#undef G8K_SETUP
}

static thread_local int g_dw_split_override = 0, g_dw_pin = 1, g_dw_xorder = -1, g_dw_kt = -1;        // xorder: -1 = environment SS_GEMM_DW_XCD, default on
extern "C" int ss_gemm_dw_set_option(int what, int value) {
    int old = 0;
    if (what == 0) { old = g_dw_split_override; g_dw_split_override = value; }
    else if (what == 1) { old = g_dw_pin; g_dw_pin = value; }
    else if (what == 2) { old = g_dw_xorder; g_dw_xorder = value; }
    else if (what == 3) { old = g_dw_kt; g_dw_kt = value; }          // 0: transposing-read kernel, 2 / 4: K-contiguous tiles (-1: environment SS_GEMM_DW_KT, default 4)
    return old;
}

extern "C" int ss_gemm_dw_grouped(int n_jobs, const ss_dw_job* jobs, void* stream)
{
    SS_CHECK(jobs && n_jobs >= 1 && n_jobs <= DW_MAX_JOBS, "ss_gemm_dw_grouped: 1..%d jobs per launch", DW_MAX_JOBS);
    DwJobs J; memset(&J, 0, sizeof(J));
    J.n = n_jobs;
    long long tiles_total = 0; int kmax = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const ss_dw_job& s = jobs[i];
        SS_CHECK(s.A && s.B && s.C, "ss_gemm_dw_grouped: null pointer in job %d", i);
        SS_CHECK(s.M > 0 && s.N > 0 && s.K > 0 && s.M % 8 == 0 && s.N % 8 == 0, "ss_gemm_dw_grouped: job %d: M=%d, N=%d must be positive multiples of 8 (16-byte rows), K=%d > 0", i, s.M, s.N, s.K);
        SS_CHECK(((uintptr_t)s.A) % 16 == 0 && ((uintptr_t)s.B) % 16 == 0, "ss_gemm_dw_grouped: job %d: operands must be 16-byte aligned", i);
        const ss_rowmap* maps[2] = {&s.amap, &s.bmap};
        for (int o = 0; o < 2; ++o)
            SS_CHECK(maps[o]->base % 8 == 0 && maps[o]->batch_stride % 8 == 0 && maps[o]->row_stride % 8 == 0, "ss_gemm_dw_grouped: job %d operand %d: strides must be multiples of 8 elements", i, o);
        DwJob& d = J.job[i];
        d.A = (const bf16_t*)s.A; d.B = (const bf16_t*)s.B; d.C = s.C; d.ldc = s.ldc; d.M = s.M; d.N = s.N; d.K = s.K;
        d.amap.base = s.amap.base; d.amap.batch_stride = s.amap.batch_stride; d.amap.row_stride = s.amap.row_stride; d.amap.rows_per_batch = s.amap.rows_per_batch > 0 ? s.amap.rows_per_batch : 0x7fffffff;
        d.bmap.base = s.bmap.base; d.bmap.batch_stride = s.bmap.batch_stride; d.bmap.row_stride = s.bmap.row_stride; d.bmap.rows_per_batch = s.bmap.rows_per_batch > 0 ? s.bmap.rows_per_batch : 0x7fffffff;
        SS_CHECK(d.amap.rows_per_batch == d.bmap.rows_per_batch, "ss_gemm_dw_grouped: job %d: both operands must split their rows into batches of the same length", i);
        d.tiles_n = (s.N + 255) / 256; d.ntiles = ((s.M + 255) / 256) * d.tiles_n;
        tiles_total += d.ntiles; kmax = s.K > kmax ? s.K : kmax;
    }
    // one K split for the whole group: the largest that keeps (tiles x split) inside ONE round of the CUs
    const int cus = g8_cus();
    int split = (int)(cus / (tiles_total > 0 ? tiles_total : 1)); split = split < 1 ? 1 : split;
    if (g_dw_split_override > 0) split = g_dw_split_override;
    int item0 = 0;
    for (int i = 0; i < n_jobs; ++i) {
        DwJob& d = J.job[i];
        const int ksteps = (d.K + 63) / 64;
        int sp = split; if (sp > ksteps / 4) sp = ksteps / 4 > 0 ? ksteps / 4 : 1;          // at least 4 K tiles per item
        const int per = (ksteps + sp - 1) / sp;
        d.k_chunk = per * 64; d.split = (ksteps + per - 1) / per;
        d.atomic = d.split > 1 ? 1 : 0;                     // one item per tile: a plain read-modify-write of C suffices
        d.item0 = item0; d.nitem = d.ntiles * d.split; item0 += d.nitem;
    }
    const int nitems = item0;
    if (g_dw_xorder < 0) { const char* e = getenv("SS_GEMM_DW_XCD"); g_dw_xorder = e ? atoi(e) : 1; }
    dim3 grid(nitems < cus ? nitems : cus), block(512);
    // K-contiguous tiles (the default) need batches of a multiple of 8 frames and of at least 64; option 3 (SS_GEMM_DW_KT): 0 = the transposing-read kernel,
    // 2 / 4 = 16-row MFMA tiles per phase
    if (g_dw_kt < 0) { const char* e = getenv("SS_GEMM_DW_KT"); g_dw_kt = e ? atoi(e) : 4; }
    bool kt_ok = g_dw_kt != 0;
    for (int i = 0; i < n_jobs; ++i) { const int rpb = J.job[i].amap.rows_per_batch; if (rpb != 0x7fffffff && (rpb % 8 != 0 || rpb < 64)) kt_ok = false; }
    if (kt_ok) {
        const size_t smem = 2 * g8::KT_STAGE;
#define G8_DWK(HR_)                                                                                                          \
    do {                                                                                                                      \
        static bool granted = false;                                                                                          \
        if (!granted) { if (g8_grant((const void*)gemm8_dwk_kernel<HR_>, smem)) return 1; granted = true; }                   \
        SS_LAUNCH(SS_KERNEL(gemm8_dwk_kernel<HR_>), grid, block, smem, stream, J, nitems, g_dw_xorder);                       \
    } while (0)
        if (g_dw_kt == 2) G8_DWK(2); else G8_DWK(4);
#undef G8_DWK
        SS_LAUNCH_CHECK("ss_gemm_dw_grouped");
        return 0;
    }
    const size_t smem = 2 * g8::TR_STAGE;
#define G8_DW(PIN_, NW_)                                                                                                     \
    do {                                                                                                                      \
        static bool granted = false;                                                                                          \
        if (!granted) { if (g8_grant((const void*)gemm8_dw_kernel<PIN_, NW_>, smem)) return 1; granted = true; }              \
        SS_LAUNCH(SS_KERNEL(gemm8_dw_kernel<PIN_, NW_>), grid, block, smem, stream, J, nitems, g_dw_xorder);                  \
    } while (0)
    // NW = 4 (128 x 128 per wave, one wave per SIMD, a third less fragment traffic) was measured and lost: 652 vs 816 TFLOP/s on the
    // layer group -- with the compiler's read-all-then-MFMA schedule a lone wave has no partner to cover its LDS waits.
    if (g_dw_pin) G8_DW(1, 8); else G8_DW(0, 8);
#undef G8_DW
    SS_LAUNCH_CHECK("ss_gemm_dw_grouped");
    return 0;
}
