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
constexpr int DW_MAX_JOBS = 24;      // (3.3 KB of kernel arguments; the hi / lo plane form of the parity-grade mode hands in three jobs per weight gradient)
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
    const int wave = wave_uniform(tid >> 6);
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
constexpr int KP = 144;                      // LDS row pitch: 128 B (64 frames of one outer row) + 16 B = 9 sixteen-byte slots
constexpr int KS = 34;                       // outer column m = 8 c + mm (c < 32, mm < 8) lives in LDS row mm * KS + c  (see StageKT)
constexpr int KROWS = 8 * KS;                // 272 row slots per operand tile (rows 32, 33 of every group of 34 are unused)
constexpr int KT_STAGE = 2 * KROWS * KP;     // A tile + B tile

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

// Per thread and K tile: TWO 4 (frames) x 8 (outer columns) blocks of ONE operand, one per 32-frame half h of the tile -- frames
// 32 h + 4 r .. + 3 of outer columns 8 c .. 8 c + 7, with
//     c = (lane & 15) + 16 (wave & 1),   r = (lane >> 4) + 4 ((wave >> 1) & 1);   waves 0..3 copy A, waves 4..7 copy B.
// Each half has its own 16 registers and its own slot in the step (write in MFMA groups 0-1 resp. 8-9, the loads of the tile after next
// right behind), so a load has a full K step (~1 us) to arrive without a second register set, and the requests of a step leave in two
// bursts of 4 instead of one of 8.  Both memory sides want something of the 16 lanes the hardware serves together:
//   * the global load: whole cache lines.  16 lanes = 16 consecutive 16-byte chunks of ONE frame = 2 full lines.  (The first version dealt
//     the lanes of a quarter-wave to 8 frames x 32 bytes: 4 x the tag look-ups, and the kernel was bound by exactly that -- without its
//     global loads the main loop ran at 1600 TFLOP/s, with them at 830; tools/gemm_bench + SS_GEMM_DW_ABL.)
//   * the LDS write of the transposed block (outer column 8 c + mm, 8 bytes = 4 frames at byte 64 h + 8 r of the row): with outer column m
//     in row mm * 34 + c and 9 sixteen-byte slots per row the 16 c of a write sit 9 slots apart (all distinct), r and r + 1 share a slot's halves.
//   * the fragment read of an MFMA operand (16 consecutive outer columns = 8 mm x 2 c, one chunk): slots 2 mm + 9 b (mod 16; 34 * 9 = 306 = 2 mod
//     16): all distinct as well.  (mm * 32 + c would put the 8 mm of a read on the SAME slot: 32 * 9 = 0 mod 16.)
// Requires batches of a multiple of 8 frames (a block never straddles a batch), of at least 64 frames (one wrap per K tile at most),
// and K slices that start at multiples of 64.
struct StageKT {
    const unsigned char* p;                  // operand base + this thread's outer column chunk (bytes)
    unsigned off[2], step, wrap, rs;         // first frame of the block of half h (byte offset) / advance per 64 frames / extra per batch wrap / row stride
    int tt[2], rpb, kb;                      // frame inside its batch (per half), frames per batch, first frame of the h = 0 block inside a K tile
    unsigned wr;                             // LDS byte offset inside a stage: (operand, row c of group mm = 0, byte 8 r)
    __device__ __forceinline__ void init(const DwJob& J, int opsel, int m0, int n0, int k_begin, int wave, int lane) {
        const int c = (lane & 15) + 16 * (wave & 1), r = (lane >> 4) + 4 * ((wave >> 1) & 1);
        const RowMap& map = opsel ? J.bmap : J.amap;
        const int outer = opsel ? n0 : m0, lim = opsel ? J.N : J.M;
        const int col = outer + c * 8 < lim ? outer + c * 8 : 0;         // beyond the matrix: column 0 (products land in C entries that are never stored)
        p = (const unsigned char*)((opsel ? J.B : J.A) + map.base + col);
        rpb = map.rows_per_batch; kb = 4 * r;
        rs = (unsigned)(map.row_stride * 2); step = (unsigned)(map.row_stride * BK8 * 2);
        wrap = (unsigned)((map.batch_stride - (long long)rpb * map.row_stride) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rr = k_begin + 32 * h + kb;
            int b_ = 0, t_ = rr;
            if (rpb != 0x7fffffff) { b_ = rr / rpb; t_ = rr - b_ * rpb; }
            tt[h] = t_;
            off[h] = (unsigned)(((long long)b_ * map.batch_stride + (long long)t_ * map.row_stride) * 2);
        }
        wr = (unsigned)((opsel * KROWS + c) * KP + r * 8);
    }
    template <int H>
    __device__ __forceinline__ void advance() {
        tt[H] += BK8; off[H] += step;                                    // batches hold >= 64 frames (host-side condition): at most one wrap, no loop
        const bool over = tt[H] >= rpb; tt[H] -= over ? rpb : 0; off[H] += over ? wrap : 0u;
    }
    // The loads of the steady state are issued from asm and counted by hand (wait_vm): left to the compiler, its wait insertion put an
    // s_waitcnt vmcnt(0) at the loop header (the loop-carried state of two register sets in flight merges to "everything pending"), which
    // also drained the OTHER half's loads a few hundred cycles after their issue.  A compiler-visible load into the same registers anywhere
    // would bring that back (it would have to guard every later use), so the predicated form (ragged last tile; prologue of short K
    // slices) loads into temporaries and copies: the copy is where the compiler waits, L itself never carries a pending load it knows of.
    template <int H, bool PRED>
    __device__ __forceinline__ void load(int k0, int kend, u32x4 (&L)[4]) {
        const unsigned char* q = p + off[H];
        if constexpr (PRED) {
            u32x4 T[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const u32x4 z = {0u, 0u, 0u, 0u}; T[i] = k0 + 32 * H + kb + i < kend ? *(const u32x4*)(q + i * rs) : z; }
#if !defined(SS_EMU)
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) as a builtin: the copies below coalesce away, and without the explicit wait the
#endif                                                   // compiler would carry "L pending" into the steady loop and wait vmcnt(0) at its header
#pragma unroll
            for (int i = 0; i < 4; ++i) L[i] = T[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#if defined(SS_EMU)
                L[i] = *(const u32x4*)(q + i * rs);
#else
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(L[i]) : "v"(q + i * rs) : "memory");
#endif
            }
        }
    }
    // all but the N most recent asm loads of this wave have arrived (N = 4: the other half's loads stay in flight); ties L to this point
    template <int N>
    __device__ __forceinline__ static void wait_vm(u32x4 (&L)[4]) {
#if !defined(SS_EMU)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(L[0]), "+v"(L[1]), "+v"(L[2]), "+v"(L[3]) : "n"(N));
#endif
    }
    // outer rows 2 W, 2 W + 1 of the block of half H: dword W of its 4 frames -> two K-contiguous 8-byte pieces
    template <int H, int W>
    __device__ __forceinline__ void write2(unsigned char* stage, const u32x4 (&L)[4]) {
        u32x2 o0, o1;
        o0[0] = perm_lo(L[1][W], L[0][W]); o0[1] = perm_lo(L[3][W], L[2][W]);
        o1[0] = perm_hi(L[1][W], L[0][W]); o1[1] = perm_hi(L[3][W], L[2][W]);
        *(u32x2*)(stage + wr + (2 * W) * KS * KP + 64 * H) = o0;
        *(u32x2*)(stage + wr + (2 * W + 1) * KS * KP + 64 * H) = o1;
    }
    template <int H>
    __device__ __forceinline__ void write_all(unsigned char* stage, const u32x4 (&L)[4]) { write2<H, 0>(stage, L); write2<H, 1>(stage, L); write2<H, 2>(stage, L); write2<H, 3>(stage, L); }
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
// the current one (in registers since the previous step) is transposed and written in two halves (MFMA groups 0-1 and 8-9 of 16), the loads
// of the tile after that leave right behind each half; the step's single barrier sits in front of its last phase.
// ABL: compile-time ablation mask for tuning (results wrong): 1 no global loads, 2 no transposes / LDS writes, 4 no MFMAs in the steady steps, 8 no C update
template <int HR, int ABL, bool SHIFT = false>
__global__ __launch_bounds__(512) void gemm8_dwk_kernel(DwJobs jobs, int nitems, int xorder)
{
    using namespace g8;
    constexpr int NPK = 8 / HR, NPH = 2 * NPK;
    SS_DYN_SMEM(lds_raw);
    unsigned char* lds = (unsigned char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int wave = wave_uniform(tid >> 6);
    const unsigned lbase = lds_byte_address(lds);
    const int wm = wave >> 2, wn = wave & 3, opsel = wave >> 2, G = gridDim.x;
    // fragment of MFMA tile i: outer columns 16 i + c (c = lane & 15) = group mm = c & 7, row 2 i + (c >> 3) (+ 16 wm | 8 wn): tile i adds 2 rows
    const unsigned aoff = (unsigned)(((c & 7) * KS + (c >> 3) + 16 * wm) * KP + q * 16), boff = (unsigned)((KROWS + (c & 7) * KS + (c >> 3) + 8 * wn) * KP + q * 16);
    StageKT st;
    u32x4 L0[4], L1[4];                      // the two half-tile blocks in flight
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
        if (nsteps == 1) { st.template load<0, true>(k_begin, k_end, L0); st.template load<1, true>(k_begin, k_end, L1); }   \
        else if (nsteps > 1) { st.template load<0, false>(0, 0, L0); st.template load<1, false>(0, 0, L1); }                 \
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
            lds_read128_async<i * 2 * KP + kk * 64>(fa[ph & 1][x], lds, lbase + sb + aoff);
        };
        auto read_b = [&](auto phc, auto jc, unsigned sb) {
            constexpr int ph = phc, j = jc, kk = ph / NPK;
            lds_read128_async<j * 2 * KP + kk * 64>(fb[kk][j], lds, lbase + sb + boff);
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
        // MFMA group g = ph * HR + x of step s (16 groups of 4 MFMAs): groups 0-1 / 8-9 transpose and write half 0 / 1 of tile s + 1 into the other
        // stage, groups 2 / 10 request that half of tile s + 2.  STEADY (s + 3 < nsteps, compile time): every condition true, tile s + 2 is not the
        // (possibly ragged) last one.
        // SH (experiment, off: SHIFT = false): the copy slots of the waves that stage B (4..7) two groups behind those of the waves that stage A.
        // Wave w and wave w + 4 share a SIMD and leave every barrier together, so both copy and both compute at the same moments; shifting
        // one of them needs the whole step loop twice (a wave-uniform branch around it), and with two copies of the loop the allocator
        // spilled 321 registers -- not measurable as a schedule.
        auto phase = [&](auto phc, auto steady_c, auto shc, const unsigned S, const unsigned O, const int s) {
            constexpr int ph = phc, nx = ph + 1 < NPH ? ph + 1 : 0, SH = shc;
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
                const bool wr_ok = (STEADY && !(ABL & 2)) || (!STEADY && s + 1 < nsteps);
                if constexpr (g == 0 + SH) { if (wr_ok) { if (STEADY) StageKT::wait_vm<4>(L0); else StageKT::wait_vm<0>(L0); st.template write2<0, 0>(lds + O, L0); st.template write2<0, 1>(lds + O, L0); } }
                if constexpr (g == 1 + SH) { if (wr_ok) { st.template write2<0, 2>(lds + O, L0); st.template write2<0, 3>(lds + O, L0); } }
                if constexpr (g == 8 + SH) { if (wr_ok) { if (STEADY) StageKT::wait_vm<4>(L1); else StageKT::wait_vm<0>(L1); st.template write2<1, 0>(lds + O, L1); st.template write2<1, 1>(lds + O, L1); } }
                if constexpr (g == 9 + SH) { if (wr_ok) { st.template write2<1, 2>(lds + O, L1); st.template write2<1, 3>(lds + O, L1); } }
                if constexpr (g == 2 + SH) {
                    if (STEADY) { st.template advance<0>(); if (!(ABL & 1)) st.template load<0, false>(0, 0, L0); }
                    else if (s + 2 < nsteps) { st.template advance<0>(); st.template load<0, true>(k_begin + (s + 2) * BK8, k_end, L0); }
                }
                if constexpr (g == 10 + SH) {
                    if (STEADY) { st.template advance<1>(); if (!(ABL & 1)) st.template load<1, false>(0, 0, L1); }
                    else if (s + 2 < nsteps) { st.template advance<1>(); st.template load<1, true>(k_begin + (s + 2) * BK8, k_end, L1); }
                }
                sched_fence();
                if (!(STEADY && (ABL & 4))) mfma_row(phc, xc);
                sched_fence();
            });
            if (rd) wait_frags(std::integral_constant<int, nx>{});
            sched_fence();
        };
        // ---- tile 0 (in registers since the previous item's epilogue) -> stage cur, tile 1 requested, first fragments
        if (nsteps > 0) { StageKT::wait_vm<0>(L0); StageKT::wait_vm<0>(L1); st.template write_all<0>(lds + cur * KT_STAGE, L0); st.template write_all<1>(lds + cur * KT_STAGE, L1); }
        if (nsteps > 1) {
            st.template advance<0>(); st.template advance<1>();
            if (nsteps == 2) { st.template load<0, true>(k_begin + BK8, k_end, L0); st.template load<1, true>(k_begin + BK8, k_end, L1); }
            else { st.template load<0, false>(0, 0, L0); st.template load<1, false>(0, 0, L1); }
        }
        barrier_lds();
        if (nsteps > 0) {
            static_for<0, HR>([&](auto x) { read_a(std::integral_constant<int, 0>{}, x, (unsigned)(cur * KT_STAGE)); });
            static_for<0, 4>([&](auto j) { read_b(std::integral_constant<int, 0>{}, j, (unsigned)(cur * KT_STAGE)); });
            wait_frags(std::integral_constant<int, 0>{});
        }
        sched_fence();
        {
            auto run = [&](auto shc) {
                int s = 0;
                for (; s + 3 < nsteps; ++s) { const unsigned S = cur * KT_STAGE, O = (cur ^ 1) * KT_STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::true_type{}, shc, S, O, s); }); cur ^= 1; }
                for (; s < nsteps; ++s) { const unsigned S = cur * KT_STAGE, O = (cur ^ 1) * KT_STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::false_type{}, shc, S, O, s); }); cur ^= 1; }
            };
            if (SHIFT && opsel) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 0>{});
        }
        // `cur` now names the stage the last K tile did NOT use: free since the barrier of the last-but-one step (or never used), so the
        // next item's tile 0 may be written there without another barrier
        const DwJob& J = jobs.job[ji];
        float* Cj = J.C; const long long ldc = J.ldc; const int Mj = J.M, Nj = J.N, atomic = J.atomic, cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) { it += G; G8K_SETUP(); }
        // ---- epilogue: lane holds rows q*4+reg, column c of each 16x16 tile (an update instruction touches 4 rows x 64 bytes).  ~22 of the ~315 us of a
        // layer group are these f32 atomics (SS_GEMM_DW_ABL=8).  Measured and dropped: exchanging the registers of neighbouring tiles half-wave-wise
        // (v_permlane32_swap) so that an instruction covers 2 rows x 128 contiguous bytes -- same time: the L2 atomic units are bound per float, not per line.
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = cn0 + (wn * 4 + j) * 16 + c, row0 = cm0 + (wm * 8 + i) * 16 + q * 4;
                if (col < Nj && !((ABL & 8) && acc[i][j][0] != 12345.f)) {
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
    int s_flags[DW_MAX_JOBS];
    for (int i = 0; i < n_jobs; ++i) {
        const ss_dw_job& s = jobs[i];
        s_flags[i] = s.flags;
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
    if (tiles_total > cus) {
        // more tiles than CUs (the plane form: three jobs per gradient): the split that minimises rounds x (K tiles per item + ~12 K tiles' worth of
        // atomic epilogue) -- 324 tiles of 344 K tiles: split 1 = 2 rounds x 356, split 3 = 4 rounds x 127
        const int ks = (kmax + 63) / 64;
        double best = 0;
        for (int cand = 1; cand <= 8; ++cand) {
            const long long items = tiles_total * cand, rounds = (items + cus - 1) / cus;
            const double cost = (double)rounds * ((ks + cand - 1) / cand + 12);
            if (cand == 1 || cost < best) { best = cost; split = cand; }
        }
    }
    if (g_dw_split_override > 0) split = g_dw_split_override;
    int item0 = 0;
    for (int i = 0; i < n_jobs; ++i) {
        DwJob& d = J.job[i];
        const int ksteps = (d.K + 63) / 64;
        int sp = split; if (sp > ksteps / 4) sp = ksteps / 4 > 0 ? ksteps / 4 : 1;          // at least 4 K tiles per item
        const int per = (ksteps + sp - 1) / sp;
        d.k_chunk = per * 64; d.split = (ksteps + per - 1) / per;
        d.atomic = (d.split > 1 || (s_flags[i] & 1)) ? 1 : 0;   // one item per tile of a C nobody else updates: a plain read-modify-write suffices
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
        if (!granted) { if (g8_grant((const void*)gemm8_dwk_kernel<HR_, ABL_>, smem)) return 1; granted = true; }             \
        SS_LAUNCH(SS_KERNEL(gemm8_dwk_kernel<HR_, ABL_>), grid, block, smem, stream, J, nitems, g_dw_xorder);                 \
    } while (0)
#define ABL_ 0
        static const int abl = getenv("SS_GEMM_DW_ABL") ? atoi(getenv("SS_GEMM_DW_ABL")) : 0;       // tuning only (tools/gemm_bench): results are wrong when set
        if (abl == 0) { if (g_dw_kt == 2) G8_DWK(2); else G8_DWK(4); }
#undef ABL_
#if defined(G8_DW_ABLATION)
#define ABL_CASE(A_) else if (abl == A_) { constexpr int ABL_ = A_; G8_DWK(4); }
        ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(8) ABL_CASE(11) ABL_CASE(7)
#undef ABL_CASE
#endif
        else { ss_set_error("ss_gemm_dw_grouped: SS_GEMM_DW_ABL needs a -DG8_DW_ABLATION build"); return 1; }
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
