// gemm.hip -- the MFMA contraction core of the transduction hot path (gfx950).
//
//   C[m][n] (+)= alpha * sum_k A(m,k) * B(n,k)   -> epilogue (bias, ReLU, dropout, gate, scatter)
//
// Every dense contraction of the reference's training step lands here:
//   * nn.Conv1d k=3 / k=1, stride 1|2   (architecture.py:18,20,24)  -- implicit GEMM: the three taps of
//     a (B, T+2, C) zero-padded activation buffer are one contiguous 3C-wide row, so im2col is just a
//     RowMap with row_stride = stride*C (overlapping rows, nothing is materialised);
//   * nn.Linear                         (architecture.py:51,55,59; transformer.py:32,34)
//   * the per-head einsum projections   (transformer.py:96-98,111)  -- fused QKV / W_o GEMMs;
//   * and their autograd transposes: dX = dY.W ("B outer-contiguous") and dW = dY^T.X (both operands
//     outer-contiguous, reduction over the B*T rows, split-K with f32 atomics).
//
// Operand modes: KC = reduction index contiguous (elem(o,r) = p[rowmap(o) + r]);
//                OC = outer index contiguous     (elem(o,r) = p[rowmap(r) + o])  -> transposed while
//                staging (register 4x8 / 4x4 transposes), so the LDS image and the MFMA loop are identical.
// Tile 128x128x(128 bytes of K), 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16 tiles,
// double-buffered LDS (64 KiB), XOR-swizzled 16-B chunks (conflict-free ds_read_b128), global->register->
// LDS staging with the next tile's loads issued before the current tile's MFMAs.
//   bf16: v_mfma_f32_16x16x32_bf16 (f32 accumulate);  f32: v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
#include "common.h"
#include "silent_speech_hip.h"
#include <stdlib.h>
#include <math.h>

#include "gemm_common.h"

// ---------------------------------------------------------------- staging: KC (copy) mode
template <class T>
struct StageKC {
    typedef u32x4 Regs[4];
    long long off[4];
    const T* p;
    int c, r0;
    __device__ __forceinline__ void init(const T* p_, const RowMap& map, int outer0, int outer_size, int tid) {
        p = p_; c = tid & 7; r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int o = outer0 + r0 + 32 * i;
            off[i] = o < outer_size ? rowmap_off(map, o) : -1;
        }
    }
    __device__ __forceinline__ void seek(int) {}
    __device__ __forceinline__ void load(int k0, int kend, Regs& reg) {
        int kk = k0 + c * Elem<T>::EPC;
        bool kv = kk < kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            reg[i] = (kv && off[i] >= 0) ? *(const u32x4*)(p + off[i] + kk) : z;
        }
    }
    __device__ __forceinline__ void store(unsigned char* tile, const Regs& reg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { int r = r0 + 32 * i; *(u32x4*)(tile + swz(r, c)) = reg[i]; }
    }
};

// ---------------------------------------------------------------- staging: OC (transposing) mode
template <class T> struct StageOC;
template <>
struct StageOC<bf16_t> {   // thread: 8 outer x 4 reduction
    typedef u32x4 Regs[4];
    const bf16_t* p; RowMap map; int ob, rb, bb, tt; bool ov;
    __device__ __forceinline__ void init(const bf16_t* p_, const RowMap& map_, int outer0, int outer_size, int tid) {
        map = map_; ob = (tid & 15) * 8; rb = (tid >> 4) * 4;
        ov = outer0 + ob < outer_size;     // outer_size % 8 == 0 (checked on the host)
        p = p_ + outer0 + ob;
    }
    __device__ __forceinline__ void seek(int k_begin) { int r = k_begin + rb; bb = r / map.rows_per_batch; tt = r - bb * map.rows_per_batch; }
    __device__ __forceinline__ void load(int k0, int kend, Regs& in) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int t = tt + j, b = bb;
            while (t >= map.rows_per_batch) { t -= map.rows_per_batch; ++b; }
            u32x4 z = {0u, 0u, 0u, 0u};
            bool v = ov && (k0 + rb + j < kend);
            in[j] = v ? *(const u32x4*)(p + map.base + (long long)b * map.batch_stride + (long long)t * map.row_stride) : z;
        }
        tt += Elem<bf16_t>::BK;
        while (tt >= map.rows_per_batch) { tt -= map.rows_per_batch; ++bb; }
    }
    __device__ __forceinline__ void store(unsigned char* tile, const Regs& in) {
        int q = rb >> 2;                    // 8-byte slot index within the row (0..15)
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            unsigned x0 = in[0][o >> 1], x1 = in[1][o >> 1], x2 = in[2][o >> 1], x3 = in[3][o >> 1];
            u32x2 w;
            if (o & 1) { w[0] = (x0 >> 16) | (x1 & 0xffff0000u); w[1] = (x2 >> 16) | (x3 & 0xffff0000u); }
            else       { w[0] = (x0 & 0xffffu) | (x1 << 16);     w[1] = (x2 & 0xffffu) | (x3 << 16); }
            int r = ob + o;
            *(u32x2*)(tile + swz(r, q >> 1) + ((q & 1) << 3)) = w;
        }
    }
};
template <>
struct StageOC<float> {    // thread: 4 outer x 4 reduction
    typedef f32x4 Regs[4];
    const float* p; RowMap map; int ob, rb, bb, tt; bool ov;
    __device__ __forceinline__ void init(const float* p_, const RowMap& map_, int outer0, int outer_size, int tid) {
        map = map_; ob = (tid & 31) * 4; rb = (tid >> 5) * 4;
        ov = outer0 + ob < outer_size;     // outer_size % 4 == 0
        p = p_ + outer0 + ob;
    }
    __device__ __forceinline__ void seek(int k_begin) { int r = k_begin + rb; bb = r / map.rows_per_batch; tt = r - bb * map.rows_per_batch; }
    __device__ __forceinline__ void load(int k0, int kend, Regs& in) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int t = tt + j, b = bb;
            while (t >= map.rows_per_batch) { t -= map.rows_per_batch; ++b; }
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bool v = ov && (k0 + rb + j < kend);
            in[j] = v ? *(const f32x4*)(p + map.base + (long long)b * map.batch_stride + (long long)t * map.row_stride) : z;
        }
        tt += Elem<float>::BK;
        while (tt >= map.rows_per_batch) { tt -= map.rows_per_batch; ++bb; }
    }
    __device__ __forceinline__ void store(unsigned char* tile, const Regs& in) {
        int chunk = rb >> 2;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            f32x4 w = {in[0][o], in[1][o], in[2][o], in[3][o]};
            *(f32x4*)(tile + swz(ob + o, chunk)) = w;
        }
    }
};

// ---------------------------------------------------------------- staging: TR mode (bf16, BOTH operands outer-contiguous: dW = dY^T X)
// The tile is copied UNtransposed ([reduction k][outer m], 16-byte chunks, like KC) and the transposition happens in the
// LDS read: ds_read_b64_tr_b16 hands lane (c, q) the 4 k-values of outer row c (tools/emu/hipemu.h has the lane map).
// Row pitch 288 B (= 256 B + 32 B pad): the 16 lanes of a group read 4 k-rows x 32 B on disjoint banks, and the two groups
// served together (q, q+1) read adjacent k-quads.  k inside a 32-deep MFMA step is enumerated as h*16 + q*4 + j for
// BOTH operands (any common permutation of k leaves the contraction unchanged).
constexpr int TR_ROWB = 288;
constexpr int TR_TILE = 64 * TR_ROWB;
struct StageTR {
    typedef u32x4 Regs[4];
    const bf16_t* p; RowMap map; int c, r0; bool ov; int bb[4], tt[4];
    __device__ __forceinline__ void init(const bf16_t* p_, const RowMap& map_, int outer0, int outer_size, int tid) {
        map = map_; c = tid & 15; r0 = tid >> 4;
        ov = outer0 + c * 8 < outer_size;          // outer_size % 8 == 0 (checked on the host)
        p = p_ + outer0 + c * 8;
    }
    __device__ __forceinline__ void seek(int k_begin) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = k_begin + r0 + 16 * i;
            if (map.rows_per_batch == 0x7fffffff) { bb[i] = 0; tt[i] = r; } else { bb[i] = r / map.rows_per_batch; tt[i] = r - bb[i] * map.rows_per_batch; }
        }
    }
    __device__ __forceinline__ void load(int k0, int kend, Regs& reg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x4 z = {0u, 0u, 0u, 0u};
            const bool v = ov && (k0 + r0 + 16 * i < kend);
            reg[i] = v ? *(const u32x4*)(p + map.base + (long long)bb[i] * map.batch_stride + (long long)tt[i] * map.row_stride) : z;
            tt[i] += 64;
            while (tt[i] >= map.rows_per_batch) { tt[i] -= map.rows_per_batch; ++bb[i]; }
        }
    }
    __device__ __forceinline__ void store(unsigned char* tile, const Regs& reg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(tile + (r0 + 16 * i) * TR_ROWB + c * 16) = reg[i];
    }
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int sub, int kk, int c, int q) {
    const unsigned char* a0 = tile + (kk * 32 + q * 4 + (c >> 2)) * TR_ROWB + sub * 32 + (c & 3) * 8;
    const s16x4 lo = lds_read_tr16(a0), hi = lds_read_tr16(a0 + 16 * TR_ROWB);
    bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
}
struct TileMmaTR {
    static __device__ __forceinline__ void run(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
        const int c = lane & 15, q = lane >> 4;
        bf16x8 a[2][4], b[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = tr_frag(As, wm * 4 + i, kk, c, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[kk][j] = tr_frag(Bs, wn * 4 + j, kk, c, q);
        }
        sched_fence();                                     // the 32 transposing reads go out together, ahead of the MFMAs
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16_16x16x32(a[kk][i], b[kk][j], acc[i][j]);
    }
};

template <class T, int MODE> struct StagerSel { typedef StageKC<T> type; };
template <class T> struct StagerSel<T, OP_OC> { typedef StageOC<T> type; };

// ---------------------------------------------------------------- MFMA over one staged K tile
template <class T> struct TileMma;
template <>
struct TileMma<bf16_t> {
    static __device__ __forceinline__ void run(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
        int r = lane & 15, q = lane >> 4;
        // both 32-deep halves of the K tile are fetched up front (16 x ds_read_b128): the second half's LDS latency hides under
        // the first half's 16 MFMAs instead of being re-exposed in front of every group of four
        bf16x8 a[2][4], b[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = *(const bf16x8*)(As + swz(wm * 64 + i * 16 + r, kk * 4 + q));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[kk][j] = *(const bf16x8*)(Bs + swz(wn * 64 + j * 16 + r, kk * 4 + q));
        }
#if !defined(SS_EMU)
        __builtin_amdgcn_sched_barrier(0);                 // keep the 16 reads ahead of the MFMAs (the scheduler would sink them again)
        if (SS_GEMM_PRIO) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16_16x16x32(a[kk][i], b[kk][j], acc[i][j]);
#if !defined(SS_EMU)
        if (SS_GEMM_PRIO) __builtin_amdgcn_s_setprio(0);
#endif
    }
};
template <>
struct TileMma<float> {
    // lane (r,q) reads the 8 consecutive k = q*8 .. q*8+7 of its row; MFMA e pairs element e of every
    // quarter: a permutation of k shared by A and B, so the sum over the 32-deep tile is unchanged.
    static __device__ __forceinline__ void run(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
        int r = lane & 15, q = lane >> 4;
        f32x4 a[4][2], b[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i][0] = *(const f32x4*)(As + swz(wm * 64 + i * 16 + r, q * 2)); a[i][1] = *(const f32x4*)(As + swz(wm * 64 + i * 16 + r, q * 2 + 1)); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { b[j][0] = *(const f32x4*)(Bs + swz(wn * 64 + j * 16 + r, q * 2)); b[j][1] = *(const f32x4*)(Bs + swz(wn * 64 + j * 16 + r, q * 2 + 1)); }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f32_16x16x4(a[i][e >> 2][e & 3], b[j][e >> 2][e & 3], acc[i][j]);
    }
};

// f32 operands, bf16 x 3 arithmetic (dtype_in = SS_F32X3): x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and
//   a.b ~ a_lo.b_hi + a_hi.b_lo + a_hi.b_hi          (the a_lo.b_lo term, 2^-18 of the product, is dropped)
// on three bf16 MFMAs with f32 accumulation: every product is exact in f32, the operands carry 16-17 significant bits instead of 24.
// The LDS image stays the f32 one (all four staging modes and the global->LDS copy are unchanged): lane (r, q) reads the same 8
// consecutive k of its row as the exact kernel, which is precisely the A / B fragment of one 16x16x32 bf16 MFMA; the split is done on
// the fragments (3 VALU per element: one packed conversion per pair and operand half, the widening and the subtraction).
__device__ __forceinline__ void split_bf16x3(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned hp = pack_bf16(v[2 * p], v[2 * p + 1]);
        h[p] = hp;
        l[p] = pack_bf16(v[2 * p] - __uint_as_float(hp << 16), v[2 * p + 1] - __uint_as_float(hp & 0xffff0000u));
    }
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
struct TileMmaX3 {
    struct Frags { bf16x8 ah[4], al[4], bh[4], bl[4]; };
    // every byte this wave needs from the staged K tile, split: after load() the LDS stage is not touched again by this wave
    static __device__ __forceinline__ void load(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane, Frags& f) {
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            split_bf16x3(*(const f32x4*)(As + swz(wm * 64 + i * 16 + r, q * 2)), *(const f32x4*)(As + swz(wm * 64 + i * 16 + r, q * 2 + 1)), f.ah[i], f.al[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            split_bf16x3(*(const f32x4*)(Bs + swz(wn * 64 + j * 16 + r, q * 2)), *(const f32x4*)(Bs + swz(wn * 64 + j * 16 + r, q * 2 + 1)), f.bh[j], f.bl[j]);
    }
    static __device__ __forceinline__ void mma(const Frags& f, f32x4 (&acc)[4][4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma_bf16_16x16x32(f.al[i], f.bh[j], acc[i][j]);        // small terms first
                acc[i][j] = mfma_bf16_16x16x32(f.ah[i], f.bl[j], acc[i][j]);
                acc[i][j] = mfma_bf16_16x16x32(f.ah[i], f.bh[j], acc[i][j]);
            }
    }
    static __device__ __forceinline__ void run(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
        Frags f;
        load(As, Bs, wm, wn, lane, f);
        mma(f, acc);
    }
};


template <bool TR, class S> struct TrSel { typedef S type; };
template <class S> struct TrSel<true, S> { typedef StageTR type; };
template <bool TR, class T, int X3> struct MmaSel { typedef TileMma<T> type; };
template <class T> struct MmaSel<true, T, 0> { typedef TileMmaTR type; };
template <> struct MmaSel<false, float, 1> { typedef TileMmaX3 type; };

// Work item `it` (tile x K-slice) of a persistent block.  Items are taken G at a time; inside each batch of G the
// XCD-aware bijective remap keeps consecutive tile ids (which share an A row panel) on one XCD's L2.
__device__ __forceinline__ void item_coord(int it, int G, int nitems, int ntiles, int tiles_n, int k_chunk, int K, int& m0, int& n0, int& k_begin, int& k_end)
{
    const int chunk0 = it / G * G, pos = it - chunk0;
    int R = nitems - chunk0; R = R > G ? G : R;
    const int xcd = pos & 7, q = R >> 3, r = R & 7;
    const int idx = chunk0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pos >> 3);
    const int z = idx / ntiles, tile = idx - z * ntiles;
    const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
    m0 = mt * BM; n0 = nt * BN;
    k_begin = z * k_chunk; k_end = min(K, k_begin + k_chunk);
}

// PERSISTENT kernel: gridDim.x = min(#items, 2 blocks x #CUs); each block walks items it, it+G, ...  The global loads of
// the NEXT item's first K-tiles are issued before the current item's epilogue, so the C-tile store burst, the next
// tile's cold loads and the launch ramp overlap instead of serialising once per "round" of tiles.
template <class T, class TO, int AMODE, int BMODE, int X3 = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T* __restrict__ A, const T* __restrict__ B, TO* __restrict__ C,
                                                      int M, int N, int K, RowMap amap, RowMap bmap, GemmEpi epi,
                                                      int k_chunk, int tiles_m, int tiles_n, int nitems)
{
    constexpr bool TR = sizeof(T) == 2 && AMODE == OP_OC && BMODE == OP_OC;     // LDS-transpose-read variant (dW GEMMs)
    constexpr int STAGE = TR ? TR_TILE : BM * ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][STAGE];
    constexpr int BK = Elem<T>::BK;
    constexpr bool D2 = false;               // second register set (prefetch 2 K-tiles ahead): measured no gain, and with the
                                             // cross-item prefetch live during the epilogue it spills -> one set
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x, ntiles = tiles_m * tiles_n;
    typedef typename TrSel<TR, typename StagerSel<T, AMODE>::type>::type SA;
    typedef typename TrSel<TR, typename StagerSel<T, BMODE>::type>::type SB;
    typedef typename MmaSel<TR, T, X3>::type MMA;
    SA sa; SB sb;
    typename SA::Regs ra0, ra1;
    typename SB::Regs rb0, rb1;
    int it = blockIdx.x, m0, n0, k_begin, k_end, nsteps;

#define SS_SETUP()                                                                                             \
    do {                                                                                                        \
        item_coord(it, G, nitems, ntiles, tiles_n, k_chunk, K, m0, n0, k_begin, k_end);                          \
        sa.init(A, amap, m0, M, tid); sb.init(B, bmap, n0, N, tid);                                              \
        sa.seek(k_begin); sb.seek(k_begin);                                                                      \
        nsteps = (k_end - k_begin + BK - 1) / BK;                                                                \
        if (nsteps > 0) {                                                                                        \
            sa.load(k_begin, k_end, ra0); sb.load(k_begin, k_end, rb0);                                          \
            if (D2 && nsteps > 1) { sa.load(k_begin + BK, k_end, ra1); sb.load(k_begin + BK, k_end, rb1); }      \
        }                                                                                                        \
    } while (0)

    SS_SETUP();
    for (;;) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        if (nsteps > 0) { sa.store(lds[0][0], ra0); sb.store(lds[0][1], rb0); }
        __syncthreads();
        // Software pipeline: at step s the MFMAs read LDS[s&1]; tile s+1 (already in registers) is written to
        // LDS[(s+1)&1] afterwards; the loads of tile s+2 (bf16) are in flight the whole time; one barrier per K-tile.
        if (D2) {
            for (int s = 0; s < nsteps; s += 2) {
                if (s + 2 < nsteps) { const int k0 = k_begin + (s + 2) * BK; sa.load(k0, k_end, ra0); sb.load(k0, k_end, rb0); }
                MMA::run(lds[0][0], lds[0][1], wm, wn, lane, acc);
                if (s + 1 < nsteps) { sa.store(lds[1][0], ra1); sb.store(lds[1][1], rb1); }
                __syncthreads();
                if (s + 1 >= nsteps) break;
                if (s + 3 < nsteps) { const int k0 = k_begin + (s + 3) * BK; sa.load(k0, k_end, ra1); sb.load(k0, k_end, rb1); }
                MMA::run(lds[1][0], lds[1][1], wm, wn, lane, acc);
                if (s + 2 < nsteps) { sa.store(lds[0][0], ra0); sb.store(lds[0][1], rb0); }
                __syncthreads();
            }
        } else {
            for (int s = 0; s < nsteps; ++s) {
                const int cur = s & 1;
                const bool more = s + 1 < nsteps;
                if (more) { const int k0 = k_begin + (s + 1) * BK; sa.load(k0, k_end, ra0); sb.load(k0, k_end, rb0); }
                if (!(epi.debug & 4)) MMA::run(lds[cur][0], lds[cur][1], wm, wn, lane, acc);
                if (more) { sa.store(lds[cur ^ 1][0], ra0); sb.store(lds[cur ^ 1][1], rb0); }
                __syncthreads();
            }
        }
        // ---- next item: its first global loads fly during this item's epilogue
        const int cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) { it += G; SS_SETUP(); }

        // ---- epilogue: lane holds rows (lane>>4)*4+reg, column lane&15 of each 16x16 tile
        const int cq = lane >> 4, cr = lane & 15;
        if (epi.fast) {
            // C tile through LDS (the staging buffers are free after the loop's final barrier) -> 16-byte coalesced stores
            TO* ct = (TO*)&lds[0][0][0];
            constexpr int LDC = BN + 16 / (int)sizeof(TO);
            constexpr int NPASS = sizeof(TO) == 4 ? 2 : 1;   // an f32 128x128 tile does not fit 64 KiB: two 64-row passes
#define SS_EPS(I, J) epilogue_stage<TO, GEN>(acc[I][J], ct, LDC, lr + I * 16 + cq * 4, wn * 64 + J * 16 + cr, epi, cm0 + wm * 64 + I * 16 + cq * 4, cn0 + wn * 64 + J * 16 + cr, M, N)
#define SS_EPS_ROW(I) SS_EPS(I, 0); SS_EPS(I, 1); SS_EPS(I, 2); SS_EPS(I, 3)
#define SS_EPS_ALL SS_EPS_ROW(0); SS_EPS_ROW(1); SS_EPS_ROW(2); SS_EPS_ROW(3)
            if (NPASS == 1) {
                const int lr = wm * 64;
                if (!(epi.debug & 2)) {
                    if (epi.general == 1) { constexpr int GEN = 1; SS_EPS_ALL; } else if (epi.general == 2) { constexpr int GEN = 2; SS_EPS_ALL; } else { constexpr int GEN = 0; SS_EPS_ALL; }
                }
                __syncthreads();
                if (!(epi.debug & 1)) epilogue_flush<TO>(ct, LDC, C, epi, cm0, BM, cn0, M, N, tid);
            } else {
                const int lr = 0;
                if (wm == 0) { if (epi.general == 1) { constexpr int GEN = 1; SS_EPS_ALL; } else if (epi.general == 2) { constexpr int GEN = 2; SS_EPS_ALL; } else { constexpr int GEN = 0; SS_EPS_ALL; } }
                __syncthreads();
                epilogue_flush<TO>(ct, LDC, C, epi, cm0, BM / 2, cn0, M, N, tid);
                __syncthreads();
                if (wm == 1) { if (epi.general == 1) { constexpr int GEN = 1; SS_EPS_ALL; } else if (epi.general == 2) { constexpr int GEN = 2; SS_EPS_ALL; } else { constexpr int GEN = 0; SS_EPS_ALL; } }
                __syncthreads();
                epilogue_flush<TO>(ct, LDC, C, epi, cm0 + BM / 2, BM / 2, cn0, M, N, tid);
            }
#undef SS_EPS_ALL
#undef SS_EPS_ROW
#undef SS_EPS
        } else {
#define SS_EPI(I, J) epilogue_tile<TO>(acc[I][J], C, epi, cm0 + wm * 64 + I * 16 + cq * 4, cn0 + wn * 64 + J * 16 + cr, M, N)
#define SS_EPI_ROW(I) SS_EPI(I, 0); SS_EPI(I, 1); SS_EPI(I, 2); SS_EPI(I, 3)
            SS_EPI_ROW(0); SS_EPI_ROW(1); SS_EPI_ROW(2); SS_EPI_ROW(3);
#undef SS_EPI_ROW
#undef SS_EPI
        }
        if (!has_next) break;
        __syncthreads();      // the C tile in LDS is fully flushed before the next item restages
    }
#undef SS_SETUP
}

// ================================================================ KC x KC kernel with direct global->LDS staging
// Same tile / MFMA / LDS image as gemm_kernel, but the K-tiles are copied by global_load_lds_dwordx4: no VGPR round trip and
// no ds_write instructions (the LDS-store path was the largest per-K-tile cost of the register-staged version).  The image
// must be lane-linear (dest = wave base + 16*lane), so the XOR swizzle is applied to the per-lane SOURCE chunk instead.
// Rows / columns beyond M / N are clamped to a valid row (their results are never stored); K must be a multiple of the
// K-tile (host-checked), so no zero fill is ever needed.  One barrier per K-tile; the copy of tile s+1 overlaps the MFMAs
// of tile s.  Persistent over items; the next item's first tile is requested before the epilogue and lands in the stage
// the epilogue does not use.
template <class T>
struct StageG {
    const T* src[4];      // per-lane source (row clamped, chunk pre-swizzled), advanced by BK per K-tile
    int wrow;             // first tile row written by this wave in each of the 4 pieces: 8*wave (+32*i)
    __device__ __forceinline__ void init(const T* p, const RowMap& map, int outer0, int outer_size, int k_begin, int tid) {
        const int lr = (tid & 63) >> 3, cpos = tid & 7, wave = tid >> 6;
        wrow = wave * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = wrow + lr + 32 * i;
            int o = outer0 + r; o = o < outer_size ? o : outer_size - 1;
            const int chunk = cpos ^ (r & 7) ^ (((r >> 3) & 3) << 1);              // inverse of swz(): same involution
            src[i] = p + rowmap_off(map, o) + k_begin + chunk * Elem<T>::EPC;
        }
    }
    __device__ __forceinline__ void issue(unsigned char* tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { glds16(src[i], tile + (wrow + 32 * i) * ROWB); src[i] += Elem<T>::BK; }
    }
};

template <class T, class TO, int X3 = 0>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const T* __restrict__ A, const T* __restrict__ B, TO* __restrict__ C,
                                                           int M, int N, int K, RowMap amap, RowMap bmap, GemmEpi epi,
                                                           int k_chunk, int tiles_m, int tiles_n, int nitems)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][BM * ROWB];
    constexpr int BK = Elem<T>::BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x, ntiles = tiles_m * tiles_n;
    StageG<T> sa, sb;
    int it = blockIdx.x, m0, n0, k_begin, k_end, cur = 0;
    item_coord(it, G, nitems, ntiles, tiles_n, k_chunk, K, m0, n0, k_begin, k_end);
    sa.init(A, amap, m0, M, k_begin, tid); sb.init(B, bmap, n0, N, k_begin, tid);
    sa.issue(lds[cur][0]); sb.issue(lds[cur][1]);
    for (;;) {
        const int nsteps = (k_end - k_begin) / BK;           // >= 1, exact
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        if constexpr (X3) {
            // bf16 x 3: a wave's split fragments of a K tile ARE its whole share of the tile (64 registers), so the stage is handed back to the copy
            // engine as soon as every wave holds them -- BEFORE the 48 MFMAs -- and the copy of tile s + 2 goes into the stage tile s was just read
            // from: two K tiles in flight per workgroup on two LDS stages.  Worth 3-8 % (22 000 x 3072 x 768: 458 -> 436 us): the copies are bound by
            // CU <-> L2 THROUGHPUT, not latency -- the launch with the MMA removed takes 276 us either way (tools/x3_gemm_probe.py: 3.2 GB of tile
            // bytes = 11.5 TB/s; FETCH_SIZE says 0.4 GB of it misses L2), because a 128 x 128 tile of f32 operands is 32 flop per byte.
            // Vector-memory operations retire in issue order: with tiles s and s + 1 outstanding (8 copies per thread each), vmcnt(8) is "tile s has landed".
            for (int s = 0; s < nsteps; ++s) {
                if (s == 0) {
                    __syncthreads();                          // tile 0 (requested by the prologue / under the previous item's epilogue) has landed; stage cur^1 is free
                    if (nsteps > 1) { sa.issue(lds[cur ^ 1][0]); sb.issue(lds[cur ^ 1][1]); }
                } else {
                    if (s + 1 < nsteps) wait_vmcnt<8>(); else wait_vmcnt<0>();
                    barrier_keep_vm();
                }
                TileMmaX3::Frags f;
                TileMmaX3::load(lds[cur][0], lds[cur][1], wm, wn, lane, f);
                if (s + 2 < nsteps) {
                    barrier_keep_vm();                        // every wave holds its fragments of tile s
                    sa.issue(lds[cur][0]); sb.issue(lds[cur][1]);
                }
                if (!(epi.debug & 4)) TileMmaX3::mma(f, acc);
                cur ^= 1;
            }
        } else {
            for (int s = 0; s < nsteps; ++s) {
                __syncthreads();                              // tile s has landed in stage `cur`; stage cur^1 is free
                if (s + 1 < nsteps) { sa.issue(lds[cur ^ 1][0]); sb.issue(lds[cur ^ 1][1]); }
                if (!(epi.debug & 4)) MmaSel<false, T, X3>::type::run(lds[cur][0], lds[cur][1], wm, wn, lane, acc);
                cur ^= 1;
            }
        }
        // `cur` now names the stage NOT read by the last K-tile: the next item's first tile goes there
        const int cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) {
            it += G;
            item_coord(it, G, nitems, ntiles, tiles_n, k_chunk, K, m0, n0, k_begin, k_end);
            sa.init(A, amap, m0, M, k_begin, tid); sb.init(B, bmap, n0, N, k_begin, tid);
            sa.issue(lds[cur][0]); sb.issue(lds[cur][1]);
        }
        barrier_keep_vm();                                    // every wave is done reading stage cur^1 -> it becomes the C tile
        // ---- epilogue through the free stage (32 KiB): R rows per pass
        TO* ct = (TO*)&lds[cur ^ 1][0][0];
        constexpr int LDC = BN + 16 / (int)sizeof(TO);
        const int cq = lane >> 4, cr = lane & 15;
        if (sizeof(TO) == 2 && epi.c2_lds) {
            // QKV / dO projections: the result AND its per-sequence transposed copy ([b][col][t], read by the attention
            // kernels) leave through LDS as 16-byte stores: 32-row passes, C piece [32][136] + transposed piece [128][40].
            constexpr int LDT = 40;
            TO* tt = ct + 32 * LDC;
#define SS_EPS(I, J) epilogue_stage<TO, 0, true>(acc[I][J], ct, LDC, (I & 1) * 16 + cq * 4, wn * 64 + J * 16 + cr, epi, cm0 + wm * 64 + I * 16 + cq * 4, cn0 + wn * 64 + J * 16 + cr, M, N, tt, LDT)
#define SS_EPS_ROW(I) SS_EPS(I, 0); SS_EPS(I, 1); SS_EPS(I, 2); SS_EPS(I, 3)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                if (wm == pass / 2) { if (pass & 1) { SS_EPS_ROW(2); SS_EPS_ROW(3); } else { SS_EPS_ROW(0); SS_EPS_ROW(1); } }
                barrier_keep_vm();
                epilogue_flush<TO>(ct, LDC, C, epi, cm0 + pass * 32, 32, cn0, M, N, tid);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = tid + 256 * i, col = idx >> 2, ch = idx & 3;
                    const int row0 = cm0 + pass * 32 + ch * 8;
                    if (cn0 + col < N && row0 < M) {
                        const u32x4 v = *(const u32x4*)(tt + col * LDT + ch * 8);
                        *(u32x4*)((TO*)epi.c2 + (long long)(cn0 + col) * epi.col_stride2 + rowmap_off(epi.cmap2, row0)) = v;
                    }
                }
                if (pass < 3) barrier_keep_vm();
            }
#undef SS_EPS_ROW
#undef SS_EPS
        } else {
            constexpr int R = sizeof(TO) == 2 ? 64 : 32;
#define SS_EPS(I, J) epilogue_stage<TO, GEN>(acc[I][J], ct, LDC, (I & (R / 16 - 1)) * 16 + cq * 4, wn * 64 + J * 16 + cr, epi, cm0 + wm * 64 + I * 16 + cq * 4, cn0 + wn * 64 + J * 16 + cr, M, N)
#define SS_EPS_ROW(I) SS_EPS(I, 0); SS_EPS(I, 1); SS_EPS(I, 2); SS_EPS(I, 3)
#pragma unroll
            for (int pass = 0; pass < BM / R; ++pass) {
#define SS_EPS_PASS                                                                                                                  \
                if (R == 64) { if (wm == pass) { SS_EPS_ROW(0); SS_EPS_ROW(1); SS_EPS_ROW(2); SS_EPS_ROW(3); } }                              \
                else if (wm == pass / 2) { if (pass & 1) { SS_EPS_ROW(2); SS_EPS_ROW(3); } else { SS_EPS_ROW(0); SS_EPS_ROW(1); } }
                if (epi.general == 1) { constexpr int GEN = 1; SS_EPS_PASS } else if (epi.general == 2) { constexpr int GEN = 2; SS_EPS_PASS } else { constexpr int GEN = 0; SS_EPS_PASS }
#undef SS_EPS_PASS
                barrier_keep_vm();
                if (!(epi.debug & 1)) epilogue_flush<TO>(ct, LDC, C, epi, cm0 + pass * R, R, cn0, M, N, tid);
                if (pass + 1 < BM / R) barrier_keep_vm();
            }
#undef SS_EPS_ROW
#undef SS_EPS
        }
        if (!has_next) break;
    }
}

// ================================================================ KC x KC bf16 kernel, 2 waves x (128 x 64) per tile
// Same 128 x 128 output tile and persistent schedule, but the tile belongs to TWO waves that each keep a 128 x 64 block
// (8 x 4 MFMA tiles, 128 accumulator registers): one K step of 32 needs 8 + 4 = 12 fragment reads for 32 MFMAs
// (0.375 ds_read_b128 per MFMA instead of 0.5), a K tile is 32 deep (16 KiB per stage, 32 KiB per workgroup) so FOUR
// workgroups share a CU (still 8 waves), and the per-tile barrier involves 2 waves instead of 4.
// LDS image: dense 64-byte rows; 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 3), which spreads the
// 8 rows served together by a ds_read_b128 over all 32 banks.  Staging is global_load_lds (the swizzle is applied to the
// per-lane source chunk, the destination stays lane-linear).
namespace w2 {
constexpr int BK2 = 32, ROW2 = 64;
// ROWS tile rows, copied 16 rows per wave instruction; wave w takes the 16-row pieces w, w+2, w+4, ...
template <int ROWS>
struct Stage {
    static constexpr int NP = (ROWS / 16 + 1) / 2;        // pieces per wave (the last one may belong to wave 0 only)
    const bf16_t* src[NP];
    int wave;
    __device__ __forceinline__ void init(const bf16_t* p, const RowMap& map, int outer0, int outer_size, int k_begin, int tid) {
        const int lane = tid & 63;
        wave = tid >> 6;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = (i * 2 + wave) * 16 + (lane >> 2);
            int o = outer0 + r; o = o < outer_size ? o : outer_size - 1;
            const int chunk = (lane & 3) ^ ((r >> 1) & 3);
            src[i] = p + rowmap_off(map, o) + k_begin + chunk * 8;
        }
    }
    __device__ __forceinline__ void issue(unsigned char* tile) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if ((i * 2 + wave) * 16 < ROWS) glds16(src[i], tile + (i * 2 + wave) * 16 * ROW2);
            src[i] += BK2;
        }
    }
};
__device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int row, int q) { return *(const bf16x8*)(tile + row * ROW2 + ((q ^ ((row >> 1) & 3)) << 4)); }

// epilogue pass P of a BMT-row tile: IPP 16-row MFMA tiles per pass go to the LDS piece, then 16-byte row-contiguous stores
template <class TO, int GEN, int NI, int IPP, int P, int NPASS>
struct Passes {
    static __device__ __forceinline__ void run(const f32x4 (&acc)[NI][4], TO* ct, int ldc, TO* C, const GemmEpi& epi, int cm0, int cn0, int M, int N, int tid, int wn, int r, int q) {
#pragma unroll
        for (int ii = 0; ii < IPP; ++ii) {
            constexpr int I0 = P * IPP;
            if (I0 + ii < NI) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    epilogue_stage<TO, GEN>(acc[I0 + ii < NI ? I0 + ii : 0][j], ct, ldc, ii * 16 + q * 4, wn * 64 + j * 16 + r, epi, cm0 + (I0 + ii) * 16 + q * 4, cn0 + wn * 64 + j * 16 + r, M, N);
            }
        }
        barrier_keep_vm();
        epilogue_flush<TO, 128>(ct, ldc, C, epi, cm0 + P * IPP * 16, IPP * 16, cn0, M, N, tid);
        barrier_keep_vm();
        Passes<TO, GEN, NI, IPP, P + 1, NPASS>::run(acc, ct, ldc, C, epi, cm0, cn0, M, N, tid, wn, r, q);
    }
};
template <class TO, int GEN, int NI, int IPP, int NPASS>
struct Passes<TO, GEN, NI, IPP, NPASS, NPASS> {
    static __device__ __forceinline__ void run(const f32x4 (&)[NI][4], TO*, int, TO*, const GemmEpi&, int, int, int, int, int, int, int, int) {}
};
}  // namespace w2

template <class TO, int BMT>
__global__ __launch_bounds__(128, 2) void gemm_w2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                         int M, int N, int K, RowMap amap, RowMap bmap, GemmEpi epi,
                                                         int k_chunk, int tiles_m, int tiles_n, int nitems)
{
    constexpr int NI = BMT / 16, STAGE = (BMT + BN) * w2::ROW2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, r = lane & 15, q = lane >> 4;
    const int G = gridDim.x, ntiles = tiles_m * tiles_n;
    w2::Stage<BMT> sa; w2::Stage<BN> sb;
    int it = blockIdx.x, m0, n0, k_begin, k_end, cur = 0;
    item_coord(it, G, nitems, ntiles, tiles_n, k_chunk, K, m0, n0, k_begin, k_end); m0 = m0 / BM * BMT;
    sa.init(A, amap, m0, M, k_begin, tid); sb.init(B, bmap, n0, N, k_begin, tid);
    sa.issue(lds[cur]); sb.issue(lds[cur] + BMT * w2::ROW2);
    for (;;) {
        const int nsteps = (k_end - k_begin) / w2::BK2;
        f32x4 acc[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        for (int s = 0; s < nsteps; ++s) {
            __syncthreads();                                  // K tile s has landed in stage `cur`; stage cur^1 is free
            if (s + 1 < nsteps) { sa.issue(lds[cur ^ 1]); sb.issue(lds[cur ^ 1] + BMT * w2::ROW2); }
            const unsigned char* As = lds[cur]; const unsigned char* Bs = lds[cur] + BMT * w2::ROW2;
            bf16x8 a[NI], b[4];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[i] = w2::frag(As, i * 16 + r, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = w2::frag(Bs, wn * 64 + j * 16 + r, q);
#if !defined(SS_EMU)
            __builtin_amdgcn_sched_barrier(0);               // all fragment reads in flight before the first MFMA
#endif
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16_16x16x32(a[i], b[j], acc[i][j]);
            cur ^= 1;
        }
        const int cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        if (has_next) {
            it += G;
            item_coord(it, G, nitems, ntiles, tiles_n, k_chunk, K, m0, n0, k_begin, k_end); m0 = m0 / BM * BMT;
            sa.init(A, amap, m0, M, k_begin, tid); sb.init(B, bmap, n0, N, k_begin, tid);
            sa.issue(lds[cur]); sb.issue(lds[cur] + BMT * w2::ROW2);
        }
        barrier_keep_vm();                                    // both waves are done reading stage cur^1 -> it becomes the C piece
        // ---- epilogue through the free stage: R-row pieces [R][BN + pad], 16-byte row-contiguous stores
        TO* ct = (TO*)&lds[cur ^ 1][0];
        constexpr int LDC = BN + 16 / (int)sizeof(TO);
        constexpr int IPP = sizeof(TO) == 2 ? (BMT % 48 == 0 ? 3 : 2) : 1;            // 16-row MFMA tiles per pass: 48 / 32 rows (bf16), 16 (f32)
        constexpr int NPASS = (NI + IPP - 1) / IPP;
        static_assert((size_t)IPP * 16 * LDC * sizeof(TO) <= (size_t)STAGE, "C piece does not fit the free stage");
        if (epi.general == 1) w2::Passes<TO, 1, NI, IPP, 0, NPASS>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid, wn, r, q);
        else if (epi.general == 2) w2::Passes<TO, 2, NI, IPP, 0, NPASS>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid, wn, r, q);
        else w2::Passes<TO, 0, NI, IPP, 0, NPASS>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid, wn, r, q);
        if (!has_next) break;
    }
}

// ---------------------------------------------------------------- host launcher
static RowMap to_rowmap(const ss_rowmap* m) {
    RowMap r; r.base = m->base; r.batch_stride = m->batch_stride; r.row_stride = m->row_stride; r.rows_per_batch = m->rows_per_batch > 0 ? m->rows_per_batch : 0x7fffffff;
    return r;
}

// resident block slots: 2 blocks (64 KiB LDS, <=256 registers) per CU
static thread_local int g_blocks_per_cu = 2;
static thread_local int g_last_kernel = 0;
extern "C" int ss_gemm_last_kernel(void) { return g_last_kernel; }
extern "C" int ss_gemm_set_blocks_per_cu(int n) { int old = g_blocks_per_cu; if (n >= 1 && n <= 2) g_blocks_per_cu = n; return old; }

// Kernel-selection knobs (process-wide; tests and the tuning tools flip them, the defaults come from the environment once):
//   0 SS_GEMM_W2     2-wave 128/144 x 128 kernel: 0 never, 1 cost model, 2 whenever legal
//   1 SS_GEMM_W2_BM  force its tile height (128 / 144), 0 = cost model
//   2 SS_GEMM8       8-wave 256/288 x 256 kernel: 0 never, 1 cost model, 2 whenever legal
//   3 SS_GEMM8_NI    force its tile height in 16-row units per M-wave (8 / 9), 0 = cost model
//   4 SS_GEMM8_PIN   bit 0: its fragment reads, bit 1: its DMA pieces spread between the MFMA groups (else a burst per phase); values 0..3.
//                    (Tuning builds, -DG8_FAST_BUILD, additionally read bits 2..3 as 16-row tiles per phase; the release build ignores
//                    values above 3 and keeps the default schedule, 3.)
//   5 SS_GEMM_DEBUG  ablation mask for tuning (results are then wrong): 1 no flush stores, 2 no epilogue staging, 4 no MFMA (128-wide
//                    kernels); 16 no MFMA, 32 no in-loop global->LDS copies, 64 no in-loop fragment reads, 128 no C flush (8-wave kernel)
//   6 SS_GEMM_SMALLK the LDS-free K <= 32 kernel (gemm_smallk.hip): 0 never, 1 whenever legal
enum { OPT_W2 = 0, OPT_W2_BM, OPT_G8, OPT_G8_NI, OPT_G8_PIN, OPT_DEBUG, OPT_SMALLK, OPT_COUNT };
static int g_opt[OPT_COUNT] = {-1, -1, -1, -1, -1, -1, -1};
static int gemm_opt(int what) {
    static const char* names[OPT_COUNT] = {"SS_GEMM_W2", "SS_GEMM_W2_BM", "SS_GEMM8", "SS_GEMM8_NI", "SS_GEMM8_PIN", "SS_GEMM_DEBUG", "SS_GEMM_SMALLK"};
    static const int defaults[OPT_COUNT] = {1, 0, 1, 0, 16, 0, 1};
    if (g_opt[what] < 0) { const char* e = getenv(names[what]); g_opt[what] = e ? atoi(e) : defaults[what]; }
    return g_opt[what];
}
extern "C" int ss_gemm_set_option(int what, int value) {
    if (what < 0 || what >= OPT_COUNT) return -1;
    const int old = gemm_opt(what);
    g_opt[what] = value < 0 ? -1 : value;          // negative: back to the environment / default
    return old;
}

static int gemm_slots() {
#if defined(SS_EMU)
    return 3;            // tiny on purpose: the emulator tests exercise the multi-item path of the persistent loop
#else
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = cus;
    }
    return slots * g_blocks_per_cu;
#endif
}

// Does this problem run the 8-wave 256 / 288 x 256 kernel (gemm8.hip: one workgroup per CU, fragment reads travelling under the MFMAs)?
// Legality (bf16, both operands K-contiguous, K % 64 == 0, LDS-staged epilogue, 32-bit operand offsets) + a measured cost model
// (tools/gemm_bench.cpp, 22 000 .. 88 000 rows, us): a tile costs 13 + 0.0245 K (256 rows) or 13 + 0.0295 K (288 rows) while the whole chip
// streams; a last round of `rem` tiles runs faster (0.55 + 0.45 rem / CUs of a tile).  The 128-wide kernels sustain ~560 TFLOP/s at
// K <= 1024 and ~680 beyond, plus ~6 us.
// kmul = 3: the hi / lo plane form (ss_gemm_planes) -- the contraction the K loop walks is 3 K long, and there is no 128-wide alternative to compare with
// which of the 8-wave kernel's epilogues carries the sign-bit paths: the producer sits in the register epilogue (bf16 out, no column statistics), the consumer in
// the C-piece epilogue of the column-sum instantiations (bf16 out)
template <class TO> static bool sign_paths_ok(const GemmEpi& epi) {
    if (sizeof(TO) != 2) return false;
    if (epi.sign_out && (epi.col_sum || epi.general == 2)) return false;
    if (epi.gate_bits && !(epi.col_sum && !epi.col_sumsq && !epi.col_shift)) return false;
    return true;
}
static bool pick_gemm8(bool bf16_in, int a_mode, int b_mode, int M, int N, int K, const RowMap& am, const RowMap& bm, const GemmEpi& epi, int split_k, int* ni_out, int* pin_out, int kmul = 1)
{
    if (!(bf16_in && gemm_opt(OPT_G8) && a_mode == OP_KC && b_mode == OP_KC && epi.fast && !epi.c2 && split_k == 1 && K % 64 == 0 && K > 0)) return false;
    if (epi.general == 1 && epi.col_sum) return false;       // column statistics of a dropout epilogue: not instantiated (nothing asks for it)
    if (epi.general == 2) return false;                      // log-clamp epilogue (the f32 mel GEMM): not instantiated in the 8-wave kernel (its logf path only cost registers there)
    const int cus = gemm_slots() / g_blocks_per_cu;
    auto max_off = [](const RowMap& m, int rows, int K_) {     // largest element offset the kernel forms for this operand
        const long long nb = m.rows_per_batch == 0x7fffffff ? 0 : (rows - 1) / m.rows_per_batch;
        const long long rr = m.rows_per_batch == 0x7fffffff ? rows - 1 : m.rows_per_batch - 1;
        return m.base + nb * (m.batch_stride > 0 ? m.batch_stride : 0) + rr * (m.row_stride > 0 ? m.row_stride : 0) + K_;
    };
    const bool fits = am.base >= 0 && bm.base >= 0 && am.row_stride >= 0 && bm.row_stride >= 0 && am.batch_stride >= 0 && bm.batch_stride >= 0 &&
                      max_off(am, M, K) * 2 < 0xffffffffLL && max_off(bm, N, K) * 2 < 0xffffffffLL;
    if (!fits) return false;
    const int ni_force = gemm_opt(OPT_G8_NI);
    double best8 = 0; int ni = 0;
    for (int cand = 8; cand <= 9; ++cand) {
        if (ni_force && cand != ni_force) continue;
        const int bmt = 32 * cand;
        const long long t = (long long)((M + bmt - 1) / bmt) * ((N + 255) / 256);
        const double tile = 13.0 + (cand == 9 ? 0.0295 : 0.0245) * K * kmul;
        const long long full = t / cus, rem = t % cus;
        const double cost = 5.0 + full * tile + (rem ? tile * (0.55 + 0.45 * (double)rem / cus) : 0.0);
        if (!ni || cost < best8) { best8 = cost; ni = cand; }
    }
    const double cost_old = 6.0 + 2.0 * M * N * K / ((K <= 1024 ? 560.0 : 680.0) * 1e6);
    if (!ni || !(gemm_opt(OPT_G8) == 2 || kmul > 1 || best8 < cost_old)) return false;
    // Spreading the fragment reads / DMA pieces between the MFMA groups (PIN = 3) is the default for both tile heights: since the
    // steady K steps are one straight-line block with scalar-base copies (gemm8.hip) the 288-row variant no longer spills in the loop and
    // beats the burst schedule by 0..16 % depending on the shape and the box (tools/gemm_bench.cpp: 22 000 x 768 x 3072 96 vs 113 us).
    *ni_out = ni;
#if defined(G8_FAST_BUILD)
    *pin_out = gemm_opt(OPT_G8_PIN) >= 0 && gemm_opt(OPT_G8_PIN) <= 15 ? gemm_opt(OPT_G8_PIN) : 3;      // tuning builds: bits 2..3 select tiles per phase
#else
    *pin_out = gemm_opt(OPT_G8_PIN) >= 0 && gemm_opt(OPT_G8_PIN) <= 3 ? gemm_opt(OPT_G8_PIN) : 3;       // release: schedules 0..3 exist; anything else = the default
#endif
    return true;
}

template <class T, class TO, int X3 = 0>
static int launch_gemm(int a_mode, int b_mode, const void* A, const void* B, void* C, int M, int N, int K,
                       const RowMap& am, const RowMap& bm, const GemmEpi& epi, int split_k, void* stream)
{
    constexpr int BK = Elem<T>::BK;
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    if (split_k < 1) split_k = 1;
    int ksteps = (K + BK - 1) / BK;
    int per = (ksteps + split_k - 1) / split_k;
    int k_chunk = per * BK;
    split_k = (ksteps + per - 1) / per;
    const int nitems = tiles_m * tiles_n * split_k;
    const int slots = gemm_slots();
    dim3 grid(nitems < slots ? nitems : slots), block(256);
    if (epi.sign_out || epi.gate_bits) {          // only the 8-wave kernel carries these epilogue paths: anything else would silently ignore them
        int ni = 0, pin = 0;
        if (!sign_paths_ok<TO>(epi) || !pick_gemm8(sizeof(T) == 2, a_mode, b_mode, M, N, K, am, bm, epi, split_k, &ni, &pin) || (epi.sign_out && pin != 3)) {
            ss_set_error("ss_gemm: sign_out / gate_bits need the 8-wave kernel (bf16; sign_out: no column statistics, default schedule; gate_bits: with col_sum) -- ask ss_gemm_sign_bits_supported / ss_gemm_fuses_column_stats first");
            return 1;
        }
        if (gemm8_launch_kc<TO>(ni, pin, A, B, C, M, N, K, am, bm, epi, stream)) return 1;
        g_last_kernel = ni == 9 ? 4 : 3;
        return 0;
    }
    if (gemm_opt(OPT_SMALLK) && gemm_smallk_ok(sizeof(T) == 2 ? SS_BF16 : SS_F32, sizeof(TO) == 2 ? SS_BF16 : SS_F32, a_mode, b_mode, A, B, C, M, N, K, am, bm, epi, split_k)) {
        if (gemm_smallk_launch(A, B, C, M, N, K, am, bm, epi, stream)) return 1;       // K <= 32: register-resident weights, no LDS (gemm_smallk.hip)
        g_last_kernel = 5;
        return 0;
    }
    {   // 8-wave 256 / 288 x 256 kernel (gemm8.hip)
        int ni = 0, pin = 0;
        if (pick_gemm8(sizeof(T) == 2, a_mode, b_mode, M, N, K, am, bm, epi, split_k, &ni, &pin)) {
            if (gemm8_launch_kc<TO>(ni, pin, A, B, C, M, N, K, am, bm, epi, stream)) return 1;
            g_last_kernel = ni == 9 ? 4 : 3;
            return 0;
        }
        if (epi.col_sum) { ss_set_error("ss_gemm: epilogue column statistics need the 8-wave kernel for this shape (ask ss_gemm_fuses_column_stats first)"); return 1; }
    }
    {   // 2-wave kernel: bf16 KC x KC with the plain LDS-staged epilogue (no transposed second output)
        const int w2_on = gemm_opt(OPT_W2);                                                  // 0 never, 1 heuristic, 2 whenever possible
        if (w2_on && sizeof(T) == 2 && a_mode == OP_KC && b_mode == OP_KC && epi.fast && !epi.c2 && split_k == 1 && K % 32 == 0) {
            // One workgroup slot = a quarter of a CU (4 resident workgroups).  The tile height (128 or 144 rows) is the one that
            // wastes the least of the last round: 22 000 rows x 768 columns are 1032 tiles of 128 rows (a round of 1024 plus an
            // 8-tile tail on 2 of a CU's 4 SIMDs) but 918 tiles of 144 rows -- one round.
            // Cost model in "rows a CU works through": a round of the 4-wave kernel is 2 tiles of 128 rows per CU, a round of this
            // kernel 4 tiles of BMT rows; a remainder below 1/8 of the slots still costs ~0.55 of a round (its tiles run alone).
            // Measured anchors (22 000 rows): N=3072, K=768: this kernel +16 %; N=768, K=3072 with 128-row tiles: -21 % (the model's
            // 794 vs 653), i.e. the lower fixed cost per tile is worth ~20 % at K <= 1024 and nothing at long K.
            const int slots2 = slots / g_blocks_per_cu * 4;
            const int bm_force = gemm_opt(OPT_W2_BM);                                                        // tests: force the tile height
            double best = 0; int bmt = 0, tm_best = 0;
            for (int cand = 128; cand <= 144; cand += 16) {
                if (bm_force && cand != bm_force) continue;
                const int tm = (M + cand - 1) / cand; const long long t = (long long)tm * tiles_n;
                const long long full = t / slots2, rem = t % slots2;
                const double rounds = (double)full + (rem == 0 ? 0.0 : (rem * 8 < slots2 ? 0.55 : 1.0));
                const double cost = rounds * 4 * cand;
                if (!bmt || cost < best) { best = cost; bmt = cand; tm_best = tm; }
            }
            const long long t4 = (long long)tiles_m * tiles_n;
            const double rounds4 = (double)(t4 / slots) + (t4 % slots == 0 ? 0.0 : ((t4 % slots) * 8 < slots ? 0.55 : 1.0));
            const double cost4 = rounds4 * 2 * 128;
            const bool w2_shape = w2_on == 2 || best * (K <= 1024 ? 0.8 : 1.0) < 0.97 * cost4;
            if (w2_shape) {
                const int nitems2 = tm_best * tiles_n;
                dim3 grid2(nitems2 < slots2 ? nitems2 : slots2);
                if (bmt == 144) SS_LAUNCH(SS_KERNEL(gemm_w2_kernel<TO, 144>), grid2, dim3(128), 0, stream, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, M, N, K, am, bm, epi, K, tm_best, tiles_n, nitems2);
                else SS_LAUNCH(SS_KERNEL(gemm_w2_kernel<TO, 128>), grid2, dim3(128), 0, stream, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, M, N, K, am, bm, epi, K, tm_best, tiles_n, nitems2);
                SS_LAUNCH_CHECK("ss_gemm(w2)");
                g_last_kernel = 2;
                return 0;
            }
        }
    }
    if (a_mode == OP_KC && b_mode == OP_KC && epi.fast && K % BK == 0 && k_chunk % BK == 0 && !(epi.debug & 8)) {
        SS_LAUNCH(SS_KERNEL(gemm_glds_kernel<T, TO, X3>), grid, block, 0, stream, (const T*)A, (const T*)B, (TO*)C, M, N, K, am, bm, epi, k_chunk, tiles_m, tiles_n, nitems);
        SS_LAUNCH_CHECK("ss_gemm(glds)");
        g_last_kernel = 1;
        return 0;
    }
    // (measured: a global_load_lds + XOR-swizzled ds_read_b64_tr_b16 variant of the OC x OC kernel is ~7 % SLOWER than the
    //  register-staged, 288-byte-pitch one below -- the transpose read's banking is not fixed by address swizzles.)
#define SS_GEMM_CASE(AM, BMD)                                                                                     \
    SS_LAUNCH(SS_KERNEL(gemm_kernel<T, TO, AM, BMD, X3>), grid, block, 0, stream, (const T*)A, (const T*)B, (TO*)C, M, N, K, am, bm, epi, k_chunk, tiles_m, tiles_n, nitems)
    if (a_mode == OP_KC && b_mode == OP_KC) SS_GEMM_CASE(OP_KC, OP_KC);
    else if (a_mode == OP_KC && b_mode == OP_OC) SS_GEMM_CASE(OP_KC, OP_OC);
    else if (a_mode == OP_OC && b_mode == OP_OC) SS_GEMM_CASE(OP_OC, OP_OC);
    else if (a_mode == OP_OC && b_mode == OP_KC) SS_GEMM_CASE(OP_OC, OP_KC);
#undef SS_GEMM_CASE
    SS_LAUNCH_CHECK("ss_gemm");
    g_last_kernel = 0;
    return 0;
}

static int build_epi(GemmEpi& epi, int dtype_out, const void* C, int M, int N, const ss_rowmap* cmap, const ss_gemm_epilogue* e, int split_k)
{
    memset(&epi, 0, sizeof(epi));
    epi.alpha = 1.f; epi.gate_scale = 1.f; epi.drop_scale = 1.f;
    epi.cmap = to_rowmap(cmap);
    if (e) {
        epi.bias = e->bias; epi.gate = e->gate; epi.gate_scale = e->gate_scale; epi.alpha = e->alpha; epi.relu = e->relu;
        if (e->dropout_p > 0.f) { epi.drop_thresh = dropout_threshold(e->dropout_p); epi.drop_scale = 1.f / (1.f - e->dropout_p); }
        epi.seed = e->seed; epi.stream = e->rng_stream; epi.mode = e->mode;
        epi.col_mod = e->col_mod; epi.col_mul = e->col_mul; epi.col_div_mul = e->col_div_mul;
        epi.log_clamp = e->log_clamp;
        if (e->c2) { epi.c2 = e->c2; epi.cmap2 = to_rowmap(&e->cmap2); epi.col_stride2 = e->col_stride2; }
        epi.col_sum = e->col_sum; epi.col_sumsq = e->col_sumsq; epi.col_shift = e->col_shift;
        epi.planes_hi = e->planes_hi; epi.planes_lo = e->planes_lo; epi.planes_only = e->planes_only;
        epi.sign_out = (unsigned char*)e->sign_out; epi.sign_pitch = e->sign_pitch; epi.gate_bits = (const unsigned char*)e->gate_bits; epi.gate_bits_pitch = e->gate_bits_pitch;
        SS_CHECK(!(e->sign_out || e->gate_bits) || N % 8 == 0, "ss_gemm: sign bits need N %% 8 == 0 (N = %d)", N);
        SS_CHECK(!e->sign_out || (dtype_out == SS_BF16 && e->sign_pitch * 8 >= N && !e->col_mod), "ss_gemm: sign_out needs bf16 results, sign_pitch >= N / 8 and no column permutation");
        SS_CHECK(!e->gate_bits || (!e->gate && e->gate_bits_pitch * 8 >= N && !e->col_mod), "ss_gemm: gate_bits replaces gate (not both), gate_bits_pitch >= N / 8, no column permutation");
        SS_CHECK((e->planes_hi != nullptr) == (e->planes_lo != nullptr) && !(e->planes_only && !e->planes_hi), "ss_gemm: planes_hi / planes_lo come together (planes_only needs them)");
        SS_CHECK(!e->planes_hi || (dtype_out == SS_F32 && e->mode != 2 && !e->c2 && !e->col_mod && ((uintptr_t)e->planes_hi | (uintptr_t)e->planes_lo) % 8 == 0),
                 "ss_gemm: plane output needs f32 results, no atomic mode / second copy / column permutation and 8-byte aligned planes");
        SS_CHECK(!(e->col_sumsq && !e->col_sum), "ss_gemm: col_sumsq needs col_sum");
        SS_CHECK(e->mode >= 0 && e->mode <= 2, "ss_gemm: bad output mode %d", e->mode);
        SS_CHECK(!(e->mode == 2 && dtype_out != SS_F32), "ss_gemm: atomic accumulation needs f32 output");
        SS_CHECK(!(split_k > 1 && e->mode != 2), "ss_gemm: split_k > 1 needs mode 2 (atomic accumulate)");
    } else {
        SS_CHECK(split_k <= 1, "ss_gemm: split_k > 1 needs mode 2 (atomic accumulate)");
    }
    epi.debug = gemm_opt(OPT_DEBUG);
    {
        const int ev = dtype_out == SS_BF16 ? 8 : 4;
        const RowMap& cm = epi.cmap;
        epi.general = epi.drop_thresh != 0 ? 1 : (epi.log_clamp > 0.f ? 2 : 0);
        epi.fast = !(epi.drop_thresh != 0 && epi.log_clamp > 0.f) && epi.mode != 2 && epi.col_mod == 0 && N % ev == 0 && cm.base % ev == 0 && cm.batch_stride % ev == 0 && cm.row_stride % ev == 0 &&
                   ((uintptr_t)C) % 16 == 0 && (!epi.gate || ((uintptr_t)epi.gate) % 16 == 0);
        if (epi.c2) {
            const int pk = 4;      // rows per lane
            const size_t osz = dtype_out == SS_BF16 ? 2 : 4;
            const RowMap& c2m = epi.cmap2;
            epi.c2_lds = dtype_out == SS_BF16 && !epi.general && c2m.row_stride == 1 && c2m.rows_per_batch % 8 == 0 && M % 8 == 0 && c2m.base % 8 == 0 &&
                         c2m.batch_stride % 8 == 0 && epi.col_stride2 % 8 == 0 && ((uintptr_t)epi.c2) % 16 == 0;
            epi.c2_pack = c2m.row_stride == 1 && c2m.rows_per_batch % pk == 0 && c2m.base % pk == 0 && c2m.batch_stride % pk == 0 && epi.col_stride2 % pk == 0 &&
                          ((uintptr_t)epi.c2) % (pk * osz) == 0;
        }
    }
    return 0;
}

extern "C" int ss_gemm(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* A, const void* B, void* C,
                       int M, int N, int K, const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap,
                       const ss_gemm_epilogue* e, int split_k, void* stream)
{
    SS_CHECK(A && B && C && amap && bmap && cmap, "ss_gemm: null pointer");
    SS_CHECK(M >= 0 && N >= 0 && K >= 0, "ss_gemm: negative size");
    if (M == 0 || N == 0) return 0;
    SS_CHECK(dtype_in == SS_F32 || dtype_in == SS_BF16 || dtype_in == SS_F32X3, "ss_gemm: bad input dtype %d", dtype_in);
    SS_CHECK(dtype_out == SS_F32 || dtype_out == SS_BF16, "ss_gemm: bad output dtype %d", dtype_out);
    SS_CHECK(!(dtype_in != SS_BF16 && dtype_out == SS_BF16), "ss_gemm: f32 inputs with bf16 output is not a supported combination");
    const int epc = dtype_in == SS_BF16 ? 8 : 4;
    const size_t esz = dtype_in == SS_BF16 ? 2 : 4;
    // 16-byte vector staging: reduction extent (KC) / outer extent (OC) and all strides in whole chunks
    if (a_mode == OP_KC) SS_CHECK(K % epc == 0, "ss_gemm: K=%d must be a multiple of %d for a KC operand", K, epc);
    else SS_CHECK(M % epc == 0, "ss_gemm: M=%d must be a multiple of %d for an OC A operand", M, epc);
    if (b_mode == OP_KC) SS_CHECK(K % epc == 0, "ss_gemm: K=%d must be a multiple of %d for a KC operand", K, epc);
    else SS_CHECK(N % epc == 0, "ss_gemm: N=%d must be a multiple of %d for an OC B operand", N, epc);
    const ss_rowmap* maps[2] = {amap, bmap};
    const void* ptrs[2] = {A, B};
    for (int i = 0; i < 2; ++i) {
        SS_CHECK(maps[i]->base % epc == 0 && maps[i]->batch_stride % epc == 0 && maps[i]->row_stride % epc == 0,
                 "ss_gemm: operand %d strides must be multiples of %d elements (16-byte rows)", i, epc);
        SS_CHECK(((uintptr_t)ptrs[i]) % 16 == 0, "ss_gemm: operand %d is not 16-byte aligned", i);
    }
    (void)esz;
    GemmEpi epi;
    if (build_epi(epi, dtype_out, C, M, N, cmap, e, split_k)) return 1;
    SS_CHECK(!epi.planes_hi, "ss_gemm: plane output is an epilogue of ss_gemm_planes only");
    RowMap am = to_rowmap(amap), bm = to_rowmap(bmap);
    if (dtype_in == SS_BF16 && dtype_out == SS_BF16) return launch_gemm<bf16_t, bf16_t>(a_mode, b_mode, A, B, C, M, N, K, am, bm, epi, split_k, stream);
    if (dtype_in == SS_BF16 && dtype_out == SS_F32) return launch_gemm<bf16_t, float>(a_mode, b_mode, A, B, C, M, N, K, am, bm, epi, split_k, stream);
    if (dtype_in == SS_F32X3) return launch_gemm<float, float, 1>(a_mode, b_mode, A, B, C, M, N, K, am, bm, epi, split_k, stream);
    return launch_gemm<float, float>(a_mode, b_mode, A, B, C, M, N, K, am, bm, epi, split_k, stream);
}

// ---- f32 operands as hi / lo bf16 planes (the parity-grade mode of the training plan on the 8-wave kernel)
static int planes_setup(int dtype_out, const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, const void* C, int M, int N, int K, const ss_rowmap* amap, const ss_rowmap* bmap,
                        const ss_rowmap* cmap, const ss_gemm_epilogue* e, GemmEpi& epi, int* ni, int* pin)
{
    if (!amap || !bmap || !cmap || dtype_out != SS_F32 || M <= 0 || N <= 0 || K <= 0 || K % 64) return 0;
    if (amap->base % 8 || amap->batch_stride % 8 || amap->row_stride % 8 || bmap->base % 8 || bmap->batch_stride % 8 || bmap->row_stride % 8) return 0;
    if (((uintptr_t)A_hi | (uintptr_t)A_lo | (uintptr_t)B_hi | (uintptr_t)B_lo) % 16) return 0;
    if (build_epi(epi, dtype_out, C, M, N, cmap, e, 1)) return 0;
    if (epi.sign_out || epi.gate_bits) return 0;                      // (bf16 results only)
    return pick_gemm8(true, OP_KC, OP_KC, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, 1, ni, pin, 3) ? 1 : 0;
}
extern "C" int ss_gemm_planes_supported(int dtype_out, const void* C, int M, int N, int K, const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* e)
{
    GemmEpi epi; int ni = 0, pin = 0;
    return planes_setup(dtype_out, (const void*)16, (const void*)16, (const void*)16, (const void*)16, C, M, N, K, amap, bmap, cmap, e, epi, &ni, &pin);
}
extern "C" int ss_gemm_planes(int dtype_out, const void* A_hi, const void* A_lo, const void* B_hi, const void* B_lo, void* C, int M, int N, int K,
                              const ss_rowmap* amap, const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* e, void* stream)
{
    SS_CHECK(A_hi && A_lo && B_hi && B_lo && C && amap && bmap && cmap, "ss_gemm_planes: null pointer");
    SS_CHECK(M >= 0 && N >= 0 && K >= 0, "ss_gemm_planes: negative size");
    if (M == 0 || N == 0) return 0;
    GemmEpi epi; int ni = 0, pin = 0;
    SS_CHECK(planes_setup(dtype_out, A_hi, A_lo, B_hi, B_lo, C, M, N, K, amap, bmap, cmap, e, epi, &ni, &pin),
             "ss_gemm_planes: M=%d N=%d K=%d with this epilogue does not run on the 8-wave kernel (ask ss_gemm_planes_supported; K %% 64 == 0, f32 out, 16-byte rows, no transposed second output)", M, N, K);
    if (gemm8_launch_kc<float>(ni, pin, A_hi, B_hi, C, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, stream, true,
                               (long long)((const char*)A_lo - (const char*)A_hi), (long long)((const char*)B_lo - (const char*)B_hi))) return 1;
    g_last_kernel = ni == 9 ? 4 : 3;
    return 0;
}

extern "C" int ss_gemm_fuses_column_stats(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* C, int M, int N, int K, const ss_rowmap* amap,
                                          const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* e, int split_k)
{
    if (!amap || !bmap || !cmap || dtype_in != SS_BF16 || (dtype_out != SS_BF16 && dtype_out != SS_F32) || M <= 0 || N <= 0) return 0;
    GemmEpi epi;
    if (build_epi(epi, dtype_out, C, M, N, cmap, e, split_k)) return 0;
    if (epi.gate_bits && !(dtype_out == SS_BF16 && sign_paths_ok<bf16_t>(epi))) return 0;
    if (!epi.gate_bits && gemm_opt(OPT_SMALLK) && gemm_smallk_ok(dtype_in, dtype_out, a_mode, b_mode, (const void*)16, (const void*)16, C, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, split_k)) return 1;
    int ni = 0, pin = 0;
    return pick_gemm8(true, a_mode, b_mode, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, split_k, &ni, &pin) ? 1 : 0;
}

extern "C" int ss_gemm_sign_bits_supported(int dtype_in, int dtype_out, int a_mode, int b_mode, const void* C, int M, int N, int K, const ss_rowmap* amap,
                                           const ss_rowmap* bmap, const ss_rowmap* cmap, const ss_gemm_epilogue* e, int split_k)
{
    if (!amap || !bmap || !cmap || dtype_in != SS_BF16 || dtype_out != SS_BF16 || M <= 0 || N <= 0 || N % 8) return 0;
    GemmEpi epi;
    if (build_epi(epi, dtype_out, C, M, N, cmap, e, split_k)) return 0;
    GemmEpi probe = epi; probe.sign_out = (unsigned char*)16; probe.sign_pitch = N / 8;
    if (!sign_paths_ok<bf16_t>(probe)) return 0;
    if (gemm_opt(OPT_SMALLK) && gemm_smallk_ok(dtype_in, dtype_out, a_mode, b_mode, (const void*)16, (const void*)16, C, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, split_k)) return 0;
    int ni = 0, pin = 0;
    return pick_gemm8(true, a_mode, b_mode, M, N, K, to_rowmap(amap), to_rowmap(bmap), epi, split_k, &ni, &pin) && pin == 3 ? 1 : 0;
}
