// gemm8.hip -- 8-wave, 256-column-tile MFMA kernels for the large contractions of the training step (gfx950, bf16).
//
// The 128 x 128 tiles of gemm.hip spend as long in per-tile fixed cost (first-tile latency, epilogue) and LDS fragment
// traffic (0.375-0.5 ds_read_b128 per MFMA, every read exposed in front of its MFMAs) as in the MFMAs themselves.
// Here ONE workgroup of 8 waves owns a CU (2 x 68 KiB LDS stages):
//
//   gemm8_kc_kernel  C[m][n] = sum_k A(m,k) B(n,k), both operands K-contiguous (forward convs / linears and every dX):
//       tile (2*NI*16) x 256, NI = 8 or 9 (256 or 288 rows -- 22 000 frames x 768 columns are 3 x 77 = 231 tiles of 288
//       rows = ONE round of the 256 CUs, where 256-row tiles need 258), waves 2 (M) x 4 (N), 16*NI x 64 per wave,
//       K tile = 64 (128-byte rows, XOR-swizzled 16-byte chunks), staged by global_load_lds (swizzle on the source chunk).
//       The K step is cut into 4 phases (K half x row half); every phase first requests the fragments of the NEXT phase
//       (ping-pong register sets) and then issues its own 16-20 MFMAs, so LDS reads travel under MFMAs; the single
//       barrier of a K step sits between phase 2 and 3, where the stage just consumed is handed back to the DMA and the
//       other stage (requested one full K step earlier) is first read.
//   gemm8_dw_kernel  C[m][n] (+)= sum_k A(k,m) B(k,n), both operands outer-contiguous, reduction over the B*T frames
//       (every dW = dY^T X), GROUPED: one persistent launch walks the (tile x K-slice) items of up to 8 problems, so
//       the 4 weight gradients of an encoder layer (108 tiles) need a split of 2, not 7 per GEMM -> 3.5x fewer f32
//       atomics.  Tiles are copied untransposed ([64 k][256 m], pitch 544 B) and transposed by ds_read_b64_tr_b16.
#include "gemm8_common.h"

#if defined(G8_STAMPS)       // measurement builds only (tools/g8_stamps.sh): wall-clock stamps of every workgroup's tiles
__device__ unsigned long long g8_stamp_buf[256 * 8 * 8];
extern "C" int ss_gemm8_debug_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g8_stamp_buf), sizeof(g8_stamp_buf)) == hipSuccess ? 1 : 0; }
#define G8_STAMP(item, k) do { if (tid == 0 && blockIdx.x < 256 && (item) < 8) { g8_stamp_buf[(blockIdx.x * 8 + (item)) * 8 + (k)] = wall_clock64(); \
        if ((k) == 1 || (k) == 2) g8_stamp_buf[(blockIdx.x * 8 + (item)) * 8 + 4 + (k)] = clock64(); } } while (0)      /* slots 5, 6: shader cycles around the K loop */
#else
#define G8_STAMP(item, k) do { } while (0)
#endif

namespace g8 {

__device__ __forceinline__ int fsw(int row) { return (row & 7) ^ (((row >> 3) & 3) << 1); }

// ROWS tile rows copied as 8-row (1 KiB) pieces, piece p by wave p % 8
template <int ROWS>
struct Stage {
    static constexpr int PIECES = ROWS / 8, NP = (PIECES + 7) / 8;
    unsigned off[NP];                 // byte offset of this lane's source chunk (row clamped, chunk pre-swizzled), K offset excluded
    // When PIECES is not a multiple of 8 the waves beyond the last piece repeat their previous one (same bytes to the same LDS
    // address): a branch around one copy would split the K loop's basic block and cost the counted LDS waits more than 1 KiB does.
    // LEAN: one offset at a time (called from inside the K loop, where ~250 registers are live: interleaved, the temporaries of the NP row maps spill)
    template <bool LEAN = false>
    __device__ __forceinline__ void init(const RowMap& map, int outer0, int outer_size, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = i * 8 + wave; p = p < PIECES ? p : p - 8;
            const int r = p * 8 + (lane >> 3);
            int o = outer0 + r; o = o < outer_size ? o : outer_size - 1;
            const int chunk = (lane & 7) ^ fsw(r);
            off[i] = (unsigned)((rowmap_off(map, o) + chunk * 8) * 2);
            if constexpr (LEAN) sched_fence();
        }
    }
    __device__ __forceinline__ void issue(const unsigned char* base_k, unsigned char* tile, int wave) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int p = i * 8 + wave; p = p < PIECES ? p : p - 8;
            glds16(base_k + off[i], tile + p * 1024);
        }
    }
    template <int I>
    __device__ __forceinline__ void issue_one(const unsigned char* base_k, unsigned char* tile, int wave) {
        int p = I * 8 + wave; p = p < PIECES ? p : p - 8;
        glds16(base_k + off[I], tile + p * 1024);
    }
    // The same copy with the address split the way the hardware takes it: a wave-uniform 64-bit base in SGPRs (operand base + K
    // offset), this lane's 32-bit byte offset in ONE VGPR, and M0 = LDS address of the wave's slot + an immediate.  Left to the
    // compiler the copy carried a 64-bit VGPR address (a v_lshl_add_u64 per piece and 2 registers per offset) and one hoisted SGPR per
    // piece for M0 -- 9 live scalars that were spilled to VGPR lanes and came back through v_readlane inside the K loop.
    // slot = LDS byte address of (tile + wave * 1024); slot_last = the same for the last piece group (waves beyond it repeat their previous piece).
    template <int I>
    __device__ __forceinline__ void issue_one_s(const unsigned char* base_k, unsigned char* tile, int wave, unsigned slot, unsigned slot_last) {
#if defined(SS_EMU)
        (void)slot; (void)slot_last;
        issue_one<I>(base_k, tile, wave);
#else
        (void)tile; (void)wave;
        constexpr bool LASTG = (I + 1 == NP) && (PIECES % 8 != 0);
        if constexpr (LASTG) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(slot_last), "v"(off[I]), "s"(base_k) : "memory", "m0");
        else asm volatile("s_add_i32 m0, %0, %1\n\tglobal_load_lds_dwordx4 %2, %3" : : "s"(slot), "n"(I * 8192), "v"(off[I]), "s"(base_k) : "memory", "m0");
#endif
    }
    // slot_last relative to the tile: piece (NP - 1) * 8 + wave when it exists, else the wave's previous piece
    __device__ __forceinline__ static unsigned last_piece_off(int wave) { int p = (NP - 1) * 8 + wave; p = p < PIECES ? p : p - 8; return (unsigned)p * 1024u; }
};

__device__ __forceinline__ void tile_coord(int it, int G, int nitems, int tiles_n, int& mt, int& nt) {
    const int chunk0 = it / G * G, pos = it - chunk0;
    int R = nitems - chunk0; R = R > G ? G : R;
    const int xcd = pos & 7, q = R >> 3, r = R & 7;
    const int idx = chunk0 + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (pos >> 3);
    mt = idx / tiles_n; nt = idx - mt * tiles_n;
}

// wait for this wave's LDS reads AND its outstanding global->LDS copies, then the workgroup barrier
__device__ __forceinline__ void barrier_all() {
#if defined(SS_EMU)
    __syncthreads();
#else
    // the wait as a BUILTIN: the compiler's own wait insertion then knows that nothing it issued (register reloads, ...) is pending
    // behind this point, and places no s_waitcnt vmcnt(0) of its own inside the K loop -- where it would also drain the copies of the next
    // K tile (issued from asm, invisible to it) a few instructions after their issue
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

// The first barrier of an item whose predecessor left through direct_store8<INTERIOR>: this item's first K tile was requested BEFORE those
// NST stores (from the last K step of the predecessor), and vector-memory operations retire in issue order -- "at most NST outstanding"
// means the K tile has landed, while the stores drain under the first K step instead of in front of it (2.2 -> 0.5 us per tile).
template <int NST>
__device__ __forceinline__ void barrier_after_stores() {
#if defined(SS_EMU)
    __syncthreads();
#else
    static_assert(NST >= 0 && NST < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt(((NST >> 4) << 14) | (7 << 4) | (NST & 15));      // vmcnt(NST) lgkmcnt(0), expcnt untouched (gfx9 encoding)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

// ---- epilogue.  Everything the epilogue reads from memory is requested EARLY and all at once: the bias of the wave's four column groups
// before the tile's first pass, and the per-element side input of a flush pass (the gate = saved activation of the ReLU / dropout backward, or
// the old C of "C += v") one pass ahead, into registers the main loop no longer needs.  Left inside the per-chunk loop of the flush, each of
// the 18 chunks of a tile paid its own memory round trip while no MFMA ran (22 000 x 3072 x 768 alone: gate +42 us, bias +14 us, C += v +14 us
// over the plain 131 us; tools/gemm_bench epi).
template <class TO, int GEN>        // GEN: 0 alpha / bias / ReLU, 1 + dropout, 2 + log-clamp
__device__ __forceinline__ void stage8(const f32x4& a, TO* __restrict__ ct, int ldc, int lrow0, int lcol, const GemmEpi& epi, int row0, int col, int N, float bias)
{
    const float lo = epi.relu ? 0.f : -INFINITY;
    bool kp[4] = {true, true, true, true};
    if (GEN == 1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) kp[reg] = dropout_keep1(epi.seed, epi.stream, (unsigned long long)(row0 + reg) * (unsigned)N + col, epi.drop_thresh);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float x = fmaxf(a[reg] * epi.alpha + bias, lo);
        if (GEN == 1) x = kp[reg] ? x * epi.drop_scale : 0.f;
        if (GEN == 2) x = logf(fmaxf(x, epi.log_clamp));
        stf(ct + (lrow0 + reg) * ldc + lcol, x);
    }
}

// The same for the TRANSPOSED accumulator layout of the kernels without dropout (MFMA operands swapped): lane (r, q) holds row r and columns
// 4 q .. 4 q + 3 of the 16 x 16 tile, so the four values leave as ONE 8-byte (bf16) / 16-byte (f32) LDS store instead of four 2-byte ones
// (a store's cost is its address + data transfer: 4 cycles for a 2-byte ds_write, 6 for 8 bytes)
template <class TO, bool DROP = false>
__device__ __forceinline__ void stage8_sw(const f32x4& a, TO* __restrict__ ct, int ldc, int lrow, int lcol0, const GemmEpi& epi, const f32x4& bias, int row = 0, int col0 = 0, int N = 0)
{
    const float lo = epi.relu ? 0.f : -INFINITY;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = fmaxf(a[e] * epi.alpha + bias[e], lo);
    if constexpr (DROP) {               // the lane's four columns are ONE group of the row-major stream (N % 4 == 0, col0 % 4 == 0)
        bool kp[4];
        dropout_keep4(epi.seed, epi.stream, ((unsigned long long)row * (unsigned)N + col0) >> 2, epi.drop_thresh, kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = kp[e] ? x[e] * epi.drop_scale : 0.f;
    }
    if constexpr (sizeof(TO) == 2) { u32x2 w; w[0] = pack_bf16(x[0], x[1]); w[1] = pack_bf16(x[2], x[3]); *(u32x2*)(ct + lrow * ldc + lcol0) = w; }
    else { f32x4 w = {x[0], x[1], x[2], x[3]}; *(f32x4*)(ct + lrow * ldc + lcol0) = w; }
}

// ---- the plain epilogue of the transposed accumulator layout WITHOUT the LDS C piece (bf16 out; no side input, no column statistics).
// Lane (r, q) holds row r and columns 4 q .. 4 q + 3 of each of the wave's four 16-column tiles j: 8 bytes per tile, 32 bytes apart.  One
// v_permlane16_swap per dword exchanges, between the lanes q and q ^ 1 of a row, the piece of tile 2 jp + 1 (from q even) against the piece
// of tile 2 jp (from q odd): afterwards lane (r, q) holds columns 8 (q >> 1) .. + 7 of tile 2 jp + (q & 1), 16 contiguous bytes, and the 16
// rows of a store instruction each receive one 64-byte run.  The C piece cost 3 x (12 ds_write_b64, barrier, 6 ds_read_b128 + address
// arithmetic per thread, barrier) = 4.6 us per 288 x 256 tile with no MFMA running (tools/g8_stamps); this form has no barrier at all.
__device__ __forceinline__ void lane16_swap(unsigned& x, unsigned& y) {
#if defined(SS_EMU)
    const int lane = (int)(threadIdx.x & 63), q = lane >> 4;
    const unsigned xo = __shfl(y, (lane - 16) & 63), yo = __shfl(x, (lane + 16) & 63);      // every lane takes part in both exchanges
    if (q & 1) x = xo; else y = yo;
#else
    const auto v = __builtin_amdgcn_permlane16_swap(x, y, false, false);     // rows 1, 3 of x <-> rows 0, 2 of y (a row = 16 lanes)
    x = v[0]; y = v[1];
#endif
}
// INTERIOR: the tile lies inside the matrix -- no per-lane predicate, so every wave issues at least 2 NI vector-memory operations behind
// the next item's first K tile (the next item's first barrier counts on that number: barrier_after_stores).
// Side inputs (addressed like C, 16 bytes per lane and store): the gate (saved activation of a ReLU / dropout backward) and / or the old C of
// "C += v".  One of them travels two row tiles ahead in registers (the fragment registers of the K loop are free here); when both
// are present the old C is read in place.  Arithmetic and rounding are those of the C-piece epilogue: bf16(alpha acc + bias), then the gate
// and the sum on that rounded value in f32, rounded once more.
// STATS = 2: per-column sums of the stored (rounded) values: 16 accumulators per lane, folded over the 16 rows of a DPP row at the end,
// one atomic per column, wave and tile.
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// [x > 0] of the eight bf16 values of four packed words as one byte (bit 2 d = low half of word d, bit 2 d + 1 = high half): a bf16 is positive exactly when
// its 16 bits, read as a signed integer, are > 0 -- packed max(.,0) / min(.,1) give (hi > 0) << 16 | (lo > 0) per word
__device__ __forceinline__ unsigned sign_byte8(const unsigned (&v)[4]) {
    unsigned sacc = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#if defined(SS_EMU)
        const unsigned r = (unsigned)((short)(v[d] & 0xffffu) > 0) | ((unsigned)((short)(v[d] >> 16) > 0) << 16);
#else
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        const s16x2 z = {0, 0}; const u16x2 one = {1, 1};
        const s16x2 w = __builtin_bit_cast(s16x2, v[d]);
        const s16x2 c = __builtin_elementwise_max(w, z);
        const unsigned r = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, c), one));
#endif
        sacc |= r << (2 * d);
    }
    return (sacc & 0x55u) | ((sacc >> 15) & 0xAAu);
}
template <int NI, bool INTERIOR, int STATS, bool DROP = false>
__device__ __forceinline__ void direct_store8(const f32x4 (&acc)[NI][4], bf16_t* __restrict__ C, const GemmEpi& epi, int m0, int n0, int M, int N, int wm, int wn, int r, int q,
                                              const f32x4 (&b16)[4])
{
    constexpr int AHEAD = 2, SLOTS = AHEAD + 1;                // the side input of row tile i + AHEAD is requested when row tile i is worked on
    const float lo = epi.relu ? 0.f : -INFINITY;
    const float alpha = epi.alpha, gs = epi.gate_scale;
    const bf16_t* gate = (const bf16_t*)epi.gate;
    const bool acc_c = epi.mode == 1;
    const bf16_t* side = gate ? gate : (acc_c ? (const bf16_t*)C : nullptr);       // the prefetched input
    const int colb = n0 + wn * 64 + (q & 1) * 16 + (q >> 1) * 8;
    const bool cok0 = INTERIOR || colb < N, cok1 = INTERIOR || colb + 32 < N;
    float cs[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[jp][e] = 0.f;
    u32x4 pre[SLOTS][2];
    auto row_of = [&](int i) { return m0 + (wm * NI + i) * 16 + r; };
    auto fetch = [&](auto ic) {
        constexpr int i = ic;
        const int row = row_of(i);
        const long long ro = rowmap_off(epi.cmap, INTERIOR || row < M ? row : M - 1);
        const u32x4 z = {0u, 0u, 0u, 0u};
        pre[i % SLOTS][0] = (INTERIOR || (row < M && cok0)) ? *(const u32x4*)(side + ro + colb) : z;
        pre[i % SLOTS][1] = (INTERIOR || (row < M && cok1)) ? *(const u32x4*)(side + ro + colb + 32) : z;
    };
    if (side) static_for<0, (AHEAD < NI ? AHEAD : NI)>([&](auto ic) { fetch(ic); });
    static_for<0, NI>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (i + AHEAD < NI) { if (side) fetch(std::integral_constant<int, i + AHEAD>{}); }
        const int row = row_of(i);
        const bool rok = INTERIOR || row < M;
        const long long ro = rowmap_off(epi.cmap, rok ? row : M - 1);
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            unsigned w[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = fmaxf(acc[i][2 * jp + t][e] * alpha + b16[2 * jp + t][e], lo);
                if constexpr (DROP) {       // before the lane16 swap: columns 4 q .. 4 q + 3 of tile 2 jp + t = one group of the row-major stream
                    bool kp[4];
                    dropout_keep4(epi.seed, epi.stream, ((unsigned long long)row * (unsigned)N + (n0 + wn * 64 + (2 * jp + t) * 16 + 4 * q)) >> 2, epi.drop_thresh, kp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = kp[e] ? x[e] * epi.drop_scale : 0.f;
                }
                w[t][0] = pack_bf16(x[0], x[1]); w[t][1] = pack_bf16(x[2], x[3]);
            }
            lane16_swap(w[0][0], w[1][0]);
            lane16_swap(w[0][1], w[1][1]);
            unsigned v[4] = {w[0][0], w[0][1], w[1][0], w[1][1]};
            const int col = colb + jp * 32;
            const bool ok = INTERIOR || (rok && (jp ? cok1 : cok0));
            if (side) {
                u32x4 old = {0u, 0u, 0u, 0u};
                if (gate && acc_c && ok) old = *(const u32x4*)(C + ro + col);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float x0 = bf_lo(v[d]), x1 = bf_hi(v[d]);
                    const unsigned sd = pre[i % SLOTS][jp][d];
                    if (gate) {
                        x0 = bf_lo(sd) > 0.f ? x0 * gs : 0.f; x1 = bf_hi(sd) > 0.f ? x1 * gs : 0.f;
                        if (acc_c) { x0 += bf_lo(old[d]); x1 += bf_hi(old[d]); }
                    } else { x0 += bf_lo(sd); x1 += bf_hi(sd); }
                    v[d] = pack_bf16(x0, x1);
                }
            }
            if (ok) {
                const u32x4 o = {v[0], v[1], v[2], v[3]};
                *(u32x4*)(C + ro + col) = o;
                if (epi.sign_out) epi.sign_out[(long long)row * epi.sign_pitch + (col >> 3)] = (unsigned char)sign_byte8(v);      // the ReLU / dropout backward's gate, 1 bit per element
                if constexpr (STATS == 2) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) { cs[jp][2 * d] += bf_lo(v[d]); cs[jp][2 * d + 1] += bf_hi(v[d]); }
                }
            }
        }
        sched_fence();
    });
    if constexpr (STATS == 2) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = cs[jp][e];
                t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
                cs[jp][e] = t;
            }
        if (r == 0) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int col = colb + jp * 32 + e; if (INTERIOR || col < N) atomicAdd(epi.col_sum + col, cs[jp][e]); }
        }
    }
}

// chunk `it` of thread `tid` in flush pass `pass`: LDS row / column chunk, output row / column, validity
template <class TO, int NI, int IPP>
struct Flush8 {
    static constexpr int EV = OutVec<TO>::N, CPR = TBN / EV, R = IPP * 16, TOTAL = 2 * R * CPR, NIT = TOTAL / 512;
    static_assert(TOTAL % 1024 == 0 && 512 % CPR == 0, "flush chunks per thread (an even number)");
    __device__ __forceinline__ static bool map(int it, int tid, int pass, int m0, int n0, int M, int N, int& lr, int& ch, int& row, int& col) {
        const int idx = it * 512 + tid;
        lr = idx / CPR; ch = idx - lr * CPR;
        const int half = lr >= R ? 1 : 0, rr = lr - half * R, trow = pass * R + rr;         // row inside the half
        row = m0 + half * NI * 16 + trow; col = n0 + ch * EV;
        return trow < NI * 16 && row < M && col < N;
    }
};
__device__ __forceinline__ void raw_to_float(const u32x4& w, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ void raw_to_float(const u32x4& w, float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(w[e]);
}
// side input (src addressed like C: the gate, or C itself) of HALF a flush pass -> registers: chunks H * NIT/2 .. of pass `pass`, all in flight at once
template <class TO, int NI, int IPP, int H>
__device__ __forceinline__ void flush8_prefetch(const TO* __restrict__ src, const GemmEpi& epi, int m0, int n0, int pass, int M, int N, int tid, u32x4 (&pre)[Flush8<TO, NI, IPP>::NIT / 2])
{
    using F = Flush8<TO, NI, IPP>;
#pragma unroll
    for (int i = 0; i < F::NIT / 2; ++i) {
        int lr, ch, row, col;
        const u32x4 z = {0u, 0u, 0u, 0u};
        pre[i] = F::map(H * (F::NIT / 2) + i, tid, pass, m0, n0, M, N, lr, ch, row, col) ? *(const u32x4*)(src + rowmap_off(epi.cmap, row) + col) : z;
    }
}
// flush of HALF an epilogue pass (the LDS piece holds IPP*16 rows of each of the two M halves of the tile).
// A thread always handles the same 16-byte column chunk (512 % CPR == 0), so the optional column statistics accumulate in registers
// (cs / cq) over all rows the thread stores, across the passes of a tile.  pf: 1 = `pre` holds the gate chunks, 2 = the old C chunks.
template <class TO, int NI, int IPP, int STATS, int H>
__device__ __forceinline__ void flush8(const TO* __restrict__ ct, int ldc, TO* __restrict__ C, const GemmEpi& epi, int m0, int n0, int pass, int M, int N, int tid,
                                       float (&cs)[OutVec<TO>::N], float (&cq)[OutVec<TO>::N], const float (&sh)[OutVec<TO>::N],
                                       const u32x4 (&pre)[Flush8<TO, NI, IPP>::NIT / 2], int pf, const unsigned* gb = nullptr)
{
    using F = Flush8<TO, NI, IPP>;
    constexpr int EV = F::EV;
#pragma unroll
    for (int i = 0; i < F::NIT / 2; ++i) {
        int lr, ch, row, col;
        if (F::map(H * (F::NIT / 2) + i, tid, pass, m0, n0, M, N, lr, ch, row, col)) {
            const long long off = rowmap_off(epi.cmap, row) + col;
            if (STATS == 0 && !epi.gate && !epi.gate_bits && epi.mode != 1 && !(sizeof(TO) == 4 && epi.planes_hi)) {
                // nothing is applied per element here (bias / ReLU / dropout went in before the LDS piece): the 16 bytes leave as they are --
                // unpacking 8 bf16 to f32 and rounding them back was ~60 of this chunk's instructions, on every plain tile of the step
                *(u32x4*)(C + off) = *(const u32x4*)(ct + lr * ldc + ch * EV);
                sched_fence();
                continue;
            }
            float v[EV];
            outvec_load(ct + lr * ldc + ch * EV, v);
            if (epi.gate) {
                float g[EV];
                if (pf == 1) raw_to_float(pre[i], g); else outvec_load((const TO*)epi.gate + off, g);
#pragma unroll
                for (int e = 0; e < EV; ++e) v[e] = g[e] > 0.f ? v[e] * epi.gate_scale : 0.f;
            } else if (epi.gate_bits) {         // the same gate from one bit per element; the tile's bytes were requested together before the first pass
                const unsigned bits = gb[pass * F::NIT + H * (F::NIT / 2) + i] >> (col & 7);
#pragma unroll
                for (int e = 0; e < EV; ++e) v[e] = ((bits >> e) & 1u) ? v[e] * epi.gate_scale : 0.f;
            }
            if (epi.mode == 1) {
                float o[EV];
                if (pf == 2) raw_to_float(pre[i], o); else outvec_load(C + off, o);
#pragma unroll
                for (int e = 0; e < EV; ++e) v[e] += o[e];
            }
            if constexpr (sizeof(TO) == 4) {
                if (epi.planes_hi) {            // the value that is stored, as hi / lo bf16 planes addressed like C (planes.hip: split_planes_kernel's arithmetic)
                    const unsigned h0 = pack_bf16(v[0], v[1]), h1 = pack_bf16(v[2], v[3]);
                    const u32x2 hw = {h0, h1};
                    const u32x2 lw = {pack_bf16(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u)),
                                      pack_bf16(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u))};
                    *(u32x2*)((bf16_t*)epi.planes_hi + off) = hw; *(u32x2*)((bf16_t*)epi.planes_lo + off) = lw;
                }
                if (!epi.planes_only) outvec_store(C + off, v);
            } else outvec_store(C + off, v);
            if (STATS == 1) {
#pragma unroll
                for (int e = 0; e < EV; ++e) { const float x = rnd<TO>(v[e]) - sh[e]; cs[e] += x; cq[e] += x * x; }
            } else if (STATS == 2) {
#pragma unroll
                for (int e = 0; e < EV; ++e) cs[e] += rnd<TO>(v[e]);
            }
        }
        sched_fence();          // one chunk at a time: interleaved, the chunks' temporaries push the epilogue past the register file (accumulators + side inputs are live)
    }
}

// The side input travels half a pass ahead of its use, and is always requested BEFORE the stores of the half in front of it (a wait for a
// load that was issued behind a store also waits for that store: vmcnt counts in order): pa = first halves, pb = second halves.
template <class TO, int GEN, int NI, int IPP, int P, int NPASS, int STATS, bool SW = false>
struct Passes8 {
    static __device__ __forceinline__ void run(const f32x4 (&acc)[NI][4], TO* ct, int ldc, TO* C, const GemmEpi& epi, int m0, int n0, int M, int N, int tid, int wm, int wn, int r, int q,
                                               float (&cs)[OutVec<TO>::N], float (&cq)[OutVec<TO>::N], const float (&sh)[OutVec<TO>::N],
                                               const float (&bias4)[4], u32x4 (&pa)[Flush8<TO, NI, IPP>::NIT / 2], u32x4 (&pb)[Flush8<TO, NI, IPP>::NIT / 2], int pf, const TO* pf_src,
                                               const f32x4 (&b16)[4], const unsigned* gb = nullptr) {
#pragma unroll
        for (int ii = 0; ii < IPP; ++ii) {
            constexpr int I0 = P * IPP;
            if (I0 + ii < NI) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SW) stage8_sw<TO, GEN == 1>(acc[I0 + ii < NI ? I0 + ii : 0][j], ct, ldc, (wm * IPP + ii) * 16 + r, wn * 64 + j * 16 + q * 4, epi, b16[j],
                                                             m0 + (wm * NI + I0 + ii) * 16 + r, n0 + wn * 64 + j * 16 + q * 4, N);
                    else stage8<TO, GEN>(acc[I0 + ii < NI ? I0 + ii : 0][j], ct, ldc, (wm * IPP + ii) * 16 + q * 4, wn * 64 + j * 16 + r, epi,
                                         m0 + (wm * NI + I0 + ii) * 16 + q * 4, n0 + wn * 64 + j * 16 + r, N, bias4[j]);
                }
            }
        }
        barrier_keep_vm();
        if (pf) flush8_prefetch<TO, NI, IPP, 1>(pf_src, epi, m0, n0, P, M, N, tid, pb);
        flush8<TO, NI, IPP, STATS, 0>(ct, ldc, C, epi, m0, n0, P, M, N, tid, cs, cq, sh, pa, pf, gb);
        if (P + 1 < NPASS && pf) flush8_prefetch<TO, NI, IPP, 0>(pf_src, epi, m0, n0, P + 1, M, N, tid, pa);
        flush8<TO, NI, IPP, STATS, 1>(ct, ldc, C, epi, m0, n0, P, M, N, tid, cs, cq, sh, pb, pf, gb);
        if (P + 1 < NPASS) barrier_keep_vm();
        Passes8<TO, GEN, NI, IPP, P + 1, NPASS, STATS, SW>::run(acc, ct, ldc, C, epi, m0, n0, M, N, tid, wm, wn, r, q, cs, cq, sh, bias4, pa, pb, pf, pf_src, b16, gb);
    }
};
template <class TO, int GEN, int NI, int IPP, int NPASS, int STATS, bool SW>
struct Passes8<TO, GEN, NI, IPP, NPASS, NPASS, STATS, SW> {
    static __device__ __forceinline__ void run(const f32x4 (&)[NI][4], TO*, int, TO*, const GemmEpi&, int, int, int, int, int, int, int, int, int,
                                               float (&)[OutVec<TO>::N], float (&)[OutVec<TO>::N], const float (&)[OutVec<TO>::N],
                                               const float (&)[4], u32x4 (&)[Flush8<TO, NI, IPP>::NIT / 2], u32x4 (&)[Flush8<TO, NI, IPP>::NIT / 2], int, const TO*,
                                               const f32x4 (&)[4], const unsigned* = nullptr) {}
};

}  // namespace g8

// ================================================================ KC x KC
// PIN: bit 0 = fragment reads, bit 1 = DMA pieces spread between the MFMA groups of a phase (else issued in a burst at the phase start).
// STATS: 1 = the epilogue also accumulates per-column sums / sums of squares (about a shift) of the stored tile, 2 = plain column sums only
// (a bias gradient: a third of the registers, which leaves room for the side-input buffers of a gated epilogue) (separate instantiations: the extra live
// registers of that path would otherwise spill in the main loop of every launch).
// ABL: compile-time ablation mask for tuning (results are wrong): 1 no MFMA, 2 no in-loop global->LDS copies, 4 no in-loop fragment reads.
// PL ("planes", the parity-grade f32 x3 mode on this kernel): A and B are the HI planes of f32 operands split into hi = bf16(x), lo = bf16(x - hi);
// the LO planes sit a_lo / b_lo bytes behind them with the same row maps.  The K loop then runs over a THREE times longer contraction
//     C = [A_lo | A_hi | A_hi] . [B_hi | B_lo | B_hi]^T        (small terms first; the a_lo.b_lo term, 2^-18 of the product, is dropped)
// -- the same arithmetic as TileMmaX3 of gemm.hip (three bf16 MFMAs per product, f32 accumulate), but on bf16 LDS tiles and this kernel's
// schedule: a K tile index t of the ring maps to (segment t / (K/64), K offset t % (K/64)), i.e. only the scalar base of a copy changes.
template <class TO, int NI, int PIN, int ABL, int STATS, int GENSEL, bool PL = false>
__global__ __launch_bounds__(512) void gemm8_kc_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                       int M, int N, int K, RowMap amap, RowMap bmap, GemmEpi epi, int tiles_n, int nitems, long long a_lo, long long b_lo)
{
    using namespace g8;
    constexpr bool SWAP = GENSEL == 0 || GENSEL == 1;        // one epilogue path known at compile time (none / dropout): transposed accumulator tiles (see stage8_sw);
                                                             // the dropout draws follow the row-major element index, so a lane's four columns are one group (common.h)
    constexpr int BMT = 2 * NI * 16, STAGE = (BMT + TBN) * RB;
    constexpr int HRSEL = PIN >> 2;                         // PIN bits 2..3: rows of 16 per phase -- 0: 3 or 4, 1: one, 2: two (even NI)
    constexpr int HR = HRSEL == 1 ? 1 : (HRSEL == 2 && NI % 2 == 0 ? 2 : (NI % 3 == 0 ? 3 : 4));                 // 16-row MFMA tiles per phase
    constexpr int NPK = NI / HR, NPH = 2 * NPK;             // phases per K half / per K step (even: the ping-pong parity carries over)
    static_assert(NI % HR == 0 && NPH % 2 == 0, "phase split");
    SS_DYN_SMEM(lds_raw);
    unsigned char* lds = (unsigned char*)lds_raw;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = wave_uniform(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x, n1 = K / BK8, nsteps = PL ? 3 * n1 : n1;
    // byte offsets (from the operand bases) of K tile t of the ring; tile 0 of an item: (ka0, 0)
    const long long ka0 = PL ? a_lo : 0;
    auto koff = [&](int t, long long& ka, long long& kbb) {
        if constexpr (PL) {
            const int seg = (t >= n1 ? 1 : 0) + (t >= 2 * n1 ? 1 : 0), kk = t - seg * n1;
            ka = (long long)kk * (BK8 * 2) + (seg == 0 ? a_lo : 0); kbb = (long long)kk * (BK8 * 2) + (seg == 1 ? b_lo : 0);
        } else { ka = (long long)t * (BK8 * 2); kbb = ka; }
    };
    Stage<BMT> sa; Stage<TBN> sb;
    // per-lane fragment read offsets: entry x serves (kk ^ i) & 1 == x (A) / (kk ^ j) & 1 == x (B)
    int aoff[2], boff[2];
    {
        const int f0 = (r & 7) ^ (((r >> 3) & 1) << 1), cq = q ^ f0, par = (wm * NI) & 1;
        aoff[0] = (wm * NI * 16 + r) * RB + ((cq ^ (par << 2)) << 4);
        aoff[1] = (wm * NI * 16 + r) * RB + ((cq ^ ((par ^ 1) << 2)) << 4);
        boff[0] = BMT * RB + (wn * 64 + r) * RB + (cq << 4);
        boff[1] = BMT * RB + (wn * 64 + r) * RB + ((cq ^ 4) << 4);
    }
    const unsigned lbase = lds_byte_address(lds);                                  // LDS byte address of the dynamic segment
    const unsigned wslot = (unsigned)wave * 1024u;                                 // this wave's 1 KiB slot inside an 8-piece group of a stage
    int it = blockIdx.x, mt, nt, cur = 0;
    tile_coord(it, G, nitems, tiles_n, mt, nt);
    int m0 = mt * BMT, n0 = nt * TBN;
    sa.init(amap, m0, M, wave, lane); sb.init(bmap, n0, N, wave, lane);
    sa.issue((const unsigned char*)A + ka0, lds, wave); sb.issue((const unsigned char*)B, lds + BMT * RB, wave);

    constexpr int SA_NP = Stage<BMT>::NP, NPW = SA_NP + Stage<TBN>::NP;     // global->LDS pieces per wave and K tile
    constexpr bool SPR = (PIN & 1) != 0, SPD = (PIN & 2) != 0;              // spread the fragment reads / the DMA pieces between the MFMA groups
    constexpr int NCH = SPD ? NPH / 2 : 1, CS = (NPW + NCH - 1) / NCH;     // the copy of a K tile is issued in NCH chunks of CS pieces

    int item = 0;
    bool stores_behind_tile0 = false;
    for (;; ++item) {
        G8_STAMP(item, 0);
        // The next item's first K tile is one more tile of the same ring: with >= 2 K steps its copies leave from the slots in which a tile
        // nsteps of THIS item would be requested (last phase of step nsteps-2, early phases of step nsteps-1), under the MFMAs of the last
        // K step.  (Requested in one burst after the loop, the 9 copies per wave cost 2.1 us per tile: tools/g8_stamps, "drain".)
        const int cm0 = m0, cn0 = n0;
        const bool has_next = it + G < nitems;
        const bool ring_next = has_next && nsteps >= 2 && !(ABL & 2);
        f32x4 acc[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
        bf16x8 fa[2][HR], fb[2][4];
        // A fragment x / B fragment j of phase ph (K half ph / NPK, rows (ph % NPK) * HR ..) from the stage at byte offset st
        auto read_a = [&](auto phc, auto xc, unsigned st) {
            constexpr int ph = phc, x = xc, kk = ph / NPK, i = (ph % NPK) * HR + x;
            lds_read128_async<i * 2048>(fa[ph & 1][x], lds, lbase + st + aoff[(i ^ kk) & 1]);
        };
        auto read_b = [&](auto phc, auto jc, unsigned st) {
            constexpr int ph = phc, j = jc, kk = ph / NPK;
            lds_read128_async<j * 2048>(fb[kk][j], lds, lbase + st + boff[(j ^ kk) & 1]);
        };
        auto wait_frags = [&](auto phc) {       // issued from the END of the phase in which the fragments of phase ph were requested
            constexpr int ph = phc;
            lds_wait_pin(fa[ph & 1]);
            if constexpr (ph % NPK == 0) lds_wait_pin(fb[ph / NPK]);
        };
        auto dma = [&](auto kc, long long kbyte, long long kbyte_b, unsigned st) {      // piece k of the K tile at byte offsets kbyte (A) / kbyte_b (B) -> stage at st
            constexpr int k = kc;
            if constexpr (k < SA_NP) sa.template issue_one_s<k>((const unsigned char*)A + kbyte, lds + st, wave, lbase + st + wslot, lbase + st + Stage<BMT>::last_piece_off(wave));
            else if constexpr (k < NPW) sb.template issue_one_s<k - SA_NP>((const unsigned char*)B + kbyte_b, lds + st + BMT * RB, wave, lbase + st + BMT * RB + wslot,
                                                                            lbase + st + BMT * RB + Stage<TBN>::last_piece_off(wave));
        };
        auto mfma_row = [&](auto phc, auto xc) {
            constexpr int ph = phc, x = xc, kk = ph / NPK, i = (ph % NPK) * HR + x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SWAP) acc[i][j] = mfma_bf16_16x16x32(fb[kk][j], fa[ph & 1][x], acc[i][j]);      // D^T: lane (r, q) <- row r, columns 4 q .. 4 q + 3
                else acc[i][j] = mfma_bf16_16x16x32(fa[ph & 1][x], fb[kk][j], acc[i][j]);
            }
        };
        // One phase = HR groups of {a share of the NEXT phase's fragment reads, a share of a DMA chunk, 4 MFMAs}, then ONE wait for
        // the fragments requested in it.  PIN = 0: reads and DMA in a burst at the phase start (every wave queues behind the LDS /
        // TA pipes before its first MFMA: measured +0.35 us per K tile each); PIN = 1: spread between the MFMA groups.
        //   DMA chunks of K tile t: chunk 0 right after the mid-step barrier of step t-2 (last phase; for tile 1: right after the barrier
        //   in front of step 0), chunks 1.. in phases 0.. of step t-1, all of them >= 3 phases ahead of the barrier (step t-1, last phase)
        //   that waits for them.
        // STEADY (compile time): K step s with s + 2 < nsteps -- every condition below is true, the step is ONE basic block
        // up to its barrier (with the conditions evaluated at run time each MFMA group of a spread phase became its own block, and the
        // compiler re-materialised addresses and spilled around the branches)
        auto phase = [&](auto phc, auto steady_c, const unsigned S, const unsigned O, const int s) {
            constexpr int ph = phc, nx = ph + 1 < NPH ? ph + 1 : 0;
            constexpr bool last = ph + 1 == NPH, STEADY = steady_c;
            constexpr int chunk = last ? 0 : ph + 1;
            bool rd = true, dm = false; unsigned rst = S, dst = 0; long long kb = 0, kb2 = 0;
            if constexpr (!STEADY && ph == NCH - 1) {
                // this item's last copy has left (chunk NCH-1 of K tile nsteps-1, one phase ago): the offsets become the next item's
                if (ring_next && s + 2 == nsteps) {
                    it += G;
                    tile_coord(it, G, nitems, tiles_n, mt, nt);
                    m0 = mt * BMT; n0 = nt * TBN;
                    sched_fence();
                    sa.template init<true>(amap, m0, M, wave, lane); sb.template init<true>(bmap, n0, N, wave, lane);
                }
            }
            if constexpr (last) {
                if (STEADY || s + 1 < nsteps) {
                    barrier_all();               // every wave holds its last fragments of stage S; K tile s+1 has landed in O
                    rst = O; dst = S;
                    if constexpr (STEADY) { dm = !(ABL & 2); koff(s + 2, kb, kb2); }
                    else { const bool mine = s + 2 < nsteps; dm = (mine && !(ABL & 2)) || ring_next; if (mine) koff(s + 2, kb, kb2); else { kb = ka0; kb2 = 0; } }      // s + 2 == nsteps: K tile 0 of the next item
                } else rd = false;
            } else if constexpr (chunk < NCH) {
                dst = O;
                if constexpr (STEADY) { dm = !(ABL & 2); koff(s + 1, kb, kb2); }
                else { const bool mine = s + 1 < nsteps; dm = (mine && !(ABL & 2)) || ring_next; if (mine) koff(s + 1, kb, kb2); else { kb = ka0; kb2 = 0; } }
            }
            if (ABL & 4) rd = false;
            static_for<0, HR>([&](auto xc) {
                constexpr int x = xc;
                constexpr int x0 = SPR ? x : 0, x1 = SPR ? x + 1 : (x == 0 ? HR : 0);           // read shares issued in front of MFMA group x
                constexpr int y0 = SPD ? x : 0, y1 = SPD ? x + 1 : (x == 0 ? HR : 0);           // DMA shares
                if (rd) {
                    static_for<x0, x1>([&](auto xr) { read_a(std::integral_constant<int, nx>{}, xr, rst); });
                    if constexpr (nx % NPK == 0) static_for<x0 * 4 / HR, x1 * 4 / HR>([&](auto j) { read_b(std::integral_constant<int, nx>{}, j, rst); });
                }
                if constexpr (chunk < NCH) {
                    if (dm) static_for<chunk * CS + y0 * CS / HR, chunk * CS + y1 * CS / HR>([&](auto k) { dma(k, kb, kb2, dst); });
                }
                sched_fence();
                if (!(ABL & 1)) mfma_row(phc, xc);
                sched_fence();
            });
            if (rd) wait_frags(std::integral_constant<int, nx>{});
            sched_fence();
        };
        if (stores_behind_tile0) barrier_after_stores<2 * NI>(); else barrier_all();       // K tile 0 of this item has landed in stage `cur`; stage cur^1 is free
        stores_behind_tile0 = false;
        G8_STAMP(item, 1);
        // chunk 0 of K tile 1 (its other chunks follow from the phases of step 0, like those of every later tile)
        if (nsteps > 1 && !(ABL & 2)) { long long k1a, k1b; koff(1, k1a, k1b); static_for<0, (NCH > 1 ? CS : NPW)>([&](auto k) { dma(k, k1a, k1b, (unsigned)((cur ^ 1) * STAGE)); }); }
        static_for<0, HR>([&](auto x) { read_a(std::integral_constant<int, 0>{}, x, (unsigned)(cur * STAGE)); });
        static_for<0, 4>([&](auto j) { read_b(std::integral_constant<int, 0>{}, j, (unsigned)(cur * STAGE)); });
        wait_frags(std::integral_constant<int, 0>{});
        sched_fence();
        {   // steady steps 0 .. nsteps - 3 (straight-line body, every condition of a phase true at compile time) | the last two steps
            int s = 0;
            for (; s + 2 < nsteps; ++s) { const unsigned S = cur * STAGE, O = (cur ^ 1) * STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::true_type{}, S, O, s); }); cur ^= 1; }
            for (; s < nsteps; ++s) { const unsigned S = cur * STAGE, O = (cur ^ 1) * STAGE; static_for<0, NPH>([&](auto ph) { phase(ph, std::false_type{}, S, O, s); }); cur ^= 1; }
        }
        // `cur` names the stage the last K tile did NOT use (free since the previous mid-step barrier): the next item's first
        // K tile goes there while this item's C tile leaves through the other stage
        G8_STAMP(item, 2);
        // The epilogue's per-thread constants (flush rows, LDS offsets: ~20 values derived from the thread number) must not be hoisted out of
        // the item loop: live across the K loop they are spilled, and every reload inside the epilogue is a scratch load behind the tile's
        // global stores (vmcnt counts in order: s_waitcnt vmcnt(0) per reload = the store pipe drained ~30 times per tile, epilogue 4.6 -> 8.5 us)
        int tid_e = tid;
        pin_vgpr(tid_e);
        const int lane_e = tid_e & 63, r_e = lane_e & 15, q_e = lane_e >> 4;
        if (has_next && !ring_next) {
            it += G;
            tile_coord(it, G, nitems, tiles_n, mt, nt);
            m0 = mt * BMT; n0 = nt * TBN;
            sa.init(amap, m0, M, wave, lane); sb.init(bmap, n0, N, wave, lane);
            sa.issue((const unsigned char*)A + ka0, lds + cur * STAGE, wave); sb.issue((const unsigned char*)B, lds + cur * STAGE + BMT * RB, wave);
        }
        // side inputs of the epilogue (after the next item's copies: a wait for these must not have to drain younger copies first)
        constexpr int IPP = sizeof(TO) == 2 ? (NI % 3 == 0 ? 3 : 2) : 1;
        constexpr int NPASS = (NI + IPP - 1) / IPP;
        float bias4[4];
        f32x4 b16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SWAP) { const int col = cn0 + wn * 64 + j * 16 + q_e * 4; bias4[j] = 0.f; b16[j] = (epi.bias && col < N) ? *(const f32x4*)(epi.bias + col) : z; }      // N % 8 == 0 (epi.fast)
            else { const int col = cn0 + wn * 64 + j * 16 + r_e; bias4[j] = (epi.bias && col < N) ? epi.bias[col] : 0.f; b16[j] = z; }
        }
        // (not in the full column-statistics instantiation: its three per-column register arrays leave no room for the side-input buffers --
        // with them the gate + column-sum epilogue of the FFN input gradient spilled and ran 219 instead of 185 us; the sums-only form has room)
        const int pf = STATS == 1 ? 0 : (epi.gate ? 1 : (epi.mode == 1 ? 2 : 0));
        const TO* pf_src = pf == 1 ? (const TO*)epi.gate : C;
        // bf16 results of the transposed accumulator layout leave straight from the registers (direct_store8: alpha / bias / ReLU, gate, C += v,
        // column sums, and -- round 6 -- the dropout of linear1's epilogue); the C piece in LDS remains for the BatchNorm statistics and f32 output
        constexpr bool DIRECT = SWAP && sizeof(TO) == 2 && STATS == 0;
        if constexpr (DIRECT) {
            G8_STAMP(item, 3);
            if (cm0 + BMT <= M && cn0 + TBN <= N) {
                direct_store8<NI, true, STATS, GENSEL == 1>(acc, (bf16_t*)C, epi, cm0, cn0, M, N, wm, wn, r_e, q_e, b16);
                stores_behind_tile0 = ring_next;
            } else direct_store8<NI, false, STATS, GENSEL == 1>(acc, (bf16_t*)C, epi, cm0, cn0, M, N, wm, wn, r_e, q_e, b16);
        } else {
            // ("some value", fixed per item: left plainly undefined on the paths without a side input, the 24 registers became a value carried
            // around the item loop -- spilled before the K loop and reloaded, behind the global stores, in the epilogue; zeroed, they are live
            // registers of every epilogue)
            u32x4 pa[Flush8<TO, NI, IPP>::NIT / 2], pb[Flush8<TO, NI, IPP>::NIT / 2];
#if !defined(SS_EMU)
#pragma unroll
            for (int i = 0; i < Flush8<TO, NI, IPP>::NIT / 2; ++i) { pa[i] = __builtin_nondeterministic_value(pa[i]); pb[i] = __builtin_nondeterministic_value(pb[i]); }
#else
            for (int i = 0; i < Flush8<TO, NI, IPP>::NIT / 2; ++i) { const u32x4 z = {0u, 0u, 0u, 0u}; pa[i] = z; pb[i] = z; }
#endif
            if (pf) flush8_prefetch<TO, NI, IPP, 0>(pf_src, epi, cm0, cn0, 0, M, N, tid_e, pa);
            // gate as sign bits (column-sum instantiations: the FFN input gradient): ALL bytes of this thread's chunks of the tile are requested here, at
            // once -- one exposed round trip per tile where the tensor form pays one per half pass (6 per tile: +15 us per 288 x 256 tile, tools/gemm_bench epi)
            unsigned gbits[STATS == 2 ? NPASS * Flush8<TO, NI, IPP>::NIT : 1];
            if constexpr (STATS == 2) {
#pragma unroll
                for (int i = 0; i < NPASS * Flush8<TO, NI, IPP>::NIT; ++i) gbits[i] = 0u;
                if (epi.gate_bits) {
#pragma unroll
                    for (int pp = 0; pp < NPASS; ++pp)
#pragma unroll
                        for (int it = 0; it < Flush8<TO, NI, IPP>::NIT; ++it) {
                            int lr, ch, row, col;
                            if (Flush8<TO, NI, IPP>::map(it, tid_e, pp, cm0, cn0, M, N, lr, ch, row, col)) gbits[pp * Flush8<TO, NI, IPP>::NIT + it] = epi.gate_bits[(long long)row * epi.gate_bits_pitch + (col >> 3)];
                        }
                }
            }
            barrier_keep_vm();                                   // every wave is done reading stage cur^1 -> it becomes the C piece
            G8_STAMP(item, 3);
            TO* ct = (TO*)(lds + (cur ^ 1) * STAGE);
            constexpr int LDC = TBN + 16 / (int)sizeof(TO);
            static_assert((size_t)2 * IPP * 16 * LDC * sizeof(TO) <= (size_t)STAGE, "C piece does not fit the free stage");
            constexpr int EV = OutVec<TO>::N, CPR = TBN / EV;
            float cs[EV], cq[EV], sh[EV];
#pragma unroll
            for (int e = 0; e < EV; ++e) { cs[e] = 0.f; cq[e] = 0.f; sh[e] = 0.f; }
            if constexpr (STATS) {
                // column statistics of the stored tile: registers (per thread: one 16-byte column chunk, all its rows) -> per-wave LDS bins
                // (plain stores: every wave covers all 256 columns) -> 512 threads add the 8 waves' bins -> one global atomic per column and
                // tile.  The bins sit behind the C piece in the free stage.
                constexpr int BINS_OFF = NI % 3 == 0 ? 51200 : 40960;
                float* bins = (float*)(lds + (cur ^ 1) * STAGE + BINS_OFF);
                static_assert((size_t)2 * IPP * 16 * LDC * sizeof(TO) <= BINS_OFF && BINS_OFF + 8 * 2 * TBN * 4 <= STAGE, "column-statistics bins overlap the C piece");
                const int ch = tid_e % CPR;
                if (STATS == 1 && epi.col_shift) {
#pragma unroll
                    for (int e = 0; e < EV; ++e) { const int col = cn0 + ch * EV + e; sh[e] = col < N ? epi.col_shift[col] : 0.f; }
                }
                if (GENSEL == 1 || (GENSEL < 0 && epi.general == 1)) Passes8<TO, 1, NI, IPP, 0, NPASS, STATS, SWAP>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid_e, wm, wn, r_e, q_e, cs, cq, sh, bias4, pa, pb, pf, pf_src, b16, gbits);
                else Passes8<TO, 0, NI, IPP, 0, NPASS, STATS, SWAP>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid_e, wm, wn, r_e, q_e, cs, cq, sh, bias4, pa, pb, pf, pf_src, b16, gbits);
                if (CPR == 32) {                  // bf16 out: lanes l and l + 32 of a wave hold the same column chunk
#pragma unroll
                    for (int e = 0; e < EV; ++e) { cs[e] += __shfl_xor(cs[e], 32); if (STATS == 1) cq[e] += __shfl_xor(cq[e], 32); }
                }
                if (CPR == 64 || lane_e < 32) {
                    float* wb = bins + wave * 2 * TBN;
#pragma unroll
                    for (int e = 0; e < EV; ++e) { wb[ch * EV + e] = cs[e]; if (STATS == 1) wb[TBN + ch * EV + e] = cq[e]; }
                }
                barrier_keep_vm();
                {
                    float t = 0.f;
                    if (STATS == 1 || tid_e < TBN) {
#pragma unroll
                        for (int w8 = 0; w8 < 8; ++w8) t += bins[w8 * 2 * TBN + tid_e];
                    }
                    const int col = cn0 + (tid_e & (TBN - 1));
                    if (col < N) {
                        if (tid_e < TBN) atomicAdd(epi.col_sum + col, t);
                        else if (STATS == 1 && epi.col_sumsq) atomicAdd(epi.col_sumsq + col, t);
                    }
                }
            } else {
                if (GENSEL == 1 || (GENSEL < 0 && epi.general == 1)) Passes8<TO, 1, NI, IPP, 0, NPASS, 0, SWAP>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid_e, wm, wn, r_e, q_e, cs, cq, sh, bias4, pa, pb, pf, pf_src, b16);
                else Passes8<TO, 0, NI, IPP, 0, NPASS, 0, SWAP>::run(acc, ct, LDC, C, epi, cm0, cn0, M, N, tid_e, wm, wn, r_e, q_e, cs, cq, sh, bias4, pa, pb, pf, pf_src, b16);
            }
        }
        G8_STAMP(item, 4);
        if (!has_next) break;
    }
}

// ---------------------------------------------------------------- host side
// which == NI (8 or 9).  Legality (bf16 in, K % 64 == 0, 32-bit operand offsets, no transposed second output) is the caller's job.
// planes: A / B are the hi planes of split f32 operands, the lo planes a_lo / b_lo BYTES behind them (gemm8_kc_kernel<..., PL = true>; f32 out, default schedule only)
template <class TO>
int gemm8_launch_kc(int ni, int pin, const void* A, const void* B, void* C, int M, int N, int K, const RowMap& am, const RowMap& bm, const GemmEpi& epi, void* stream,
                    bool planes, long long a_lo, long long b_lo)
{
    const int bmt = 2 * ni * 16, tiles_m = (M + bmt - 1) / bmt, tiles_n = (N + 255) / 256, nitems = tiles_m * tiles_n;
    const size_t smem = (size_t)2 * (bmt + 256) * 128;
    const int cus = g8_cus();
    dim3 grid(nitems < cus ? nitems : cus), block(512);
#define G8_CASE(NI_, PIN_, ABL_, ST_) G8_CASE6(NI_, PIN_, ABL_, ST_, -1)
#define G8_CASE6(NI_, PIN_, ABL_, ST_, GS_)                                                                                        \
    do {                                                                                                                      \
        static bool granted = false;                                                                                          \
        if (!granted) { if (g8_grant((const void*)gemm8_kc_kernel<TO, NI_, PIN_, ABL_, ST_, GS_>, smem)) return 1; granted = true; } \
        SS_LAUNCH(SS_KERNEL(gemm8_kc_kernel<TO, NI_, PIN_, ABL_, ST_, GS_>), grid, block, smem, stream, (const bf16_t*)A, (const bf16_t*)B, (TO*)C, M, N, K, am, bm, epi, tiles_n, nitems, 0LL, 0LL); \
    } while (0)
#define G8_PLANES(NI_, ST_, GS_)                                                                                               \
    do {                                                                                                                      \
        static bool granted = false;                                                                                          \
        if (!granted) { if (g8_grant((const void*)gemm8_kc_kernel<float, NI_, 3, 0, ST_, GS_, true>, smem)) return 1; granted = true; } \
        SS_LAUNCH(SS_KERNEL(gemm8_kc_kernel<float, NI_, 3, 0, ST_, GS_, true>), grid, block, smem, stream, (const bf16_t*)A, (const bf16_t*)B, (float*)C, M, N, K, am, bm, epi, tiles_n, nitems, a_lo, b_lo); \
    } while (0)
    if (planes) {
        if constexpr (sizeof(TO) == 4) {
            const int st = epi.col_sum ? ((!epi.col_sumsq && !epi.col_shift) ? 2 : 1) : 0, gs = epi.general == 1 ? 1 : 0;
            if (st && gs) { ss_set_error("gemm8 (planes): column statistics of a dropout epilogue are not instantiated"); return 1; }
            if (ni == 9) { if (st == 2) G8_PLANES(9, 2, 0); else if (st == 1) G8_PLANES(9, 1, 0); else if (gs) G8_PLANES(9, 0, 1); else G8_PLANES(9, 0, 0); }
            else { if (st == 2) G8_PLANES(8, 2, 0); else if (st == 1) G8_PLANES(8, 1, 0); else if (gs) G8_PLANES(8, 0, 1); else G8_PLANES(8, 0, 0); }
            SS_LAUNCH_CHECK("ss_gemm_planes(gemm8)");
            return 0;
        } else { ss_set_error("gemm8 (planes): f32 output only"); return 1; }
    }
    const int abl = (epi.debug >> 4) & 7;
#if defined(G8_FAST_BUILD)        // tuning builds: three bf16-out main-loop variants only (a full build of this file takes minutes)
    if constexpr (sizeof(TO) == 2) {
        if (epi.col_sum && ni == 9) G8_CASE6(9, 3, 0, 2, 0);
        else if (ni == 9) { switch (pin) { case 3: if (epi.general == 1) G8_CASE6(9, 3, 0, 0, 1); else G8_CASE6(9, 3, 0, 0, 0); break; case 4: G8_CASE(9, 4, 0, false); break; case 7: G8_CASE(9, 7, 0, false); break; default: G8_CASE(9, 0, 0, false); } }
        else { switch (pin) { case 7: G8_CASE(8, 7, 0, false); break; case 11: G8_CASE(8, 11, 0, false); break; default: G8_CASE(8, 3, 0, false); } }
        SS_LAUNCH_CHECK("ss_gemm(gemm8)");
        return 0;
    } else { ss_set_error("gemm8: tuning build"); return 1; }
#else
    // (column statistics never come with the dropout epilogue -- pick_gemm8 refuses the combination -- so these kernels carry one epilogue path)
    if (epi.col_sum && !epi.col_sumsq && !epi.col_shift) {      // plain column sums (a bias gradient)
        if (ni == 9) { if (pin & 3) G8_CASE6(9, 3, 0, 2, 0); else G8_CASE6(9, 0, 0, 2, 0); }
        else { if (pin & 3) G8_CASE6(8, 3, 0, 2, 0); else G8_CASE6(8, 0, 0, 2, 0); }
    }
    else if (epi.col_sum) {                 // column statistics: burst or spread schedule of either tile height
        if (ni == 9) { if (pin & 3) G8_CASE6(9, 3, 0, 1, 0); else G8_CASE6(9, 0, 0, 1, 0); }
        else { if (pin & 3) G8_CASE6(8, 3, 0, 1, 0); else G8_CASE6(8, 0, 0, 1, 0); }
    }
    else if (abl && sizeof(TO) == 2) {      // tuning builds only (bf16 out): which of MFMA / DMA / fragment reads bounds the loop
        if (ni == 9) { switch (abl) { case 1: G8_CASE(9, 0, 1, false); break; case 2: G8_CASE(9, 0, 2, false); break; case 4: G8_CASE(9, 0, 4, false); break; case 5: G8_CASE(9, 0, 5, false); break; case 6: G8_CASE(9, 0, 6, false); break; default: G8_CASE(9, 0, 7, false); } }
        else { switch (abl) { case 1: G8_CASE(8, 0, 1, false); break; case 2: G8_CASE(8, 0, 2, false); break; case 4: G8_CASE(8, 0, 4, false); break; case 5: G8_CASE(8, 0, 5, false); break; case 6: G8_CASE(8, 0, 6, false); break; default: G8_CASE(8, 0, 7, false); } }
    }
    // the default schedule comes in two instantiations: without / with the dropout epilogue (one epilogue path per kernel: fewer registers)
    else if (ni == 9) { switch (pin) { case 1: G8_CASE(9, 1, 0, false); break; case 2: G8_CASE(9, 2, 0, false); break; case 3: if (epi.general == 1) G8_CASE6(9, 3, 0, 0, 1); else G8_CASE6(9, 3, 0, 0, 0); break; default: G8_CASE(9, 0, 0, false); } }
    else { switch (pin) { case 1: G8_CASE(8, 1, 0, false); break; case 2: G8_CASE(8, 2, 0, false); break; case 3: if (epi.general == 1) G8_CASE6(8, 3, 0, 0, 1); else G8_CASE6(8, 3, 0, 0, 0); break; default: G8_CASE(8, 0, 0, false); } }
    SS_LAUNCH_CHECK("ss_gemm(gemm8)");
    return 0;
#endif
#undef G8_CASE
#undef G8_CASE6
#undef G8_PLANES
}
template int gemm8_launch_kc<bf16_t>(int, int, const void*, const void*, void*, int, int, int, const RowMap&, const RowMap&, const GemmEpi&, void*, bool, long long, long long);
template int gemm8_launch_kc<float>(int, int, const void*, const void*, void*, int, int, int, const RowMap&, const RowMap&, const GemmEpi&, void*, bool, long long, long long);
