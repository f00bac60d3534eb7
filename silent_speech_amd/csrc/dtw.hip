// dtw.hip -- DTW cumulative cost + backtrace on gfx950 (reference align.py:5-14 time_warp,
// align.py:16-34 align_from_distances; call site transduction_model.py:126).
//
//   dtw[0][0] = 0, dtw[0][j>0] = dtw[i>0][0] = +inf
//   dtw[i][j] = costs[i][j] + min(dtw[i-1][j], dtw[i][j-1], dtw[i-1][j-1])          i,j >= 1
//   backtrace from (N-1, M-1) while i>0 and j>0: results[i] = j; step to the FIRST minimum of
//   (up, left, diag)  [Python min() tie order]  -> results[i] = smallest j visited in row i.
// f32, one add per cell, no reassociation => bit-exact against the reference.
//
// Mapping (one 256-thread workgroup = 4 waves per matrix; matrices are independent => one per CU slot):
//   * rows are dealt to lanes: lane l of wave w owns R=4 consecutive rows; the 4 waves x 64 lanes x R rows
//     form a 1024-row strip (taller matrices take several strips, chained through a boundary row in HBM).
//   * skewed wavefront: at wave-local step t lane l handles column j = t + 1 - l, so lane l-1 finished the
//     same column one step earlier and hands its last row down with ONE cross-lane shift per step (no LDS,
//     no barrier inside a wave).  Across waves the hand-off goes through an LDS ring and a barrier every
//     G = 64 steps: wave w runs two super-steps behind wave w-1 (blocked wavefront).
//   * costs come from one of two sources, decided per matrix (dtw_source): a matrix with one unit-stride axis of at most 1025 cells is read
//     IN PLACE -- that axis is dealt to the lanes (for a row-major matrix the lanes therefore own COLUMNS and the sweep solves the transposed
//     problem), 8-lane groups fetch their shared 128-byte lines together into a per-wave LDS ring and every lane picks its 16 bytes up
//     when its step comes (see "the cost FIFO in LDS" below).  Everything else is consumed in a SKEWED layout
//     sk[(strip*4+wave)][t][lane][r], one fully coalesced, 16-byte-aligned 1 KiB load per wave and step (prefetched 24 steps ahead);
//     out-of-matrix cells hold +inf.  ss_dtw_align() builds it from an arbitrarily strided cost matrix, the fused loss path writes it
//     directly (loss.hip).
//   * the 2-bit first-minimum direction of every cell (1 byte per lane per step, dirs[wave strip][t / 4][lane][t % 4]) goes to HBM instead
//     of the 4-byte cumulative matrix (8 B/cell algorithmic traffic -> 4.25 B/cell); the workgroup then stages the direction bytes of a
//     wave strip in LDS and walks the path back with scalar arithmetic (results[]).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#if defined(SS_EMU)
inline void __threadfence() {}
#endif

namespace {
constexpr int DW = 4;          // waves per matrix
constexpr int DR = 4;          // rows per lane
constexpr int DG = 64;         // steps per super-step (barrier interval)
constexpr int DRING = 256;     // LDS boundary ring (columns)
constexpr int DNB = 4;           // register buffers of 8 steps in the fast sweep path (4 or 8: a super-step is 8 groups), requested DNB - 1 groups ahead
constexpr int DCH = 2048;      // backtrace window (steps) staged in LDS: 128 KB of dynamic shared memory
constexpr int DESC = 10;       // descriptor fields per matrix
enum { D_N = 0, D_M, D_COST_OFF, D_SI, D_SJ, D_SK_OFF, D_DIRS_OFF, D_BND_OFF, D_RES_OFF };
}

__host__ __device__ static inline long long dtw_strips(long long n) { long long rows = n - 1; long long cap = DW * 64 * DR; return rows <= 0 ? 0 : (rows + cap - 1) / cap; }
__host__ __device__ static inline long long dtw_tsteps(long long m) { return m <= 1 ? 0 : (m - 1) + 63; }
__host__ __device__ static inline long long dtw_tpitch(long long m) { return (dtw_tsteps(m) + DG - 1) / DG * DG; }     // steps per wave strip in `dirs`: whole super-steps
// Where the sweep takes its costs from (per matrix).  0: the skewed strips (written by dtw_skew_kernel or by the fused loss kernels).
// 1 / 2: IN PLACE from the caller's matrix -- possible when one axis is unit-stride, that axis fits one strip (<= 1024 interior cells) and
// is dealt to the lanes: 1 = lanes own ROWS (a column-major matrix: the costs.T view of transduction_model.py:126), 2 = lanes own COLUMNS
// (row-major; the sweep then solves the transposed problem and the backtrace reads its codes accordingly).
__host__ __device__ static inline int dtw_source(bool have_costs, long long n, long long m, long long si, long long sj)
{
    if (!have_costs || n <= 1 || m <= 1) return 0;
    if (sj == 1 && m >= 5 && m - 1 <= DW * 64 * DR && si >= 4 && si < (1 << 22)) return 2;
    if (si == 1 && n >= 5 && n - 1 <= DW * 64 * DR && sj >= 4 && sj < (1 << 22)) return 1;
    return 0;
}

extern "C" int ss_dtw_source(int n, int m, int64_t stride_i, int64_t stride_j) { return dtw_source(true, n, m, stride_i, stride_j); }

extern "C" int64_t ss_dtw_workspace_bytes(int n, int m, int64_t* sk_bytes, int64_t* dirs_bytes, int64_t* bnd_bytes)
{
    long long strips = dtw_strips(n), ts = dtw_tsteps(m);
    long long sk = strips * DW * ts * 64 * DR * 4, dirs = strips * dtw_tpitch(m) * 256, bnd = 2LL * (m > 0 ? m : 0) * 4;
    const long long dirs_t = dtw_strips(m) * dtw_tpitch(n) * 256;          // the transposed sweep of a row-major matrix (dtw_source() == 2)
    if (dirs_t > dirs) dirs = dirs_t;
    sk = (sk + 255) / 256 * 256; dirs = (dirs + 255) / 256 * 256; bnd = (bnd + 255) / 256 * 256;
    if (sk_bytes) *sk_bytes = sk;
    if (dirs_bytes) *dirs_bytes = dirs;
    if (bnd_bytes) *bnd_bytes = bnd;
    return sk + dirs + bnd;
}

// ------------------------------------------------------------------ cost matrix -> skewed strips
// One workgroup = 64 rows (16 lanes of a wave strip) x SKT steps, through an LDS tile [64 rows][SKT]: the skewed layout needs, for
// row 4 l + r, the SKT columns t0 + 1 - l .. -- a parallelogram of the matrix.  It is fetched along the matrix' unit-stride axis
// (rows of 128 bytes for a row-major matrix; for a column-major one -- the costs.T view of transduction_model.py:126 -- column j is
// needed by the lanes l = t0 + 1 - j .. of the tile: up to 64 consecutive rows) and leaves as 256-byte pieces of the strip's
// 1 KiB rows.  Every thread has all of its 8 (12) loads in flight at once.  (A gather straight from the matrix, one element per
// thread, ran at 1.4 TB/s of combined traffic for a batch: 64 distinct cache lines per 256 outputs.)
constexpr int SKT = 32, SKR = 64;
__global__ __launch_bounds__(256) void dtw_skew_kernel(const float* __restrict__ costs, const long long* __restrict__ desc, unsigned char* __restrict__ ws,
                                                       int* __restrict__ results, int ntiles_max, int dbg)
{
    __shared__ float tile[SKR][SKT + 1];
    const long long* d = desc + (long long)blockIdx.y * DESC;
    const int N = (int)d[D_N], M = (int)d[D_M];
    int* res = results + d[D_RES_OFF];
    const int tid = threadIdx.x;
    for (int i = blockIdx.x * 256 + tid; i < N; i += gridDim.x * 256) res[i] = 0;
    if (dtw_source(!(dbg & 8), N, M, d[D_SI], d[D_SJ])) return;                  // the sweep reads this matrix in place
    const int ts = (int)dtw_tsteps(M), nrt = (int)dtw_strips(N) * DW * (256 / SKR);      // row tiles: quarters of the wave strips
    const int rt = blockIdx.x / ntiles_max, t0 = (blockIdx.x - rt * ntiles_max) * SKT;
    if (rt >= nrt || t0 >= ts) return;
    float* sk = (float*)(ws + d[D_SK_OFF]);
    const float* c = costs + d[D_COST_OFF];
    const long long si = d[D_SI], sj = d[D_SJ];
    const int row0 = 1 + rt * SKR;                                        // matrix row of tile row 0
    const int l0 = (rt * (SKR / DR)) & 63;                                // lane of tile row 0 inside its wave strip
    if (sj == 1) {
#pragma unroll
        for (int pp = 0; pp < SKR / 8; ++pp) {                            // 8 rows x 32 columns per pass
            const int rr = pp * 8 + (tid >> 5), cc = tid & 31;
            const int i = row0 + rr, j = t0 + cc + 1 - (l0 + (rr >> 2));
            tile[rr][cc] = (i < N && j >= 1 && j < M) ? c[(long long)i * si + j] : INFINITY;
        }
    } else if (si == 1) {
        const int rr = tid & 63, l = l0 + (rr >> 2), i = row0 + rr;
#pragma unroll
        for (int pp = 0; pp < (SKT + SKR / DR - 1 + 3) / 4; ++pp) {       // 4 columns x 64 rows per pass; column j serves the lanes with 0 <= j - 1 + l - t0 < SKT
            const int j = t0 + 1 - (l0 + SKR / DR - 1) + pp * 4 + (tid >> 6), cc = j - 1 + l - t0;
            if (cc >= 0 && cc < SKT) tile[rr][cc] = (i < N && j >= 1 && j < M) ? c[(long long)j * sj + i] : INFINITY;
        }
    } else {
        const int rr = tid & 63, l = l0 + (rr >> 2), i = row0 + rr;
#pragma unroll
        for (int pp = 0; pp < SKT / 4; ++pp) {
            const int cc = pp * 4 + (tid >> 6), j = t0 + cc + 1 - l;
            tile[rr][cc] = (i < N && j >= 1 && j < M) ? c[(long long)i * si + (long long)j * sj] : INFINITY;
        }
    }
    __syncthreads();
    // strip row t: [64 lanes][4 rows] floats; this tile owns the 64 consecutive floats from lane l0
    const int kw = rt / (256 / SKR);
    float* out = sk + ((long long)kw * ts + t0) * (64 * DR) + l0 * DR + (tid & 63);
#pragma unroll
    for (int pp = 0; pp < SKT / 4; ++pp) {
        const int cc = pp * 4 + (tid >> 6);
        if (t0 + cc < ts) out[(long long)cc * (64 * DR)] = tile[tid & 63][cc];
    }
}

// ------------------------------------------------------------------ cumulative cost + backtrace
// lane l <- lane l-1, lane 0 <- `first`: one DPP move (wave_shr:1) instead of a ds_bpermute round trip through the LDS
// crossbar -- this shift sits on the serial dependency chain of every DTW step.
__device__ __forceinline__ float wave_shift_in(float v, float first, int lane) {
#if defined(SS_EMU)
    const float u = __shfl_up(v, 1);
    return lane == 0 ? first : u;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(first), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
#endif
}
// lane l <- lane l+1 (wave_shl:1): lane 0 of the boundary register walks through the 64 values of a super-step, one per step, so the
// step reads "the row above the strip" from its OWN lane-0 slot -- no v_readlane (SGPR round trip + 4-5 hazard nops on the serial chain)
__device__ __forceinline__ float wave_rotate_down(float v) {
#if defined(SS_EMU)
    return __shfl_down(v, 1);
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
#endif
}
// bits = 2 * bits + (x == y): compare into VCC, add-with-carry doubles and inserts the bit (2 instructions per flag; the
// select / shift / or form the compiler builds from C costs 5-6 per cell)
__device__ __forceinline__ void push_eq(unsigned& bits, float x, float y) {
#if defined(SS_EMU)
    bits = (bits << 1) | (x == y ? 1u : 0u);
#else
    asm("v_cmp_eq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(x), "v"(y) : "vcc");
#endif
}

// ring slot of column s in the LDS hand-over between waves: (s + 62) & 255.  Lane 63 handles column t - 62 at step t, so inside a
// super-step (64 steps from a multiple of 64) it writes the CONSECUTIVE slots (t0 & 255) + c: a constant ds_write offset, no address
// arithmetic on the step
__device__ __forceinline__ int ring_slot(int s) { return (s + 62) & (DRING - 1); }

// One wavefront step.  No column predicate: cells outside the matrix carry cost +inf in the skewed layout, so a lane that
// has not reached column 1 yet (or is past M-1) only turns +inf into +inf and its state stays what the recurrence needs
// (dtw[i][0] = dtw[0][j] = +inf); the direction bytes it writes there are never read.  Branch-free: the per-step chain is
// the 4 dependent (min3, add) pairs plus one DPP shift.  Returns this step's direction byte; slot: ring word (lane 63: the ring
// of the next wave; the other lanes hit a dump row).
__device__ __forceinline__ unsigned dtw_step(int lane, const f32x4& cv, float top, float (&prev)[DR], float& diag_sv, float& last_out, float* slot)
{
    const float up_in = wave_shift_in(last_out, top, lane);          // lane 0 takes the row above the strip / the previous wave's last row
    float a = up_in, dg = diag_sv;
    unsigned bits = 0;
#pragma unroll
    for (int r = 0; r < DR; ++r) {
        const float b = prev[r];
        // first minimum of (up, left, diag) [Python min() tie order]: the VALUE is min3 -- one instruction on the serial chain
        // (min3 -> add -> next row's min3); which candidate it was is recovered off the chain as two equality flags per cell: the
        // backtrace takes "up" if best == up, else "left" if best == left, else "diag" (up wins ties, then left)
        const float best = fminf(fminf(a, b), dg);
        push_eq(bits, best, b); push_eq(bits, best, a);                    // cell code = 2 [best == left] + [best == up], row r at bits 2 (3 - r)
        const float nv = cv[r] + best;
        dg = b; a = nv; prev[r] = nv;
    }
    last_out = a;
    diag_sv = up_in;
    *slot = a;
    return bits;
}
// Direction bytes: dirs[wave strip][t / 4][lane][t % 4] -- the four steps of a lane share a dword, so the fast paths store once per 4 steps
// (a quarter of the store instructions, and fewer stores between the counted waits of the cost prefetch)
__device__ __forceinline__ long long dir_byte(int t, int lane) { return ((long long)(t >> 2) * 64 + lane) * 4 + (t & 3); }

// ---- costs of a group of 8 steps -> registers WITHOUT the compiler knowing that these are loads.  Left to hipcc, every step is its
// own basic block and the first use of a prefetched group is preceded by s_waitcnt vmcnt(0): that drains the 8 loads issued just before
// (the NEXT group), i.e. one full memory round trip on the serial chain per 8 steps -- 40 % of the kernel.  Loads issued from asm are
// invisible to the wait insertion; the counted waits below are written by hand (vector memory operations retire in issue order).
template <int OFF>
__device__ __forceinline__ void gload128(f32x4& v, const float* p) {
#if defined(SS_EMU)
    __builtin_memcpy(&v, (const char*)p + OFF, 16);
#else
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
#endif
}
// ---- in-place source: the cost FIFO in LDS.  Read straight into registers, lane l needs 16 bytes of line t + 1 - l at step t: 64 distinct
// cache lines per load instruction, and the texture addresser takes them at one line per clock -- 256 clocks per step for the four waves of a
// matrix, more than the step's arithmetic (measured: tools/ta_probe, 262 clocks).  But the 8 lanes of a group g = l / 8 own the 8 pieces of the
// SAME 128-byte line, only at different times.  So fetch-step T loads line T - 8 g for all 8 lanes of group g at once (8 lines per instruction,
// straight into LDS: global_load_lds, 1 KiB slot T % 32 of the wave's ring), and lane l = 8 g + i picks its piece up i steps later: at step t it
// reads slot (t + 1 - i) % 32.  Every lane reads only what it fetched itself (a per-lane FIFO), so no barrier is involved; completion of the
// fetches is counted by hand (vmcnt), the reads travel one group of 8 steps ahead of their use (lgkmcnt).
constexpr int CR_SLOTS = 32, CR_WAVE = CR_SLOTS * 1024, CR_PAD = 8192;     // pad: the lanes' read base addresses reach 7 KiB below their ring
static_assert(DW * CR_WAVE == DCH * 64, "the four cost rings alias the backtrace window");
__device__ __forceinline__ void cost_fetch_s(unsigned m0, unsigned voff, const float* sbase, unsigned char* ring_emu) {   // scalar line base + lane offset
#if defined(SS_EMU)
    __builtin_memcpy(ring_emu + m0 + (threadIdx.x & 63) * 16, (const char*)sbase + voff, 16);
#else
    (void)ring_emu;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0), "v"(voff), "s"(sbase) : "memory", "m0");
#endif
}
__device__ __forceinline__ void cost_fetch_v(unsigned m0, const float* p, unsigned char* ring_emu) {                      // per-lane address (clamped lines)
#if defined(SS_EMU)
    __builtin_memcpy(ring_emu + m0 + (threadIdx.x & 63) * 16, p, 16);
#else
    (void)ring_emu;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0), "v"(p) : "memory", "m0");
#endif
}
template <int OFF>
__device__ __forceinline__ void cost_read(f32x4& v, unsigned addr, const unsigned char* ring_emu) {
#if defined(SS_EMU)
    __builtin_memcpy(&v, ring_emu + addr + OFF, 16);
#else
    (void)ring_emu;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
#endif
}
__device__ __forceinline__ void wait_lds() {
#if !defined(SS_EMU)
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
#endif
}
// p = this lane's cost pointer at step tg + 4 (8 steps x 1 KiB around it: offsets -4096 .. 3072 fit the 13-bit signed immediate)
__device__ __forceinline__ void issue_group(f32x4 (&buf)[8], const float* p) {
    gload128<-4096>(buf[0], p); gload128<-3072>(buf[1], p); gload128<-2048>(buf[2], p); gload128<-1024>(buf[3], p);
    gload128<0>(buf[4], p); gload128<1024>(buf[5], p); gload128<2048>(buf[6], p); gload128<3072>(buf[7], p);
}
__device__ __forceinline__ void pin_group(f32x4 (&buf)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) pin_vgpr(buf[e]);
}
template <int N> __device__ __forceinline__ void wait_vm() {       // s_waitcnt vmcnt(N), expcnt / lgkmcnt untouched
#if !defined(SS_EMU)
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x0f70 | ((N >> 4) << 14));
#endif
}
template <int I, int N, class F> __device__ __forceinline__ void dfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); dfor<I + 1, N>(f); }
}
__device__ __forceinline__ int dtw_uniform(int v) { return wave_uniform(v); }
__device__ __forceinline__ unsigned dtw_readlane(unsigned v, int l) {      // l wave-uniform
#if defined(SS_EMU)
    return __shfl(v, l);
#else
    return __builtin_amdgcn_readlane(v, l);
#endif
}

// dbg (SS_DTW_DEBUG, tuning only -- results are wrong): 1 no backtrace, 2 no sweep, 4 every super-step on the generic path, 8 never in place
// HAVE_COSTS = false: the instantiation behind ss_dtw_align_skewed (the training step's loss): strips only, none of the in-place paths'
// registers (the three straight-line 64-step bodies side by side push the kernel past 256 VGPRs, which costs the strip path moves)
template <bool HAVE_COSTS>
__global__ __launch_bounds__(256) void dtw_kernel(const long long* __restrict__ desc, unsigned char* __restrict__ ws, int* __restrict__ results, int dbg,
                                                  const float* __restrict__ costs)
{
    __shared__ float lds_bnd[DW + 1][DRING];              // [w] = ring read by wave w (written by wave w-1); [DW] = dump / last wave's unused ring
    SS_DYN_SMEM(chunk_raw);                               // DCH x 64 direction bytes of the backtrace window
    unsigned char* chunk = (unsigned char*)chunk_raw;
    const long long* d = desc + (long long)blockIdx.x * DESC;
    const int N = (int)d[D_N], M = (int)d[D_M];
    if (N <= 1 || M <= 1) return;                                   // no interior cell: results stay 0 (align.py:24)
    // in-place sources: the sweep runs on (Np, Mp) = (cells dealt to lanes, cells along the steps); for source 2 that is the transposed problem
    const int source = HAVE_COSTS ? dtw_source(!(dbg & 8), N, M, d[D_SI], d[D_SJ]) : 0;
    const int Np = source == 2 ? M : N, Mp = source == 2 ? N : M;
    const long long ld = source == 2 ? d[D_SI] : d[D_SJ];           // element stride along the steps (the lanes' axis is unit-stride)
    const float* sk = (const float*)(ws + d[D_SK_OFF]);
    unsigned char* dirs = ws + d[D_DIRS_OFF];
    float* bnd = (float*)(ws + d[D_BND_OFF]);
    int* res = results + d[D_RES_OFF];
    const int tid = threadIdx.x, lane = tid & 63, w = dtw_uniform(tid >> 6);      // scalar: step bounds become s_cbranch, not exec masks
    const int ts = (int)dtw_tsteps(Mp), nstrips = (int)dtw_strips(Np), tsp = (int)dtw_tpitch(Mp);
    const int nss = (ts + DG - 1) / DG;
    const bool multi = nstrips > 1;
    if (source) for (int i = tid; i < N; i += 256) res[i] = 0;       // (the skew launch zeroes them too; kept here so that this kernel does not depend on it)

    // ---- in-place source, per lane: owned cells o0 .. o0 + 3 of the lanes' axis (16 contiguous bytes of every step's line)
    const float* cmat = source ? costs + d[D_COST_OFF] : nullptr;
    const int o0 = 1 + (w * 64 + lane) * DR, grp = lane >> 3, sub = lane & 7;
    const bool straddle = source && o0 <= Np - 1 && o0 + DR - 1 > Np - 1;       // the 16 bytes run past the end of the line: fine on every line but the last
    const int o0c = (!source || o0 <= Np - 1) ? o0 : Np - DR;                   // lanes past the matrix read valid memory; their cells are never looked at
    const int hi_l = Mp - 1 - (straddle ? 1 : 0);                               // last line this lane may load 16 bytes from
    const float* lane_base = cmat + o0c;
    const unsigned voff = source ? (unsigned)(((56 - 8 * grp) * ld + o0c) * 4) : 0u;      // fetch-step T: line T - 8 grp = (T - 56) + (56 - 8 grp)
    f32x4 last4 = {0.f, 0.f, 0.f, 0.f};
    if (straddle) {
#pragma unroll
        for (int r = 0; r < DR; ++r) if (o0 + r <= Np - 1) last4[r] = cmat[(long long)(Mp - 1) * ld + o0 + r];
    }
    wait_vm<0>();                                        // vmcnt(0) HERE: left to the compiler the wait lands in front of the first patched step of every
    pin_vgpr(last4);                                     // clamped super-step, where it also drains the cost fetches just issued
    const unsigned lbase = lds_byte_address(chunk);      // LDS byte address of the dynamic segment
    unsigned char* const ring_emu = chunk;               // (emulator: ring addresses are offsets into the dynamic segment)
    const unsigned ring_w = lbase + CR_PAD + (unsigned)w * CR_WAVE;           // this wave's cost ring (aliases the backtrace window: the phases do not overlap)
    const unsigned rd1 = ring_w + lane * 16 - sub * 1024;                     // read base: slot (tau - sub) for tau >= sub ..
    unsigned rsel[7];                                                          // .. and for tau = 0 .. 6 the lanes with sub > tau wrap to the top of the ring
#pragma unroll
    for (int q = 0; q < 7; ++q) rsel[q] = sub > q ? rd1 + CR_WAVE : rd1;

    for (int k = 0; k < nstrips && !(dbg & 2); ++k) {
        const int rowbase = ((k * DW + w) * 64 + lane) * DR;
        float prev[DR];
#pragma unroll
        for (int r = 0; r < DR; ++r) prev[r] = INFINITY;             // dtw[i][0] = inf
        float diag_sv = rowbase == 0 ? 0.f : INFINITY;               // dtw[i-1][0]; dtw[0][0] = 0
        float last_out = INFINITY;
        const float* skp = sk + ((long long)(k * DW + w) * ts) * (64 * DR) + lane * DR;
        unsigned char* const dstrip = dirs + ((long long)(k * DW + w) * tsp) * 64;   // dirs[wave strip][t / 4][lane][t % 4]: a wave's steps are contiguous (full cache lines)
        const float* bnd_prev = bnd + ((k + 1) & 1) * Mp;
        float* bnd_cur = bnd + (k & 1) * Mp;
        float* const ring_next = lds_bnd[w + 1];
        float* const dump = &lds_bnd[DW][lane];                       // lanes 0..62: words lane .. lane + 63 of the dump row
        // Fast super-steps of the STRIP source (all 64 steps inside the matrix' ts, one strip): DNB register buffers of 8 steps, group g in
        // buffer g % DNB, requested DNB - 1 groups ahead (8 buffers would keep 7 KiB per wave in flight for batches that stream their costs
        // from HBM rather than L2, but the 64-step straight-line body then runs out of registers: 512 + spills).  At every super-step boundary
        // the first DNB - 1 groups of the NEXT super-step are loaded AND settled, so no request is in flight across control flow (a register
        // copy the compiler places at a join would copy a value that has not arrived).  Group start beyond ts - 8: clamped (never consumed:
        // that super-step takes the generic path).
        const bool fast_ok = !source && !multi && !(dbg & 4) && ts >= DG;
        f32x4 cbuf[DNB][8];
        auto group_ptr = [&](int tg) { const int tc = tg + 8 <= ts ? tg : ts - 8; return skp + (long long)(tc + 4) * (64 * DR); };
        if (fast_ok) {
            dfor<0, DNB - 1>([&](auto gc) { constexpr int g = gc; issue_group(cbuf[g], group_ptr(g * 8)); });
            wait_vm<0>();
            dfor<0, DNB - 1>([&](auto gc) { constexpr int g = gc; pin_group(cbuf[g]); });
        }
        auto super_fast = [&](int t0, float topv) {
            unsigned* const dpb = (unsigned*)(dstrip + dir_byte(t0, lane));
            float* const slot0 = lane == 63 ? ring_next + (t0 & (DRING - 1)) : dump;
            unsigned dacc = 0;
            dfor<0, 8>([&](auto jc) {
                constexpr int j = jc;
                issue_group(cbuf[(j + DNB - 1) % DNB], group_ptr(t0 + (j + DNB - 1) * 8));
                // younger than the loads of group j: the 8 (DNB - 1) loads of the groups requested since, and the direction stores of
                // the steps between.  Counting only the loads keeps the wait correct whatever the compiler does with the stores
                // (vector memory operations retire in issue order)
                if constexpr (j >= DNB - 1) wait_vm<8 * (DNB - 1)>();
                pin_group(cbuf[j % DNB]);
                dfor<0, 8>([&](auto ec) {
                    constexpr int e = ec, c = j * 8 + e;
                    const unsigned bits = dtw_step(lane, cbuf[j % DNB][e], topv, prev, diag_sv, last_out, slot0 + c);
                    dacc = (c & 3) ? dacc | (bits << (8 * (c & 3))) : bits;
                    if constexpr ((c & 3) == 3) dpb[(c >> 2) * 64] = dacc;
                    topv = wave_rotate_down(topv);
                });
            });
            wait_vm<0>();
            dfor<0, DNB - 1>([&](auto gc) { constexpr int g = gc; pin_group(cbuf[g]); });
        };
        // ---- in-place sources run EVERY super-step on the cost ring.  Cells outside the matrix need no +inf: a cell left of column 1 only ever
        // combines +inf states (inf + any finite cost = inf), and cells past the last line / past the lanes' axis feed only cells that are
        // outside too -- so out-of-range fetches are merely clamped to valid addresses (CLAMP: per-lane line index clamped to [0, hi_l]; the
        // straddling lane takes its last line from `last4`), and super-steps whose every fetch is in range use the scalar-base form.
        // Fetch group D = fetch-steps 8 D + 1 .. 8 D + 8; the reads of sweep group J (steps 8 J .. 8 J + 7) need groups J - 1 and J.  At the start
        // of sweep group J: groups <= J + 1 have landed (vmcnt(8): group J + 2, requested one sweep group ago, may still be in flight) -> the reads
        // of group J + 1 are issued -> group J + 3 is requested (its slots are 9 .. 30 behind the oldest slot those reads touch) -> the 8 steps of
        // group J -> one lgkmcnt(0) settles the reads.  Nothing but LDS-bound fetches crosses control flow (no register destination).
        f32x4 rb[2][8];
        const float* fbase = nullptr;                                         // scalar-base form: line base of the next fetch-step (fetches are requested in order)
        auto fetch = [&](auto clampc, int t0, auto crelc) {                   // fetch-step T = t0 + crel (t0 a multiple of 64 -> slot = crel & 31)
            constexpr bool CLAMP = decltype(clampc)::value;
            constexpr int crel = decltype(crelc)::value;
            const unsigned m0 = ring_w + (unsigned)((crel & (CR_SLOTS - 1)) * 1024);
            if constexpr (!CLAMP) { cost_fetch_s(m0, voff, fbase, ring_emu); fbase += ld; }
            else {
                int li = t0 + crel - 8 * grp;
                li = li < 0 ? 0 : (li > hi_l ? hi_l : li);
                cost_fetch_v(m0, lane_base + (long long)li * ld, ring_emu);
            }
        };
        auto reads = [&](auto jc) {                                           // the 8 reads of sweep group j (relative to the super-step; j = 8: the next one's first)
            constexpr int j = jc;
            dfor<0, 8>([&](auto ec) {
                constexpr int e = ec, tau = (j * 8 + e + 1) & (CR_SLOTS - 1);
                if constexpr (tau < 7) cost_read<tau * 1024>(rb[j & 1][e], rsel[tau], ring_emu);
                else cost_read<tau * 1024>(rb[j & 1][e], rd1, ring_emu);
            });
        };
        auto settle = [&](int par) {
            wait_lds();
#pragma unroll
            for (int e = 0; e < 8; ++e) { if (par) pin_vgpr(rb[1][e]); else pin_vgpr(rb[0][e]); }
        };
        if (source) {
            dfor<0, 32>([&](auto cc) { constexpr int c = cc; fetch(std::true_type{}, 0, std::integral_constant<int, c - 7>{}); });     // groups -1 .. 2
            wait_vm<16>();
            reads(std::integral_constant<int, 0>{});
            settle(0);
        }
        auto super_ring = [&](auto clampc, int t0, float topv) {
            constexpr bool CLAMP = decltype(clampc)::value;
            unsigned* const dpb = (unsigned*)(dstrip + dir_byte(t0, lane));
            float* const slot0 = lane == 63 ? ring_next + (t0 & (DRING - 1)) : dump;
            unsigned dacc = 0;
            if constexpr (!CLAMP) fbase = cmat + (long long)(t0 + 25 - 56) * ld;        // first fetch-step of this super-step: t0 + 25
            dfor<0, 8>([&](auto jc) {
                constexpr int j = jc;
                wait_vm<8>();
                reads(std::integral_constant<int, j + 1>{});
                dfor<0, 8>([&](auto ec) { constexpr int e = ec; fetch(clampc, t0, std::integral_constant<int, (j + 3) * 8 + 1 + e>{}); });
                dfor<0, 8>([&](auto ec) {
                    constexpr int e = ec, c = j * 8 + e;
                    unsigned bits;
                    if constexpr (CLAMP) {
                        f32x4 cv = rb[j & 1][e];
                        const bool patch = straddle && t0 + c + 1 - lane >= Mp - 1;
#pragma unroll
                        for (int r = 0; r < DR; ++r) cv[r] = patch ? last4[r] : cv[r];
                        bits = dtw_step(lane, cv, topv, prev, diag_sv, last_out, slot0 + c);
                    } else bits = dtw_step(lane, rb[j & 1][e], topv, prev, diag_sv, last_out, slot0 + c);
                    dacc = (c & 3) ? dacc | (bits << (8 * (c & 3))) : bits;
                    if constexpr ((c & 3) == 3) dpb[(c >> 2) * 64] = dacc;
                    topv = wave_rotate_down(topv);
                });
                settle((j + 1) & 1);
            });
        };
        for (int ss = 0; ss < nss + 2 * (DW - 1); ++ss) {
            const int u = ss - 2 * w;
            if (u >= 0 && u < nss) {
                const int t0 = u * DG;
                // the 64 values lane 0 will need from above during this super-step (one per step), fetched up front
                // (single-strip matrices -- every fast super-step -- have nothing above wave 0.  The global load of the strip boundary must not
                // even sit on a path that reaches a fast super-step: the compiler would wait vmcnt(0) for it in front of the first step, and that
                // also drains the cost loads issued just before -- one memory round trip per super-step)
                auto top_lds = [&]() {
                    const float v = w == 0 ? INFINITY : lds_bnd[w][ring_slot(t0 + 1 + lane)];
                    // settle the LDS read NOW: left pending, the compiler re-waits lgkmcnt(0) at EVERY step, which also
                    // drains that step's ring write (an LDS round trip on the serial chain of each of the 64 steps)
                    wait_lds();
                    return v;
                };
                if (source) {
                    // every fetch this super-step requests (fetch-steps t0 + 25 .. t0 + 88, lines T - 56 .. T) in range for every lane, the last line excluded?
                    if (t0 >= 64 && t0 + 90 <= Mp) super_ring(std::false_type{}, t0, top_lds());
                    else super_ring(std::true_type{}, t0, top_lds());
                } else if (fast_ok && t0 + DG <= ts) {
                    super_fast(t0, top_lds());
                } else {
                    float topv = top_lds();
                    { const int sb = t0 + 1 + lane; if (w == 0 && k > 0 && sb < Mp) topv = bnd_prev[sb]; }
                    f32x4 cb[8];
                    const f32x4 inf4 = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
                    for (int e = 0; e < 8; ++e) cb[e] = t0 + e < ts ? *(const f32x4*)(skp + (long long)(t0 + e) * (64 * DR)) : inf4;
                    for (int g = 0; g < DG / 8; ++g) {
                        f32x4 nb[8];
                        const int tn = t0 + (g + 1) * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) nb[e] = (g + 1 < DG / 8 && tn + e < ts) ? *(const f32x4*)(skp + (long long)(tn + e) * (64 * DR)) : inf4;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int t = t0 + g * 8 + e;
                            if (t < ts) {
                                const int s = t + 1 - lane;
                                dstrip[dir_byte(t, lane)] = (unsigned char)dtw_step(lane, cb[e], topv, prev, diag_sv, last_out, lane == 63 ? ring_next + ring_slot(s) : dump);
                                if (multi && w == DW - 1 && lane == 63 && s >= 1 && s < Mp) bnd_cur[s] = last_out;
                                topv = wave_rotate_down(topv);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) cb[e] = nb[e];
                    }
                }
            }
            __syncthreads();
        }
        wait_vm<0>();         // cost fetches still on their way into the ring (it becomes the backtrace window)
        __threadfence();      // strip boundary row + direction bytes visible before they are re-read
        __syncthreads();
    }
    if (dbg & 1) return;

    // ---- backtrace: all four waves walk the path in lockstep (wave-uniform = scalar arithmetic; wave 0 writes results), so that the
    // whole workgroup stages the direction bytes of a wave strip: window [t_lo, t_hi] x 64 lanes, up to DCH steps = the whole strip
    // for M <= DCH - 62.  The walk is organised by LANE COLUMN (the 4 rows of one lane): inside a column the bytes come from a
    // register cache (lane i = step c_top - i of column l) through v_readlane, and a cell costs ~12 scalar instructions (a wave on
    // its own issues one instruction per ~5 cycles: the instruction count IS the latency).  The cache of the next column (l - 1) is
    // requested on entry to a column -- its top step is known: t only decreases -- so that its LDS latency passes under the walk.
    // results[i] = the column at which the path LEAVES row i (smallest j visited): collected in lanes 0..3 of a register (compare +
    // select: v_writelane would need value and lane select in scalar registers, two constant-bus reads gfx9 does not encode) and
    // stored once per column.
    int p = N - 1, s = M - 1;
    int cur_kw = -1, t_lo = 0;
    unsigned cache = 0, cache_nx = 0, rv = 0;
    int c_top = 0, nx_top = 0;
    bool nx_ok = false;
    auto stage = [&](int kw, int t) {                                     // window [.., t] of wave strip kw -> LDS (whole workgroup); whole dwords: 4-step units
        __syncthreads();
        const int t_hi = t | 3;
        t_lo = t_hi - DCH + 1 < 0 ? 0 : t_hi - DCH + 1; cur_kw = kw; nx_ok = false;
        const unsigned char* src = dirs + ((long long)kw * tsp + t_lo) * 64;
        const int nd = (t_hi - t_lo + 1) * 4;                             // 16-byte pieces, contiguous in the workspace
        for (int base = 0; base < nd; base += 256 * 8) {                  // 8 loads in flight per thread, then the 8 LDS writes
            u32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = base + e * 256 + tid;
                if (idx < nd) v[e] = *(const u32x4*)(src + (long long)idx * 16);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = base + e * 256 + tid;
                if (idx < nd) *(u32x4*)(chunk + idx * 16) = v[e];
            }
        }
        __syncthreads();
    };
    auto column = [&](int l, int top) -> unsigned {                       // lane i <- byte of step top - i (0 below the window)
        const int tt = top - lane;
        return tt >= t_lo ? (unsigned)chunk[dir_byte(tt - t_lo, l)] : 0u;
    };
    if (source == 2) {
        // ---- transposed sweep (lanes own COLUMNS of the matrix, steps run along its rows): a byte holds the codes of columns 4 l + 1 .. 4 l + 4 of
        // row t + 1 - l, code = 2 [best == up] + [best == left] in the MATRIX' terms (the sweep's "left" is the matrix' "up"), up wins ties.
        // Same organisation as below with the roles swapped: inside a lane column a move up steps through the register cache (idx), a move
        // left shifts to the next code of the byte (sh).  results[i] = the column at the LAST cell visited in row i: lane idx of `rv` follows
        // the row at cache position idx; rows are stored when the path has left them (the row the walk stands in is carried on).
        int nx_l = -1;
        while (p > 0 && s > 0) {
            const int q = s - 1;
            const int kw = q / (64 * DR), l = (q / DR) & 63, r = q % DR;
            const int t = p - 1 + l;
            if (kw != cur_kw || t < t_lo) stage(kw, t);
            if (nx_ok && nx_l == l && nx_top >= t && nx_top - t < 64) { cache = cache_nx; c_top = nx_top; }
            else { c_top = t; cache = column(l, t); }
            nx_ok = l > 0;
            if (nx_ok) { nx_top = t - 1; nx_l = l - 1; cache_nx = column(l - 1, nx_top); }
            const int p0c = c_top + 1 - l;                                // row at cache position 0
            const int idx_in = c_top - t, idx_max = c_top - t_lo < 63 ? c_top - t_lo : 63;
            const int ilim = idx_max < p0c - 1 ? idx_max : p0c - 1;       // cache exhausted | row 0 reached: one limit on idx
            int idx = idx_in, sh = 2 * (DR - 1 - r), up;
            do {
                rv = lane == idx ? (unsigned)s : rv;
                const unsigned code = (dtw_readlane(cache, idx) >> sh) & 3u;
                up = code != 1u;                                          // up or diagonal: the row is left
                const int lf = (int)((code >> 1) ^ 1u);                   // left or diagonal: the column moves
                idx += up; s -= lf; sh += 2 * lf;
            } while (((2 * (DR - 1) - sh) | (s - 1) | (ilim - idx)) >= 0);      // column left | path ended (s == 0 | p == 0) | cache exhausted
            const int idx_hi = s == 0 ? idx - up : idx - 1;               // rows left in this visit (+ the current one where the path ends in it)
            if (w == 0 && lane >= idx_in && lane <= idx_hi) res[p0c - lane] = (int)rv;
            p = p0c - idx;
        }
        return;
    }
    while (p > 0 && s > 0) {
        const int q = p - 1;
        const int kw = q / (64 * DR), l = (q / DR) & 63;
        int r = q % DR;
        const int pbase = p - r;                                          // row of r = 0
        int t = s - 1 + l;
        if (kw != cur_kw || t < t_lo) stage(kw, t);
        if (nx_ok && nx_top >= t && nx_top - t < 64) { cache = cache_nx; c_top = nx_top; }
        else { c_top = t; cache = column(l, t); }
        nx_ok = l > 0;
        if (nx_ok) { nx_top = t - 1; cache_nx = column(l - 1, nx_top); }
        // Walk state of the column, kept small (every instruction of this loop is ~5 cycles of latency): idx = c_top - t, the bit
        // position sh = 2 (3 - r) of the current row's code (moving up a row = sh + 2, leaving the column = sh > 6) and s.  Lane k of
        // rv holds the result of row 3 - k; the rows left in this visit are exactly those with sh_in <= 2 k < sh_out.
        int idx = c_top - t, idx_max = c_top - t_lo < 63 ? c_top - t_lo : 63;            // the cache holds steps c_top - idx_max .. c_top
        int sh = 2 * (DR - 1 - r);
        const int sh_in = sh, lane2 = 2 * lane;
        int sh_out;
        for (;;) {
            // one cell per trip, branch-free (the compiler's if / else form of this loop was 35 instructions and three taken branches per
            // cell): left (code 2) moves one column and leaves no row; up (code & 1) leaves the row at column s; diagonal does both
            unsigned code;
            // s and idx move together (s + idx is constant in this loop): "the path ended" (s < 1) and "cache exhausted" (idx > idx_max) are ONE limit on s
            const int s_lim = s + idx - idx_max > 1 ? s + idx - idx_max : 1;
            for (;;) {
                code = (dtw_readlane(cache, idx) >> sh) & 3u;                         // 2 [best == left] + [best == up]
                const bool isl = code == 2u;
                const int rec_sh = isl ? 64 : sh;
                rv = lane2 == rec_sh ? (unsigned)s : rv;
                const int dgl = (int)(~code & 1u);
                s -= dgl; idx += dgl;
                sh += isl ? 0 : 2;
                if (((2 * (DR - 1) - sh) | (s - s_lim)) < 0) break;      // left the column upward | the path ended or the cache is exhausted
            }
            if (s == 0) {                                                 // the path ends here; after a left move the current row ends at column 1
                if (code == 2u) { rv = lane2 == sh ? 1u : rv; sh_out = sh + 2; } else sh_out = sh;
                break;
            }
            if (sh > 2 * (DR - 1)) { sh_out = sh; break; }
            const int tt = c_top - idx;                                   // a long horizontal run: the register cache is used up
            if (tt < t_lo) { sh_out = sh; break; }                        // .. and so is the LDS window: back out, the column is re-entered behind a reload
            c_top = tt; cache = column(l, tt); idx = 0; idx_max = c_top - t_lo < 63 ? c_top - t_lo : 63;
        }
        p = pbase + (DR - 1) - (sh >> 1);                                 // the row the walk stands in now (the row above the column for sh = 8)
        if (w == 0 && lane < DR && lane2 >= sh_in && lane2 < sh_out) res[pbase + (DR - 1) - lane] = (int)rv;
    }
}

static int dtw_launch(const long long* desc_dev, int n, int max_n, int max_m, void* ws, int* results, void* stream, const float* costs)
{
    static const int dbg = getenv("SS_DTW_DEBUG") ? atoi(getenv("SS_DTW_DEBUG")) : 0;
    if (costs) {
        const long long nrt = dtw_strips(max_n) * DW * (256 / SKR), ntiles = (dtw_tsteps(max_m) + SKT - 1) / SKT;
        long long blocks = nrt * ntiles;
        if (blocks < 1) blocks = 1;                                       // results are zeroed by this launch
        SS_CHECK(blocks < (1LL << 31), "ss_dtw_align: matrix too large (%d x %d)", max_n, max_m);
        SS_LAUNCH(dtw_skew_kernel, dim3((unsigned)blocks, n), dim3(256), 0, stream, costs, desc_dev, (unsigned char*)ws, results, (int)(ntiles > 0 ? ntiles : 1), dbg);
        SS_LAUNCH_CHECK("ss_dtw_align(skew)");
    }
    const size_t smem = (size_t)DCH * 64 + CR_PAD;        // backtrace window; during the sweep of in-place sources: pad + 4 cost rings
    static bool granted = false;
    if (!granted) {
        if (!ss_grant_lds((const void*)dtw_kernel<false>, smem) || !ss_grant_lds((const void*)dtw_kernel<true>, smem)) { ss_set_error("ss_dtw_align: cannot reserve %zu bytes of LDS", smem); return 1; }
        granted = true;
    }
    if (costs) SS_LAUNCH(SS_KERNEL(dtw_kernel<true>), dim3(n), dim3(256), smem, stream, desc_dev, (unsigned char*)ws, results, dbg, costs);
    else SS_LAUNCH(SS_KERNEL(dtw_kernel<false>), dim3(n), dim3(256), smem, stream, desc_dev, (unsigned char*)ws, results, dbg, costs);
    SS_LAUNCH_CHECK("ss_dtw_align");
    return 0;
}

extern "C" int ss_dtw_align(const float* costs, const int64_t* desc_dev, int n, int max_n, int max_m, void* workspace, int32_t* results, void* stream)
{
    SS_CHECK(n >= 0, "ss_dtw_align: negative batch");
    if (n == 0) return 0;
    SS_CHECK(costs && desc_dev && workspace && results, "ss_dtw_align: null pointer");
    SS_CHECK(max_n >= 1 && max_m >= 1, "ss_dtw_align: empty matrix (N=%d, M=%d); the reference indexes shape[0]-1", max_n, max_m);
    return dtw_launch((const long long*)desc_dev, n, max_n, max_m, workspace, results, stream, costs);
}

extern "C" int ss_dtw_align_skewed(const int64_t* desc_dev, int n, void* workspace, int32_t* results, void* stream)
{
    SS_CHECK(n >= 0, "ss_dtw_align_skewed: negative batch");
    if (n == 0) return 0;
    SS_CHECK(desc_dev && workspace && results, "ss_dtw_align_skewed: null pointer");
    return dtw_launch((const long long*)desc_dev, n, 0, 0, workspace, results, stream, nullptr);
}

// ================================================================ dense cumulative matrix: align.py:5-14 `time_warp`
// The batched path above never writes the cumulative matrix (2-bit directions are all a backtrace needs).  Callers of the
// reference's public `time_warp` want the matrix itself (its last entry is the alignment cost), and `zeros_like(costs)` keeps the
// input dtype (align.py:6: f32 from the torch path, f64 stays f64), so this entry point is templated on the element type.  Not on
// the training path: one workgroup sweeps the anti-diagonals (cells of one diagonal are independent), a barrier per diagonal.
//   dtw[0][0] = 0, dtw[0][1:] = dtw[1:][0] = +inf  (align.py:7-9: row 0 / column 0 of `costs` are never read)
//   dtw[i][j] = costs[i][j] + min(dtw[i-1][j], dtw[i][j-1], dtw[i-1][j-1])                       (align.py:11-13)
// With `results` != NULL the same launch walks the matrix back (align.py:21-26: first minimum of up, left, diag).
template <class T>
__global__ __launch_bounds__(1024) void dtw_cumulative_kernel(const T* __restrict__ costs, long long s0, long long s1, T* out, int N, int M, int* __restrict__ results)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const T inf = (T)INFINITY;
    for (int j = tid; j < M; j += nt) out[j] = j ? inf : (T)0;
    for (int i = 1 + tid; i < N; i += nt) out[(long long)i * M] = inf;
    if (results) for (int i = tid; i < N; i += nt) results[i] = 0;
    __syncthreads();
    for (int d = 2; d <= N + M - 2; ++d) {
        const int lo = d - (M - 1) > 1 ? d - (M - 1) : 1, hi = d - 1 < N - 1 ? d - 1 : N - 1;
        for (int i = lo + tid; i <= hi; i += nt) {
            const int j = d - i;
            const T* up = out + (long long)(i - 1) * M + j;
            const T a = up[0], b = out[(long long)i * M + j - 1], c = up[-1];
            T best = a <= b ? a : b;
            best = best <= c ? best : c;
            out[(long long)i * M + j] = costs[i * s0 + j * s1] + best;
        }
        __syncthreads();
    }
    if (results && tid == 0) {
        int i = N - 1, j = M - 1;
        while (i > 0 && j > 0) {
            results[i] = j;
            const T up = out[(long long)(i - 1) * M + j], left = out[(long long)i * M + j - 1], diag = out[(long long)(i - 1) * M + j - 1];
            if (up <= left && up <= diag) --i;
            else if (left <= diag) --j;
            else { --i; --j; }
        }
    }
}

extern "C" int ss_dtw_cumulative(int dtype, const void* costs, int64_t stride_i, int64_t stride_j, int N, int M, void* dtw_out, int32_t* results, void* stream)
{
    SS_CHECK(dtype == SS_F32 || dtype == SS_F64, "ss_dtw_cumulative: dtype must be f32 or f64 (align.py:6 keeps the input dtype)");
    SS_CHECK(costs && dtw_out, "ss_dtw_cumulative: null pointer");
    SS_CHECK(N >= 1 && M >= 1, "ss_dtw_cumulative: empty matrix (N=%d, M=%d)", N, M);
    const int threads = N < 64 ? 64 : (N > 1024 ? 1024 : (N + 63) / 64 * 64);
    if (dtype == SS_F32) SS_LAUNCH(SS_KERNEL(dtw_cumulative_kernel<float>), dim3(1), dim3(threads), 0, stream, (const float*)costs, (long long)stride_i, (long long)stride_j, (float*)dtw_out, N, M, (int*)results);
    else SS_LAUNCH(SS_KERNEL(dtw_cumulative_kernel<double>), dim3(1), dim3(threads), 0, stream, (const double*)costs, (long long)stride_i, (long long)stride_j, (double*)dtw_out, N, M, (int*)results);
    SS_LAUNCH_CHECK("ss_dtw_cumulative");
    return 0;
}
