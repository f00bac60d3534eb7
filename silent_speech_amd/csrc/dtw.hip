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
//   * costs are consumed in a SKEWED layout  sk[(strip*4+wave)][t][lane][r]  so that every step of a wave
//     is one fully coalesced, 16-byte-aligned 1 KiB load (prefetched 8 steps ahead); out-of-matrix cells
//     hold +inf.  ss_dtw_align() builds it from an arbitrarily strided cost matrix (e.g. the non-contiguous
//     costs.T view of transduction_model.py:126); the fused loss path writes it directly (loss.hip).
//   * the 2-bit first-minimum direction of every cell (1 byte per lane per step) goes to HBM instead of the
//     4-byte cumulative matrix (8 B/cell algorithmic traffic -> 4.25 B/cell); wave 0 then walks the path
//     back through LDS-staged direction chunks and writes results[].
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>

#if defined(SS_EMU)
inline void __threadfence() {}
#endif

namespace {
constexpr int DW = 4;          // waves per matrix
constexpr int DR = 4;          // rows per lane
constexpr int DG = 64;         // steps per super-step (barrier interval)
constexpr int DRING = 256;     // LDS boundary ring (columns)
constexpr int DCH = 512;       // backtrace chunk (steps)
constexpr int DESC = 10;       // descriptor fields per matrix
enum { D_N = 0, D_M, D_COST_OFF, D_SI, D_SJ, D_SK_OFF, D_DIRS_OFF, D_BND_OFF, D_RES_OFF };
}

__host__ __device__ static inline long long dtw_strips(long long n) { long long rows = n - 1; long long cap = DW * 64 * DR; return rows <= 0 ? 0 : (rows + cap - 1) / cap; }
__host__ __device__ static inline long long dtw_tsteps(long long m) { return m <= 1 ? 0 : (m - 1) + 63; }

extern "C" int64_t ss_dtw_workspace_bytes(int n, int m, int64_t* sk_bytes, int64_t* dirs_bytes, int64_t* bnd_bytes)
{
    long long strips = dtw_strips(n), ts = dtw_tsteps(m);
    long long sk = strips * DW * ts * 64 * DR * 4, dirs = strips * ts * 256, bnd = 2LL * (m > 0 ? m : 0) * 4;
    sk = (sk + 255) / 256 * 256; dirs = (dirs + 255) / 256 * 256; bnd = (bnd + 255) / 256 * 256;
    if (sk_bytes) *sk_bytes = sk;
    if (dirs_bytes) *dirs_bytes = dirs;
    if (bnd_bytes) *bnd_bytes = bnd;
    return sk + dirs + bnd;
}

// ------------------------------------------------------------------ cost matrix -> skewed strips
__global__ void dtw_skew_kernel(const float* __restrict__ costs, const long long* __restrict__ desc, unsigned char* __restrict__ ws,
                                int* __restrict__ results)
{
    const long long* d = desc + (long long)blockIdx.y * DESC;
    const int N = (int)d[D_N], M = (int)d[D_M];
    int* res = results + d[D_RES_OFF];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) res[i] = 0;
    const long long ts = dtw_tsteps(M), total = dtw_strips(N) * DW * ts * 64 * DR;
    float* sk = (float*)(ws + d[D_SK_OFF]);
    const float* c = costs + d[D_COST_OFF];
    const long long si = d[D_SI], sj = d[D_SJ];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int r = (int)(e % DR); long long x = e / DR; int l = (int)(x % 64); x /= 64; long long t = x % ts; long long kw = x / ts;
        long long i = 1 + (kw * 64 + l) * DR + r, j = t + 1 - l;
        sk[e] = (i < N && j >= 1 && j < M) ? c[i * si + j * sj] : INFINITY;
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

// One wavefront step.  No column predicate: cells outside the matrix carry cost +inf in the skewed layout, so a lane that
// has not reached column 1 yet (or is past M-1) only turns +inf into +inf and its state stays what the recurrence needs
// (dtw[i][0] = dtw[0][j] = +inf); the direction bytes it writes there are never read.  Branch-free: the per-step chain is
// the 4 dependent (min3, add) pairs plus one DPP shift.
__device__ __forceinline__ void dtw_step(int t, int lane, int w, bool multi_strip, int M, const f32x4& cv, float top, float (&prev)[DR], float& diag_sv, float& last_out,
                                         unsigned char*& dp, float* ring_next, float* dump, float* bnd_cur)
{
    const int s = t + 1 - lane;
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
    *dp = (unsigned char)bits;
    dp += 256;
    // hand the strip's last row to the next wave through the LDS ring: only lane 63 owns a ring slot, the others hit a dump word
    float* slot = lane == 63 ? ring_next + (s & (DRING - 1)) : dump;
    *slot = a;
    if (multi_strip && w == DW - 1 && lane == 63 && s >= 1 && s < M) bnd_cur[s] = a;
}

__global__ __launch_bounds__(256) void dtw_kernel(const long long* __restrict__ desc, unsigned char* __restrict__ ws, int* __restrict__ results)
{
    __shared__ float lds_bnd[DW + 1][DRING];              // [w] = ring read by wave w (written by wave w-1); [DW] = dump / last wave's unused ring
    __shared__ __attribute__((aligned(16))) unsigned char chunk[DCH * 64];
    const long long* d = desc + (long long)blockIdx.x * DESC;
    const int N = (int)d[D_N], M = (int)d[D_M];
    if (N <= 1 || M <= 1) return;                                   // no interior cell: results stay 0 (align.py:24)
    const float* sk = (const float*)(ws + d[D_SK_OFF]);
    unsigned char* dirs = ws + d[D_DIRS_OFF];
    float* bnd = (float*)(ws + d[D_BND_OFF]);
    int* res = results + d[D_RES_OFF];
#if defined(SS_EMU)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#else
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: step bounds become s_cbranch, not exec masks
#endif
    const int ts = (int)dtw_tsteps(M), nstrips = (int)dtw_strips(N);
    const int nss = (ts + DG - 1) / DG;

    for (int k = 0; k < nstrips; ++k) {
        const int rowbase = ((k * DW + w) * 64 + lane) * DR;
        float prev[DR];
#pragma unroll
        for (int r = 0; r < DR; ++r) prev[r] = INFINITY;             // dtw[i][0] = inf
        float diag_sv = rowbase == 0 ? 0.f : INFINITY;               // dtw[i-1][0]; dtw[0][0] = 0
        float last_out = INFINITY;
        const float* skp = sk + ((long long)(k * DW + w) * ts) * (64 * DR) + lane * DR;
        unsigned char* dp0 = dirs + ((long long)k * ts) * 256 + w * 64 + lane;
        const float* bnd_prev = bnd + ((k + 1) & 1) * M;
        float* bnd_cur = bnd + (k & 1) * M;
        for (int ss = 0; ss < nss + 2 * (DW - 1); ++ss) {
            const int u = ss - 2 * w;
            if (u >= 0 && u < nss) {
                const int t0 = u * DG;
                unsigned char* dp = dp0 + (long long)t0 * 256;
                // the 64 values lane 0 will need from above during this super-step (one per step), fetched up front
                float topv;
                { const int sb = t0 + 1 + lane;
                  if (w == 0) topv = (k == 0 || sb >= M) ? INFINITY : bnd_prev[sb];
                  else topv = lds_bnd[w][sb & (DRING - 1)]; }
#if !defined(SS_EMU)
                // settle the LDS read NOW: left pending, the compiler re-waits lgkmcnt(0) at the v_readlane of EVERY step, which also
                // drains that step's ring write (an LDS round trip on the serial chain of each of the 64 steps)
                __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
                f32x4 cb[8];
                const f32x4 inf4 = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll
                for (int e = 0; e < 8; ++e) cb[e] = t0 + e < ts ? *(const f32x4*)(skp + (long long)(t0 + e) * (64 * DR)) : inf4;
                for (int g = 0; g < DG / 8; ++g) {
                    f32x4 nb[8];
                    const int tn = t0 + (g + 1) * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) nb[e] = (g + 1 < DG / 8 && tn + e < ts) ? *(const f32x4*)(skp + (long long)(tn + e) * (64 * DR)) : inf4;
#if !defined(SS_EMU)
                    // settle THIS group's costs (requested one group ago) now, leaving the 8 loads just issued in flight: every step is its own
                    // basic block, and left to itself the compiler waits vmcnt(0) at the top of each -- i.e. for the previous step's direction
                    // store (vmcnt counts stores too): one memory round trip on the serial chain of every step
                    __builtin_amdgcn_s_waitcnt(0x0f78);          // vmcnt(8), expcnt / lgkmcnt untouched
#endif
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int t = t0 + g * 8 + e;
                        if (t < ts) { dtw_step(t, lane, w, nstrips > 1, M, cb[e], topv, prev, diag_sv, last_out, dp, lds_bnd[w + 1], &lds_bnd[DW][lane], bnd_cur); topv = wave_rotate_down(topv); }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) cb[e] = nb[e];
                }
            }
            __syncthreads();
        }
        __threadfence();      // strip boundary row + direction bytes visible before they are re-read
        __syncthreads();
    }

    if (w != 0) return;
    // ---- backtrace (wave 0, wave-uniform walk; direction bytes staged through LDS in chunks)
    int p = N - 1, s = M - 1;
    int cur_kw = -1, t_lo = 0, t_hi = -1;
    while (p > 0 && s > 0) {
        if (lane == 0) res[p] = s;
        const int q = p - 1;
        const int kw = q / (64 * DR), l = (q / DR) & 63, r = q % DR;
        const int t = s - 1 + l;
        if (kw != cur_kw || t < t_lo || t > t_hi) {
            __syncthreads();
            t_hi = t; t_lo = t - DCH + 1 < 0 ? 0 : t - DCH + 1; cur_kw = kw;
            const int kk = kw / DW, ww = kw % DW;
            const unsigned char* src = dirs + ((long long)kk * ts + t_lo) * 256 + ww * 64;
            const int nd = (t_hi - t_lo + 1) * 16;
            for (int idx = lane; idx < nd; idx += 64) {
                const int tt = idx >> 4, c = idx & 15;
                *(unsigned*)(chunk + tt * 64 + c * 4) = *(const unsigned*)(src + (long long)tt * 256 + c * 4);
            }
            __syncthreads();
        }
        const unsigned byte = chunk[(t - t_lo) * 64 + l];
        const unsigned pm = (byte >> (2 * (DR - 1 - r))) & 3u;            // 2 [best == left] + [best == up]
        if (pm & 1u) { --p; } else if (pm & 2u) { --s; } else { --p; --s; }
    }
}

static int dtw_launch(const long long* desc_dev, int n, int max_n, int max_m, void* ws, int* results, void* stream, const float* costs)
{
    if (costs) {
        long long total = dtw_strips(max_n) * DW * dtw_tsteps(max_m) * 64 * DR;
        long long blocks = (total + 255) / 256;
        if (blocks < (max_n + 255) / 256) blocks = (max_n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        SS_LAUNCH(dtw_skew_kernel, dim3((unsigned)blocks, n), dim3(256), 0, stream, costs, desc_dev, (unsigned char*)ws, results);
        SS_LAUNCH_CHECK("ss_dtw_align(skew)");
    }
    SS_LAUNCH(dtw_kernel, dim3(n), dim3(256), 0, stream, desc_dev, (unsigned char*)ws, results);
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
