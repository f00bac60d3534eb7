// norm.hip -- the HBM-bound normalisation / elementwise kernels of the transduction hot path (gfx950):
//   BatchNorm1d training/eval forward + backward, fused with ReLU and the residual add of ResBlock
//   (architecture.py:19,21,25,32,36,40), LayerNorm fused with the residual add and dropout of the
//   post-norm encoder layer (transformer.py:55-56,58-59), bias-gradient column sums, and the EMG
//   input conditioning (shift augmentation architecture.py:64-68 + cast + zero padding).
// Activations are (B, T[+2], C) row-major with C contiguous; every access is a 16-byte vector of 8
// (bf16) / 2x16 bytes (f32) channels, statistics are f32.  Padded buffers carry one zero row on each
// side of every sequence so the k=3 convolutions read their taps as one contiguous 3C row (gemm.hip).
#include "common.h"
#include <stdlib.h>
#include "silent_speech_hip.h"

namespace {
struct Seq {            // logical row r of a (B, T + 2*pad, C) buffer
    int T, pad;
    __device__ __forceinline__ long long row(int r) const { int b = r / T, t = r - b * T; return (long long)b * (T + 2 * pad) + pad + t; }
    __device__ __forceinline__ long long at(int b, int t) const { return (long long)b * (T + 2 * pad) + pad + t; }     // the division of row() done once by the caller
};

// 8 consecutive per-channel f32 constants as TWO 16-byte loads.  Element-wise (and conditional) they became 8 dword loads each behind its own
// wait: with 6..8 such arrays a workgroup spent its first ~8 us in a chain of dependent round trips before touching the data.
__device__ __forceinline__ void ld8(const float* __restrict__ p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// thread -> (vector column cx0 (+k*CVb), row lane ry) for column reductions over a [rows][C] matrix
struct ColMap {
    int CV, CVb, RY, cx0, ry; bool active;
    __device__ __forceinline__ ColMap(int C, int tid, int nthreads) {
        CV = C >> 3; CVb = CV < nthreads ? CV : nthreads; RY = nthreads / CVb;
        cx0 = tid % CVb; ry = tid / CVb; active = ry < RY;
    }
};
constexpr int RED_THREADS = 256;

// sum over the per-chunk partials of NQ per-channel quantities: 256 threads = 32 channels x 8 chunk lanes
template <int NQ>
__device__ __forceinline__ bool chunk_reduce(const float* __restrict__ partial, int nchunks, int C, float (&out)[NQ], int& c_out)
{
    __shared__ float red[NQ][8][32];
    const int cl = threadIdx.x & 31, ln = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    if (c < C) {
        int k = ln;
        for (; k + 56 < nchunks; k += 64) {                 // 8 independent loads per quantity in flight (the serial loop was latency-bound)
            float v[8][NQ];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) v[u][q] = partial[((long long)(k + 8 * u) * NQ + q) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] += v[u][q];
        }
        for (; k < nchunks; k += 8)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] += partial[((long long)k * NQ + q) * C + c];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[q][ln][cl] = acc[q];
    __syncthreads();
    c_out = c;
    if (ln != 0 || c >= C) return false;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { float t = 0.f; for (int y = 0; y < 8; ++y) t += red[q][y][cl]; out[q] = t; }
    return true;
}
}

// =========================================================================== BatchNorm statistics
template <class T>
__global__ __launch_bounds__(RED_THREADS) void bn_partial_kernel(const T* __restrict__ x, Seq sx, int rows, int C, int rows_per_chunk, const float* __restrict__ shift, float* __restrict__ partial)
{
    __shared__ float red[2][RED_THREADS * 8];
    ColMap m(C, threadIdx.x, (int)blockDim.x);
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    for (int cb = 0; cb < m.CV; cb += m.CVb) {
        const int cx = cb + m.cx0;
        float s1[8], s2[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; sh[e] = 0.f; }
        const bool cv = m.active && cx < m.CV;
        if (cv) {
            if (shift) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sh[e] = shift[cx * 8 + e];
            } else Vec8<T>::load(x + sx.row(0) * C + cx * 8, sh);   // shift = first row: tames E[x^2]-E[x]^2 cancellation
            int r = r0 + m.ry;
            for (; r + 3 * m.RY < r1; r += 4 * m.RY) {                    // 4 independent 16-byte loads in flight per thread
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Vec8<T>::load(x + sx.row(r + u * m.RY) * C + cx * 8, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float d = v[u][e] - sh[e]; s1[e] += d; s2[e] += d * d; }
            }
            for (; r < r1; r += m.RY) {
                float v[8]; Vec8<T>::load(x + sx.row(r) * C + cx * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { float d = v[e] - sh[e]; s1[e] += d; s2[e] += d * d; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = s1[e]; red[1][threadIdx.x * 8 + e] = s2[e]; }
        __syncthreads();
        if (cv && m.ry == 0) {
            for (int y = 1; y < m.RY; ++y)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += red[0][(y * m.CVb + m.cx0) * 8 + e]; s2[e] += red[1][(y * m.CVb + m.cx0) * 8 + e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                partial[((long long)blockIdx.x * 2 + 0) * C + cx * 8 + e] = s1[e];
                partial[((long long)blockIdx.x * 2 + 1) * C + cx * 8 + e] = s2[e];
            }
        }
    }
}

// sums[0][c] = sum(x - shift), sums[1][c] = sum (x - shift)^2, sums[2][c] = shift
template <class T>
__global__ __launch_bounds__(256) void bn_sums_kernel(const T* __restrict__ x, Seq sx, const float* __restrict__ partial, int nchunks, int C, const float* __restrict__ shift, float* __restrict__ sums)
{
    float t[2]; int c;
    if (!chunk_reduce<2>(partial, nchunks, C, t, c)) return;
    sums[c] = t[0]; sums[C + c] = t[1]; sums[2 * C + c] = shift ? shift[c] : ldf(x + sx.row(0) * C + c);
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* shift, float n, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                                   float momentum, float eps, int training)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!training) { mean[c] = running_mean[c]; invstd[c] = rsqrtf(running_var[c] + eps); return; }
    const float d = sums[c] / n, mu = (shift ? shift[c] : sums[2 * C + c]) + d;      // shift may alias running_mean: read before the update below
    float var = sums[C + c] / n - d * d; var = var < 0.f ? 0.f : var;
    mean[c] = mu; invstd[c] = rsqrtf(var + eps);
    if (running_mean) {   // torch: running = (1-m)*running + m*stat, unbiased variance for the running estimate
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
    }
}

// threads of a column-reduction workgroup: the largest multiple of the C / 8 column chunks within RED_THREADS, so that no lane idles (C = 768:
// 96 chunks x 2 rows = 192 threads; with 256 a quarter of the lanes had no column).  SS_BN_RED_THREADS overrides (tuning).
static int red_threads(int C) {
    static const int forced = getenv("SS_BN_RED_THREADS") ? atoi(getenv("SS_BN_RED_THREADS")) : 0;
    if (forced > 0) return forced < RED_THREADS ? forced : RED_THREADS;
    const int cv = C >> 3;
    if (cv >= RED_THREADS || cv <= 0) return RED_THREADS;
    const int t = RED_THREADS / cv * cv;
    return t >= 64 ? t : RED_THREADS;
}
// row chunks of the column reductions = workgroups: 3 per CU (SS_BN_CHUNKS; bn_bwd_sums per step: 512 -> 0.504, 768 -> 0.475, 1024 -> 0.53 ms)
static int red_chunks(int rows) { static const int cap = getenv("SS_BN_CHUNKS") ? atoi(getenv("SS_BN_CHUNKS")) : 768; int c = (rows + 63) / 64; if (c > cap) c = cap; if (c < 1) c = 1; return c; }

extern "C" int64_t ss_bn_scratch_floats(int B, int T, int C) { return (int64_t)red_chunks(B * T) * 4 * C + 8 * (int64_t)C; }

extern "C" int ss_bn_stats_sums(int dtype, const void* x, int B, int T, int C, int pad, float* scratch, const float* shift, float* sums, void* stream)
{
    SS_CHECK(x && scratch && sums, "ss_bn_stats_sums: null pointer");
    SS_CHECK(C % 8 == 0 && C > 0, "ss_bn_stats_sums: C=%d must be a positive multiple of 8", C);
    SS_CHECK(B > 0 && T > 0, "ss_bn_stats_sums: empty batch");
    const int rows = B * T, nch = red_chunks(rows), rpc = (rows + nch - 1) / nch;
    Seq sx = {T, pad};
    if (dtype == SS_BF16) {
        SS_LAUNCH(bn_partial_kernel<bf16_t>, dim3(nch), dim3(red_threads(C)), 0, stream, (const bf16_t*)x, sx, rows, C, rpc, shift, scratch);
        SS_LAUNCH(bn_sums_kernel<bf16_t>, dim3((C + 31) / 32), dim3(256), 0, stream, (const bf16_t*)x, sx, (const float*)scratch, nch, C, shift, sums);
    } else {
        SS_LAUNCH(bn_partial_kernel<float>, dim3(nch), dim3(red_threads(C)), 0, stream, (const float*)x, sx, rows, C, rpc, shift, scratch);
        SS_LAUNCH(bn_sums_kernel<float>, dim3((C + 31) / 32), dim3(256), 0, stream, (const float*)x, sx, (const float*)scratch, nch, C, shift, sums);
    }
    SS_LAUNCH_CHECK("ss_bn_stats_sums");
    return 0;
}

extern "C" int ss_bn_finalize(const float* sums, double n_total, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                              float momentum, float eps, int training, void* stream)
{
    SS_CHECK(mean && invstd && C > 0, "ss_bn_finalize: null pointer");
    SS_CHECK(training ? (sums != nullptr && n_total >= 1.0) : (running_mean && running_var), "ss_bn_finalize: missing sums (training) or running statistics (eval)");
    SS_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, (const float*)nullptr, (float)n_total, C, mean, invstd, running_mean, running_var, momentum, eps, training);
    SS_LAUNCH_CHECK("ss_bn_finalize");
    return 0;
}
extern "C" int ss_bn_finalize_shift(const float* sums, const float* shift, double n_total, int C, float* mean, float* invstd, float* running_mean, float* running_var,
                                    float momentum, float eps, int training, void* stream)
{
    SS_CHECK(mean && invstd && C > 0, "ss_bn_finalize_shift: null pointer");
    SS_CHECK(training ? (sums != nullptr && shift != nullptr && n_total >= 1.0) : (running_mean && running_var), "ss_bn_finalize_shift: missing sums / shift (training) or running statistics (eval)");
    SS_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sums, shift, (float)n_total, C, mean, invstd, running_mean, running_var, momentum, eps, training);
    SS_LAUNCH_CHECK("ss_bn_finalize_shift");
    return 0;
}

// =========================================================================== BatchNorm apply (+residual, +ReLU)
// Elementwise kernels over (B, T + 2 pad, C) in 16-byte chunks.  The launch makes the grid stride a multiple of C / 8 whenever it can
// (ew_grid), so that a thread stays on ONE column chunk: the per-channel constants then sit in registers for all of its rows, and
// (b, t) advance without a division.  (Re-read per chunk they were 128 .. 192 bytes of L1 traffic per 16 bytes of data, and the 64-bit
// row arithmetic -- a division per tensor and chunk -- outweighed the memory instructions.)
struct RowWalk {
    unsigned ro, b, tp, rstep, db, dtp, TP; int cx;
    __device__ __forceinline__ RowWalk(unsigned gtid, unsigned step, unsigned CV, unsigned TP_) {
        TP = TP_; ro = gtid / CV; cx = (int)(gtid - ro * CV); rstep = step / CV;
        b = ro / TP; tp = ro - b * TP; db = rstep / TP; dtp = rstep - db * TP;
    }
    __device__ __forceinline__ void next() { ro += rstep; b += db; tp += dtp; if (tp >= TP) { tp -= TP; ++b; } }
};

template <class T>
__global__ void bn_apply_kernel(const T* __restrict__ xa, Seq sa, const float* __restrict__ mean_a, const float* __restrict__ invstd_a, const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                                const T* __restrict__ xb, Seq sb, const float* __restrict__ mean_b, const float* __restrict__ invstd_b, const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                T* __restrict__ y, Seq sy, int B, int C, int relu)
{
    const int CV = C >> 3, TT = sy.T;
    const unsigned TP = (unsigned)(TT + 2 * sy.pad), rows_out = (unsigned)B * TP, step32 = gridDim.x * blockDim.x;     // the host checks rows_out * CV < 2^31
    if (step32 % (unsigned)CV == 0) {
        RowWalk w(blockIdx.x * blockDim.x + threadIdx.x, step32, (unsigned)CV, TP);
        float ma[8], sca[8], ba[8], mb[8], scb[8], bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mb[e] = 0.f; scb[e] = 0.f; bb[e] = 0.f; }
        {
            float t[8];
            ld8(mean_a + w.cx * 8, ma); ld8(gamma_a + w.cx * 8, sca); ld8(invstd_a + w.cx * 8, t); ld8(beta_a + w.cx * 8, ba);
#pragma unroll
            for (int e = 0; e < 8; ++e) sca[e] *= t[e];
            if (xb) {
                ld8(mean_b + w.cx * 8, mb); ld8(gamma_b + w.cx * 8, scb); ld8(invstd_b + w.cx * 8, t); ld8(beta_b + w.cx * 8, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) scb[e] *= t[e];
            }
        }
        // the raw chunks of the thread's NEXT row are requested before the current row is converted and stored (one row per trip otherwise means
        // load -> wait -> store with a single 16-byte load or two in flight per thread)
        typedef typename RawVec8<T>::type Raw;
        Raw nv = RawVec8<T>::zero(), nu = RawVec8<T>::zero();
        auto fetch = [&](unsigned b, unsigned tp) {
            const int t = (int)tp - sy.pad;
            if (t >= 0 && t < TT) {
                nv = RawVec8<T>::load(xa + sa.at((int)b, t) * C + w.cx * 8);
                if (xb) nu = RawVec8<T>::load(xb + sb.at((int)b, t) * C + w.cx * 8);
            }
        };
        if (w.ro < rows_out) fetch(w.b, w.tp);
        for (; w.ro < rows_out; w.next()) {
            const int t = (int)w.tp - sy.pad;
            const Raw cv = nv, cu = nu;
            {   // position of the next row of this thread (RowWalk::next without committing it)
                unsigned nb = w.b + w.db, ntp = w.tp + w.dtp;
                if (ntp >= w.TP) { ntp -= w.TP; ++nb; }
                if (w.ro + w.rstep < rows_out) fetch(nb, ntp);
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;                 // zero halo rows
            if (t >= 0 && t < TT) {
                float v[8], u[8];
                RawVec8<T>::unpack(cv, v);
                if (xb) RawVec8<T>::unpack(cu, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (v[e] - ma[e]) * sca[e] + ba[e];
                if (xb) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += (u[e] - mb[e]) * scb[e] + bb[e];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
                }
            }
            Vec8<T>::store(y + (long long)w.ro * C + w.cx * 8, o);
        }
        return;
    }
    const unsigned total32 = rows_out * (unsigned)CV;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += step32) {
        const unsigned ro = i / (unsigned)CV; const int cx = (int)(i - ro * (unsigned)CV);
        const unsigned bq = ro / TP; const int b = (int)bq, tp = (int)(ro - bq * TP), t = tp - sy.pad;
        float o[8];
        if (t < 0 || t >= TT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;             // zero halo rows
        } else {
            float v[8]; Vec8<T>::load(xa + sa.at(b, t) * C + cx * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int c = cx * 8 + e; const float s = gamma_a[c] * invstd_a[c]; o[e] = (v[e] - mean_a[c]) * s + beta_a[c]; }
            if (xb) {
                float u[8]; Vec8<T>::load(xb + sb.at(b, t) * C + cx * 8, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int c = cx * 8 + e; const float s = gamma_b[c] * invstd_b[c]; o[e] += (u[e] - mean_b[c]) * s + beta_b[c]; }
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
            }
        }
        Vec8<T>::store(y + (long long)ro * C + cx * 8, o);
    }
}

// grid of an elementwise kernel over `total` 16-byte chunks.  CV > 0: rows of CV chunks with per-column constants -- few, long-lived
// workgroups (768 = 3 per CU; SS_BN_GRID) whose thread total is a multiple of CV, so that a thread keeps its column chunk and its
// constants.  Measured on the bench shapes (bn_apply / bn_bwd_apply per step): 8190 workgroups 1.10 / 1.33 ms (the set-up -- four
// divisions, 6..14 constant loads -- is paid per 1..4 chunks), 2046: 0.45 / 0.62, 768: 0.335 / 0.50, 510: 0.34 / 0.53; the per-chunk form
// this replaces: 0.39 / 0.56.  Two rows in flight per thread: no gain (bn_apply), spills (bn_bwd_apply).
static dim3 ew_grid(long long total, int block, int CV = 0) {
    static const long long cap_cols = getenv("SS_BN_GRID") ? atoll(getenv("SS_BN_GRID")) : 768;
    long long g = (total + block - 1) / block;
    const long long cap = CV > 0 ? cap_cols : 8192;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if (CV > 0) {
        long long a = CV, bb = block; while (bb) { const long long t = a % bb; a = bb; bb = t; }      // gcd(CV, block)
        const long long m = CV / a;
        if (g >= 4 * m) g = g / m * m;
    }
    return dim3((unsigned)g);
}

extern "C" int ss_bn_apply(int dtype, const void* xa, const float* mean_a, const float* invstd_a, const float* gamma_a, const float* beta_a, int pad_xa,
                           const void* xb, const float* mean_b, const float* invstd_b, const float* gamma_b, const float* beta_b, int pad_xb,
                           void* y, int pad_y, int B, int T, int C, int relu, void* stream)
{
    SS_CHECK(xa && mean_a && invstd_a && gamma_a && beta_a && y, "ss_bn_apply: null pointer");
    SS_CHECK(!xb || (mean_b && invstd_b && gamma_b && beta_b), "ss_bn_apply: second branch incomplete");
    SS_CHECK(C % 8 == 0 && C > 0 && B > 0 && T > 0, "ss_bn_apply: bad shape B=%d T=%d C=%d", B, T, C);
    Seq sa = {T, pad_xa}, sb = {T, pad_xb}, sy = {T, pad_y};
    const long long total = (long long)B * (T + 2 * pad_y) * (C / 8);
    SS_CHECK(total < (1LL << 31) - (1LL << 22), "ss_bn_apply: tensor too large for 32-bit chunk indices");
    if (dtype == SS_BF16) SS_LAUNCH(bn_apply_kernel<bf16_t>, ew_grid(total, 256, C / 8), dim3(256), 0, stream, (const bf16_t*)xa, sa, mean_a, invstd_a, gamma_a, beta_a, (const bf16_t*)xb, sb, mean_b, invstd_b, gamma_b, beta_b, (bf16_t*)y, sy, B, C, relu);
    else SS_LAUNCH(bn_apply_kernel<float>, ew_grid(total, 256, C / 8), dim3(256), 0, stream, (const float*)xa, sa, mean_a, invstd_a, gamma_a, beta_a, (const float*)xb, sb, mean_b, invstd_b, gamma_b, beta_b, (float*)y, sy, B, C, relu);
    SS_LAUNCH_CHECK("ss_bn_apply");
    return 0;
}

// =========================================================================== BatchNorm backward
// g = dy * 1[y>0]  (ReLU of the fused output);  per branch: dgamma = sum g*xhat, dbeta = sum g,
// dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).
template <class T, bool REGATE>
__global__ __launch_bounds__(RED_THREADS, sizeof(T) == 2 ? 3 : 2) void bn_bwd_partial_kernel(const T* __restrict__ dy, Seq sdy, const T* __restrict__ y, Seq sy,
                                                                     const T* __restrict__ xa, Seq sa, const float* __restrict__ mean_a, const float* __restrict__ invstd_a,
                                                                     const T* __restrict__ xb, Seq sb, const float* __restrict__ mean_b, const float* __restrict__ invstd_b,
                                                                     const float* __restrict__ gamma_a, const float* __restrict__ beta_a, const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
                                                                     int rows, int C, int rows_per_chunk, int relu, float* __restrict__ partial)
{
    // beta_a != null: the ReLU gate is RECOMPUTED from the inputs that are read anyway -- y > 0  <=>  (xa - mean_a) gamma_a invstd_a + beta_a
    // [+ the b branch] > 0, the forward's own expression (bn_apply_kernel) -- instead of read from the saved output: one tensor less per pass
    // (4 -> 3 reads here, 4 -> 3 of the 6 streams of bn_bwd_apply).
    // REGATE is a template parameter: the two forms need different per-channel constants, and both in one kernel overflowed the 128-register
    // budget of 4 workgroups per CU (the pass ran 3 x slower).  The recomputing form keeps (mean, gamma invstd) per branch + the summed betas and
    // applies invstd to the x-hat sums once, after the row loop.
    __shared__ float red[3][RED_THREADS * 8];
    ColMap m(C, threadIdx.x, (int)blockDim.x);
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    for (int cb = 0; cb < m.CV; cb += m.CVb) {
        const int cx = cb + m.cx0;
        float sg[8], sga[8], sgb[8], ma[8], ka[8], mb[8], kb[8], bs[8];          // ka / kb: invstd (gate from y) or gamma invstd (gate recomputed)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg[e] = sga[e] = sgb[e] = 0.f; ma[e] = ka[e] = mb[e] = kb[e] = bs[e] = 0.f; }
        const bool cv = m.active && cx < m.CV;
        if (cv) {
            {   // all per-channel constants as 16-byte loads, issued together (the b-branch arrays fall back to the a-branch ones when absent: no branch)
                float t0[8], t1[8], t2[8], t3[8];
                ld8(mean_a + cx * 8, ma); ld8(invstd_a + cx * 8, ka);
                ld8((xb ? mean_b : mean_a) + cx * 8, mb); ld8((xb ? invstd_b : invstd_a) + cx * 8, kb);
                if (REGATE) {
                    ld8(gamma_a + cx * 8, t0); ld8(beta_a + cx * 8, t1); ld8((xb ? gamma_b : gamma_a) + cx * 8, t2); ld8((xb ? beta_b : beta_a) + cx * 8, t3);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ka[e] *= t0[e]; kb[e] *= t2[e]; bs[e] = t1[e] + (xb ? t3[e] : 0.f); }
                }
                if (!xb) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { mb[e] = 0.f; kb[e] = 0.f; }
                }
            }
            // Two rows per trip, SOFTWARE-PIPELINED: the raw 16-byte chunks of trip k + 1 are requested before trip k is unpacked and summed, so the
            // loads of one trip travel under the ~200 VALU instructions of the previous one.  (Round 3 issued a trip's loads, waited for all of them and
            // computed: ~30 dependent round trips per thread, 2.6 TB/s of real traffic for an "HBM-bound" pass.)  Rows past the chunk re-load its last
            // row and are skipped at the use; the sums keep their row order.
            typedef typename RawVec8<T>::type Raw;
            Raw ng[2], no[2], na[2], nb[2];
            auto fetch = [&](int r, Raw (&fg)[2], Raw (&fo)[2], Raw (&fa)[2], Raw (&fb)[2]) {
                const int rc0 = r < r1 ? r : r1 - 1, rc1 = r + m.RY < r1 ? r + m.RY : r1 - 1;
                const int b0 = rc0 / sdy.T, t0 = rc0 - b0 * sdy.T, b1 = rc1 / sdy.T, t1 = rc1 - b1 * sdy.T;      // ONE division per row (Seq::row costs one per tensor)
                fg[0] = RawVec8<T>::load(dy + sdy.at(b0, t0) * C + cx * 8); fg[1] = RawVec8<T>::load(dy + sdy.at(b1, t1) * C + cx * 8);
                if (relu && !REGATE) { fo[0] = RawVec8<T>::load(y + sy.at(b0, t0) * C + cx * 8); fo[1] = RawVec8<T>::load(y + sy.at(b1, t1) * C + cx * 8); }
                fa[0] = RawVec8<T>::load(xa + sa.at(b0, t0) * C + cx * 8); fa[1] = RawVec8<T>::load(xa + sa.at(b1, t1) * C + cx * 8);
                if (xb) { fb[0] = RawVec8<T>::load(xb + sb.at(b0, t0) * C + cx * 8); fb[1] = RawVec8<T>::load(xb + sb.at(b1, t1) * C + cx * 8); }
            };
            if (r0 + m.ry < r1) fetch(r0 + m.ry, ng, no, na, nb);
            for (int r = r0 + m.ry; r < r1; r += 2 * m.RY) {
                const bool two = r + m.RY < r1;
                Raw cg[2] = {ng[0], ng[1]}, co[2] = {no[0], no[1]}, ca[2] = {na[0], na[1]}, cb[2] = {nb[0], nb[1]};
                if (r + 2 * m.RY < r1) fetch(r + 2 * m.RY, ng, no, na, nb);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
                    float g[8], o[8], va[8], vb[8];
                    RawVec8<T>::unpack(cg[u], g); RawVec8<T>::unpack(ca[u], va);
                    if (relu && !REGATE) RawVec8<T>::unpack(co[u], o);
                    if (xb) RawVec8<T>::unpack(cb[u], vb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (REGATE) {       // pre = (xa - mean_a) gamma_a invstd_a [+ b branch] + betas: the forward's expression (bn_apply_kernel)
                            const float da = va[e] - ma[e], db = xb ? vb[e] - mb[e] : 0.f;
                            const float gg = (!relu || da * ka[e] + db * kb[e] + bs[e] > 0.f) ? g[e] : 0.f;
                            sg[e] += gg; sga[e] += gg * da;
                            if (xb) sgb[e] += gg * db;
                        } else {
                            const float gg = (!relu || o[e] > 0.f) ? g[e] : 0.f;
                            sg[e] += gg; sga[e] += gg * (va[e] - ma[e]) * ka[e];
                            if (xb) sgb[e] += gg * (vb[e] - mb[e]) * kb[e];
                        }
                    }
                }
            }
            if (REGATE) {                   // the x-hat sums were taken without invstd: apply it once per channel (16-byte loads, no branch)
                float ia[8], ib[8];
                ld8(invstd_a + cx * 8, ia); ld8((xb ? invstd_b : invstd_a) + cx * 8, ib);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sga[e] *= ia[e]; sgb[e] *= ib[e]; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][threadIdx.x * 8 + e] = sg[e]; red[1][threadIdx.x * 8 + e] = sga[e]; red[2][threadIdx.x * 8 + e] = sgb[e]; }
        __syncthreads();
        if (cv && m.ry == 0) {
            for (int yy = 1; yy < m.RY; ++yy)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int o = (yy * m.CVb + m.cx0) * 8 + e; sg[e] += red[0][o]; sga[e] += red[1][o]; sgb[e] += red[2][o]; }
            Vec8<float>::store(partial + ((long long)blockIdx.x * 3 + 0) * C + cx * 8, sg);
            Vec8<float>::store(partial + ((long long)blockIdx.x * 3 + 1) * C + cx * 8, sga);
            Vec8<float>::store(partial + ((long long)blockIdx.x * 3 + 2) * C + cx * 8, sgb);
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nchunks, int C, float* __restrict__ coef /* [3][C] sums */,
                                                              float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b)
{
    float t[3]; int c;
    if (!chunk_reduce<3>(partial, nchunks, C, t, c)) return;
    coef[c] = t[0]; coef[C + c] = t[1]; coef[2 * C + c] = t[2];
    if (dgamma_a) dgamma_a[c] += t[1];
    if (dbeta_a) dbeta_a[c] += t[0];
    if (dgamma_b) dgamma_b[c] += t[2];
    if (dbeta_b) dbeta_b[c] += t[0];
}

template <class T, bool REGATE>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, Seq sdy, const T* __restrict__ y, Seq sy,
                                    const T* __restrict__ xa, Seq sa, const float* __restrict__ mean_a, const float* __restrict__ invstd_a, const float* __restrict__ gamma_a,
                                    const T* __restrict__ xb, Seq sb, const float* __restrict__ mean_b, const float* __restrict__ invstd_b, const float* __restrict__ gamma_b,
                                    const float* __restrict__ coef, float inv_n, T* __restrict__ dxa, Seq sda, T* __restrict__ dxb, Seq sdb, int B, int C, int relu,
                                    const float* __restrict__ beta_a, const float* __restrict__ beta_b)
{
    constexpr bool regate = REGATE;                         // recompute the ReLU gate from xa / xb instead of reading the saved output (see bn_bwd_partial_kernel;
                                                            // a template parameter for the same reason: both forms in one kernel spilled 60 registers)
    const int CV = C >> 3, TT = sdy.T;
    const int padmax = sda.pad > sdb.pad ? sda.pad : sdb.pad;
    const long long total = (long long)B * (TT + 2 * padmax) * CV;
    const unsigned TP = (unsigned)(TT + 2 * padmax), total32 = (unsigned)total, step32 = gridDim.x * blockDim.x;      // 32-bit index arithmetic, see bn_apply_kernel
    if (step32 % (unsigned)CV == 0) {
        // thread <-> column chunk fixed (see bn_apply_kernel): dx = k (g - c0 - (x - mean) q) with k = gamma invstd, c0 = mean(g),
        // q = invstd mean(g xhat) in registers
        RowWalk w(blockIdx.x * blockDim.x + threadIdx.x, step32, (unsigned)CV, TP);
        float ma[8], ka[8], qa[8], c0[8], mb[8], kb[8], qb[8], bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mb[e] = 0.f; kb[e] = 0.f; qb[e] = 0.f; bs[e] = 0.f; }
        {
            float ia[8], t[8];
            ld8(mean_a + w.cx * 8, ma); ld8(gamma_a + w.cx * 8, ka); ld8(invstd_a + w.cx * 8, ia); ld8(coef + C + w.cx * 8, qa); ld8(coef + w.cx * 8, c0);
            if (regate) ld8(beta_a + w.cx * 8, bs);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ka[e] *= ia[e]; qa[e] = ia[e] * (qa[e] * inv_n); c0[e] *= inv_n; }
            if (xb) {
                ld8(mean_b + w.cx * 8, mb); ld8(gamma_b + w.cx * 8, kb); ld8(invstd_b + w.cx * 8, ia); ld8(coef + 2 * C + w.cx * 8, qb);
#pragma unroll
                for (int e = 0; e < 8; ++e) { kb[e] *= ia[e]; qb[e] = ia[e] * (qb[e] * inv_n); }
                if (regate) {
                    ld8(beta_b + w.cx * 8, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs[e] += t[e];
                }
            }
        }
        const unsigned rows_out = (unsigned)B * TP;
        for (; w.ro < rows_out; w.next()) {
            const int b = (int)w.b, t = (int)w.tp - padmax, cx = w.cx;
            float oa[8], ob[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { oa[e] = 0.f; ob[e] = 0.f; }
            if (t >= 0 && t < TT) {
                float g[8], o[8], v[8], u[8];
                Vec8<T>::load(dy + sdy.at(b, t) * C + cx * 8, g);
                if (relu && !regate) Vec8<T>::load(y + sy.at(b, t) * C + cx * 8, o);
                Vec8<T>::load(xa + sa.at(b, t) * C + cx * 8, v);
                if (xb) Vec8<T>::load(xb + sb.at(b, t) * C + cx * 8, u);
                if (regate) {                                   // ka = gamma invstd is the forward's scale: (x - mean) ka + beta, both branches, as bn_apply_kernel formed it
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float pre = (v[e] - ma[e]) * ka[e] + (xb ? (u[e] - mb[e]) * kb[e] : 0.f) + bs[e]; g[e] = pre > 0.f ? g[e] : 0.f; }
                } else if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) oa[e] = ka[e] * (g[e] - c0[e] - (v[e] - ma[e]) * qa[e]);
                if (xb) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ob[e] = kb[e] * (g[e] - c0[e] - (u[e] - mb[e]) * qb[e]);
                }
            }
            const int ta = t + sda.pad, tb = t + sdb.pad;
            if (dxa && ta >= 0 && ta < TT + 2 * sda.pad) Vec8<T>::store(dxa + ((long long)b * (TT + 2 * sda.pad) + ta) * C + cx * 8, oa);
            if (dxb && tb >= 0 && tb < TT + 2 * sdb.pad) Vec8<T>::store(dxb + ((long long)b * (TT + 2 * sdb.pad) + tb) * C + cx * 8, ob);
        }
        return;
    }
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += step32) {
        const unsigned ro = i / (unsigned)CV; const int cx = (int)(i - ro * (unsigned)CV);
        const unsigned bq = ro / TP; const int b = (int)bq, t = (int)(ro - bq * TP) - padmax;
        float oa[8], ob[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { oa[e] = 0.f; ob[e] = 0.f; }
        const bool halo = t < 0 || t >= TT;
        if (!halo) {
            float g[8], v[8];
            Vec8<T>::load(dy + sdy.at(b, t) * C + cx * 8, g);
            Vec8<T>::load(xa + sa.at(b, t) * C + cx * 8, v);
            if (regate) {
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = 0.f;
                if (xb) Vec8<T>::load(xb + sb.at(b, t) * C + cx * 8, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = cx * 8 + e;
                    float pre = (v[e] - mean_a[c]) * (gamma_a[c] * invstd_a[c]) + beta_a[c];
                    if (xb) pre += (u[e] - mean_b[c]) * (gamma_b[c] * invstd_b[c]) + beta_b[c];
                    g[e] = pre > 0.f ? g[e] : 0.f;
                }
            } else if (relu) { float o[8]; Vec8<T>::load(y + sy.at(b, t) * C + cx * 8, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f; }
#pragma unroll
            for (int e = 0; e < 8; ++e) { const int c = cx * 8 + e; const float xh = (v[e] - mean_a[c]) * invstd_a[c]; oa[e] = gamma_a[c] * invstd_a[c] * (g[e] - coef[c] * inv_n - xh * coef[C + c] * inv_n); }
            if (xb) { Vec8<T>::load(xb + sb.at(b, t) * C + cx * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int c = cx * 8 + e; const float xh = (v[e] - mean_b[c]) * invstd_b[c]; ob[e] = gamma_b[c] * invstd_b[c] * (g[e] - coef[c] * inv_n - xh * coef[2 * C + c] * inv_n); } }
        }
        // halo rows (only for padded outputs) are written as zeros
        const int ta = t + sda.pad, tb = t + sdb.pad;
        if (dxa && ta >= 0 && ta < TT + 2 * sda.pad) Vec8<T>::store(dxa + ((long long)b * (TT + 2 * sda.pad) + ta) * C + cx * 8, oa);
        if (dxb && tb >= 0 && tb < TT + 2 * sdb.pad) Vec8<T>::store(dxb + ((long long)b * (TT + 2 * sdb.pad) + tb) * C + cx * 8, ob);
    }
}

extern "C" int ss_bn_backward_sums(int dtype, const void* dy, int pad_dy, const void* y, int pad_y,
                                   const void* xa, int pad_xa, const float* mean_a, const float* invstd_a,
                                   const void* xb, int pad_xb, const float* mean_b, const float* invstd_b,
                                   float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                                   float* scratch, float* sums, int B, int T, int C, int relu,
                                   const float* gate_gamma_a, const float* gate_beta_a, const float* gate_gamma_b, const float* gate_beta_b, void* stream)
{
    SS_CHECK(dy && xa && mean_a && invstd_a && scratch && sums, "ss_bn_backward_sums: null pointer");
    SS_CHECK(!relu || y || gate_beta_a, "ss_bn_backward_sums: relu backward needs the saved output or the affine parameters to recompute its sign");
    SS_CHECK(!gate_beta_a || (gate_gamma_a && (!xb || (gate_gamma_b && gate_beta_b))), "ss_bn_backward_sums: gate recomputation needs gamma and beta of every branch");
    SS_CHECK(!xb || (mean_b && invstd_b), "ss_bn_backward_sums: second branch incomplete");
    SS_CHECK(C % 8 == 0 && C > 0 && B > 0 && T > 0, "ss_bn_backward_sums: bad shape");
    const int rows = B * T, nch = red_chunks(rows), rpc = (rows + nch - 1) / nch;
    Seq sdy = {T, pad_dy}, sy = {T, pad_y}, sa = {T, pad_xa}, sb = {T, pad_xb};
#define SS_BNB(TT)                                                                                                                             \
    if (relu && gate_beta_a)                                                                                                                  \
        SS_LAUNCH(SS_KERNEL(bn_bwd_partial_kernel<TT, true>), dim3(nch), dim3(red_threads(C)), 0, stream, (const TT*)dy, sdy, (const TT*)y, sy, (const TT*)xa, sa, mean_a, invstd_a, \
                  (const TT*)xb, sb, mean_b, invstd_b, gate_gamma_a, gate_beta_a, gate_gamma_b, gate_beta_b, rows, C, rpc, relu, scratch);       \
    else                                                                                                                                      \
        SS_LAUNCH(SS_KERNEL(bn_bwd_partial_kernel<TT, false>), dim3(nch), dim3(red_threads(C)), 0, stream, (const TT*)dy, sdy, (const TT*)y, sy, (const TT*)xa, sa, mean_a, invstd_a, \
                  (const TT*)xb, sb, mean_b, invstd_b, gate_gamma_a, gate_beta_a, gate_gamma_b, gate_beta_b, rows, C, rpc, relu, scratch);       \
    SS_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, (const float*)scratch, nch, C, sums, dgamma_a, dbeta_a, dgamma_b, dbeta_b)
    if (dtype == SS_BF16) { SS_BNB(bf16_t); } else { SS_BNB(float); }
#undef SS_BNB
    SS_LAUNCH_CHECK("ss_bn_backward_sums");
    return 0;
}

extern "C" int ss_bn_backward_apply(int dtype, const void* dy, int pad_dy, const void* y, int pad_y,
                                    const void* xa, int pad_xa, const float* mean_a, const float* invstd_a, const float* gamma_a,
                                    const void* xb, int pad_xb, const float* mean_b, const float* invstd_b, const float* gamma_b,
                                    const float* sums, double n_total, void* dxa, int pad_dxa, void* dxb, int pad_dxb,
                                    int B, int T, int C, int relu, const float* gate_beta_a, const float* gate_beta_b, void* stream)
{
    SS_CHECK(dy && xa && mean_a && invstd_a && gamma_a && dxa && sums, "ss_bn_backward_apply: null pointer");
    SS_CHECK(!relu || y || gate_beta_a, "ss_bn_backward_apply: relu backward needs the saved output or the affine parameters to recompute its sign");
    SS_CHECK(!gate_beta_a || !xb || gate_beta_b, "ss_bn_backward_apply: gate recomputation needs beta of every branch");
    SS_CHECK(!xb || (mean_b && invstd_b && gamma_b && dxb), "ss_bn_backward_apply: second branch incomplete");
    SS_CHECK(C % 8 == 0 && C > 0 && B > 0 && T > 0 && n_total >= 1.0, "ss_bn_backward_apply: bad shape");
    Seq sdy = {T, pad_dy}, sy = {T, pad_y}, sa = {T, pad_xa}, sb = {T, pad_xb}, sda = {T, pad_dxa}, sdb = {T, pad_dxb};
    const int padmax = pad_dxa > pad_dxb ? pad_dxa : pad_dxb;
    const long long total = (long long)B * (T + 2 * padmax) * (C / 8);
    SS_CHECK(total < (1LL << 31) - (1LL << 22), "ss_bn_backward_apply: tensor too large for 32-bit chunk indices");
    const float inv_n = (float)(1.0 / n_total);
#define SS_BNA(TT)                                                                                                                             \
    if (relu && gate_beta_a)                                                                                                                  \
        SS_LAUNCH(SS_KERNEL(bn_bwd_apply_kernel<TT, true>), ew_grid(total, 256, C / 8), dim3(256), 0, stream, (const TT*)dy, sdy, (const TT*)y, sy, (const TT*)xa, sa, mean_a, invstd_a, gamma_a, \
                  (const TT*)xb, sb, mean_b, invstd_b, gamma_b, sums, inv_n, (TT*)dxa, sda, (TT*)dxb, sdb, B, C, relu, gate_beta_a, gate_beta_b); \
    else                                                                                                                                      \
        SS_LAUNCH(SS_KERNEL(bn_bwd_apply_kernel<TT, false>), ew_grid(total, 256, C / 8), dim3(256), 0, stream, (const TT*)dy, sdy, (const TT*)y, sy, (const TT*)xa, sa, mean_a, invstd_a, gamma_a, \
                  (const TT*)xb, sb, mean_b, invstd_b, gamma_b, sums, inv_n, (TT*)dxa, sda, (TT*)dxb, sdb, B, C, relu, gate_beta_a, gate_beta_b)
    if (dtype == SS_BF16) { SS_BNA(bf16_t); } else { SS_BNA(float); }
#undef SS_BNA
    SS_LAUNCH_CHECK("ss_bn_backward_apply");
    return 0;
}

// =========================================================================== column sums (bias gradients)
template <class T>
__global__ __launch_bounds__(RED_THREADS) void colsum_partial_kernel(const T* __restrict__ x, int rows, int C, long long ld, int rows_per_chunk, float* __restrict__ partial)
{
    __shared__ float red[RED_THREADS * 8];
    ColMap m(C, threadIdx.x, (int)blockDim.x);
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    for (int cb = 0; cb < m.CV; cb += m.CVb) {
        const int cx = cb + m.cx0;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        const bool cv = m.active && cx < m.CV;
        if (cv) {
            int r = r0 + m.ry;
            for (; r + 3 * m.RY < r1; r += 4 * m.RY) {            // 4 independent 16-byte loads in flight per thread
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Vec8<T>::load(x + (long long)(r + u * m.RY) * ld + cx * 8, v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
            }
            for (; r < r1; r += m.RY) { float v[8]; Vec8<T>::load(x + (long long)r * ld + cx * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
        __syncthreads();
        if (cv && m.ry == 0) {
            for (int yy = 1; yy < m.RY; ++yy)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += red[(yy * m.CVb + m.cx0) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) partial[(long long)blockIdx.x * C + cx * 8 + e] = s[e];
        }
    }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nchunks, int C, float* __restrict__ out)
{
    float t[1]; int c;
    if (!chunk_reduce<1>(partial, nchunks, C, t, c)) return;
    out[c] += t[0];
}

static int colsum_chunks(int rows) { int c = (rows + 31) / 32; if (c > 512) c = 512; if (c < 1) c = 1; return c; }   // more chunks only move the time into the finalize pass
extern "C" int64_t ss_colsum_scratch_floats(int rows, int C) { return (int64_t)colsum_chunks(rows) * C; }

extern "C" int ss_colsum(int dtype, const void* x, int rows, int C, int64_t ld, float* scratch, float* out_accum, void* stream)
{
    SS_CHECK(x && out_accum && scratch, "ss_colsum: null pointer");
    SS_CHECK(C % 8 == 0 && C > 0 && rows >= 0 && ld % 8 == 0, "ss_colsum: C and ld must be multiples of 8");
    if (rows == 0) return 0;
    const int nch = colsum_chunks(rows), rpc = (rows + nch - 1) / nch;
    if (dtype == SS_BF16) SS_LAUNCH(colsum_partial_kernel<bf16_t>, dim3(nch), dim3(red_threads(C)), 0, stream, (const bf16_t*)x, rows, C, (long long)ld, rpc, scratch);
    else SS_LAUNCH(colsum_partial_kernel<float>, dim3(nch), dim3(red_threads(C)), 0, stream, (const float*)x, rows, C, (long long)ld, rpc, scratch);
    SS_LAUNCH(colsum_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, (const float*)scratch, nch, C, out_accum);
    SS_LAUNCH_CHECK("ss_colsum");
    return 0;
}

// =========================================================================== residual + dropout + LayerNorm
// One wave per row; a lane keeps NV 8-channel vectors of the row in registers (C <= 512*NV).
template <class T, int NV>
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ a_z, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int rows, int C, float eps,
                                                                 unsigned thresh, float keep_scale, unsigned long long seed, unsigned stream_id,
                                                                 bf16_t* __restrict__ y_hi = nullptr, bf16_t* __restrict__ y_lo = nullptr)
{
    // y_hi / y_lo (f32 only; the parity-grade mode): y also leaves as hi / lo bf16 planes (planes.hip's arithmetic) for the plane GEMMs that consume it
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, CV = C >> 3;
    // gamma / beta of this lane's chunks stay in registers for all of its rows (they were re-read -- 4 x 16 bytes per chunk -- behind their own wait on
    // every row), and the raw chunks of the NEXT row are requested before the current row is reduced: a wave otherwise runs load -> wait -> reduce -> load
    // -> wait -> store, three dependent round trips per row.
    typedef typename RawVec8<T>::type Raw;
    constexpr bool HOIST = NV <= 2;                     // wide rows (NV = 8: 128 registers of constants) keep the per-row loads
    float gm[HOIST ? NV : 1][8], bt[HOIST ? NV : 1][8];
    if (HOIST) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i, cc = cx < CV ? cx : CV - 1;
            ld8(gamma + cc * 8, gm[HOIST ? i : 0]); ld8(beta + cc * 8, bt[HOIST ? i : 0]);
        }
    }
    const int rstep = gridDim.x * wpb;
    int r = blockIdx.x * wpb + (threadIdx.x >> 6);
    Raw nx[NV], na[NV];
    auto fetch = [&](int row) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i;
            if (cx < CV) { nx[i] = RawVec8<T>::load(x + (long long)row * C + cx * 8); na[i] = RawVec8<T>::load(a_z + (long long)row * C + cx * 8); }
            else { nx[i] = RawVec8<T>::zero(); na[i] = RawVec8<T>::zero(); }
        }
    };
    if (r < rows) fetch(r);
    for (; r < rows; r += rstep) {
        Raw cx_[NV], ca_[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { cx_[i] = nx[i]; ca_[i] = na[i]; }
        if (r + rstep < rows) fetch(r + rstep);
        float z[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i;
            if (cx < CV) {
                float xv[8], av[8];
                RawVec8<T>::unpack(cx_[i], xv); RawVec8<T>::unpack(ca_[i], av);
                bool kp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) kp[e] = true;
                if (thresh) {                               // element r*C + c  <->  Philox block (r*C + c) >> 2, word c & 3
                    bool k0[4], k1[4];
                    dropout_keep4(seed, stream_id, ((unsigned long long)r * C + cx * 8) >> 2, thresh, k0);
                    dropout_keep4(seed, stream_id, (((unsigned long long)r * C + cx * 8) >> 2) + 1, thresh, k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { kp[e] = k0[e]; kp[4 + e] = k1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float br = av[e];
                    if (thresh) br = kp[e] ? br * keep_scale : 0.f;
                    z[i][e] = rnd<T>(xv[e] + br);           // z is stored (and re-read by backward) in T
                    s += z[i][e];
                }
                Vec8<T>::store(a_z + (long long)r * C + cx * 8, z[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) z[i][e] = 0.f;
            }
        }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lane + 64 * i < CV)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = z[i][e] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i;
            if (cx < CV) {
                float o[8];
                if (!HOIST) { ld8(gamma + cx * 8, gm[0]); ld8(beta + cx * 8, bt[0]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (z[i][e] - mu) * rs * gm[HOIST ? i : 0][e] + bt[HOIST ? i : 0][e];
                Vec8<T>::store(y + (long long)r * C + cx * 8, o);
                if constexpr (sizeof(T) == 4) {
                    if (y_hi) {
                        u32x4 h, l;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const unsigned hp = pack_bf16(o[2 * p], o[2 * p + 1]);
                            h[p] = hp; l[p] = pack_bf16(o[2 * p] - __uint_as_float(hp << 16), o[2 * p + 1] - __uint_as_float(hp & 0xffff0000u));
                        }
                        *(u32x4*)(y_hi + (long long)r * C + cx * 8) = h; *(u32x4*)(y_lo + (long long)r * C + cx * 8) = l;
                    }
                }
            }
        }
    }
}

// Backward of z = x + dropout(branch); y = LN(z): one wave per row.  The loads of the NEXT row of a wave (dy, z: 2 x NV 16-byte
// loads per lane) are issued before the current row's reductions and stores: a wave otherwise has only those 4 loads in flight and the
// kernel is latency-bound (2.3 TB/s); optional dbsum accumulates the column sums of dbranch (the bias gradient of the nn.Linear that
// produced the branch, transformer.py:58), which saves a separate pass over dbranch.
template <class T, int NV, bool BS>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, T* __restrict__ dres, T* __restrict__ dbranch, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dbsum, int rows, int C, unsigned thresh, float keep_scale, unsigned long long seed, unsigned stream_id, int dbg)
{
    __shared__ float red[4][NV * 64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6, CV = C >> 3;
    float dg[NV][8], db[NV][8], dbs[BS ? NV : 1][8];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; if (BS) dbs[BS ? i : 0][e] = 0.f; }
    float gm[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[i][e] = lane + 64 * i < CV ? gamma[(lane + 64 * i) * 8 + e] : 0.f;
    const int stride = gridDim.x * wpb;
    int r = blockIdx.x * wpb + w;
    // raw 16-byte (bf16) / 2 x 16-byte (f32) chunks: the conversion to float happens at the USE, so the loads of the next row stay in flight
    typedef typename RawVec8<T>::type Raw;
    Raw dcur[NV], zcur[NV];
    auto load_row = [&](int rr, Raw (&d)[NV], Raw (&zz)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i;
            if (cx < CV) { d[i] = RawVec8<T>::load(dy + (long long)rr * C + cx * 8); zz[i] = RawVec8<T>::load(z + (long long)rr * C + cx * 8); }
            else { d[i] = RawVec8<T>::zero(); zz[i] = RawVec8<T>::zero(); }
        }
    };
    float mu = 0.f, rs = 0.f;
    if (r < rows) { load_row(r, dcur, zcur); mu = mean[r]; rs = rstd[r]; }
    for (; r < rows; r += stride) {
        Raw dnx[NV], znx[NV]; float mun = 0.f, rsn = 0.f;
        const bool more = r + stride < rows;
        if (more) { load_row(r + stride, dnx, znx); mun = mean[r + stride]; rsn = rstd[r + stride]; }   // in flight during this row's reductions and stores
        float dc[NV][8], zc[NV][8];
#pragma unroll
        for (int i = 0; i < NV; ++i) { RawVec8<T>::unpack(dcur[i], dc[i]); RawVec8<T>::unpack(zcur[i], zc[i]); }
        float gy[NV][8], xh[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = lane + 64 * i < CV;
                xh[i][e] = ok ? (zc[i][e] - mu) * rs : 0.f;
                dg[i][e] += dc[i][e] * xh[i][e]; db[i][e] += dc[i][e];
                gy[i][e] = dc[i][e] * gm[i][e];
                s1 += gy[i][e]; s2 += gy[i][e] * xh[i][e];
            }
        }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cx = lane + 64 * i;
            if (cx < CV) {
                float o[8], ob[8];
                bool kp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) kp[e] = true;
                if (thresh && dbranch) {
                    bool k0[4], k1[4];
                    dropout_keep4(seed, stream_id, ((unsigned long long)r * C + cx * 8) >> 2, thresh, k0);
                    dropout_keep4(seed, stream_id, (((unsigned long long)r * C + cx * 8) >> 2) + 1, thresh, k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { kp[e] = k0[e]; kp[4 + e] = k1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = rs * (gy[i][e] - s1 - xh[i][e] * s2);
                    ob[e] = (thresh && !kp[e]) ? 0.f : (thresh ? o[e] * keep_scale : o[e]);
                    if (BS) dbs[BS ? i : 0][e] += rnd<T>(ob[e]);  // what the consumers of dbranch read (stored in T)
                }
                Vec8<T>::store(dres + (long long)r * C + cx * 8, o);
                if (dbranch) Vec8<T>::store(dbranch + (long long)r * C + cx * 8, ob);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { dcur[i] = dnx[i]; zcur[i] = znx[i]; }
            mu = mun; rs = rsn;
        }
    }
    // block-level reduction of the affine gradients (and the branch column sums), then one atomic per channel per block
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
        float* const dst = qn == 0 ? dgamma : (qn == 1 ? dbeta : dbsum);
        if (!dst || (qn == 2 && !BS)) continue;               // uniform
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[w][(i * 64 + lane) * 8 + e] = qn == 0 ? dg[i][e] : (qn == 1 ? db[i][e] : dbs[BS ? i : 0][e]);
        __syncthreads();
        for (int idx = threadIdx.x; idx < NV * 64 * 8; idx += blockDim.x) {
            const int i = idx / 512, l = (idx / 8) & 63, e = idx & 7, cx = l + 64 * i;
            if (cx < CV) {
                float a = 0.f;
                for (int ww = 0; ww < wpb; ++ww) a += red[ww][idx];
                if (!(dbg & 1)) atomicAdd(dst + cx * 8 + e, a);
            }
        }
        __syncthreads();
    }
}

// ---- LayerNorm backward, second form: 16 waves per workgroup (<= 128 registers), partial column sums to a scratch buffer.
// The kernel above keeps one row per wave in 64 x 16-byte chunks: for C = 768 the second chunk round uses half the lanes, the column
// accumulators of both rounds plus a prefetched row cost ~200 registers (2 waves per SIMD, ~32 KB of loads in flight per CU) and every
// workgroup ends with 3 C atomics onto the same 3 C addresses (8.5 us of the 44 us measured alone).  Here a lane owns C / 256 pieces of
// 4 consecutive elements (no idle lanes, 8-byte bf16 loads / stores), the second pass recomputes from the raw row instead of keeping
// floats, which fits 16 waves per CU, and the per-workgroup sums go to scratch[workgroup][3][C], added up by a second tiny kernel.
template <class T> struct Piece4;
template <> struct Piece4<bf16_t> {
    typedef u32x2 raw;
    static __device__ __forceinline__ raw load(const bf16_t* p) { return *(const u32x2*)p; }
    static __device__ __forceinline__ void unpack(const raw& r, float (&v)[4]) {
        v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u); v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) { u32x2 r = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}; *(u32x2*)p = r; }
};
template <> struct Piece4<float> {
    typedef f32x4 raw;
    static __device__ __forceinline__ raw load(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void unpack(const raw& r, float (&v)[4]) { v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3]; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { f32x4 r = {v[0], v[1], v[2], v[3]}; *(f32x4*)p = r; }
};
constexpr int LNB2_WAVES = 16;

// DROP: dropout on the branch gradient (uniform).  The row loop is ONE basic block: the next row is always requested (its index
// clamped to the last row: one redundant row per wave at the end), nothing is predicated -- with an `if (more)` prefetch block the
// compiler drained vmcnt(0) at the top of every iteration, i.e. waited for the previous row's STORES before requesting the next row.
template <class T, int P, bool BS, bool DROP>
__global__ __launch_bounds__(LNB2_WAVES * 64) void ln_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, T* __restrict__ dres, T* __restrict__ dbranch, float* __restrict__ partial,
                                                                  int rows, unsigned thresh, float keep_scale, unsigned long long seed, unsigned stream_id,
                                                                  float* dgamma, float* dbeta, float* dbsum, int direct,
                                                                  bf16_t* __restrict__ db_hi = nullptr, bf16_t* __restrict__ db_lo = nullptr)
{
    // db_hi / db_lo (f32 only; the parity-grade mode): dbranch also leaves as hi / lo bf16 planes for the plane GEMMs that consume it
    constexpr int C = P * 256;
    __shared__ float red[LNB2_WAVES][C];
    typedef typename Piece4<T>::raw Raw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float dg[P][4], db[P][4], dbs[BS ? P : 1][4], gm[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[p][e] = 0.f; db[p][e] = 0.f; if (BS) dbs[BS ? p : 0][e] = 0.f; gm[p][e] = gamma[(p * 64 + lane) * 4 + e]; }
    const int stride = gridDim.x * LNB2_WAVES;
    int r = blockIdx.x * LNB2_WAVES + w;
    Raw dcur[P], zcur[P];
    float mu = 0.f, rs = 0.f;
    auto load_row = [&](int rr, Raw (&d)[P], Raw (&zz)[P]) {
#pragma unroll
        for (int p = 0; p < P; ++p) { d[p] = Piece4<T>::load(dy + (long long)rr * C + (p * 64 + lane) * 4); zz[p] = Piece4<T>::load(z + (long long)rr * C + (p * 64 + lane) * 4); }
    };
    if (r < rows) { load_row(r, dcur, zcur); mu = mean[r]; rs = rstd[r]; }
    for (; r < rows; r += stride) {
        Raw dnx[P], znx[P];
        const int rn = r + stride < rows ? r + stride : rows - 1;
        load_row(rn, dnx, znx);
        const float mun = mean[rn], rsn = rstd[rn];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float dc[4], zc[4]; Piece4<T>::unpack(dcur[p], dc); Piece4<T>::unpack(zcur[p], zc);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (zc[e] - mu) * rs, gy = dc[e] * gm[p][e];
                dg[p][e] += dc[e] * xh; db[p][e] += dc[e];
                s1 += gy; s2 += gy * xh;
            }
        }
        s1 = wave_sum(s1) * (1.f / (float)C); s2 = wave_sum(s2) * (1.f / (float)C);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float dc[4], zc[4], o[4], ob[4]; Piece4<T>::unpack(dcur[p], dc); Piece4<T>::unpack(zcur[p], zc);
            bool kp[4] = {true, true, true, true};
            const long long col = (p * 64 + lane) * 4;
            if (DROP) dropout_keep4(seed, stream_id, ((unsigned long long)r * C + col) >> 2, thresh, kp);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (zc[e] - mu) * rs;
                o[e] = rs * (dc[e] * gm[p][e] - s1 - xh * s2);
                ob[e] = DROP ? (kp[e] ? o[e] * keep_scale : 0.f) : o[e];
                if (BS) dbs[BS ? p : 0][e] += rnd<T>(ob[e]);
            }
            Piece4<T>::store(dres + (long long)r * C + col, o);
            Piece4<T>::store(dbranch + (long long)r * C + col, ob);
            if constexpr (sizeof(T) == 4) {
                if (db_hi) {
                    const unsigned h0 = pack_bf16(ob[0], ob[1]), h1 = pack_bf16(ob[2], ob[3]);
                    const u32x2 hw = {h0, h1};
                    const u32x2 lw = {pack_bf16(ob[0] - __uint_as_float(h0 << 16), ob[1] - __uint_as_float(h0 & 0xffff0000u)),
                                      pack_bf16(ob[2] - __uint_as_float(h1 << 16), ob[3] - __uint_as_float(h1 & 0xffff0000u))};
                    *(u32x2*)(db_hi + (long long)r * C + col) = hw; *(u32x2*)(db_lo + (long long)r * C + col) = lw;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) { dcur[p] = dnx[p]; zcur[p] = znx[p]; }
        mu = mun; rs = rsn;
    }
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
        if (qn == 2 && !BS) continue;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[w][(p * 64 + lane) * 4 + e] = qn == 0 ? dg[p][e] : (qn == 1 ? db[p][e] : dbs[BS ? p : 0][e]);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += LNB2_WAVES * 64) {
            float a = 0.f;
#pragma unroll
            for (int ww = 0; ww < LNB2_WAVES; ++ww) a += red[ww][c];
            // direct: one atomic per column and workgroup (#CU workgroups: a quarter of the atomics of the 4-wave form, spread over the
            // moments the workgroups finish) instead of a partial row + the second launch
            if (direct) atomicAdd((qn == 0 ? dgamma : (qn == 1 ? dbeta : dbsum)) + c, a);
            else partial[((long long)blockIdx.x * 3 + qn) * C + c] = a;
        }
        __syncthreads();
    }
}

// 64 columns x 16 slices of the workgroup range per block: 16 independent loads per thread instead of one thread walking all partials
__global__ __launch_bounds__(1024) void ln_bwd2_finalize_kernel(const float* __restrict__ partial, int nblocks, int C, float* dgamma, float* dbeta, float* dbsum)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                        // 3 C is a multiple of 64
    const int qn = i / C, c = i - qn * C;
    float a = 0.f;
    for (int b = sl; b < nblocks; b += 16) a += partial[((long long)b * 3 + qn) * C + c];
    red[sl][lane] = a;
    __syncthreads();
    if (sl == 0) {
        float* const dst = qn == 0 ? dgamma : (qn == 1 ? dbeta : dbsum);
        if (dst) {
#pragma unroll
            for (int k = 1; k < 16; ++k) a += red[k][lane];
            dst[c] += a;
        }
    }
}

static int lnb2_blocks(int rows) {
    static int cus = 0;
    if (!cus) {
#if defined(SS_EMU)
        cus = 2;
#else
        int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
#endif
    }
    const int need = (rows + LNB2_WAVES - 1) / LNB2_WAVES;
    return need < cus ? need : cus;
}
// floats of scratch ss_layernorm_backward_ws wants for (rows, C); 0 if this shape runs the atomic form (C not 256, 512 or 768)
extern "C" int64_t ss_layernorm_backward_scratch_floats(int rows, int C)
{
    if (rows <= 0 || (C != 256 && C != 512 && C != 768)) return 0;
    return (int64_t)lnb2_blocks(rows) * 3 * C;
}

extern "C" int ss_add_dropout_layernorm_forward(int dtype, const void* x, void* branch_inout, const float* gamma, const float* beta, void* y,
                                                float* mean, float* rstd, int rows, int C, float eps, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_add_dropout_layernorm_forward_planes(dtype, x, branch_inout, gamma, beta, y, nullptr, nullptr, mean, rstd, rows, C, eps, dropout_p, seed, rng_stream, stream);
}

extern "C" int ss_add_dropout_layernorm_forward_planes(int dtype, const void* x, void* branch_inout, const float* gamma, const float* beta, void* y, void* y_hi, void* y_lo,
                                                       float* mean, float* rstd, int rows, int C, float eps, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    SS_CHECK(x && branch_inout && gamma && beta && y && mean && rstd, "ss_add_dropout_layernorm_forward: null pointer");
    SS_CHECK((y_hi != nullptr) == (y_lo != nullptr) && (!y_hi || (dtype == SS_F32 && ((uintptr_t)y_hi | (uintptr_t)y_lo) % 16 == 0)), "ss_add_dropout_layernorm_forward_planes: planes come as a pair, for f32 data, 16-byte aligned");
    SS_CHECK(C % 8 == 0 && C > 0 && C <= 4096, "ss_add_dropout_layernorm_forward: C=%d must be a multiple of 8 and <= 4096", C);
    SS_CHECK(dropout_p >= 0.f && dropout_p < 1.f, "dropout p out of range");
    if (rows <= 0) return 0;
    const unsigned th = dropout_threshold(dropout_p); const float ks = 1.f / (1.f - dropout_p);
    int blocks = (rows + 3) / 4; if (blocks > 4096) blocks = 4096;
#define SS_LNF(TT, NV) SS_LAUNCH(SS_KERNEL(add_dropout_ln_fwd_kernel<TT, NV>), dim3(blocks), dim3(256), 0, stream, (const TT*)x, (TT*)branch_inout, gamma, beta, (TT*)y, mean, rstd, rows, C, eps, th, ks, (unsigned long long)seed, rng_stream, (bf16_t*)y_hi, (bf16_t*)y_lo)
    if (dtype == SS_BF16) { if (C <= 1024) SS_LNF(bf16_t, 2); else SS_LNF(bf16_t, 8); }
    else { if (C <= 1024) SS_LNF(float, 2); else SS_LNF(float, 8); }
#undef SS_LNF
    SS_LAUNCH_CHECK("ss_add_dropout_layernorm_forward");
    return 0;
}

extern "C" int ss_layernorm_backward(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                     void* dres, void* dbranch, float* dgamma, float* dbeta, int rows, int C, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_layernorm_backward_bias(dtype, dy, z, mean, rstd, gamma, dres, dbranch, dgamma, dbeta, nullptr, rows, C, dropout_p, seed, rng_stream, stream);
}

extern "C" int ss_layernorm_backward_bias(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                          void* dres, void* dbranch, float* dgamma, float* dbeta, float* dbranch_colsum, int rows, int C, float dropout_p,
                                          uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_layernorm_backward_ws(dtype, dy, z, mean, rstd, gamma, dres, dbranch, dgamma, dbeta, dbranch_colsum, nullptr, 0, rows, C, dropout_p, seed, rng_stream, stream);
}

extern "C" int ss_layernorm_backward_ws(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                        void* dres, void* dbranch, float* dgamma, float* dbeta, float* dbranch_colsum, float* scratch, int64_t scratch_floats,
                                        int rows, int C, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    return ss_layernorm_backward_ws_planes(dtype, dy, z, mean, rstd, gamma, dres, dbranch, nullptr, nullptr, dgamma, dbeta, dbranch_colsum, scratch, scratch_floats, rows, C, dropout_p, seed, rng_stream, stream);
}

// the same with dbranch ALSO written as hi / lo bf16 planes (f32 data, the workspace form only: returns an error when the shape would take the atomic form)
extern "C" int ss_layernorm_backward_ws_planes(int dtype, const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                               void* dres, void* dbranch, void* dbranch_hi, void* dbranch_lo, float* dgamma, float* dbeta, float* dbranch_colsum, float* scratch,
                                               int64_t scratch_floats, int rows, int C, float dropout_p, uint64_t seed, uint32_t rng_stream, void* stream)
{
    SS_CHECK((dbranch_hi != nullptr) == (dbranch_lo != nullptr) && (!dbranch_hi || (dtype == SS_F32 && dbranch && ((uintptr_t)dbranch_hi | (uintptr_t)dbranch_lo) % 8 == 0)),
             "ss_layernorm_backward_ws_planes: planes come as a pair, for f32 data with dbranch, 8-byte aligned");
    SS_CHECK(dy && z && mean && rstd && gamma && dres && dgamma && dbeta, "ss_layernorm_backward: null pointer");
    SS_CHECK(!dbranch_colsum || dbranch, "ss_layernorm_backward_bias: the column sums are those of dbranch");
    SS_CHECK(C % 8 == 0 && C > 0 && C <= 4096, "ss_layernorm_backward: C=%d must be a multiple of 8 and <= 4096", C);
    if (rows <= 0) return 0;
    const unsigned th = dropout_threshold(dropout_p); const float ks = 1.f / (1.f - dropout_p);
    {
        const int64_t want = ss_layernorm_backward_scratch_floats(rows, C);
        const char* e = getenv("SS_LN_BWD2");                 // "0": the atomic form (A/B measurements, tests of both)
        if (scratch && dbranch && want > 0 && scratch_floats >= want && !(e && e[0] == '0')) {
            const int nb = lnb2_blocks(rows);
            int direct = 1; { const char* e2 = getenv("SS_LN_DIRECT"); if (e2) direct = atoi(e2); }
#define SS_LNB2K(TT, PP, BSV, DRV) SS_LAUNCH(SS_KERNEL(ln_bwd2_kernel<TT, PP, BSV, DRV>), dim3(nb), dim3(LNB2_WAVES * 64), 0, stream, (const TT*)dy, (const TT*)z, mean, rstd, gamma, (TT*)dres, (TT*)dbranch, scratch, rows, th, ks, (unsigned long long)seed, rng_stream, dgamma, dbeta, dbranch_colsum, direct, (bf16_t*)dbranch_hi, (bf16_t*)dbranch_lo)
#define SS_LNB2(TT, PP) do { if (dbranch_colsum) { if (th) SS_LNB2K(TT, PP, true, true); else SS_LNB2K(TT, PP, true, false); } \
                             else { if (th) SS_LNB2K(TT, PP, false, true); else SS_LNB2K(TT, PP, false, false); } } while (0)
            if (dtype == SS_BF16) { if (C == 256) SS_LNB2(bf16_t, 1); else if (C == 512) SS_LNB2(bf16_t, 2); else SS_LNB2(bf16_t, 3); }
            else { if (C == 256) SS_LNB2(float, 1); else if (C == 512) SS_LNB2(float, 2); else SS_LNB2(float, 3); }
#undef SS_LNB2K
#undef SS_LNB2
            if (!direct) SS_LAUNCH(ln_bwd2_finalize_kernel, dim3(3 * C / 64), dim3(1024), 0, stream, (const float*)scratch, nb, C, dgamma, dbeta, dbranch_colsum);
            SS_LAUNCH_CHECK("ss_layernorm_backward_ws");
            return 0;
        }
    }
    SS_CHECK(!dbranch_hi, "ss_layernorm_backward_ws_planes: plane output needs the workspace form (C = 256, 512 or 768 with scratch of ss_layernorm_backward_scratch_floats)");
    // 2 blocks per CU (~200 registers).  On gfx9 a wait for the prefetched loads also drains the previous row's stores (one vmcnt for both),
    // so consecutive rows of one wave overlap only partly; forcing 3 waves per SIMD (168 registers) spills the column accumulators.
    int blocks = (rows + 15) / 16; if (blocks > 512) blocks = 512;
    int dbg = 0; { const char* e = getenv("SS_LN_DEBUG"); if (e) dbg = atoi(e); const char* b = getenv("SS_LN_BLOCKS"); if (b) blocks = atoi(b); }
#define SS_LNB(TT, NV) do { if (dbranch_colsum) SS_LAUNCH(SS_KERNEL(ln_bwd_kernel<TT, NV, true>), dim3(blocks), dim3(256), 0, stream, (const TT*)dy, (const TT*)z, mean, rstd, gamma, (TT*)dres, (TT*)dbranch, dgamma, dbeta, dbranch_colsum, rows, C, th, ks, (unsigned long long)seed, rng_stream, dbg); \
                            else SS_LAUNCH(SS_KERNEL(ln_bwd_kernel<TT, NV, false>), dim3(blocks), dim3(256), 0, stream, (const TT*)dy, (const TT*)z, mean, rstd, gamma, (TT*)dres, (TT*)dbranch, dgamma, dbeta, dbranch_colsum, rows, C, th, ks, (unsigned long long)seed, rng_stream, dbg); } while (0)
    if (dtype == SS_BF16) { if (C <= 1024) SS_LNB(bf16_t, 2); else SS_LNB(bf16_t, 8); }
    else { if (C <= 1024) SS_LNB(float, 2); else SS_LNB(float, 8); }
#undef SS_LNB
    SS_LAUNCH_CHECK("ss_layernorm_backward");
    return 0;
}

// =========================================================================== EMG input conditioning
// out (B, T0+2, 8) in the compute dtype = left-shift-by-r of x_raw (B, T0, 8) f32, zero tail, zero halo rows;
// optionally also writes the shifted f32 copy back (the reference mutates its input in place).
template <class T>
__global__ void emg_prepare_kernel(const float* __restrict__ x, T* __restrict__ out, float* __restrict__ shifted, int B, int T0, int Cin, int r)
{
    const long long total = (long long)B * (T0 + 2) * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin); const long long ro = i / Cin;
        const int b = (int)(ro / (T0 + 2)), t = (int)(ro - (long long)b * (T0 + 2)) - 1;
        float v = 0.f;
        if (t >= 0 && t < T0) {
            v = t + r < T0 ? x[((long long)b * T0 + t + r) * Cin + c] : 0.f;
            if (shifted) shifted[((long long)b * T0 + t) * Cin + c] = v;
        }
        stf(out + i, v);
    }
}

extern "C" int ss_emg_prepare(int dtype, const float* x_raw, void* out_padded, float* shifted_copy, int B, int T0, int Cin, int shift, void* stream)
{
    SS_CHECK(x_raw && out_padded, "ss_emg_prepare: null pointer");
    SS_CHECK(B > 0 && T0 > 0 && Cin > 0 && shift >= 0 && shift < T0, "ss_emg_prepare: bad shape/shift");
    const long long total = (long long)B * (T0 + 2) * Cin;
    if (dtype == SS_BF16) SS_LAUNCH(emg_prepare_kernel<bf16_t>, ew_grid(total, 256), dim3(256), 0, stream, x_raw, (bf16_t*)out_padded, shifted_copy, B, T0, Cin, shift);
    else SS_LAUNCH(emg_prepare_kernel<float>, ew_grid(total, 256), dim3(256), 0, stream, x_raw, (float*)out_padded, shifted_copy, B, T0, Cin, shift);
    SS_LAUNCH_CHECK("ss_emg_prepare");
    return 0;
}
