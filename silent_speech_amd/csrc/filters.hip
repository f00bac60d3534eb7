// filters.hip -- offline EMG conditioning on the MI355X ("next" row N4 of SURVEY section 8f): the zero-phase IIR cascade and the
// linear-interpolation resampling that reference read_emg.py applies to every raw 1 kHz recording before anything else sees it:
//     notch_harmonics  read_emg.py:31-38   7 x scipy.signal.filtfilt(iirnotch(60 h, Q = 30))        (2nd order each)
//     remove_drift     read_emg.py:27-29   scipy.signal.filtfilt(butter(3, 2 Hz, 'highpass'))        (3rd order)
//     subsample        read_emg.py:40-44   np.interp onto the 689.06 Hz / 516.79 Hz grids
// All arithmetic is f64 like numpy's.  An IIR recurrence is sequential in time, so each filtfilt pass is parallelised by CHUNKS:
// scipy's transposed-direct-form-II recurrence is linear in its state, z' = A z + B x, y = b0 x + z0, hence
//   1. every chunk of L samples is run from a zero state -> its end state p_k                          (chunk_state_kernel)
//   2. the true chunk-initial states follow from s_{k+1} = A^L s_k + p_k, s_0 = zi * x_ext[0]            (prefix_kernel, serial in k, tiny)
//   3. every chunk is re-run from its true initial state and writes its outputs                          (apply_kernel)
// with A^L computed on the host.  The backward pass is the same three kernels on the reversed index; scipy's odd extension
// (padlen = 3 max(len a, len b)) and its initial conditions (lfilter_zi) are reproduced exactly, so results agree with
// scipy.signal.filtfilt to f64 round-off (the chunking changes the order of a few additions).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>
#include <vector>

// numpy / scipy evaluate these expressions with separate multiplies and adds (x86-64 baseline, no FMA); contraction into f64 FMAs
// would move results by an ulp -- and np.interp parity is checked bit for bit
#pragma clang fp contract(off)

namespace {
constexpr int FL_MIN = 256, FL_MAX = 4096;   // samples per chunk: chosen per filter, see ss_iir_filtfilt
constexpr int FN = 3;              // maximum state dimension (order-3 Butterworth)

struct Filt { double b[FN + 1], a[FN + 1], zi[FN], AL[FN][FN]; int n, padlen, L; };

__device__ __forceinline__ long long fidx(long long t, long long Te, int rev) { return rev ? Te - 1 - t : t; }

// x (T, C) -> odd extension (T + 2 padlen, C); src may carry `skip` leading rows of a previous extension
__global__ void ext_kernel(const double* __restrict__ src, long long skip, double* __restrict__ dst, int T, int C, int padlen)
{
    const long long Te = (long long)T + 2 * padlen, total = Te * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long t = i / C;
        const double* x = src + skip * C + c;
        double v;
        if (t < padlen) v = 2.0 * x[0] - x[(long long)(padlen - t) * C];
        else if (t < padlen + T) v = x[(t - padlen) * C];
        else v = 2.0 * x[(long long)(T - 1) * C] - x[(long long)(T - 2 - (t - padlen - T)) * C];
        dst[i] = v;
    }
}

__device__ __forceinline__ double tdf2_step(const Filt& f, double x, double (&z)[FN]) {
    const double y = f.b[0] * x + z[0];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const double zn = i + 1 < FN ? z[i + 1] : 0.0;
        z[i] = i < f.n ? f.b[i + 1] * x + zn - f.a[i + 1] * y : 0.0;
    }
    return y;
}


// The recurrences are serial in time, but their LOADS are not: a chunk is walked in blocks of IIR_BLK samples whose loads are all requested
// before the first step runs (round 5: the plain `for t: step(x[t])` loops paid one memory round trip per sample -- 165 us per launch for
// 10 MB of signal; the arithmetic, its order and the results are unchanged).
constexpr int IIR_BLK = 16;
template <class GET, class PUT>
__device__ __forceinline__ void iir_walk(long long t0, long long t1, GET&& get, PUT&& put)
{
    if (t0 >= t1) return;
    double v[IIR_BLK], nx[IIR_BLK];
#pragma unroll
    for (int i = 0; i < IIR_BLK; ++i) v[i] = get(t0 + i < t1 ? t0 + i : t1 - 1);
    for (long long t = t0; t < t1; t += IIR_BLK) {
        const long long tn = t + IIR_BLK;                          // the next block travels while this one is stepped through
        if (tn < t1) {
#pragma unroll
            for (int i = 0; i < IIR_BLK; ++i) nx[i] = get(tn + i < t1 ? tn + i : t1 - 1);
        }
#pragma unroll
        for (int i = 0; i < IIR_BLK; ++i) if (t + i < t1) put(t + i, v[i]);
#pragma unroll
        for (int i = 0; i < IIR_BLK; ++i) v[i] = nx[i];
    }
}

__global__ void chunk_state_kernel(const double* __restrict__ x, int Te, int C, int nchunks, int rev, Filt f, double* __restrict__ P)
{
    const int c = threadIdx.x % C, k = blockIdx.x * (blockDim.x / C) + threadIdx.x / C;
    if (k >= nchunks || threadIdx.x >= blockDim.x / C * C) return;
    double z[FN] = {0.0, 0.0, 0.0};
    const long long t0 = (long long)k * f.L, t1 = min((long long)Te, t0 + f.L);
    iir_walk(t0, t1, [&](long long t) { return x[fidx(t, Te, rev) * C + c]; }, [&](long long, double v) { tdf2_step(f, v, z); });
#pragma unroll
    for (int i = 0; i < FN; ++i) P[((long long)k * C + c) * FN + i] = z[i];
}

__global__ void prefix_kernel(const double* __restrict__ x, int Te, int C, int nchunks, int rev, Filt f, const double* __restrict__ P, double* __restrict__ S)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double x0 = x[fidx(0, Te, rev) * C + c];
    double s[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) s[i] = i < f.n ? f.zi[i] * x0 : 0.0;
    for (int k = 0; k < nchunks; ++k) {
        double nx[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            S[((long long)k * C + c) * FN + i] = s[i];
            double acc = P[((long long)k * C + c) * FN + i];
#pragma unroll
            for (int j = 0; j < FN; ++j) acc += f.AL[i][j] * s[j];
            nx[i] = acc;
        }
        // the last chunk may be shorter than L: its end state is never used
#pragma unroll
        for (int i = 0; i < FN; ++i) s[i] = nx[i];
    }
}

__global__ void apply_kernel(const double* __restrict__ x, double* __restrict__ y, int Te, int C, int nchunks, int rev, Filt f, const double* __restrict__ S)
{
    const int c = threadIdx.x % C, k = blockIdx.x * (blockDim.x / C) + threadIdx.x / C;
    if (k >= nchunks || threadIdx.x >= blockDim.x / C * C) return;
    double z[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) z[i] = S[((long long)k * C + c) * FN + i];
    const long long t0 = (long long)k * f.L, t1 = min((long long)Te, t0 + f.L);
    iir_walk(t0, t1, [&](long long t) { return x[fidx(t, Te, rev) * C + c]; }, [&](long long t, double v) { y[fidx(t, Te, rev) * C + c] = tdf2_step(f, v, z); });
}

__global__ void crop_kernel(const double* __restrict__ src, long long skip, double* __restrict__ dst, int T, int C)
{
    const long long total = (long long)T * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) dst[i] = src[skip * C + i];
}

// np.interp(sample_times, times, x) with times[j] = j / old_freq, sample_times[i] = i / new_freq   (read_emg.py:40-44)
__global__ void resample_kernel(const double* __restrict__ x, double* __restrict__ y, int T, int C, double old_freq, double new_freq, int T_out)
{
    const long long total = (long long)T_out * C;
    const double step = 1.0 / new_freq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long o = i / C;
        const double t = (double)o * step;
        long long j = (long long)(t * old_freq);
        if (j > T - 1) j = T - 1;
        while (j > 0 && (double)j / old_freq > t) --j;               // numpy searches the xp VALUES: settle rounding at chunk boundaries the same way
        while (j + 1 < T && (double)(j + 1) / old_freq <= t) ++j;
        double v;
        if (j >= T - 1) v = x[(long long)(T - 1) * C + c];
        else {
            const double x0 = (double)j / old_freq, x1 = (double)(j + 1) / old_freq, f0 = x[j * C + c], f1 = x[(j + 1) * C + c];
            const double slope = (f1 - f0) / (x1 - x0);
            v = slope * (t - x0) + f0;
        }
        y[i] = v;
    }
}

// ---- ragged batch: R recordings of different lengths through the same cascade in ONE launch sequence.  Recording u occupies rows
// [sum T_<u, +T_u) of the packed (sum T, C) input / output; per filter a small device table says where its extended copy, its chunks
// and its source sit.  Work items are (chunk, channel) pairs over ALL recordings, found by binary search in the table.
struct RagRow { long long src_off, T, ext_off, Te, chunk_off, nch; };     // elements (doubles) / rows / chunks

__device__ __forceinline__ int rag_find_ext(const RagRow* __restrict__ tab, int R, long long e) {
    int lo = 0, hi = R - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].ext_off <= e) lo = mid; else hi = mid - 1; }
    return lo;
}
__device__ __forceinline__ int rag_find_chunk(const RagRow* __restrict__ tab, int R, long long g) {
    int lo = 0, hi = R - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].chunk_off <= g) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void ext_ragged_kernel(const double* __restrict__ src, double* __restrict__ dst, const RagRow* __restrict__ tab, int R, int C, int padlen, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int u = rag_find_ext(tab, R, i);
        const long long loc = i - tab[u].ext_off, t = loc / C, T = tab[u].T;
        const int c = (int)(loc - t * C);
        const double* x = src + tab[u].src_off + c;
        double v;
        if (t < padlen) v = 2.0 * x[0] - x[(long long)(padlen - t) * C];
        else if (t < padlen + T) v = x[(t - padlen) * C];
        else v = 2.0 * x[(T - 1) * C] - x[(T - 2 - (t - padlen - T)) * C];
        dst[i] = v;
    }
}
__global__ void chunk_state_ragged_kernel(const double* __restrict__ x, const RagRow* __restrict__ tab, int R, int C, long long nchunks, int rev, Filt f, double* __restrict__ P)
{
    const int c = threadIdx.x % C; const long long g = (long long)blockIdx.x * (blockDim.x / C) + threadIdx.x / C;
    if (g >= nchunks || threadIdx.x >= blockDim.x / C * C) return;
    const int u = rag_find_chunk(tab, R, g);
    const long long Te = tab[u].Te, k = g - tab[u].chunk_off;
    const double* xe = x + tab[u].ext_off;
    double z[FN] = {0.0, 0.0, 0.0};
    const long long t0 = k * f.L, t1 = min(Te, t0 + f.L);
    iir_walk(t0, t1, [&](long long t) { return xe[fidx(t, Te, rev) * C + c]; }, [&](long long, double v) { tdf2_step(f, v, z); });
#pragma unroll
    for (int i = 0; i < FN; ++i) P[(g * C + c) * FN + i] = z[i];
}
__global__ void prefix_ragged_kernel(const double* __restrict__ x, const RagRow* __restrict__ tab, int R, int C, int rev, Filt f, const double* __restrict__ P, double* __restrict__ S)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= R * C) return;
    const int u = id / C, c = id - u * C;
    const long long Te = tab[u].Te, g0 = tab[u].chunk_off; const int nch = (int)tab[u].nch;
    const double x0 = x[tab[u].ext_off + fidx(0, Te, rev) * C + c];
    double s[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) s[i] = i < f.n ? f.zi[i] * x0 : 0.0;
    for (int k = 0; k < nch; ++k) {
        double nx[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            S[((g0 + k) * C + c) * FN + i] = s[i];
            double acc = P[((g0 + k) * C + c) * FN + i];
#pragma unroll
            for (int j = 0; j < FN; ++j) acc += f.AL[i][j] * s[j];
            nx[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i) s[i] = nx[i];
    }
}
__global__ void apply_ragged_kernel(const double* __restrict__ x, double* __restrict__ y, const RagRow* __restrict__ tab, int R, int C, long long nchunks, int rev, Filt f, const double* __restrict__ S)
{
    const int c = threadIdx.x % C; const long long g = (long long)blockIdx.x * (blockDim.x / C) + threadIdx.x / C;
    if (g >= nchunks || threadIdx.x >= blockDim.x / C * C) return;
    const int u = rag_find_chunk(tab, R, g);
    const long long Te = tab[u].Te, k = g - tab[u].chunk_off, base = tab[u].ext_off;
    double z[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) z[i] = S[(g * C + c) * FN + i];
    const long long t0 = k * f.L, t1 = min(Te, t0 + f.L);
    iir_walk(t0, t1, [&](long long t) { return x[base + fidx(t, Te, rev) * C + c]; }, [&](long long t, double v) { y[base + fidx(t, Te, rev) * C + c] = tdf2_step(f, v, z); });
}
// packed output row r of recording u <- extended row padlen + r   (out_off = prefix sum of T * C: the src_off column of table `tout`)
__global__ void crop_ragged_kernel(const double* __restrict__ src, double* __restrict__ dst, const RagRow* __restrict__ tab, const RagRow* __restrict__ tout, int R, int C, int padlen, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = R - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tout[mid].src_off <= i) lo = mid; else hi = mid - 1; }
        dst[i] = src[tab[lo].ext_off + (long long)padlen * C + (i - tout[lo].src_off)];
    }
}
// tab4[u] = {in_off (rows), T, out_off (rows), T_out}: np.interp per recording onto its own grid
__global__ void resample_ragged_kernel(const double* __restrict__ x, double* __restrict__ y, const long long* __restrict__ tab4, int R, int C, double old_freq, double new_freq, long long total_out_rows)
{
    const long long total = total_out_rows * C;
    const double step = 1.0 / new_freq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long og = i / C;
        int lo = 0, hi = R - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab4[mid * 4 + 2] <= og) lo = mid; else hi = mid - 1; }
        const long long o = og - tab4[lo * 4 + 2], T = tab4[lo * 4 + 1];
        const double* xs = x + tab4[lo * 4] * C;
        const double t = (double)o * step;
        long long j = (long long)(t * old_freq);
        if (j > T - 1) j = T - 1;
        while (j > 0 && (double)j / old_freq > t) --j;
        while (j + 1 < T && (double)(j + 1) / old_freq <= t) ++j;
        double v;
        if (j >= T - 1) v = xs[(T - 1) * C + c];
        else {
            const double x0 = (double)j / old_freq, x1 = (double)(j + 1) / old_freq, f0 = xs[j * C + c], f1 = xs[(j + 1) * C + c];
            const double slope = (f1 - f0) / (x1 - x0);
            v = slope * (t - x0) + f0;
        }
        y[i] = v;
    }
}

// host side of one filter: coefficients, state matrix of the transposed direct form II (z' = A z + (b[1:] - a[1:] b0) x, A[i][0] = -a[i+1],
// A[i][i+1] = 1), chunk length L and A^L.  The drift filter's companion matrix is far from normal (three poles at 0.99: |A^k| climbs to
// ~1e4 before it decays), and whatever |A^L| is multiplies the rounding of the chunk states; so L is the first of 256, 512, .. at which
// max |A^L| <= 2 (256 for the notches, 1024 for the drift filter), and A^L comes from L plain products in extended precision rather than
// repeated squaring in f64.
static int make_filt(const double* cf, int q, Filt& f)
{
    memset(&f, 0, sizeof(f));
    for (int i = 0; i < 4; ++i) { f.b[i] = cf[i]; f.a[i] = cf[4 + i]; }
    for (int i = 0; i < 3; ++i) f.zi[i] = cf[8 + i];
    f.n = (int)cf[11]; f.padlen = (int)cf[12];
    SS_CHECK(f.n >= 1 && f.n <= FN && f.a[0] == 1.0, "ss_iir_filtfilt: filter %d: order 1..3 with a[0] == 1 expected", q);
    double A[FN][FN]; memset(A, 0, sizeof(A));
    for (int i = 0; i < f.n; ++i) { A[i][0] = -f.a[i + 1]; if (i + 1 < f.n) A[i][i + 1] = 1.0; }
    long double Rm[FN][FN], Tm[FN][FN];
    for (int i = 0; i < FN; ++i) for (int j = 0; j < FN; ++j) Rm[i][j] = i == j ? 1.0L : 0.0L;
    for (int e = 1; e <= FL_MAX; ++e) {
        for (int i = 0; i < FN; ++i) for (int j = 0; j < FN; ++j) { long double acc = 0; for (int k = 0; k < FN; ++k) acc += (long double)A[i][k] * Rm[k][j]; Tm[i][j] = acc; }
        memcpy(Rm, Tm, sizeof(Rm));
        if (e >= FL_MIN && (e & (e - 1)) == 0) {
            long double mx = 0; for (int i = 0; i < FN; ++i) for (int j = 0; j < FN; ++j) mx = fabsl(Rm[i][j]) > mx ? fabsl(Rm[i][j]) : mx;
            if (mx <= 2.0L || e == FL_MAX) { f.L = e; break; }
        }
    }
    for (int i = 0; i < FN; ++i) for (int j = 0; j < FN; ++j) f.AL[i][j] = (double)Rm[i][j];
    return 0;
}

}  // namespace

extern "C" int64_t ss_iir_filtfilt_workspace_bytes(int T, int C, int max_padlen)
{
    if (T <= 0 || C <= 0) return 0;
    const long long Te = (long long)T + 2 * max_padlen, nch = (Te + FL_MIN - 1) / FL_MIN;
    return (2 * Te * C + 2 * nch * C * FN) * 8 + 1024;
}

// coef: n_filt x { b[4], a[4], zi[3], n (as double), padlen (as double) } = 13 doubles per filter, host memory
extern "C" int ss_iir_filtfilt(const double* x, double* y, int T, int C, int n_filt, const double* coef, void* workspace, int64_t workspace_bytes, void* stream)
{
    SS_CHECK(x && y && coef && workspace, "ss_iir_filtfilt: null pointer");
    SS_CHECK(T > 0 && C > 0 && C <= 64 && n_filt >= 1 && n_filt <= 16, "ss_iir_filtfilt: bad sizes");
    int max_pad = 0;
    for (int q = 0; q < n_filt; ++q) { const int pl = (int)coef[q * 13 + 12]; SS_CHECK(pl >= 1 && pl < T, "ss_iir_filtfilt: the signal must be longer than padlen = %d", pl); max_pad = pl > max_pad ? pl : max_pad; }
    SS_CHECK(workspace_bytes >= ss_iir_filtfilt_workspace_bytes(T, C, max_pad), "ss_iir_filtfilt: workspace too small");
    const long long Temax = (long long)T + 2 * max_pad, nchmax = (Temax + FL_MIN - 1) / FL_MIN;
    double* bufA = (double*)workspace; double* bufB = bufA + Temax * C; double* P = bufB + Temax * C; double* S = P + nchmax * C * FN;
    const double* src = x; long long skip = 0;
    const int tpb = 256 / C * C;                 // whole (chunk, channel) groups per block
    for (int q = 0; q < n_filt; ++q) {
        Filt f;
        if (make_filt(coef + q * 13, q, f)) return 1;      // coefficients, chunk length L and A^L
        const int Te = T + 2 * f.padlen, nch = (Te + f.L - 1) / f.L;
        const long long tot = (long long)Te * C;
        int eg = (int)((tot + 255) / 256); if (eg > 4096) eg = 4096;
        double* E = src == bufA ? bufB : bufA;    // the extension never overwrites the buffer it reads
        double* F = E == bufA ? bufB : bufA;
        SS_LAUNCH(ext_kernel, dim3(eg), dim3(256), 0, stream, src, skip, E, T, C, f.padlen);
        const int cpb = tpb / C, blocks = (nch + cpb - 1) / cpb;
        for (int rev = 0; rev < 2; ++rev) {
            const double* in = rev ? F : E; double* out = rev ? E : F;
            SS_LAUNCH(chunk_state_kernel, dim3(blocks), dim3(256), 0, stream, in, Te, C, nch, rev, f, P);
            SS_LAUNCH(prefix_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, in, Te, C, nch, rev, f, (const double*)P, S);
            SS_LAUNCH(apply_kernel, dim3(blocks), dim3(256), 0, stream, in, out, Te, C, nch, rev, f, (const double*)S);
        }
        src = E; skip = f.padlen;                // the filtered signal sits in E[padlen : padlen + T]
    }
    { const long long tot = (long long)T * C; int g = (int)((tot + 255) / 256); if (g > 4096) g = 4096;
      SS_LAUNCH(crop_kernel, dim3(g), dim3(256), 0, stream, src, skip, y, T, C); }
    SS_LAUNCH_CHECK("ss_iir_filtfilt");
    return 0;
}

extern "C" int ss_linear_resample(const double* x, double* y, int T, int C, double old_freq, double new_freq, int T_out, void* stream)
{
    SS_CHECK(x && y, "ss_linear_resample: null pointer");
    SS_CHECK(T >= 1 && C >= 1 && T_out >= 0 && old_freq > 0 && new_freq > 0, "ss_linear_resample: bad sizes");
    if (T_out == 0) return 0;
    const long long tot = (long long)T_out * C; int g = (int)((tot + 255) / 256); if (g > 4096) g = 4096;
    SS_LAUNCH(resample_kernel, dim3(g), dim3(256), 0, stream, x, y, T, C, old_freq, new_freq, T_out);
    SS_LAUNCH_CHECK("ss_linear_resample");
    return 0;
}

// ---------------------------------------------------------------- ragged batch entry points
static long long rag_ext_rows(const int32_t* lengths, int R, int pad) { long long t = 0; for (int u = 0; u < R; ++u) t += (long long)lengths[u] + 2 * pad; return t; }
static long long rag_chunks(const int32_t* lengths, int R, int pad, int L) { long long n = 0; for (int u = 0; u < R; ++u) n += ((long long)lengths[u] + 2 * pad + L - 1) / L; return n; }

extern "C" int64_t ss_iir_filtfilt_batch_workspace_bytes(const int32_t* lengths_host, int R, int C, int max_padlen, int n_filt)
{
    if (!lengths_host || R <= 0 || C <= 0) return 0;
    const long long te = rag_ext_rows(lengths_host, R, max_padlen), nch = rag_chunks(lengths_host, R, max_padlen, FL_MIN);
    return (2 * te * C + 2 * nch * C * FN) * 8 + (long long)(n_filt + 1) * R * (long long)sizeof(RagRow) + 1024;
}

// x, y: packed (sum T_u, C) f64 (recording u = rows [sum T_<u, + T_u)); lengths_host: R ints (host memory, like coef)
extern "C" int ss_iir_filtfilt_batch(const double* x, double* y, const int32_t* lengths_host, int R, int C, int n_filt, const double* coef, void* workspace, int64_t workspace_bytes, void* stream)
{
    SS_CHECK(x && y && coef && workspace && lengths_host, "ss_iir_filtfilt_batch: null pointer");
    SS_CHECK(R > 0 && C > 0 && C <= 64 && n_filt >= 1 && n_filt <= 16, "ss_iir_filtfilt_batch: bad sizes");
    int max_pad = 0, min_len = lengths_host[0];
    for (int u = 0; u < R; ++u) min_len = lengths_host[u] < min_len ? lengths_host[u] : min_len;
    for (int q = 0; q < n_filt; ++q) { const int pl = (int)coef[q * 13 + 12]; SS_CHECK(pl >= 1 && pl < min_len, "ss_iir_filtfilt_batch: every signal must be longer than padlen = %d", pl); max_pad = pl > max_pad ? pl : max_pad; }
    SS_CHECK(workspace_bytes >= ss_iir_filtfilt_batch_workspace_bytes(lengths_host, R, C, max_pad, n_filt), "ss_iir_filtfilt_batch: workspace too small");
    const long long te_max = rag_ext_rows(lengths_host, R, max_pad), nch_max = rag_chunks(lengths_host, R, max_pad, FL_MIN);
    double* bufA = (double*)workspace; double* bufB = bufA + te_max * C; double* P = bufB + te_max * C; double* S = P + nch_max * C * FN;
    RagRow* tabs_dev = (RagRow*)(S + nch_max * C * FN);
    // tables of every filter (and the packed output offsets as table n_filt), built on the host, uploaded in one copy
    std::vector<RagRow> tabs((size_t)(n_filt + 1) * R);
    std::vector<Filt> filts(n_filt);
    long long total_rows = 0;
    for (int q = 0; q < n_filt; ++q) {
        if (make_filt(coef + q * 13, q, filts[q])) return 1;
        const Filt& f = filts[q];
        long long ext = 0, ch = 0, rows = 0;
        for (int u = 0; u < R; ++u) {
            RagRow& r = tabs[(size_t)q * R + u];
            r.T = lengths_host[u]; r.Te = r.T + 2 * f.padlen; r.ext_off = ext; r.chunk_off = ch; r.nch = (r.Te + f.L - 1) / f.L;
            r.src_off = q == 0 ? rows * C : tabs[(size_t)(q - 1) * R + u].ext_off + (long long)filts[q - 1].padlen * C;
            ext += r.Te * C; ch += r.nch; rows += r.T;
        }
        total_rows = rows;
    }
    { long long rows = 0; for (int u = 0; u < R; ++u) { RagRow& r = tabs[(size_t)n_filt * R + u]; memset(&r, 0, sizeof(r)); r.src_off = rows * C; r.T = lengths_host[u]; rows += r.T; } }
    if (ss_upload_table(tabs_dev, tabs.data(), tabs.size() * sizeof(RagRow), stream)) return 1;      // through pinned staging: `tabs` is a local
    const int tpb = 256 / C * C, cpb = tpb / C;
    const double* src = x;
    for (int q = 0; q < n_filt; ++q) {
        const Filt& f = filts[q];
        const RagRow* tab = tabs_dev + (size_t)q * R;
        const RagRow& last = tabs[(size_t)q * R + R - 1];
        const long long tot = last.ext_off + last.Te * C, nch = last.chunk_off + last.nch;
        int eg = (int)((tot + 255) / 256); if (eg > 8192) eg = 8192;
        double* E = src == bufA ? bufB : bufA;
        double* F = E == bufA ? bufB : bufA;
        SS_LAUNCH(ext_ragged_kernel, dim3(eg), dim3(256), 0, stream, src, E, tab, R, C, f.padlen, tot);
        const int blocks = (int)((nch + cpb - 1) / cpb);
        for (int rev = 0; rev < 2; ++rev) {
            const double* in = rev ? F : E; double* out = rev ? E : F;
            SS_LAUNCH(chunk_state_ragged_kernel, dim3(blocks), dim3(256), 0, stream, in, tab, R, C, nch, rev, f, P);
            SS_LAUNCH(prefix_ragged_kernel, dim3((R * C + 63) / 64), dim3(64), 0, stream, in, tab, R, C, rev, f, (const double*)P, S);
            SS_LAUNCH(apply_ragged_kernel, dim3(blocks), dim3(256), 0, stream, in, out, tab, R, C, nch, rev, f, (const double*)S);
        }
        src = E;
    }
    { const long long tot = total_rows * C; int g = (int)((tot + 255) / 256); if (g > 8192) g = 8192;
      SS_LAUNCH(crop_ragged_kernel, dim3(g), dim3(256), 0, stream, src, y, tabs_dev + (size_t)(n_filt - 1) * R, tabs_dev + (size_t)n_filt * R, R, C, filts[n_filt - 1].padlen, tot); }
    SS_LAUNCH_CHECK("ss_iir_filtfilt_batch");
    return 0;
}

// table_dev: int64 [R][4] = {first input row, T, first output row, T_out} per recording (device memory); x / y packed (rows, C) f64
extern "C" int ss_linear_resample_batch(const double* x, double* y, const int64_t* table_dev, int R, int C, double old_freq, double new_freq, int64_t total_out_rows, void* stream)
{
    SS_CHECK(x && y && table_dev, "ss_linear_resample_batch: null pointer");
    SS_CHECK(R >= 1 && C >= 1 && total_out_rows >= 0 && old_freq > 0 && new_freq > 0, "ss_linear_resample_batch: bad sizes");
    if (total_out_rows == 0) return 0;
    const long long tot = total_out_rows * C; int g = (int)((tot + 255) / 256); if (g > 8192) g = 8192;
    SS_LAUNCH(resample_ragged_kernel, dim3(g), dim3(256), 0, stream, x, y, (const long long*)table_dev, R, C, old_freq, new_freq, (long long)total_out_rows);
    SS_LAUNCH_CHECK("ss_linear_resample_batch");
    return 0;
}
