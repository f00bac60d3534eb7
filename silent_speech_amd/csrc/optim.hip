// optim.hip -- fused AdamW over the flat parameter arena (reference transduction_model.py:178,210:
// torch.optim.AdamW, betas (.9,.999), eps 1e-8, decoupled weight decay FLAGS.l2) and small elementwise
// helpers of the mel-target path (data_utils.py:51,57).  HBM-bound: 16 B read + 12 B written per parameter.
#include "common.h"
#include "silent_speech_hip.h"

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float grad_scale)
{
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * grad_scale;
            float x = pp[e] * (1.f - lr * wd);
            mm[e] = beta1 * mm[e] + (1.f - beta1) * gr;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pp[e] = x - (lr / bc1) * (mm[e] / denom);
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gr = g[i] * grad_scale;
        float x = p[i] * (1.f - lr * wd);
        float mm = beta1 * m[i] + (1.f - beta1) * gr, vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mm; v[i] = vv;
        p[i] = x - (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}

extern "C" int ss_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, float grad_scale, void* stream)
{
    SS_CHECK(p && g && m && v, "ss_adamw_step: null pointer");
    SS_CHECK(step >= 1, "ss_adamw_step: step is 1-based");
    SS_CHECK(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "ss_adamw_step: arenas must be 16-byte aligned");
    if (n <= 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = ((n >> 2) + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    SS_LAUNCH(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    SS_LAUNCH_CHECK("ss_adamw_step");
    return 0;
}

// ---------------------------------------------------------------- f32 -> compute-dtype cast of a flat range (weight shadow copies)
template <class TO>
__global__ void cast_kernel(const float* __restrict__ in, TO* __restrict__ out, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) stf(out + i, in[i]);
}
extern "C" int ss_cast_f32(const float* in, void* out, int out_dtype, int64_t n, void* stream)
{
    SS_CHECK(in && out, "ss_cast_f32: null pointer");
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (out_dtype == SS_BF16) SS_LAUNCH(cast_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, in, (bf16_t*)out, (long long)n);
    else SS_LAUNCH(cast_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, in, (float*)out, (long long)n);
    SS_LAUNCH_CHECK("ss_cast_f32");
    return 0;
}

// ---------------------------------------------------------------- soft clipping of the input pipeline ("next" row N3)
// read_emg.py:227-228  raw_emg = raw_emg / 20; raw_emg = 50 * tanh(raw_emg / 50)      (pre_div = 20, limit = 50)
// read_emg.py:233      emg = 8 * tanh(emg / 8) after FeatureNormalizer.normalize         (pre_div = 1,  limit = 8)
// optional per-column affine first (FeatureNormalizer: (x - mean[c]) / std[c], data_utils.py:228-231).
__global__ void soft_clip_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, int C, const float* __restrict__ mean, const float* __restrict__ stdv,
                                 float pre_div, float limit)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (mean) { const int c = (int)(i % C); v = (v - mean[c]) / stdv[c]; }
        v = v / pre_div;
        out[i] = limit > 0.f ? limit * tanhf(v / limit) : v;
    }
}
extern "C" int ss_soft_clip(const float* x, float* out, int64_t n, int C, const float* mean, const float* stdv, float pre_div, float limit, void* stream)
{
    SS_CHECK(x && out, "ss_soft_clip: null pointer");
    SS_CHECK((mean == nullptr) == (stdv == nullptr) && (!mean || C > 0) && pre_div != 0.f, "ss_soft_clip: bad normaliser arguments");
    if (n <= 0) return 0;
    long long blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    SS_LAUNCH(soft_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, out, (long long)n, C, mean, stdv, pre_div, limit);
    SS_LAUNCH_CHECK("ss_soft_clip");
    return 0;
}

// ---------------------------------------------------------------- mel targets: reflect padding and |STFT|
// data_utils.py:51  F.pad(y, (p, p), mode='reflect')
__global__ void reflect_pad_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int L, int pad, long long ld_out)
{
    const int Lp = L + 2 * pad;
    const long long total = (long long)B * Lp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / Lp), t = (int)(i - (long long)b * Lp) - pad;
        const int s = t < 0 ? -t : (t >= L ? 2 * (L - 1) - t : t);
        out[(long long)b * ld_out + t + pad] = y[(long long)b * L + s];
    }
}
extern "C" int ss_reflect_pad(const float* y, float* out, int B, int L, int pad, int64_t ld_out, void* stream)
{
    SS_CHECK(y && out, "ss_reflect_pad: null pointer");
    SS_CHECK(B > 0 && L > pad && pad >= 0 && ld_out >= L + 2 * pad, "ss_reflect_pad: bad sizes (reflect needs pad < L)");
    long long total = (long long)B * (L + 2 * pad), blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    SS_LAUNCH(reflect_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, out, B, L, pad, (long long)ld_out);
    SS_LAUNCH_CHECK("ss_reflect_pad");
    return 0;
}

// The same for a RAGGED batch (one launch for every utterance of a batch): row b of `out` holds the clipped (data_utils.py:76
// np.clip(audio, -1, 1)) and reflect-padded signal of utterance b, L_b + 2 pad samples, zeros behind it up to ld_out.
__global__ void reflect_pad_ragged_kernel(const float* __restrict__ y, const long long* __restrict__ offs, const int* __restrict__ lens, float* __restrict__ out,
                                          int B, int pad, long long ld_out, int clip)
{
    const long long total = (long long)B * ld_out;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / ld_out), j = (int)(i - (long long)b * ld_out);
        const int L = lens[b], t = j - pad;
        float v = 0.f;
        if (j < L + 2 * pad) {
            const int s = t < 0 ? -t : (t >= L ? 2 * (L - 1) - t : t);
            v = y[offs[b] + s];
            if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        }
        out[i] = v;
    }
}
extern "C" int ss_reflect_pad_ragged(const float* y, const int64_t* offsets_dev, const int32_t* lengths_dev, float* out, int B, int min_len, int pad, int64_t ld_out, int clip, void* stream)
{
    SS_CHECK(y && out && offsets_dev && lengths_dev, "ss_reflect_pad_ragged: null pointer");
    SS_CHECK(B > 0 && pad >= 0 && min_len > pad && ld_out >= min_len + 2 * pad, "ss_reflect_pad_ragged: bad sizes (reflect needs pad < the shortest signal)");
    long long total = (long long)B * ld_out, blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    SS_LAUNCH(reflect_pad_ragged_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, (const long long*)offsets_dev, (const int*)lengths_dev, out, B, pad, (long long)ld_out, clip);
    SS_LAUNCH_CHECK("ss_reflect_pad_ragged");
    return 0;
}

// data_utils.py:56-57  sqrt(re^2 + im^2 + 1e-9); spec rows hold [re(0..nb-1) | im(0..nb-1)] with row stride ld_spec
__global__ void stft_magnitude_kernel(const float* __restrict__ spec, long long ld_spec, int nb, float* __restrict__ mag, long long ld_mag, int rows)
{
    const long long total = (long long)rows * ld_mag;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld_mag), k = (int)(i - (long long)r * ld_mag);
        float v = 0.f;
        if (k < nb) { const float re = spec[(long long)r * ld_spec + k], im = spec[(long long)r * ld_spec + nb + k]; v = sqrtf(re * re + im * im + 1e-9f); }
        mag[i] = v;
    }
}
extern "C" int ss_stft_magnitude(const float* spec, int64_t ld_spec, int n_bins, float* mag, int64_t ld_mag, int rows, void* stream)
{
    SS_CHECK(spec && mag, "ss_stft_magnitude: null pointer");
    SS_CHECK(ld_spec >= 2 * n_bins && ld_mag >= n_bins, "ss_stft_magnitude: bad leading dimensions");
    if (rows <= 0) return 0;
    long long total = (long long)rows * ld_mag, blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    SS_LAUNCH(stft_magnitude_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, spec, (long long)ld_spec, n_bins, mag, (long long)ld_mag, rows);
    SS_LAUNCH_CHECK("ss_stft_magnitude");
    return 0;
}

// ---------------------------------------------------------------- packing utterances into fixed-length rows (data_utils.py:158-167)
// combine_fixed_length = concatenate the utterances along time and zero-pad to a whole number of rows.  In bytes that is one gather:
// output granule g belongs to the utterance u with off[u] <= g * G < off[u + 1] (binary search over the cumulative byte offsets) or to
// the zero tail.  One launch instead of torch.cat over ~40 inputs + a padding tensor; G = 16, 8, 4 or 1 bytes, whatever every
// utterance size and pointer is a multiple of.
template <class V>
__global__ void concat_pad_kernel(const unsigned long long* __restrict__ ptrs, const long long* __restrict__ offs, int n, V* __restrict__ out, long long total_granules)
{
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total_granules; g += (long long)gridDim.x * blockDim.x) {
        const long long byte = g * (long long)sizeof(V);
        V v;
        memset(&v, 0, sizeof(V));
        if (byte < offs[n]) {
            int lo = 0, hi = n - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= byte) lo = mid; else hi = mid - 1; }
            v = *(const V*)((const unsigned char*)(uintptr_t)ptrs[lo] + (byte - offs[lo]));
        }
        out[g] = v;
    }
}
// table_dev: [n] source pointers followed by [n + 1] cumulative byte offsets (int64 each); granule: 16 / 8 / 4 / 1
extern "C" int ss_concat_pad(const void* table_dev, int n, void* out, int64_t total_bytes, int granule, void* stream)
{
    SS_CHECK(table_dev && out && n >= 1 && total_bytes >= 0, "ss_concat_pad: bad arguments");
    SS_CHECK((granule == 16 || granule == 8 || granule == 4 || granule == 1) && total_bytes % granule == 0, "ss_concat_pad: granule must be 16, 8, 4 or 1 and divide the output size");
    if (total_bytes == 0) return 0;
    const unsigned long long* ptrs = (const unsigned long long*)table_dev;
    const long long* offs = (const long long*)table_dev + n;
    const long long tg = total_bytes / granule;
    long long blocks = (tg + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (granule == 16) SS_LAUNCH(SS_KERNEL(concat_pad_kernel<u32x4>), dim3((unsigned)blocks), dim3(256), 0, stream, ptrs, offs, n, (u32x4*)out, tg);
    else if (granule == 8) SS_LAUNCH(SS_KERNEL(concat_pad_kernel<u32x2>), dim3((unsigned)blocks), dim3(256), 0, stream, ptrs, offs, n, (u32x2*)out, tg);
    else if (granule == 4) SS_LAUNCH(SS_KERNEL(concat_pad_kernel<unsigned>), dim3((unsigned)blocks), dim3(256), 0, stream, ptrs, offs, n, (unsigned*)out, tg);
    else SS_LAUNCH(SS_KERNEL(concat_pad_kernel<unsigned char>), dim3((unsigned)blocks), dim3(256), 0, stream, ptrs, offs, n, (unsigned char*)out, tg);
    SS_LAUNCH_CHECK("ss_concat_pad");
    return 0;
}

// ---------------------------------------------------------------- ss_counters_add
struct CounterPtrs { long long* p[16]; };
__global__ void counters_add_kernel(CounterPtrs c, int n, long long delta) { const int i = threadIdx.x; if (i < n && c.p[i]) c.p[i][0] += delta; }
extern "C" int ss_counters_add(int n, int64_t** counters, int64_t delta, void* stream)
{
    SS_CHECK(counters && n >= 0 && n <= 16, "ss_counters_add: 0..16 counters");
    if (n == 0) return 0;
    CounterPtrs c; memset(&c, 0, sizeof(c));
    for (int i = 0; i < n; ++i) c.p[i] = (long long*)counters[i];
    SS_LAUNCH(counters_add_kernel, dim3(1), dim3(64), 0, stream, c, n, (long long)delta);
    SS_LAUNCH_CHECK("ss_counters_add");
    return 0;
}
