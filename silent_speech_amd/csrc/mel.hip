// mel.hip -- log-mel target extraction as an LDS FFT (reference data_utils.py:39-62 mel_spectrogram; call site :76-78 load_audio:
//     y    = np.clip(audio, -1, 1)                                                                              (:76)
//     y    = F.pad(y, ((n_fft - hop) / 2,) * 2, mode = 'reflect')                                               (:51)
//     spec = torch.stft(y, n_fft = 1024, hop 256, hann(1024), center = False, onesided)                        (:54-55)
//     spec = sqrt(re^2 + im^2 + 1e-9)                                                                           (:57)
//     spec = log(clamp(mel_basis @ spec, min = 1e-5))                                                           (:59-60)
// Rounds 1-5 ran the STFT as a dense windowed-DFT GEMM (2.1 MFLOP per frame on exact-f32 MFMA tiles: 0.5 % of the 1344 B / frame HBM roofline,
// SURVEY 8d) behind a clip + reflect-pad pass into a zeroed buffer.  Here ONE WAVE owns a frame from the caller's samples to the 80 log-mel values:
//   * the frame's 1024 samples are read straight from the (ragged batch of) signals -- clip and reflection are index arithmetic on the few frames
//     that touch a signal's ends, the padded copy does not exist -- and packed into 512 complex numbers z[n] = x[2n] + i x[2n+1], windowed on the way in;
//   * 512 = 8 x 8 x 8: three radix-8 passes in registers (lane = one butterfly of 8; complex numbers are float pairs, so adds / twiddle products
//     are v_pk_add_f32 / v_pk_fma_f32), two transposes through the wave's own 4.6 KiB LDS tile (8-byte elements, pitches 72 / 9 and a skew of the
//     last layout: every access of a pass is conflict-free), twiddles W_512^(r k0), W_64^(r0 k0') per lane in registers;
//   * the spectrum of the real signal is recombined from Z[k] and conj Z[512 - k] (k = lane + 64 j), the magnitudes go to LDS;
//   * the mel filterbank is SPARSE (a band is a triangle over <= ~35 bins): the host deals the bands to the lanes so that every lane has about the
//     same number of bins (two bands per lane at most), a lane sums its bands from the LDS magnitudes (runs padded to 4 with zero weights), log-clamp, store.
// No barrier inside a frame (LDS serves a wave's operations in order; wave_lds_sync is a compiler-level fence), ~26 kFLOP per frame.
// A wave walks frames g, g + (waves in the grid), ...: its 60 per-lane constants (window, twiddles) are computed once.
// n_fft is fixed at 1024 (the only value the reference uses); other sizes keep the GEMM formulation (data_utils._stft_logmel).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>

namespace {
constexpr int MF_N = 1024, MF_H = 512, MF_WAVES = 4;
constexpr int MF_P1 = 72, MF_P2 = 9;                       // LDS pitches of the two transposes (complex elements)
constexpr int MF_TILE = 8 * MF_P1;                         // complex elements of a wave's tile
constexpr int MF_MAXW = 4096, MF_MAXB = 128;               // packed filterbank weights / bands held in LDS
constexpr int MF_MAGPAD = 8;                               // zeroed magnitudes behind bin 512 (runs are padded to multiples of 4)

typedef float cpx __attribute__((ext_vector_type(2)));     // (re, im)
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { const cpx bs = {-b[1], b[0]}; return a[0] * b + a[1] * bs; }      // two packed operations
__device__ __forceinline__ cpx mul_mi(cpx a) { return cpx{a[1], -a[0]}; }                                               // a * (-i)
// e^(-2 pi i num / den)
__device__ __forceinline__ cpx twiddle(int num, int den) {
#if defined(SS_EMU)
    const double a = -2.0 * 3.14159265358979323846 * (double)num / (double)den;
    return cpx{(float)cos(a), (float)sin(a)};
#else
    float s, c;
    sincospif(2.0f * (float)num / (float)den, &s, &c);      // the argument is exact (den a power of two)
    return cpx{c, -s};
#endif
}
__device__ __forceinline__ float fast_sqrt(float x) {
#if defined(SS_EMU)
    return sqrtf(x);
#else
    return __builtin_amdgcn_sqrtf(x);                        // v_sqrt_f32, 1 ulp; the argument is >= 1e-9 (no denormal scaling needed)
#endif
}
// X[k] = sum_q x[q] e^(-2 pi i q k / 8), in place, natural order: one radix-2 split (decimation in frequency) and two 4-point transforms
__device__ __forceinline__ void dft4(cpx c0, cpx c1, cpx c2, cpx c3, cpx& y0, cpx& y1, cpx& y2, cpx& y3) {
    const cpx e0 = c0 + c2, e1 = c0 - c2, o0 = c1 + c3, o1 = mul_mi(c1 - c3);
    y0 = e0 + o0; y2 = e0 - o0; y1 = e1 + o1; y3 = e1 - o1;
}
__device__ __forceinline__ void dft8(cpx (&x)[8]) {
    constexpr float H = 0.70710678118654752f;
    cpx a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = x[q] + x[q + 4]; b[q] = x[q] - x[q + 4]; }
    b[1] = cpx{b[1][0] + b[1][1], b[1][1] - b[1][0]} * H;            // (1 - i) / sqrt 2
    b[2] = mul_mi(b[2]);                                               // -i
    b[3] = cpx{b[3][1] - b[3][0], -(b[3][0] + b[3][1])} * H;         // (-1 - i) / sqrt 2
    dft4(a[0], a[1], a[2], a[3], x[0], x[2], x[4], x[6]);
    dft4(b[0], b[1], b[2], b[3], x[1], x[3], x[5], x[7]);
}
// position of Z[k] in the tile: a skew of 4 elements per 32, so that the pass-3 stores (k = k0 + 8 k0' + 64 k1' over the lanes (k0, k0')) and the
// recombination's reads (k = lane + 64 j, and 512 - k) both touch 32 distinct 8-byte bank pairs per half wave
__device__ __forceinline__ int zpos(int k) { return k + 4 * (k >> 5); }
static_assert(MF_H - 1 + 4 * ((MF_H - 1) >> 5) < MF_TILE, "the skewed spectrum fits the tile");
}

// (3 workgroups per CU: 168 registers; left alone hipcc takes 224 -- two waves per SIMD -- and the kernel runs 15-30 % slower: tools/mel_probe.py)
__global__ __launch_bounds__(MF_WAVES * 64, 3) void stft_logmel_fft_kernel(const float* __restrict__ y, const long long* __restrict__ offs, const int* __restrict__ lens, long long uniform_len,
                                                                         int B, int F, int pad, int clip, int hop, const float* __restrict__ window,
                                                                         const int* __restrict__ band_lo, const int* __restrict__ band_cnt, const int* __restrict__ band_off,
                                                                         const float* __restrict__ band_w, const int* __restrict__ lane_bands, int n_mels, int n_w, float log_clamp,
                                                                         float* __restrict__ out, long long sb, long long sf, long long sm)
{
    __shared__ cpx tile_s[MF_WAVES][MF_TILE];
    __shared__ float mag_s[MF_WAVES][MF_H + MF_MAGPAD];
    __shared__ float w_s[MF_MAXW];
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_uniform(tid >> 6);
    for (int i = tid; i < n_w; i += MF_WAVES * 64) w_s[i] = band_w[i];
    cpx* tl = tile_s[wv]; float* mg = mag_s[wv];
    if (lane < MF_MAGPAD) mg[MF_H + lane] = 0.f;                 // bins behind 512 (index 512 itself is written per frame): read against zero weights only
    __syncthreads();

    // ---- per-lane constants
    const int r = lane, k0l = lane >> 3, r0 = lane & 7;            // pass 1: lane = r; pass 2: lane = (k0, r0); pass 3: lane = (k0, k0')
    cpx win[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int n = r + 64 * q; win[q] = cpx{window[2 * n], window[2 * n + 1]}; }
    cpx tw1[8], tw2[8], tw3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { tw1[k] = twiddle(r * k, MF_H); tw2[k] = twiddle(r0 * k, 64); tw3[k] = twiddle(lane + 64 * k, MF_N); }
    const cpx tw_nyq = twiddle(MF_H, MF_N);
    int bm[2], blo[2], bcnt[2], boff[2];                           // this lane's (at most two) bands
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        bm[it] = lane_bands[it * 64 + lane];
        const int m = bm[it] >= 0 ? bm[it] : 0;
        blo[it] = band_lo[m]; bcnt[it] = bm[it] >= 0 ? band_cnt[m] : 0; boff[it] = band_off[m];
    }

    const long long total = (long long)B * F;
    const long long wave0 = (long long)blockIdx.x * MF_WAVES + wv, nwaves = (long long)gridDim.x * MF_WAVES;
    for (long long g = wave0; g < total; g += nwaves) {
        const int b = (int)(g / F), f = (int)(g - (long long)b * F);
        const float* x = y + (offs ? offs[b] : (long long)b * uniform_len);
        const int L = wave_uniform(offs ? lens[b] : (int)uniform_len);
        const int i0 = wave_uniform(f * hop - pad);                // signal index of the frame's first sample
        // ---- pass 1: z[r + 64 q], q = 0 .. 7 -> radix 8 over q -> T[k0][r] = W_512^(r k0) sum_q z[r + 64 q] W_8^(q k0)
        cpx v[8];
        if (i0 >= 0 && i0 + MF_N <= L) {                           // (wave-uniform) the frame lies inside the signal
#pragma unroll
            for (int q = 0; q < 8; ++q) { const float* p = x + i0 + 2 * (r + 64 * q); v[q] = cpx{p[0], p[1]}; }
        } else {                                                   // reflection about the first / last sample (:51), zeros behind the padded signal
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int t = i0 + 2 * (r + 64 * q) + h;
                    int s = t < 0 ? -t : (t >= L ? 2 * (L - 1) - t : t);
                    const bool in = t >= -pad && t < L + pad;
                    s = s < 0 ? 0 : (s > L - 1 ? L - 1 : s);
                    v[q][h] = in ? x[s] : 0.f;
                }
            }
        }
        if (clip) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { v[q][0] = fminf(fmaxf(v[q][0], -1.f), 1.f); v[q][1] = fminf(fmaxf(v[q][1], -1.f), 1.f); }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= win[q];
        dft8(v);
        wave_lds_sync();                                           // (the previous frame's reads of the tile are behind us)
#pragma unroll
        for (int k = 0; k < 8; ++k) tl[k * MF_P1 + r] = k ? cmul(v[k], tw1[k]) : v[k];
        wave_lds_sync();
        // ---- pass 2: lane (k0, r0): T[k0][r0 + 8 r1], r1 = 0 .. 7 -> radix 8 over r1 -> U[k0][r0][k0'] = W_64^(r0 k0') sum_r1 T W_8^(r1 k0')
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) v[r1] = tl[k0l * MF_P1 + r0 + 8 * r1];
        dft8(v);
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) tl[k0l * MF_P1 + r0 * MF_P2 + k] = k ? cmul(v[k], tw2[k]) : v[k];
        wave_lds_sync();
        // ---- pass 3: lane (k0, k0' = lane & 7): U[k0][r0][k0'], r0 = 0 .. 7 -> radix 8 over r0 -> Z[k0 + 8 k0' + 64 k1']
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = tl[k0l * MF_P1 + q * MF_P2 + r0];
        dft8(v);
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) tl[zpos(k0l + 8 * r0 + 64 * k)] = v[k];
        wave_lds_sync();
        // ---- the real signal's bins k = lane + 64 j (and k = 512): X[k] = E + W_1024^k O, E = (Z[k] + conj Z[512 - k]) / 2, O = (Z[k] - conj Z[512 - k]) / (2 i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j, km = (MF_H - k) & (MF_H - 1);
            const cpx a = tl[zpos(k)], c0 = tl[zpos(km)], c = cpx{c0[0], -c0[1]};
            const cpx e = 0.5f * (a + c), o = mul_mi(0.5f * (a - c));
            const cpx xk = e + cmul(o, tw3[j]);
            mg[k] = fast_sqrt(xk[0] * xk[0] + xk[1] * xk[1] + 1e-9f);
        }
        if (lane == 0) {                                           // k = 512: Z[512] = Z[0]: E = re Z[0], O = im Z[0]
            const cpx a = tl[0];
            const cpx xk = cpx{a[0], 0.f} + cmul(cpx{a[1], 0.f}, tw_nyq);
            mg[MF_H] = fast_sqrt(xk[0] * xk[0] + xk[1] * xk[1] + 1e-9f);
        }
        wave_lds_sync();
        // ---- sparse mel filterbank + log clamp
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float acc0 = 0.f, acc1 = 0.f;
            const float* wp = w_s + boff[it]; const float* mp = mg + blo[it];
            for (int j = 0; j < bcnt[it]; j += 4) {                // runs are padded to multiples of 4 (zero weights)
                acc0 += wp[j] * mp[j]; acc1 += wp[j + 1] * mp[j + 1];
                acc0 += wp[j + 2] * mp[j + 2]; acc1 += wp[j + 3] * mp[j + 3];
            }
            if (bm[it] >= 0) out[(long long)b * sb + (long long)f * sf + (long long)bm[it] * sm] = logf(fmaxf(acc0 + acc1, log_clamp));
        }
    }
    (void)n_mels;
}

extern "C" int ss_stft_logmel_fft(const float* y, const int64_t* offsets_dev, const int32_t* lengths_dev, int64_t uniform_len, int B, int F, int pad, int clip,
                                  int n_fft, int hop, const float* window,
                                  const int32_t* band_lo, const int32_t* band_cnt, const int32_t* band_off, const float* band_w, const int32_t* lane_bands, int n_mels, int n_w,
                                  float log_clamp, float* out, int64_t stride_b, int64_t stride_f, int64_t stride_m, void* stream)
{
    SS_CHECK(y && window && band_lo && band_cnt && band_off && band_w && lane_bands && out, "ss_stft_logmel_fft: null pointer");
    SS_CHECK((offsets_dev != nullptr) == (lengths_dev != nullptr), "ss_stft_logmel_fft: offsets and lengths come together (ragged batch) or not at all (rows of uniform_len samples)");
    SS_CHECK(n_fft == MF_N, "ss_stft_logmel_fft: n_fft must be %d (other sizes: the DFT-GEMM path)", MF_N);
    SS_CHECK(hop > 0 && pad >= 0 && (offsets_dev || uniform_len > pad), "ss_stft_logmel_fft: bad hop %d / pad %d / length %lld (reflect needs pad < length)", hop, pad, (long long)uniform_len);
    SS_CHECK(B >= 0 && F >= 0, "ss_stft_logmel_fft: negative sizes");
    SS_CHECK(n_mels >= 1 && n_mels <= MF_MAXB && n_w >= 0 && n_w <= MF_MAXW, "ss_stft_logmel_fft: %d bands / %d packed weights exceed the LDS tables (%d / %d)", n_mels, n_w, MF_MAXB, MF_MAXW);
    const long long total = (long long)B * F;
    if (total == 0) return 0;
    long long blocks = (total + MF_WAVES - 1) / MF_WAVES;
    const long long cap = (long long)ss_cu_count(2) * 3;         // 3 workgroups of 4 waves per CU
    if (blocks > cap) blocks = cap;
    SS_LAUNCH(stft_logmel_fft_kernel, dim3((unsigned)blocks), dim3(MF_WAVES * 64), 0, stream, y, (const long long*)offsets_dev, lengths_dev, (long long)uniform_len, B, F, pad, clip, hop, window,
              band_lo, band_cnt, band_off, band_w, lane_bands, n_mels, n_w, log_clamp, out, (long long)stride_b, (long long)stride_f, (long long)stride_m);
    SS_LAUNCH_CHECK("ss_stft_logmel_fft");
    return 0;
}
