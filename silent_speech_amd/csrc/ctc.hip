// ctc.hip -- CTC loss of the recognition trainer on gfx950 (reference recognition_model.py:96-101):
//     pred = log_softmax(model(...), 2); pred = pad_sequence(decollate_tensor(pred, lengths))
//     loss = F.ctc_loss(pred, y, lengths, text_int_lengths, blank=n_chars)            (reduction 'mean')
// The kernels work on the PACKED frame layout the encoder produces (utterance u = frames [frame0, frame0+T) of the
// flattened (rows*200, V) logits, data_utils.py:159-179), so neither the decollate/pad copy nor the (T_max, N, V)
// log-prob tensor exists.  log_softmax is folded in through the per-frame log-sum-exp (ss_frame_lse).
//   alpha/beta : one workgroup per (utterance, direction); a thread owns up to 4 of the 2S+1 extended states, the
//                previous column lives in LDS, 32 frames of log-probs are staged in LDS at a time.  beta is the same
//                recursion on the reversed label string and reversed time.
//   gradient   : one wave per packed frame: occupancy per class via LDS bins, then
//                d loss / d logit[t][c] = (softmax[t][c] - exp(log occ[t][c] + nll - logp[t][c])) / (max(S,1) * N)
//                which is ATen's ctc_loss backward composed with log_softmax's (frames outside any utterance get 0).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>

namespace {
constexpr int CD = 5;                      // descriptor row: frame0, T, target0, S, workspace offset (floats)
constexpr int CTC_THREADS = 256, CTC_NS = 4, CTC_FC = 32;

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
}

__global__ __launch_bounds__(CTC_THREADS) void ctc_alpha_beta_kernel(const float* __restrict__ logits, long long ld, int V, int blank, const float* __restrict__ lse,
                                                                     const long long* __restrict__ desc, const int* __restrict__ targets,
                                                                     float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll, int sp_cap)
{
    SS_DYN_SMEM(smem);
    const int u = blockIdx.x; const bool rev = blockIdx.y != 0;
    const long long f0 = desc[u * CD + 0], T = desc[u * CD + 1], g0 = desc[u * CD + 2], S = desc[u * CD + 3], w0 = desc[u * CD + 4];
    const int SP = (int)(2 * S + 1);
    float* col[2] = {(float*)smem, (float*)smem + (sp_cap + 2)};        // two columns, each with 2 leading -inf pads
    float* lpc = (float*)smem + 2 * (sp_cap + 2);                       // [CTC_FC][V] staged log-probs
    float* out = (rev ? beta : alpha) + w0;
    const int tid = threadIdx.x;
    if (T <= 0) { if (!rev && tid == 0) nll[u] = S == 0 ? 0.f : INFINITY; return; }

    int lab[CTC_NS]; bool skip[CTC_NS];
#pragma unroll
    for (int k = 0; k < CTC_NS; ++k) {
        const int s = tid + k * CTC_THREADS;                             // logical state (reversed string when rev)
        lab[k] = blank; skip[k] = false;
        if (s < SP && (s & 1)) {
            const int j = s >> 1;
            const int c = targets[g0 + (rev ? S - 1 - j : j)];
            lab[k] = c;
            if (j >= 1) skip[k] = targets[g0 + (rev ? S - j : j - 1)] != c;
        }
    }
    if (tid < 2) { col[0][tid] = -INFINITY; col[1][tid] = -INFINITY; }

    int cur = 0;
    for (long long tl = 0; tl < T; ++tl) {
        const int fi = (int)(tl % CTC_FC);
        if (fi == 0) {                                                    // stage the next CTC_FC frames of log-probs
            const long long left = T - tl; const int nf = left < CTC_FC ? (int)left : CTC_FC;
            for (int i = tid; i < nf * V; i += CTC_THREADS) {
                const int f = i / V, v = i - f * V;
                const long long fr = f0 + (rev ? T - 1 - (tl + f) : tl + f);
                lpc[f * V + v] = logits[fr * ld + v] - lse[fr];
            }
            __syncthreads();
        }
        const long long tp = rev ? T - 1 - tl : tl;
        const float* prev = col[cur ^ 1]; float* now = col[cur];
#pragma unroll
        for (int k = 0; k < CTC_NS; ++k) {
            const int s = tid + k * CTC_THREADS;
            if (s < SP) {
                float v;
                if (tl == 0) v = s < 2 ? lpc[lab[k]] : -INFINITY;
                else v = lse3(prev[s + 2], prev[s + 1], skip[k] ? prev[s] : -INFINITY) + lpc[fi * V + lab[k]];
                now[s + 2] = v;
                out[tp * SP + (rev ? SP - 1 - s : s)] = v;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (!rev && tid == 0) {
        const float* last = col[cur ^ 1];
        const float l = lse3(last[SP - 1 + 2], SP > 1 ? last[SP - 2 + 2] : -INFINITY, -INFINITY);
        nll[u] = -l;
    }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, long long ld, int V, int blank, const float* __restrict__ lse,
                                                       const long long* __restrict__ desc, int n_utt, long long rows, const int* __restrict__ targets,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ nll,
                                                       float inv_n, float* __restrict__ dlogits, float* __restrict__ loss)
{
    SS_DYN_SMEM(smem);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    float* bins = (float*)smem + w * V;
    if (blockIdx.x == 0 && threadIdx.x == 0) {                           // mean over utterances of nll / max(S, 1), in a fixed order
        float s = 0.f;
        for (int u = 0; u < n_utt; ++u) { const long long S = desc[u * CD + 3]; s += nll[u] / (float)(S > 1 ? S : 1); }
        loss[0] = s * inv_n;
    }
    for (long long r = (long long)blockIdx.x * wpb + w; r < rows; r += (long long)gridDim.x * wpb) {
        int lo = 0, hi = n_utt;                                           // last utterance starting at or before r
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (desc[mid * CD] <= r) lo = mid; else hi = mid; }
        const long long f0 = desc[lo * CD], T = desc[lo * CD + 1], g0 = desc[lo * CD + 2], S = desc[lo * CD + 3], w0 = desc[lo * CD + 4];
        float* d = dlogits + r * ld;
        if (n_utt == 0 || r < f0 || r >= f0 + T) { for (int c = lane; c < ld; c += 64) d[c] = 0.f; continue; }
        const int SP = (int)(2 * S + 1);
        const float* a = alpha + w0 + (r - f0) * SP; const float* b = beta + w0 + (r - f0) * SP;
        float m = -INFINITY;
        for (int s = lane; s < SP; s += 64) m = fmaxf(m, a[s] + b[s]);
        m = wave_max(m);
        for (int c = lane; c < V; c += 64) bins[c] = 0.f;
        wave_lds_sync();
        for (int s = lane; s < SP; s += 64) {
            const float v = a[s] + b[s];
            if (v > -INFINITY) atomicAdd(&bins[(s & 1) ? targets[g0 + (s >> 1)] : blank], expf(v - m));
        }
        wave_lds_sync();
        const float L = lse[r], nl = nll[lo], sc = inv_n / (float)(S > 1 ? S : 1);
        for (int c = lane; c < ld; c += 64) {
            float g = 0.f;
            if (c < V) {
                const float lp = logits[r * ld + c] - L, occ = bins[c];
                g = expf(lp) - (occ > 0.f ? expf(logf(occ) + m + nl - lp) : 0.f);
                if (nl == INFINITY) g = NAN;                        // no valid alignment: ATen's exp(-inf + inf - lp), stated explicitly
            }
            d[c] = g * sc;
        }
        wave_lds_sync();
    }
}

extern "C" int ss_ctc_loss(const float* logits, int64_t ld, int V, int blank, const float* lse, const int64_t* desc, int n_utt, int max_target_len,
                           int64_t rows, const int32_t* targets, float* alpha_ws, float* beta_ws, float* nll, float* dlogits, float* loss, void* stream)
{
    SS_CHECK(logits && lse && dlogits && loss, "ss_ctc_loss: null pointer");
    SS_CHECK(V >= 1 && V <= 4096 && blank >= 0 && blank < V && ld >= V, "ss_ctc_loss: bad class count %d / blank %d / row stride %lld", V, blank, (long long)ld);
    SS_CHECK(n_utt >= 0 && max_target_len >= 0, "ss_ctc_loss: negative sizes");
    const int sp_cap = 2 * max_target_len + 1;
    SS_CHECK(sp_cap <= CTC_NS * CTC_THREADS, "ss_ctc_loss: target length %d exceeds the %d-label limit", max_target_len, (CTC_NS * CTC_THREADS - 1) / 2);
    if (n_utt > 0) {
        SS_CHECK(desc && alpha_ws && beta_ws && nll, "ss_ctc_loss: null workspace");
        SS_CHECK(targets || max_target_len == 0, "ss_ctc_loss: null targets");
        const size_t smem = sizeof(float) * (2 * (size_t)(sp_cap + 2) + (size_t)CTC_FC * V);
        SS_CHECK(smem <= 160 * 1024, "ss_ctc_loss: %zu bytes of LDS needed", smem);
        SS_LAUNCH(ctc_alpha_beta_kernel, dim3(n_utt, 2), dim3(CTC_THREADS), smem, stream, logits, (long long)ld, V, blank, lse, (const long long*)desc, targets,
                  alpha_ws, beta_ws, nll, sp_cap);
        SS_LAUNCH_CHECK("ss_ctc_loss(alpha/beta)");
    }
    {
        long long blocks = (rows + 3) / 4; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
        SS_LAUNCH(ctc_grad_kernel, dim3((unsigned)blocks), dim3(256), sizeof(float) * 4 * V, stream, logits, (long long)ld, V, blank, lse, (const long long*)desc, n_utt,
                  (long long)rows, targets, alpha_ws, beta_ws, nll, n_utt > 0 ? 1.f / (float)n_utt : 0.f, dlogits, loss);
        SS_LAUNCH_CHECK("ss_ctc_loss(grad)");
    }
    return 0;
}
