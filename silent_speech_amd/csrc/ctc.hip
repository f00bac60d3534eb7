// ctc.hip -- CTC loss of the recognition trainer on gfx950 (reference recognition_model.py:96-101):
//     pred = log_softmax(model(...), 2); pred = pad_sequence(decollate_tensor(pred, lengths))
//     loss = F.ctc_loss(pred, y, lengths, text_int_lengths, blank=n_chars)            (reduction 'mean')
// The kernels work on the PACKED frame layout the encoder produces (utterance u = frames [frame0, frame0+T) of the
// flattened (rows*200, V) logits, data_utils.py:159-179), so neither the decollate/pad copy nor the (T_max, N, V)
// log-prob tensor exists.  log_softmax is folded in through the per-frame log-sum-exp (ss_frame_lse).
//   alpha/beta : a wave-level log-domain scan, one wave per (utterance, direction), a lane owns consecutive states (see the kernel).  beta is the same
//                recursion on the reversed label string and reversed time.
//   gradient   : one wave per packed frame: occupancy per class via LDS bins, then
//                d loss / d logit[t][c] = (softmax[t][c] - exp(log occ[t][c] + nll - logp[t][c])) / (max(S,1) * N)
//                which is ATen's ctc_loss backward composed with log_softmax's (frames outside any utterance get 0).
#include "common.h"
#include "silent_speech_hip.h"
#include <math.h>

namespace {
constexpr int CD = 5;                      // descriptor row: frame0, T, target0, S, workspace offset (floats)
constexpr int CTC_MAX_STATES = 1024, CTC_FC = 32;      // 16 states per lane; 32 frames of class probabilities staged per wave

}

// alpha / beta as a WAVE-LEVEL scan (round 6).  One wave per (utterance, direction) -- both directions of an utterance in one workgroup of two waves --,
// lane l owns the NS consecutive extended states l NS .. l NS + NS - 1 (NS even: 2, 4, 6, 8 or 16 >= ceil((2S+1) / 64)), so that of the two predecessors
// of a state only those of a lane's first two states live in another lane: two wave shifts per frame, no workgroup barrier, no LDS column.
// The recursion stays in the LOG domain (a probability-domain scan with one normalisation per frame was built first: 3 VALU operations per state, but in f32
// the paths that are "ahead of schedule" -- the only ones that still reach the last label of a long utterance -- underflow against the bulk of the mass;
// 0.04 off in the nll of the 267-frame golden case), in base 2 on the transcendental unit, and a lane's states alternate blank / label at COMPILE time (NS even):
//     blank state (two predecessors):   m + log2(1 + 2^(lo - m))                      1 v_exp + 1 v_log
//     label state (three):              m + log2(1 + 2^(md - m) + 2^(lo - m))         2 v_exp + 1 v_log     (m / md / lo = v_max3 / v_med3 / v_min3)
// instead of 3 expf + 1 logf (library forms) per state.  Stored: ln alpha_t(s), as before.
// (Rounds 3-5: one 256-thread workgroup per (utterance, direction), the previous column in LDS, a workgroup barrier per frame: 0.9 us per frame, 0.785 ms for a
// 128 000-sample batch whose longest utterance has 860 frames.)
namespace {
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;
__device__ __forceinline__ float fast_log2(float x) {
#if defined(SS_EMU)
    return log2f(x);
#else
    return __builtin_amdgcn_logf(x);
#endif
}
// log2(2^a + 2^b [+ 2^c]) on FINITE numbers: "minus infinity" is the sentinel CTC_NEG inside the scan (-1e30 absorbs every log-probability added to it,
// 2^(x - CTC_NEG) never occurs with x > CTC_NEG as the larger argument is subtracted), so the per-frame chain carries no NaN guard and no branch --
// the first version tested m == -inf per state and hipcc made each test a branch around the transcendentals: ~40 taken / not-taken branches per frame.
constexpr float CTC_NEG = -1e30f;
__device__ __forceinline__ float lse2_b2(float a, float b) {
    const float m = fmaxf(a, b), lo = fminf(a, b);
    return m + fast_log2(1.f + fast_exp2(lo - m));
}
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c)), lo = fminf(a, fminf(b, c));
    const float md = fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));             // the median
    return m + fast_log2(1.f + fast_exp2(md - m) + fast_exp2(lo - m));
}
}
// The states of one (utterance, direction) are dealt to W WAVES, 128 consecutive states each (lane l: states 2 l (blank), 2 l + 1 (label) of the wave's range).
// State s only depends on s, s - 1, s - 2 of the previous frame, so wave w needs exactly two numbers per frame from wave w - 1 -- and never the reverse: the
// waves form a PIPELINE through an LDS ring (CTC_R frames x 2 floats per wave pair) with a progress word per wave, no workgroup barrier; wave w simply runs
// a frame or more behind wave w - 1.  (One wave per direction was built first: its frame costs ~150 instructions for the 6 states of a lane at SP = 283,
// 0.47 us -- instruction issue of a single wave, with 250 CUs idle.  Two states per lane: ~45 instructions per frame.)
// Progress protocol: a wave with a consumer publishes prog = t after writing frame t's pair (same lane, LDS operations of a wave are served in order); a
// consumer polls the producer's word only when it has caught up with what it last saw.  The ring is protected the other way round: every wave publishes its
// progress at least every 8 frames and a producer never runs more than CTC_R - 16 frames ahead of what its consumer last published.
namespace {
constexpr int CTC_NSW = 2, CTC_C = 64 * CTC_NSW, CTC_R = 64, CTC_LAG = 8;
__device__ __forceinline__ void ctc_spin() {
#if defined(SS_EMU)
    hipemu::yield_to_sched();
#else
    __builtin_amdgcn_s_sleep(1);
#endif
}
}
__global__ __launch_bounds__(1024) void ctc_alpha_beta_kernel(const float* __restrict__ logits, long long ld, int V, int blank, const float* __restrict__ lse,
                                                              const long long* __restrict__ desc, const int* __restrict__ targets,
                                                              float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll, int W)
{
    constexpr int NS = CTC_NSW;
    SS_DYN_SMEM(smem);
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = wave_uniform(tid >> 6);
    const bool rev = wv >= W;
    const int w = rev ? wv - W : wv;
    const long long f0 = desc[u * CD + 0], T = desc[u * CD + 1], g0 = desc[u * CD + 2], S = desc[u * CD + 3], w0 = desc[u * CD + 4];
    const int SP = (int)(2 * S + 1);
    float* pc = (float*)smem + (size_t)wv * (CTC_FC * V);                // this wave's [CTC_FC][V] staged log2-probabilities
    volatile float* rings = (volatile float*)((float*)smem + (size_t)2 * W * (CTC_FC * V));
    volatile int* prog = (volatile int*)(rings + (size_t)2 * W * CTC_R * 2);
    if (lane == 0) prog[wv] = -1;
    __syncthreads();
    float* out = (rev ? beta : alpha) + w0;
    if (T <= 0) { if (wv == 0 && lane == 0) nll[u] = S == 0 ? 0.f : INFINITY; return; }
    const int s0 = w * CTC_C;
    if (s0 >= SP) return;                                                // this wave owns no state of this utterance
    const bool has_prod = w > 0, has_cons = w + 1 < W && (w + 1) * CTC_C < SP;
    volatile float* ring_in = rings + (size_t)(wv - 1) * (CTC_R * 2);   // written by wave w - 1 of this direction (has_prod only)
    volatile float* ring_out = rings + (size_t)wv * (CTC_R * 2);

    int lab[NS]; bool skip[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int s = s0 + lane * NS + k;                                // logical state (reversed string when rev); odd k = label states
        lab[k] = blank; skip[k] = false;
        if (s < SP && (k & 1)) {
            const int j = s >> 1;
            const int c = targets[g0 + (rev ? S - 1 - j : j)];
            lab[k] = c;
            if (j >= 1) skip[k] = targets[g0 + (rev ? S - j : j - 1)] != c;
        }
    }
    float a[NS];
    int oidx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) { const int s_ = s0 + lane * NS + k; a[k] = CTC_NEG; oidx[k] = rev ? SP - 1 - s_ : s_; }
    float* row = out + (rev ? (T - 1) * SP : 0);
    const long long rstep = rev ? -(long long)SP : (long long)SP;
    int seen_prod = -1, seen_cons = -1;                                  // what this wave last read of its neighbours' progress words
    float hn0 = CTC_NEG, hn1 = CTC_NEG;                                  // the producer's pair for the next frame
    // chunks of CTC_FC frames: stage their log2-probabilities, settle every load, then run the frames of the chunk with NO load in flight -- with the
    // staging inside one flat frame loop hipcc carried "loads pending" around the back edge and waited vmcnt(0) at the top of EVERY frame, which on gfx9
    // also waits for the previous frame's alpha stores (a store round trip per frame on the serial chain)
    for (long long t0 = 0; t0 < T; t0 += CTC_FC) {
        const long long left = T - t0; const int nf = left < CTC_FC ? (int)left : CTC_FC;
        wave_lds_sync();
        for (int i = lane; i < nf * V; i += 64) {
            const int f = i / V, v = i - f * V;
            const long long fr = f0 + (rev ? T - 1 - (t0 + f) : t0 + f);
            pc[f * V + v] = (logits[fr * ld + v] - lse[fr]) * LOG2E_F;
        }
#if !defined(SS_EMU)
        __builtin_amdgcn_s_waitcnt(0);
#endif
        wave_lds_sync();
        // Software pipeline over the frames: the log2-probabilities of frame t + 1 and the producer's pair for frame t + 1 (= its frame t) are requested
        // at the top of frame t, so neither LDS round trip sits on the chain; the producer's progress word is polled only when this wave has used up what
        // it last saw, and then it waits for CTC_LAG frames at once (a wave that polled every frame ran no faster than one wave doing all the states)
        float lpn[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) lpn[k] = pc[lab[k]];
        for (int fi = 0; fi < nf; ++fi) {
            const int t = (int)t0 + fi;
            float lp[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) lp[k] = lpn[k];
            const float h0 = hn0, h1 = hn1;                             // the two states below this wave's range, frame t - 1
            if (fi + 1 < nf) {
#pragma unroll
                for (int k = 0; k < NS; ++k) lpn[k] = pc[(fi + 1) * V + lab[k]];
            }
            if (has_prod && t + 1 < (int)T) {                            // the pair of frame t, for frame t + 1
                if (seen_prod < t) {
                    const int want = t + CTC_LAG < (int)T - 1 ? t + CTC_LAG : (int)T - 1;
                    while (seen_prod < want) { seen_prod = prog[wv - 1]; if (seen_prod < want) ctc_spin(); }
                }
                hn0 = ring_in[(t & (CTC_R - 1)) * 2]; hn1 = ring_in[(t & (CTC_R - 1)) * 2 + 1];
            }
            float nv[NS];
            if (t == 0) {
#pragma unroll
                for (int k = 0; k < NS; ++k) { const int s_ = s0 + lane * NS + k; nv[k] = s_ < 2 && s_ < SP ? lp[k] : CTC_NEG; }
            } else {
                // predecessors of a lane's first two states: the last two states of the lane below (lane 0: of the wave below)
                const float p1 = wave_shr1(a[NS - 1], h1), p2 = wave_shr1(a[NS - 2], h0);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const float a1 = k >= 1 ? a[k >= 1 ? k - 1 : 0] : p1, a2 = k >= 2 ? a[k >= 2 ? k - 2 : 0] : (k == 1 ? p1 : p2);
                    const float v = (k & 1) ? lse3_b2(a[k], a1, skip[k] ? a2 : CTC_NEG) : lse2_b2(a[k], a1);
                    nv[k] = fmaxf(v + lp[k], CTC_NEG);                  // (lp = -inf, a class of probability 0, lands on the sentinel; phantom states beyond SP only ever feed higher phantom states)
                }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                a[k] = nv[k];
                if (s0 + lane * NS + k < SP) row[oidx[k]] = a[k] > 0.5f * CTC_NEG ? a[k] * LN2_F : -INFINITY;
            }
            row += rstep;
            if (has_cons) {
                // never more than CTC_R - 16 frames ahead of what the consumer last published (it publishes at least every 8 frames)
                while (t - seen_cons > CTC_R - 16) { seen_cons = prog[wv + 1]; if (t - seen_cons > CTC_R - 16) ctc_spin(); }
                if (lane == 63) { ring_out[(t & (CTC_R - 1)) * 2] = a[0]; ring_out[(t & (CTC_R - 1)) * 2 + 1] = a[1]; prog[wv] = t; }
            } else if ((t & 7) == 7 && lane == 63) prog[wv] = t;
        }
    }
    if (lane == 63) prog[wv] = (int)T - 1 + CTC_R;                       // done: a producer still ahead of this wave never waits for it again
    if (!rev && s0 <= SP - 1 && SP - 1 < s0 + CTC_C) {
        // nll = -ln(alpha_T(SP-1) + alpha_T(SP-2)); SP - 2 may be the last state of the wave below (its last ring entry)
        float e1 = CTC_NEG, e2 = CTC_NEG;
#pragma unroll
        for (int k = 0; k < NS; ++k) { const int s_ = s0 + lane * NS + k; if (s_ == SP - 1) e1 = a[k]; if (s_ == SP - 2) e2 = a[k]; }
        e1 = wave_max(e1); e2 = wave_max(e2);
        if (SP - 2 >= 0 && SP - 2 < s0) {
            while (seen_prod < (int)T - 1) { seen_prod = prog[wv - 1]; if (seen_prod < (int)T - 1) ctc_spin(); }
            e2 = ring_in[(((int)T - 1) & (CTC_R - 1)) * 2 + 1];
        }
        const float l2 = lse2_b2(e1, e2);
        if (lane == 0) nll[u] = l2 > 0.5f * CTC_NEG ? -l2 * LN2_F : INFINITY;
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
    SS_CHECK(sp_cap <= CTC_MAX_STATES, "ss_ctc_loss: target length %d exceeds the %d-label limit", max_target_len, (CTC_MAX_STATES - 1) / 2);
    if (n_utt > 0) {
        SS_CHECK(desc && alpha_ws && beta_ws && nll, "ss_ctc_loss: null workspace");
        SS_CHECK(targets || max_target_len == 0, "ss_ctc_loss: null targets");
        const int W = (sp_cap + CTC_C - 1) / CTC_C;                      // waves per direction: 128 states each
        const size_t smem = sizeof(float) * ((size_t)2 * W * CTC_FC * V + (size_t)2 * W * CTC_R * 2 + 2 * W);
        SS_CHECK(smem <= 160 * 1024, "ss_ctc_loss: %zu bytes of LDS needed", smem);
        static size_t granted = 0;
        if (granted < smem) { if (!ss_grant_lds((const void*)ctc_alpha_beta_kernel, smem)) { ss_set_error("ss_ctc_loss: cannot reserve %zu bytes of LDS", smem); return 1; } granted = smem; }
        SS_LAUNCH(ctc_alpha_beta_kernel, dim3(n_utt), dim3(2 * W * 64), smem, stream, logits, (long long)ld, V, blank, lse, (const long long*)desc, targets, alpha_ws, beta_ws, nll, W);
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
