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
constexpr int CTC_MAX_STATES = 1024, CTC_FC = 32;      // 8 compute waves of 128 states; frames per staged chunk

}

// alpha / beta as a WAVE-LEVEL scan (round 6).  One workgroup per (utterance, direction); its W waves each own 128 consecutive extended states, lane l the
// states 2 l (a blank) and 2 l + 1 (a label) of the wave's range: of the predecessors s, s - 1, s - 2 of a state only the label state of the lane below lives in
// another lane -- one DPP wave shift per frame, no workgroup barrier, no LDS column.  beta is the same recursion on the reversed label string and reversed time.
// The recursion stays in the LOG domain (a probability-domain scan with one normalisation per frame was built first: 3 VALU operations per state, but in f32
// the paths that are "ahead of schedule" -- the only ones that still reach the last label of a long utterance -- underflow against the bulk of the mass;
// 0.04 off in the nll of the 267-frame golden case), in base 2 on the transcendental unit:
//     blank state (two predecessors):   m + log2(1 + 2^(lo - m))                      1 v_exp + 1 v_log
//     label state (three):              m + log2(1 + 2^(md - m) + 2^(lo - m))         2 v_exp + 1 v_log     (m / md / lo = v_max3 / v_med3 / v_min3)
// instead of 3 expf + 1 logf (library forms) per state.  Stored: log2 alpha_t(s), "minus infinity" as the finite sentinel CTC_NEG (ctc_grad_kernel reads it so).
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
// (v_med3_f32 through its builtin: fminf / fmaxf make hipcc canonicalise each operand first -- v_max_f32 x, x, x, seven extra instructions per frame)
__device__ __forceinline__ float med3(float a, float b, float c) {
#if defined(SS_EMU)
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
#else
    return __builtin_amdgcn_fmed3f(a, b, c);
#endif
}
__device__ __forceinline__ float lse2_b2(float a, float b) {
    const float m = med3(a, b, INFINITY), lo = med3(a, b, -INFINITY);      // max, min
    return m + fast_log2(1.f + fast_exp2(lo - m));
}
__device__ __forceinline__ float lse3_b2(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c)), lo = fminf(a, fminf(b, c)), md = med3(a, b, c);
    return m + fast_log2(1.f + fast_exp2(md - m) + fast_exp2(lo - m));
}
}
// The waves of a workgroup form a PIPELINE: state s only depends on s, s - 1, s - 2 of the previous frame, so wave w needs exactly one number per frame from
// wave w - 1 (the label state below its range) -- and never the reverse.  It travels through an LDS ring (CTC_R frames per wave pair) with a progress word per
// wave, no workgroup barrier; wave w simply runs a few frames behind wave w - 1.  (One wave per direction was built first: its frame costs ~150 instructions
// for the 6 states of a lane at SP = 283, 0.47 us -- instruction issue of a single wave, with 250 CUs idle.)
// Progress protocol, per block of CTC_B = 8 frames: a wave publishes prog = (last frame of the block) after writing the block's ring entries (same lane; LDS
// serves a wave's operations in order); a consumer polls the producer's word before a block, until the block's last frame is there.  The ring is protected
// the other way round: before a block a producer checks that it is at most 48 frames ahead of what its consumer last published (so never more than 55 < CTC_R).
// What the measurements of the second version said (tools/bin/ctcdbg variants, 850-frame batch, 0.425 ms): the polled words were `volatile` FLAT accesses with
// a vmcnt(0) behind each (the alpha stores' round trip on the chain, see common.h lds_peek / lds_post: -0.14 ms), the chunk staging loaded two values at a
// time behind 20-instruction divisions (0.06 ms), both directions of an utterance shared the four SIMDs of one CU, the loop's lgkmcnt(0) waited for the ring
// WRITE of the frame, and each alpha store cost five VALU operations (scale to ln, -inf select, 64-bit address).
namespace {
constexpr int CTC_C = 128, CTC_R = 64, CTC_B = 8, CTC_NB = 4;
static_assert(CTC_FC % CTC_B == 0 && CTC_R % CTC_FC == 0, "a block of frames occupies consecutive ring slots");
__device__ __forceinline__ void ctc_spin() {
#if defined(SS_EMU)
    hipemu::yield_to_sched();
#else
    __builtin_amdgcn_s_sleep(1);
#endif
}
}
// row[byte offset] = v with the row in scalar registers and a 32-bit lane offset (hipcc forms a 64-bit per-lane address on the VALU for the C expression)
__device__ __forceinline__ void ctc_store(float* row, unsigned byte_off, float v) {
#if defined(SS_EMU)
    *(float*)((char*)row + byte_off) = v;
#else
    asm volatile("global_store_dword %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(row) : "memory");
#endif
}
// Waves 0 .. W - 1 compute; wave W is the LOADER: it stages the log2-probabilities of chunk after chunk (CTC_FC = 32 frames x V classes) into a ring of
// CTC_NB chunk buffers that all compute waves read (they need the same frames, a few frames apart), so that a compute wave's chunk boundary is one
// progress-word poll instead of 64 loads + 32 LDS writes + their address arithmetic (~1.4 us per chunk on the chain of every wave, 27 chunks).
__global__ __launch_bounds__(576) void ctc_alpha_beta_kernel(const float* __restrict__ logits, long long ld, int V, int blank, const float* __restrict__ lse,
                                                             const long long* __restrict__ desc, const int* __restrict__ targets,
                                                             float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll, int W)
{
    SS_DYN_SMEM(smem);
    const int u = blockIdx.x >> 1, tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    const bool rev = blockIdx.x & 1;
    const long long f0 = desc[u * CD + 0], T = desc[u * CD + 1], g0 = desc[u * CD + 2], S = desc[u * CD + 3], w0 = desc[u * CD + 4];
    const int SP = (int)(2 * S + 1);
    float* chunks = (float*)smem;                                        // [CTC_NB][CTC_FC][V] staged log2-probabilities
    float* rings = (float*)smem + (size_t)CTC_NB * (CTC_FC * V);        // polled words: lds_peek / lds_post only (see common.h)
    int* prog = (int*)(rings + (size_t)W * CTC_R);                       // [0, W): frames done per compute wave; [W]: chunks staged
    if (lane == 0) lds_post_i32(prog + w, -1);
    __syncthreads();
    float* out = (rev ? beta : alpha) + w0;
    if (T <= 0) { if (!rev && w == 0 && lane == 0) nll[u] = S == 0 ? 0.f : INFINITY; return; }
    if (w == W) {
        // ---- the loader.  Chunk c goes to buffer c % CTC_NB once the LAST active compute wave is done with chunk c - CTC_NB; 8 values per lane are in
        // flight at a time (the divisions are off everybody's chain here).
        const int wl = (SP - 1) / CTC_C;                                  // last compute wave that owns a state
        const float* lg0 = logits + f0 * ld; const float* ls0 = lse + f0;
        int seen = -1;
        for (int c = 0, t0 = 0; t0 < (int)T; ++c, t0 += CTC_FC) {
            const int left = (int)T - t0, n = (left < CTC_FC ? left : CTC_FC) * V;
            const int need = (c - CTC_NB + 1) * CTC_FC - 1;              // last frame of chunk c - CTC_NB
            while (seen < need) { seen = wave_uniform(lds_peek_i32(prog + wl)); if (seen < need) ctc_spin(); }
            compiler_fence();
            float* pc = chunks + (size_t)(c % CTC_NB) * (CTC_FC * V);
            for (int base = 0; base < n; base += 64 * 8) {
                float x[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = base + e * 64 + lane, ic = i < n ? i : n - 1;
                    const int f = ic / V, v = ic - f * V, fr = rev ? (int)T - 1 - (t0 + f) : t0 + f;
                    x[e] = lg0[fr * (int)ld + v]; l[e] = ls0[fr];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { const int i = base + e * 64 + lane; if (i < n) pc[i] = (x[e] - l[e]) * LOG2E_F; }
            }
            compiler_fence();
            if (lane == 0) lds_post_i32(prog + W, c);
        }
        return;
    }
    const int s0 = w * CTC_C;
    if (s0 >= SP) return;                                                // this wave owns no state of this utterance
    const bool has_prod = w > 0, has_cons = w + 1 < W && (w + 1) * CTC_C < SP;
    const float* ring_in = rings + (size_t)(w > 0 ? w - 1 : 0) * CTC_R; // written by wave w - 1 (has_prod only)
    float* ring_out = rings + (size_t)w * CTC_R;
    float* dump = (float*)(prog + W + 1) + (size_t)w * (64 + CTC_B) + lane; // lanes 0 .. 62: words lane .. lane + 7 of this wave's dump row

    // lane l: states sb = s0 + 2 l (blank) and sb + 1 (label j = sb / 2 of the -- for beta reversed -- string)
    const int sb = s0 + 2 * lane;
    int lab1 = blank; bool skip1 = false;
    if (sb + 1 < SP) {
        const int j = sb >> 1;
        lab1 = targets[g0 + (rev ? S - 1 - j : j)];
        if (j >= 1) skip1 = targets[g0 + (rev ? S - j : j - 1)] != lab1;
    }
    const bool val0 = sb < SP, val1 = sb + 1 < SP;
    // the state before frame 0: probability 1 in a virtual state 0, nothing anywhere else -- the recursion itself then yields alpha_0(0) = lp(blank),
    // alpha_0(1) = lp(first label) (2^(CTC_NEG - 0) = 0 exactly, log2(1) = 0 exactly), so that frame 0 is no special case inside the loop
    float a0 = sb == 0 ? 0.f : CTC_NEG, a1 = CTC_NEG;
    const unsigned ob0 = 4u * (unsigned)(rev ? SP - 1 - sb : sb), ob1 = 4u * (unsigned)(rev ? SP - 2 - sb : sb + 1);      // 32-bit byte offsets: the stores take the row from scalar registers
    float* row = out + (rev ? (T - 1) * SP : 0);
    const long long rstep = rev ? -(long long)SP : (long long)SP;
    int seen_prod = -1, seen_cons = -1, seen_staged = -1;                                  // what this wave last read of its neighbours' progress words
    float hn1 = CTC_NEG;                                                 // the producer's last label state, for the next frame

    // chunks of CTC_FC frames, staged by the loader wave
    for (int c = 0, t0 = 0; t0 < (int)T; ++c, t0 += CTC_FC) {
        const int left = (int)T - t0, nf = left < CTC_FC ? left : CTC_FC;
        while (seen_staged < c) { seen_staged = wave_uniform(lds_peek_i32(prog + W)); if (seen_staged < c) ctc_spin(); }
        compiler_fence();
        const float* pc = chunks + (size_t)(c % CTC_NB) * (CTC_FC * V);
        // Software pipeline over the frames: the log2-probabilities of frame t + 1 and the producer's state for frame t + 1 (= its frame t) are requested
        // at the top of frame t and SETTLED before the frame's own ring write is issued, so that no LDS round trip sits on the chain.  The frames run in
        // blocks of CTC_B = 8 (a chunk starts at a multiple of 32, so a block's ring slots are consecutive: immediate offsets) and the pipeline is
        // synchronised per BLOCK: a consumer waits until its producer has published the block's last frame, a producer checks the ring once and publishes
        // once per block -- per frame there is one ring read and one ring write (every lane writes: lane 63 into the ring, the others into a dump row, no
        // exec-mask change).  Per-frame checks had cost ~45 scalar instructions around ~50 of arithmetic.
        float lpn0 = pc[blank], lpn1 = pc[lab1];
        const float* pcn0 = pc + blank; const float* pcn1 = pc + lab1;  // frame fi + 1's entries
        for (int fb = 0; fb < nf; fb += CTC_B) {
            const int tb = t0 + fb, nb = nf - fb < CTC_B ? nf - fb : CTC_B, tl = tb + nb - 1;       // frames tb .. tl
            if (has_prod) {
                // the ring entries read in this block: the producer's frames tb .. min(tl, T - 2)
                const int want = tl < (int)T - 1 ? tl : (int)T - 1;
                while (seen_prod < want) { seen_prod = wave_uniform(lds_peek_i32(prog + w - 1)); if (seen_prod < want) ctc_spin(); }
                compiler_fence();
            }
            if (has_cons) {                                               // never more than 48 + 7 frames ahead of what the consumer last published
                while (tb - seen_cons > CTC_R - 16) { seen_cons = wave_uniform(lds_peek_i32(prog + w + 1)); if (tb - seen_cons > CTC_R - 16) ctc_spin(); }
            }
            const float* rin = ring_in + (tb & (CTC_R - 1));
            float* rout = (lane == 63 ? ring_out + (tb & (CTC_R - 1)) : dump);
#pragma unroll
            for (int e = 0; e < CTC_B; ++e) {
                if (e < nb) {
                    const float lp0 = lpn0, lp1 = lpn1, h1 = hn1;
                    pcn0 += V; pcn1 += V;
                    if (fb + e + 1 < nf) { lpn0 = *pcn0; lpn1 = *pcn1; }
                    if (has_prod) hn1 = lds_peek_f32(rin + e);           // (the last frame's entry is read and never used)
                    const float p1 = wave_shr1(a1, h1);                  // the label state below this lane's blank (lane 0: of the wave below)
                    const float n0 = fmaxf(lse2_b2(a0, p1) + lp0, CTC_NEG);             // (lp = -inf, a class of probability 0, lands on the sentinel;
                    const float n1 = fmaxf(lse3_b2(a1, a0, skip1 ? p1 : CTC_NEG) + lp1, CTC_NEG);      //  phantom states beyond SP only ever feed higher phantom states)
                    a0 = n0; a1 = n1;
                    if (val0) ctc_store(row, ob0, a0);
                    if (val1) ctc_store(row, ob1, a1);
                    row += rstep;
                    pin_vgpr(lpn0); pin_vgpr(lpn1); pin_vgpr(hn1);       // the reads have landed HERE (else the wait sits behind the ring write below and waits for it too)
                    if (has_cons) lds_post_f32(rout + e, a1);
                }
            }
            compiler_fence();
            if (lane == 63) lds_post_i32(prog + w, tl);
        }
    }
    if (lane == 63) lds_post_i32(prog + w, (int)T - 1 + CTC_R);         // done: a producer still ahead of this wave never waits for it again
    if (!rev && s0 <= SP - 1 && SP - 1 < s0 + CTC_C) {
        // nll = -ln(alpha_T(SP-1) + alpha_T(SP-2)); SP - 2 may be the last state of the wave below (its last ring entry)
        float e1 = CTC_NEG, e2 = CTC_NEG;
        if (sb == SP - 1) e1 = a0;
        if (sb + 1 == SP - 1) e1 = a1;
        if (sb == SP - 2) e2 = a0;
        if (sb + 1 == SP - 2) e2 = a1;
        e1 = wave_max(e1); e2 = wave_max(e2);
        if (SP - 2 >= 0 && SP - 2 < s0) {
            while (seen_prod < (int)T - 1) { seen_prod = wave_uniform(lds_peek_i32(prog + w - 1)); if (seen_prod < (int)T - 1) ctc_spin(); }
            compiler_fence();
            e2 = lds_peek_f32(ring_in + (((int)T - 1) & (CTC_R - 1)));
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
        // alpha / beta are log2 values with the finite sentinel CTC_NEG for "no path" (a sum with a sentinel in it stays below CTC_NEG / 2)
        float m = CTC_NEG;
        for (int s = lane; s < SP; s += 64) m = fmaxf(m, a[s] + b[s]);
        m = wave_max(m);
        for (int c = lane; c < V; c += 64) bins[c] = 0.f;
        wave_lds_sync();
        for (int s = lane; s < SP; s += 64) {
            const float v = a[s] + b[s];
            if (v > 0.5f * CTC_NEG) atomicAdd(&bins[(s & 1) ? targets[g0 + (s >> 1)] : blank], fast_exp2(v - m));
        }
        wave_lds_sync();
        const float L = lse[r], nl = nll[lo], sc = inv_n / (float)(S > 1 ? S : 1), mln = m * LN2_F;
        for (int c = lane; c < ld; c += 64) {
            float g = 0.f;
            if (c < V) {
                const float lp = logits[r * ld + c] - L, occ = bins[c];
                g = expf(lp) - (occ > 0.f ? expf(logf(occ) + mln + nl - lp) : 0.f);
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
        const int W = (sp_cap + CTC_C - 1) / CTC_C;                      // compute waves per (utterance, direction): 128 states each; + 1 loader wave
        const size_t smem = sizeof(float) * ((size_t)CTC_NB * CTC_FC * V + (size_t)W * CTC_R + W + 1 + (size_t)W * (64 + CTC_B));
        SS_CHECK(smem <= 160 * 1024, "ss_ctc_loss: %zu bytes of LDS needed", smem);
        static size_t granted = 0;
        if (granted < smem) { if (!ss_grant_lds((const void*)ctc_alpha_beta_kernel, smem)) { ss_set_error("ss_ctc_loss: cannot reserve %zu bytes of LDS", smem); return 1; } granted = smem; }
        SS_LAUNCH(ctc_alpha_beta_kernel, dim3(2 * n_utt), dim3((W + 1) * 64), smem, stream, logits, (long long)ld, V, blank, lse, (const long long*)desc, targets, alpha_ws, beta_ws, nll, W);
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
