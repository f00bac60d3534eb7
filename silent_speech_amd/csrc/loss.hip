// loss.hip -- dtw_loss of the transduction trainer on gfx950 (reference transduction_model.py:98-157).
//   voiced utterance : sum_t ||y_t - pred_t + 1e-6||_2 + lambda * CE_sum(aux, y_phone)          (:141-145)
//   silent utterance : costs[q][k] = ||pred_q - y_k||_2 - lambda * log_softmax(aux_q)[phone_k]   (:116-124)
//                      alignment = DTW(costs.T) (dtw.hip), loss = sum_k costs[alignment[k]][k]    (:126-128)
//   batch            : sum(losses) / sum(T2);  phoneme accuracy on the side                       (:157)
// The T1 x T2 cost matrix exists only in the DTW kernel's skewed strip layout and is never read back: the
// loss and its gradient touch only the T2 aligned (q, k) pairs, which are recomputed here, so no dense
// T1 x T2 gradient is ever materialised.  Forward value and gradient are produced by the same kernels
// (d loss / d head is written directly; upstream gradient of the scalar loss is folded in via inv_total).
// `head` is the fused output of the two linear heads: row = packed frame, columns [0,n_mel) = mel
// prediction (w_out, architecture.py:55), [n_mel, n_mel+n_phone) = phoneme logits (w_aux, :59), f32.
#include "common.h"
#include <stdlib.h>
#include "silent_speech_hip.h"
#include <math.h>

namespace {
constexpr int DESC = 10;
enum { D_N = 0, D_M, D_PRED_ROW0, D_TGT_ROW0, D_UNUSED, D_SK_OFF, D_DIRS_OFF, D_BND_OFF, D_RES_OFF };
constexpr int DW = 4, DR = 4;
__device__ __forceinline__ long long dtw_strips_d(long long n) { long long rows = n - 1, cap = DW * 64 * DR; return rows <= 0 ? 0 : (rows + cap - 1) / cap; }
__device__ __forceinline__ long long dtw_tsteps_d(long long m) { return m <= 1 ? 0 : (m - 1) + 63; }
}

// ---------------------------------------------------------------- per-frame log-sum-exp + argmax of the phoneme logits
__global__ void frame_lse_kernel(const float* __restrict__ head, long long ld, int col0, int ncls, int rows, float* __restrict__ lse, int* __restrict__ amax)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb) {
        const float* a = head + (long long)r * ld + col0;
        float mx = -INFINITY; int mi = 0x7fffffff;
        for (int c = lane; c < ncls; c += 64) { float v = a[c]; if (v > mx) { mx = v; mi = c; } }
        const float gmx = wave_max(mx);
        int cand = (mx == gmx) ? mi : 0x7fffffff;               // first maximal index, like torch.argmax
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(cand, m); cand = o < cand ? o : cand; }
        float s = 0.f;
        for (int c = lane; c < ncls; c += 64) s += expf(a[c] - gmx);
        s = wave_sum(s);
        if (lane == 0) { lse[r] = gmx + logf(s); amax[r] = cand; }
    }
}

extern "C" int ss_frame_lse(const float* head, int64_t ld, int col0, int ncls, int rows, float* lse, int32_t* argmax, void* stream)
{
    SS_CHECK(head && lse && argmax, "ss_frame_lse: null pointer");
    if (rows <= 0) return 0;
    int blocks = (rows + 3) / 4; if (blocks > 4096) blocks = 4096;
    SS_LAUNCH(frame_lse_kernel, dim3(blocks), dim3(256), 0, stream, head, (long long)ld, col0, ncls, rows, lse, argmax);
    SS_LAUNCH_CHECK("ss_frame_lse");
    return 0;
}

// ---------------------------------------------------------------- per-frame index tables of a batch, expanded on the device
// The packed-row <-> utterance bookkeeping of decollate_tensor + zip (transduction_model.py:101-111) as data: one row of 8 int64 per
// utterance [n_pred, n_tgt, silent, pred_row0, tgt_row0, res_off, vo_off, si_off] (44 rows for a reference-size batch: ONE small upload)
// instead of five per-frame int32 arrays built on the host (22 000 entries each) and copied over per step.
constexpr int UTT = 8;
__global__ void loss_index_tables_kernel(const long long* __restrict__ utt, int* __restrict__ vo_pred, int* __restrict__ vo_tgt,
                                         int* __restrict__ si_tgt, int* __restrict__ si_base, int* __restrict__ si_res)
{
    const long long* u = utt + (long long)blockIdx.x * UTT;
    const int n1 = (int)u[0], n2 = (int)u[1];
    const int p0 = (int)u[3], t0 = (int)u[4], r0 = (int)u[5];
    if (u[2]) {
        const long long o = u[7];
        for (int i = threadIdx.x; i < n2; i += blockDim.x) { si_tgt[o + i] = t0 + i; si_base[o + i] = p0; si_res[o + i] = r0 + i; }
    } else {
        const long long o = u[6];
        for (int i = threadIdx.x; i < n1; i += blockDim.x) { vo_pred[o + i] = p0 + i; vo_tgt[o + i] = t0 + i; }
    }
}

extern "C" int ss_loss_index_tables(const int64_t* utt_dev, int n_utt, int32_t* vo_pred, int32_t* vo_tgt, int32_t* si_tgt, int32_t* si_base,
                                    int32_t* si_res, void* stream)
{
    SS_CHECK(n_utt >= 0, "ss_loss_index_tables: negative utterance count");
    if (n_utt == 0) return 0;
    SS_CHECK(utt_dev && vo_pred && vo_tgt && si_tgt && si_base && si_res, "ss_loss_index_tables: null pointer");
    SS_LAUNCH(loss_index_tables_kernel, dim3((unsigned)n_utt), dim3(256), 0, stream, (const long long*)utt_dev, (int*)vo_pred, (int*)vo_tgt, (int*)si_tgt,
              (int*)si_base, (int*)si_res);
    SS_LAUNCH_CHECK("ss_loss_index_tables");
    return 0;
}

// per-wave partial sums (held by lane 0) -> ONE atomic per workgroup: thousands of waves hitting the same two addresses
// serialise in the L2 atomic unit (that, not the arithmetic, was most of these kernels' time)
__device__ __forceinline__ void loss_block_commit(float lsum, int csum, float inv_total, float* loss_acc, int* correct_acc)
{
    __shared__ float ls[16];
    __shared__ int cs[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    if (lane == 0) { ls[w] = lsum; cs[w] = csum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f; int c = 0;
        for (int i = 0; i < wpb; ++i) { a += ls[i]; c += cs[i]; }
        if (a != 0.f) atomicAdd(loss_acc, a * inv_total);
        if (c) atomicAdd(correct_acc, c);
    }
}

// ---------------------------------------------------------------- voiced frames: value + gradient, one wave per frame
__global__ void voiced_loss_kernel(const float* __restrict__ head, long long ld, int n_mel, int n_ph, const float* __restrict__ lse, const int* __restrict__ amax,
                                   const float* __restrict__ Y, const long long* __restrict__ phones, const int* __restrict__ pred_row, const int* __restrict__ tgt_row,
                                   int nframes, float lam, float inv_total, float* __restrict__ dhead, float* __restrict__ loss_acc, int* __restrict__ correct_acc)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float lsum = 0.f; int csum = 0;
    for (int f = blockIdx.x * wpb + (threadIdx.x >> 6); f < nframes; f += gridDim.x * wpb) {
        const int pr = pred_row[f], tr = tgt_row[f];
        const float* p = head + (long long)pr * ld; const float* y = Y + (long long)tr * n_mel;
        float* dp = dhead + (long long)pr * ld;
        float ss = 0.f;
        for (int c = lane; c < n_mel; c += 64) { float d = y[c] - p[c] + 1e-6f; ss += d * d; }        // F.pairwise_distance eps (:141)
        const float dist = sqrtf(wave_sum(ss));
        const float gs = dist > 0.f ? inv_total / dist : 0.f;
        for (int c = lane; c < n_mel; c += 64) dp[c] = -(y[c] - p[c] + 1e-6f) * gs;
        const int ph = (int)phones[tr];
        const float L = lse[pr];
        for (int c = lane; c < n_ph; c += 64) dp[n_mel + c] = lam * inv_total * (expf(p[n_mel + c] - L) - (c == ph ? 1.f : 0.f));
        if (lane == 0) { lsum += dist + lam * (L - p[n_mel + ph]); csum += amax[pr] == ph; }
    }
    loss_block_commit(lsum, csum, inv_total, loss_acc, correct_acc);
}

extern "C" int ss_voiced_loss(const float* head, int64_t ld, int n_mel, int n_phone, const float* lse, const int32_t* argmax, const float* Y, const int64_t* phones,
                              const int32_t* pred_row, const int32_t* tgt_row, int nframes, float lam, float inv_total,
                              float* dhead, float* loss_accum, int32_t* correct_accum, void* stream)
{
    SS_CHECK(nframes >= 0, "ss_voiced_loss: negative frame count");
    if (nframes == 0) return 0;
    SS_CHECK(head && lse && argmax && Y && phones && pred_row && tgt_row && dhead && loss_accum && correct_accum, "ss_voiced_loss: null pointer");
    int blocks = (nframes + 3) / 4; if (blocks > 1024) blocks = 1024;
    SS_LAUNCH(voiced_loss_kernel, dim3(blocks), dim3(256), 0, stream, head, (long long)ld, n_mel, n_phone, lse, (const int*)argmax, Y, (const long long*)phones,
              (const int*)pred_row, (const int*)tgt_row, nframes, lam, inv_total, dhead, loss_accum, (int*)correct_accum);
    SS_LAUNCH_CHECK("ss_voiced_loss");
    return 0;
}

// ---------------------------------------------------------------- silent utterances: cost matrix straight into the DTW strip layout
// DTW runs on costs.T: row i = target frame k, column j = predicted frame q (transduction_model.py:126).
// One thread owns one (lane, row) slot of a wave-strip -- i.e. ONE target frame i -- for a chunk of CT (16) consecutive steps t:
// the target row y_i stays in registers (n_mel <= 128), only the predicted rows stream in, and every step's 256 results
// (64 lanes x 4 rows) leave as one contiguous 1 KiB store.  The 80-term sum runs in the reference order (bit-exact costs).
constexpr int YMAX = 128;
// value of lane (quad base + k) for every lane of the quad: quad_perm [k,k,k,k] as a DPP operand of the consuming instruction
__device__ __forceinline__ float quad_bcast(float v, int k) {
#if defined(SS_EMU)
    return __shfl(v, (int)((threadIdx.x & 63) & ~3) + k);
#else
    switch (k) {
    case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00, 0xf, 0xf, true));
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x55, 0xf, 0xf, true));
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xAA, 0xf, 0xf, true));
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xFF, 0xf, 0xf, true));
    }
#endif
}
// squared distance of the predicted row p (NQ 16-byte pieces per quad lane, shared through quad broadcasts) to this thread's target row yv,
// the 16 NQ terms in the reference order
template <int NQ>
__device__ __forceinline__ float quad_shared_sqdist(const float* __restrict__ p, int r, const f32x4 (&yv)[YMAX / 4]) {
    f32x4 own[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) own[q] = *(const f32x4*)(p + (r * NQ + q) * 4);
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 b = yv[k * NQ + q];
            const float a0 = quad_bcast(own[q][0], k), a1 = quad_bcast(own[q][1], k), a2 = quad_bcast(own[q][2], k), a3 = quad_bcast(own[q][3], k);
            const float d0 = a0 - b[0], d1 = a1 - b[1], d2 = a2 - b[2], d3 = a3 - b[3];
            ss += d0 * d0; ss += d1 * d1; ss += d2 * d2; ss += d3 * d3;
        }
    }
    return ss;
}

__global__ __launch_bounds__(256) void silent_cost_skewed_kernel(const float* __restrict__ head, long long ld, int n_mel, const float* __restrict__ lse, const float* __restrict__ Y,
                                                                 const long long* __restrict__ phones, const long long* __restrict__ desc, float lam, unsigned char* __restrict__ ws, int* __restrict__ results, int CT)
{
    const long long* d = desc + (long long)blockIdx.y * DESC;
    const int N = (int)d[D_N], M = (int)d[D_M];
    const long long p0 = d[D_PRED_ROW0], y0 = d[D_TGT_ROW0];
    int* res = results + d[D_RES_OFF];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) res[i] = 0;
    const long long ts = dtw_tsteps_d(M), nkw = dtw_strips_d(N) * DW;
    if (ts <= 0) return;
    const long long tchunks = (ts + CT - 1) / CT, nwork = nkw * tchunks;
    float* sk = (float*)(ws + d[D_SK_OFF]);
    const int l = threadIdx.x >> 2, r = threadIdx.x & 3;
    for (long long wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
        const long long kw = wk / tchunks, t0 = (wk - kw * tchunks) * CT;
        const long long i = 1 + (kw * 64 + l) * DR + r;
        const bool iv = i < N;
        f32x4 yv[YMAX / 4];
        int ph = 0;
        if (iv) {
            const float* y = Y + (y0 + i) * n_mel;
#pragma unroll
            for (int q = 0; q < YMAX / 4; ++q) if (q * 4 < n_mel) yv[q] = *(const f32x4*)(y + q * 4);
            ph = (int)phones[y0 + i];
        }
        const long long tend = t0 + CT < ts ? t0 + CT : ts;
        // The four threads of a quad (the 4 rows of one lane) need the SAME predicted row: each loads a quarter of it (5 instead of 20
        // 16-byte loads per cell; the kernel was bound by those) and the others read it through quad-broadcast DPP operands.  The 80-term
        // sum keeps the reference order.  (80 mel bins; any other width: every thread loads the row itself.)
        if (n_mel == 80) {                                               // the model's mel bins (NQ = 5)
            for (long long t = t0; t < tend; ++t) {
                const long long j = t + 1 - l;
                const bool jv = j >= 1 && j < M;                          // quad-uniform
                const float* p = head + (p0 + (jv ? j : 1)) * ld;           // columns outside the matrix: any valid row, result discarded --
                const float ss = quad_shared_sqdist<5>(p, r, yv);         // every lane takes part in the broadcasts (no divergence around them)
                float c = INFINITY;
                if (jv && iv) c = sqrtf(ss) + lam * (lse[p0 + j] - p[n_mel + ph]);
                sk[((kw * ts + t) * 64 + l) * DR + r] = c;
            }
            continue;
        }
        for (long long t = t0; t < tend; ++t) {
            const long long j = t + 1 - l;
            float c = INFINITY;
            if (iv && j >= 1 && j < M) {
                const float* p = head + (p0 + j) * ld;
                float ss = 0.f;
#pragma unroll
                for (int q = 0; q < YMAX / 4; ++q) {
                    if (q * 4 < n_mel) {
                        const f32x4 a = *(const f32x4*)(p + q * 4), b = yv[q];
                        const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2], d3 = a[3] - b[3];
                        ss += d0 * d0; ss += d1 * d1; ss += d2 * d2; ss += d3 * d3;
                    }
                }
                c = sqrtf(ss) + lam * (lse[p0 + j] - p[n_mel + ph]);
            }
            sk[((kw * ts + t) * 64 + l) * DR + r] = c;
        }
    }
}

extern "C" int ss_silent_cost_skewed(const float* head, int64_t ld, int n_mel, const float* lse, const float* Y, const int64_t* phones,
                                     const int64_t* desc_dev, int n, int max_n, int max_m, float lam, void* workspace, int32_t* results, void* stream)
{
    SS_CHECK(n >= 0, "ss_silent_cost_skewed: negative batch");
    if (n == 0) return 0;
    SS_CHECK(head && lse && Y && phones && desc_dev && workspace && results, "ss_silent_cost_skewed: null pointer");
    SS_CHECK(n_mel % 4 == 0 && ld % 4 == 0, "ss_silent_cost_skewed: n_mel and ld must be multiples of 4 (16-byte rows)");
    SS_CHECK(n_mel <= YMAX, "ss_silent_cost_skewed: at most %d mel bins", YMAX);
    long long strips = max_n <= 1 ? 0 : (max_n - 1 + DW * 64 * DR - 1) / (DW * 64 * DR);
    const long long ts_max = max_m <= 1 ? 0 : max_m - 1 + 63;
    // steps per work item: every step is a dependent load -> 80-term sum round trip, so the kernel lives on the number of work items in
    // flight (32 steps: 135 us for the bench batch, 16: 84, 8: 80; below that the re-loaded target row starts to show)
    static const int CT = getenv("SS_COST_CT") ? atoi(getenv("SS_COST_CT")) : 16;
    long long blocks = strips * DW * ((ts_max + CT - 1) / CT);            // one workgroup per (wave-strip, chunk of CT steps)
    if (blocks < (max_n + 255) / 256) blocks = (max_n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    SS_LAUNCH(silent_cost_skewed_kernel, dim3((unsigned)blocks, n), dim3(256), 0, stream, head, (long long)ld, n_mel, lse, Y, (const long long*)phones,
              (const long long*)desc_dev, lam, (unsigned char*)workspace, (int*)results, CT);
    SS_LAUNCH_CHECK("ss_silent_cost_skewed");
    return 0;
}

// ---------------------------------------------------------------- silent utterances: value + gradient over the aligned pairs only
__global__ void silent_loss_kernel(const float* __restrict__ head, long long ld, int n_mel, int n_ph, const float* __restrict__ lse, const int* __restrict__ amax,
                                   const float* __restrict__ Y, const long long* __restrict__ phones, const int* __restrict__ results,
                                   const int* __restrict__ tgt_row, const int* __restrict__ pred_base, const int* __restrict__ res_idx,
                                   int nframes, float lam, float inv_total, float* __restrict__ dhead, float* __restrict__ loss_acc, int* __restrict__ correct_acc)
{
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    float lsum = 0.f; int csum = 0;
    for (int f = blockIdx.x * wpb + (threadIdx.x >> 6); f < nframes; f += gridDim.x * wpb) {
        const int tr = tgt_row[f], pr = pred_base[f] + results[res_idx[f]];
        const float* p = head + (long long)pr * ld; const float* y = Y + (long long)tr * n_mel;
        float* dp = dhead + (long long)pr * ld;
        float ss = 0.f;
        for (int c = lane; c < n_mel; c += 64) { float d = p[c] - y[c]; ss += d * d; }
        const float dist = sqrtf(wave_sum(ss));
        const float gs = dist > 0.f ? inv_total / dist : 0.f;                          // cdist backward: 0 where the distance is 0
        for (int c = lane; c < n_mel; c += 64) atomicAdd(dp + c, (p[c] - y[c]) * gs);     // several k may align to one q
        const int ph = (int)phones[tr];
        const float L = lse[pr];
        for (int c = lane; c < n_ph; c += 64) atomicAdd(dp + n_mel + c, lam * inv_total * (expf(p[n_mel + c] - L) - (c == ph ? 1.f : 0.f)));
        if (lane == 0) { lsum += dist + lam * (L - p[n_mel + ph]); csum += amax[pr] == ph; }
    }
    loss_block_commit(lsum, csum, inv_total, loss_acc, correct_acc);
}

extern "C" int ss_silent_loss(const float* head, int64_t ld, int n_mel, int n_phone, const float* lse, const int32_t* argmax, const float* Y, const int64_t* phones,
                              const int32_t* results, const int32_t* tgt_row, const int32_t* pred_base, const int32_t* res_idx, int nframes,
                              float lam, float inv_total, float* dhead, float* loss_accum, int32_t* correct_accum, void* stream)
{
    SS_CHECK(nframes >= 0, "ss_silent_loss: negative frame count");
    if (nframes == 0) return 0;
    SS_CHECK(head && lse && argmax && Y && phones && results && tgt_row && pred_base && res_idx && dhead && loss_accum && correct_accum, "ss_silent_loss: null pointer");
    int blocks = (nframes + 3) / 4; if (blocks > 1024) blocks = 1024;
    SS_LAUNCH(silent_loss_kernel, dim3(blocks), dim3(256), 0, stream, head, (long long)ld, n_mel, n_phone, lse, (const int*)argmax, Y, (const long long*)phones,
              (const int*)results, (const int*)tgt_row, (const int*)pred_base, (const int*)res_idx, nframes, lam, inv_total, dhead, loss_accum, (int*)correct_accum);
    SS_LAUNCH_CHECK("ss_silent_loss");
    return 0;
}

// ---------------------------------------------------------------- phoneme confusion matrix (evaluation), accumulated on the device
// transduction_model.py:130-137 (silent: predictions gathered through the DTW alignment) and :147-152 (voiced): confusion[pred][target] += 1
// per target frame.  One thread per frame of the two index tables, one integer atomic each; the matrix stays on the device across batches,
// so test() reads it back ONCE per epoch (the reference calls .item() / .cpu() per utterance).
__global__ void phoneme_confusion_kernel(const int* __restrict__ amax, const long long* __restrict__ phones, const int* __restrict__ results,
                                         const int* __restrict__ vo_pred, const int* __restrict__ vo_tgt, int n_voiced,
                                         const int* __restrict__ si_tgt, const int* __restrict__ si_base, const int* __restrict__ si_res, int n_silent,
                                         int* __restrict__ conf, int n_phone)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_voiced + n_silent; i += gridDim.x * blockDim.x) {
        int p, t;
        if (i < n_voiced) { p = amax[vo_pred[i]]; t = (int)phones[vo_tgt[i]]; }
        else { const int k = i - n_voiced; p = amax[si_base[k] + results[si_res[k]]]; t = (int)phones[si_tgt[k]]; }
        if ((unsigned)p < (unsigned)n_phone && (unsigned)t < (unsigned)n_phone) atomicAdd(conf + p * n_phone + t, 1);
        else atomicAdd(conf + n_phone * n_phone, 1);     // a label outside the inventory: the reference raises IndexError (or wraps a negative one); counted, the host checks
    }
}

extern "C" int ss_phoneme_confusion(const int32_t* argmax, const int64_t* phones, const int32_t* results, const int32_t* vo_pred, const int32_t* vo_tgt,
                                    int n_voiced, const int32_t* si_tgt, const int32_t* si_base, const int32_t* si_res, int n_silent_frames,
                                    int32_t* confusion, int n_phone, void* stream)
{
    SS_CHECK(argmax && phones && confusion, "ss_phoneme_confusion: null pointer");
    SS_CHECK(n_voiced == 0 || (vo_pred && vo_tgt), "ss_phoneme_confusion: voiced tables missing");
    SS_CHECK(n_silent_frames == 0 || (results && si_tgt && si_base && si_res), "ss_phoneme_confusion: silent tables missing");
    const int n = n_voiced + n_silent_frames;
    if (n <= 0) return 0;
    int blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    SS_LAUNCH(phoneme_confusion_kernel, dim3(blocks), dim3(256), 0, stream, argmax, (const long long*)phones, results, vo_pred, vo_tgt, n_voiced,
              si_tgt, si_base, si_res, n_silent_frames, confusion, n_phone);
    SS_LAUNCH_CHECK("ss_phoneme_confusion");
    return 0;
}
