// attn_bench: times the fused relative-position attention (forward, backward) at one shape through the C ABI, no torch.
//   attn_bench [B H T dp D p iters]      default: the benchmark step's shape 110 8 200 96 100 0.2 20
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "silent_speech_hip.h"
#include <dlfcn.h>
// measurement hook of csrc/attention_t.hip: exists only in a -DATTN_T_MEASURE build of the library (tools/Makefile: stamplib), looked up at run time
static int ss_attn_t_debug_stamps(unsigned long long* out) {
    typedef int (*fn_t)(unsigned long long*);
    static fn_t fn = (fn_t)dlsym(RTLD_DEFAULT, "ss_attn_t_debug_stamps");
    return fn ? fn(out) : 0;
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float frand(uint32_t& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    int B = 110, H = 8, T = 200, dp = 96, D = 100, iters = 20; float pdrop = 0.2f;
    if (argc > 1) B = atoi(argv[1]); if (argc > 2) H = atoi(argv[2]); if (argc > 3) T = atoi(argv[3]); if (argc > 4) dp = atoi(argv[4]);
    if (argc > 5) D = atoi(argv[5]); if (argc > 6) pdrop = (float)atof(argv[6]); if (argc > 7) iters = atoi(argv[7]);
    const int MPt = (2 * D - 1 + 31) / 32 * 32, Tp = (T + 7) / 8 * 8;
    const size_t nqkv = (size_t)B * T * 3 * H * dp, nE = (size_t)H * (2 * D - 1) * dp, nET = (size_t)H * dp * MPt, nO = (size_t)B * T * H * dp;
    std::vector<uint16_t> h(nqkv); uint32_t s = 12345;
    for (auto& v : h) v = f2bf(frand(s) * 0.8f);
    std::vector<uint16_t> hE(nE), hET(nET, 0), hdO(nO);
    for (auto& v : hE) v = f2bf(frand(s) * 0.1f);
    for (int hh = 0; hh < H; ++hh) for (int m = 0; m < 2 * D - 1; ++m) for (int d = 0; d < dp; ++d) hET[((size_t)hh * dp + d) * MPt + m] = hE[((size_t)hh * (2 * D - 1) + m) * dp + d];
    for (auto& v : hdO) v = f2bf(frand(s));
    void *qkv, *E, *ET, *out, *dO, *dqkv; float *lse, *dsc;
    CK(hipMalloc(&qkv, nqkv * 2)); CK(hipMalloc(&E, nE * 2)); CK(hipMalloc(&ET, nET * 2)); CK(hipMalloc(&out, nO * 2)); CK(hipMalloc(&dO, nO * 2)); CK(hipMalloc(&dqkv, nqkv * 2));
    CK(hipMalloc(&lse, (size_t)B * H * T * 4)); CK(hipMalloc(&dsc, (size_t)B * H * T * 4));
    CK(hipMemcpy(qkv, h.data(), nqkv * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(E, hE.data(), nE * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ET, hET.data(), nET * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dO, hdO.data(), nO * 2, hipMemcpyHostToDevice));
    if (ss_relpos_attention_needs_transposed(SS_BF16, T, dp, D)) { fprintf(stderr, "shape runs the per-tile kernels (needs transposed copies): not benchmarked here\n"); return 2; }
    const int64_t nsaved = ss_relpos_attention_saved_bytes(SS_BF16, B, H, T, dp, D);          // 0 with SS_ATTN_SAVE_P=0: backward recomputes the probabilities
    void* pimg = nullptr; if (nsaved) CK(hipMalloc(&pimg, (size_t)nsaved));
    printf("saved probabilities: %.1f MB\n", nsaved / 1e6);
    hipStream_t st; CK(hipStreamCreate(&st));
    // family 2 (transposed 32 x 32 score tiles): the embedding table E / scale in fragment order, from the f32 embeddings [H][2D-1][dp]
    void* tab = nullptr;
    const int family = ss_relpos_attention_family(SS_BF16, T, dp, D);
    printf("kernel family %d (0 per-tile, 1 LDS-resident 16x16, 2 transposed 32x32)\n", family);
    if (family == 2) {
        std::vector<float> hf(nE); for (size_t i = 0; i < nE; ++i) { uint32_t u = (uint32_t)hE[i] << 16; memcpy(&hf[i], &u, 4); }
        float* embf; CK(hipMalloc(&embf, nE * 4)); CK(hipMemcpy(embf, hf.data(), nE * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&tab, (size_t)ss_relpos_attention_table_bytes(H, dp, D)));
        if (ss_relpos_attention_prepare_tables(embf, tab, H, D, dp, dp, 1.0f / sqrtf((float)dp), st)) { fprintf(stderr, "prepare_tables: %s\n", ss_last_error()); return 1; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float scale = 1.0f / sqrtf((float)dp);
    auto fwd = [&]() { return ss_relpos_attention_forward_p(SS_BF16, qkv, nullptr, E, tab, out, lse, pimg, B, H, T, Tp, dp, D, scale, pdrop, 77, 3, st); };
    auto bwd = [&]() { return ss_relpos_attention_backward_p(SS_BF16, qkv, nullptr, E, ET, tab, out, lse, dO, nullptr, dsc, dqkv, pimg, B, H, T, Tp, dp, D, scale, pdrop, 77, 3, st); };
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i) if ((pass ? bwd() : fwd())) { fprintf(stderr, "launch failed: %s\n", ss_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) (pass ? bwd() : fwd());
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / iters;
        // band-limited useful flops: 2*dp per (q, k) pair in the band x {QK, QE, PV} forward, x {QK, QE, dP, dQ(K), dQ(E), dK, dV} backward (one recompute counted once)
        double pairs = 0; for (int q = 0; q < T; ++q) for (int k = 0; k < T; ++k) if (abs(k - q) <= D - 1) pairs += 1;
        const double flops = pairs * 2.0 * dp * (pass ? 7.0 : 3.0) * B * H;
        printf("%s B=%d H=%d T=%d dp=%d D=%d p=%.2f : %.1f us  (%.1f TFLOP/s band-limited)\n", pass ? "backward" : "forward ", B, H, T, dp, D, pdrop, us, flops / us * 1e-6);
    }
    {   // SS_ATTN_T_STAMPS=1: phase time stamps of workgroup 0 of the last forward launch (s_memtime ticks = shader cycles)
        if ((fwd())) return 1;
        unsigned long long st8[8 * 4 * 8 * 2];
        if (ss_attn_t_debug_stamps(st8)) {
            const char* names[8] = {"top", "logits", "barrier A", "softmax", "-", "image+P~V", "O", "barrier B"};
            for (int w = 0; w < 7; ++w) for (int it = 0; it < 4; ++it) {
                const unsigned long long* r = st8 + (w * 4 + it) * 8;
                if (!r[7]) continue;
                printf("wave %d pair %d:", w, it);
                for (int k = 1; k < 8; ++k) printf("  %s %llu", names[k], r[k] - r[k - 1]);
                printf("  | total %llu\n", r[7] - r[0]);
            }
            for (int w = 0; w < 7; ++w) {
                const unsigned long long* r = st8 + 256 + w * 32;
                if (!r[0]) continue;
                printf("wave %d pair 1 logits iterations (skew-write+R | skew-read+S):", w);
                for (int ub = 0; ub < 8; ++ub) printf("  %llu|%llu", r[3 * ub + 1] - r[3 * ub], r[3 * ub + 2] - r[3 * ub + 1]);
                printf("\n");
            }
        }
    }
    // checksum so two builds can be compared
    std::vector<uint16_t> ho(nO); CK(hipMemcpy(ho.data(), out, nO * 2, hipMemcpyDeviceToHost));
    double cs = 0; for (size_t i = 0; i < nO; ++i) { uint32_t u = (uint32_t)ho[i] << 16; float f; memcpy(&f, &u, 4); cs += fabs((double)f); }
    std::vector<uint16_t> hg(nqkv); CK(hipMemcpy(hg.data(), dqkv, nqkv * 2, hipMemcpyDeviceToHost));
    double cg = 0; for (size_t i = 0; i < nqkv; ++i) { uint32_t u = (uint32_t)hg[i] << 16; float f; memcpy(&f, &u, 4); cg += fabs((double)f); }
    printf("checksum |O| = %.6e   |dqkv| = %.6e\n", cs, cg);
    return 0;
}
