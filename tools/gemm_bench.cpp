// gemm_bench.cpp -- stand-alone (no torch, starts in a second on a fresh GPU box) timing + cross-check of the ss_gemm kernel
// variants on the GEMM shapes of the reference-size training step.  Build: make -C tools gemm_bench (hipcc, links the C ABI).
//   gemm_bench [iters]      prints one line per (shape, variant): microseconds, TFLOP/s, max |diff| vs the 128-wide kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "silent_speech_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define SS(x) do { if ((x) != 0) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, ss_last_error()); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned rng_state = 12345u;
static float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

static void* dev_bf16(size_t n, float scale) {
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(rnd() * scale);
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static ss_rowmap plain(long long ld) { ss_rowmap m; m.base = 0; m.batch_stride = 0; m.row_stride = ld; m.rows_per_batch = 0; return m; }

static float time_us(int iters, void (*fn)(void*), void* arg) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn(arg);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) fn(arg);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1000.f / iters;
}

struct KcArgs { void *A, *B, *C; int M, N, K; ss_rowmap am, bm, cm; ss_gemm_epilogue epi; };
static void run_kc(void* p) {
    KcArgs* a = (KcArgs*)p;
    SS(ss_gemm(SS_BF16, SS_BF16, SS_OP_KC, SS_OP_KC, a->A, a->B, a->C, a->M, a->N, a->K, &a->am, &a->bm, &a->cm, &a->epi, 1, 0));
}

static double max_diff_bf16(const void* d0, const void* d1, size_t n) {
    std::vector<unsigned short> h0(n), h1(n);
    CK(hipMemcpy(h0.data(), d0, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, n * 2, hipMemcpyDeviceToHost));
    double m = 0; size_t nbad = 0;
    for (size_t i = 0; i < n; ++i) { double d = fabs((double)bf2f(h0[i]) - (double)bf2f(h1[i])); if (!(d <= m)) m = d; if (h0[i] != h1[i]) ++nbad; }
    if (nbad) fprintf(stdout, "      (%zu of %zu elements differ bitwise)\n", nbad, n);
    return m;
}

static void bench_kc(const char* tag, int M, int N, int K, int iters, bool relu_bias) {
    KcArgs a; memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K;
    a.A = dev_bf16((size_t)M * K, 1.0f); a.B = dev_bf16((size_t)N * K, 0.05f);
    void *Cref, *Cnew; CK(hipMalloc(&Cref, (size_t)M * N * 2)); CK(hipMalloc(&Cnew, (size_t)M * N * 2));
    a.am = plain(K); a.bm = plain(K); a.cm = plain(N);
    a.epi.alpha = 1.f; a.epi.gate_scale = 1.f;
    float* bias = nullptr;
    if (relu_bias) { std::vector<float> hb(N); for (int i = 0; i < N; ++i) hb[i] = rnd(); CK(hipMalloc((void**)&bias, N * 4)); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice)); a.epi.bias = bias; a.epi.relu = 1; }
    const double flops = 2.0 * M * N * K;
    // reference: the 128-wide kernels
    ss_gemm_set_option(2, 0);
    a.C = Cref; CK(hipMemset(Cref, 0, (size_t)M * N * 2));
    float t = time_us(iters, run_kc, &a);
    printf("%-10s M=%6d N=%5d K=%5d  %-14s kernel %d  %8.1f us  %7.1f TF\n", tag, M, N, K, "128-wide", ss_gemm_last_kernel(), t, flops / t / 1e6);
    for (int ni = 8; ni <= 9; ++ni)
        for (int pin = 0; pin <= 3; pin += 3) {              // burst / spread schedule
            ss_gemm_set_option(2, 2); ss_gemm_set_option(3, ni); ss_gemm_set_option(4, pin);
            a.C = Cnew; CK(hipMemset(Cnew, 0xff, (size_t)M * N * 2));
            t = time_us(iters, run_kc, &a);
            CK(hipDeviceSynchronize());
            char name[32]; snprintf(name, sizeof name, "gemm8 ni%d pin%d", ni, pin);
            const int k = ss_gemm_last_kernel();
            const double d = max_diff_bf16(Cref, Cnew, (size_t)M * N);
            printf("%-10s M=%6d N=%5d K=%5d  %-14s kernel %d  %8.1f us  %7.1f TF   max|diff| %.3g\n", tag, M, N, K, name, k, t, flops / t / 1e6, d);
        }
    ss_gemm_set_option(2, -1); ss_gemm_set_option(3, -1); ss_gemm_set_option(4, -1);
    // what the cost model picks
    a.C = Cnew;
    t = time_us(iters, run_kc, &a);
    printf("%-10s M=%6d N=%5d K=%5d  %-14s kernel %d  %8.1f us  %7.1f TF\n", tag, M, N, K, "auto", ss_gemm_last_kernel(), t, flops / t / 1e6);
    CK(hipFree(a.A)); CK(hipFree(a.B)); CK(hipFree(Cref)); CK(hipFree(Cnew)); if (bias) CK(hipFree(bias));
    fflush(stdout);
}

struct DwArgs { int n; ss_dw_job jobs[8]; };
static void run_dw_grouped(void* p) { DwArgs* a = (DwArgs*)p; SS(ss_gemm_dw_grouped(a->n, a->jobs, 0)); }
static void run_dw_old(void* p) {
    DwArgs* a = (DwArgs*)p;
    for (int i = 0; i < a->n; ++i) {
        const ss_dw_job& j = a->jobs[i];
        ss_gemm_epilogue e; memset(&e, 0, sizeof(e)); e.alpha = 1.f; e.gate_scale = 1.f; e.mode = 2;
        ss_rowmap cm = plain(j.ldc);
        const int tiles = ((j.M + 127) / 128) * ((j.N + 127) / 128);
        int split = 512 / (tiles > 0 ? tiles : 1); if (split < 1) split = 1; if (split > 64) split = 64; if (split > j.K / 512) split = j.K / 512 > 0 ? j.K / 512 : 1;
        SS(ss_gemm(SS_BF16, SS_F32, SS_OP_OC, SS_OP_OC, j.A, j.B, j.C, j.M, j.N, j.K, &j.amap, &j.bmap, &cm, &e, split, 0));
    }
}

static void bench_dw(const char* tag, int rows, int n, const int (*mn)[2], int iters) {
    DwArgs a; memset(&a, 0, sizeof(a)); a.n = n;
    DwArgs b = a;
    double flops = 0; size_t total = 0;
    for (int i = 0; i < n; ++i) {
        ss_dw_job& j = a.jobs[i];
        j.M = mn[i][0]; j.N = mn[i][1]; j.K = rows;
        j.A = dev_bf16((size_t)rows * j.M, 0.05f); j.B = dev_bf16((size_t)rows * j.N, 1.0f);
        CK(hipMalloc((void**)&j.C, (size_t)j.M * j.N * 4)); CK(hipMemset(j.C, 0, (size_t)j.M * j.N * 4));
        j.amap = plain(j.M); j.bmap = plain(j.N); j.ldc = j.N;
        b.jobs[i] = j; CK(hipMalloc((void**)&b.jobs[i].C, (size_t)j.M * j.N * 4)); CK(hipMemset(b.jobs[i].C, 0, (size_t)j.M * j.N * 4));
        flops += 2.0 * rows * j.M * j.N; total += (size_t)j.M * j.N;
    }
    float t_old = time_us(iters, run_dw_old, &b);
    printf("%-10s rows=%6d jobs=%d  %-22s %8.1f us  %7.1f TF\n", tag, rows, n, "128-wide TR, per GEMM", t_old, flops / t_old / 1e6);
    const int kts[3] = {0, 4, 2};                       // transposing-read kernel (rounds 2-3) / K-contiguous tiles, 4 or 2 MFMA row tiles per phase
    for (int v = 0; v < 3; ++v) {
        ss_gemm_dw_set_option(3, kts[v]);
        float t = time_us(iters, run_dw_grouped, &a);
        char name[40]; snprintf(name, sizeof name, kts[v] ? "grouped KT hr%d" : "grouped TR (r3)", kts[v]);
        printf("%-10s rows=%6d jobs=%d  %-22s %8.1f us  %7.1f TF\n", tag, rows, n, name, t, flops / t / 1e6);
    }
    ss_gemm_dw_set_option(3, -1);
    ss_gemm_dw_set_option(1, 1);
    // cross-check one accumulation of each (buffers re-zeroed)
    for (int i = 0; i < n; ++i) { CK(hipMemset(a.jobs[i].C, 0, (size_t)a.jobs[i].M * a.jobs[i].N * 4)); CK(hipMemset(b.jobs[i].C, 0, (size_t)a.jobs[i].M * a.jobs[i].N * 4)); }
    run_dw_grouped(&a); run_dw_old(&b); CK(hipDeviceSynchronize());
    for (int i = 0; i < n; ++i) {
        const size_t cnt = (size_t)a.jobs[i].M * a.jobs[i].N;
        std::vector<float> h0(cnt), h1(cnt);
        CK(hipMemcpy(h0.data(), a.jobs[i].C, cnt * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), b.jobs[i].C, cnt * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; for (size_t k = 0; k < cnt; ++k) { md = fmax(md, fabs((double)h0[k] - h1[k])); mx = fmax(mx, fabs((double)h1[k])); }
        printf("      job %d (%d x %d): max|diff| %.3g of max %.3g\n", i, a.jobs[i].M, a.jobs[i].N, md, mx);
    }
    for (int i = 0; i < n; ++i) { CK(hipFree((void*)a.jobs[i].A)); CK(hipFree((void*)a.jobs[i].B)); CK(hipFree(a.jobs[i].C)); CK(hipFree(b.jobs[i].C)); }
    fflush(stdout);
}

// what the heavier epilogues of the step cost on top of the plain GEMM of the same shape (22 000 x 3072 x 768: FFN1 forward with bias + ReLU +
// dropout; its input gradient with the ReLU / dropout gate read from the saved activation and the column sums of linear1.bias.grad; C += v)
static void bench_epi(int iters) {
    const int M = 22000, N = 3072, K = 768;
    KcArgs a; memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K;
    a.A = dev_bf16((size_t)M * K, 1.0f); a.B = dev_bf16((size_t)N * K, 0.05f);
    CK(hipMalloc(&a.C, (size_t)M * N * 2)); CK(hipMemset(a.C, 0, (size_t)M * N * 2));
    void* gate = dev_bf16((size_t)M * N, 1.0f);
    float *bias, *cs; std::vector<float> hb(N); for (int i = 0; i < N; ++i) hb[i] = rnd();
    CK(hipMalloc((void**)&bias, N * 4)); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMalloc((void**)&cs, N * 4)); CK(hipMemset(cs, 0, N * 4));
    a.am = plain(K); a.bm = plain(K); a.cm = plain(N);
    const double flops = 2.0 * M * N * K;
    const char* names[6] = {"plain", "bias+relu", "bias+relu+dropout", "gate", "gate+colsum", "C += v"};
    for (int v = 0; v < 6; ++v) {
        memset(&a.epi, 0, sizeof(a.epi)); a.epi.alpha = 1.f; a.epi.gate_scale = 1.f;
        if (v == 1 || v == 2) { a.epi.bias = bias; a.epi.relu = 1; }
        if (v == 2) { a.epi.dropout_p = 0.2f; a.epi.seed = 77; a.epi.rng_stream = 3; }
        if (v == 3 || v == 4) { a.epi.gate = gate; a.epi.gate_scale = 1.25f; }
        if (v == 4) a.epi.col_sum = cs;
        if (v == 5) a.epi.mode = 1;
        const float t = time_us(iters, run_kc, &a);
        if (v == 5) CK(hipMemset(a.C, 0, (size_t)M * N * 2));
        run_kc(&a); if (v == 5) run_kc(&a);
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> hc((size_t)M * N); CK(hipMemcpy(hc.data(), a.C, hc.size() * 2, hipMemcpyDeviceToHost));
        double cks = 0; for (size_t i = 0; i < hc.size(); i += 7) cks += fabs((double)bf2f(hc[i]));
        printf("epilogue   M=%6d N=%5d K=%5d  %-18s kernel %d  %8.1f us  %7.1f TF   checksum %.9e\n", M, N, K, names[v], ss_gemm_last_kernel(), t, flops / t / 1e6, cks);
    }
    // where the gate's cost comes from: the same two launches with every output row mapped onto row 0 (the C stores and the gate loads then hit L2:
    // what is left of the difference is instruction / latency cost, what disappeared was HBM traffic of the epilogue bursts)
    for (int v = 0; v < 2; ++v) {
        memset(&a.epi, 0, sizeof(a.epi)); a.epi.alpha = 1.f; a.epi.gate_scale = 1.f;
        if (v == 1) { a.epi.gate = gate; a.epi.gate_scale = 1.25f; }
        a.cm = plain(0);
        const float t = time_us(iters, run_kc, &a);
        printf("epilogue   M=%6d N=%5d K=%5d  %-18s kernel %d  %8.1f us  %7.1f TF   (timing only: all output rows aliased)\n", M, N, K, v ? "gate, rows -> 0" : "plain, rows -> 0", ss_gemm_last_kernel(), t, flops / t / 1e6);
        a.cm = plain(N);
    }
    CK(hipFree(a.A)); CK(hipFree(a.B)); CK(hipFree(a.C)); CK(hipFree(gate)); CK(hipFree(bias)); CK(hipFree(cs));
    // the output map of a convolution: (B, T + 2, C) with a zero halo row at each end of every sequence -> a division per output row
    {
        const int Bq = 110, T = 400, Cc = 768, Mc = Bq * T, Kc = 2304;
        KcArgs c; memset(&c, 0, sizeof(c));
        c.M = Mc; c.N = Cc; c.K = Kc;
        c.A = dev_bf16((size_t)Mc * Kc, 1.0f); c.B = dev_bf16((size_t)Cc * Kc, 0.05f);
        CK(hipMalloc(&c.C, (size_t)Bq * (T + 2) * Cc * 2)); CK(hipMemset(c.C, 0, (size_t)Bq * (T + 2) * Cc * 2));
        c.am = plain(Kc); c.bm = plain(Kc); c.epi.alpha = 1.f; c.epi.gate_scale = 1.f;
        for (int v = 0; v < 2; ++v) {
            if (v == 0) c.cm = plain(Cc);
            else { c.cm.base = Cc; c.cm.batch_stride = (long long)(T + 2) * Cc; c.cm.row_stride = Cc; c.cm.rows_per_batch = T; }
            const float t = time_us(iters, run_kc, &c);
            printf("conv out   M=%6d N=%5d K=%5d  %-18s kernel %d  %8.1f us  %7.1f TF\n", Mc, Cc, Kc, v ? "halo rows (B,T+2,C)" : "plain rows", ss_gemm_last_kernel(), t, 2.0 * Mc * Cc * Kc / t / 1e6);
        }
        CK(hipFree(c.A)); CK(hipFree(c.B)); CK(hipFree(c.C));
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const char* only = argc > 2 ? argv[2] : "";
    printf("arch %s, abi %d\n", ss_target_arch(), ss_abi_version());
    if (!*only || strstr(only, "kc")) {
        const int M = 22000;                                   // B*T of the reference-size batch (110 rows of 200 frames)
        bench_kc("lin768", M, 768, 768, iters, true);          // w_raw_in, W_o, dX of QKV..., x13 per step
        bench_kc("qkv", M, 2304, 768, iters, false);
        bench_kc("ffn1", M, 3072, 768, iters, true);
        bench_kc("ffn2", M, 768, 3072, iters, true);
        bench_kc("dqkv", M, 768, 2304, iters, false);
        bench_kc("conv2@800", 88000, 768, 2304, iters, true);  // ResBlock 0 conv2 (rows = 110 x 800)
        bench_kc("conv@400", 44000, 768, 2304, iters, true);
        bench_kc("res@400", 44000, 768, 768, iters, true);
        bench_kc("heads", M, 128, 768, iters, true);
        bench_kc("dheads", M, 768, 128, iters, false);
    }
    if (strstr(only, "epi")) bench_epi(iters);
    if (strstr(only, "abl")) {        // where does the time of the 8-wave kernel go?  (results are wrong under a non-zero mask)
        const int masks[7] = {0, 16, 32, 64, 16 | 64, 32 | 64, 16 | 32 | 64};
        const char* names[7] = {"full", "no mfma", "no glds", "no frag reads", "glds only", "mfma only", "skeleton"};
        const int shapes[3][3] = {{22000, 768, 768}, {22000, 768, 3072}, {88000, 768, 2304}};
        for (int sh = 0; sh < 3; ++sh)
            for (int ni = 8; ni <= 9; ++ni) {
                KcArgs a; memset(&a, 0, sizeof(a));
                a.M = shapes[sh][0]; a.N = shapes[sh][1]; a.K = shapes[sh][2];
                a.A = dev_bf16((size_t)a.M * a.K, 1.0f); a.B = dev_bf16((size_t)a.N * a.K, 0.05f);
                CK(hipMalloc(&a.C, (size_t)a.M * a.N * 2));
                a.am = plain(a.K); a.bm = plain(a.K); a.cm = plain(a.N); a.epi.alpha = 1.f; a.epi.gate_scale = 1.f;
                ss_gemm_set_option(2, 2); ss_gemm_set_option(3, ni); ss_gemm_set_option(4, 0);
                for (int m = 0; m < 7; ++m) {
                    ss_gemm_set_option(5, masks[m]);
                    const float t = time_us(iters, run_kc, &a);
                    printf("ablate M=%6d N=%5d K=%5d ni%d  %-14s %8.1f us\n", a.M, a.N, a.K, ni, names[m], t);
                }
                ss_gemm_set_option(5, 0);
                CK(hipFree(a.A)); CK(hipFree(a.B)); CK(hipFree(a.C));
                fflush(stdout);
            }
        ss_gemm_set_option(2, -1); ss_gemm_set_option(3, -1); ss_gemm_set_option(4, -1);
    }
    if (!*only || strstr(only, "dw")) {
        const int layer[4][2] = {{768, 3072}, {3072, 768}, {768, 768}, {2304, 768}};
        bench_dw("dW layer", 22000, 4, layer, iters);
        const int blk1[3][2] = {{768, 2304}, {768, 2304}, {768, 768}};
        bench_dw("dW block1", 44000, 3, blk1, iters);
        const int blk0[1][2] = {{768, 2304}};
        bench_dw("dW b0conv2", 88000, 1, blk0, iters);
        const int raw[1][2] = {{768, 768}};
        bench_dw("dW raw_in", 22000, 1, raw, iters);
    }
    return 0;
}
