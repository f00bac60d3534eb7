// g8_stamps: where the time of a gemm8_kc_kernel tile goes.  Links a measurement build of the library (tools/g8_stamps.sh: -DG8_STAMPS),
// runs one K-contiguous GEMM and prints, per item (= round) of the persistent workgroups, the wall-clock phases of a tile:
//   wait0 = tile start -> first K tile landed, kloop, drain = last MFMA -> every wave done with the stage, epilogue = C piece passes.
//   g8_stamps [M N K]     default 22000 2304 768
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "silent_speech_hip.h"
#include <dlfcn.h>
typedef int (*stamps_fn)(unsigned long long*);      // present in the measurement build only
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    int M = 22000, N = 2304, K = 768;
    if (argc > 3) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K); unsigned s = 1;
    for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = 0x3c00 + ((s >> 9) & 0x3ff); }
    for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = 0x3a00 + ((s >> 9) & 0x3ff); }
    void *A, *B, *C; CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    ss_rowmap am = {0, 0, K, 0}, bm = {0, 0, K, 0}, cm = {0, 0, N, 0};
    ss_gemm_epilogue e; memset(&e, 0, sizeof e); e.alpha = 1.f; e.gate_scale = 1.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // argv[4] = number of rotating output / A buffers (cold lines: inside the training step an output never overwrites a cache-resident buffer)
    const int rot = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<void*> Cs(rot, C), As(rot, A);
    for (int i = 1; i < rot; ++i) { CK(hipMalloc(&Cs[i], (size_t)M * N * 2)); CK(hipMalloc(&As[i], ha.size() * 2)); CK(hipMemcpy(As[i], A, ha.size() * 2, hipMemcpyDeviceToDevice)); }
    if (rot > 1) {
        for (int i = 0; i < 2 * rot; ++i) ss_gemm(SS_BF16, SS_BF16, SS_OP_KC, SS_OP_KC, As[i % rot], B, Cs[i % rot], M, N, K, &am, &bm, &cm, &e, 1, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 4 * rot; ++i) ss_gemm(SS_BF16, SS_BF16, SS_OP_KC, SS_OP_KC, As[i % rot], B, Cs[i % rot], M, N, K, &am, &bm, &cm, &e, 1, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float msr; CK(hipEventElapsedTime(&msr, e0, e1));
        printf("rotating over %d output / A buffers: %.1f us per launch (back to back)\n", rot, msr * 1e3 / (4 * rot));
    }
    for (int i = 0; i < 4; ++i) {
        CK(hipEventRecord(e0, 0));
        if (ss_gemm(SS_BF16, SS_BF16, SS_OP_KC, SS_OP_KC, As[i % rot], B, Cs[i % rot], M, N, K, &am, &bm, &cm, &e, 1, 0)) { fprintf(stderr, "%s\n", ss_last_error()); return 1; }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(256 * 8 * 8);
    stamps_fn stamps = (stamps_fn)dlsym(RTLD_DEFAULT, "ss_gemm8_debug_stamps");
    if (!stamps) { printf("M %d N %d K %d: kernel %d, %.1f us (event); this library carries no stamps\n", M, N, K, ss_gemm_last_kernel(), ms * 1e3); return 0; }
    if (!stamps(st.data())) { fprintf(stderr, "no stamps\n"); return 1; }
    const int tiles = ((M + 287) / 288) * ((N + 255) / 256);
    printf("M %d N %d K %d: kernel %d, %.1f us (event), %d tiles of 288 x 256 on 256 workgroups; 100 MHz ticks below, in us\n", M, N, K, ss_gemm_last_kernel(), ms * 1e3, tiles);
    unsigned long long t0 = ~0ull; for (int b = 0; b < 256; ++b) if (b < tiles) t0 = std::min(t0, st[(b * 8) * 8]);
    for (int item = 0; item < 8; ++item) {
        double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}; double first = 1e30, last = 0, endmin = 1e30, endmax = 0, mhz = 0; int n = 0;
        for (int b = 0; b < 256; ++b) {
            if (b + item * 256 >= tiles) continue;
            const unsigned long long* p = &st[(b * 8 + item) * 8];
            ++n;
            for (int k = 0; k < 4; ++k) { const double d = (double)(p[k + 1] - p[k]) * 0.01; sum[k] += d; mx[k] = std::max(mx[k], d); }
            const double a = (double)(p[0] - t0) * 0.01, z = (double)(p[4] - t0) * 0.01;
            first = std::min(first, a); last = std::max(last, a); endmin = std::min(endmin, z); endmax = std::max(endmax, z);
            sum[4] += z - a;
            mhz += (double)(p[6] - p[5]) / ((double)(p[2] - p[1]) * 0.01);
        }
        if (!n) break;
        printf("item %d (%3d workgroups): start %6.2f .. %6.2f  end %6.2f .. %6.2f | mean wait0 %5.2f  kloop %5.2f  drain %5.2f  epilogue %5.2f  tile %5.2f | max wait0 %5.2f kloop %5.2f drain %5.2f epilogue %5.2f | shader clock in the K loop %4.0f MHz\n",
               item, n, first, last, endmin, endmax, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, mx[0], mx[1], mx[2], mx[3], mhz / n);
    }
    return 0;
}
