// dtw_bench: times ss_dtw_align on a batch of square f32 cost matrices through the C ABI, no torch (BASELINE configs[2]).
//   dtw_bench [nb n iters]         default 64 1000 10;  SS_DTW_DEBUG=8 forces the skewed-strip copy instead of the in-place sources
// Each batch is aligned twice: as row-major matrices (strides (n, 1): the lanes own columns) and as the column-major view of the
// same memory (strides (1, n): the lanes own rows, i.e. the alignment of the transposed matrices).  Prints time, matrices / s, the
// 8 N M-byte roofline figure and a checksum of the results, so that two builds / two sources can be compared bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "silent_speech_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    int nb = 64, n = 1000, iters = 10;
    if (argc > 1) nb = atoi(argv[1]); if (argc > 2) n = atoi(argv[2]); if (argc > 3) iters = atoi(argv[3]);
    const size_t cells = (size_t)nb * n * n;
    std::vector<float> h(cells); uint32_t s = 2024;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((s >> 8) & 0xffffff) / 16777216.0f; }
    float* costs; CK(hipMalloc(&costs, cells * 4)); CK(hipMemcpy(costs, h.data(), cells * 4, hipMemcpyHostToDevice));
    int64_t sk, dr, bd; const int64_t per = ss_dtw_workspace_bytes(n, n, &sk, &dr, &bd);
    void* ws; CK(hipMalloc(&ws, (size_t)per * nb));
    int32_t* res; CK(hipMalloc(&res, (size_t)nb * n * 4));
    int64_t* desc; CK(hipMalloc(&desc, (size_t)nb * 10 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool alias = getenv("DTW_BENCH_ALIAS") != nullptr;      // every problem reads matrix 0 (L2-resident costs: separates HBM latency from the rest)
    for (int orient = 0; orient < 2; ++orient) {
        std::vector<int64_t> hd((size_t)nb * 10, 0);
        for (int b = 0; b < nb; ++b) {
            int64_t* d = &hd[(size_t)b * 10];
            d[0] = n; d[1] = n; d[2] = alias ? 0 : (int64_t)b * n * n; d[3] = orient ? 1 : n; d[4] = orient ? n : 1;
            d[5] = (int64_t)b * per; d[6] = d[5] + sk; d[7] = d[6] + dr; d[8] = (int64_t)b * n;
        }
        CK(hipMemcpy(desc, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
        for (int i = 0; i < 2; ++i) if (ss_dtw_align(costs, desc, nb, n, n, ws, res, st)) { fprintf(stderr, "ss_dtw_align: %s\n", ss_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) ss_dtw_align(costs, desc, nb, n, n, ws, res, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        std::vector<int32_t> hr((size_t)nb * n); CK(hipMemcpy(hr.data(), res, hr.size() * 4, hipMemcpyDeviceToHost));
        uint64_t cs = 1469598103934665603ull; for (int32_t v : hr) cs = (cs ^ (uint32_t)v) * 1099511628211ull;
        printf("%3d x %d x %d  %-28s source %d : %8.1f us  %9.0f matrices/s  %6.3f TB/s of 8NM bytes (%.1f %% of 8 TB/s)  results %016llx\n", nb, n, n,
               orient ? "column-major (lanes own rows)" : "row-major (lanes own columns)", ss_dtw_source(n, n, orient ? 1 : n, orient ? n : 1), ms * 1e3, nb / (ms * 1e-3),
               8.0 * cells / (ms * 1e-3) / 1e12, 8.0 * cells / (ms * 1e-3) / 8e12 * 100.0, (unsigned long long)cs);
    }
    return 0;
}
