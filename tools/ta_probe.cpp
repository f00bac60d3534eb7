// ta_probe.cpp -- how fast does a CU serve the load pattern of a DTW sweep that reads the cost matrix IN PLACE?  (tuning tool, no torch)
// A wave of the sweep needs, per wavefront step, 16 bytes per lane.  In the skewed strip layout that is one contiguous 1 KiB (mode 0).
// Read in place from a matrix whose unit-stride axis is the one the lanes own, lane l reads 16 bytes of line (t + 1 - l): 64 distinct
// cache lines per instruction, each line re-used by 8 neighbouring lanes over the next 8 steps (mode 1).  Mode 2: the same with 32-byte
// loads every second step (2 x dwordx4 from one line).  4 waves per workgroup = one matrix per CU, `nwg` workgroups.
// Prints ns and shader clocks per step and wave: the sweep's arithmetic costs ~140 clocks per step, the loads must stay below that.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ mat, float* out, int n, int steps, unsigned long long* clk)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* m = mat + (size_t)blockIdx.x * n * n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        const float* p = m + (size_t)w * steps * 256 + lane * 4;
        for (int t = 0; t < steps; t += 8) {
            f32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *(const f32x4*)(p + (size_t)(t + e) * 256);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += v[e];
        }
    } else {
        const float* p = m + w * 256 + lane * 4 + 1;              // columns 1 + 4 (64 w + lane) .. : dword-aligned, not 16-byte-aligned, like the real matrix
        for (int t = 0; t < steps; t += 8) {
            f32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int r = t + e + 1 - lane; r = r < 0 ? 0 : (r > n - 2 ? n - 2 : r);
                v[e] = *(const f32x4*)(p + (size_t)r * n);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += v[e];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, const float* mat, float* out, unsigned long long* clk, int nwg, int n)
{
    const int steps = (n - 1 + 63) / 8 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, mat, out, n, steps, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, mat, out, n, steps, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    unsigned long long h; CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    printf("%-44s nwg %3d: %8.1f us  %7.1f ns/step  %7.1f clk/step (wave 0 of wg 0)  %6.2f TB/s of matrix bytes\n", name, nwg, ms * 1e3, ms * 1e6 / steps,
           (double)h / steps, (double)nwg * n * n * 4 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv)
{
    const int n = 1000, maxwg = 256;
    float* mat; float* out; unsigned long long* clk;
    CK(hipMalloc(&mat, (size_t)maxwg * n * n * 4 + (1 << 20))); CK(hipMalloc(&out, maxwg * 256 * 4)); CK(hipMalloc(&clk, 8));
    CK(hipMemset(mat, 0, (size_t)maxwg * n * n * 4 + (1 << 20)));
    for (int nwg : {64, 256}) {
        run<0>("strip layout (1 KiB per step, coalesced)", mat, out, clk, nwg, n);
        run<1>("in place, 64 lines per load (diagonal)", mat, out, clk, nwg, n);
    }
    return 0;
}
