// mfma_peak.cpp -- what the MI355X sustains on bf16 MFMA (16x16x32) when nothing else limits it, and what happens to that rate
// when LDS traffic of the size a GEMM main loop needs runs beside it.  Stand-alone (no torch): make -C tools mfma_peak.
//   mode 0: MFMAs only (register operands, 36 independent accumulator tiles per wave = the 288 x 256 tile of gemm8.hip)
//   mode 1: + the fragment reads of that tile (26 ds_read_b128 per 72 MFMAs per wave), results discarded
//   mode 2: + 8.5 LDS writes of 1 KiB per wave and 72 MFMAs (the copy traffic of one K tile; ds_write_b128 stands in for the DMA)
//   mode 3: the grouped weight-gradient kernel's traffic: 24 TRANSPOSING 8-byte reads (ds_read_b64_tr_b16) + 4 ds_write_b128 per 32 MFMAs and wave
//   mode 4: the same with 16 transposing reads (what 128 x 128 per wave would need)
// Prints TFLOP/s and the shader clock that rate implies (1024 flop / clk / SIMD), plus s_memtime / s_memrealtime deltas of one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* clk, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[9][4];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 4; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
    bf16x8 a[9], b[4];
    for (int i = 0; i < 9; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (lane + i + e));
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.002f * (lane + j + e));
    for (int i = threadIdx.x; i < 139264 / 16; i += 512) ((f32x4*)lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned base = (unsigned)(wave * 16384 + lane * 16);
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (MODE >= 1) {        // 13 reads of 1 KiB per wave and K half; results folded into one register that is never consumed by an MFMA
                f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 13; ++r) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(0)); sink += v; }
                asm volatile("" :: "v"(sink));
            }
            if (MODE >= 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { f32x4 v = acc[0][0]; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base + 8192), "v"(v), "n"(0) : "memory"); }
            }
            if (MODE == 3 || MODE == 4) {     // the weight-gradient kernel's K half: 24 (8 waves) / 32 per 64 MFMAs (4-wave form) transposing 8-byte reads, 4 LDS writes
                typedef short s16x4 __attribute__((ext_vector_type(4)));
                s16x4 sink = {0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < (MODE == 3 ? 24 : 16); ++r) { s16x4 v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(0)); sink += v; }
                asm volatile("" :: "v"(sink));
#pragma unroll
                for (int r = 0; r < 4; ++r) { f32x4 v = acc[0][0]; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base + 8192), "v"(v), "n"(0) : "memory"); }
            }
#pragma unroll
            for (int i = 0; i < (MODE >= 3 ? 8 : 9); ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// the same probe on 32x32x16 MFMAs: a 128 x 64 wave tile (4 x 2 accumulator tiles of 16 registers), 16 MFMAs per K half of 32;
// READS: + 12 fragment reads of 1 KiB per wave and K half (what that tile needs), as above never consumed by an MFMA
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int READS>
__global__ __launch_bounds__(512) void probe32(float* out, unsigned long long* clk, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (lane + i + e));
    for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.002f * (lane + j + e));
    for (int i = threadIdx.x; i < 139264 / 16; i += 512) ((f32x4*)lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned base = (unsigned)(wave * 16384 + lane * 16);
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (READS) {
                f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 12; ++r) { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(0)); sink += v; }
                asm volatile("" :: "v"(sink));
            }
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int READS>
static void run32(const char* name, int iters)
{
    int dev = 0, cus = 0; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* out; unsigned long long* clk; CK(hipMalloc(&out, (size_t)cus * 512 * 4)); CK(hipMalloc(&clk, 16));
    CK(hipFuncSetAttribute((const void*)probe32<READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe32<READS>, dim3(cus), dim3(512), 139264, 0, out, clk, iters / 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe32<READS>, dim3(cus), dim3(512), 139264, 0, out, clk, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flops = (double)cus * 8 * iters * 32.0 * 32 * 32 * 16 * 2;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %8.1f TFLOP/s  => %.2f GHz at 1024 flop/clk/SIMD;  s_memtime %llu, s_memrealtime %llu (ratio %.2f)\n", name, ms, tf,
           tf * 1e12 / (cus * 4 * 1024.0) / 1e9, h[0], h[1], h[1] ? (double)h[0] / h[1] : 0.0);
    CK(hipFree(out)); CK(hipFree(clk));
}

template <int MODE>
static void run(const char* name, int iters)
{
    int dev = 0, cus = 0; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* out; unsigned long long* clk; CK(hipMalloc(&out, (size_t)cus * 512 * 4)); CK(hipMalloc(&clk, 16));
    CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 139264, 0, out, clk, iters / 8);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 139264, 0, out, clk, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flops = (double)cus * 8 * iters * (MODE >= 3 ? 64.0 : 72.0) * 16 * 16 * 32 * 2;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %8.1f TFLOP/s  => %.2f GHz at 1024 flop/clk/SIMD;  s_memtime %llu, s_memrealtime %llu (ratio %.2f)\n", name, ms, tf,
           tf * 1e12 / (cus * 4 * 1024.0) / 1e9, h[0], h[1], h[1] ? (double)h[0] / h[1] : 0.0);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    run<0>("MFMA only", iters);
    run<1>("MFMA + fragment reads", iters);
    run<2>("MFMA + fragment reads + LDS writes", iters);
    run<3>("dW K half: 24 tr reads + 4 writes / 32 MFMA", iters);
    run<4>("dW 4-wave form: 16 tr reads / 32 MFMA", iters);
    run<0>("MFMA only (again)", iters);
    run32<0>("32x32x16 MFMA only", iters);
    run32<1>("32x32x16 MFMA + fragment reads", iters);
    run<0>("MFMA only (16x16x32, again)", iters);
    run32<0>("32x32x16 MFMA only (again)", iters);
    return 0;
}
