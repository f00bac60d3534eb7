// runtime.hip -- error plumbing and small layout utilities of the C ABI.
#include "common.h"
#include "silent_speech_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ss_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ss_last_error(void) { return g_err; }
extern "C" int ss_abi_version(void) { return 1; }
extern "C" const char* ss_target_arch(void) {
#if defined(SS_EMU)
    return "host-emulator";
#else
    return "gfx950";
#endif
}
int ss_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ss_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e)); return 2; }
    return 0;
}

// ---------------------------------------------------------------- ss_permute3d
template <class TI, class TO>
__global__ void permute3d_kernel(const TI* __restrict__ in, TO* __restrict__ out, int d0, int d1, int d2,
                                 long long s0, long long s1, long long s2, int valid1, int valid2, float scale, int accumulate)
{
    long long total = (long long)d0 * d1 * d2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % d2); long long t = i / d2; int b = (int)(t % d1); int a = (int)(t / d1);
        float v = (b < valid1 && c < valid2) ? ldf(in + a * s0 + b * s1 + c * s2) * scale : 0.f;
        if (accumulate) v += ldf(out + i);
        stf(out + i, v);
    }
}

extern "C" int ss_permute3d(const void* in, int in_dtype, void* out, int out_dtype, int d0, int d1, int d2,
                            int64_t s0, int64_t s1, int64_t s2, int valid1, int valid2, float scale, int accumulate, void* stream)
{
    SS_CHECK(in && out, "ss_permute3d: null pointer");
    SS_CHECK(d0 >= 0 && d1 >= 0 && d2 >= 0, "ss_permute3d: negative extent");
    long long total = (long long)d0 * d1 * d2;
    if (total == 0) return 0;
    int block = 256; long long g = (total + block - 1) / block; if (g > 4096) g = 4096;
    dim3 grid((unsigned)g), blk(block);
#define SS_P3(TI, TO) SS_LAUNCH(SS_KERNEL(permute3d_kernel<TI, TO>), grid, blk, 0, stream, (const TI*)in, (TO*)out, d0, d1, d2, (long long)s0, (long long)s1, (long long)s2, valid1, valid2, scale, accumulate)
    if (in_dtype == SS_F32 && out_dtype == SS_F32) SS_P3(float, float);
    else if (in_dtype == SS_F32 && out_dtype == SS_BF16) SS_P3(float, bf16_t);
    else if (in_dtype == SS_BF16 && out_dtype == SS_F32) SS_P3(bf16_t, float);
    else if (in_dtype == SS_BF16 && out_dtype == SS_BF16) SS_P3(bf16_t, bf16_t);
    else SS_CHECK(false, "ss_permute3d: bad dtype");
#undef SS_P3
    SS_LAUNCH_CHECK("ss_permute3d");
    return 0;
}

// ---------------------------------------------------------------- ss_permute3d_batch
// Many ss_permute3d jobs in ONE launch (the per-step weight re-layout / gradient un-layout is ~130 small tensors).
struct PermuteJob {       // mirrored by ctypes in silent_speech_amd/_lib.py
    const void* in; void* out;
    long long s0, s1, s2, o0, o1;        // input strides; output element (a,b,c) lives at a*o0 + b*o1 + c
    int d0, d1, d2, valid1, valid2, in_dtype, out_dtype, accumulate;
    float scale; int first_block, nblocks, pad_;
};

template <class TI, class TO>
__device__ __forceinline__ void permute_job(const PermuteJob& j, int lb, int nthreads, int tid)
{
    const long long total = (long long)j.d0 * j.d1 * j.d2;
    const TI* in = (const TI*)j.in; TO* out = (TO*)j.out;
    for (long long i = (long long)lb * nthreads + tid; i < total; i += (long long)j.nblocks * nthreads) {
        const int c = (int)(i % j.d2); const long long t = i / j.d2; const int b = (int)(t % j.d1); const int a = (int)(t / j.d1);
        float v = (b < j.valid1 && c < j.valid2) ? ldf(in + a * j.s0 + b * j.s1 + c * j.s2) * j.scale : 0.f;
        TO* o = out + a * j.o0 + b * j.o1 + c;
        if (j.accumulate) v += ldf(o);
        stf(o, v);
    }
}

__global__ void permute3d_batch_kernel(const PermuteJob* __restrict__ jobs, const int* __restrict__ job_of_block)
{
    const PermuteJob j = jobs[job_of_block[blockIdx.x]];
    const int lb = blockIdx.x - j.first_block;
    if (j.in_dtype == SS_F32 && j.out_dtype == SS_F32) permute_job<float, float>(j, lb, blockDim.x, threadIdx.x);
    else if (j.in_dtype == SS_F32) permute_job<float, bf16_t>(j, lb, blockDim.x, threadIdx.x);
    else if (j.out_dtype == SS_F32) permute_job<bf16_t, float>(j, lb, blockDim.x, threadIdx.x);
    else permute_job<bf16_t, bf16_t>(j, lb, blockDim.x, threadIdx.x);
}

extern "C" int ss_permute3d_batch(const void* jobs_dev, const int32_t* job_of_block_dev, int total_blocks, void* stream)
{
    SS_CHECK(total_blocks >= 0, "ss_permute3d_batch: negative block count");
    if (total_blocks == 0) return 0;
    SS_CHECK(jobs_dev && job_of_block_dev, "ss_permute3d_batch: null pointer");
    SS_LAUNCH(permute3d_batch_kernel, dim3(total_blocks), dim3(256), 0, stream, (const PermuteJob*)jobs_dev, (const int*)job_of_block_dev);
    SS_LAUNCH_CHECK("ss_permute3d_batch");
    return 0;
}
