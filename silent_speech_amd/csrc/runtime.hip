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
