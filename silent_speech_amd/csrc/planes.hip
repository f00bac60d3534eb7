// planes.hip -- f32 tensors as two bf16 planes, hi = bf16(x) and lo = bf16(x - hi)  (x = hi + lo to 2^-17 relative).
//
// The parity-grade mode of the training plan (f32 storage between the kernels, every contraction on three bf16 MFMAs per product:
// a.b ~ a_lo.b_hi + a_hi.b_lo + a_hi.b_hi, gemm.hip: split_bf16x3) used to split its operands in registers on the 128 x 128 kernels, whose
// f32 LDS tiles made them copy-bound.  With the operands split ONCE, in memory, the 8-wave bf16 kernels run the same arithmetic as a three
// times longer contraction over the planes (gemm8.hip: PL; gemm8_dw.hip: three jobs per weight gradient).  This file is the split pass:
// an HBM-bound elementwise kernel, 4 bytes in and 4 bytes out per element.
// Reference arithmetic it serves: the f32 matmuls of architecture.py:61-84 and transformer.py:87-112.
#include "common.h"
#include "silent_speech_hip.h"

namespace {

// 8 elements per thread and trip: two 16-byte loads, one 16-byte store per plane
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long long n8, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f32x4 a = *(const f32x4*)(x + 8 * i), b = *(const f32x4*)(x + 8 * i + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 h, l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned hp = pack_bf16(v[2 * p], v[2 * p + 1]);
            h[p] = hp;
            l[p] = pack_bf16(v[2 * p] - __uint_as_float(hp << 16), v[2 * p + 1] - __uint_as_float(hp & 0xffff0000u));
        }
        *(u32x4*)(hi + 8 * i) = h; *(u32x4*)(lo + 8 * i) = l;
    }
    // tail (n % 8 elements): the first threads of block 0
    const long long t = 8 * n8 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        const float v = x[t];
        const bf16_t h = f2bf(v);
        hi[t] = h; lo[t] = f2bf(v - bf2f(h));
    }
}

}  // namespace

extern "C" int ss_split_planes(const float* x, void* hi, void* lo, int64_t n, void* stream)
{
    SS_CHECK(x && hi && lo, "ss_split_planes: null pointer");
    SS_CHECK(n >= 0, "ss_split_planes: negative length");
    SS_CHECK(((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) % 16 == 0, "ss_split_planes: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    const long long n8 = n / 8;
    long long blocks = (n8 + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    SS_LAUNCH(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, (bf16_t*)hi, (bf16_t*)lo, n8, (long long)n);
    SS_LAUNCH_CHECK("ss_split_planes");
    return 0;
}
