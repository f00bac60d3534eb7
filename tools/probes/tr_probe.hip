// Probe of ds_read_b64_tr_b16 semantics on gfx950: which lane receives which 16-bit element.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    int addr = mode == 0 ? l * 4                      // contiguous: lane l -> elements 4l..4l+3
                         : ((l & 15) >> 2) * 100 + (l & 3) * 4 + (l >> 4) * 1000;   // 4 rows of stride 100 per 16-lane group
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2); short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
