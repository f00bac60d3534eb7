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
extern "C" int ss_abi_version(void) { return SS_ABI_VERSION; }
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

// ---------------------------------------------------------------- host tables -> device, from memory that outlives the call
// A host-built table that a kernel sequence reads must not be copied from a local with hipMemcpyAsync: a pageable source is only safe
// as long as the runtime stages it before returning, which it does not promise (and does not do under stream capture).  The table is
// copied into a slot of a small process-wide ring of pinned buffers first; a slot is reused only after the event recorded behind its
// last copy has completed.
#if !defined(SS_EMU)
#include <mutex>
namespace {
struct PinnedRing {
    static constexpr int N = 8;
    void* buf[N] = {}; size_t cap[N] = {}; hipEvent_t ev[N] = {}; bool used[N] = {}; int at = 0; std::mutex mu;
};
PinnedRing g_ring;
}
#endif
int ss_upload_table(void* dst_dev, const void* src_host, size_t bytes, void* stream)
{
    if (bytes == 0) return 0;
#if defined(SS_EMU)
    memcpy(dst_dev, src_host, bytes);
    return 0;
#else
    std::lock_guard<std::mutex> lock(g_ring.mu);
    const int i = g_ring.at; g_ring.at = (i + 1) % PinnedRing::N;
    if (g_ring.used[i] && hipEventSynchronize(g_ring.ev[i]) != hipSuccess) { ss_set_error("table upload: event wait failed"); return 1; }
    if (g_ring.cap[i] < bytes) {
        if (g_ring.buf[i]) (void)hipHostFree(g_ring.buf[i]);
        size_t cap = 1 << 16; while (cap < bytes) cap <<= 1;
        if (hipHostMalloc(&g_ring.buf[i], cap, hipHostMallocDefault) != hipSuccess) { g_ring.buf[i] = nullptr; g_ring.cap[i] = 0; ss_set_error("table upload: pinned allocation of %zu bytes failed", cap); return 1; }
        g_ring.cap[i] = cap;
    }
    if (!g_ring.ev[i] && hipEventCreateWithFlags(&g_ring.ev[i], hipEventDisableTiming) != hipSuccess) { ss_set_error("table upload: event creation failed"); return 1; }
    memcpy(g_ring.buf[i], src_host, bytes);
    if (hipMemcpyAsync(dst_dev, g_ring.buf[i], bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) { ss_set_error("table upload failed"); return 1; }
    if (hipEventRecord(g_ring.ev[i], (hipStream_t)stream) != hipSuccess) { ss_set_error("table upload: event record failed"); return 1; }
    g_ring.used[i] = true;
    return 0;
#endif
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

// Three paths per job (wave-uniform choice):
//   rows      input contiguous along c (s2 == 1): 4 elements per thread, 16-byte loads
//   transpose input contiguous along a or b, strided along c: 64 x 64 tiles through LDS, both sides coalesced
//             (the per-step weight re-layouts are mostly [n][k] -> [k][n] and (O, I, 3) -> (O, 3, I) of this kind)
//   generic   anything else (negative strides of the flipped-tap conv forms, tiny jobs)
template <class TI> struct Vec4;
template <> struct Vec4<float> { static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const f32x4 a = *(const f32x4*)p; v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; } };
template <> struct Vec4<bf16_t> { static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) { const u32x2 a = *(const u32x2*)p; v[0] = __uint_as_float(a[0] << 16); v[1] = __uint_as_float(a[0] & 0xffff0000u); v[2] = __uint_as_float(a[1] << 16); v[3] = __uint_as_float(a[1] & 0xffff0000u); } };

template <class TI, class TO, bool GROUPED_RMW = false>
__device__ __forceinline__ void permute_job(const PermuteJob& j, int lb, int nthreads, int tid, float* tile)
{
    const long long total = (long long)j.d0 * j.d1 * j.d2;
    const TI* in = (const TI*)j.in; TO* out = (TO*)j.out;
    // X = the outer dimension along which the input is (nearly) contiguous: |stride| <= 4 elements, the longer one if both are
    const long long as0 = j.s0 < 0 ? -j.s0 : j.s0, as1 = j.s1 < 0 ? -j.s1 : j.s1, as2 = j.s2 < 0 ? -j.s2 : j.s2;
    const bool c0 = j.d0 > 1 && as0 >= 1 && as0 <= 4, c1 = j.d1 > 1 && as1 >= 1 && as1 <= 4;
    const bool fx1 = c1 && (!c0 || j.d1 >= j.d0), fx0 = c0 && !fx1;
    const int dXq = fx1 ? j.d1 : j.d0;
    const bool al16 = ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    // ---- cast path: the same dense layout on both sides (the plain [N][K] weights, more than half of the parameters): one linear
    // index, 8 elements per thread, no index arithmetic at all (the row path below spends two 64-bit divisions per 4 elements)
    if (!GROUPED_RMW && !j.accumulate && j.s2 == 1 && j.s1 == j.d2 && j.o1 == j.d2 && (j.d0 == 1 || (j.s0 == (long long)j.d1 * j.d2 && j.o0 == j.s0)) &&
        j.valid1 >= j.d1 && j.valid2 >= j.d2 && (total & 7) == 0 && al16 && total < (1LL << 34)) {
        const unsigned n8 = (unsigned)(total >> 3), step = (unsigned)j.nblocks * nthreads;
        for (unsigned i = (unsigned)lb * nthreads + tid; i < n8; i += step) {
            float v[8]; Vec8<TI>::load(in + (size_t)i * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= j.scale;
            Vec8<TO>::store(out + (size_t)i * 8, v);
        }
        return;
    }
    // ---- vector transpose path (16-bit output): input unit-stride along X, output along c, everything in multiples of 8 elements.
    // 64 (X) x 64 (c) tiles: 16/32-byte loads along X, a 16-bit LDS tile, 16-byte stores along c -- the scalar transpose path below
    // moves 2 bytes per lane and instruction on both sides.
    if (!GROUPED_RMW && sizeof(TO) == 2 && !j.accumulate && as2 >= 8 && (fx0 || fx1) && nthreads == 256 && al16) {
        const int dX = fx1 ? j.d1 : j.d0, dO = fx1 ? j.d0 : j.d1;
        const long long sX = fx1 ? j.s1 : j.s0, sO = fx1 ? j.s0 : j.s1, oX = fx1 ? j.o1 : j.o0, oO = fx1 ? j.o0 : j.o1;
        const int vX = fx1 ? (j.valid1 < j.d1 ? j.valid1 : j.d1) : dX, vO = fx1 ? dO : (j.valid1 < j.d1 ? j.valid1 : j.d1), vC = j.valid2 < j.d2 ? j.valid2 : j.d2;
        if (sX == 1 && !(dX & 7) && !(j.d2 & 7) && !(sO & 7) && !(j.s2 & 7) && !(oX & 7) && !(oO & 7) && !(vX & 7) && !(vC & 7) && dX >= 64 && j.d2 >= 64) {
            unsigned short* t16 = (unsigned short*)tile;                    // [64 c][66]
            const int tiles_x = (dX + 63) >> 6, tiles_c = (j.d2 + 63) >> 6;
            const long long ntiles = (long long)dO * tiles_x * tiles_c;
            for (long long t = lb; t < ntiles; t += j.nblocks) {
                const int oi = (int)(t / (tiles_x * tiles_c)), rem = (int)(t - (long long)oi * (tiles_x * tiles_c));
                const int tx = rem / tiles_c, tc = rem - tx * tiles_c;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int f = k * 256 + tid, cc = f >> 3, x0 = (f & 7) * 8, gx = tx * 64 + x0, gc = tc * 64 + cc;
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (gx < vX && gc < vC && oi < vO) { Vec8<TI>::load(in + oi * sO + gx + (long long)gc * j.s2, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= j.scale; }
                    unsigned* d = (unsigned*)(t16 + cc * 66 + x0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = pack_bf16(v[2 * e], v[2 * e + 1]);
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int f = k * 256 + tid, x = f >> 3, c0 = (f & 7) * 8, gx = tx * 64 + x, gc = tc * 64 + c0;
                    if (gx < dX && gc < j.d2) {
                        u32x4 r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = (unsigned)t16[(c0 + 2 * e) * 66 + x] | ((unsigned)t16[(c0 + 2 * e + 1) * 66 + x] << 16);
                        *(u32x4*)(out + oi * oO + gx * oX + gc) = r;
                    }
                }
                __syncthreads();
            }
            return;
        }
    }
    // (a 4096-element tile must be reasonably full: an (O, 8, 3) conv weight would put 24 elements in each and its few
    //  workgroups would walk hundreds of almost empty tiles -- the generic path is faster there)
    // ---- vector transpose path, f32 -> f32 (the gradient un-layout: GEMM-layout weight gradients accumulated into the parameter-layout
    // .grad arena): 64 x 64 tiles, 16-byte loads along X, 16-byte read-modify-write along c
    if (sizeof(TI) == 4 && sizeof(TO) == 4 && as2 >= 4 && (fx0 || fx1) && nthreads == 256 && al16) {
        const int dX = fx1 ? j.d1 : j.d0, dO = fx1 ? j.d0 : j.d1;
        const long long sX = fx1 ? j.s1 : j.s0, sO = fx1 ? j.s0 : j.s1, oX = fx1 ? j.o1 : j.o0, oO = fx1 ? j.o0 : j.o1;
        const int vX = fx1 ? (j.valid1 < j.d1 ? j.valid1 : j.d1) : dX, vO = fx1 ? dO : (j.valid1 < j.d1 ? j.valid1 : j.d1), vC = j.valid2 < j.d2 ? j.valid2 : j.d2;
        if (sX == 1 && !(dX & 3) && !(j.d2 & 3) && !(sO & 3) && !(j.s2 & 3) && !(oX & 3) && !(oO & 3) && !(vX & 3) && !(vC & 3) && dX >= 32 && j.d2 >= 32) {
            const int tiles_x = (dX + 63) >> 6, tiles_c = (j.d2 + 63) >> 6;
            const long long ntiles = (long long)dO * tiles_x * tiles_c;
            for (long long t = lb; t < ntiles; t += j.nblocks) {
                const int oi = (int)(t / (tiles_x * tiles_c)), rem = (int)(t - (long long)oi * (tiles_x * tiles_c));
                const int tx = rem / tiles_c, tc = rem - tx * tiles_c;
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = k * 256 + tid, cc = f >> 4, x0 = (f & 15) * 4, gx = tx * 64 + x0, gc = tc * 64 + cc;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    v[k] = (gx < vX && gc < vC && oi < vO) ? *(const f32x4*)((const float*)in + oi * sO + gx + (long long)gc * j.s2) : z;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = k * 256 + tid, cc = f >> 4, x0 = (f & 15) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) tile[cc * 65 + x0 + e] = v[k][e] * j.scale;
                }
                __syncthreads();
                f32x4 w[4]; float* op[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = k * 256 + tid, x = f >> 4, c0 = (f & 15) * 4, gx = tx * 64 + x, gc = tc * 64 + c0;
                    op[k] = (gx < dX && gc < j.d2) ? (float*)out + oi * oO + gx * oX + gc : nullptr;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    w[k] = (op[k] && j.accumulate) ? *(const f32x4*)op[k] : z;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = k * 256 + tid, x = f >> 4, c0 = (f & 15) * 4;
                    if (op[k]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[k][e] += tile[(c0 + e) * 65 + x];
                        *(f32x4*)op[k] = w[k];
                    }
                }
                __syncthreads();
            }
            return;
        }
    }
    if (as2 > 4 && (fx0 || fx1) && nthreads == 256 && total >= 4096 && (long long)(dXq < 64 ? dXq : 64) * (j.d2 < 64 ? j.d2 : 64) >= 192) {
        // ---- transpose path: X as above, O = the other outer dimension.  Tile = TX (along X) x TC (along c) = 4096 elements;
        // the shape follows the short side (3 conv taps along X on the way in, along c on the way out).
        const int dX = fx1 ? j.d1 : j.d0, dO = fx1 ? j.d0 : j.d1;
        const long long sX = fx1 ? j.s1 : j.s0, sO = fx1 ? j.s0 : j.s1, oX = fx1 ? j.o1 : j.o0, oO = fx1 ? j.o0 : j.o1;
        const int lx = dX <= 4 ? 2 : (dX <= 16 ? 4 : (j.d2 <= 4 ? 10 : (j.d2 <= 16 ? 8 : 6)));
        const int TX = 1 << lx, lc = 12 - lx, TC = 1 << lc, P = TX + 1;
        const int tiles_x = (dX + TX - 1) >> lx, tiles_c = (j.d2 + TC - 1) >> lc;
        const long long ntiles = (long long)dO * tiles_x * tiles_c;
        for (long long t = lb; t < ntiles; t += j.nblocks) {
            const int oi = (int)(t / (tiles_x * tiles_c)), rem = (int)(t - (long long)oi * (tiles_x * tiles_c));
            const int tx = rem / tiles_c, tc = rem - tx * tiles_c;
            {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int f = k * 256 + tid, x = f & (TX - 1), cc = f >> lx;                 // X fastest: coalesced input
                    const int gx = tx * TX + x, gc = tc * TC + cc, bb = fx1 ? gx : oi;
                    v[k] = 0.f;
                    if (gx < dX && gc < j.d2 && bb < j.valid1 && gc < j.valid2) v[k] = ldf(in + oi * sO + gx * sX + (long long)gc * j.s2) * j.scale;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) { const int f = k * 256 + tid; tile[(f >> lx) * P + (f & (TX - 1))] = v[k]; }
            }
            __syncthreads();
            if (!GROUPED_RMW || !j.accumulate) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int f = k * 256 + tid, x = f >> lc, cc = f & (TC - 1);                 // c fastest: coalesced output
                    const int gx = tx * TX + x, gc = tc * TC + cc;
                    if (gx < dX && gc < j.d2) {
                        TO* o = out + oi * oO + gx * oX + gc;
                        float w = tile[cc * P + x];
                        if (j.accumulate) w += ldf(o);
                        stf(o, w);
                    }
                }
            } else {
                // read-modify-write: 8 loads in flight, then 8 stores (a load / add / store chain per element costs 16 round trips)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float w[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int f = (h * 8 + k) * 256 + tid, x = f >> lc, cc = f & (TC - 1);
                        const int gx = tx * TX + x, gc = tc * TC + cc;
                        w[k] = tile[cc * P + x];
                        if (gx < dX && gc < j.d2) w[k] += ldf(out + oi * oO + gx * oX + gc);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int f = (h * 8 + k) * 256 + tid, x = f >> lc, cc = f & (TC - 1);
                        const int gx = tx * TX + x, gc = tc * TC + cc;
                        if (gx < dX && gc < j.d2) stf(out + oi * oO + gx * oX + gc, w[k]);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    if (j.s2 == 1 && !(j.d2 & 3) && !(j.s0 & 3) && !(j.s1 & 3) && !(j.o0 & 3) && !(j.o1 & 3) && !(j.valid2 & 3) &&
        ((uintptr_t)in % (4 * sizeof(TI))) == 0 && ((uintptr_t)out % (4 * sizeof(TO))) == 0) {
        // ---- row path
        const int c4n = j.d2 >> 2;
        for (long long i = (long long)lb * nthreads + tid; i < (total >> 2); i += (long long)j.nblocks * nthreads) {
            const int c = (int)(i % c4n) * 4; const long long t = i / c4n; const int b = (int)(t % j.d1); const int a = (int)(t / j.d1);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (b < j.valid1 && c < j.valid2) { Vec4<TI>::load(in + a * j.s0 + b * j.s1 + c, v); v[0] *= j.scale; v[1] *= j.scale; v[2] *= j.scale; v[3] *= j.scale; }
            TO* o = out + a * j.o0 + b * j.o1 + c;
            if (j.accumulate) { float w[4]; Vec4<TO>::load(o, w); v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3]; }
            if (sizeof(TO) == 4) { const f32x4 r = {v[0], v[1], v[2], v[3]}; *(f32x4*)o = r; }
            else { const u32x2 r = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}; *(u32x2*)o = r; }
        }
        return;
    }
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
    __shared__ float tile[1024 * 5];                    // TC x (TX + 1) for TX = 4 / 16 / 64
    const PermuteJob j = jobs[job_of_block[blockIdx.x]];
    const int lb = blockIdx.x - j.first_block;
    if (j.in_dtype == SS_F32 && j.out_dtype == SS_F32) permute_job<float, float>(j, lb, blockDim.x, threadIdx.x, tile);
    else if (j.in_dtype == SS_F32) permute_job<float, bf16_t>(j, lb, blockDim.x, threadIdx.x, tile);
    else if (j.out_dtype == SS_F32) permute_job<bf16_t, float>(j, lb, blockDim.x, threadIdx.x, tile);
    else permute_job<bf16_t, bf16_t>(j, lb, blockDim.x, threadIdx.x, tile);
}

// the gradient un-layout batch is all f32 -> f32 read-modify-write: its own kernel, so that the 8-deep load groups of that path
// do not push the mixed-dtype kernel above 128 registers
__global__ __launch_bounds__(256) void permute3d_batch_f32_kernel(const PermuteJob* __restrict__ jobs, const int* __restrict__ job_of_block)
{
    __shared__ float tile[1024 * 5];
    const PermuteJob j = jobs[job_of_block[blockIdx.x]];
    permute_job<float, float, true>(j, blockIdx.x - j.first_block, blockDim.x, threadIdx.x, tile);
}

extern "C" int ss_permute3d_batch(const void* jobs_dev, const int32_t* job_of_block_dev, int total_blocks, int all_f32, void* stream)
{
    SS_CHECK(total_blocks >= 0, "ss_permute3d_batch: negative block count");
    if (total_blocks == 0) return 0;
    SS_CHECK(jobs_dev && job_of_block_dev, "ss_permute3d_batch: null pointer");
    if (all_f32) SS_LAUNCH(permute3d_batch_f32_kernel, dim3(total_blocks), dim3(256), 0, stream, (const PermuteJob*)jobs_dev, (const int*)job_of_block_dev);
    else SS_LAUNCH(permute3d_batch_kernel, dim3(total_blocks), dim3(256), 0, stream, (const PermuteJob*)jobs_dev, (const int*)job_of_block_dev);
    SS_LAUNCH_CHECK("ss_permute3d_batch");
    return 0;
}
