// common.h -- shared device/host helpers for the gfx950 kernels of the transduction hot path.
// One source, two builds: hipcc --offload-arch=gfx950 (the product) and, for the CPU test tier
// only, host clang with -DSS_EMU against tools/emu/hipemu.h (see that header).
#pragma once
#if defined(SS_EMU)
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#define SS_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// ------------------------------------------------------------------ error reporting (C ABI)
extern "C" const char* ss_last_error(void);
void ss_set_error(const char* fmt, ...);
#define SS_CHECK(cond, ...)                         \
    do {                                            \
        if (!(cond)) { ss_set_error(__VA_ARGS__); return 1; } \
    } while (0)
int ss_check_launch(const char* what);
int ss_upload_table(void* dst_dev, const void* src_host, size_t bytes, void* stream);   // runtime.hip: via a pinned staging ring
#define SS_LAUNCH_CHECK(what) do { int rc_ = ss_check_launch(what); if (rc_) return rc_; } while (0)

#if defined(SS_EMU)
#define SS_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipemu::launch(grid, block, smem, [&]() { kernel(__VA_ARGS__); })
#else
#define SS_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#endif
#define SS_KERNEL(...) (__VA_ARGS__)   // protects template commas inside SS_LAUNCH

// ------------------------------------------------------------------ element types
typedef unsigned short bf16_t;   // raw bfloat16 bits
enum { SS_F32 = 0, SS_BF16 = 1, SS_F64 = 2, SS_F32X3 = 3 };

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // round-to-nearest-even, NaN-preserving
#if defined(SS_EMU)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#else
    return __builtin_bit_cast(bf16_t, (__bf16)f);      // gfx950: v_cvt_pk_bf16_f32 (hardware RNE), no branches
#endif
}
template <class T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <class T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <class T> __device__ __forceinline__ float rnd(float v);   // round through the storage type
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }

// 8-element vector load/store as floats (16 B for bf16, 32 B for f32); p must be 16-B aligned
template <class T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        u32x4 r = *(const u32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
        *(u32x4*)p = r;
    }
};

// raw (unconverted) 8-element chunks: load now, convert at the use -> the load stays in flight across unrelated work
template <class T> struct RawVec8;
template <> struct RawVec8<bf16_t> {
    typedef u32x4 type;
    static __device__ __forceinline__ type load(const bf16_t* p) { return *(const u32x4*)p; }
    static __device__ __forceinline__ type zero() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }
    static __device__ __forceinline__ void unpack(const type& r, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
    }
};
struct f32x8_raw { f32x4 a, b; };
template <> struct RawVec8<float> {
    typedef f32x8_raw type;
    static __device__ __forceinline__ type load(const float* p) { type r; r.a = *(const f32x4*)p; r.b = *(const f32x4*)(p + 4); return r; }
    static __device__ __forceinline__ type zero() { type r; f32x4 z = {0.f, 0.f, 0.f, 0.f}; r.a = z; r.b = z; return r; }
    static __device__ __forceinline__ void unpack(const type& r, float (&v)[8]) { v[0] = r.a[0]; v[1] = r.a[1]; v[2] = r.a[2]; v[3] = r.a[3]; v[4] = r.b[0]; v[5] = r.b[1]; v[6] = r.b[2]; v[7] = r.b[3]; }
};

// ------------------------------------------------------------------ wave helpers (wave = 64)
// Sum over the 64 lanes, returned in every lane.  DPP adds (quad swaps, row half-mirror / mirror, row broadcasts) + one readlane: six
// full-rate VALU ops instead of six ds_bpermute round trips through the LDS crossbar (their latency sat on the critical path of every
// LayerNorm row).
__device__ __forceinline__ float wave_sum(float v) {
#if defined(SS_EMU)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
#else
#define SS_DPP_ADD(CTRL, ROWMASK) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, true))
    SS_DPP_ADD(0xB1, 0xf);        // quad_perm [1,0,3,2]
    SS_DPP_ADD(0x4E, 0xf);        // quad_perm [2,3,0,1]
    SS_DPP_ADD(0x141, 0xf);       // row_half_mirror: lanes of 8
    SS_DPP_ADD(0x140, 0xf);       // row_mirror: every lane of a 16-lane row holds the row sum
    SS_DPP_ADD(0x142, 0xa);       // row_bcast:15 into rows 1 and 3
    SS_DPP_ADD(0x143, 0xc);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef SS_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}
// lane l receives lane l - 1's value, lane 0 receives `fill` (DPP wave_shr:1: one full-rate VALU operation; __shfl_up would be a ds_bpermute round trip)
__device__ __forceinline__ float wave_shr1(float v, float fill) {
#if defined(SS_EMU)
    const float r = __shfl_up(v, 1);
    return (threadIdx.x & 63) == 0 ? fill : r;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// LDS hand-off inside ONE wave (each wave owns its tiles): LDS operations of a wave are served in order, so only
// the compiler must be kept from reordering; no workgroup barrier -> the 4 waves of a block run independently.
__device__ __forceinline__ void wave_lds_sync() {
#if defined(SS_EMU)
    hipemu::sync_wave();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// 2^x and 1/x straight on the transcendental unit (v_exp_f32 / v_rcp_f32, 1 ulp; no denormal fix-up code)
__device__ __forceinline__ void compiler_fence();
// Words in LDS that one wave writes and another wave of the workgroup polls (progress words, mailboxes).  A `volatile` access through a pointer derived
// from the dynamic LDS block stays a FLAT access with system scope (the address-space inference pass leaves volatile operations alone):
// flat_load/store ... sc0 sc1 followed by s_waitcnt vmcnt(0), which on gfx9 also waits for every global store the wave has in flight.  These go through
// an address-space-3 pointer as relaxed workgroup-scope atomics instead: plain ds_read_b32 / ds_write_b32 the compiler neither caches in a register nor
// reorders against each other (LDS serves one wave's operations in order, so "data, then progress word" needs no fence on the hardware side).
#if defined(SS_EMU)
__device__ __forceinline__ float lds_peek_f32(const float* p) { return *(const volatile float*)p; }
__device__ __forceinline__ int lds_peek_i32(const int* p) { return *(const volatile int*)p; }
__device__ __forceinline__ void lds_post_f32(float* p, float v) { *(volatile float*)p = v; }
__device__ __forceinline__ void lds_post_i32(int* p, int v) { *(volatile int*)p = v; }
#else
__device__ __forceinline__ float lds_peek_f32(const float* p) { return __hip_atomic_load((__attribute__((address_space(3))) float*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_peek_i32(const int* p) { return __hip_atomic_load((__attribute__((address_space(3))) int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_post_f32(float* p, float v) { __hip_atomic_store((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_post_i32(int* p, int v) { __hip_atomic_store((__attribute__((address_space(3))) int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
__device__ __forceinline__ float fast_exp2(float x) {
#if defined(SS_EMU)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float fast_rcp(float x) {
#if defined(SS_EMU)
    return 1.f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// two floats -> packed bf16 pair (lo in bits 0..15)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
#if defined(SS_EMU)
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
#else
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw));      // ONE v_cvt_pk_bf16_f32 (two separate conversions + shift/or otherwise)
#endif
}

// ------------------------------------------------------------------ MFMA wrappers
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(SS_EMU)
    return hipemu::mfma_16x16x32_bf16(a, b, c);
#else
    typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
#if defined(SS_EMU)
    return hipemu::mfma_16x16x4_f32(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// direct global -> LDS copy (global_load_lds_dwordx4): lane l writes its 16 bytes at lds_wave_base + 16*l; the base must be
// wave-uniform (it travels in M0).  Completion is tracked by vmcnt (a following __syncthreads() drains it).
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
#if defined(SS_EMU)
    hipemu::global_load_lds16(gptr, lds_wave_base);
#else
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gptr, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
#endif
}
// barrier that does NOT drain outstanding global->LDS copies (only LDS accesses of this wave are waited for)
__device__ __forceinline__ void barrier_keep_vm() {
#if defined(SS_EMU)
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

// hand-counted wait on this wave's vector-memory operations (they retire in issue order): at most N still outstanding afterwards
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if !defined(SS_EMU)
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
#endif
}

// ---- host side of the same seam
// compute units of the current device (the emulator pretends `emu_cus`, so that persistent kernels walk several items per workgroup in tests)
static inline int ss_cu_count(int emu_cus) {
#if defined(SS_EMU)
    return emu_cus;
#else
    static int cus = 0;
    if (!cus) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
    return cus;
#endif
}
// a kernel that wants more than 64 KiB of dynamic LDS has to be granted it once; false = refused
static inline bool ss_grant_lds(const void* kernel, size_t bytes) {
#if defined(SS_EMU)
    (void)kernel; (void)bytes;
    return true;
#else
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
#endif
}
static inline bool ss_memset_async(void* p, int value, size_t bytes, void* stream) {
#if defined(SS_EMU)
    (void)stream; memset(p, value, bytes);
    return true;
#else
    return hipMemsetAsync(p, value, bytes, (hipStream_t)stream) == hipSuccess;
#endif
}
static inline bool ss_copy_d2d_async(void* dst, const void* src, size_t bytes, void* stream) {
#if defined(SS_EMU)
    (void)stream; memcpy(dst, src, bytes);
    return true;
#else
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess;
#endif
}

// ---- the backend seam of the hand-scheduled kernels: every sequence that exists twice (gfx950 / the host emulator of tests/) lives in ONE
// named helper here; the kernels call the helper and carry no #if of their own for it.
// a wave-uniform value as a scalar (v_readfirstlane): drives uniform branches, M0 / LDS base addresses
__device__ __forceinline__ int wave_uniform(int v) {
#if defined(SS_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// the value of the first lane in every lane (v_readfirstlane again; the emulator has to exchange it: every live lane of the wave calls this)
__device__ __forceinline__ int wave_first(int v) {
#if defined(SS_EMU)
    return __shfl(v, 0);
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
// "this value is opaque here": the compiler can neither hoist what is derived from it out of the enclosing loop nor move a consumer above
// this point (used after hand-counted waits to pin the registers an asm load has written)
template <class T> __device__ __forceinline__ void pin_vgpr(T& x) {
#if !defined(SS_EMU)
    asm volatile("" : "+v"(x));
#else
    (void)x;
#endif
}
template <class T> __device__ __forceinline__ void pin_sgpr(T& x) {
#if !defined(SS_EMU)
    asm volatile("" : "+s"(x));
#else
    (void)x;
#endif
}
// compiler-level memory fence (no instruction)
__device__ __forceinline__ void compiler_fence() {
#if !defined(SS_EMU)
    asm volatile("" ::: "memory");
#endif
}
// byte address of an LDS pointer inside the workgroup's segment (what ds_* / M0 take); the emulator addresses LDS through the pointer itself
__device__ __forceinline__ unsigned lds_byte_address(const void* lds_ptr) {
#if defined(SS_EMU)
    (void)lds_ptr;
    return 0u;
#else
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds_ptr);
#endif
}
// nothing crosses this point in the instruction schedule
__device__ __forceinline__ void sched_fence() {
#if !defined(SS_EMU)
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// LDS transpose read (ds_read_b64_tr_b16): see tools/emu/hipemu.h for the lane map (verified on gfx950)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_ptr) {
#if defined(SS_EMU)
    return hipemu::ds_read_tr16_b64((const short*)lds_ptr);
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds_ptr);
#endif
}

// ------------------------------------------------------------------ dropout RNG
// Four keep-decisions for dropout group g4 (any bijective group numbering shared by forward and backward).
// Dropout only needs decorrelated, reproducible Bernoulli draws, not Philox-grade streams: two rounds of a 32-bit
// multiply-xorshift finaliser per 2 draws (2 v_mul_lo per hash) cost ~1/7 of Philox4x32-10 (40 quarter-rate multiplies),
// which was the largest item of the FFN epilogue.  16-bit uniforms -> the keep probability is exact to 2^-16.
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void dropout_keep4(unsigned long long seed, unsigned stream, unsigned long long g4, unsigned thresh, bool (&keep)[4]) {
    const unsigned lo = (unsigned)g4, hi = (unsigned)(g4 >> 32);
    const unsigned key = (unsigned)seed ^ ((unsigned)(seed >> 32) * 0x9E3779B9u) ^ (stream * 0x85EBCA6Bu) ^ (hi * 0xC2B2AE35u);
    const unsigned a = mix32(lo * 2u + key), b = mix32(lo * 2u + 1u + (key ^ 0x68E31DA4u));
    const unsigned t16 = thresh >> 16;
    keep[0] = (a & 0xffffu) >= t16; keep[1] = (a >> 16) >= t16; keep[2] = (b & 0xffffu) >= t16; keep[3] = (b >> 16) >= t16;
}
// One element of the same stream: element e <-> group e >> 2, slot e & 3 (only the word that holds the slot is mixed).  The GEMM epilogues draw by the ROW-MAJOR
// element index e = row * N + col (round 6; before: group (row >> 2) * N + col, slot row & 3): a lane of the transposed accumulator layout holds four consecutive
// columns of one row, i.e. exactly one group -- the dropout epilogue of the 8-wave kernel can leave through the register epilogue (direct_store8) instead of the LDS
// C piece.  Kernels whose lanes hold four ROWS of one column draw element by element.
__device__ __forceinline__ bool dropout_keep1(unsigned long long seed, unsigned stream, unsigned long long e, unsigned thresh) {
    const unsigned long long g4 = e >> 2;
    const unsigned lo = (unsigned)g4, hi = (unsigned)(g4 >> 32), slot = (unsigned)e & 3u;
    const unsigned key = (unsigned)seed ^ ((unsigned)(seed >> 32) * 0x9E3779B9u) ^ (stream * 0x85EBCA6Bu) ^ (hi * 0xC2B2AE35u);
    const unsigned w = (slot & 2u) ? mix32(lo * 2u + 1u + (key ^ 0x68E31DA4u)) : mix32(lo * 2u + key);
    return ((slot & 1u) ? (w >> 16) : (w & 0xffffu)) >= (thresh >> 16);
}
static inline unsigned dropout_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0) return 0u;
    if (t >= 4294967295.0) return 4294967295u;
    return (unsigned)t;
}

// ------------------------------------------------------------------ row addressing
// Logical row i of a (possibly batched / strided / overlapping) matrix starts at element offset
//   base + (i / rows_per_batch) * batch_stride + (i % rows_per_batch) * row_stride.
// This one map expresses plain matrices, the zero-padded (B, T+2, C) activation buffers, the
// overlapping 3C-wide im2col rows of the k=3 convolutions (architecture.py:18,20) with stride 1/2,
// and the stride-2 scatter of their input gradients.
struct RowMap {
    long long base;
    long long batch_stride;
    long long row_stride;
    int rows_per_batch;
};
__device__ __host__ __forceinline__ long long rowmap_off(const RowMap& m, int i) {
    if (m.rows_per_batch == 0x7fffffff) return m.base + (long long)i * m.row_stride;      // plain matrix: no division
    int b = i / m.rows_per_batch, t = i - b * m.rows_per_batch;
    return m.base + (long long)b * m.batch_stride + (long long)t * m.row_stride;
}
static inline RowMap rowmap_plain(long long ld) { RowMap m; m.base = 0; m.batch_stride = 0; m.row_stride = ld; m.rows_per_batch = 0x7fffffff; return m; }
