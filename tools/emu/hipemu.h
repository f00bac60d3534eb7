// hipemu.h -- a minimal host-side SIMT emulator for the kernels in silent_speech_amd/csrc.
//
// PURPOSE: test infrastructure only.  There is no GPU in the build container, so the CPU test tier
// compiles the SAME kernel sources with the host clang (-DSS_EMU) against this header and runs them
// with one fiber per GPU thread.  It validates index arithmetic, LDS staging, barrier structure,
// wave-level exchanges and the MFMA fragment maps before a kernel ever reaches the MI355X.  It is
// NOT a product path: the package loads only libsilent_speech_hip.so (gfx950) and raises when that
// is missing; the emulator library is injected explicitly by tests (tests/emu_backend.py).
//
// Model: blocks run sequentially; the threads of a block are ucontext fibers scheduled round-robin;
// __syncthreads() and every wave-level primitive (__shfl*, __ballot, MFMA) are rendezvous points
// (block-wide / wave-wide).  A wave is 64 consecutive threads (gfx950).  MFMA is emulated from the
// documented gfx950 fragment maps (16x16x32 bf16, 16x16x4 f32): see mfma_* below.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t*) { return 0; }

namespace hipemu {

constexpr int kWave = 64;
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    int wait = 0;        // 0 runnable, 1 wave rendezvous, 2 block rendezvous
    unsigned gen = 0;    // generation waited on
    dim3 tid;
};

struct State {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    int cur = -1;
    int nthreads = 0, live = 0;
    int wave_live[kMaxThreads / kWave];
    int wave_arr[kMaxThreads / kWave];
    unsigned wave_gen[kMaxThreads / kWave];
    int blk_arr = 0;
    unsigned blk_gen = 0;
    std::function<void()> body;
    // exchange areas
    uint64_t xch[kMaxThreads];
    float big[kMaxThreads][40];
    char* dyn_smem = nullptr;
};

inline State& S() { static State s; return s; }

}  // namespace hipemu

// CUDA/HIP built-in coordinates (set by the scheduler before a fiber is resumed)
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {

inline int flat_tid() { return S().cur; }

inline void yield_to_sched() {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    swapcontext(&f.ctx, &s.sched);
}

inline void sync_block() {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    f.wait = 2; f.gen = s.blk_gen; s.blk_arr++;
    yield_to_sched();
}

inline void sync_wave() {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    int w = s.cur / kWave;
    f.wait = 1; f.gen = s.wave_gen[w]; s.wave_arr[w]++;
    yield_to_sched();
}

inline void fiber_entry() {
    State& s = S();
    s.body();
    s.fibers[s.cur].done = true;
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void run_block(dim3 block) {
    State& s = S();
    int n = block.x * block.y * block.z;
    if (n > kMaxThreads) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if ((int)s.fibers.size() < n) s.fibers.resize(n);
    s.nthreads = s.live = n;
    int nw = (n + kWave - 1) / kWave;
    for (int w = 0; w < nw; ++w) { s.wave_live[w] = (w + 1) * kWave <= n ? kWave : n - w * kWave; s.wave_arr[w] = 0; s.wave_gen[w] = 0; }
    s.blk_arr = 0; s.blk_gen = 0;
    for (int i = 0; i < n; ++i) {
        Fiber& f = s.fibers[i];
        if (!f.stack) f.stack = (char*)malloc(kStack);
        f.done = false; f.wait = 0;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &s.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    while (s.live > 0) {
        bool progress = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = s.fibers[i];
            if (f.done) continue;
            int w = i / kWave;
            if (f.wait == 1) {
                if (f.gen == s.wave_gen[w]) {
                    if (s.wave_arr[w] >= s.wave_live[w]) { s.wave_gen[w]++; s.wave_arr[w] = 0; } else continue;
                }
            } else if (f.wait == 2) {
                if (f.gen == s.blk_gen) {
                    if (s.blk_arr >= s.live) { s.blk_gen++; s.blk_arr = 0; } else continue;
                }
            }
            f.wait = 0;
            s.cur = i;
            threadIdx = f.tid;
            swapcontext(&s.sched, &f.ctx);
            progress = true;
            if (f.done) { s.live--; s.wave_live[w]--; }
        }
        if (!progress) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier / wave rendezvous) in block (%u,%u,%u)\n", blockIdx.x, blockIdx.y, blockIdx.z);
            abort();
        }
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t smem, F&& body) {
    State& s = S();
    s.body = body;
    std::vector<char> dyn(smem + 64, (char)0xFF);      // LDS is not zeroed on hardware: 0xFF.. = NaN in bf16 and f32, so stale reads show up in the CPU tier
    s.dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    gridDim = grid; blockDim = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                blockIdx = dim3(x, y, z);
                run_block(block);
            }
    s.dyn_smem = nullptr;
}

// ---- wave-level exchange -------------------------------------------------------------
template <class T>
inline T xchg_read(T v, int src_lane_abs_or_neg) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    State& s = S();
    int me = s.cur;
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
    s.xch[me] = raw;
    sync_wave();
    T r = v;
    if (src_lane_abs_or_neg >= 0 && src_lane_abs_or_neg < s.nthreads) { uint64_t q = s.xch[src_lane_abs_or_neg]; memcpy(&r, &q, sizeof(T)); }
    sync_wave();
    return r;
}

}  // namespace hipemu

inline void __syncthreads() { hipemu::sync_block(); }

template <class T> inline T __shfl(T v, int src, int width = 64) {
    int me = hipemu::flat_tid(); int lane = me & 63; int seg = lane / width * width;
    return hipemu::xchg_read(v, (me - lane) + seg + ((src % width) + width) % width);
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int me = hipemu::flat_tid(); int lane = me & 63; int seg = lane / width * width;
    int t = (lane - seg) ^ mask;
    return hipemu::xchg_read(v, t < width ? (me - lane) + seg + t : me);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int me = hipemu::flat_tid(); int lane = me & 63; int seg = lane / width * width;
    int t = (lane - seg) - (int)d;
    return hipemu::xchg_read(v, t >= 0 ? (me - lane) + seg + t : me);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int me = hipemu::flat_tid(); int lane = me & 63; int seg = lane / width * width;
    int t = (lane - seg) + (int)d;
    return hipemu::xchg_read(v, t < width ? (me - lane) + seg + t : me);
}
inline unsigned long long __ballot(int pred) {
    hipemu::State& s = hipemu::S();
    int me = s.cur, lane = me & 63, base = me - lane;
    s.xch[me] = pred ? 1 : 0;
    hipemu::sync_wave();
    unsigned long long m = 0;
    for (int l = 0; l < 64 && base + l < s.nthreads; ++l)
        if (!s.fibers[base + l].done && s.xch[base + l]) m |= 1ull << l;
    hipemu::sync_wave();
    return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) {
    hipemu::State& s = hipemu::S();
    int me = s.cur, lane = me & 63, base = me - lane;
    unsigned long long live = 0;
    for (int l = 0; l < 64 && base + l < s.nthreads; ++l) if (!s.fibers[base + l].done) live |= 1ull << l;
    return (__ballot(p) & live) == live;
}

// ---- atomics (blocks and fibers are sequential on the host) ---------------------------
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }

// ---- math / bit helpers ---------------------------------------------------------------
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

// ---- MFMA emulation (gfx950 fragment maps; cdna_hip_programming.md section 3) ----------
// v_mfma_f32_16x16x32_bf16: lane l holds A[i=l&15][k=(l>>4)*8+e], B[k=(l>>4)*8+e][j=l&15], e<8;
//                           C/D: col j=l&15, row i=(l>>4)*4+reg.
// v_mfma_f32_16x16x4_f32  : lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; C/D as above.
namespace hipemu {
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
inline float bf16_bits_to_f(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

inline v4f mfma_16x16x32_bf16(v8s a, v8s b, v4f c) {
    State& s = S();
    int me = s.cur, lane = me & 63, base = me - lane;
    for (int e = 0; e < 8; ++e) { s.big[me][e] = bf16_bits_to_f((unsigned short)a[e]); s.big[me][8 + e] = bf16_bits_to_f((unsigned short)b[e]); }
    sync_wave();
    v4f d = c;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(s.big[base + i + 16 * q][e], s.big[base + j + 16 * q][8 + e], acc);
        d[r] = acc;
    }
    sync_wave();
    return d;
}

inline v4f mfma_16x16x4_f32(float a, float b, v4f c) {
    State& s = S();
    int me = s.cur, lane = me & 63, base = me - lane;
    s.big[me][0] = a; s.big[me][1] = b;
    sync_wave();
    v4f d = c;
    int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int i = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int q = 0; q < 4; ++q) acc = fmaf(s.big[base + i + 16 * q][0], s.big[base + j + 16 * q][1], acc);
        d[r] = acc;
    }
    sync_wave();
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=(l>>5)*8+e], B[k=(l>>5)*8+e][j=l&31], e<8;
//                           C/D: col j=l&31, row i=(reg&3)+8*(reg>>2)+4*(l>>5), reg<16  (cdna_hip_programming.md section 3).
typedef float v16f __attribute__((ext_vector_type(16)));
inline v16f mfma_32x32x16_bf16(v8s a, v8s b, v16f c) {
    State& s = S();
    int me = s.cur, lane = me & 63, base = me - lane;
    for (int e = 0; e < 8; ++e) { s.big[me][e] = bf16_bits_to_f((unsigned short)a[e]); s.big[me][8 + e] = bf16_bits_to_f((unsigned short)b[e]); }
    sync_wave();
    v16f d = c;
    int j = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int q = 0; q < 2; ++q)
            for (int e = 0; e < 8; ++e)
                acc = fmaf(s.big[base + i + 32 * q][e], s.big[base + j + 32 * q][8 + e], acc);
        d[r] = acc;
    }
    sync_wave();
    return d;
}
// global_load_lds_dwordx4: every lane copies 16 bytes from its own global address to  lds_base (wave-uniform) + lane*16
inline void global_load_lds16(const void* gptr, void* lds_wave_base) {
    State& s = S();
    memcpy((char*)lds_wave_base + (s.cur & 63) * 16, gptr, 16);
}

// ds_read_b64_tr_b16 (gfx950), verified on hardware (tools/probes/tr_probe.hip): inside each 16-lane group, lane i
// supplies the address of 4 contiguous 16-bit elements; lane c receives, for j = 0..3, element (c % 4) of the
// 4 elements addressed by lane (4*j + c/4)  -- i.e. column c of the 4 x 16 block whose row j is the 16 elements
// addressed by lanes 4j..4j+3.
typedef short v4s_t __attribute__((ext_vector_type(4)));
inline v4s_t ds_read_tr16_b64(const short* p) {
    State& s = S();
    int me = s.cur, lane = me & 63, base = me - lane, grp = lane & ~15, c = lane & 15;
    uint64_t raw; memcpy(&raw, p, 8);
    s.xch[me] = raw;
    sync_wave();
    v4s_t r;
    for (int j = 0; j < 4; ++j) {
        uint64_t q = s.xch[base + grp + 4 * j + (c >> 2)];
        short e[4]; memcpy(e, &q, 8);
        r[j] = e[c & 3];
    }
    sync_wave();
    return r;
}
}  // namespace hipemu

#define SS_DYN_SMEM(name) char* name = hipemu::S().dyn_smem
