// gemm8_common.h -- pieces shared by the 8-wave KC kernel (gemm8.hip) and the grouped weight-gradient kernel (gemm8_dw.hip).
#pragma once
#include "gemm_common.h"
#include "silent_speech_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace g8 {
constexpr int TBN = 256, RB = 128, BK8 = 64;
using ::sched_fence;
// ---- fragment reads whose completion the KERNEL counts, not the compiler.  hipcc either waits lgkmcnt(0) right after the
// reads of the next phase (fenced order) or re-serialises "one ds_read -> wait -> 4 MFMAs" (its own order): in both cases the
// LDS latency sits in front of the MFMAs.  The reads are therefore issued from inline asm (invisible to the compiler's wait
// insertion) and waited for by ONE s_waitcnt at the END of the phase in which they were issued, i.e. after that phase's
// MFMAs; the wait statement names every destination register "+v" so no consumer can be scheduled above it, and nothing is
// in flight at a loop back-edge (a register copy there would copy stale data).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ void lds_read128_async(bf16x8& d, const unsigned char* lds, unsigned addr) {
#if defined(SS_EMU)
    d = *(const bf16x8*)(lds + addr + OFF);
#else
    (void)lds;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
#endif
}
template <int N>
__device__ __forceinline__ void lds_wait_pin(bf16x8 (&f)[N]) {
#if !defined(SS_EMU)
    static_assert(N >= 1 && N <= 4, "fragment set size");
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]));
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
    else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
#endif
}

}  // namespace g8

static inline int g8_cus() { return ss_cu_count(3); }
static inline int g8_grant(const void* fn, size_t smem) {
    if (!ss_grant_lds(fn, smem)) { ss_set_error("gemm8: cannot reserve %zu bytes of LDS", smem); return 1; }
    return 0;
}

