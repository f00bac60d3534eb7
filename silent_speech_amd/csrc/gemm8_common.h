// gemm8_common.h -- pieces shared by the 8-wave KC kernel (gemm8.hip) and the grouped weight-gradient kernel (gemm8_dw.hip).
#pragma once
#include "gemm_common.h"
#include "silent_speech_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace g8 {
constexpr int TBN = 256, RB = 128, BK8 = 64;
__device__ __forceinline__ void sched_fence() {
#if !defined(SS_EMU)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
}  // namespace g8

static inline int g8_cus() {
#if defined(SS_EMU)
    return 3;
#else
    static int cus = 0;
    if (!cus) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
    return cus;
#endif
}
static inline int g8_grant(const void* fn, size_t smem) {
#if !defined(SS_EMU)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { ss_set_error("gemm8: cannot reserve %zu bytes of LDS", smem); return 1; }
#endif
    return 0;
}

