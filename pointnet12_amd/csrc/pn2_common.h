// Shared helpers for the gfx950 kernels of libpn2_hip.so (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pn2.h"

#define PN2_WAVE 64

#define PN2_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return PN2_EINVAL; \
    } while (0)

static inline int pn2_launch_status() { return hipGetLastError() == hipSuccess ? PN2_OK : PN2_ELAUNCH; }

static inline hipStream_t pn2_s(pn2_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t pn2_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-bit max across a wave with xor-shuffles; every lane ends with the result.
__device__ __forceinline__ unsigned long long pn2_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double pn2_wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
