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

// 64-bit max over each row of 16 lanes with DPP (no LDS round trip): xor-1 and xor-2 inside the quads
// (quad_perm), then row_half_mirror and row_mirror fold the quads; every lane of the row ends with the row max.
__device__ __forceinline__ unsigned long long pn2_row_max_u64(unsigned long long v) {
#define PN2_DPP_MAX_STEP(ctrl)                                                                  \
    {                                                                                           \
        const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);                         \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xF, 0xF, false); \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xF, 0xF, false); \
        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;                     \
        v = o > v ? o : v;                                                                      \
    }
    PN2_DPP_MAX_STEP(0xB1)    // quad_perm [1,0,3,2]
    PN2_DPP_MAX_STEP(0x4E)    // quad_perm [2,3,0,1]
    PN2_DPP_MAX_STEP(0x141)   // row_half_mirror
    PN2_DPP_MAX_STEP(0x140)   // row_mirror
#undef PN2_DPP_MAX_STEP
    return v;
}

// 64-bit max over the wave, result uniform (SGPRs): DPP inside the four rows, then four readlanes.
__device__ __forceinline__ unsigned long long pn2_wave_max_u64_dpp(unsigned long long v) {
    v = pn2_row_max_u64(v);
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    unsigned long long best = 0;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned long long r = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, row * 16) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane(lo, row * 16);
        best = r > best ? r : best;
    }
    return best;
}

// Buffer clears are plain kernels, not hipMemsetAsync: a memset NODE captured into a hipGraph on memory that was
// allocated during the capture faults at replay on this ROCm ("write access to a read-only page").
namespace {
__global__ __launch_bounds__(256) void pn2_fill_u32_kernel(unsigned *__restrict__ p, unsigned value, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = value;
}
}  // namespace

static inline void pn2_fill_u32(void *p, unsigned value, int64_t n_words, hipStream_t s) {
    if (n_words <= 0) return;
    hipLaunchKernelGGL(pn2_fill_u32_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<unsigned *>(p), value, n_words);
}

__device__ __forceinline__ double pn2_wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
