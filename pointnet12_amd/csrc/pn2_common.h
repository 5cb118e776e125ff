// Shared helpers for the gfx950 kernels of libpn2_hip.so (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <set>
#include <utility>
#include "../../include/pn2.h"

#define PN2_WAVE 64
#define PN2_CF_REPL 8            // replicas of the first-layer closed-form weight gradient's scratch (mlp.hip; added to by mlp_res.hip's fused form)

// Stores of activations that only LATER kernels read.  Round 1 measured nontemporal stores 2-6 % faster on the streamed forward
// GEMMs; round 3 measured the opposite on the store-bound gather + conv kernel (92 -> 80 us at 1 M rows).  One switch for A/B
// builds of the whole library: make XFLAGS=-DPN2_PLAIN_STORES.
#ifdef PN2_PLAIN_STORES
#define PN2_STREAM_STORE(value, ptr) (*(ptr) = (value))
#else
#define PN2_STREAM_STORE(value, ptr) __builtin_nontemporal_store((value), (ptr))
#endif

#define PN2_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return PN2_EINVAL; \
    } while (0)

static inline int pn2_launch_status() { return hipGetLastError() == hipSuccess ? PN2_OK : PN2_ELAUNCH; }

static inline hipStream_t pn2_s(pn2_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t pn2_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ----------------------------------------------------------------------------- library options
// Dispatch / tuning switches of the library.  They are set EXPLICITLY through pn2_set_option() (include/pn2.h): the library
// never reads the process environment, so an external caller's results depend on its arguments and on the options it set,
// nothing else (the Python binding forwards PN2_* environment variables for A/B runs -- _lib.load()).
// Every default is the measured winner.  Read per call (a relaxed load): an option changed between two calls takes effect.
#define PN2_OPTION_LIST(X) \
    X(FPS_ROWS_MIN_PPT, 24) /* FPS rows kernel: points per thread from which it replaces the wave-level pruned kernel */ \
    X(FPS_ROWMAP, 2) /* FPS rows kernel: row ownership (0 contiguous run per wave, 1 round robin, 2 groups of four rows round robin) */ \
    X(FPS_SINGLE_MAX, 24576) /* largest cloud of the single-workgroup FPS kernel (clamped to 16384 .. 28672) */ \
    X(FPS_PIECE, 0) /* 2048 < N <= 4096: the sampling chain as launches of this many iterations each (0: one launch); state through the workspace */ \
    X(FPS_COOP, 1) /* cooperative multi-workgroup FPS above FPS_SINGLE_MAX */ \
    X(FPS_PRUNE, 1) /* spatially pruned FPS kernels (8192 < N) */ \
    X(BQ_ORDER, 1) /* ball query: centres taken in Morton order (pn2_ball_query_ws) */ \
    X(FEWROW_MAX_TILES, 1024) /* fewrow_nt_kernel (forward): largest 32 x 64 tile count it takes (0: off) */ \
    X(FEWROW_MAX_TILES_DGRAD, 512) /* the same for the data gradient */ \
    X(FEWROW_KS, 0) /* fewrow_nt_kernel: waves per tile (0: automatic) */ \
    X(NT_CFG, 0) /* NT GEMM tile override (tuning) */ \
    X(NT_SMALL_DEPTH, 1) /* few-row NT GEMM: k-steps in flight / accumulator chains / rotated k order (1 .. 4) */ \
    X(NT_NSPLIT, 1) /* 129 .. 224 output columns as 128 + remainder */ \
    X(TN_SPLITDIV, 1) /* weight-gradient split depth divisor */ \
    X(TN_CFG, 0) /* TN GEMM tile override (tuning) */ \
    X(TN_NARROW, 1) /* 32 x 32 / 64 x 32 wave-split tiles for narrow weight gradients */ \
    X(TN_SMALLP, 65536) /* 64 x 64 tiles up to this many rows */ \
    X(TN_SMALL_DEPTH, 2) /* few-row TN GEMM: stages in flight */ \
    X(CF_WGS_PER_CU, 2) /* closed-form first-layer weight gradient: workgroups per CU */ \
    X(WGRAD_SKINNY, 1) /* streaming first-layer weight-gradient kernel */ \
    X(BWD_PAIR, 1) /* pn2_conv1x1_bwd_pair: dgrad + wgrad bodies in one launch */ \
    X(RES, 1) /* weight-resident kernels (csrc/mlp_res.hip) */ \
    X(RES_MIN_ROWS, 32768) /* rows from which they take a layer */ \
    X(RES_HALF, 1) /* two 4-wave workgroups per CU on the fused backward pairs that fit twice */ \
    X(WIDE, 1) /* register-stationary kernels (csrc/mlp_wide.hip) */ \
    X(WIDE_MIN_ROWS, 65536) /* rows from which they take a layer */ \
    X(SPLIT, 1) /* wide layers: fp32 products as six bf16 MFMA products of exact three-way splits (split_nt_kernel) */ \
    X(SPLIT_WGRAD, 1) /* ... the full-tile weight gradients too (split_tn_kernel) */ \
    X(SPLIT_K256, 1) /* ... the pooled data gradients with C_out = 256 (contraction split over wave pairs) */ \
    X(SPLIT_NARROW, 1) /* ... also on the narrow sa1 forward layers the weight-resident kernels served */ \
    X(SPLIT_RES, 1) /* ... and the fused data + weight gradient of the narrow long layers (split_bwd_res_kernel) */ \
    X(FUSE_FIRST, 1) /* ... with the FIRST layer's dZ^T x formed in the second layer's fused backward (sa1: dZ1 never reaches memory; pn2_conv1x1_bwd_first).  Exact, -0.9 GB of traffic per MSG step (14.7 -> 13.9 GB).  Alone the fused kernels are slower than what they replace (serial step: 96 x 64 243 -> 282 us, 64 x 64 113 -> 140 against the 2 x 53 us of weight-gradient launches they shorten), and with the geometry branch forked at the top of the step the step was too (4.96 -> 4.98 ms: off until the end of round 6); with the branch behind sa2 the captured MSG step measures 4.706 - 4.767 against 4.764 - 4.829 ms and 4.739 - 4.772 against 4.775 - 4.945 (eight and six alternating runs on two boxes), cfg5 MSG 40.3 - 40.9 against 40.1 - 40.8, SSG equal: on */ \
    X(SPLIT_WG2, 1) /* ... its four-wave forms (sa1 of MSG) as two workgroups per CU (alone: 96 -> 128 pooled 244 -> 209 us; cfg5 MSG 41.4 -> 41.0 ms) */ \
    X(POOL_CF, 2) /* the pooled last layers of sa1 WITHOUT their pre-BN output: forward stores nothing, backward from the layer's input (split_bwd_cf_kernel); 1: 128 x 96 only, 2: also 128 x 64 */ \
    X(SPLIT_RES_MIN_TILES_128, 1024) /* ... its 128 x 128 pair from this many 64-row tiles on (below: the streamed pair kernel) */ \
    X(SPLIT_MIN_ROWS_128, 98304) /* ... 128 -> 128 forward / data gradient from this many rows on (below: the streamed fp32 kernels) */ \
    X(RING, 0) /* forward: the LDS-DMA ring form of the register-stationary kernel (measured equal: DESIGN.md section 3) */ \
    X(WIDE_POOL, 1) /* pooling extrema in the register-stationary forward's epilogue */ \
    X(WIDE_ADB196, 1) /* 196 -> 128 data gradient: operand reads one k block ahead */ \
    X(WIDE_WGRAD, 1) /* full-tile weight gradient */ \
    X(WIDE_WGRAD_MIN_ROWS, 131072) /* rows from which it takes a layer */ \
    X(WGRAD_TWO_PHASE, 0) /* two-phase dW flush through caller scratch (256 x 196) */ \
    X(WGRAD_TWO_PHASE_ALL, 0) /* ... for all three full-tile shapes */ \
    X(SEG_CHUNK, 0) /* segmented scatter: members per lane group (0: automatic) */

enum Pn2Option {
#define PN2_X(name, dflt) PN2_OPT_##name,
    PN2_OPTION_LIST(PN2_X)
#undef PN2_X
    PN2_OPT_COUNT
};
extern int pn2_option_table[PN2_OPT_COUNT];                          // api.hip
static inline int pn2_opt(int id) { return __atomic_load_n(&pn2_option_table[id], __ATOMIC_RELAXED); }

// ----------------------------------------------------------------------------- diagnostics
// pn2_last_kernel() (include/pn2.h): the kernel (template instantiation) the calling thread's most recent GEMM launcher enqueued, as
// the profiler prints it.  The launchers store the kernel's host-side function pointer (one thread-local store); nothing inside the
// library reads it: bench.py prices each launch of its instrumented pass against the peak of the pipe that kernel runs on.
extern thread_local const void *pn2_last_kernel_fn;                  // api.hip
#define PN2_NOTE_KERNEL(...) (pn2_last_kernel_fn = reinterpret_cast<const void *>(__VA_ARGS__))

// Current device, clamped to the per-device tables below (a process driving more than PN2_MAX_DEVICES GPUs shares the
// last slot, which only costs it a redundant attribute call or a CU count of the wrong device for grid sizing).
#define PN2_MAX_DEVICES 16
static inline int pn2_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < PN2_MAX_DEVICES ? dev : PN2_MAX_DEVICES - 1;
}

// Compute units of the CURRENT device (a CPX partition or a CU-masked queue reports its own count); cached per device:
// a process may drive several GPUs (one rank per GPU is the supported layout, but nothing here assumes it).
static inline int pn2_num_cus() {
    static int cus[PN2_MAX_DEVICES] = {0};
    const int slot = pn2_device_slot();
    if (cus[slot] == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cus[slot] = n > 0 ? n : 256;
    }
    return cus[slot];
}

// Kernels that use more than 64 KiB of dynamic LDS raise the function attribute once PER DEVICE (the attribute belongs to the
// device's code object: a flag per process would skip it on the second GPU).  Usage: static Pn2PerDevice done; if
// (!done.get()) { hipFuncSetAttribute(...); done.set(); }
struct Pn2PerDevice {
    bool flag[PN2_MAX_DEVICES] = {false};
    bool get() const { return flag[pn2_device_slot()]; }
    void set() { flag[pn2_device_slot()] = true; }
};
static inline int pn2_raise_dynamic_lds(const void *kernel, Pn2PerDevice &done) {
    if (done.get()) return PN2_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PN2_ELAUNCH;
    done.set();
    return PN2_OK;
}

// The same for launchers that pick the kernel at run time: remembered per (kernel, device).
static inline int pn2_raise_dynamic_lds_once(const void *kernel) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    const std::pair<const void *, int> key(kernel, pn2_device_slot());
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return PN2_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PN2_ELAUNCH;
    done.insert(key);
    return PN2_OK;
}

// PN2_OPAQUE (round 6; HISTORY.md "pn2_fps beside the pooled bf16-split forward"; tools/exp/lds_reader_probe.hip kinds 48 .. 62):
// ON THIS HARDWARE a packed-fp32 operation (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose SECOND source is a VGPR pair read
// through its HIGH half for the low result (op_sel:[0,1], with or without op_sel_hi / neg modifiers) does not write the LOW half of its
// result in lanes 48 .. 63 (the register keeps its previous content; the high half is right) while another wave of the same SIMD executes bf16 MFMAs (v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16): 1.7 M wrong
// results in 960 workgroup runs beside the pooled bf16-split forward, 19 M beside a bare 16x16x32 spinner, 0 alone, 0 beside fp32
// MFMAs or vector-only work; the same select on src0, on src2 of an fma, on an SGPR pair, and the LOW-half broadcast
// (op_sel_hi:[1,0]) measured 0.  hipcc emits the form when it broadcasts the second register of a tuple -- e.g. {cy, cy} from the
// ds_read_b96 that fetched (cx, cy, cz), which is what pn2_fps did: co-resident with the pooled forward of the captured step, 4 .. 100 %
// of its launches returned a different sample list.  Passing the scalars through an empty asm statement makes them plain 32-bit
// values again (no tuple to select a half of): the compiler then broadcasts with the LOW half, a v_mov where needed.
// tools/check_isa.py `pkhi` (tests/test_isa_cpu.py) fails the build if the form appears anywhere in the library.
#define PN2_OPAQUE1(a) asm volatile("" : "+v"(a))
#define PN2_OPAQUE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#define PN2_OPAQUE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// 64-bit max across a wave with xor-shuffles; every lane ends with the result.
__device__ __forceinline__ unsigned long long pn2_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = __shfl_xor(v, m, 64);
        v = o > v ? o : v;
    }
    return v;
}

// 32-bit max over each row of 16 lanes with DPP: xor-1 and xor-2 inside the quads (quad_perm), then row_half_mirror and
// row_mirror fold the quads; every lane of the row ends with the row max.  Written as v_max_u32 with a DPP operand (one
// instruction per step; through __builtin_amdgcn_update_dpp the compiler emits mov + nop + dpp-mov + max).  The
// hazard recogniser does not look inside inline asm, so the two wait states a DPP read needs after a VALU write of
// the same register are spelled out.
__device__ __forceinline__ unsigned pn2_row_max_u32(unsigned v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return v;
}

// 64-bit max over each row of 16 lanes as two 32-bit reductions: the maximum of the high words, then the maximum of
// the low words among the lanes that hold that high word (a 64-bit compare-and-select step costs eight instructions,
// a 32-bit DPP max one; the FPS iteration is bound by exactly this instruction count).
__device__ __forceinline__ unsigned long long pn2_row_max_u64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = pn2_row_max_u32(hi);
    const unsigned ml = pn2_row_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

// 32-bit max over the wave, result uniform: DPP inside the four rows, row_bcast:15 / row_bcast:31 across them (the
// gfx9 wave64 idiom: lane 63 ends with the maximum), one readlane.
__device__ __forceinline__ unsigned pn2_wave_max_u32(unsigned v) {
    v = pn2_row_max_u32(v);
    asm volatile("v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"      // rows 1, 3 take rows 0, 2
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"      // rows 2, 3 take lane 31
                 "s_nop 1"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Four independent 32-bit wave maxima at once: the DPP steps of the four chains interleaved, so each chain's two wait states
// are filled by the other three instead of s_nops (a lone chain is latency-bound: ~12 cycles per step for 4 of work).
// Lane 63 of every value ends with its maximum; the caller reads it (v_readlane).
__device__ __forceinline__ void pn2_wave_max_u32_x4(unsigned &a, unsigned &b, unsigned &c, unsigned &d) {
#define PN2_DPP4(ctl)                                                   \
    "v_max_u32_dpp %0, %0, %0 " ctl "\n\t"                              \
    "v_max_u32_dpp %1, %1, %1 " ctl "\n\t"                              \
    "v_max_u32_dpp %2, %2, %2 " ctl "\n\t"                              \
    "v_max_u32_dpp %3, %3, %3 " ctl "\n\t"
    asm volatile("s_nop 1\n\t"
                 PN2_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 PN2_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 PN2_DPP4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 PN2_DPP4("row_mirror row_mask:0xf bank_mask:0xf")
                 PN2_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 PN2_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef PN2_DPP4
}

// 64-bit max over the wave, result uniform (SGPRs).
__device__ __forceinline__ unsigned long long pn2_wave_max_u64_dpp(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = pn2_wave_max_u32(hi);
    const unsigned ml = pn2_wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
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
