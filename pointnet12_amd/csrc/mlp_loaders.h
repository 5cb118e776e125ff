// Operand loaders shared by the shared-MLP GEMM kernels (mlp.hip: streamed-weight NT / TN cores; mlp_res.hip: the
// weight-resident forward and fused backward kernels).  Everything lives in an anonymous namespace: each translation
// unit gets its own copy.
#pragma once
#include "pn2_common.h"
#include "bn_tail.h"
#include <stdlib.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int NTHREADS = 256; // 4 waves

// relu(bn(y)) exactly as every consumer applies it: the ReLU mask of the backward pass must
// agree bit-for-bit with the forward activation, so there is exactly one spelling of it.
__device__ __forceinline__ float bn_act(float y, float mean, float scale, float beta) {
    return __builtin_fmaf(y - mean, scale, beta);
}


__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ int4 ld4i(const int32_t *p) { return *reinterpret_cast<const int4 *>(p); }

// ----------------------------------------------------------------------------- operand loaders
// A loader produces 4 consecutive k-values of one row of the (virtual) GEMM operand in two halves:
//   issue()  requests the raw global data (and the per-channel constants of those 4 columns) -- nothing
//            is computed, so the requests stay in flight while the previous k-step's MFMAs run;
//   finish() turns the raw registers into operand values right before they are written to LDS.
// (Computing the transform inside the load, as a first version did, makes the loaded value live
// immediately: the wave then waits out the full HBM latency every k-step with no MFMA work to cover it --
// 58 % of the workgroup's cycles, measured with the in-kernel stamps.)
// Rows of one loader thread are m + i*stride, i < IT, all at the same 4 columns k..k+3.

const float4 kZero4 = {0.f, 0.f, 0.f, 0.f};

// Element offset of a row of a position-major matrix.  The GEMM entry points take P < 2^31 rows (checked on the host),
// so row * pitch is ONE v_mad_u64_u32 -- as int64 * int the compiler forms it from two 32-bit multiplies, a 64-bit
// multiply-add and an add3, per request and k-step.
__device__ __forceinline__ int64_t row_off(int64_t row, int ld) {
    return (int64_t)((uint64_t)(uint32_t)row * (uint64_t)(uint32_t)ld);
}

__device__ float4 pn2_zero_page[4];          // always-zero source for predicated-off operand requests
// Its address travels to the kernels as an argument (last member of every loader / of BMat / of the dgrad epilogue): a
// __device__ symbol is reached through the GOT, and inside the stage loops that was one s_load + s_waitcnt lgkmcnt(0)
// per predicated request -- the wait also drains the LDS stores issued just before it.
static inline int pow2_shift(int v) {          // log2(v) for a power of two, else -1
    if (v <= 0 || (v & (v - 1))) return -1;
    int sft = 0;
    while ((1 << sft) < v) ++sft;
    return sft;
}

static const float *zero_page_dev() {
    static const float *p = [] {
        void *q = nullptr;
        return hipGetSymbolAddress(&q, HIP_SYMBOL(pn2_zero_page)) == hipSuccess ? reinterpret_cast<const float *>(q) : nullptr;
    }();
    return p;
}

// The per-channel constants of a loader's 4 columns are (re)loaded by params() at finish time -- they hit L1/L2,
// and keeping them out of the in-flight register set is what lets the kernels run at 3-4 workgroups per CU
// without spilling.

// Per-channel constants of the Dy loaders (the four coefficient rows of pn2_bn_bwd_coef) are copied into LDS once per
// workgroup (kTab rows of K4 floats, dynamic shared memory) and read from there every k-step: a global/L1 round trip
// right before the operand transform would otherwise sit exposed in front of every LDS store of the dgrad kernel.
struct LoadPlain {          // X as stored
    static constexpr int kTab = 0;
    __device__ __forceinline__ const float *tab_src() const { return nullptr; }
    const float *X; int ldx; const float *zp;
    static constexpr int kRegs = 4;
    __device__ __forceinline__ void prologue() const {}
    LoadPlain without_lazy() const { return *this; }
    template <int IT> struct Raw { float4 x[IT]; };
    struct Params {};
    template <int IT>
    __device__ __forceinline__ void issue(Raw<IT> &r, int64_t m, int stride, int k, int64_t rows, bool kvalid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t mi = m + (int64_t)i * stride;
            r.x[i] = ld4((kvalid && mi < rows) ? X + row_off(mi, ldx) + k : zp);
        }
    }
    __device__ __forceinline__ Params params(int, bool) const { return Params(); }
    __device__ __forceinline__ Params params_tab(const float *, int, int, bool) const { return Params(); }
    template <int IT>
    __device__ __forceinline__ float4 finish(const Raw<IT> &r, int i, bool, const Params &) const { return r.x[i]; }
};

struct LoadBnRelu {         // relu(bn(Y_prev)) formed from the pre-BN tensor
    const float *X; int ldx; const float *aff; const float *zp;
    LazyBn lz;              // consumer-side BatchNorm: `aff` is filled from the producer's sums by the kernel's prologue (bn_tail.h)
    __device__ __forceinline__ void prologue() const { lazy_bn_prologue(lz); }
    LoadBnRelu without_lazy() const { LoadBnRelu o = *this; o.lz = LazyBn{}; return o; }
    static constexpr int kRegs = 4;
    static constexpr int kTab = 0;          // its constants ride in the prefetched Raw registers
    __device__ __forceinline__ const float *tab_src() const { return nullptr; }
    template <int IT> struct Raw { float4 x[IT]; float4 mu, sc, be; };   // 12 constant registers: fits at 4 WG/CU
    struct Params {};
    template <int IT>
    __device__ __forceinline__ void issue(Raw<IT> &r, int64_t m, int stride, int k, int64_t rows, bool kvalid) const {
        Affine a(aff, ldx);
        // every request is always issued (invalid ones read the zero page): straight-line code lets the compiler
        // count the outstanding requests instead of draining them
        r.mu = ld4(kvalid ? a.mean + k : zp);          // pad / invalid columns: scale = beta = 0 -> operand 0
        r.sc = ld4(kvalid ? a.scale + k : zp);
        r.be = ld4(kvalid ? a.beta + k : zp);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t mi = m + (int64_t)i * stride;
            r.x[i] = ld4((kvalid && mi < rows) ? X + row_off(mi, ldx) + k : zp);
        }
    }
    __device__ __forceinline__ Params params(int, bool) const { return Params(); }
    __device__ __forceinline__ Params params_tab(const float *, int, int, bool) const { return Params(); }
    template <int IT>
    __device__ __forceinline__ float4 finish(const Raw<IT> &r, int i, bool valid, const Params &) const {
        const float4 x = r.x[i];
        float4 o;
        o.x = fmaxf(bn_act(x.x, r.mu.x, r.sc.x, r.be.x), 0.f);
        o.y = fmaxf(bn_act(x.y, r.mu.y, r.sc.y, r.be.y), 0.f);
        o.z = fmaxf(bn_act(x.z, r.mu.z, r.sc.z, r.be.z), 0.f);
        o.w = fmaxf(bn_act(x.w, r.mu.w, r.sc.w, r.be.w), 0.f);
        return valid ? o : kZero4;                      // rows past the end must contribute nothing
    }
};

// The same operand for the TN (wgrad) kernel: there a thread's channel quad is fixed for the whole launch, so the three
// constant rows are fetched ONCE (params(), hoisted in front of the position loop) instead of with every stage.
struct LoadBnReluFixed {
    const float *X; int ldx; const float *aff; const float *zp;
    __device__ __forceinline__ void prologue() const {}
    template <int IT> struct Raw { float4 x[IT]; };
    struct Params { float4 mu, sc, be; };
    template <int IT>
    __device__ __forceinline__ void issue(Raw<IT> &r, int64_t m, int stride, int k, int64_t rows, bool kvalid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t mi = m + (int64_t)i * stride;
            r.x[i] = ld4((kvalid && mi < rows) ? X + row_off(mi, ldx) + k : zp);
        }
    }
    __device__ __forceinline__ Params params(int k, bool kvalid) const {
        Affine a(aff, ldx);
        Params q;
        q.mu = ld4(kvalid ? a.mean + k : zp);          // pad / invalid columns: scale = beta = 0 -> operand 0
        q.sc = ld4(kvalid ? a.scale + k : zp);
        q.be = ld4(kvalid ? a.beta + k : zp);
        return q;
    }
    template <int IT>
    __device__ __forceinline__ float4 finish(const Raw<IT> &r, int i, bool valid, const Params &q) const {
        const float4 x = r.x[i];
        float4 o;
        o.x = fmaxf(bn_act(x.x, q.mu.x, q.sc.x, q.be.x), 0.f);
        o.y = fmaxf(bn_act(x.y, q.mu.y, q.sc.y, q.be.y), 0.f);
        o.z = fmaxf(bn_act(x.z, q.mu.z, q.sc.z, q.be.z), 0.f);
        o.w = fmaxf(bn_act(x.w, q.mu.w, q.sc.w, q.be.w), 0.f);
        return valid ? o : kZero4;
    }
};

struct DyParams { float4 c0, q1, q0, mu; };

__device__ __forceinline__ DyParams dy_params(const float *coef, int ldc, int k, bool kvalid, const float *zp) {
    DyParams q;
    q.c0 = ld4(kvalid ? coef + k : zp);
    q.q1 = ld4(kvalid ? coef + ldc + k : zp);
    q.q0 = ld4(kvalid ? coef + 2 * ldc + k : zp);
    q.mu = ld4(kvalid ? coef + 3 * ldc + k : zp);
    return q;
}

__device__ __forceinline__ DyParams dy_params_tab(const float *tab, int K4, int k, bool kvalid) {
    DyParams q;
    const int kk = kvalid ? k : 0;                 // the table holds the four rows back to back, K4 floats each
    q.c0 = *reinterpret_cast<const float4 *>(tab + kk);
    q.q1 = *reinterpret_cast<const float4 *>(tab + K4 + kk);
    q.q0 = *reinterpret_cast<const float4 *>(tab + 2 * K4 + kk);
    q.mu = *reinterpret_cast<const float4 *>(tab + 3 * K4 + kk);
    return q;
}

// dY = c0*dZ + q1*(y-mean) + q0   (BatchNorm backward folded into per-channel coefficients)
__device__ __forceinline__ float4 dy_from(const float4 dz, const float4 y, const DyParams &q) {
    float4 o;
    o.x = __builtin_fmaf(q.c0.x, dz.x, __builtin_fmaf(q.q1.x, y.x - q.mu.x, q.q0.x));
    o.y = __builtin_fmaf(q.c0.y, dz.y, __builtin_fmaf(q.q1.y, y.y - q.mu.y, q.q0.y));
    o.z = __builtin_fmaf(q.c0.z, dz.z, __builtin_fmaf(q.q1.z, y.z - q.mu.z, q.q0.z));
    o.w = __builtin_fmaf(q.c0.w, dz.w, __builtin_fmaf(q.q1.w, y.w - q.mu.w, q.q0.w));
    return o;
}

struct LoadDyDense {
    const float *dZ; int ldz; const float *Y; int ldy; const float *coef; int ldc; const float *zp;
    LazyCoef lc;            // consumer-side BatchNorm backward: `coef` is filled from the reductions by the kernel's prologue
    __device__ __forceinline__ void prologue() const { lazy_coef_prologue(lc); }
    LoadDyDense without_lazy() const { LoadDyDense o = *this; o.lc = LazyCoef{}; return o; }
    static constexpr int kRegs = 8;
    template <int IT> struct Raw { float4 dz[IT], y[IT]; };
    typedef DyParams Params;
    template <int IT>
    __device__ __forceinline__ void issue(Raw<IT> &r, int64_t m, int stride, int k, int64_t rows, bool kvalid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t mi = m + (int64_t)i * stride;
            const bool v = kvalid && mi < rows;
                r.dz[i] = ld4(v ? dZ + row_off(mi, ldz) + k : zp);
            r.y[i] = ld4(v ? Y + row_off(mi, ldy) + k : zp);
        }
    }
    __device__ __forceinline__ Params params(int k, bool kvalid) const { return dy_params(coef, ldc, k, kvalid, zp); }
    static constexpr int kTab = 4;
    __device__ __forceinline__ const float *tab_src() const { return coef; }       // 4 rows of pitch ldc == K4
    __device__ __forceinline__ Params params_tab(const float *tab, int K4, int k, bool kvalid) const {
        return dy_params_tab(tab, K4, k, kvalid);
    }
    template <int IT>
    __device__ __forceinline__ float4 finish(const Raw<IT> &r, int i, bool valid, const Params &q) const {
        return valid ? dy_from(r.dz[i], r.y[i], q) : kZero4;
    }
};

// Same, with dZ implied by the max-pool: dZ[g*Kp+kk, c] = dZp[g,c] if kk == arg[g,c], where
// dZp = dOut * (out > 0) was written once by pn2_pool_bwd_reduce (keeps this loader at 3 requests per row).
struct LoadDyPooled {
    const float *dZp; int ldo; const int32_t *arg; int Kp;
    const float *Y; int ldy; const float *coef; int ldc; const float *zp; int kshift;   // kshift: log2(Kp) or -1
    LazyCoef lc;
    __device__ __forceinline__ void prologue() const { lazy_coef_prologue(lc); }
    LoadDyPooled without_lazy() const { LoadDyPooled o = *this; o.lc = LazyCoef{}; return o; }
    static constexpr int kRegs = 13;
    template <int IT> struct Raw { float4 go[IT], y[IT]; int4 a[IT]; int kk[IT]; };
    typedef DyParams Params;
    template <int IT>
    __device__ __forceinline__ void issue(Raw<IT> &r, int64_t m, int stride, int k, int64_t rows, bool kvalid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t mi = m + (int64_t)i * stride;
            const bool v = kvalid && mi < rows;
            // group and position inside it; P < 2^31 (checked by the host wrapper).  Kp is a power of two in every network
            // of the reference (16 .. 128): a shift and a mask instead of the ~12-instruction division sequence.
            const unsigned g = kshift >= 0 ? (unsigned)mi >> kshift : (unsigned)mi / (unsigned)Kp;
            r.kk[i] = (int)((unsigned)mi - g * (unsigned)Kp);
                r.go[i] = ld4(v ? dZp + row_off(g, ldo) + k : zp);           // invalid: dZp = 0 -> dz = 0 whatever arg says
            r.a[i] = ld4i(v ? arg + row_off(g, ldo) + k : reinterpret_cast<const int32_t *>(zp));
            r.y[i] = ld4(v ? Y + row_off(mi, ldy) + k : zp);
        }
    }
    __device__ __forceinline__ Params params(int k, bool kvalid) const { return dy_params(coef, ldc, k, kvalid, zp); }
    static constexpr int kTab = 4;
    __device__ __forceinline__ const float *tab_src() const { return coef; }       // 4 rows of pitch ldc == K4
    __device__ __forceinline__ Params params_tab(const float *tab, int K4, int k, bool kvalid) const {
        return dy_params_tab(tab, K4, k, kvalid);
    }
    template <int IT>
    __device__ __forceinline__ float4 finish(const Raw<IT> &r, int i, bool valid, const Params &q) const {
        const float4 go = r.go[i];
        const int4 a = r.a[i];
        const int kk = r.kk[i];
        float4 dz;
        dz.x = a.x == kk ? go.x : 0.f;
        dz.y = a.y == kk ? go.y : 0.f;
        dz.z = a.z == kk ? go.z : 0.f;
        dz.w = a.w == kk ? go.w : 0.f;
        return valid ? dy_from(dz, r.y[i], q) : kZero4;
    }
};

}  // namespace
