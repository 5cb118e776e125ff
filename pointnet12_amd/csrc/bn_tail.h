// BatchNorm affine blocks and the fused "tails" that finish a per-channel reduction inside its producer kernel.
#pragma once
#include "pn2_common.h"

// (outside the anonymous namespace: these cross translation units as arguments of the pn2_wide_* / pn2_fwd_res launchers)
struct LazyBn {
    const double *stats;        // nullptr: nothing to do (the block was written by pn2_bn_finalize or by an earlier launch)
    const float *gamma, *beta;
    float *affine, *rmean, *rvar;
    int64_t *nbt;
    double inv_p, unbias;
    float eps, momentum;
    int C;
};

struct LazyCoef {
    const double *red;          // nullptr: nothing to do
    const float *gamma, *aff;
    float *coef, *dgamma, *dbeta;
    double inv_p;
    int C, accumulate;
};

namespace {

struct Affine {   // views into a float[4*ld] affine block (see pn2.h)
    const float *mean, *scale, *beta, *invstd;
    __device__ __host__ Affine(const float *base, int ld) : mean(base), scale(base + ld), beta(base + 2 * ld), invstd(base + 3 * ld) {}
};

// ----------------------------------------------------------------------------- fused BatchNorm tails
// The per-channel work that follows a reduction (statistics -> affine block in the forward pass, reductions ->
// backward coefficients in the backward pass) is a few hundred flops, but as a kernel of its own it is one more
// ~5 us hop on the stream's dependency chain, 50 times per step.  Producers therefore take an optional "tail":
// every workgroup bumps a ticket after its atomics are visible device-wide, and the workgroup that draws the last
// ticket reads the finished sums (device-scope loads) and does the per-channel work before the kernel ends.
__device__ __forceinline__ double ld_f64_device(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FinTail {            // forward: statistics -> affine block (+ running statistics)
    unsigned *ticket;       // nullptr: no tail
    const float *gamma, *beta;
    float eps, momentum;
    float *rmean, *rvar;
    int64_t *nbt;
    float *affine;
    double inv_p, unbias;
};

struct CoefTail {           // backward: reductions -> coefficients of dY = c0*dZ + q1*(y-mean) + q0, dgamma, dbeta
    unsigned *ticket;       // nullptr: no tail
    const float *gamma, *aff;
    int use_batch;
    float *coef, *dgamma, *dbeta;
    int accumulate;
    double inv_p;
};

// training-mode statistics of channel c -> affine block entries (the arithmetic of pn2_bn_finalize)
__device__ __forceinline__ void bn_finalize_channel(double s0, double s1, int c, int ld, double inv_p, double unbias,
                                                    const float *gamma, const float *beta, float eps, float momentum,
                                                    float *rmean, float *rvar, float *affine) {
    const double mean = s0 * inv_p;
    double var = s1 * inv_p - mean * mean;
    if (var < 0.0) var = 0.0;
    if (rmean) rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
    if (rvar) rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * var * unbias);
    const double invstd = 1.0 / sqrt(var + (double)eps);
    affine[c] = (float)mean;
    affine[ld + c] = (float)((double)gamma[c] * invstd);
    affine[2 * ld + c] = beta[c];
    affine[3 * ld + c] = (float)invstd;
}

__device__ __forceinline__ void bn_coef_channel(double r0, double r1, int c, int ld, double inv_p, const float *gamma,
                                                const float *aff, int use_batch, float *coef, float *dgamma,
                                                float *dbeta, int accumulate) {
    Affine a(aff, ld);
    const double c0 = (double)gamma[c] * (double)a.invstd[c];
    coef[c] = (float)c0;
    coef[ld + c] = use_batch ? (float)(-c0 * (double)a.invstd[c] * r1 * inv_p) : 0.f;
    coef[2 * ld + c] = use_batch ? (float)(-c0 * r0 * inv_p) : 0.f;
    coef[3 * ld + c] = a.mean[c];
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)r1 : (float)r1;     // one writer per channel
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)r0 : (float)r0;
}

// Called by every thread of every workgroup after the workgroup's atomics were issued.  Returns true in the
// workgroup that drew the last ticket.
//
// Ordering without a device-scope fence: the sums are agent-scope atomics (performed at the device's point of
// coherence, not in an XCD's L2), so "this thread's atomics are done" only needs their completion
// (s_waitcnt, a workgroup-scope release), then the barrier, then the ticket -- itself an agent-scope atomic.  The
// reader uses agent-scope atomic loads.  A full __threadfence() here costs a write-back + invalidate of the
// XCD's L2 per workgroup (measured: MSG-SemSeg step 9.2 -> 12.8 ms), which is what this avoids; the tail's own
// plain stores (affine block / coefficients) are released by the end of the kernel as usual.
__device__ __forceinline__ bool tail_is_last_block(unsigned *ticket, unsigned total_blocks) {
    __shared__ unsigned s_last;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total_blocks - 1 ? 1u : 0u;
    __syncthreads();
    return s_last != 0;
}

__device__ __forceinline__ void run_fin_tail(const FinTail &f, const double *stats, int C, int nthreads) {
    const int ld = (C + 3) & ~3;
    for (int c = threadIdx.x; c < C; c += nthreads) {
        double s0 = 0.0, s1 = 0.0;
        for (int r = 0; r < PN2_STAT_REPLICAS; ++r) {
            s0 += ld_f64_device(stats + r * 2 * C + c);
            s1 += ld_f64_device(stats + r * 2 * C + C + c);
        }
        bn_finalize_channel(s0, s1, c, ld, f.inv_p, f.unbias, f.gamma, f.beta, f.eps, f.momentum, f.rmean, f.rvar, f.affine);
    }
    if (threadIdx.x == 0) {
        if (f.nbt) *f.nbt += 1;
        *f.ticket = 0;                                 // reusable without another memset
    }
}

__device__ __forceinline__ void run_coef_tail(const CoefTail &t, const double *red, int C, int nthreads) {
    const int ld = (C + 3) & ~3;
    for (int c = threadIdx.x; c < C; c += nthreads) {
        double r0 = 0.0, r1 = 0.0;
        for (int r = 0; r < PN2_STAT_REPLICAS; ++r) {
            r0 += ld_f64_device(red + r * 2 * C + c);
            r1 += ld_f64_device(red + r * 2 * C + C + c);
        }
        bn_coef_channel(r0, r1, c, ld, t.inv_p, t.gamma, t.aff, t.use_batch, t.coef, t.dgamma, t.dbeta, t.accumulate);
    }
    if (threadIdx.x == 0) *t.ticket = 0;
}

// ----------------------------------------------------------------------------- consumer-side BatchNorm (round 4)
// The statistics -> affine block step (forward) and the reductions -> coefficients step (backward) as a PROLOGUE of the first
// kernel that reads the block, instead of a 4.8 us launch of its own on the dependency chain (50 such hops per MSG-SemSeg step).
// The producing launch has ended, so its fp64 sums are complete and visible; EVERY workgroup of the consumer recomputes the
// block (the same fp64 arithmetic as bn_finalize_channel / bn_coef_channel: bit-identical values from every workgroup),
// stores it to the block's global buffer -- where the kernel's loaders and every later kernel read it exactly as before --
// waits for its own stores and meets at a workgroup barrier.  A CU's vector L1 is write-through and shared by the waves of a
// workgroup: loads issued behind that barrier see the workgroup's own stores (workgroup-scope coherence, no fence needed),
// and whatever another workgroup has written to the same lines is the same bytes.  Workgroup (0, 0, 0) alone updates what must
// be written once: running statistics, num_batches_tracked, dgamma / dbeta.
__device__ __forceinline__ void lazy_bn_prologue(const LazyBn &z) {
    if (z.stats == nullptr) return;                                // (uniform: a kernel argument)
    const bool first = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
    const int ld = (z.C + 3) & ~3;
    for (int c = threadIdx.x; c < z.C; c += blockDim.x) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int r = 0; r < PN2_STAT_REPLICAS; ++r) { s0 += z.stats[r * 2 * z.C + c]; s1 += z.stats[r * 2 * z.C + z.C + c]; }
        bn_finalize_channel(s0, s1, c, ld, z.inv_p, z.unbias, z.gamma, z.beta, z.eps, z.momentum, first ? z.rmean : nullptr,
                            first ? z.rvar : nullptr, z.affine);
    }
    if (first && threadIdx.x == 0 && z.nbt) *z.nbt += 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's stores have reached the L2
    __syncthreads();
}

__device__ __forceinline__ void lazy_coef_prologue(const LazyCoef &z) {
    if (z.red == nullptr) return;
    const bool first = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
    const int ld = (z.C + 3) & ~3;
    for (int c = threadIdx.x; c < z.C; c += blockDim.x) {
        double r0 = 0.0, r1 = 0.0;
#pragma unroll
        for (int r = 0; r < PN2_STAT_REPLICAS; ++r) { r0 += z.red[r * 2 * z.C + c]; r1 += z.red[r * 2 * z.C + z.C + c]; }
        bn_coef_channel(r0, r1, c, ld, z.inv_p, z.gamma, z.aff, 1, z.coef, first ? z.dgamma : nullptr, first ? z.dbeta : nullptr,
                        z.accumulate);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

static inline LazyBn make_lazy_bn(const pn2_bn_lazy *z) {
    LazyBn o{};
    if (z && z->stats) {
        const double n = (double)z->count;
        o = LazyBn{z->stats, z->gamma, z->beta, z->affine, z->running_mean, z->running_var, z->num_batches_tracked, 1.0 / n,
                   z->count > 1 ? n / (n - 1.0) : 1.0, z->eps, z->momentum, z->C};
    }
    return o;
}

static inline LazyCoef make_lazy_coef(const pn2_bn_coef_lazy *z) {
    LazyCoef o{};
    if (z && z->red) o = LazyCoef{z->red, z->gamma, z->affine, z->coef, z->dgamma, z->dbeta, 1.0 / (double)z->count, z->C, z->accumulate};
    return o;
}

static inline bool lazy_bn_ok(const pn2_bn_lazy *z, const float *block, int C) {
    return z == nullptr || (z->stats && z->gamma && z->beta && z->affine && z->affine == block && z->C == C && z->count > 0);
}

static inline bool lazy_coef_ok(const pn2_bn_coef_lazy *z, const float *block, int C) {
    return z == nullptr || (z->red && z->gamma && z->affine && z->coef && z->coef == block && z->C == C && z->count > 0);
}

}  // namespace
