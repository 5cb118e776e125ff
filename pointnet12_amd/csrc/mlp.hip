// Shared MLP of the set-abstraction / feature-propagation modules on gfx950:
// 1x1 convolution (a [P,K] x [N,K]^T GEMM on v_mfma_f32_32x32x2_f32, exact fp32), training-mode
// BatchNorm (statistics accumulated in the GEMM epilogue, applied on the fly by the consumer),
// ReLU, max over the K neighbours, and the matching backward (dgrad / wgrad / BN reductions).
//
// Activations are position-major: row p = one grouped position, channels contiguous.
// Replaces model/pointnet_util.py:194-199, :251-256, :309-312 and their autograd.
#include "pn2_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;        // k-depth of one LDS stage
constexpr int LDP = BK + 4;   // LDS row pitch in floats: 16 consecutive rows cover all 64 banks once
constexpr int NTHREADS = 256; // 4 waves

// relu(bn(y)) exactly as every consumer applies it: the ReLU mask of the backward pass must
// agree bit-for-bit with the forward activation, so there is exactly one spelling of it.
__device__ __forceinline__ float bn_act(float y, float mean, float scale, float beta) {
    return __builtin_fmaf(y - mean, scale, beta);
}

struct Affine {   // views into a float[4*ld] affine block (see pn2.h)
    const float *mean, *scale, *beta, *invstd;
    __device__ __host__ Affine(const float *base, int ld) : mean(base), scale(base + ld), beta(base + 2 * ld), invstd(base + 3 * ld) {}
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ int4 ld4i(const int32_t *p) { return *reinterpret_cast<const int4 *>(p); }

// ----------------------------------------------------------------------------- operand loaders
// Each loader returns 4 consecutive k-values of one row of the (virtual) GEMM operand.

struct LoadPlain {          // X as stored
    const float *X; int ldx;
    __device__ __forceinline__ float4 operator()(int64_t m, int k) const { return ld4(X + m * ldx + k); }
};

struct LoadBnRelu {         // relu(bn(Y_prev)) formed on the fly from the pre-BN tensor
    const float *X; int ldx; const float *aff;
    __device__ __forceinline__ float4 operator()(int64_t m, int k) const {
        float4 x = ld4(X + m * ldx + k);
        Affine a(aff, ldx);
        float4 mu = ld4(a.mean + k), sc = ld4(a.scale + k), be = ld4(a.beta + k);
        float4 r;
        r.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
        r.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
        r.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
        r.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
        return r;
    }
};

// dY = c0*dZ + q1*(y-mean) + q0   (BatchNorm backward folded into per-channel coefficients)
struct LoadDyDense {
    const float *dZ; int ldz; const float *Y; int ldy; const float *coef; int ldc;
    __device__ __forceinline__ float4 operator()(int64_t m, int k) const {
        float4 dz = ld4(dZ + m * ldz + k), y = ld4(Y + m * ldy + k);
        float4 c0 = ld4(coef + k), q1 = ld4(coef + ldc + k), q0 = ld4(coef + 2 * ldc + k), mu = ld4(coef + 3 * ldc + k);
        float4 r;
        r.x = __builtin_fmaf(c0.x, dz.x, __builtin_fmaf(q1.x, y.x - mu.x, q0.x));
        r.y = __builtin_fmaf(c0.y, dz.y, __builtin_fmaf(q1.y, y.y - mu.y, q0.y));
        r.z = __builtin_fmaf(c0.z, dz.z, __builtin_fmaf(q1.z, y.z - mu.z, q0.z));
        r.w = __builtin_fmaf(c0.w, dz.w, __builtin_fmaf(q1.w, y.w - mu.w, q0.w));
        return r;
    }
};

// Same, with dZ implied by the max-pool: dZ[g*Kp+kk, c] = dOut[g,c] if kk == arg[g,c] and out[g,c] > 0.
struct LoadDyPooled {
    const float *dOut; int ldo; const float *out; const int32_t *arg; int Kp;
    const float *Y; int ldy; const float *coef; int ldc;
    __device__ __forceinline__ float4 operator()(int64_t m, int k) const {
        int64_t g = m / Kp;
        int kk = (int)(m - g * Kp);
        float4 go = ld4(dOut + g * ldo + k), o = ld4(out + g * ldo + k);
        int4 a = ld4i(arg + g * ldo + k);
        float4 y = ld4(Y + m * ldy + k);
        float4 c0 = ld4(coef + k), q1 = ld4(coef + ldc + k), q0 = ld4(coef + 2 * ldc + k), mu = ld4(coef + 3 * ldc + k);
        float4 dz;
        dz.x = (a.x == kk && o.x > 0.f) ? go.x : 0.f;
        dz.y = (a.y == kk && o.y > 0.f) ? go.y : 0.f;
        dz.z = (a.z == kk && o.z > 0.f) ? go.z : 0.f;
        dz.w = (a.w == kk && o.w > 0.f) ? go.w : 0.f;
        float4 r;
        r.x = __builtin_fmaf(c0.x, dz.x, __builtin_fmaf(q1.x, y.x - mu.x, q0.x));
        r.y = __builtin_fmaf(c0.y, dz.y, __builtin_fmaf(q1.y, y.y - mu.y, q0.y));
        r.z = __builtin_fmaf(c0.z, dz.z, __builtin_fmaf(q1.z, y.z - mu.z, q0.z));
        r.w = __builtin_fmaf(c0.w, dz.w, __builtin_fmaf(q1.w, y.w - mu.w, q0.w));
        return r;
    }
};

// ----------------------------------------------------------------------------- epilogues (NT GEMM)

struct EpiFwd {             // y = acc + bias -> Y; per-channel sum(y), sum(y*y) -> stats
    float *Y; int ldy; const float *bias; double *stats;
    static constexpr bool kHasStats = true;
    __device__ __forceinline__ bool want_stats() const { return stats != nullptr; }
    __device__ __forceinline__ void prep(int n, bool nvalid, float (&c)[4]) const { c[0] = nvalid ? bias[n] : 0.f; }
    __device__ __forceinline__ void elem(int64_t m, int n, float acc, const float (&c)[4], float &s0, float &s1) const {
        float y = acc + c[0];
        Y[m * ldy + n] = y;
        s0 += y;
        s1 = __builtin_fmaf(y, y, s1);
    }
    __device__ __forceinline__ void flush(int n, int N, double s0, double s1) const {
        atomicAdd(stats + n, s0);
        atomicAdd(stats + N + n, s1);
    }
};

struct EpiDgradMask {       // dZprev = acc * relu'(prev) -> dXout; sum(dZprev), sum(dZprev*yhat_prev) -> red
    float *dX; int ldx; const float *prevY; int ldp; const float *aff; int lda; double *red;
    static constexpr bool kHasStats = true;
    __device__ __forceinline__ bool want_stats() const { return red != nullptr; }
    __device__ __forceinline__ void prep(int n, bool nvalid, float (&c)[4]) const {
        Affine a(aff, lda);
        c[0] = nvalid ? a.mean[n] : 0.f; c[1] = nvalid ? a.scale[n] : 0.f;
        c[2] = nvalid ? a.beta[n] : 0.f; c[3] = nvalid ? a.invstd[n] : 0.f;
    }
    __device__ __forceinline__ void elem(int64_t m, int n, float acc, const float (&c)[4], float &s0, float &s1) const {
        float y = prevY[m * ldp + n];
        float dz = bn_act(y, c[0], c[1], c[2]) > 0.f ? acc : 0.f;
        dX[m * ldx + n] = dz;
        s0 += dz;
        s1 = __builtin_fmaf(dz, (y - c[0]) * c[3], s1);
    }
    __device__ __forceinline__ void flush(int n, int N, double s0, double s1) const {
        atomicAdd(red + n, s0);
        atomicAdd(red + N + n, s1);
    }
};

struct EpiStore {           // first layer: dX0 = acc
    float *dX; int ldx;
    static constexpr bool kHasStats = false;
    __device__ __forceinline__ bool want_stats() const { return false; }
    __device__ __forceinline__ void prep(int, bool, float (&)[4]) const {}
    __device__ __forceinline__ void elem(int64_t m, int n, float acc, const float (&)[4], float &, float &) const {
        dX[m * ldx + n] = acc;
    }
    __device__ __forceinline__ void flush(int, int, double, double) const {}
};

// ----------------------------------------------------------------------------- NT GEMM core
// C[P,N] = A[P,K] * Bw[N,K]^T.  A rows come from a loader, Bw is a plain padded matrix.
// 4 waves as WR x WC, wave tile (BM/WR) x (BN/WC) built from 32x32 MFMA tiles.
// Each workgroup walks row tiles blockIdx.x, +gridDim.x, ... so per-channel reductions are
// kept in registers across tiles and flushed once (one fp64 atomic per channel per workgroup).
//
// LDS operands are K-contiguous.  One ds_read_b128 gives a lane 4 k-values (k = 8*kb + 4*(lane>>5) + e);
// MFMA e of the group consumes element e from both operands, i.e. the k-order inside an
// 8-block is permuted identically for A and B, which leaves every product pair intact.
template <int BM, int BN, int WR, int WC, class ALoad, class Epi>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(ALoad aload, const float *__restrict__ Bw, int ldb,
                                                           int64_t P, int K4, int N, Epi epi) {
    static_assert(WR * WC == 4, "four waves");
    constexpr int WTM = BM / WR, WTN = BN / WC;      // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;      // MFMA tiles per wave
    constexpr int A_IT = BM * (BK / 4) / NTHREADS, B_IT = BN * (BK / 4) / NTHREADS;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 loader threads");

    __shared__ float As[BM * LDP];
    __shared__ float Bs[BN * LDP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int l31 = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int64_t tiles_m = (P + BM - 1) / BM;

    double st0[TN], st1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { st0[j] = 0.0; st1[j] = 0.0; }

    for (int64_t tile = blockIdx.x; tile < tiles_m; tile += gridDim.x) {
        const int64_t m0 = tile * BM;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int k0 = 0; k0 < K4; k0 += BK) {
            float4 ra[A_IT], rb[B_IT];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int f = t + i * NTHREADS, row = f >> 3, kq = (f & 7) * 4;
                int64_t m = m0 + row;
                ra[i] = (m < P && k0 + kq < K4) ? aload(m, k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int f = t + i * NTHREADS, row = f >> 3, kq = (f & 7) * 4;
                int n = n0 + row;
                rb[i] = (n < N && k0 + kq < K4) ? ld4(Bw + (int64_t)n * ldb + k0 + kq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int f = t + i * NTHREADS, row = f >> 3, kq = (f & 7) * 4;
                *reinterpret_cast<float4 *>(&As[row * LDP + kq]) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int f = t + i * NTHREADS, row = f >> 3, kq = (f & 7) * 4;
                *reinterpret_cast<float4 *>(&Bs[row * LDP + kq]) = rb[i];
            }
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < BK / 8; ++kb) {
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&As[(wr * WTM + i * 32 + l31) * LDP + kb * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(&Bs[(wc * WTN + j * 32 + l31) * LDP + kb * 8 + lh * 4]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }

        // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * WTN + j * 32 + l31;
            const bool nvalid = n < N;
            float c[4];
            epi.prep(n, nvalid, c);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (nvalid && m < P) epi.elem(m, n, acc[i][j][r], c, s0, s1);
                }
            }
            if (Epi::kHasStats) { st0[j] += (double)s0; st1[j] += (double)s1; }
        }
    }

    if (Epi::kHasStats && epi.want_stats()) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double a0 = st0[j] + __shfl_xor(st0[j], 32, 64);
            double a1 = st1[j] + __shfl_xor(st1[j], 32, 64);
            const int n = n0 + wc * WTN + j * 32 + l31;
            if (lh == 0 && n < N) epi.flush(n, N, a0, a1);
        }
    }
}

template <int BM, int BN, int WR, int WC, class ALoad, class Epi>
int launch_nt(ALoad aload, const float *Bw, int ldb, int64_t P, int K4, int N, Epi epi, hipStream_t s) {
    int64_t tiles_m = pn2_cdiv(P, BM);
    unsigned tiles_n = (unsigned)pn2_cdiv(N, BN);
    // enough workgroups to fill 256 CUs several times over, few enough that the per-workgroup
    // statistics flush stays negligible
    int64_t cap = 2048 / tiles_n;
    if (cap < 256) cap = 256;
    unsigned gx = (unsigned)(tiles_m < cap ? tiles_m : cap);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WR, WC, ALoad, Epi>), dim3(gx, tiles_n), dim3(NTHREADS), 0, s, aload, Bw,
                       ldb, P, K4, N, epi);
    return pn2_launch_status();
}

template <class ALoad, class Epi>
int dispatch_nt(ALoad aload, const float *Bw, int ldb, int64_t P, int K4, int N, Epi epi, hipStream_t s) {
    if (N <= 32) return launch_nt<128, 32, 4, 1>(aload, Bw, ldb, P, K4, N, epi, s);
    if (N <= 64) return launch_nt<128, 64, 2, 2>(aload, Bw, ldb, P, K4, N, epi, s);
    if (N <= 128 || N > 256) return launch_nt<128, 128, 2, 2>(aload, Bw, ldb, P, K4, N, epi, s);
    return launch_nt<64, 256, 1, 4>(aload, Bw, ldb, P, K4, N, epi, s);
}

// ----------------------------------------------------------------------------- TN GEMM (wgrad)
// dW[M,N] += sum_p dY[p,m] * X[p,n]: both operands arrive position-major and are consumed
// "down the columns" (ds_read_b32, consecutive lanes on consecutive channels: conflict free).
// Split over P across gridDim.z; partial tiles are combined with fp32 atomics (256-B contiguous
// per wave-instruction).
constexpr int WG_BP = 32;     // positions per LDS stage

template <int BM, int BN, int WR, int WC, class DyLoad, class XLoad>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(DyLoad dyload, XLoad xload, int64_t P, int64_t chunk, int M,
                                                           int N, float *__restrict__ dW, int lddw,
                                                           float *__restrict__ dbias) {
    constexpr int WTM = BM / WR, WTN = BN / WC;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = WG_BP * (BM / 4) / NTHREADS, B_IT = WG_BP * (BN / 4) / NTHREADS;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");
    __shared__ float As[WG_BP * (BM + 4)];
    __shared__ float Bs[WG_BP * (BN + 4)];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int64_t p_begin = (int64_t)blockIdx.z * chunk;
    const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[A_IT][4];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) bsum[i][0] = bsum[i][1] = bsum[i][2] = bsum[i][3] = 0.f;

    for (int64_t p0 = p_begin; p0 < p_end; p0 += WG_BP) {
        float4 ra[A_IT], rb[B_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int f = t + i * NTHREADS, row = f / (BM / 4), cq = (f % (BM / 4)) * 4;
            int64_t p = p0 + row;
            ra[i] = (p < p_end && m0 + cq < M) ? dyload(p, m0 + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
            bsum[i][0] += ra[i].x; bsum[i][1] += ra[i].y; bsum[i][2] += ra[i].z; bsum[i][3] += ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int f = t + i * NTHREADS, row = f / (BN / 4), cq = (f % (BN / 4)) * 4;
            int64_t p = p0 + row;
            rb[i] = (p < p_end && n0 + cq < N) ? xload(p, n0 + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int f = t + i * NTHREADS, row = f / (BM / 4), cq = (f % (BM / 4)) * 4;
            *reinterpret_cast<float4 *>(&As[row * (BM + 4) + cq]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int f = t + i * NTHREADS, row = f / (BN / 4), cq = (f % (BN / 4)) * 4;
            *reinterpret_cast<float4 *>(&Bs[row * (BN + 4) + cq]) = rb[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < WG_BP / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk * 2 + lh) * (BM + 4) + wr * WTM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk * 2 + lh) * (BN + 4) + wc * WTN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * WTN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < N) atomicAdd(dW + (int64_t)m * lddw + n, acc[i][j][r]);
            }
        }
    if (dbias != nullptr && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int f = t + i * NTHREADS, cq = (f % (BM / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (m0 + cq + e < M) atomicAdd(dbias + m0 + cq + e, bsum[i][e]);
        }
    }
}

template <int BM, int BN, int WR, int WC, class DyLoad, class XLoad>
int launch_tn(DyLoad dyload, XLoad xload, int64_t P, int M, int N, float *dW, int lddw, float *dbias, hipStream_t s) {
    unsigned tm = (unsigned)pn2_cdiv(M, BM), tn = (unsigned)pn2_cdiv(N, BN);
    int64_t want = 2048 / ((int64_t)tm * tn);
    if (want < 1) want = 1;
    int64_t max_split = pn2_cdiv(P, 8 * WG_BP);
    int64_t split = want < max_split ? want : max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    int64_t chunk = pn2_cdiv(pn2_cdiv(P, split), WG_BP) * WG_BP;
    split = pn2_cdiv(P, chunk);
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WR, WC, DyLoad, XLoad>), dim3(tm, tn, (unsigned)split), dim3(NTHREADS), 0, s,
                       dyload, xload, P, chunk, M, N, dW, lddw, dbias);
    return pn2_launch_status();
}

template <class DyLoad, class XLoad>
int dispatch_tn(DyLoad dyload, XLoad xload, int64_t P, int M, int N, float *dW, int lddw, float *dbias, hipStream_t s) {
    if (N <= 32) return launch_tn<128, 32, 4, 1>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (M <= 32) return launch_tn<32, 128, 1, 4>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (M <= 64 && N <= 64) return launch_tn<64, 64, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    return launch_tn<128, 128, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
}

// ----------------------------------------------------------------------------- small kernels

__global__ void bn_finalize_kernel(const double *__restrict__ stats, double inv_p, double unbias, int C, int ld,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float momentum, int training, float *__restrict__ rmean, float *__restrict__ rvar,
                                   int64_t *__restrict__ nbt, float *__restrict__ affine) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt) *nbt += 1;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = stats[c] * inv_p;
        var = stats[C + c] * inv_p - mean * mean;
        if (var < 0.0) var = 0.0;
        if (rmean) rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
        if (rvar) rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * var * unbias);
    } else {
        mean = (double)rmean[c];
        var = (double)rvar[c];
    }
    double invstd = 1.0 / sqrt(var + (double)eps);
    affine[c] = (float)mean;
    affine[ld + c] = (float)((double)gamma[c] * invstd);
    affine[2 * ld + c] = beta[c];
    affine[3 * ld + c] = (float)invstd;
}

// out[g,c] = max_k relu(bn(Y[g*K+k,c])); arg = first k attaining it.
__global__ __launch_bounds__(256) void bn_relu_max_kernel(const float *__restrict__ Y, int ldy,
                                                          const float *__restrict__ aff, int lda, int64_t G, int K,
                                                          int C, float *__restrict__ out, int ldo,
                                                          int32_t *__restrict__ arg) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t g = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= C || g >= G) return;
    Affine a(aff, lda);
    const float mu = a.mean[c], sc = a.scale[c], be = a.beta[c];
    const float *y = Y + g * K * ldy + c;
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        float v = fmaxf(bn_act(y[(int64_t)k * ldy], mu, sc, be), 0.f);
        if (v > best) { best = v; bk = k; }
    }
    out[g * ldo + c] = best;
    if (arg) arg[g * ldo + c] = bk;
}

// red[c] += sum_g dZ, red[C+c] += sum_g dZ*yhat at the pooled positions.
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float *__restrict__ dOut, int ldo,
                                                              const float *__restrict__ out,
                                                              const int32_t *__restrict__ arg,
                                                              const float *__restrict__ Y, int ldy,
                                                              const float *__restrict__ aff, int lda, int64_t G, int K,
                                                              int C, double *__restrict__ red) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        Affine a(aff, lda);
        const float mu = a.mean[c], is = a.invstd[c];
        for (int64_t g = (int64_t)blockIdx.y * 4 + gl; g < G; g += (int64_t)gridDim.y * 4) {
            float o = out[g * ldo + c];
            if (o > 0.f) {
                float dz = dOut[g * ldo + c];
                float y = Y[(g * K + arg[g * ldo + c]) * ldy + c];
                s0 += (double)dz;
                s1 += (double)(dz * ((y - mu) * is));
            }
        }
    }
    sh[0][gl][cl] = s0; sh[1][gl][cl] = s1;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        double a1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        atomicAdd(red + c, a0);
        atomicAdd(red + C + c, a1);
    }
}

// Dense last layer (FP): dZ = dOut * (out > 0), same two reductions.
__global__ __launch_bounds__(256) void relu_bwd_reduce_kernel(const float *__restrict__ dOut, int ldo,
                                                              const float *__restrict__ out,
                                                              const float *__restrict__ Y, int ldy,
                                                              const float *__restrict__ aff, int lda, int64_t P, int C,
                                                              float *__restrict__ dZ, int ldz, double *__restrict__ red) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        Affine a(aff, lda);
        const float mu = a.mean[c], is = a.invstd[c];
        for (int64_t p = (int64_t)blockIdx.y * 4 + gl; p < P; p += (int64_t)gridDim.y * 4) {
            float dz = out[p * ldo + c] > 0.f ? dOut[p * ldo + c] : 0.f;
            dZ[p * ldz + c] = dz;
            float y = Y[p * ldy + c];
            s0 += (double)dz;
            s1 += (double)(dz * ((y - mu) * is));
        }
    }
    sh[0][gl][cl] = s0; sh[1][gl][cl] = s1;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        double a1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        atomicAdd(red + c, a0);
        atomicAdd(red + C + c, a1);
    }
}

__global__ void bn_bwd_coef_kernel(const double *__restrict__ red, double inv_p, int C, int ld,
                                   const float *__restrict__ gamma, const float *__restrict__ aff, int use_batch,
                                   float *__restrict__ coef, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    Affine a(aff, ld);
    double c0 = (double)gamma[c] * (double)a.invstd[c];
    double r0 = red[c], r1 = red[C + c];
    coef[c] = (float)c0;
    coef[ld + c] = use_batch ? (float)(-c0 * (double)a.invstd[c] * r1 * inv_p) : 0.f;
    coef[2 * ld + c] = use_batch ? (float)(-c0 * r0 * inv_p) : 0.f;
    coef[3 * ld + c] = a.mean[c];
    if (dgamma) dgamma[c] = (float)r1;
    if (dbeta) dbeta[c] = (float)r0;
}

inline int round4(int x) { return (x + 3) & ~3; }

}  // namespace

extern "C" {

int pn2_conv1x1_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y,
                    int ldy, int64_t P, int K, int N, double *stats, pn2_stream_t stream) {
    PN2_CHECK_ARG(X && W && bias && Y && P > 0 && K > 0 && N > 0);
    PN2_CHECK_ARG(ldx % 4 == 0 && ldw % 4 == 0 && ldx >= round4(K) && ldw >= round4(K) && ldy >= N);
    const int K4 = round4(K);
    EpiFwd epi{Y, ldy, bias, stats};
    if (in_affine) return dispatch_nt(LoadBnRelu{X, ldx, in_affine}, W, ldw, P, K4, N, epi, pn2_s(stream));
    return dispatch_nt(LoadPlain{X, ldx}, W, ldw, P, K4, N, epi, pn2_s(stream));
}

int pn2_bn_finalize(const double *stats, int64_t P, int C, const float *gamma, const float *beta, float eps,
                    float momentum, int training, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                    float *affine, pn2_stream_t stream) {
    PN2_CHECK_ARG(gamma && beta && affine && C > 0 && P > 0);
    PN2_CHECK_ARG(training ? stats != nullptr : (running_mean && running_var));
    const int ld = round4(C);
    double unbias = P > 1 ? (double)P / (double)(P - 1) : 1.0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pn2_cdiv(C, 128)), dim3(128), 0, pn2_s(stream), stats, 1.0 / (double)P,
                       unbias, C, ld, gamma, beta, eps, momentum, training, running_mean, running_var,
                       num_batches_tracked, affine);
    return pn2_launch_status();
}

int pn2_bn_relu_max(const float *Y, int ldy, const float *affine, int64_t G, int K, int C, float *out, int ldo,
                    int32_t *arg, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && affine && out && G > 0 && K > 0 && C > 0 && ldy >= C && ldo >= C);
    int64_t gy = pn2_cdiv(G, 4);
    PN2_CHECK_ARG(gy <= 0x7fffffff);
    // grid.y is limited to 65535: fold the group index into x-major order when needed
    if (gy > 65535) {
        // split into slabs of 65535*4 groups
        int64_t done = 0;
        while (done < G) {
            int64_t take = G - done < 65535LL * 4 ? G - done : 65535LL * 4;
            hipLaunchKernelGGL(bn_relu_max_kernel, dim3((unsigned)pn2_cdiv(C, 64), (unsigned)pn2_cdiv(take, 4)), dim3(256), 0,
                               pn2_s(stream), Y + done * K * ldy, ldy, affine, (C + 3) & ~3, take, K, C, out + done * ldo, ldo,
                               arg ? arg + done * ldo : nullptr);
            done += take;
        }
        return pn2_launch_status();
    }
    hipLaunchKernelGGL(bn_relu_max_kernel, dim3((unsigned)pn2_cdiv(C, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), Y, ldy,
                       affine, (C + 3) & ~3, G, K, C, out, ldo, arg);
    return pn2_launch_status();
}

int pn2_pool_bwd_reduce(const float *dOut, int ldo, const float *out, const int32_t *arg, const float *Y, int ldy,
                        const float *affine, int64_t G, int K, int C, double *red, pn2_stream_t stream) {
    PN2_CHECK_ARG(dOut && out && arg && Y && affine && red && G > 0 && K > 0 && C > 0);
    int64_t gy = pn2_cdiv(G, 4 * 16);
    if (gy > 256) gy = 256;
    hipLaunchKernelGGL(pool_bwd_reduce_kernel, dim3((unsigned)pn2_cdiv(C, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), dOut,
                       ldo, out, arg, Y, ldy, affine, (C + 3) & ~3, G, K, C, red);
    return pn2_launch_status();
}

int pn2_relu_bwd_reduce(const float *dOut, int ldo, const float *out, const float *Y, int ldy, const float *affine,
                        int64_t P, int C, float *dZ, int ldz, double *red, pn2_stream_t stream) {
    PN2_CHECK_ARG(dOut && out && Y && affine && dZ && red && P > 0 && C > 0);
    int64_t gy = pn2_cdiv(P, 4 * 16);
    if (gy > 512) gy = 512;
    hipLaunchKernelGGL(relu_bwd_reduce_kernel, dim3((unsigned)pn2_cdiv(C, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), dOut,
                       ldo, out, Y, ldy, affine, (C + 3) & ~3, P, C, dZ, ldz, red);
    return pn2_launch_status();
}

int pn2_bn_bwd_coef(const double *red, int64_t P, int C, const float *gamma, const float *affine, int use_batch_stats,
                    float *coef, float *dgamma, float *dbeta, pn2_stream_t stream) {
    PN2_CHECK_ARG(red && gamma && affine && coef && P > 0 && C > 0);
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((unsigned)pn2_cdiv(C, 128)), dim3(128), 0, pn2_s(stream), red, 1.0 / (double)P,
                       C, (C + 3) & ~3, gamma, affine, use_batch_stats, coef, dgamma, dbeta);
    return pn2_launch_status();
}

int pn2_conv1x1_dgrad(const float *dZ, int ldz, const float *dOut, int ldo, const float *out, const int32_t *arg,
                      int Kpool, const float *Y, int ldy, const float *coef, const float *Wt, int ldw,
                      const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                      double *prev_red, int64_t P, int K, int N, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && coef && Wt && dXout && P > 0 && K > 0 && N > 0);
    PN2_CHECK_ARG(dZ != nullptr || (dOut && out && arg && Kpool > 0));
    PN2_CHECK_ARG(ldw % 4 == 0 && ldw >= round4(K) && ldy % 4 == 0 && ldy >= round4(K) && ldxo >= N);
    PN2_CHECK_ARG(prev_Y == nullptr || prev_affine != nullptr);
    const int K4 = round4(K), ldc = round4(K);
    hipStream_t s = pn2_s(stream);
    if (dZ) {
        PN2_CHECK_ARG(ldz % 4 == 0 && ldz >= K4);
        LoadDyDense ld{dZ, ldz, Y, ldy, coef, ldc};
        if (prev_Y)
            return dispatch_nt(ld, Wt, ldw, P, K4, N,
                               EpiDgradMask{dXout, ldxo, prev_Y, ld_prev, prev_affine, round4(N), prev_red}, s);
        return dispatch_nt(ld, Wt, ldw, P, K4, N, EpiStore{dXout, ldxo}, s);
    }
    PN2_CHECK_ARG(ldo % 4 == 0 && ldo >= K4);
    LoadDyPooled ld{dOut, ldo, out, arg, Kpool, Y, ldy, coef, ldc};
    if (prev_Y)
        return dispatch_nt(ld, Wt, ldw, P, K4, N, EpiDgradMask{dXout, ldxo, prev_Y, ld_prev, prev_affine, round4(N), prev_red},
                           s);
    return dispatch_nt(ld, Wt, ldw, P, K4, N, EpiStore{dXout, ldxo}, s);
}

int pn2_conv1x1_wgrad(const float *dZ, int ldz, const float *dOut, int ldo, const float *out, const int32_t *arg,
                      int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                      const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                      pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && coef && X && dW && P > 0 && M > 0 && N > 0);
    PN2_CHECK_ARG(dZ != nullptr || (dOut && out && arg && Kpool > 0));
    PN2_CHECK_ARG(ldy % 4 == 0 && ldy >= round4(M) && ldx % 4 == 0 && ldx >= round4(N) && lddw >= N);
    const int ldc = round4(M);
    hipStream_t s = pn2_s(stream);
    if (dZ) {
        PN2_CHECK_ARG(ldz % 4 == 0 && ldz >= round4(M));
        LoadDyDense dy{dZ, ldz, Y, ldy, coef, ldc};
        if (x_affine) return dispatch_tn(dy, LoadBnRelu{X, ldx, x_affine}, P, M, N, dW, lddw, dbias, s);
        return dispatch_tn(dy, LoadPlain{X, ldx}, P, M, N, dW, lddw, dbias, s);
    }
    PN2_CHECK_ARG(ldo % 4 == 0 && ldo >= round4(M));
    LoadDyPooled dy{dOut, ldo, out, arg, Kpool, Y, ldy, coef, ldc};
    if (x_affine) return dispatch_tn(dy, LoadBnRelu{X, ldx, x_affine}, P, M, N, dW, lddw, dbias, s);
    return dispatch_tn(dy, LoadPlain{X, ldx}, P, M, N, dW, lddw, dbias, s);
}

}  // extern "C"
