// Factorised first layer of a set-abstraction MLP.
//
// The grouped input row of position p = (b, s, k) is [f_j, xyz_j - c_s] with j = idx[p], so the first
// 1x1 convolution splits into a per-SOURCE-POINT part and a 3-term geometric part:
//     y[p, :] = (W_f f_j + bias)  +  W_x (xyz_j - c_s)  =  Zf[b, j, :] + W_x (xyz_j - c_s)
// Zf is one small GEMM over the B*N source points instead of the B*S*K grouped positions (K = 32..128
// times fewer rows), the grouped tensor [P, 3+D] is never materialised, and the forward kernel is a row
// gather of Zf plus three FMAs per output with the exactly centred coordinates (the subtraction happens
// before the multiply, as in the reference, pointnet_util.py:128,244 -- no cancellation).
// Replaces, for the first layer only: index_points + cat (pointnet_util.py:127-131, :243-247) and
// nn.Conv2d (pointnet_util.py:197, :254), and their autograd.
#include "pn2_common.h"
#include "bn_tail.h"

namespace {

// Per-channel sum(y), sum(y*y) for the training-mode BatchNorm are accumulated like in the GEMM epilogue:
// fp32 per 64 rows -> fp64 registers -> LDS -> one fp64 atomic per channel per workgroup.
__global__ __launch_bounds__(256) void group_affine_fwd_kernel(const float *__restrict__ Zf, int ldz,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const int64_t *__restrict__ idx,
                                                               const float *__restrict__ Wx, int ldwx, int N, int S,
                                                               int K, int C, int64_t P, float *__restrict__ Y, int ldy,
                                                               double *__restrict__ stats, FinTail fin) {
    __shared__ double red[256 * 8];
    const int CG = (C + 3) >> 2;                  // float4 column groups per row
    const int RPB = 256 / CG;                     // rows per pass
    const int t = threadIdx.x, cg = t % CG, r = t / CG;
    const bool live = r < RPB;
    const int c = cg * 4;
    float wx[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 3; ++a) wx[e][a] = (live && c + e < C) ? Wx[(c + e) * ldwx + a] : 0.f;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int since = 0;
    if (live) {
        // four independent rows per trip: idx -> (xyz, centre, Zf row) is a two-deep dependent chain per row; with
        // the four chains interleaved a trip costs two memory round trips instead of eight
        const int64_t stride = (int64_t)gridDim.x * RPB;
        for (int64_t p0 = (int64_t)blockIdx.x * RPB + r; p0 < P; p0 += 4 * stride) {
            int64_t j[4], g[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                const bool v = p < P;
                g[u] = v ? p / K : 0;
                b[u] = g[u] / S;
                j[u] = v ? idx[p] : 0;
            }
            float4 z[4];
            float d[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *q = xyz + (b[u] * N + j[u]) * 3, *ctr = new_xyz + g[u] * 3;
                d[u][0] = q[0] - ctr[0]; d[u][1] = q[1] - ctr[1]; d[u][2] = q[2] - ctr[2];
                z[u] = *reinterpret_cast<const float4 *>(Zf + (b[u] * N + j[u]) * ldz + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                if (p >= P) break;
                float y[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = __builtin_fmaf(wx[e][0], d[u][0], y[e]);
                    y[e] = __builtin_fmaf(wx[e][1], d[u][1], y[e]);
                    y[e] = __builtin_fmaf(wx[e][2], d[u][2], y[e]);
                    if (c + e >= C) y[e] = 0.f;
                    s0[e] += y[e];
                    s1[e] = __builtin_fmaf(y[e], y[e], s1[e]);
                }
                *reinterpret_cast<float4 *>(Y + p * ldy + c) = make_float4(y[0], y[1], y[2], y[3]);
                if (++since == 64) {              // fold the fp32 partials into fp64 every 64 rows
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st[e] += (double)s0[e]; st[4 + e] += (double)s1[e];
                        s0[e] = 0.f; s1[e] = 0.f;
                    }
                    since = 0;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[e] += (double)s0[e]; st[4 + e] += (double)s1[e]; }
    }
    if (stats == nullptr) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[t * 8 + e] = st[e];
    __syncthreads();
    for (int ch = t; ch < C; ch += 256) {
        const int g4 = ch >> 2, e = ch & 3;
        double a0 = 0.0, a1 = 0.0;
        for (int rr = 0; rr < RPB; ++rr) {
            a0 += red[(rr * CG + g4) * 8 + e];
            a1 += red[(rr * CG + g4) * 8 + 4 + e];
        }
        double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * C;
        atomicAdd(rep + ch, a0);
        atomicAdd(rep + C + ch, a1);
    }
    if (fin.ticket != nullptr && tail_is_last_block(fin.ticket, gridDim.x)) run_fin_tail(fin, stats, C, 256);
}

// Backward: dY = c0*dZ + q1*(y-mean) + q0 (BatchNorm backward folded into `coef`, see mlp.hip) is
// scattered back to the source points, G[b, j, :] += dY[p, :], and contracted with the centred
// coordinates, dWx[c, a] += dY[p, c] * (xyz_j - c_s)[a].  The weight / feature gradients then come from
// two small GEMMs over the B*N source points (dW_f = G^T F, dF = G W_f).
__global__ __launch_bounds__(256) void group_affine_bwd_kernel(const float *__restrict__ dZ, int ldz,
                                                               const float *__restrict__ Y, int ldy,
                                                               const float *__restrict__ coef, int ldc,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const int64_t *__restrict__ idx, int N, int S, int K,
                                                               int C, int64_t P, float *__restrict__ G, int ldg,
                                                               float *__restrict__ dWx, int ldwx) {
    // One lane per CHANNEL (not per float4): every atomic wave-instruction then adds 64 consecutive floats of one
    // G row -- the 256-byte contiguous shape the memory-side float atomics run at full rate with; the float4
    // mapping issued four 16-byte-strided instructions per row segment and ran ~4x slower.
    __shared__ float red[256 * 3];
    const int t = threadIdx.x;
    const int CT = C < 256 ? C : 256;             // threads per row
    const int RPB = 256 / CT;                     // rows per pass (1 when C >= 129)
    const int r = t / CT, cl = t % CT;
    const bool live = r < RPB;
    for (int cb = 0; cb < C; cb += CT) {          // C > 256: a thread owns channels cl, cl + 256, ... (uniform trip count)
        const int c = cb + cl;
        float acc[3] = {0.f, 0.f, 0.f};
        if (live && c < C) {
            const float c0 = coef[c], q1 = coef[ldc + c], q0 = coef[2 * ldc + c], mu = coef[3 * ldc + c];
            // four independent rows per trip: the idx -> xyz -> atomic chains of a lane overlap instead of queueing
            const int64_t stride = (int64_t)gridDim.x * RPB;
            for (int64_t p0 = (int64_t)blockIdx.x * RPB + r; p0 < P; p0 += 4 * stride) {
                int64_t j[4], g[4];
                float dzv[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t p = p0 + u * stride;
                    const bool v = p < P;
                    g[u] = v ? p / K : 0;
                    j[u] = v ? idx[p] : 0;
                    dzv[u] = v ? dZ[p * ldz + c] : 0.f;
                    yv[u] = v ? Y[p * ldy + c] : mu;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p0 + u * stride >= P) break;
                    const int64_t b = g[u] / S;
                    const float *q = xyz + (b * N + j[u]) * 3, *ctr = new_xyz + g[u] * 3;
                    const float dy = __builtin_fmaf(c0, dzv[u], __builtin_fmaf(q1, yv[u] - mu, q0));
                    atomicAdd(G + (b * N + j[u]) * ldg + c, dy);
                    acc[0] = __builtin_fmaf(dy, q[0] - ctr[0], acc[0]);
                    acc[1] = __builtin_fmaf(dy, q[1] - ctr[1], acc[1]);
                    acc[2] = __builtin_fmaf(dy, q[2] - ctr[2], acc[2]);
                }
            }
        }
        red[t * 3] = acc[0]; red[t * 3 + 1] = acc[1]; red[t * 3 + 2] = acc[2];
        __syncthreads();
        if (t < CT && c < C) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int rr = 0; rr < RPB; ++rr) {
                s0 += red[(rr * CT + t) * 3]; s1 += red[(rr * CT + t) * 3 + 1]; s2 += red[(rr * CT + t) * 3 + 2];
            }
            atomicAdd(dWx + c * ldwx, s0);
            atomicAdd(dWx + c * ldwx + 1, s1);
            atomicAdd(dWx + c * ldwx + 2, s2);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int pn2_group_affine_fwd(const float *Zf, int ldz, const float *xyz, const float *new_xyz, const int64_t *idx,
                         const float *Wx, int ldwx, int B, int N, int S, int K, int C, float *Y, int ldy, double *stats,
                         const pn2_bn_finalize_tail *fin, pn2_stream_t stream) {
    PN2_CHECK_ARG(fin == nullptr || (stats && fin->ticket && fin->gamma && fin->beta && fin->affine));
    FinTail ft{};
    if (fin) {
        const int64_t rows = (int64_t)B * S * K;
        ft = FinTail{fin->ticket, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean, fin->running_var,
                     fin->num_batches_tracked, fin->affine, 1.0 / (double)rows,
                     rows > 1 ? (double)rows / (double)(rows - 1) : 1.0};
    }
    PN2_CHECK_ARG(Zf && xyz && new_xyz && idx && Wx && Y && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 && C <= 1024 &&
                  ldwx >= 3);
    PN2_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && ldz >= ((C + 3) & ~3) && ldy >= ((C + 3) & ~3));
    const int64_t P = (int64_t)B * S * K;
    const int rpb = 256 / ((C + 3) >> 2);
    int64_t blocks = pn2_cdiv(P, (int64_t)rpb * 8);
    if (blocks > 1024) blocks = 1024;    // 2*C same-address fp64 atomics per workgroup at the end
    hipLaunchKernelGGL(group_affine_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), Zf, ldz, xyz, new_xyz, idx,
                       Wx, ldwx, N, S, K, C, P, Y, ldy, stats, ft);
    return pn2_launch_status();
}

int pn2_group_affine_bwd(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *xyz,
                         const float *new_xyz, const int64_t *idx, int B, int N, int S, int K, int C, float *G, int ldg,
                         float *dWx, int ldwx, pn2_stream_t stream) {
    PN2_CHECK_ARG(dZ && Y && coef && xyz && new_xyz && idx && G && dWx && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 &&
                  C <= 1024);
    PN2_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && ldg % 4 == 0 && ldwx >= 3);
    const int64_t P = (int64_t)B * S * K;
    const int rpb = C < 256 ? 256 / C : 1;
    int64_t blocks = pn2_cdiv(P, (int64_t)rpb * 16);
    if (blocks > 768) blocks = 768;      // every workgroup ends with 3*C same-address atomics on dWx: keep that queue short
    hipLaunchKernelGGL(group_affine_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), dZ, ldz, Y, ldy, coef,
                       (C + 3) & ~3, xyz, new_xyz, idx, N, S, K, C, P, G, ldg, dWx, ldwx);
    return pn2_launch_status();
}

}  // extern "C"
