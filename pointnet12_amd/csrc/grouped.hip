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


// ---------------------------------------------------------------------------------------------------------------------
// Gather + first layer in one launch, for the narrow first layers (3 + D <= 12 input channels, 32 or 64 outputs: the sa1
// stacks).  The stand-alone pair was pn2_group (a gather whose time is the latency of 1 M x 3 dependent row reads: 87 us
// for 50 MB) followed by a GEMM that is pure store bandwidth (268 MB of Y at 1 M x 64): here a lane gathers one grouped
// row (index, centre, xyz, features), keeps it in registers, writes it out once (the backward's weight gradient reads it),
// evaluates the CO outputs as an fma chain against the weight image in LDS, and the 64 x CO tile leaves through LDS as
// 16-byte stores, 256 contiguous bytes per row; the BatchNorm statistics are column sums of the same tile.
// Measured (1 M rows x 64): 102 us alone -- the pair it replaces took 26 + 73 -- of which the stores are 14 and the gather
// nothing (sequential indices: 103): the kernel is bound by the return path of its 192 broadcast ds_read_b128 of weights
// per slab (a broadcast still delivers 64 x 16 bytes).  Inside the step, where the three sa1 branches overlap, one launch
// instead of two is worth 50 us (MSG 6.50 -> 6.45 ms).  An MFMA form (the weight-resident forward with a gathering loader)
// would run at the store bound, ~75 us.  model/pointnet_util.py:127-131 + :197 (first conv), :243-247 + :254.
// ---------------------------------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void group_conv_fwd_kernel(const float *__restrict__ xyz, const float *__restrict__ points,
                                                             const float *__restrict__ new_xyz, const int64_t *__restrict__ idx,
                                                             int N, int S, int K, int D, int xyz_first,
                                                             const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                             float *__restrict__ X, int ldx, float *__restrict__ Y, int ldy,
                                                             int64_t slabs, double *__restrict__ stats) {
    constexpr int LT = CO + 4;                                     // tile pitch: 4 mod 8 floats (conflict-free b128 rows)
    __shared__ __attribute__((aligned(16))) float Wl[CO * 12];    // [CO][12]: the row's 3 + D coefficients, zero padded
    __shared__ __attribute__((aligned(16))) float bl[CO];         // bias (from LDS: a scalar load per channel would drain lgkmcnt)
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cin = 3 + D;
    for (int i = t; i < CO * 12; i += 256) {
        const int c = i / 12, k = i - c * 12;
        Wl[i] = k < cin ? W[(size_t)c * ldw + k] : 0.f;
    }
    for (int i = t; i < CO; i += 256) bl[i] = bias[i];
    float *T = tile[wave];
    double st0 = 0.0, st1 = 0.0;                                   // lane = channel (CO = 32: two row halves per channel)
    __syncthreads();
    const int xo = xyz_first ? 0 : D, fo = xyz_first ? 3 : 0;
    // Every wave works on slabs of its own (its own LDS tile: no workgroup barrier in the loop -- LDS operations of one wave
    // execute in order), and the NEXT slab's index / centre / point reads are in flight while this one is computed and stored.
    const int64_t stride = (int64_t)gridDim.x * 4;
    float gx, gy, gz, f[9];
    auto fetch = [&](int64_t slab) {
        const int64_t r = (slab < slabs ? slab : slabs - 1) * 64 + lane;      // grouped row (past the end: a valid dummy)
        const int64_t g = r / K, bb = g / S;
        const int64_t j = idx ? idx[r] : r - g * K;
        const float *px = xyz + (bb * N + j) * 3;
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (new_xyz) { const float *q = new_xyz + g * 3; cx = q[0]; cy = q[1]; cz = q[2]; }
        gx = px[0] - cx; gy = px[1] - cy; gz = px[2] - cz;
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = 0.f;
        if (D > 0) {
            const float *pf = points + (bb * N + j) * D;
#pragma unroll
            for (int k = 0; k < 9; ++k) f[k] = k < D ? pf[k] : 0.f;
        }
    };
    int64_t slab = (int64_t)blockIdx.x * 4 + wave;
    fetch(slab);
    for (; slab < slabs; slab += stride) {
        float x[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float v = 0.f;
            v = k == xo ? gx : v; v = k == xo + 1 ? gy : v; v = k == xo + 2 ? gz : v;
#pragma unroll
            for (int e = 0; e < 9; ++e) v = (k == fo + e && e < D) ? f[e] : v;
            x[k] = v;
        }
        fetch(slab + stride);
        typedef float v4f __attribute__((ext_vector_type(4)));
        {
            float *xr = X + (slab * 64 + lane) * ldx;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (4 * q < ldx) {
                    const v4f v = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(xr + 4 * q));
                }
            }
        }
        // y[c] = (fma chain over k, from 0) + bias[c]: the rounding order of the MFMA path it replaces
#pragma unroll 2
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float acc[4];
            const float4 b4 = *reinterpret_cast<const float4 *>(&bl[4 * c4]);
            const float bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 w0 = *reinterpret_cast<const float4 *>(&Wl[(4 * c4 + e) * 12]);
                const float4 w1 = *reinterpret_cast<const float4 *>(&Wl[(4 * c4 + e) * 12 + 4]);
                const float4 w2 = *reinterpret_cast<const float4 *>(&Wl[(4 * c4 + e) * 12 + 8]);
                float a = 0.f;
                a = __builtin_fmaf(x[0], w0.x, a); a = __builtin_fmaf(x[1], w0.y, a); a = __builtin_fmaf(x[2], w0.z, a);
                a = __builtin_fmaf(x[3], w0.w, a); a = __builtin_fmaf(x[4], w1.x, a); a = __builtin_fmaf(x[5], w1.y, a);
                a = __builtin_fmaf(x[6], w1.z, a); a = __builtin_fmaf(x[7], w1.w, a); a = __builtin_fmaf(x[8], w2.x, a);
                a = __builtin_fmaf(x[9], w2.y, a); a = __builtin_fmaf(x[10], w2.z, a); a = __builtin_fmaf(x[11], w2.w, a);
                acc[e] = a + bq[e];
            }
            *reinterpret_cast<float4 *>(&T[lane * LT + 4 * c4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __builtin_amdgcn_wave_barrier();                           // the tile is this wave's own: program order is enough
        {
            float *yb = Y + slab * 64 * (int64_t)ldy;
            constexpr int QPR = CO / 4;                            // 16-byte pieces per row
#pragma unroll
            for (int i = 0; i < QPR; ++i) {
                const int q = lane + 64 * i, row = q / QPR, quad = q - row * QPR;
                const float4 v = *reinterpret_cast<const float4 *>(&T[row * LT + 4 * quad]);
                const v4f vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<v4f *>(yb + (int64_t)row * ldy + 4 * quad));
            }
            if (stats != nullptr) {
                const int c = lane & (CO - 1), r0 = CO == 64 ? 0 : 32 * (lane >> 5), nr = CO == 64 ? 64 : 32;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
                for (int rr = 0; rr < nr; ++rr) {
                    const float y = T[(r0 + rr) * LT + c];
                    s0 += y;
                    s1 = __builtin_fmaf(y, y, s1);
                }
                st0 += (double)s0;
                st1 += (double)s1;
            }
        }
        __builtin_amdgcn_wave_barrier();                           // ... before the next slab overwrites it
    }
    if (stats != nullptr) {
        if (CO == 32) { st0 += __shfl_xor(st0, 32, 64); st1 += __shfl_xor(st1, 32, 64); }
        __shared__ double red[4][64][2];
        red[wave][lane][0] = st0; red[wave][lane][1] = st1;
        __syncthreads();
        if (t < CO) {
            double a0 = 0.0, a1 = 0.0;
            for (int w = 0; w < 4; ++w) { a0 += red[w][t][0]; a1 += red[w][t][1]; }
            double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * CO;
            atomicAdd(rep + t, a0);
            atomicAdd(rep + CO + t, a1);
        }
    }
}

}  // namespace

extern "C" {

int pn2_group_conv_fwd(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S,
                       int K, int D, int xyz_first, const float *W, int ldw, const float *bias, float *X, int ldx, float *Y, int ldy,
                       int C_out, double *stats, pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz && W && bias && X && Y && B > 0 && N > 0 && S > 0 && K > 0 && D >= 0 && (D == 0 || points) && (idx || K == N));
    PN2_CHECK_ARG(ldw >= 3 + D && ldx % 4 == 0 && ldx >= ((3 + D + 3) & ~3) && ldy % 4 == 0 && ldy >= C_out);
    const int64_t P = (int64_t)B * S * K;
    if (3 + D > 12 || D > 9 || ldx > 12 || !(C_out == 32 || C_out == 64) || P % 64 != 0 || P >= (1LL << 31)) return PN2_EUNSUPPORTED;
    const int64_t slabs = P / 64;
    int64_t grid = pn2_cdiv(slabs, 4);
    const int64_t cap = (int64_t)pn2_num_cus() * 2;                // persistent: two workgroups per CU (LDS: 4 tiles + W each)
    if (grid > cap) grid = cap;
    if (C_out == 64)
        hipLaunchKernelGGL(group_conv_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, pn2_s(stream), xyz, points, new_xyz, idx, N, S, K,
                           D, xyz_first, W, ldw, bias, X, ldx, Y, ldy, slabs, stats);
    else
        hipLaunchKernelGGL(group_conv_fwd_kernel<32>, dim3((unsigned)grid), dim3(256), 0, pn2_s(stream), xyz, points, new_xyz, idx, N, S, K,
                           D, xyz_first, W, ldw, bias, X, ldx, Y, ldy, slabs, stats);
    return pn2_launch_status();
}

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
