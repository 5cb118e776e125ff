// Row gathers / scatters around the shared MLP: index_points, grouping, 3-point interpolation.
// All of these are HBM/L2-bandwidth work: one lane per output element, consecutive lanes on
// consecutive channels of one source row so every wave moves contiguous row segments.
#include "pn2_common.h"

namespace {

constexpr int TPB = 256;

// index_points (pointnet_util.py:58-59): out[b,m,:] = points[b, idx[b,m], :]
__global__ __launch_bounds__(TPB) void gather_rows_kernel(const float *__restrict__ points,
                                                          const int64_t *__restrict__ idx, int N, int C, int M,
                                                          int64_t total, float *__restrict__ out,
                                                          int *__restrict__ err) {
    int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= total) return;
    int c = (int)(e % C);
    int64_t r = e / C;            // b*M + m
    int64_t b = r / M;
    int64_t j = idx[r];
    float v = 0.f;
    if (j >= 0 && j < N) v = points[(b * N + j) * C + c];
    else if (err) *err = 1;
    out[e] = v;
}

__global__ __launch_bounds__(TPB) void gather_rows_bwd_kernel(const float *__restrict__ grad_out,
                                                              const int64_t *__restrict__ idx, int N, int C, int M,
                                                              int64_t total, float *__restrict__ grad_points) {
    int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= total) return;
    int c = (int)(e % C);
    int64_t r = e / C;
    int64_t b = r / M;
    int64_t j = idx[r];
    if (j >= 0 && j < N) atomicAdd(grad_points + (b * N + j) * C + c, grad_out[e]);
}

// Grouping (pointnet_util.py:127-133 / :243-251): row p = (b,s,k) of the position-major
// matrix = [xyz[j]-centre, feat[j]] or [feat[j], xyz[j]-centre], zero padded to ld.
// One thread assembles one float4 of a row (ld is a multiple of 4) and stores it with a single 16-byte write; the
// index arithmetic is 32-bit (IT = unsigned) whenever the element count allows -- per-element 64-bit divisions
// made the first version ALU-bound at 0.3 TB/s on the 1 M-row groups of MSG sa1.
template <typename IT>
__global__ __launch_bounds__(TPB) void group_kernel(const float *__restrict__ xyz, const float *__restrict__ points,
                                                    const float *__restrict__ new_xyz,
                                                    const int64_t *__restrict__ idx, int N, int S, int K, int D,
                                                    int xyz_first, int ld4, int64_t total4, float *__restrict__ out,
                                                    int *__restrict__ err) {
    const int64_t e64 = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e64 >= total4) return;
    const IT e = (IT)e64;
    const IT p = e / (IT)ld4;           // (b*S + s)*K + k
    const int c0 = (int)(e - p * (IT)ld4) * 4;
    const IT g = p / (IT)K;             // b*S + s
    const IT b = g / (IT)S;
    const int64_t j = idx ? idx[p] : (int64_t)(p - g * (IT)K);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < 0 || j >= N) {
        if (err) *err = 1;
    } else {
        const float *px = xyz + ((int64_t)b * N + j) * 3;
        const float *pf = points ? points + ((int64_t)b * N + j) * D : nullptr;
        const float *ctr = new_xyz ? new_xyz + (int64_t)g * 3 : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i;
            if (c >= 3 + D) continue;
            const int cx = xyz_first ? c : c - D;          // position inside the xyz triple, if any
            if (cx >= 0 && cx < 3) {
                float t = px[cx];
                if (ctr) t = t - ctr[cx];
                v[i] = t;
            } else {
                v[i] = pf[xyz_first ? c - 3 : c];
            }
        }
    }
    *reinterpret_cast<float4 *>(out + (int64_t)e64 * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(TPB) void group_bwd_kernel(const float *__restrict__ grad_rows,
                                                        const int64_t *__restrict__ idx, int N, int S, int K, int D,
                                                        int xyz_first, int ld, int64_t total,
                                                        float *__restrict__ grad_points) {
    int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;   // over P * D feature elements
    if (e >= total) return;
    int cf = (int)(e % D);
    int64_t p = e / D;
    int64_t b = p / ((int64_t)S * K);
    int64_t j = idx ? idx[p] : (p % K);
    if (j < 0 || j >= N) return;
    float gval = grad_rows[p * ld + (xyz_first ? 3 + cf : cf)];
    atomicAdd(grad_points + (b * N + j) * D + cf, gval);
}

// pointnet_util.py:301: interpolated[b,n,c] = ((p2[i0,c]*w0 + p2[i1,c]*w1) + p2[i2,c]*w2), written at column col0 + c of
// the concatenated row; with points1 != NULL the same launch also copies points1[b,n,0..col0) in front of it (:305,
// cat([points1, interpolated], -1)) -- one kernel per FeaturePropagation input instead of two.
// One WAVE per output row (its three neighbour indices and weights are wave-uniform), lanes across the columns: coalesced
// dword accesses whatever col0 is, no per-element division (the element-per-thread form spent 39 us on the 37 MB of FP1's
// input at B = 16 x 4096: two 64-bit divisions and six broadcast loads per element).
__global__ __launch_bounds__(256) void three_interp_kernel(const float *__restrict__ points2,
                                                           const int64_t *__restrict__ idx,
                                                           const float *__restrict__ w, int N, int S, int D,
                                                           int rows, float *__restrict__ out, int ld, int col0,
                                                           int zero_tail, const float *__restrict__ points1) {
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nwaves) {
        const int b = r / N;
        const int64_t *i3 = idx + (int64_t)r * 3;
        const float *w3 = w + (int64_t)r * 3;
        const float w0 = w3[0], w1 = w3[1], w2 = w3[2];
        const float *base = points2 + (int64_t)b * S * D;
        const float *p0 = base + i3[0] * D, *p1 = base + i3[1] * D, *p2 = base + i3[2] * D;
        float *o = out + (int64_t)r * ld;
        if (points1)
            for (int c = lane; c < col0; c += 64) o[c] = points1[(int64_t)r * col0 + c];
        for (int c = lane; c < D; c += 64)
            o[col0 + c] = __fadd_rn(__fadd_rn(__fmul_rn(p0[c], w0), __fmul_rn(p1[c], w1)), __fmul_rn(p2[c], w2));
        if (zero_tail)
            for (int c = col0 + D + lane; c < ld; c += 64) o[c] = 0.f;
    }
}

__global__ __launch_bounds__(TPB) void three_interp_bwd_kernel(const float *__restrict__ grad_out, int ld, int col0,
                                                               const int64_t *__restrict__ idx,
                                                               const float *__restrict__ w, int N, int S, int D,
                                                               int64_t total, float *__restrict__ grad_points2) {
    int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= total) return;
    int c = (int)(e % D);
    int64_t r = e / D;
    int64_t b = r / N;
    const int64_t *i3 = idx + r * 3;
    const float *w3 = w + r * 3;
    float g = grad_out[r * ld + col0 + c];
    float *base = grad_points2 + b * S * D + c;
    atomicAdd(base + i3[0] * D, g * w3[0]);
    atomicAdd(base + i3[1] * D, g * w3[1]);
    atomicAdd(base + i3[2] * D, g * w3[2]);
}

// S == 1 (FeaturePropagation below a group_all stage, reference pointnet_util.py:292-293: the single row is repeated N times): every
// entry of idx is 0, so the scatter is a column sum -- grad_points2[b, 0, c] = sum_n (w0 + w1 + w2)[b, n] * grad_out[b, n, c].
// One workgroup per (cloud, 64 columns): four row lanes per column, rows in a fixed order, the four partials added in a fixed
// order; stored, not added (3 N same-address atomics per channel otherwise: 51 us at B = 16, N = 128, D = 1024).
__global__ __launch_bounds__(256) void three_interp_bwd_single_kernel(const float *__restrict__ grad_out, int ld, int col0,
                                                                      const float *__restrict__ w, int N, int D,
                                                                      float *__restrict__ grad_points2) {
    __shared__ float part[4][64];
    const int b = blockIdx.x, cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float acc = 0.f;
    if (c < D) {
        for (int n = rl; n < N; n += 4) {
            const int64_t r = (int64_t)b * N + n;
            const float *w3 = w + r * 3;
            const float g = grad_out[r * ld + col0 + c];
            acc += g * w3[0];
            acc += g * w3[1];
            acc += g * w3[2];
        }
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < D) grad_points2[(int64_t)b * D + c] = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
}

__global__ __launch_bounds__(TPB) void copy_cols_kernel(const float *__restrict__ src, int lds, int scol0,
                                                        float *__restrict__ dst, int ldd, int dcol0, int cols,
                                                        int64_t total) {
    int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= total) return;
    int c = (int)(e % cols);
    int64_t r = e / cols;
    dst[r * ldd + dcol0 + c] = src[r * lds + scol0 + c];
}

inline unsigned blocks_for(int64_t total) { return (unsigned)pn2_cdiv(total, TPB); }

}  // namespace

extern "C" {

int pn2_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M, float *out, int *err,
                    pn2_stream_t stream) {
    PN2_CHECK_ARG(points && idx && out && B > 0 && N > 0 && C > 0 && M > 0);
    int64_t total = (int64_t)B * M * C;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(total)), dim3(TPB), 0, pn2_s(stream), points, idx, N, C, M,
                       total, out, err);
    return pn2_launch_status();
}

int pn2_gather_rows_bwd(const float *grad_out, const int64_t *idx, int B, int N, int C, int M, float *grad_points,
                        pn2_stream_t stream) {
    PN2_CHECK_ARG(grad_out && idx && grad_points && B > 0 && N > 0 && C > 0 && M > 0);
    int64_t total = (int64_t)B * M * C;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(blocks_for(total)), dim3(TPB), 0, pn2_s(stream), grad_out, idx, N, C,
                       M, total, grad_points);
    return pn2_launch_status();
}

int pn2_group(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S,
              int K, int D, int xyz_first, int ld, float *out, int *err, pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz && out && B > 0 && N > 0 && S > 0 && K > 0 && D >= 0 && ld >= 3 + D && ld % 4 == 0);
    PN2_CHECK_ARG(D == 0 || points != nullptr);
    PN2_CHECK_ARG(idx != nullptr || K == N);
    const int64_t total4 = (int64_t)B * S * K * (ld / 4);
    if (total4 < (1LL << 32))
        hipLaunchKernelGGL(group_kernel<unsigned>, dim3(blocks_for(total4)), dim3(TPB), 0, pn2_s(stream), xyz, points, new_xyz,
                           idx, N, S, K, D, xyz_first, ld / 4, total4, out, err);
    else
        hipLaunchKernelGGL(group_kernel<uint64_t>, dim3(blocks_for(total4)), dim3(TPB), 0, pn2_s(stream), xyz, points, new_xyz,
                           idx, N, S, K, D, xyz_first, ld / 4, total4, out, err);
    return pn2_launch_status();
}

int pn2_group_bwd(const float *grad_rows, const int64_t *idx, int B, int N, int S, int K, int D, int xyz_first, int ld,
                  float *grad_points, pn2_stream_t stream) {
    PN2_CHECK_ARG(grad_rows && grad_points && B > 0 && N > 0 && S > 0 && K > 0 && D > 0 && ld >= 3 + D);
    PN2_CHECK_ARG(idx != nullptr || K == N);
    int64_t total = (int64_t)B * S * K * D;
    hipLaunchKernelGGL(group_bwd_kernel, dim3(blocks_for(total)), dim3(TPB), 0, pn2_s(stream), grad_rows, idx, N, S, K, D,
                       xyz_first, ld, total, grad_points);
    return pn2_launch_status();
}

int pn2_three_interp(const float *points2, const int64_t *idx, const float *weight, int B, int N, int S, int D,
                     float *out, int ld, int col0, int zero_tail, const float *points1, pn2_stream_t stream) {
    PN2_CHECK_ARG(points2 && idx && weight && out && B > 0 && N > 0 && S > 0 && D > 0 && col0 >= 0 && ld >= col0 + D);
    PN2_CHECK_ARG(points1 == nullptr || col0 > 0);
    PN2_CHECK_ARG((int64_t)B * N < (1LL << 31));
    const int rows = B * N;
    int64_t blocks = ((int64_t)rows + 3) / 4;
    const int64_t cap = (int64_t)pn2_num_cus() * 8;          // 32 waves per CU, each walking its rows
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(three_interp_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), points2, idx, weight, N,
                       S, D, rows, out, ld, col0, zero_tail, points1);
    return pn2_launch_status();
}

int pn2_three_interp_bwd(const float *grad_out, int ld, int col0, const int64_t *idx, const float *weight, int B, int N,
                         int S, int D, float *grad_points2, pn2_stream_t stream) {
    PN2_CHECK_ARG(grad_out && idx && weight && grad_points2 && B > 0 && N > 0 && S > 0 && D > 0 && col0 >= 0 &&
                  ld >= col0 + D);
    if (S == 1) {                                      // (idx is all zeros by construction: it is not read)
        PN2_CHECK_ARG(B <= 65535 * 32);
        hipLaunchKernelGGL(three_interp_bwd_single_kernel, dim3(B, (D + 63) / 64), dim3(256), 0, pn2_s(stream), grad_out, ld, col0,
                           weight, N, D, grad_points2);
        return pn2_launch_status();
    }
    int64_t total = (int64_t)B * N * D;
    hipLaunchKernelGGL(three_interp_bwd_kernel, dim3(blocks_for(total)), dim3(TPB), 0, pn2_s(stream), grad_out, ld, col0,
                       idx, weight, N, S, D, total, grad_points2);
    return pn2_launch_status();
}

int pn2_copy_cols(const float *src, int lds, int scol0, float *dst, int ldd, int dcol0, int64_t rows, int cols,
                  pn2_stream_t stream) {
    PN2_CHECK_ARG(src && dst && rows > 0 && cols > 0 && scol0 >= 0 && dcol0 >= 0 && lds >= scol0 + cols &&
                  ldd >= dcol0 + cols);
    int64_t total = rows * cols;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks_for(total)), dim3(TPB), 0, pn2_s(stream), src, lds, scol0, dst, ldd,
                       dcol0, cols, total);
    return pn2_launch_status();
}

}  // extern "C"
