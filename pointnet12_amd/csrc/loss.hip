// Negative log-likelihood over log-probabilities: the loss either side of the hot path (reference
// semseg.py:143 `F.nll_loss(pred, target)`, pcdseg.py:179 the weighted form).
//     loss = - sum_r w[t_r] * logp[r, t_r] / sum_r w[t_r]      over rows with t_r != ignore_index
// ATen's own kernel for this reduction is a single workgroup (66 us forward + 37 us backward at 65 536 rows
// on MI355X, fully exposed between the forward and the backward pass); here every CU takes a slice, the
// per-workgroup partials are fp64 and the last workgroup to finish (ticket) adds them in a fixed order, so
// the result does not depend on the order the workgroups ran in.
#include "pn2_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void nll_fwd_kernel(const float *__restrict__ logp, int ld,
                                                           const int64_t *__restrict__ target,
                                                           const float *__restrict__ weight, int64_t R, int C,
                                                           int64_t ignore_index, double *__restrict__ ws,
                                                           unsigned *__restrict__ ticket, float *__restrict__ loss,
                                                           float *__restrict__ denom) {
    __shared__ double sh[2][kThreads / 64];
    __shared__ bool last;
    double num = 0.0, den = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < R; r += (int64_t)gridDim.x * kThreads) {
        const int64_t t = target[r];
        if (t == ignore_index) continue;
        if (t < 0 || t >= C) { num = __builtin_nan(""); continue; }      // ATen asserts; here the loss turns NaN
        const float w = weight ? weight[t] : 1.f;
        num -= (double)(w * logp[r * ld + t]);
        den += (double)w;
    }
    num = pn2_wave_sum_f64(num);
    den = pn2_wave_sum_f64(den);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = num; sh[1][wave] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double n = 0.0, d = 0.0;
        for (int i = 0; i < kThreads / 64; ++i) { n += sh[0][i]; d += sh[1][i]; }
        ws[blockIdx.x] = n;
        ws[gridDim.x + blockIdx.x] = d;
        __threadfence();                                   // partials visible device-wide before the ticket
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // <= 1024 partials: thread t adds partials t, t + 256, ... in index order, then a fixed-shape tree over the 256
    // threads -- the summation order is a function of gridDim only, never of which workgroup finished when.
    __shared__ double tree[2][kThreads];
    double n = 0.0, d = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += kThreads) {
        n += __builtin_nontemporal_load(ws + i);
        d += __builtin_nontemporal_load(ws + gridDim.x + i);
    }
    tree[0][threadIdx.x] = n;
    tree[1][threadIdx.x] = d;
    __syncthreads();
    for (int w = kThreads / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            tree[0][threadIdx.x] += tree[0][threadIdx.x + w];
            tree[1][threadIdx.x] += tree[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = (float)(tree[0][0] / tree[1][0]);          // 0/0 = NaN when every row is ignored, as ATen
        *denom = (float)tree[1][0];
        *ticket = 0;                                       // the workspace is reusable without another memset
    }
}

__global__ __launch_bounds__(kThreads) void nll_bwd_kernel(const int64_t *__restrict__ target,
                                                           const float *__restrict__ weight, int64_t R, int C,
                                                           int64_t ignore_index, const float *__restrict__ grad_loss,
                                                           const float *__restrict__ denom, float *__restrict__ dlogp,
                                                           int ld) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= R * ld) return;
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    const int64_t t = target[r];
    float v = 0.f;
    if (t == c && t != ignore_index) v = -(*grad_loss) * (weight ? weight[t] : 1.f) / *denom;
    dlogp[i] = v;
}

// log_softmax over the C leading columns of rows of pitch ldx (model/pointnet2.py:175, F.log_softmax(x, dim = -1) on the head's
// [B*N, classes] logits): one thread per row, the row read as float4 quads (a 13-class row of pitch 16 is one 64-byte line).
// ATen needs a copy (the logits arrive as a column slice of the padded GEMM output) plus its softmax kernel, and in the
// backward a zero fill plus a strided copy to re-pad the gradient; here both directions read and write the padded layout.
constexpr int kMaxClasses = 64;

__global__ __launch_bounds__(kThreads) void log_softmax_fwd_kernel(const float *__restrict__ x, int ldx, int64_t R, int C,
                                                                   float *__restrict__ out, int ldo) {
    const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (r >= R) return;
    const float *row = x + r * ldx;
    float v[kMaxClasses];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < kMaxClasses / 4; ++q) {
        if (4 * q >= C) break;
        const float4 t = *reinterpret_cast<const float4 *>(row + 4 * q);     // ldx >= round4(C): the quad is inside the row
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
        if (c < C) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
        if (c < C) s += expf(v[c] - m);
    const float lse = m + logf(s);
    float *o = out + r * ldo;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
        if (c < C) o[c] = v[c] - lse;
}

// gx = g - exp(out) * sum(g) on the C leading columns of a row of pitch ldgx; the pad columns are written as zeros (the GEMM
// backward that consumes gx reads whole float4 quads)
__global__ __launch_bounds__(kThreads) void log_softmax_bwd_kernel(const float *__restrict__ g, int ldg, const float *__restrict__ out,
                                                                   int ldo, int64_t R, int C, float *__restrict__ gx, int ldgx) {
    const int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (r >= R) return;
    const float *gr = g + r * ldg, *orow = out + r * ldo;
    float gv[kMaxClasses];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c)
        if (c < C) { gv[c] = gr[c]; s += gv[c]; }
    float *d = gx + r * ldgx;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
        if (c < C) d[c] = gv[c] - expf(orow[c]) * s;
        else if (c < ldgx) d[c] = 0.f;
    }
}

}  // namespace

extern "C" {

int pn2_log_softmax_fwd(const float *x, int ldx, int64_t R, int C, float *out, int ldo, pn2_stream_t stream) {
    PN2_CHECK_ARG(x && out && R > 0 && C > 0 && C <= kMaxClasses && ldx % 4 == 0 && ldx >= ((C + 3) & ~3) && ldo >= C);
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((unsigned)pn2_cdiv(R, kThreads)), dim3(kThreads), 0, pn2_s(stream), x, ldx, R, C, out, ldo);
    return pn2_launch_status();
}

int pn2_log_softmax_bwd(const float *grad_out, int ldg, const float *out, int ldo, int64_t R, int C, float *grad_x, int ldgx,
                        pn2_stream_t stream) {
    PN2_CHECK_ARG(grad_out && out && grad_x && R > 0 && C > 0 && C <= kMaxClasses && ldg >= C && ldo >= C && ldgx >= C && ldgx <= kMaxClasses);
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)pn2_cdiv(R, kThreads)), dim3(kThreads), 0, pn2_s(stream), grad_out, ldg, out, ldo,
                       R, C, grad_x, ldgx);
    return pn2_launch_status();
}

int64_t pn2_nll_loss_workspace_bytes(int64_t R) {
    (void)R;
    return (int64_t)(2 * 1024 * sizeof(double) + 16);
}

int pn2_nll_loss_fwd(const float *logp, int ld, const int64_t *target, const float *weight, int64_t R, int C,
                     int64_t ignore_index, void *workspace, float *loss, float *denom, pn2_stream_t stream) {
    PN2_CHECK_ARG(logp && target && workspace && loss && denom && R > 0 && C > 0 && ld >= C);
    int64_t blocks = pn2_cdiv(R, kThreads);
    if (blocks > 1024) blocks = 1024;
    double *ws = reinterpret_cast<double *>(workspace);
    unsigned *ticket = reinterpret_cast<unsigned *>(ws + 2 * 1024);
    hipLaunchKernelGGL(nll_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, pn2_s(stream), logp, ld, target, weight, R, C,
                       ignore_index, ws, ticket, loss, denom);
    return pn2_launch_status();
}

int pn2_nll_loss_bwd(const int64_t *target, const float *weight, int64_t R, int C, int64_t ignore_index,
                     const float *grad_loss, const float *denom, float *dlogp, int ld, pn2_stream_t stream) {
    PN2_CHECK_ARG(target && grad_loss && denom && dlogp && R > 0 && C > 0 && ld >= C);
    PN2_CHECK_ARG(R * ld < (1LL << 40));
    hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)pn2_cdiv(R * ld, kThreads)), dim3(kThreads), 0, pn2_s(stream), target,
                       weight, R, C, ignore_index, grad_loss, denom, dlogp, ld);
    return pn2_launch_status();
}

}  // extern "C"
