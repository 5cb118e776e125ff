// The steps either side of the hot path that the reference runs on the host or as a swarm of small ATen
// launches:
//   * pn2_adam_step       torch.optim.Adam(lr, betas, eps, weight_decay) as semseg.py:106-111 / pcdseg.py:133-138
//                         build it, over ONE flat fp32 parameter buffer: 28 B/element, a single launch instead of
//                         ~10 foreach launches over 150 tensors.
//   * pn2_prepare_clouds  pcd_normalize + pcd_jitter + the with-replacement resampling of
//                         data_utils/SemKITTI_Loader.py:17-30,93-113, as one gather over the raw [M,4] scans.
// Both are HBM-bound streaming kernels: float4 lanes, nothing staged.
#include "pn2_common.h"

namespace {

constexpr int kThreads = 256;

struct AdamScalars {
    float step_size, bc2_sqrt, beta1_w, beta2, one_m_beta2, eps, wd;
};

// Scalars exactly as torch/optim/adam.py::_single_tensor_adam forms them (Python doubles, rounded to fp32
// where ATen hands them to a float kernel).
__device__ __host__ inline AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double wd,
                                                    int64_t t) {
    const double bc1 = 1.0 - pow(beta1, (double)t);
    const double bc2 = 1.0 - pow(beta2, (double)t);
    AdamScalars s;
    s.step_size = (float)(lr / bc1);
    s.bc2_sqrt = (float)sqrt(bc2);
    s.beta1_w = (float)(1.0 - beta1);
    s.beta2 = (float)beta2;
    s.one_m_beta2 = (float)(1.0 - beta2);
    s.eps = (float)eps;
    s.wd = (float)wd;
    return s;
}

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, const AdamScalars &s) {
    if (s.wd != 0.f) g = g + s.wd * p;                       // grad.add(param, alpha=weight_decay)
    // exp_avg.lerp_(grad, 1 - beta1): ATen's lerp formula (weight < 0.5 branch first)
    const float d = g - m;
    m = s.beta1_w < 0.5f ? m + s.beta1_w * d : g - d * (1.f - s.beta1_w);
    v = v * s.beta2 + s.one_m_beta2 * g * g;                 // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
    p = p - s.step_size * (m / denom);                       // addcdiv_(exp_avg, denom, value=-step_size)
}

template <bool kVec>
__global__ __launch_bounds__(kThreads) void adam_kernel(float *__restrict__ param, float *__restrict__ grad,
                                                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                        int64_t n, double lr, double beta1, double beta2, double eps,
                                                        double wd, int64_t step, const float *__restrict__ lr_dev,
                                                        int64_t *__restrict__ step_dev, int zero_grad) {
    __shared__ AdamScalars sh;
    __shared__ int64_t t_sh;
    if (threadIdx.x == 0) {
        // device-resident step counter: "steps taken so far"; this launch is step t = taken + 1
        const int64_t t = step_dev ? __hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 : step;
        sh = adam_scalars(lr_dev ? (double)*lr_dev : lr, beta1, beta2, eps, wd, t);
        t_sh = t;
    }
    __syncthreads();
    const AdamScalars s = sh;
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    if constexpr (kVec) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
            float4 p = reinterpret_cast<float4 *>(param)[i];
            const float4 g = reinterpret_cast<const float4 *>(grad)[i];
            float4 m = reinterpret_cast<float4 *>(exp_avg)[i];
            float4 v = reinterpret_cast<float4 *>(exp_avg_sq)[i];
            adam_elem(p.x, g.x, m.x, v.x, s);
            adam_elem(p.y, g.y, m.y, v.y, s);
            adam_elem(p.z, g.z, m.z, v.z, s);
            adam_elem(p.w, g.w, m.w, v.w, s);
            reinterpret_cast<float4 *>(param)[i] = p;
            reinterpret_cast<float4 *>(exp_avg)[i] = m;
            reinterpret_cast<float4 *>(exp_avg_sq)[i] = v;
            if (zero_grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
            adam_elem(p, grad[i], m, v, s);
            param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
            if (zero_grad) grad[i] = 0.f;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
            float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
            adam_elem(p, grad[i], m, v, s);
            param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
            if (zero_grad) grad[i] = 0.f;
        }
    }
    if (step_dev) {
        // Every workgroup read the counter before it got here; the last one to finish publishes t and re-arms
        // the ticket, so the launch can be replayed from a hipGraph without host-side arguments changing.
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long *ticket = reinterpret_cast<unsigned long long *>(step_dev + 1);
            const unsigned long long done =
                __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (done == gridDim.x - 1) {
                __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(step_dev, t_sh, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// One thread per output point: row = choice[b, n] of cloud b's raw scan, normalised as pcd_normalize
// (x/70, y/70, z/3, (i-0.5)*2, clip to [-1,1]; fp32 IEEE division as numpy's), plus the per-raw-point jitter row.
__device__ __forceinline__ float clip1(float v) { return v < -1.f ? -1.f : (v > 1.f ? 1.f : v); }   // NaN passes, as np.clip

__global__ __launch_bounds__(kThreads) void prepare_kernel(const float4 *__restrict__ raw, const int64_t *__restrict__ row_begin,
                                                           const int64_t *__restrict__ row_count,
                                                           const int32_t *__restrict__ raw_label,
                                                           const float4 *__restrict__ noise,
                                                           const int64_t *__restrict__ noise_begin,
                                                           const int64_t *__restrict__ choice, int B, int N,
                                                           float4 *__restrict__ points, int64_t *__restrict__ labels,
                                                           int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)B * N) return;
    const int b = (int)(i / N);
    const int64_t lo = row_begin[b], M = row_count[b];
    int64_t c = choice[i];
    if (c < 0 || c >= M) {                                  // numpy raises IndexError; here: flagged, row 0 used
        if (bad) atomicOr(bad, 1);
        c = 0;
        if (M <= 0) { points[i] = make_float4(0.f, 0.f, 0.f, 0.f); if (labels) labels[i] = 0; return; }
    }
    const float4 r = raw[lo + c];
    float4 o;
    o.x = r.x / 70.f;
    o.y = r.y / 70.f;
    o.z = r.z / 3.f;
    o.w = (r.w - 0.5f) * 2.f;
    o.x = clip1(o.x); o.y = clip1(o.y); o.z = clip1(o.z); o.w = clip1(o.w);
    if (noise) {                                            // jittered_data += pcd  (noise already clipped, fp32)
        const float4 z = noise[(noise_begin ? noise_begin[b] : lo) + c];
        o.x = z.x + o.x; o.y = z.y + o.y; o.z = z.z + o.z; o.w = z.w + o.w;
    }
    points[i] = o;
    if (labels) labels[i] = raw_label ? (int64_t)raw_label[lo + c] : 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int pn2_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, const float *lr_dev, int64_t *step_dev,
                  int zero_grad, pn2_stream_t stream) {
    PN2_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0);
    PN2_CHECK_ARG(step_dev || step >= 1);
    PN2_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0);
    PN2_CHECK_ARG(lr_dev || lr >= 0.0);
    const bool vec = aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq);
    int64_t blocks = pn2_cdiv(vec ? pn2_cdiv(n, 4) : n, kThreads);
    if (blocks > 8192) blocks = 8192;
    if (vec)
        hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, pn2_s(stream), param, grad, exp_avg,
                           exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, lr_dev, step_dev, zero_grad);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, pn2_s(stream), param, grad, exp_avg,
                           exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, lr_dev, step_dev, zero_grad);
    return pn2_launch_status();
}

int pn2_prepare_clouds(const float *raw, const int64_t *row_begin, const int64_t *row_count, const int32_t *raw_label,
                       const float *noise, const int64_t *noise_begin, const int64_t *choice, int B, int N,
                       float *points, int64_t *labels, int *bad_index, pn2_stream_t stream) {
    PN2_CHECK_ARG(raw && row_begin && row_count && choice && points && B > 0 && N > 0);
    PN2_CHECK_ARG(aligned16(raw) && aligned16(points) && (!noise || aligned16(noise)));
    hipLaunchKernelGGL(prepare_kernel, dim3((unsigned)pn2_cdiv((int64_t)B * N, kThreads)), dim3(kThreads), 0, pn2_s(stream),
                       reinterpret_cast<const float4 *>(raw), row_begin, row_count, raw_label, reinterpret_cast<const float4 *>(noise),
                       noise_begin, choice, B, N, reinterpret_cast<float4 *>(points), labels, bad_index);
    return pn2_launch_status();
}

}  // extern "C"
