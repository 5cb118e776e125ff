// Eval-mode fused shared MLP: gather -> L x (linear + ReLU) -> max over the neighbours, ONE launch per module.
//
// Under .eval() BatchNorm is a fixed per-channel affine map, so it folds into the conv that precedes it
// (W' = diag(gamma / sqrt(var + eps)) W, b' = (b - mean) gamma / sqrt(var + eps) + beta; the host does that once per
// weight version, in fp64) and a whole set-abstraction / feature-propagation MLP becomes a chain of GEMM + ReLU with
// no statistics barrier between the layers.  A workgroup owns 32 consecutive grouped rows end to end: the gathered,
// centred, concatenated input rows are formed straight in LDS, every layer's activations stay in LDS (two buffers,
// ping-pong), the last layer's accumulators are reduced over the rows of a group in registers, and only the pooled
// [G, C_L] result is written -- no Y tensors, no grouped tensor, no argmax (nothing to differentiate).
// Replaces model/pointnet_util.py:127-133 / :243-251 + :194-199 / :251-256 / :309-312 under .eval()
// (the reference's viewer loop, pcdvis.py:118-136, runs exactly this).
#include "mlp_loaders.h"

namespace {

constexpr int EV_ROWS = 32, EV_THREADS = 256, EV_MAXL = 4;

struct EvalArgs {
    const float *W[EV_MAXL];          // folded weights [N_l][ldw_l], zero padded to ldw_l = round8(K_l)
    const float *bias[EV_MAXL];       // folded bias [N_l]
    int K[EV_MAXL], N[EV_MAXL], ldw[EV_MAXL];
    int L;
};

struct EvalInput {
    const float *X; int ldx;          // plain rows [P, ldx] (X != nullptr) ...
    const float *xyz, *points, *new_xyz; const int64_t *idx;   // ... or grouped: gather + centre + concat
    int N, S, Knb, D, xyz_first;
};

extern __shared__ __attribute__((aligned(16))) float ev_lds[];

__global__ __launch_bounds__(EV_THREADS) void fused_eval_kernel(EvalInput in, EvalArgs ar, int pool, float *__restrict__ out, int ldo,
                                                               int64_t P, int LP) {
    float *bufA = ev_lds, *bufB = ev_lds + EV_ROWS * LP;           // [32][LP], LP = 4 mod 8: conflict-free ds_read_b128
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * EV_ROWS;

    // ---- input rows -> bufA, zero padded to round8(K_0) columns
    const int K0 = ar.K[0], K0p = (K0 + 7) & ~7;
    {
        const int r = t >> 3, c0 = t & 7;                          // 8 threads per row
        const int64_t m = m0 + r;
        float *dst = bufA + r * LP;
        if (m < P && in.X != nullptr) {
            const float *src = in.X + m * in.ldx;
            for (int c = c0; c < K0p; c += 8) dst[c] = c < K0 ? src[c] : 0.f;
        } else if (m < P) {
            const int64_t g = m / in.Knb;                          // (b, s) of the row; idx == nullptr: group_all (un-centred)
            const int64_t b = g / in.S;
            const int64_t j = in.idx ? in.idx[m] : m - g * in.Knb;
            const float *px = in.xyz + (b * in.N + j) * 3;
            const float *pf = in.points ? in.points + (b * in.N + j) * in.D : nullptr;
            float cx = 0.f, cy = 0.f, cz = 0.f;
            if (in.new_xyz) { const float *q = in.new_xyz + g * 3; cx = q[0]; cy = q[1]; cz = q[2]; }
            const int xo = in.xyz_first ? 0 : in.D, fo = in.xyz_first ? 3 : 0;
            for (int c = c0; c < K0p; c += 8) {
                float v = 0.f;
                if (c >= xo && c < xo + 3) v = c - xo == 0 ? px[0] - cx : (c - xo == 1 ? px[1] - cy : px[2] - cz);
                else if (c >= fo && c < fo + in.D) v = pf[c - fo];
                dst[c] = v;
            }
        } else {
            for (int c = c0; c < K0p; c += 8) dst[c] = 0.f;
        }
    }
    __syncthreads();

    float *src = bufA, *dstb = bufB;
    for (int l = 0; l < ar.L; ++l) {
        const int K8 = (ar.K[l] + 7) & ~7, N = ar.N[l], ldw = ar.ldw[l];
        const bool last = l == ar.L - 1;
        const float *W = ar.W[l], *bias = ar.bias[l];
        const int nblk = (N + 31) >> 5;
        for (int j = wave; j < nblk; j += 4) {                     // 32-column blocks of this layer, dealt to the four waves
            const int col = 32 * j + l31;
            const float *wrow = W + (int64_t)(col < N ? col : N - 1) * ldw + 4 * lh;      // clamped: columns >= N are discarded
            const float *arow = src + l31 * LP + 4 * lh;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            int kb = 0;
            for (; kb + 4 <= K8 / 8; kb += 4) {                    // four weight quads in flight per lane
                float4 b[4], a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = ld4(wrow + 8 * (kb + u));
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4 *>(arow + 8 * (kb + u));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
                }
            }
            for (; kb < K8 / 8; ++kb) {
                const float4 b = ld4(wrow + 8 * kb);
                const float4 a = *reinterpret_cast<const float4 *>(arow + 8 * kb);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            const float bc = col < N ? bias[col] : 0.f;
            if (!last) {                                           // relu -> the other buffer (pad columns up to round8(N): zero)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (col < ((N + 7) & ~7)) dstb[row * LP + col] = col < N ? fmaxf(acc[r] + bc, 0.f) : 0.f;   // (a row holds round8(N) columns)
                }
            } else if (pool == 0) {                                // FP / head: rows out
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < P && col < N) out[m * ldo + col] = fmaxf(acc[r] + bc, 0.f);
                }
            } else {                                               // SA: max over the rows of a group (relu(max) = max(relu))
                float m_lo = -INFINITY, m_hi = -INFINITY;          // rows 0..15 / 16..31 of the tile (registers 0..7 / 8..15)
#pragma unroll
                for (int r = 0; r < 8; ++r) { m_lo = fmaxf(m_lo, acc[r]); m_hi = fmaxf(m_hi, acc[r + 8]); }
                m_lo = fmaxf(m_lo, __shfl_xor(m_lo, 32, 64));
                m_hi = fmaxf(m_hi, __shfl_xor(m_hi, 32, 64));
                if (lh == 0 && col < N && m0 < P) {
                    if (pool == 16) {                              // two groups per tile
                        const int64_t g = m0 / 16;
                        out[g * ldo + col] = fmaxf(m_lo + bc, 0.f);
                        if (m0 + 16 < P) out[(g + 1) * ldo + col] = fmaxf(m_hi + bc, 0.f);
                    } else {
                        const float v = fmaxf(fmaxf(m_lo, m_hi) + bc, 0.f);
                        const int64_t g = m0 / pool;
                        if (pool == 32) out[g * ldo + col] = v;
                        else atomicMax(reinterpret_cast<int *>(out + g * ldo + col), __float_as_int(v));   // v >= 0: int order
                    }
                }
            }
        }
        __syncthreads();
        float *tmp = src; src = dstb; dstb = tmp;
    }
}

}  // namespace

extern "C" int pn2_fused_eval(const float *X, int ldx, const float *xyz, const float *points, const float *new_xyz, const int64_t *idx,
                              int B, int N, int S, int Knb, int D, int xyz_first, const pn2_eval_layer *layers, int L, int pool,
                              float *out, int ldo, pn2_stream_t stream) {
    PN2_CHECK_ARG(layers && out && L >= 1 && L <= EV_MAXL && B > 0 && ldo > 0);
    PN2_CHECK_ARG(X != nullptr || (xyz && B > 0 && N > 0 && S > 0 && Knb > 0 && D >= 0 && (D == 0 || points)));
    PN2_CHECK_ARG(pool == 0 || (pool == Knb && (pool == 16 || pool % 32 == 0)));
    EvalArgs ar;
    ar.L = L;
    int cmax = 8;
    for (int l = 0; l < L; ++l) {
        PN2_CHECK_ARG(layers[l].W && layers[l].bias && layers[l].K > 0 && layers[l].N > 0 && layers[l].ldw >= ((layers[l].K + 7) & ~7) &&
                      (reinterpret_cast<uintptr_t>(layers[l].W) & 15) == 0 && layers[l].ldw % 4 == 0);
        PN2_CHECK_ARG(l == 0 || layers[l].K == layers[l - 1].N);
        ar.W[l] = layers[l].W; ar.bias[l] = layers[l].bias; ar.K[l] = layers[l].K; ar.N[l] = layers[l].N; ar.ldw[l] = layers[l].ldw;
        const int k8 = (layers[l].K + 7) & ~7;
        if (k8 > cmax) cmax = k8;
        if (l + 1 < L && ((layers[l].N + 7) & ~7) > cmax) cmax = (layers[l].N + 7) & ~7;
    }
    PN2_CHECK_ARG(X != nullptr ? layers[0].K <= ldx : layers[0].K == 3 + D);
    PN2_CHECK_ARG(ldo >= layers[L - 1].N);
    const int LP = cmax + 4;
    const size_t lds = sizeof(float) * 2 * EV_ROWS * LP;
    PN2_CHECK_ARG(lds <= 160 * 1024);
    const int64_t P = X ? (int64_t)B : (int64_t)B * S * Knb;      // plain rows: the caller passes the row count in B
    PN2_CHECK_ARG(pool == 0 || P % pool == 0);
    hipStream_t s = pn2_s(stream);
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&fused_eval_kernel), raised) != PN2_OK) return PN2_ELAUNCH;
    if (pool > 32) pn2_fill_u32(out, 0u, (P / pool) * (int64_t)ldo, s);       // atomicMax target: relu outputs are >= 0
    EvalInput in{X, ldx, xyz, points, new_xyz, idx, N, S, Knb, D, xyz_first};
    hipLaunchKernelGGL(fused_eval_kernel, dim3((unsigned)pn2_cdiv(P, EV_ROWS)), dim3(EV_THREADS), lds, s, in, ar, pool, out, ldo, P, LP);
    return pn2_launch_status();
}
