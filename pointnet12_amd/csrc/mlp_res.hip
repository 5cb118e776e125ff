// Weight-resident shared-MLP kernels for the narrow, long layers (C_in, C_out multiples of 32, <= 128; P ~ 1e5..1e6 rows):
//   pn2_conv1x1_fwd  -> fwd_res_kernel   Y = act(X) W^T + b, BatchNorm statistics            (model/pointnet_util.py:197,254,312)
//   pn2_conv1x1_bwd  -> bwd_res_kernel   dX = dY W (masked by the previous ReLU), dW += dY^T act(X), the previous layer's
//                                        BatchNorm-backward reductions -- dgrad AND wgrad in ONE pass over dZ / Y / Y_prev
//
// Why a second family next to mlp.hip's streamed-weight GEMMs: on these layers the weight matrix is 4..64 KB -- it fits
// in LDS next to a row tile -- while the streamed kernels re-stage it through LDS for every 64-row tile (more LDS
// write traffic than the activations themselves, one barrier per 16-deep k-step), and the backward pass reads dZ, Y and
// Y_prev twice (dgrad, then wgrad) although both form the same dY.  These layers sit at 3..5.5 TB/s in the streamed
// kernels: HBM-bound, so halving the bytes is worth more than any issue-level tuning.
//
//   * W lives in LDS for the lifetime of a persistent workgroup (one per CU), loaded once.
//   * forward: every WAVE owns 32-row slabs end to end (private LDS staging buffer, 32 x N accumulator slab), so the
//     main loop has no workgroup barrier at all; the waves of a CU drift apart and fill each other's load / epilogue
//     phases on the matrix pipe the way separate workgroups would, but share one copy of W.
//   * backward: the 8 waves of a workgroup share one 64-row tile of dY (formed once, in LDS) and of Y_prev; the tile's
//     work -- dX tiles (contraction over C_out) and dW tiles (contraction over the 64 rows; divisible by row halves
//     because dW is accumulated with atomics anyway) -- is dealt to the waves by a host-side LPT plan so every wave
//     issues the same number of MFMAs.  dW accumulators stay in registers across all tiles of the workgroup.
#include "mlp_loaders.h"

namespace {

constexpr int RES_BM = 64;

// ----------------------------------------------------------------------------------------------- backward plan
// Per wave: at most one dX tile (row block rb, column block cj) and up to three dW units (co block cb, ci block cj,
// rows: 0 = rows 0..31 of the tile, 1 = rows 32..63, 2 = all 64).
struct ResPlan {
    signed char dx_rb[8], dx_cj[8];                 // -1: none
    signed char dw_n[8];
    signed char dw_cb[8][3], dw_cj[8][3], dw_rows[8][3];
};

inline bool make_res_plan(int CO_T, int CI_T, ResPlan *out) {
    ResPlan p;
    int load[8];
    for (int w = 0; w < 8; ++w) { p.dx_rb[w] = p.dx_cj[w] = -1; p.dw_n[w] = 0; load[w] = 0; }
    if (2 * CI_T > 8) return false;
    for (int i = 0; i < 2 * CI_T; ++i) { p.dx_rb[i] = (signed char)(i & 1); p.dx_cj[i] = (signed char)(i >> 1); load[i] = 16 * CO_T; }
    const int target = 8 * CO_T * CI_T;             // MFMAs per wave and tile if perfectly balanced
    auto least = [&](int skip) {
        int best = -1;
        for (int w = 0; w < 8; ++w)
            if (w != skip && p.dw_n[w] < 3 && (best < 0 || load[w] < load[best])) best = w;
        return best;
    };
    auto give = [&](int w, int cb, int cj, int rows) {
        const int u = p.dw_n[w]++;
        p.dw_cb[w][u] = (signed char)cb; p.dw_cj[w][u] = (signed char)cj; p.dw_rows[w][u] = (signed char)rows;
        load[w] += rows == 2 ? 32 : 16;
    };
    for (int cb = 0; cb < CO_T; ++cb)
        for (int cj = 0; cj < CI_T; ++cj) {
            const int w = least(-1);
            if (w < 0) return false;
            if (load[w] + 32 <= target) { give(w, cb, cj, 2); continue; }
            const int w2 = least(w);
            if (w2 < 0) { give(w, cb, cj, 2); continue; }
            give(w, cb, cj, 0);
            give(w2, cb, cj, 1);
        }
    *out = p;
    return true;
}

extern __shared__ __attribute__((aligned(16))) float res_lds[];

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }   // 32x32 C/D layout

// ----------------------------------------------------------------------------------------------- fused backward
// dyload: LoadDyDense / LoadDyPooled over this layer's (dZ | pooled dZ, Y, coef).  Yp: the previous layer's pre-BN
// output [P, Ci] (MASKED: X = relu(bn(Yp)) with aff_p, dX masked by X > 0 and reduced into red_p) or the plain layer
// input (!MASKED: X = Yp as stored, dX = dY W unmasked, no reductions).
template <int CO_T, class DyLoad, bool MASKED>
__global__ __launch_bounds__(512, 2) void bwd_res_kernel(DyLoad dyload, const float *__restrict__ Yp, int ldp,
                                                         const float *__restrict__ aff_p, const float *__restrict__ W, int ldw,
                                                         int64_t P, int Ci, float *__restrict__ dX, int ldxo,
                                                         double *__restrict__ red_p, float *__restrict__ dW, int lddw,
                                                         ResPlan plan, const float *zp) {
    constexpr int Co = 32 * CO_T, LDY = Co + 4, QD = Co / 4, IT_D = RES_BM * QD / 512;
    static_assert(RES_BM * QD % 512 == 0, "dY tile must split evenly over 512 threads");
    const int LDP = Ci + 4, QP = Ci >> 2, IT_P = QP >> 3;          // 64 * QP / 512 quads of Y_prev per thread (Ci % 32 == 0)
    float *Wt = res_lds;                                           // [Ci][LDY]: W transposed, co contiguous
    float *dYs = Wt + Ci * LDY;                                    // [64][LDY]
    float *Yps = dYs + RES_BM * LDY;                               // [64][LDP]
    float *tab = Yps + RES_BM * LDP;                               // coefficient rows c0, q1, q0, mean of this layer: 4 * Co

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int64_t tiles = (P + RES_BM - 1) / RES_BM;

    // ---- one-time: W^T, coefficient table
    for (int i = t; i < Co * Ci; i += 512) {
        const int co = i / Ci, ci = i - co * Ci;
        Wt[ci * LDY + co] = W[(int64_t)co * ldw + ci];
    }
    for (int i = t; i < 4 * Co; i += 512) tab[i] = dyload.tab_src()[i];

    // ---- this wave's share of every tile (fixed for the whole launch)
    const int dx_rb = plan.dx_rb[wave], dx_cj = plan.dx_cj[wave], n_dw = plan.dw_n[wave];
    int dw_a[3], dw_b[3], dw_p0[3], dw_p1[3];                      // LDS column offsets of the unit, its row range
    float xmu[3], xsc[3], xbe[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const bool on = u < n_dw;
        const int cb = on ? plan.dw_cb[wave][u] : 0, cj = on ? plan.dw_cj[wave][u] : 0, rows = on ? plan.dw_rows[wave][u] : 0;
        dw_a[u] = cb * 32 + l31;
        dw_b[u] = cj * 32 + l31;
        dw_p0[u] = on ? (rows == 1 ? 16 : 0) : 0;                  // in units of row PAIRS
        dw_p1[u] = on ? (rows == 0 ? 16 : 32) : 0;
        if (MASKED) {
            Affine a(aff_p, Ci);
            xmu[u] = a.mean[dw_b[u]]; xsc[u] = a.scale[dw_b[u]]; xbe[u] = a.beta[dw_b[u]];
        }
    }
    float emu = 0.f, esc = 0.f, ebe = 0.f, eis = 0.f;             // epilogue constants of the dX tile's column
    const int ecol = (dx_cj < 0 ? 0 : dx_cj) * 32 + l31;
    if (MASKED && dx_rb >= 0) {
        Affine a(aff_p, Ci);
        emu = a.mean[ecol]; esc = a.scale[ecol]; ebe = a.beta[ecol]; eis = a.invstd[ecol];
    }

    f32x16 accw[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[u][r] = 0.f;
    double st0 = 0.0, st1 = 0.0;

    // ---- tile loaders: dY quads idx = t + 512 i -> (row, quad); Y_prev quads likewise
    typename DyLoad::template Raw<1> ra[IT_D];
    float4 rp[4];
    auto fetch = [&](int64_t tile) {
        const int64_t m0 = tile * RES_BM;
        const bool tv = tile < tiles;
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            const int idx = t + 512 * i, row = idx / QD, q = idx - row * QD;
            dyload.template issue<1>(ra[i], m0 + row, 0, 4 * q, P, tv);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < IT_P) {
                const int idx = t + 512 * i, row = idx / QP, q = idx - row * QP;
                rp[i] = ld4((tv && m0 + row < P) ? Yp + row_off(m0 + row, ldp) + 4 * q : zp);
            }
        }
    };
    int64_t tile = blockIdx.x;
    fetch(tile);
    __syncthreads();                                               // Wt and tab are in place

    for (; tile < tiles; tile += gridDim.x) {
        const int64_t m0 = tile * RES_BM;
        // ---- registers -> LDS (dY formed here, once per row)
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            const int idx = t + 512 * i, row = idx / QD, q = idx - row * QD;
            const DyParams dp = dy_params_tab(tab, Co, 4 * q, true);
            *reinterpret_cast<float4 *>(&dYs[row * LDY + 4 * q]) = dyload.template finish<1>(ra[i], 0, m0 + row < P, dp);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < IT_P) {
                const int idx = t + 512 * i, row = idx / QP, q = idx - row * QP;
                *reinterpret_cast<float4 *>(&Yps[row * LDP + 4 * q]) = rp[i];
            }
        }
        fetch(tile + gridDim.x);                                   // in flight under this tile's MFMAs
        __syncthreads();

        // ---- dX tile: rows rb*32.., columns cj*32.. ; contraction over Co, both operands 4 k-values per ds_read_b128
        if (dx_rb >= 0) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *ap = &dYs[(dx_rb * 32 + l31) * LDY + 4 * lh];
            const float *bp = &Wt[ecol * LDY + 4 * lh];
#pragma unroll
            for (int kb = 0; kb < Co / 8; ++kb) {
                const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * kb);
                const float4 b = *reinterpret_cast<const float4 *>(bp + 8 * kb);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = dx_rb * 32 + acc_row(r, lh);
                float dz = acc[r];
                if (MASKED) {
                    const float y = Yps[row * LDP + ecol];
                    dz = bn_act(y, emu, esc, ebe) > 0.f ? dz : 0.f;
                    s0 += dz;
                    s1 = __builtin_fmaf(dz, (y - emu) * eis, s1);
                }
                if (m0 + row < P) __builtin_nontemporal_store(dz, dX + row_off(m0 + row, ldxo) + ecol);
            }
            if (MASKED) { st0 += (double)s0; st1 += (double)s1; }
        }
        // ---- dW units: contraction over the rows of the tile, one MFMA per row pair
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u < n_dw) {
                const float *ap = &dYs[lh * LDY + dw_a[u]];
                const float *bp = &Yps[lh * LDP + dw_b[u]];
#pragma unroll 8
                for (int pp = dw_p0[u]; pp < dw_p1[u]; ++pp) {
                    const float a = ap[2 * pp * LDY];
                    float b = bp[2 * pp * LDP];
                    if (MASKED) b = fmaxf(bn_act(b, xmu[u], xsc[u], xbe[u]), 0.f);
                    accw[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accw[u], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                           // tile consumed: the next one may land
    }

    // ---- flush: dW partial tiles (256 contiguous bytes per wave-instruction), the dX column's two reductions
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        if (u < n_dw) {
            const int cb32 = dw_a[u] - l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(dW + (int64_t)(cb32 + acc_row(r, lh)) * lddw + dw_b[u], accw[u][r]);
        }
    }
    if (MASKED && red_p != nullptr && dx_rb >= 0) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (lh == 0) {
            double *rep = red_p + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * Ci;
            atomicAdd(rep + ecol, st0);
            atomicAdd(rep + Ci + ecol, st1);
        }
    }
}

inline size_t bwd_res_lds_bytes(int Co, int Ci) {
    return sizeof(float) * ((size_t)Ci * (Co + 4) + RES_BM * (Co + 4) + RES_BM * (Ci + 4) + 4 * Co);
}

template <int CO_T, class DyLoad, bool MASKED>
int launch_bwd_res(DyLoad dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t P, int Ci, float *dX,
                   int ldxo, double *red_p, float *dW, int lddw, hipStream_t s) {
    ResPlan plan;
    if (!make_res_plan(CO_T, Ci / 32, &plan)) return PN2_EINVAL;
    const size_t lds = bwd_res_lds_bytes(32 * CO_T, Ci);
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&bwd_res_kernel<CO_T, DyLoad, MASKED>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PN2_ELAUNCH;
        raised = true;
    }
    const int64_t tiles = pn2_cdiv(P, RES_BM);
    const int64_t cap = pn2_num_cus();                             // one 8-wave workgroup per CU
    hipLaunchKernelGGL((bwd_res_kernel<CO_T, DyLoad, MASKED>), dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(512), lds, s, dy, Yp,
                       ldp, aff_p, W, ldw, P, Ci, dX, ldxo, red_p, dW, lddw, plan, zero_page_dev());
    return pn2_launch_status();
}

template <class DyLoad, bool MASKED>
int dispatch_bwd_res(int Co, DyLoad dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t P, int Ci,
                     float *dX, int ldxo, double *red_p, float *dW, int lddw, hipStream_t s) {
    switch (Co / 32) {
        case 1: return launch_bwd_res<1, DyLoad, MASKED>(dy, Yp, ldp, aff_p, W, ldw, P, Ci, dX, ldxo, red_p, dW, lddw, s);
        case 2: return launch_bwd_res<2, DyLoad, MASKED>(dy, Yp, ldp, aff_p, W, ldw, P, Ci, dX, ldxo, red_p, dW, lddw, s);
        case 3: return launch_bwd_res<3, DyLoad, MASKED>(dy, Yp, ldp, aff_p, W, ldw, P, Ci, dX, ldxo, red_p, dW, lddw, s);
        case 4: return launch_bwd_res<4, DyLoad, MASKED>(dy, Yp, ldp, aff_p, W, ldw, P, Ci, dX, ldxo, red_p, dW, lddw, s);
    }
    return PN2_EINVAL;
}

// ----------------------------------------------------------------------------------------------- resident forward
// Every wave owns 32-row slabs: global -> registers (in flight under the previous slab's MFMAs) -> BN + ReLU -> its private
// LDS buffer [32][K+4] -> MFMA against the shared W image [N][K+4] -> bias, store, statistics straight from the
// accumulators.  ACT: X = relu(bn(X_raw)) with the affine block `aff` (hidden layers) or X as stored.
template <int K_T, int N_T, bool ACT>
__global__ __launch_bounds__(512, 2) void fwd_res_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ aff,
                                                         const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                         float *__restrict__ Y, int ldy, int64_t P, double *__restrict__ stats,
                                                         const float *zp) {
    constexpr int K = 32 * K_T, N = 32 * N_T, LDA = K + 4, QK = K / 4, IT = 32 * QK / 64;
    const int NW = blockDim.x >> 6;
    float *Ws = res_lds;                                           // [N][LDA]
    float *atab = Ws + N * LDA;                                    // mean, scale, beta rows of the input BatchNorm: 3 * K
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    float *Ab = atab + 3 * K + wave * (32 * LDA);                  // this wave's staging buffer [32][LDA]

    for (int i = t; i < N * QK; i += blockDim.x) {
        const int n = i / QK, q = i - n * QK;
        const float *src = W + (int64_t)n * ldw + 4 * q;
        float4 v;
        if ((ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) v = ld4(src);
        else v = make_float4(src[0], src[1], src[2], src[3]);
        *reinterpret_cast<float4 *>(&Ws[n * LDA + 4 * q]) = v;
    }
    if (ACT)
        for (int i = t; i < 3 * K; i += blockDim.x) atab[i] = aff[i];          // affine block rows of pitch K (K % 4 == 0)
    float bj[N_T];
#pragma unroll
    for (int j = 0; j < N_T; ++j) bj[j] = bias[32 * j + l31];

    const int64_t slabs = (P + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * NW;
    float4 rx[IT];
    auto fetch = [&](int64_t slab) {
        const int64_t m0 = slab * 32;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = lane + 64 * i, row = idx / QK, q = idx - row * QK;
            rx[i] = ld4((slab < slabs && m0 + row < P) ? X + row_off(m0 + row, ldx) + 4 * q : zp);
        }
    };
    int64_t slab = (int64_t)blockIdx.x * NW + wave;
    fetch(slab);
    double st[N_T][2];
#pragma unroll
    for (int j = 0; j < N_T; ++j) st[j][0] = st[j][1] = 0.0;
    __syncthreads();                                               // W image and table complete (the only barrier before the end)

    for (; slab < slabs; slab += stride) {
        const int64_t m0 = slab * 32;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = lane + 64 * i, row = idx / QK, q = idx - row * QK;
            float4 x = rx[i];
            if (ACT) {
                const float4 mu = *reinterpret_cast<const float4 *>(&atab[4 * q]);
                const float4 sc = *reinterpret_cast<const float4 *>(&atab[K + 4 * q]);
                const float4 be = *reinterpret_cast<const float4 *>(&atab[2 * K + 4 * q]);
                x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
                x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
                x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
                x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
                if (m0 + row >= P) x = kZero4;                       // rows past the end contribute nothing to the statistics
            }
            *reinterpret_cast<float4 *>(&Ab[row * LDA + 4 * q]) = x;
        }
        fetch(slab + stride);

        f32x16 acc[N_T];
#pragma unroll
        for (int j = 0; j < N_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float *ap = &Ab[l31 * LDA + 4 * lh];
        const float *bp = &Ws[l31 * LDA + 4 * lh];
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) {
            const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * kb);
#pragma unroll
            for (int j = 0; j < N_T; ++j) {
                const float4 b = *reinterpret_cast<const float4 *>(bp + j * 32 * LDA + 8 * kb);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < N_T; ++j) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + acc_row(r, lh);
                if (m < P) {
                    const float y = acc[j][r] + bj[j];
                    __builtin_nontemporal_store(y, Y + row_off(m, ldy) + 32 * j + l31);
                    s0 += y;
                    s1 = __builtin_fmaf(y, y, s1);
                }
            }
            st[j][0] += (double)s0;
            st[j][1] += (double)s1;
        }
    }

    if (stats != nullptr) {                                        // fold the waves in LDS (the W image is dead), one atomic per channel
        __syncthreads();
        double *red = reinterpret_cast<double *>(res_lds);        // [NW][N][2]
#pragma unroll
        for (int j = 0; j < N_T; ++j) {
            double a0 = st[j][0], a1 = st[j][1];
            a0 += __shfl_xor(a0, 32, 64);
            a1 += __shfl_xor(a1, 32, 64);
            if (lh == 0) {
                red[(wave * N + 32 * j + l31) * 2] = a0;
                red[(wave * N + 32 * j + l31) * 2 + 1] = a1;
            }
        }
        __syncthreads();
        if (t < N) {
            double a0 = 0.0, a1 = 0.0;
            for (int w = 0; w < NW; ++w) { a0 += red[(w * N + t) * 2]; a1 += red[(w * N + t) * 2 + 1]; }
            double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
            atomicAdd(rep + t, a0);
            atomicAdd(rep + N + t, a1);
        }
    }
}

template <int K_T, int N_T, bool ACT>
int launch_fwd_res(const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y, int ldy,
                   int64_t P, double *stats, hipStream_t s) {
    constexpr int K = 32 * K_T, N = 32 * N_T, LDA = K + 4;
    const size_t fixed = sizeof(float) * ((size_t)N * LDA + 3 * K), per_wave = sizeof(float) * 32 * LDA;
    int nw = (int)((160 * 1024 - fixed) / per_wave);
    if (nw > 8) nw = 8;
    if (nw < 4) return PN2_EINVAL;
    size_t lds = fixed + nw * per_wave;
    const size_t red = sizeof(double) * 2 * N * nw;                 // the final fold reuses the image
    if (lds < red) lds = red;
    static bool raised = false;
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_res_kernel<K_T, N_T, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return PN2_ELAUNCH;
        raised = true;
    }
    const int64_t slabs = pn2_cdiv(P, 32);
    int64_t grid = pn2_cdiv(slabs, nw);
    if (grid > pn2_num_cus()) grid = pn2_num_cus();
    hipLaunchKernelGGL((fwd_res_kernel<K_T, N_T, ACT>), dim3((unsigned)grid), dim3(64 * nw), lds, s, X, ldx, aff, W, ldw, bias, Y, ldy, P,
                       stats, zero_page_dev());
    return pn2_launch_status();
}

template <int K_T, bool ACT>
int dispatch_fwd_res_n(int N, const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y, int ldy,
                       int64_t P, double *stats, hipStream_t s) {
    switch (N / 32) {
        case 1: return launch_fwd_res<K_T, 1, ACT>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 2: return launch_fwd_res<K_T, 2, ACT>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 3: return launch_fwd_res<K_T, 3, ACT>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 4: return launch_fwd_res<K_T, 4, ACT>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
    }
    return PN2_EINVAL;
}

template <bool ACT>
int dispatch_fwd_res(int K, int N, const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y,
                     int ldy, int64_t P, double *stats, hipStream_t s) {
    switch (K / 32) {
        case 1: return dispatch_fwd_res_n<1, ACT>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 2: return dispatch_fwd_res_n<2, ACT>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 3: return dispatch_fwd_res_n<3, ACT>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
        case 4: return dispatch_fwd_res_n<4, ACT>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, s);
    }
    return PN2_EINVAL;
}

inline int res_min_rows() {
    static const int v = [] { const char *e = getenv("PN2_RES_MIN_ROWS"); return e ? atoi(e) : 32768; }();
    return v;
}

}  // namespace

// Shapes the weight-resident kernels take: both channel counts multiples of 32 and <= 128, enough rows to give every
// CU a few tiles.  PN2_RES=0 turns the family off (A/B runs against the streamed kernels).
static bool res_shape_ok(int C_out, int C_in) {
    return C_out % 32 == 0 && C_in % 32 == 0 && C_out >= 32 && C_out <= 128 && C_in >= 32 && C_in <= 128;
}

extern "C" int pn2_res_supported(int64_t P, int C_out, int C_in) {
    static const int on = [] { const char *e = getenv("PN2_RES"); return e ? atoi(e) : 1; }();
    return on && P >= res_min_rows() && P < (1LL << 31) && res_shape_ok(C_out, C_in);
}

// Called by pn2_conv1x1_fwd (mlp.hip) for supported shapes when no fused BatchNorm tail is requested.
int pn2_fwd_res(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                int64_t P, int K, int N, double *stats, hipStream_t s) {
    if (in_affine) return dispatch_fwd_res<true>(K, N, X, ldx, in_affine, W, ldw, bias, Y, ldy, P, stats, s);
    return dispatch_fwd_res<false>(K, N, X, ldx, nullptr, W, ldw, bias, Y, ldy, P, stats, s);
}

extern "C" int pn2_conv1x1_bwd(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y,
                               int ldy, const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev,
                               const float *prev_affine, float *dXout, int ldxo, double *prev_red, float *dW, int lddw,
                               int64_t P, int C_out, int C_in, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && coef && W && prev_Y && dXout && dW && P > 0 && P < (1LL << 31) && res_shape_ok(C_out, C_in));
    PN2_CHECK_ARG(dZ != nullptr || (dZp && arg && Kpool > 0));
    PN2_CHECK_ARG(ldw >= C_in && lddw >= C_in && ldy % 4 == 0 && ldy >= C_out && ld_prev % 4 == 0 && ld_prev >= C_in && ldxo >= C_in);
    PN2_CHECK_ARG(prev_affine != nullptr || prev_red == nullptr);
    hipStream_t s = pn2_s(stream);
    const float *zp = zero_page_dev();
    if (dZ) {
        PN2_CHECK_ARG(ldz % 4 == 0 && ldz >= C_out);
        LoadDyDense ld{dZ, ldz, Y, ldy, coef, C_out, zp};
        if (prev_affine)
            return dispatch_bwd_res<LoadDyDense, true>(C_out, ld, prev_Y, ld_prev, prev_affine, W, ldw, P, C_in, dXout, ldxo, prev_red, dW, lddw, s);
        return dispatch_bwd_res<LoadDyDense, false>(C_out, ld, prev_Y, ld_prev, nullptr, W, ldw, P, C_in, dXout, ldxo, nullptr, dW, lddw, s);
    }
    PN2_CHECK_ARG(ldo % 4 == 0 && ldo >= C_out && prev_affine != nullptr);
    LoadDyPooled ld{dZp, ldo, arg, Kpool, Y, ldy, coef, C_out, zp, pow2_shift(Kpool)};
    return dispatch_bwd_res<LoadDyPooled, true>(C_out, ld, prev_Y, ld_prev, prev_affine, W, ldw, P, C_in, dXout, ldxo, prev_red, dW, lddw, s);
}
