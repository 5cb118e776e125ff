// Experiment (round 3): "weights stationary in registers" NT GEMM for the wide shared-MLP layers.
//   Y[P][N] = relu(bn(X[P][K])) W[N][K]^T + bias, per-channel sum(y), sum(y^2)
// Every wave owns ONE 32-column block of W for the whole launch as the B operand of v_mfma_f32_32x32x2_f32 (K/2 registers per
// lane); the activations go through LDS in 32-deep chunks shared by all waves of the workgroup, one barrier per chunk, TM * 16
// MFMAs per wave between barriers (the streamed kernel of mlp.hip: 16).  Standalone: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float bn_act(float y, float mean, float scale, float beta) { return __builtin_fmaf(y - mean, scale, beta); }

// KP: K padded to a multiple of 8.  NCB: 32-column blocks of N (one per wave column).  RS: row splits (waves = NCB * RS).
// TM: 32-row blocks per wave.  Workgroup tile: BM = 32 * TM * RS rows x 32 * NCB columns.
template <int KP, int NCB, int RS, int TM, bool ACT, int SCHED, int ABL = 0>
__global__ __launch_bounds__(64 * NCB * RS) void regw_fwd_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ aff,
                                                                 const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                                 float *__restrict__ Y, int ldy, int64_t P, int K, int N,
                                                                 double *__restrict__ stats) {
    constexpr int NT = 64 * NCB * RS, BM = 32 * TM * RS, KC = 32, LDP = KC + 4, NCH = (KP + KC - 1) / KC;
    constexpr int QPR = KC / 4;                                 // float4 per row of a chunk
    constexpr int A_IT = (BM * QPR + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDP];
    __shared__ __attribute__((aligned(16))) float tab[3 * KP];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int cb = wave % NCB, rs = wave / NCB;
    const int n = cb * 32 + l31;

    // ---- one-time: this lane's slice of W (B operand of every MFMA it will issue), the BatchNorm table
    float w[KP / 2];
#pragma unroll
    for (int kb = 0; kb < KP / 8; ++kb) {
        const int k = 8 * kb + 4 * lh;
        if ((ldw & 3) == 0 && n < N && k + 3 < K) {
            const float4 v = ld4(W + (int64_t)n * ldw + k);
            w[kb * 4 + 0] = v.x; w[kb * 4 + 1] = v.y; w[kb * 4 + 2] = v.z; w[kb * 4 + 3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) w[kb * 4 + e] = (n < N && k + e < K) ? W[(int64_t)n * ldw + k + e] : 0.f;
        }
    }
    if (ACT)
        for (int i = t; i < 3 * KP; i += NT) {
            const int r = i / KP, k = i - r * KP;
            tab[i] = k < K ? aff[r * ((K + 3) & ~3) + k] : 0.f;
        }
    const float bj = n < N ? bias[n] : 0.f;

    const int64_t tiles = (P + BM - 1) / BM;
    float4 ra[A_IT];
    // flattened (tile, chunk) sequence; chunk c of a tile covers k in [32 c, 32 c + 32) (the last one may be narrower)
    auto fetch = [&](int64_t tile, int c) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = t + NT * i, row = idx / QPR, kq = idx % QPR;
            const int64_t m = tile * BM + row;
            const int k = c * KC + 4 * kq;
            const bool v = idx < BM * QPR && tile < tiles && m < P && k < ((K + 3) & ~3);
            if (!(ABL & 1) || (tile == blockIdx.x && c == 0)) ra[i] = v ? ld4(X + m * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](float *dst, int64_t tile, int c) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = t + NT * i, row = idx / QPR, kq = idx % QPR;
            const int k = c * KC + 4 * kq;
            float4 x = ra[i];
            if (ACT) {
                const int kk = k < KP ? k : 0;
                const float4 mu = *reinterpret_cast<const float4 *>(&tab[kk]);
                const float4 sc = *reinterpret_cast<const float4 *>(&tab[KP + kk]);
                const float4 be = *reinterpret_cast<const float4 *>(&tab[2 * KP + kk]);
                x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
                x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
                x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
                x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
                const int64_t m = tile * BM + row;
                if (!(m < P) || k >= ((K + 3) & ~3)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (idx < BM * QPR && (!(ABL & 2) || (tile == blockIdx.x && c == 0))) *reinterpret_cast<float4 *>(&dst[row * LDP + 4 * kq]) = x;
        }
    };

    double st0 = 0.0, st1 = 0.0;
    int64_t tile = blockIdx.x;
    float *cur = As[0], *nxt = As[1];
    fetch(tile, 0);
    __syncthreads();                                            // tab complete
    stage(cur, tile, 0);
    if (NCH > 1) fetch(tile, 1); else fetch(tile + gridDim.x, 0);
    while (tile < tiles) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                    // chunk c is in `cur`; everybody is done with `nxt`
            const int kbs = (KP - c * KC) >= KC ? KC / 8 : (KP - c * KC) / 8;     // static after unrolling
            const float *ap = cur + (rs * TM * 32 + l31) * LDP + 4 * lh;
            const bool last = c == NCH - 1;
            const int64_t t1 = last ? tile + gridDim.x : tile;
            const int c1 = last ? 0 : c + 1;
            const bool last2 = c1 == NCH - 1;
            // operand reads one k-block ahead of the MFMAs that consume them (two register sets); the memory clobbers
            // keep the compiler from hoisting every read of the chunk to its top (16 x 4 registers: spills)
            float4 a[2][TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) if (!(ABL & 8) || c == 0) a[0][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDP);
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                if (kb < kbs) {
                    asm volatile("" ::: "memory");
                    if (kb + 1 < kbs) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) { if (!(ABL & 8)) a[(kb + 1) & 1][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDP + 8 * (kb + 1)); else a[(kb + 1) & 1][i] = a[kb & 1][i]; }
                    }
                    if (SCHED == 0 && kb == kbs - 1) {           // staging of the next chunk behind the last reads of this one
                        stage(nxt, t1, c1);
                        fetch(last2 ? t1 + gridDim.x : t1, last2 ? 0 : c1 + 1);
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float4 av = a[kb & 1][i];
                        const int wi = (c * (KC / 8) + kb) * 4;
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w[wi + 0], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w[wi + 1], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w[wi + 2], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w[wi + 3], acc[i], 0, 0, 0);
                    }
                }
            }
            if (SCHED == 1) {                                    // staging after the chunk's MFMAs (the partner wave covers it)
                stage(nxt, t1, c1);
                fetch(last2 ? t1 + gridDim.x : t1, last2 ? 0 : c1 + 1);
            }
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        // ---- epilogue straight from the accumulators: column on the lane, 128 contiguous bytes per half-wave and register
        {
            float s0 = 0.f, s1 = 0.f;
            float *yb = Y + (tile * BM + rs * TM * 32) * (int64_t)ldy;     // wave-uniform base, 32-bit offsets below
            const int rows_left = (int)((P - (tile * BM + rs * TM * 32)) < (int64_t)(TM * 32) ? (P - (tile * BM + rs * TM * 32)) : (int64_t)(TM * 32));
            // the lane offset is made opaque once per tile: loop-invariant store addresses would otherwise be hoisted out of
            // the tile loop into 64 x 2 registers (and spilled)
            unsigned off = (unsigned)(4 * lh) * (unsigned)ldy + (unsigned)n;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                    const float y = acc[i][r] + bj;
                    if (row < rows_left && n < N) {
                        if (!(ABL & 4) || y == 123456.f) __builtin_nontemporal_store(y, yb + off);
                        s0 += y;
                        s1 = __builtin_fmaf(y, y, s1);
                    }
                    off += (r & 3) == 3 ? 5u * (unsigned)ldy : (unsigned)ldy;     // rows (r&3) + 8 (r>>2): +1, +1, +1, +5
                }
            }
            st0 += (double)s0; st1 += (double)s1;
        }
        tile += gridDim.x;
    }
    if (stats != nullptr) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (lh == 0 && n < N) {
            atomicAdd(stats + n, st0);
            atomicAdd(stats + N + n, st1);
        }
    }
}

template <int KP, int NCB, int RS, int TM, bool ACT, int SCHED, int ABL>
float run(const char *name, const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y, int ldy,
          int64_t P, int K, int N, double *stats, int reps) {
    constexpr int BM = 32 * TM * RS;
    int cus = 256;
    int64_t tiles = (P + BM - 1) / BM;
    unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
    auto k = regw_fwd_kernel<KP, NCB, RS, TM, ACT, SCHED, ABL>;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NCB * RS), 0, 0, X, ldx, aff, W, ldw, bias, Y, ldy, P, K, N, stats);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NCB * RS), 0, 0, X, ldx, aff, W, ldw, bias, Y, ldy, P, K, N, stats);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, tf = 2.0 * P * K * N / (us * 1e-6) / 1e12;
    printf("%-34s P=%-8lld K=%-4d N=%-4d  %8.1f us  %7.2f TF  frac %.3f\n", name, (long long)P, K, N, us, tf, tf / 157.3);
    fflush(stdout);
    return (float)us;
}

static inline int r4(int c) { return (c + 3) & ~3; }

template <int KP, int NCB, int RS, int TM, int SCHED, int ABL = 0>
void test_shape(const char *name, int64_t P, int K, int N, bool check) {
    const int ldx = r4(K), ldy = r4(N);
    std::vector<float> hX((size_t)P * ldx, 0.f), hW((size_t)N * K), hb(N), haff(4 * r4(K), 0.f);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (int64_t p = 0; p < P; ++p) for (int k = 0; k < K; ++k) hX[p * ldx + k] = rnd() * 2.f;
    for (auto &v : hW) v = rnd();
    for (auto &v : hb) v = rnd();
    for (int k = 0; k < K; ++k) { haff[k] = rnd() * 0.1f; haff[r4(K) + k] = 1.f + rnd() * 0.1f; haff[2 * r4(K) + k] = rnd() * 0.1f; }
    float *X, *W, *b, *aff, *Y; double *st;
    CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&aff, haff.size() * 4));
    CK(hipMalloc(&Y, (size_t)P * ldy * 4)); CK(hipMalloc(&st, 2 * N * 8));
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(aff, haff.data(), haff.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(Y, 0, (size_t)P * ldy * 4)); CK(hipMemset(st, 0, 2 * N * 8));
    run<KP, NCB, RS, TM, true, SCHED, ABL>(name, X, ldx, aff, W, K, b, Y, ldy, P, K, N, st, check ? 1 : 20);
    if (check) {
        std::vector<float> hY((size_t)P * ldy);
        CK(hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0.0;
        for (int64_t p = 0; p < P; p += (P > 4096 ? 97 : 1))
            for (int nn = 0; nn < N; ++nn) {
                double acc = hb[nn];
                for (int k = 0; k < K; ++k) {
                    float x = fmaxf(fmaf(hX[p * ldx + k] - haff[k], haff[r4(K) + k], haff[2 * r4(K) + k]), 0.f);
                    acc += (double)x * hW[(size_t)nn * K + k];
                }
                maxerr = fmax(maxerr, fabs(acc - hY[p * ldy + nn]));
            }
        printf("   check %s: max abs err %.3e %s\n", name, maxerr, maxerr < 2e-4 ? "OK" : "FAIL");
    }
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(b)); CK(hipFree(aff)); CK(hipFree(Y)); CK(hipFree(st));
}

int main(int argc, char **argv) {
    const int sel = argc > 1 ? atoi(argv[1]) : -1;     // -1: everything; k: only timing case k (PMC runs)
    int id = 0;
#define CASE(...) do { if (sel < 0 || sel == id) { __VA_ARGS__; } ++id; } while (0)
    if (sel < 0) {
        test_shape<200, 8, 1, 4, 0>("chk 196->256 ncb8 tm4", 1000, 196, 256, true);
        test_shape<128, 7, 1, 4, 0>("chk 128->196 ncb7 tm4", 777, 128, 196, true);
        test_shape<128, 4, 2, 2, 0>("chk 128->128 ncb4 rs2 tm2", 515, 128, 128, true);
    }
    CASE(test_shape<200, 8, 1, 4, 0, 0>("0 196->256 full", 262144, 196, 256, false));
    CASE(test_shape<200, 8, 1, 4, 0, 0>("1 196->256 full P/2", 131072, 196, 256, false));
    CASE(test_shape<200, 8, 1, 4, 0, 0>("2 196->256 full P/4", 65536, 196, 256, false));
    CASE(test_shape<200, 8, 1, 4, 0, 0>("3 196->256 full P/8 (1 tile per WG)", 32768, 196, 256, false));
    CASE(test_shape<200, 8, 1, 4, 0, 15>("4 196->256 MFMA only P/8", 32768, 196, 256, false));
    CASE(test_shape<200, 8, 1, 4, 0, 0>("5 196->256 full 2P", 524288, 196, 256, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("6 128->128 full 32k (1 tile)", 32768, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("7 128->128 full 64k", 65536, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("8 128->128 full 128k", 131072, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("9 128->128 full 256k", 262144, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("10 128->128 full 512k", 524288, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 15>("11 128->128 MFMA only 32k", 32768, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 15>("12 128->128 MFMA only 512k", 524288, 128, 128, false));
    CASE(test_shape<128, 4, 2, 2, 0, 0>("13 128->128 full 256 rows (2 WG)", 256, 128, 128, false));
    return 0;
}
