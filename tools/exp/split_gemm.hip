// PROBE (round 5, not product code): an fp32-accurate GEMM on the bf16 matrix pipe.
//
// Where the shared-MLP GEMMs stand (profiles/r05_ring_fwd.txt): v_mfma_f32_32x32x2_f32 tops out at 157 TF,
// the kernels keep the pipe ~88 % busy inside their loops at the ~2.07 GHz the chip holds -- the fp32 pipe IS the limit, and the
// same silicon does 16x that rate on bf16.  Every fp32 number is exactly hi + mid + lo with three bf16 pieces (8 + 8 + 8 mantissa
// bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the subtractions are exact in fp32).  A product a*b is then
//     a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid) + [terms <= 2^-24 |a b|, dropped]
// -- six bf16 MFMAs, each product exact in the fp32 accumulator's input (8 x 8 bits), against sixteen passes' worth of fp32 MFMA:
// 6 x 32 cycles per 32x32x16 block instead of 8 x 64.  The dropped terms are below one fp32 rounding of the product; what is
// measured here: (1) the error of this scheme against an fp64 evaluation next to the error of a plain fp32 fma chain on the
// same data, (2) the time of a forward-shaped kernel (Y = X W^T, rows streamed, weights pre-split in registers) against the
// HBM time of its bytes and against the shipped fp32 kernel (tools/bench_kernels.py fwd) on the same shapes.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/split_gemm.hip -o tools/exp/split_gemm && tools/exp/split_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

// two fp32 values -> three packed bf16 pairs (hi, mid, lo), each pair = one dword
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &mid, unsigned &lo) {
    f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    v = v - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(v, bf16x2);
    v = v - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h); mid = __builtin_bit_cast(unsigned, m); lo = __builtin_bit_cast(unsigned, l);
}

union Frag { bf16x8 v; unsigned u[4]; uint4 q; };

// K: contraction length (multiple of 64), N = 256 (8 waves x 32 columns), 64-row tiles (2 row blocks per wave), persistent.
// TERMS: 6 = the scheme above; 3 = hi*hi + hi*mid + mid*hi only (bf16x2, 16-bit operands); 1 = plain bf16.
template <int K, int TERMS>
__global__ __launch_bounds__(512) void split_fwd_kernel(const float *__restrict__ X, const float *__restrict__ W, float *__restrict__ Y,
                                                        int64_t tiles) {
    constexpr int KC = 64, NCH = K / KC, KB = K / 16, BM = 64;
    constexpr int IMG = BM * KC * 2;                                // bytes of one bf16 component image of a chunk
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int n = wave * 32 + l31;
    // ---- weights: this lane's column n, k = 16 kb + 8 lh + 0..7, pre-split into three fragments per k block
    Frag wh[KB], wm[KB], wl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float4 a = *reinterpret_cast<const float4 *>(W + (size_t)n * K + 16 * kb + 8 * lh);
        const float4 b = *reinterpret_cast<const float4 *>(W + (size_t)n * K + 16 * kb + 8 * lh + 4);
        split2(a.x, a.y, wh[kb].u[0], wm[kb].u[0], wl[kb].u[0]);
        split2(a.z, a.w, wh[kb].u[1], wm[kb].u[1], wl[kb].u[1]);
        split2(b.x, b.y, wh[kb].u[2], wm[kb].u[2], wl[kb].u[2]);
        split2(b.z, b.w, wh[kb].u[3], wm[kb].u[3], wl[kb].u[3]);
    }
    // ---- staging: item i of thread t: float4 quad q = idx % 16 of row idx / 16 (idx = t + 512 i), two items per chunk
    // LDS: [buffer 2][component 3][row 64][128 bytes]; 16-byte slot s of a row holds k block s / 2, half s % 2, at s ^ ((row >> 1) & 7)
    auto lds_off = [](int buf, int comp, int row, int slot) {
        return (unsigned)(buf * 3 * IMG + comp * IMG + row * 128 + 16 * (slot ^ ((row >> 1) & 7)));
    };
    float4 raw[2];
    auto fetch = [&](int64_t tile, int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 512 * i, row = idx >> 4, q = idx & 15;
            raw[i] = *reinterpret_cast<const float4 *>(X + ((size_t)tile * BM + row) * K + c * KC + 4 * q);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 512 * i, row = idx >> 4, q = idx & 15;
            unsigned h0, m0, l0, h1, m1, l1;
            split2(raw[i].x, raw[i].y, h0, m0, l0);
            split2(raw[i].z, raw[i].w, h1, m1, l1);
            const unsigned o = lds_off(buf, 0, row, q >> 1) + 8u * (q & 1);
            *reinterpret_cast<uint2 *>(lds_raw + o) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(lds_raw + o + IMG) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(lds_raw + o + 2 * IMG) = make_uint2(l0, l1);
        }
    };
    int64_t tile = blockIdx.x;
    if (tile >= tiles) return;
    fetch(tile, 0);
    stage(0);
    int buf = 0;
    // next chunk in flight
    {
        const bool last = NCH == 1;
        fetch(last ? (tile + gridDim.x < tiles ? tile + gridDim.x : tile) : tile, last ? 0 : 1);
    }
    while (tile < tiles) {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                        // chunk c is staged in `buf`; everyone is done with buf ^ 1
            stage(buf ^ 1);                                         // the chunk fetched one step ago
            {
                // request the chunk after that
                int c2 = c + 2;
                int64_t t2 = tile;
                while (c2 >= NCH) { c2 -= NCH; t2 += gridDim.x; }
                if (t2 >= tiles) t2 = tile;
                fetch(t2, c2);
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int kg = c * 4 + kb;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 32 + l31;
                    const unsigned o = lds_off(buf, 0, row, 2 * kb + lh);
                    Frag ah, am, al;
                    ah.q = *reinterpret_cast<const uint4 *>(lds_raw + o);
                    if (TERMS >= 3) am.q = *reinterpret_cast<const uint4 *>(lds_raw + o + IMG);
                    if (TERMS >= 6) al.q = *reinterpret_cast<const uint4 *>(lds_raw + o + 2 * IMG);
                    if (TERMS >= 6) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, wh[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wl[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wm[kg].v, acc[i], 0, 0, 0);
                    }
                    if (TERMS >= 3) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wh[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wm[kg].v, acc[i], 0, 0, 0);
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wh[kg].v, acc[i], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
        // epilogue: column on the lane, rows (r & 3) + 8 (r >> 2) + 4 lh
        float *yb = Y + (size_t)tile * BM * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                __builtin_nontemporal_store(acc[i][r], yb + (size_t)row * 256 + n);
            }
        tile += gridDim.x;
    }
}

template <int K, int TERMS>
void run(const char *label, int64_t P, bool check) {
    const int N = 256;
    std::vector<float> hX((size_t)P * K), hW((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto &v : hX) v = rnd() * 3.f + 0.3f * rnd() * rnd();
    for (auto &v : hW) v = rnd() * 0.2f;
    float *X, *W, *Y;
    CK(hipMalloc(&X, hX.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&Y, (size_t)P * N * 4));
    CK(hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    auto k = split_fwd_kernel<K, TERMS>;
    const size_t ldsb = 2 * 3 * 64 * 64 * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int64_t tiles = P / 64;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ldsb, 0, X, W, Y, tiles);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), ldsb, 0, X, W, Y, tiles);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps, bytes = 4.0 * ((double)P * K + (double)P * N);
    printf("%-34s P=%-7lld K=%-4d  %8.1f us  %7.1f fp32-equivalent TF  %6.0f GB/s algorithmic (HBM time at 5.0 TB/s: %.1f us)\n", label,
           (long long)P, K, us, 2.0 * P * K * N / (us * 1e-6) / 1e12, bytes / (us * 1e-6) / 1e9, bytes / 5.0e12 * 1e6);
    if (check) {
        std::vector<float> hY((size_t)P * N);
        CK(hipMemcpy(hY.data(), Y, hY.size() * 4, hipMemcpyDeviceToHost));
        double e_split = 0, e_f32 = 0, ymax = 0, s_split = 0, s_f32 = 0;
        size_t cnt = 0;
        for (int64_t p = 0; p < P; p += 997)
            for (int nn = 0; nn < N; nn += 3) {
                double ref = 0;
                float chain = 0.f;
                for (int kk = 0; kk < K; ++kk) {
                    ref += (double)hX[p * K + kk] * (double)hW[(size_t)nn * K + kk];
                    chain = fmaf(hX[p * K + kk], hW[(size_t)nn * K + kk], chain);
                }
                const double d1 = fabs(hY[p * N + nn] - ref), d2 = fabs((double)chain - ref);
                e_split = fmax(e_split, d1); e_f32 = fmax(e_f32, d2); ymax = fmax(ymax, fabs(ref));
                s_split += d1 * d1; s_f32 += d2 * d2; ++cnt;
            }
        printf("   accuracy over %zu sampled outputs (|y| up to %.2f): this kernel max %.3e rms %.3e   fp32 fma chain max %.3e rms %.3e\n", cnt, ymax,
               e_split, sqrt(s_split / cnt), e_f32, sqrt(s_f32 / cnt));
    }
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y));
}

int main() {
    run<128, 6>("bf16x3 (6 products)  128->256", 131072, true);
    run<128, 3>("bf16x2 (3 products)  128->256", 131072, true);
    run<128, 1>("bf16    (1 product)  128->256", 131072, true);
    run<192, 6>("bf16x3 (6 products)  192->256", 262144, true);
    run<256, 6>("bf16x3 (6 products)  256->256", 262144, true);
    run<128, 6>("bf16x3 (6 products)  128->256", 131072, false);
    run<192, 6>("bf16x3 (6 products)  192->256", 262144, false);
    return 0;
}
