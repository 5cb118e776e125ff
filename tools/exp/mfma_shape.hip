// Which fp32 MFMA shape holds the higher clock under a REALISTIC operand stream?  (MI355X_MICROARCH.md, DVFS give-back item 7: on
// bf16 the 16x16x32 loop delivered 1.12 - 1.15x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP.)  The register-
// stationary GEMMs of csrc/mlp_wide.hip hold 2.06 GHz in their tile loop (tools/stamp_wide.py) against the 2.39 GHz of a bare
// MFMA loop on operands that never change (tools/exp/mfma_peak.hip).  Here: the same per-wave work as those kernels' inner loop --
// a 128 x 32 output tile per wave, the A operand re-read from LDS by ds_read_b128 (random data, a fresh 32 KB chunk image every
// step), the B operand (weights) in 64 registers -- with v_mfma_f32_32x32x2_f32 and with v_mfma_f32_16x16x4_f32: identical LDS
// bytes, registers and FLOPs, twice the MFMA instructions at half the passes.  No global traffic in the loop.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_shape.hip -o tools/exp/mfma_shape && tools/exp/mfma_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

extern __shared__ __attribute__((aligned(16))) float lds[];

// SHAPE 0: 32x32x2, 1: 16x16x4.  LDSREAD: operand A from LDS each k block (1) or held in registers (0).
template <int SHAPE, int LDSREAD>
__global__ __launch_bounds__(512) void loop_kernel(const float *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ stamps, int steps) {
    constexpr int KC = 64, BM = 128;                                    // chunk image: BM rows x KC floats, 16-byte slots XOR-swizzled
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float4 *lds4 = reinterpret_cast<float4 *>(lds);
    for (int i = t; i < 4 * BM * KC / 4; i += 512) lds4[i] = reinterpret_cast<const float4 *>(in)[(i * 7 + blockIdx.x) & 16383];
    float w[64];                                                        // K = 128 deep weights of this wave's 32 columns
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = in[(t * 64 + i * 13 + wave) & 65535];
    __syncthreads();
    float s = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const int l31 = lane & 31, lh = lane >> 5;
        const unsigned arow = (unsigned)l31 * (KC / 4), ay = (unsigned)lh ^ (unsigned)(l31 & 15);
        float4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds4[arow + i * 32 * (KC / 4) + ay];
        for (int st2 = 0; st2 < steps; st2 += 2)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {                                    // (two steps per trip: static weight-register indices)
            const int st = st2 + hs;
            const unsigned slot = (unsigned)(st & 3) * (BM * KC / 4);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                if (LDSREAD) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = lds4[slot + arow + i * 32 * (KC / 4) + ((unsigned)(2 * kb) ^ ay)];
                }
                const int wi = (hs * 8 + kb) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, w[wi + 0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, w[wi + 1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, w[wi + 2], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, w[wi + 3], acc[i], 0, 0, 0);
                }
            }
            if ((st & 63) == 63) {                                      // keep the sums finite: fold and restart
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s += acc[j][r] * 1e-9f; acc[j][r] = 0.f; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        // 16x16x4: A[m = lane % 16][k = lane / 16], B[k = lane / 16][n = lane % 16].  A float4 of row m at k quad (lane / 16) of a
        // 16-deep block feeds four MFMAs (k = e, 4 + e, 8 + e, 12 + e).  Output tile 128 x 32 = 8 row sixteenths x 2 column halves.
        f32x4 acc[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int l15 = lane & 15, lq = lane >> 4;
        const unsigned arow = (unsigned)l15 * (KC / 4), ay = (unsigned)lq ^ (unsigned)l15;       // 16 distinct rows per service group
        float4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = lds4[arow + i * 16 * (KC / 4) + ay];
        for (int st2 = 0; st2 < steps; st2 += 2)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            const int st = st2 + hs;
            const unsigned slot = (unsigned)(st & 3) * (BM * KC / 4);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {                             // 16-deep k blocks
                if (LDSREAD) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = lds4[slot + arow + i * 16 * (KC / 4) + ((unsigned)(4 * kb) ^ ay)];
                }
                const int wi = (hs * 4 + kb) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, w[wi + 4 * j + 0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, w[wi + 4 * j + 1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, w[wi + 4 * j + 2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, w[wi + 4 * j + 3], acc[i][j], 0, 0, 0);
                    }
            }
            if ((st & 63) == 63) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { s += acc[i][j][r] * 1e-9f; acc[i][j][r] = 0.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 512 + t] = s;
    if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE, int LDSREAD>
int run(const char *label, int steps, float *in, float *out, unsigned long long *stamps, int cus) {
    auto k = loop_kernel<SHAPE, LDSREAD>;
    const size_t ldsb = 4 * 128 * 64 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL(k, dim3(cus), dim3(512), ldsb, 0, in, out, stamps, steps);      // ~2 s of warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), ldsb, 0, in, out, stamps, steps);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * cus);
    CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz(cus), cyc(cus);
    for (int i = 0; i < cus; ++i) { ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; cyc[i] = (double)h[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = (double)cus * 8.0 * steps * 128.0 * 32.0 * 64.0 * 2.0;               // 8 waves x (128 x 32 tile, 64-deep) per step
    const double pipe_cycles = (double)steps * 2.0 * (128.0 * 32.0 * 64.0 * 2.0 / 4096.0) * 64.0;   // two waves per SIMD, 4096 FLOP per 64 cycles
    printf("%-46s %8.3f ms  %7.2f TFLOP/s  in-kernel clock %.3f GHz (min %.3f max %.3f)  pipe busy %.3f\n", label, ms,
           flop / (ms * 1e-3) / 1e12, ghz[cus / 2], ghz.front(), ghz.back(), pipe_cycles / cyc[cus / 2]);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *in, *out; unsigned long long *stamps;
    std::vector<float> h(65536);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&out, (size_t)cus * 512 * 4)); CK(hipMalloc(&stamps, (size_t)cus * 16));
    CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
    const int steps = 2000;                                                 // ~7 ms per launch
    printf("device %s, %d CUs\n", prop.gcnArchName, cus);
    run<0, 0>("32x32x2, A in registers", steps, in, out, stamps, cus);
    run<1, 0>("16x16x4, A in registers", steps, in, out, stamps, cus);
    run<0, 1>("32x32x2, A re-read from LDS (ds_read_b128)", steps, in, out, stamps, cus);
    run<1, 1>("16x16x4, A re-read from LDS (ds_read_b128)", steps, in, out, stamps, cus);
    run<0, 1>("32x32x2, A re-read from LDS (again)", steps, in, out, stamps, cus);
    run<1, 1>("16x16x4, A re-read from LDS (again)", steps, in, out, stamps, cus);
    return 0;
}
