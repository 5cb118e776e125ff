// What does the fp32 matrix pipe of THIS pool's MI355X deliver, and at which clock?  (VERDICT round 3 #5a: settle the ceiling.)
//
// Barrier-free, memory-free loops of v_mfma_f32_32x32x2_f32 on random operands held in registers, four independent accumulators
// per wave, one workgroup per CU: first one wave per SIMD (256 threads), then two (512 threads).  Each launch runs >= 10 ms so
// that the clock the chip settles on under the load is what is measured (MI355X_MICROARCH.md, DVFS give-back item 6):
//   * TFLOP/s from HIP events around the launch,
//   * the in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped once around the loop, median over the
//     workgroups (the stamps go to a buffer of their own),
//   * cycles per MFMA per SIMD = loop cycles / MFMAs issued on that SIMD (64 = back-to-back issue).
// A third variant interleaves a workgroup barrier every 64 MFMAs per wave (the interval of the register-stationary GEMMs).
//
//   hipcc -O3 --offload-arch=gfx950 tools/exp/mfma_peak.hip -o tools/exp/mfma_peak && tools/exp/mfma_peak [seconds_warm]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int BARRIER_EVERY>
__global__ void mfma_loop(const float *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ stamps, int iters) {
    const int t = threadIdx.x;
    // random operands, different per lane; kept in registers for the whole loop
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 4095]; b[i] = in[(t * 16 + 8 + i) & 4095]; }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {                   // 8 x 4 = 32 MFMAs per trip, four independent chains
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + j) & 7], acc[j], 0, 0, 0);
        }
        if (BARRIER_EVERY > 0 && (it % (BARRIER_EVERY / 32)) == BARRIER_EVERY / 32 - 1) __syncthreads();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + t] = s;               // keeps the chains alive; nobody reads it
    if (t == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BE>
int run(const char *label, int threads, int iters, float *in, float *out, unsigned long long *stamps, int cus) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm: >= 2 s of back-to-back launches on the same random data, then the measured launch
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(mfma_loop<BE>, dim3(cus), dim3(threads), 0, 0, in, out, stamps, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_loop<BE>, dim3(cus), dim3(threads), 0, 0, in, out, stamps, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * cus);
    CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost));
    std::vector<double> ghz(cus), cyc(cus);
    for (int i = 0; i < cus; ++i) { ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; cyc[i] = (double)h[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double waves = threads / 64.0, mfma_per_wave = (double)iters * 32.0;
    const double flop = (double)cus * waves * mfma_per_wave * 32.0 * 32.0 * 2.0 * 2.0;
    const double per_simd = mfma_per_wave * (waves / 4.0);
    printf("%-44s %8.3f ms  %7.2f TFLOP/s  in-kernel clock %.3f GHz (min %.3f max %.3f)  %.2f cycles per MFMA per SIMD\n", label, ms,
           flop / (ms * 1e-3) / 1e12, ghz[cus / 2], ghz.front(), ghz.back(), cyc[cus / 2] / per_simd);
    return 0;
}

int main(int argc, char **argv) {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDevice(&dev));
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.0f MHz (API)\n", prop.gcnArchName, cus, prop.clockRate / 1e3);
    float *in, *out;
    unsigned long long *stamps;
    std::vector<float> h(4096);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;     // uniform [-1, 1): the guide benches on random data
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, (size_t)cus * 1024 * 4)); CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * cus));
    CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    const int iters = argc > 1 ? atoi(argv[1]) : 12000;             // 12 000 x 32 MFMAs x 64 cycles = 24.6 M cycles ~ 10 ms per wave
    if (run<0>("1 wave / SIMD, no barrier", 256, iters, in, out, stamps, cus)) return 1;
    if (run<0>("2 waves / SIMD, no barrier", 512, iters / 2, in, out, stamps, cus)) return 1;
    if (run<64>("2 waves / SIMD, barrier every 64 MFMAs/wave", 512, iters / 2, in, out, stamps, cus)) return 1;
    if (run<128>("2 waves / SIMD, barrier every 128 MFMAs/wave", 512, iters / 2, in, out, stamps, cus)) return 1;
    return 0;
}
