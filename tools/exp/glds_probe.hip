// Probe: global -> LDS by LDS-DMA (global_load_lds_dwordx4 issued from inline asm, M0 = wave-uniform LDS byte address, the
// lane's 16 bytes land at M0 + 16 * lane), counted s_waitcnt vmcnt, raw s_barrier.  Copies n pieces of 1 KiB through a 2-slot
// LDS ring and writes them back; the host checks the bytes.   hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void probe(const float *__restrict__ src, float *__restrict__ dst, int pieces_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float lds[];                 // [2 slots][4 waves][256 floats]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)lds;                        // LDS byte address of the array (address space 3 offset)
    auto issue = [&](int i) {
        const float *g = src + ((size_t)(blockIdx.x * pieces_per_wave + i) * 4 + wave) * 256 + lane * 4;
        glds16(g, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((i & 1) * 4 + wave) * 1024)));
    };
    issue(0);
    for (int i = 0; i < pieces_per_wave; ++i) {
        if (i + 1 < pieces_per_wave) { issue(i + 1); asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // read ANOTHER wave's piece (cross-wave visibility behind the barrier)
        const int ow = (wave + 1) & 3;
        const float4 v = *reinterpret_cast<const float4 *>(&lds[((i & 1) * 4 + ow) * 256 + lane * 4]);
        *reinterpret_cast<float4 *>(dst + ((size_t)(blockIdx.x * pieces_per_wave + i) * 4 + ow) * 256 + lane * 4) = v;
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");     // (the store; keeps the next DMA's count simple: at most the DMA in flight)
        asm volatile("s_barrier" ::: "memory");              // slot free before it is refilled two trips later
    }
}

int main() {
    const int blocks = 64, ppw = 16;
    const size_t n = (size_t)blocks * ppw * 4 * 256;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003) * 0.5f;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, n * 4);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 2 * 4 * 1024, 0, a, b, ppw);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> r(n);
    hipMemcpy(r.data(), b, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += r[i] != h[i];
    printf("glds probe: %zu of %zu floats differ\n", bad, n);
    return bad != 0;
}
