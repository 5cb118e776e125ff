// Which XCD does workgroup L of a (W, B) grid land on?  hipcc --offload-arch=gfx950 -O2 xcc_probe.hip -o xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
    unsigned xcc, cu;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    if (threadIdx.x == 0) { out[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = xcc; out[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = cu; }
    // stay resident a while so that all 64 workgroups are placed like a cooperative launch's
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    unsigned *d, h[2 * 128];
    hipMalloc(&d, sizeof(h));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(8, 8), dim3(1024), 0, 0, d);
        hipMemcpy(h, d, sizeof(unsigned) * 2 * 64, hipMemcpyDeviceToHost);
        printf("grid (8, 8), 1024 threads: XCC_ID (raw & 0xf) by launch index L = y * 8 + x\n");
        for (int L = 0; L < 64; ++L) printf("%u%s", h[2 * L] & 15u, (L & 7) == 7 ? "\n" : " ");
    }
    printf("raw XCC_ID register of L = 0..7: ");
    for (int L = 0; L < 8; ++L) printf("%08x ", h[2 * L]);
    printf("\n");
    return 0;
}
