// Round 6: source-level bisection of tools/exp/pk_probe.hip (same idea: the FPS distances of 8 points per thread formed twice from the same
// centre registers, differences counted).  VAR bits: 1 = path A scalar instead of packed; 2 = path B packed instead of scalar;
// 4 = branch-free accounting; 8 = the centre read right before its use (s_waitcnt, then the arithmetic) instead of an iteration ahead;
// 16 = no running minima; 32 = path B reads the centre registers without a fence copy; 64 = a v_mov_b32 reads x, y, z before path A;
// 128 = a v_mov_b32 reads x before path A; 256 = path A takes y and z from v_mov copies (its packed operands then select the LOW half only);
// 512 = path A takes x from a v_mov copy; 1024 = path B's copies are explicit v_mov_b32 after path A's first result; 2048 = ... behind s_nop 3;
// 4096 = a third path C from global memory through SGPRs: which of A and B differs from it?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -o tools/exp/libpk_probe2.so tools/exp/pk_probe2.hip
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(512) void pk_probe2_kernel(const float *__restrict__ pts, int N, int iters, unsigned *__restrict__ out) {
    extern __shared__ float4 cloud[];
    const int t = threadIdx.x, b = blockIdx.x;
    const float *p = pts + (size_t)b * N * 3;
    float px[8], py[8], pz[8], mda[8], mdb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = (t + i * 512) % N;
        px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2];
        mda[i] = mdb[i] = 1e10f;
        if (t + i * 512 < N) cloud[t + i * 512] = make_float4(px[i], py[i], pz[i], 0.f);
    }
    __syncthreads();
    unsigned bad = 0, bad_ac = 0, bad_bc = 0;
    int far = b % N;
    float4 nxt = cloud[far];
    for (int it = 0; it < iters; ++it) {
        float cx, cy, cz;
        if (VAR & 8) { const float4 c = cloud[far]; cx = c.x; cy = c.y; cz = c.z; }
        else {
            cx = nxt.x; cy = nxt.y; cz = nxt.z;
            nxt = cloud[(far * 7 + 13 + it) % N];
        }
        if (VAR & 64) {                              // a 32-bit read of the three registers before anything else touches them
            float d0, d1, d2;
            asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(d0), "=v"(d1), "=v"(d2) : "v"(cx), "v"(cy), "v"(cz));
            asm volatile("" :: "v"(d0), "v"(d1), "v"(d2));
        }
        if (VAR & 128) {                             // ... of x only
            float d0;
            asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(cx));
            asm volatile("" :: "v"(d0));
        }
        float da[8], db[8];
        auto packed = [&](float qx, float qy, float qz, float *d) {
            const f2 cx2 = {qx, qx}, cy2 = {qy, qy}, cz2 = {qz, qz};
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const f2 x = {px[i], px[i + 1]}, y = {py[i], py[i + 1]}, z = {pz[i], pz[i + 1]};
                const f2 dx = x - cx2, dy = y - cy2, dz = z - cz2;
                const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const f2 dd = (xx + yy) + zz;
                d[i] = dd.x; d[i + 1] = dd.y;
            }
        };
        auto scalar = [&](float qx, float qy, float qz, float *d) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float dx = px[i] - qx, dy = py[i] - qy, dz = pz[i] - qz;
                asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
                float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                asm volatile("" : "+v"(xx), "+v"(yy), "+v"(zz));
                d[i] = (xx + yy) + zz;
            }
        };
        float ax = cx, ay = cy, az = cz;
        if (VAR & 256) { asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(ay), "=&v"(az) : "v"(cy), "v"(cz)); }     // y, z of path A from copies
        if (VAR & 512) { asm volatile("v_mov_b32 %0, %1" : "=&v"(ax) : "v"(cx)); }                                           // x of path A from a copy
        if (VAR & 1) scalar(ax, ay, az, da); else packed(ax, ay, az, da);
        float cxs = cx, cys = cy, czs = cz;
        if (VAR & 1024) {                            // path B's copies made by explicit instructions (after the packed path in program order)
            float d0 = da[0];                        // (ties the statement behind path A's first result)
            if (VAR & 2048) asm volatile("s_nop 3\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(cxs), "=&v"(cys), "=&v"(czs) : "v"(cx), "v"(cy), "v"(cz), "v"(d0));
            else asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(cxs), "=&v"(cys), "=&v"(czs) : "v"(cx), "v"(cy), "v"(cz), "v"(d0));
        } else
        if (!(VAR & 32)) asm volatile("" : "+v"(cxs), "+v"(cys), "+v"(czs));
        if (VAR & 2) { asm volatile("" : "+v"(cxs), "+v"(cys), "+v"(czs)); packed(cxs, cys, czs, db); } else scalar(cxs, cys, czs, db);
        if (VAR & 4096) {                            // path C: the centre from global memory through SGPRs (the form that never failed)
            const int f = __builtin_amdgcn_readfirstlane(far);
            const float gx = p[3 * f], gy = p[3 * f + 1], gz = p[3 * f + 2];
            float dc[8];
            scalar(gx, gy, gz, dc);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bad_ac += __float_as_uint(da[i]) != __float_as_uint(dc[i]) ? 1u : 0u;
                bad_bc += __float_as_uint(db[i]) != __float_as_uint(dc[i]) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VAR & 4) bad += (__float_as_uint(da[i]) ^ __float_as_uint(db[i])) != 0u ? 1u : 0u;
            else if (__float_as_uint(da[i]) != __float_as_uint(db[i])) ++bad;
            if (!(VAR & 16)) { mda[i] = fminf(mda[i], da[i]); mdb[i] = fminf(mdb[i], db[i]); }
        }
        far = (far * 7 + 13 + it) % N;
    }
    if (!(VAR & 16)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bad += __float_as_uint(mda[i]) != __float_as_uint(mdb[i]) ? 0x10000u : 0u;
    }
    if (bad) { atomicAdd(&out[0], bad & 0xFFFFu); atomicAdd(&out[1], bad >> 16); atomicAdd(&out[2], 1u); }
    if (bad_ac) atomicAdd(&out[4], bad_ac);
    if (bad_bc) atomicAdd(&out[5], bad_bc);
    if (t == 0) atomicAdd(&out[3], 1u);
}

template <int VAR>
static int launch(const float *pts, int B, int N, int iters, unsigned *out, hipStream_t s) {
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pk_probe2_kernel<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); raised = true; }
    hipLaunchKernelGGL((pk_probe2_kernel<VAR>), dim3(B), dim3(512), sizeof(float4) * (size_t)N, s, pts, N, iters, out);
    return (int)hipGetLastError();
}

extern "C" int pk_probe2(const float *pts, int B, int N, int iters, int var, unsigned *out, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (var) {
#define V(x) case x: return launch<x>(pts, B, N, iters, out, s);
        V(0) V(1) V(2) V(3) V(4) V(8) V(9) V(10) V(12) V(16) V(20) V(32) V(36) V(5) V(6) V(24) V(28) V(52) V(60) V(64) V(72) V(128) V(136) V(256) V(264) V(512) V(520) V(768) V(776) V(1032) V(3080) V(4104) V(4096)
#undef V
    }
    return -1;
}
