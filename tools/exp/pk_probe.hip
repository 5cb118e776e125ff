// Round 6: is a packed-fp32 VALU result (v_pk_add_f32 / v_pk_mul_f32) ever wrong beside another kernel's bf16 MFMAs?
// Every thread forms the FPS distance ((dx*dx + dy*dy) + dz*dz) of 8 points to a moving centre twice -- packed (two points per
// instruction, as fps_kernel does) and scalar (fenced copies of the operands) -- and counts the bit differences.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC -o tools/exp/libpk_probe.so tools/exp/pk_probe.hip
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool USE_LDS, int MODE>
__global__ __launch_bounds__(512) void pk_probe_kernel(const float *__restrict__ pts, int N, int iters, unsigned *__restrict__ out) {
    extern __shared__ float4 cloud[];
    const int t = threadIdx.x, b = blockIdx.x;
    const float *p = pts + (size_t)b * N * 3;
    float px[8], py[8], pz[8], mdp[8], mds[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = (t + i * 512) % N;
        px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2];
        mdp[i] = mds[i] = 1e10f;
        if (USE_LDS && t + i * 512 < N) cloud[t + i * 512] = make_float4(px[i], py[i], pz[i], 0.f);
    }
    if (USE_LDS) __syncthreads();
    unsigned bad_d = 0, bad_md = 0, first_it = 0xFFFFFFFFu, first_slot = 0;
    float first_dp = 0.f, first_ds = 0.f; int first_far = 0, first_prev = 0, prev_far = -1;
    int far = b % N;
    float4 nxt = USE_LDS ? cloud[far] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        float cx, cy, cz;
        if (USE_LDS) {
            if (MODE == 11) {                               // the read of THIS centre was issued an iteration ago
                cx = nxt.x; cy = nxt.y; cz = nxt.z;
                const int far_n = (far * 7 + 13 + it) % N;
                nxt = cloud[far_n];
            } else if (MODE == 3) {                                // three 4-byte reads
                const volatile float *vc = reinterpret_cast<const volatile float *>(cloud + far);
                cx = vc[0]; cy = vc[1]; cz = vc[2];
            } else if (MODE == 5) {                         // only lane 0 reads; the others take it through SGPRs
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((t & 63) == 0) c = cloud[far];
                cx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c.x)));
                cy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c.y)));
                cz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c.z)));
            } else {
                float4 c = cloud[far];
                if (MODE == 4) asm volatile("" : "+v"(c.w));                                   // ds_read_b128
                if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(c.x), "+v"(c.y), "+v"(c.z));
                if (MODE == 6) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 3" : "+v"(c.x), "+v"(c.y), "+v"(c.z));
                if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0" : "+v"(c.x), "+v"(c.y), "+v"(c.z));
                if (MODE == 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" : "+v"(c.x), "+v"(c.y), "+v"(c.z));
                if (MODE == 9) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c.x), "+v"(c.y), "+v"(c.z));
                cx = c.x; cy = c.y; cz = c.z;
                if (MODE == 2) {                            // all lanes read (broadcast), lane 0's copy through SGPRs
                    cx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cx)));
                    cy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cy)));
                    cz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cz)));
                }
            }
        }
        else { const int f = __builtin_amdgcn_readfirstlane(far); cx = p[3 * f]; cy = p[3 * f + 1]; cz = p[3 * f + 2];
               if (MODE == 10) asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz)); }
        const f2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
        float dp[8], ds[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f2 x = {px[i], px[i + 1]}, y = {py[i], py[i + 1]}, z = {pz[i], pz[i + 1]};
            const f2 dx = x - cx2, dy = y - cy2, dz = z - cz2;
            const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
            const f2 d = (xx + yy) + zz;
            dp[i] = d.x; dp[i + 1] = d.y;
        }
        float cxs = cx, cys = cy, czs = cz;
        asm volatile("" : "+v"(cxs), "+v"(cys), "+v"(czs));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float dx = px[i] - cxs, dy = py[i] - cys, dz = pz[i] - czs;
            asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
            float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            asm volatile("" : "+v"(xx), "+v"(yy), "+v"(zz));
            ds[i] = (xx + yy) + zz;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (__float_as_uint(dp[i]) != __float_as_uint(ds[i])) {
                ++bad_d;
                if (first_it == 0xFFFFFFFFu) { first_it = (unsigned)it; first_slot = (unsigned)i; first_dp = dp[i]; first_ds = ds[i]; first_far = far; first_prev = prev_far; }
            }
            mdp[i] = fminf(mdp[i], dp[i]);
            mds[i] = fminf(mds[i], ds[i]);
        }
        prev_far = far;
        far = (far * 7 + 13 + it) % N;                   // (uniform: every thread computes the same next centre)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) bad_md += __float_as_uint(mdp[i]) != __float_as_uint(mds[i]) ? 1u : 0u;
    if (bad_d || bad_md) {
        atomicAdd(&out[0], bad_d);
        atomicAdd(&out[1], bad_md);
        const unsigned k = atomicAdd(&out[2], 1u);
        if (k < 32) { out[8 + 4 * k] = (unsigned)b; out[9 + 4 * k] = (unsigned)t; out[10 + 4 * k] = first_it; out[11 + 4 * k] = first_slot;
                      out[136 + 4 * k] = __float_as_uint(first_dp); out[137 + 4 * k] = __float_as_uint(first_ds); out[138 + 4 * k] = (unsigned)first_far; out[139 + 4 * k] = (unsigned)first_prev; }
    }
    if (t == 0) atomicAdd(&out[3], 1u);
}

template <int MODE>
static int launch_lds(const float *pts, int B, int N, int iters, unsigned *out, hipStream_t s) {
    const size_t lds = sizeof(float4) * (size_t)N;
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&pk_probe_kernel<true, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); raised = true; }
    hipLaunchKernelGGL((pk_probe_kernel<true, MODE>), dim3(B), dim3(512), lds, s, pts, N, iters, out);
    return (int)hipGetLastError();
}

// use_lds: 0 = the centre from global memory; 1 + MODE = from the LDS mirror (MODE 0 plain ds_read_b96, 1 wait + 16 idle states,
// 2 lane 0's copy through SGPRs, 3 three ds_read_b32, 4 ds_read_b128, 5 only lane 0 reads, 6 wait + 4 idle states)
extern "C" int pk_probe(const float *pts, int B, int N, int iters, int use_lds, unsigned *out, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (use_lds) {
        case 0: hipLaunchKernelGGL((pk_probe_kernel<false, 0>), dim3(B), dim3(512), 0, s, pts, N, iters, out); return (int)hipGetLastError();
        case 1: return launch_lds<0>(pts, B, N, iters, out, s);
        case 2: return launch_lds<1>(pts, B, N, iters, out, s);
        case 3: return launch_lds<2>(pts, B, N, iters, out, s);
        case 4: return launch_lds<3>(pts, B, N, iters, out, s);
        case 5: return launch_lds<4>(pts, B, N, iters, out, s);
        case 6: return launch_lds<5>(pts, B, N, iters, out, s);
        case 7: return launch_lds<6>(pts, B, N, iters, out, s);
        case 8: return launch_lds<7>(pts, B, N, iters, out, s);
        case 9: return launch_lds<8>(pts, B, N, iters, out, s);
        case 10: return launch_lds<9>(pts, B, N, iters, out, s);
        case 11: hipLaunchKernelGGL((pk_probe_kernel<false, 10>), dim3(B), dim3(512), 0, s, pts, N, iters, out); return (int)hipGetLastError();
        case 12: return launch_lds<11>(pts, B, N, iters, out, s);
    }
    return -1;
}

// An LDS hammer to run beside the probe: every wave reads and writes 16-byte words of its workgroup's 16 KiB for `iters` rounds.
__global__ __launch_bounds__(256) void lds_hammer_kernel(float *sink, int iters) {
    __shared__ float4 buf[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 256) buf[i] = make_float4((float)i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = buf[(t * 5 + it + 131 * k) & 1023];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        buf[(t + it) & 1023] = acc;
    }
    if (acc.x == 12345.678f) sink[t] = acc.y + acc.z + acc.w;
}
extern "C" int lds_hammer(float *sink, int blocks, int iters, void *stream) {
    hipLaunchKernelGGL(lds_hammer_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sink, iters);
    return (int)hipGetLastError();
}
