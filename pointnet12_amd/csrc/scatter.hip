// Scatter-adds of the backward pass turned into gathers.
//
// Two backward steps of the path add rows into a much smaller tensor through an index that comes from the geometry
// alone: the 3-NN interpolation (pointnet_util.py:301, dP2[b, idx[n,k], :] += w[n,k] * dRows[n, :]; 3*N sources onto S
// targets) and the factorised first layer (G[b, idx[p], :] += dY[p, :]; S*K grouped positions onto N source
// points).  As atomics they are the two slowest non-GEMM kernels of a training step (0.25 + 0.38 ms of MSG-SemSeg,
// every target element receives 24..32 same-address adds).  The index is known as soon as the neighbour search
// is done -- on the prefetch stream, one step ahead -- so it is sorted once by target (per cloud: counts -> offsets ->
// member lists, a counting sort) and the backward becomes a SEGMENTED reduction over the sorted member array: every
// lane group takes a chunk of 16 / 32 / 64 consecutive members (by level size), sums runs of equal target in registers and
// flushes a run when the target changes: a run that lies INSIDE the chunk is complete and is STORED (the buffer must arrive
// zeroed all the same: targets without members are never written), only the (at most two) runs that straddle a chunk's ends
// are added atomically.  Perfectly balanced whatever the list lengths are (a
// KITTI-shaped cloud has targets with hundreds of members next to targets with none: one-wave-per-target gathers ran
// 2x SLOWER than the atomics), and ~10x fewer atomics than the element-wise scatter.
#include "pn2_common.h"
#include "bn_tail.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// counts[b, v] += 1 for every entry; one thread per entry
__global__ __launch_bounds__(256) void invert_count_kernel(const int64_t *__restrict__ idx, int M, int T,
                                                           int *__restrict__ counts) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int64_t v = idx[(int64_t)b * M + m];
    if (v >= 0 && v < T) atomicAdd(counts + (int64_t)b * T + v, 1);
}

// exclusive scan of counts[b, 0..T) -> offsets[b, 0..T]; counts is left holding a copy of the offsets (fill cursors).
// One workgroup per cloud, chunks of 1024 values, carry in a register.
__global__ __launch_bounds__(1024) void invert_scan_kernel(int *__restrict__ counts, int T, int *__restrict__ offsets) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int *cnt = counts + (int64_t)b * T;
    int *off = offsets + (int64_t)b * (T + 1);
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int i = base + t;
        const int v = i < T ? cnt[i] : 0;
        int x = v;                                         // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int carry = carry_s;
        const int excl = carry + wbase + x - v;
        if (i < T) { off[i] = excl; cnt[i] = excl; }
        __syncthreads();
        if (t == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (t == 0) off[T] = carry_s;
}

// members[b, pos] = m, owners[b, pos] = v with pos = cursor[b, v]++
__global__ __launch_bounds__(256) void invert_fill_kernel(const int64_t *__restrict__ idx, int M, int T,
                                                          int *__restrict__ cursor, int *__restrict__ members,
                                                          int *__restrict__ owners) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int64_t v = idx[(int64_t)b * M + m];
    if (v < 0 || v >= T) return;
    const int pos = atomicAdd(cursor + (int64_t)b * T + v, 1);
    members[(int64_t)b * M + pos] = m;
    owners[(int64_t)b * M + pos] = (int)v;
}

// The whole counting sort of ONE cloud in ONE workgroup, histogram and cursors in LDS (T <= 16 384 targets: 4 (T + 20) bytes,
// 65 616 at the limit -- the launcher raises the dynamic-LDS attribute above 64 KiB):
// count with ds_add, scan in place, fill through returning ds_add cursors, pad the tail with -1.  One launch instead of
// two fills + three kernels whose global atomics queue up on a few thousand hot counters (a dense scan's 3-NN index
// puts 196 608 entries of a cloud on 1 024 targets: the three passes took 0.37 ms per call, a quarter of the cfg5 SSG
// step).  A cloud keeps one CU busy for ~20 us; the launch runs on the geometry-prefetch stream beside the MLP kernels.
__global__ __launch_bounds__(1024) void invert_lds_kernel(const int64_t *__restrict__ idx, int M, int T,
                                                          int *__restrict__ members, int *__restrict__ owners,
                                                          int *__restrict__ offsets) {
    extern __shared__ int inv_lds[];                      // hist[T], then wsum[16], carry
    int *hist = inv_lds, *wsum = inv_lds + T, *carry_s = wsum + 16;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t *src = idx + (int64_t)b * M;
    int *mem = members + (int64_t)b * M, *own = owners + (int64_t)b * M;
    for (int i = t; i < T; i += 1024) hist[i] = 0;
    if (t == 0) *carry_s = 0;
    __syncthreads();
    for (int m0 = t; m0 < M; m0 += 4096) {                // four independent entries per trip
        int64_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = m0 + 1024 * u < M ? src[m0 + 1024 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] >= 0 && v[u] < T) atomicAdd(&hist[(int)v[u]], 1);
    }
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {          // exclusive scan in place (chunks of 1024, carry in LDS)
        const int i = base + t;
        const int c = i < T ? hist[i] : 0;
        int x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int carry = *carry_s;
        const int excl = carry + wbase + x - c;
        if (i < T) { hist[i] = excl; offsets[(int64_t)b * (T + 1) + i] = excl; }
        __syncthreads();
        if (t == 1023) *carry_s = carry + wbase + x;
        __syncthreads();
    }
    const int total = *carry_s;
    if (t == 0) offsets[(int64_t)b * (T + 1) + T] = total;
    for (int m0 = t; m0 < M; m0 += 4096) {
        int64_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = m0 + 1024 * u < M ? src[m0 + 1024 * u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (v[u] >= 0 && v[u] < T) {
                const int pos = atomicAdd(&hist[(int)v[u]], 1);
                mem[pos] = m0 + 1024 * u;
                own[pos] = (int)v[u];
            }
        }
    }
    for (int i = total + t; i < M; i += 1024) { mem[i] = -1; own[i] = -1; }     // dropped entries: "no member"
}

// Members per lane group: long chunks mean few segments straddle a boundary (those add atomically, the rest is stored), short
// ones mean more lane groups in flight; 64 where that still leaves eight waves per CU, else 32 (four), else 16.  PN2_SEG_CHUNK: A/B.
static int seg_chunk(int64_t members, int lanes_per_row) {
    const int forced = pn2_opt(PN2_OPT_SEG_CHUNK);
    if (forced == 16 || forced == 32 || forced == 64) return forced;
    const int64_t per_cu = (int64_t)pn2_num_cus() * (64 / lanes_per_row);           // lane groups of one wave per CU
    if (members / 64 >= 8 * per_cu) return 64;                                       // (measured on the MSG / SSG level sizes:
    if (members / 32 >= 4 * per_cu) return 32;                                       //  tools/bench_seg.py with PN2_SEG_CHUNK)
    return 16;
}

__device__ __forceinline__ void row_atomic_add(float *dst, int c, int D, float4 v) {
    atomicAdd(dst + c, v.x);
    if (c + 1 < D) atomicAdd(dst + c + 1, v.y);
    if (c + 2 < D) atomicAdd(dst + c + 2, v.z);
    if (c + 3 < D) atomicAdd(dst + c + 3, v.w);
}

// A segment (the members of one owner) that lies entirely inside one chunk has a single writer: its sum is STORED (the
// buffer is zeroed for the owners without members and for the segments that straddle a chunk boundary, which still add
// atomically).  With 64-member chunks over ~12-member segments that is 2 atomic rows per chunk instead of 6.
__device__ __forceinline__ void row_flush(float *dst, int c, int D, float4 v, bool exclusive) {
    if (!exclusive) { row_atomic_add(dst, c, D, v); return; }
    dst[c] = v.x;
    if (c + 1 < D) dst[c + 1] = v.y;
    if (c + 2 < D) dst[c + 2] = v.z;
    if (c + 3 < D) dst[c + 3] = v.w;
}

// 3-NN interpolation backward: dP2[b, s, :] += w[b, n, k] * dRows[b*N + n, col0 + :] for the members m = 3n + k of s.
// LPR lanes (a power of two) hold one row as float4s; each lane group walks one chunk of the target-sorted members.
__global__ __launch_bounds__(256) void three_interp_bwd_seg_kernel(const float *__restrict__ grad_out, int ld, int col0,
                                                                   const int *__restrict__ members,
                                                                   const int *__restrict__ owners,
                                                                   const float *__restrict__ w, int N, int S, int D,
                                                                   int lpr_log2, int chunk, int chunks_per_cloud, int64_t chunks,
                                                                   float *__restrict__ grad_points2) {
    const int lane = threadIdx.x & 63;
    const int LPR = 1 << lpr_log2;
    const int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 >> lpr_log2) + (lane >> lpr_log2);
    const int sub = lane & (LPR - 1);
    if (q >= chunks) return;
    const int64_t b = q / chunks_per_cloud;
    const int M = N * 3;
    const int e0 = (int)(q - b * chunks_per_cloud) * chunk;
    const int e1 = e0 + chunk < M ? e0 + chunk : M;
    const int *mem = members + b * M, *own = owners + b * M;
    const float *wb = w + b * (int64_t)M;
    const bool vec = ((ld | col0) & 3) == 0;
    const int head = own[e0];
    const bool head_shared = e0 > 0 && own[e0 - 1] == head;       // the chunk starts inside a segment
    const int tail_next = e1 < M ? own[e1] : -1;                   // the owner the next chunk starts with
    for (int c = sub * 4; c < D; c += LPR * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = -1;
        bool first = true;                                         // `cur` is the chunk's first segment
        for (int e = e0; e < e1; e += 4) {
            int m[4], tg[4];
            float wt[4];
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool v = e + u < e1;
                m[u] = v ? mem[e + u] : -1;
                tg[u] = v ? own[e + u] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                wt[u] = 0.f;
                if (m[u] < 0) continue;
                wt[u] = wb[m[u]];
                const float *row = grad_out + (b * N + m[u] / 3) * ld + col0 + c;
                if (vec && c + 3 < D) g[u] = ld4(row);
                else {
                    g[u].x = row[0];
                    if (c + 1 < D) g[u].y = row[1];
                    if (c + 2 < D) g[u].z = row[2];
                    if (c + 3 < D) g[u].w = row[3];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (m[u] < 0) continue;
                if (tg[u] != cur) {
                    if (cur >= 0) {
                        row_flush(grad_points2 + (b * S + cur) * D, c, D, acc, !(first && head_shared));
                        first = false;
                    }
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    cur = tg[u];
                }
                float wu = wt[u];
                PN2_OPAQUE1(wu);                       // (a plain scalar: hipcc packed these fmas with a HIGH-half select of a (weight, weight) pair -- pn2_common.h)
                acc.x = __builtin_fmaf(g[u].x, wu, acc.x); acc.y = __builtin_fmaf(g[u].y, wu, acc.y);
                acc.z = __builtin_fmaf(g[u].z, wu, acc.z); acc.w = __builtin_fmaf(g[u].w, wu, acc.w);
            }
        }
        if (cur >= 0) row_flush(grad_points2 + (b * S + cur) * D, c, D, acc, !(first && head_shared) && cur != tail_next);
    }
}

// Factorised first layer, backward: G[b*N + j, :] += dY[p, :] over the members p of source j,
// dY = c0*dZ + q1*(y - mean) + q0, and dWx[c, a] += dY[p, c] * (xyz_j - centre(p))[a] (per-thread partials, folded per
// workgroup, one atomic per (c, a) and workgroup).  A lane owns a float4 of channels; LPR lanes per member row.
__global__ __launch_bounds__(256) void group_affine_bwd_seg_kernel(const float *__restrict__ dZ, int ldz,
                                                                   const float *__restrict__ Y, int ldy,
                                                                   const float *coef, int ldc,   // (no __restrict__: the prologue writes it)
                                                                   const float *__restrict__ xyz,
                                                                   const float *__restrict__ new_xyz,
                                                                   const int *__restrict__ members,
                                                                   const int *__restrict__ owners, int N, int S, int K,
                                                                   int C, int lpr_log2, int chunk, int chunks_per_cloud,
                                                                   int64_t chunks, float *__restrict__ G, int ldg,
                                                                   float *__restrict__ dWx, int ldwx,
                                                                   float *__restrict__ rep, LazyCoef lc) {
    __shared__ float red[256 * 12];
    lazy_coef_prologue(lc);                            // consumer-side BatchNorm backward (bn_tail.h): `coef` filled here
    const int t = threadIdx.x, lane = t & 63;
    const int LPR = 1 << lpr_log2, GPW = 64 >> lpr_log2;
    const int sub = lane & (LPR - 1);
    const int M = S * K;
    const int C4 = (C + 3) & ~3;
    float wacc[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) wacc[e][0] = wacc[e][1] = wacc[e][2] = 0.f;
    const int c = sub * 4;                                  // C4 <= 4 * LPR: one float4 of channels per lane
    const bool cv = c < C4;
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = c0, q0 = c0, mu = c0;
    if (cv) { c0 = ld4(coef + c); q1 = ld4(coef + ldc + c); q0 = ld4(coef + 2 * ldc + c); mu = ld4(coef + 3 * ldc + c); }
    const int64_t qstride = (int64_t)gridDim.x * 4 * GPW;
    for (int64_t q = ((int64_t)blockIdx.x * 4 + (t >> 6)) * GPW + (lane >> lpr_log2); q < chunks; q += qstride) {
        if (!cv) continue;
        const int64_t b = q / chunks_per_cloud;
        const int e0 = (int)(q - b * chunks_per_cloud) * chunk;
        const int e1 = e0 + chunk < M ? e0 + chunk : M;
        const int *mem = members + b * M, *own = owners + b * M;
        const bool head_shared = e0 > 0 && own[e0 - 1] == own[e0];
        const int tail_next = e1 < M ? own[e1] : -1;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = -1;
        bool first = true;
        for (int e = e0; e < e1; e += 4) {
            int m[4], src[4];
            float4 dz[4], y[4];
            float dx[4], dy_[4], dzc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool v = e + u < e1;
                m[u] = v ? mem[e + u] : -1;
                src[u] = v ? own[e + u] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dz[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                y[u] = mu;
                dx[u] = dy_[u] = dzc[u] = 0.f;
                if (m[u] < 0) continue;
                const int64_t p = b * M + m[u];
                dz[u] = ld4(dZ + p * ldz + c);
                y[u] = ld4(Y + p * ldy + c);
                const float *qp = xyz + (b * N + src[u]) * 3;
                const float *ctr = new_xyz + (b * S + m[u] / K) * 3;
                dx[u] = qp[0] - ctr[0]; dy_[u] = qp[1] - ctr[1]; dzc[u] = qp[2] - ctr[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (m[u] < 0) continue;
                if (src[u] != cur) {
                    if (cur >= 0) {
                        row_flush(G + (b * N + cur) * ldg, c, C4, acc, !(first && head_shared));
                        first = false;
                    }
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    cur = src[u];
                }
                float d[4];
                d[0] = __builtin_fmaf(c0.x, dz[u].x, __builtin_fmaf(q1.x, y[u].x - mu.x, q0.x));
                d[1] = __builtin_fmaf(c0.y, dz[u].y, __builtin_fmaf(q1.y, y[u].y - mu.y, q0.y));
                d[2] = __builtin_fmaf(c0.z, dz[u].z, __builtin_fmaf(q1.z, y[u].z - mu.z, q0.z));
                d[3] = __builtin_fmaf(c0.w, dz[u].w, __builtin_fmaf(q1.w, y[u].w - mu.w, q0.w));
                acc.x += d[0]; acc.y += d[1]; acc.z += d[2]; acc.w += d[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    wacc[k][0] = __builtin_fmaf(d[k], dx[u], wacc[k][0]);
                    wacc[k][1] = __builtin_fmaf(d[k], dy_[u], wacc[k][1]);
                    wacc[k][2] = __builtin_fmaf(d[k], dzc[u], wacc[k][2]);
                }
            }
        }
        if (cur >= 0) row_flush(G + (b * N + cur) * ldg, c, C4, acc, !(first && head_shared) && cur != tail_next);
    }
    // fold dWx over the workgroup: threads with the same `sub` own the same four channels
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a) red[t * 12 + k * 3 + a] = wacc[k][a];
    __syncthreads();
    if (t < LPR && cv) {
        // All resident workgroups finish together and each adds 3*C values: straight into dWx that is up to 1024
        // same-address atomics per word on a handful of cache lines (280 us of a 350 us launch when dWx is a dense
        // [C,3] block).  With a scratch block the adds go to one of PN2_DWX_REPLICAS copies (32x less contention per
        // line) and dwx_fold_kernel sums the copies into dWx.
        float *dst = rep ? rep + (size_t)(blockIdx.x % PN2_DWX_REPLICAS) * 3 * C4 : nullptr;
        for (int k = 0; k < 4; ++k) {
            if (c + k >= C) break;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int r = t; r < 256; r += LPR) {
                s0 += red[r * 12 + k * 3]; s1 += red[r * 12 + k * 3 + 1]; s2 += red[r * 12 + k * 3 + 2];
            }
            float *o = dst ? dst + (c + k) * 3 : dWx + (int64_t)(c + k) * ldwx;
            atomicAdd(o, s0);
            atomicAdd(o + 1, s1);
            atomicAdd(o + 2, s2);
        }
    }
}

__global__ __launch_bounds__(256) void dwx_fold_kernel(const float *__restrict__ rep, int C, int C4,
                                                       float *__restrict__ dWx, int ldwx) {
    for (int j = threadIdx.x; j < 3 * C; j += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < PN2_DWX_REPLICAS; ++r) s += rep[(size_t)r * 3 * C4 + j];
        atomicAdd(dWx + (int64_t)(j / 3) * ldwx + j % 3, s);
    }
}

}  // namespace

extern "C" {

int pn2_invert_index(const int64_t *idx, int B, int M, int T, int32_t *members, int32_t *owners, int32_t *scratch,
                     pn2_stream_t stream) {
    PN2_CHECK_ARG(idx && members && owners && scratch && B > 0 && M > 0 && T > 0 && B <= 65535);
    hipStream_t s = pn2_s(stream);
    int32_t *counts = scratch, *offsets = scratch + (size_t)B * T;        // scratch: int32 [B, 2T + 1]
    if (T <= 16384) {                                                      // one LDS-resident pass per cloud
        // T = 16 384 (sa1 of the dense scans) needs 65 616 bytes: above the default 64 KiB dynamic-LDS window
        static Pn2PerDevice raised;
        if (sizeof(int) * ((size_t)T + 20) > 64 * 1024 &&
            pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&invert_lds_kernel), raised) != PN2_OK)
            return PN2_ELAUNCH;
        hipLaunchKernelGGL(invert_lds_kernel, dim3((unsigned)B), dim3(1024), sizeof(int) * ((size_t)T + 20), s, idx, M, T, members,
                           owners, offsets);
        return pn2_launch_status();
    }
    pn2_fill_u32(counts, 0u, (int64_t)B * T, s);
    // out-of-range entries are dropped: their slots at the end of a cloud's member array must read "no member"
    pn2_fill_u32(members, 0xFFFFFFFFu, (int64_t)B * M, s);
    const dim3 grid((unsigned)pn2_cdiv(M, 256), (unsigned)B);
    hipLaunchKernelGGL(invert_count_kernel, grid, dim3(256), 0, s, idx, M, T, counts);
    hipLaunchKernelGGL(invert_scan_kernel, dim3((unsigned)B), dim3(1024), 0, s, counts, T, offsets);
    hipLaunchKernelGGL(invert_fill_kernel, grid, dim3(256), 0, s, idx, M, T, counts, members, owners);
    return pn2_launch_status();
}

int pn2_three_interp_bwd_seg(const float *grad_out, int ld, int col0, const int32_t *members, const int32_t *owners,
                             const float *weight, int B, int N, int S, int D, float *grad_points2, pn2_stream_t stream) {
    PN2_CHECK_ARG(grad_out && members && owners && weight && grad_points2 && B > 0 && N > 0 && S > 0 && D > 0 && col0 >= 0 &&
                  ld >= col0 + D);
    int lpr_log2 = 0;
    while ((4 << lpr_log2) < D && lpr_log2 < 6) ++lpr_log2;       // lanes per row (float4 each), at most a wave
    const int chunk = seg_chunk((int64_t)B * N * 3, 1 << lpr_log2);
    const int cpc = (int)pn2_cdiv((int64_t)N * 3, chunk);
    const int64_t chunks = (int64_t)B * cpc;
    const int64_t waves = pn2_cdiv(chunks, 64 >> lpr_log2);
    hipLaunchKernelGGL(three_interp_bwd_seg_kernel, dim3((unsigned)pn2_cdiv(waves, 4)), dim3(256), 0, pn2_s(stream), grad_out, ld,
                       col0, members, owners, weight, N, S, D, lpr_log2, chunk, cpc, chunks, grad_points2);
    return pn2_launch_status();
}

int pn2_group_affine_bwd_seg(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *xyz,
                             const float *new_xyz, const int32_t *members, const int32_t *owners, int B, int N, int S,
                             int K, int C, float *G, int ldg, float *dWx, int ldwx, float *dwx_scratch,
                             const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(dZ && Y && coef && xyz && new_xyz && members && owners && G && dWx && B > 0 && N > 0 && S > 0 && K > 0 &&
                  C > 0 && C <= 256 && lazy_coef_ok(coef_lazy, coef, C));
    PN2_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && ldg % 4 == 0 && ldg >= ((C + 3) & ~3) && ldwx >= 3);
    int lpr_log2 = 0;
    while ((4 << lpr_log2) < C && lpr_log2 < 6) ++lpr_log2;
    const int C4 = (C + 3) & ~3;
    const int chunk = seg_chunk((int64_t)B * S * K, 1 << lpr_log2);
    const int cpc = (int)pn2_cdiv((int64_t)S * K, chunk);
    const int64_t chunks = (int64_t)B * cpc;
    int64_t blocks = pn2_cdiv(pn2_cdiv(chunks, 64 >> lpr_log2), 4);
    if (blocks > 1024) blocks = 1024;                  // every workgroup ends with 3*C atomics for dWx
    hipLaunchKernelGGL(group_affine_bwd_seg_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), dZ, ldz, Y, ldy, coef,
                       C4, xyz, new_xyz, members, owners, N, S, K, C, lpr_log2, chunk, cpc, chunks, G, ldg, dWx, ldwx, dwx_scratch,
                       make_lazy_coef(coef_lazy));
    if (dwx_scratch)
        hipLaunchKernelGGL(dwx_fold_kernel, dim3(1), dim3(256), 0, pn2_s(stream), dwx_scratch, C, C4, dWx, ldwx);
    return pn2_launch_status();
}

}  // extern "C"
