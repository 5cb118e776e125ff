// Index-exact geometry kernels: farthest-point sampling, ball query, 3-NN, pair distances.
//
// COMPILED WITH -ffp-contract=off.  The reference's results depend on the exact fp32
// expression forms (pinned in oracle/pn2_oracle.c against the reference itself):
//   FPS distance   d = ((dx*dx + dy*dy) + dz*dz)            no fused multiply-add
//   pair distance  dot = fma(az,bz, fma(ay,by, ax*bx)); n(p) = ((x*x + y*y) + z*z)
//                  d = ((-2*dot) + n(query)) + n(candidate)
// so every fused step below is an explicit __builtin_fmaf and nothing else may contract.
#include "pn2_common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float sq_norm3(float x, float y, float z) {
    float xx = x * x, yy = y * y, zz = z * z;
    return (xx + yy) + zz;
}

// min of two non-NaN floats as ONE v_min_f32: __builtin_fminf adds a canonicalising v_max_f32 x, x per operand in IEEE
// mode, and the FPS iteration is bound by its instruction count.
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float nq, float px, float py, float pz,
                                           float np) {
    float dot = qx * px;
    dot = __builtin_fmaf(qy, py, dot);
    dot = __builtin_fmaf(qz, pz, dot);
    float d = -2.0f * dot;
    d = d + nq;
    d = d + np;
    return d;
}

// ---------------------------------------------------------------------------------------------
// Farthest-point sampling: one workgroup per cloud, the cloud and its running min-distance
// live in registers (PPT points per thread, point j = t + i*THREADS), the cloud is mirrored in
// LDS as float4 so the winner's coordinates are one broadcast ds_read_b128 away.
// Each of the npoint dependent iterations costs: PPT distance updates, a per-thread argmax,
// a 64-lane DPP max on a packed (distance bits, ~index) key -- distances are >= 0 so their
// fp32 bit patterns order like unsigned ints and "largest key" = largest distance, lowest
// index -- one ds_max_u64 per wave on a shared LDS word and ONE barrier (three words in rotation).
// pointnet_util.py:77-83.
// ---------------------------------------------------------------------------------------------
// PIECE (round 6, option FPS_PIECE: the launch-cadence experiment of HISTORY.md): the npoint iterations as several launches of
// `it1 - it0` iterations each -- the running distances and the current sample travel between the launches through a workspace
// (md_ws [B][PPT][THREADS] floats, far_ws [B] ints).  Same arithmetic in the same order: bit-identical indices.
template <int THREADS, int PPT, bool XYZ_LDS, bool PIECE = false>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float *__restrict__ xyz, int N,
                                                      const int64_t *__restrict__ start, int npoint,
                                                      int64_t *__restrict__ out, int it0 = 0, int it1 = 0,
                                                      float *__restrict__ md_ws = nullptr, int *__restrict__ far_ws = nullptr) {
    constexpr int NW = THREADS / 64;
    extern __shared__ float4 fps_lds[];
    float4 *cloud = fps_lds;                                                     // [N] when XYZ_LDS
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(fps_lds + (XYZ_LDS ? N : 0));  // [3] (+ spare)

    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const float *p = xyz + (size_t)b * N * 3;

    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int j = t + i * THREADS;
        if (j < N) {
            px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2];
            md[i] = (PIECE && it0 > 0) ? md_ws[((size_t)b * PPT + i) * THREADS + t] : 1e10f;
            if (XYZ_LDS) cloud[j] = make_float4(px[i], py[i], pz[i], 0.f);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            md[i] = -1.f;                             // never a candidate: distances are >= 0
        }
    }
    if (NW > 1 && t < 3) slots[t] = 0ull;
    if (XYZ_LDS || NW > 1) __syncthreads();
    int rot = 0;

    // an out-of-range start (only the Python `start=` override can produce one) must not index past the cloud
    int far = (int)(start[b] < 0 ? 0 : (start[b] >= N ? N - 1 : start[b]));
    if (PIECE && it0 > 0) far = far_ws[b];
    int64_t *o = out + (size_t)b * npoint;
    const int it_begin = PIECE ? it0 : 0, it_end = PIECE ? (it1 < npoint ? it1 : npoint) : npoint;
    for (int it = it_begin; it < it_end; ++it) {
        if (t == 0) o[it] = far;
        float cx, cy, cz;
        if (XYZ_LDS) {
            float4 c = cloud[far];
            PN2_OPAQUE3(c.x, c.y, c.z);                            // (no high-half selects in the packed arithmetic below: pn2_common.h)
            cx = c.x; cy = c.y; cz = c.z;
        } else {
            int f = __builtin_amdgcn_readfirstlane(far);
            cx = p[3 * f]; cy = p[3 * f + 1]; cz = p[3 * f + 2];
        }
        if (PPT >= 2) {
            // two points per instruction: v_pk_add_f32 / v_pk_mul_f32 are IEEE single operations on both halves, so
            // ((dx*dx + dy*dy) + dz*dz) keeps its exact un-fused form (the file is built with -ffp-contract=off)
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
            for (int i = 0; i + 1 < PPT; i += 2) {
                const f2 x = {px[i], px[i + 1]}, y = {py[i], py[i + 1]}, z = {pz[i], pz[i + 1]};
                const f2 dx = x - cx2, dy = y - cy2, dz = z - cz2;
                const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const f2 d = (xx + yy) + zz;
                md[i] = min_raw(d.x, md[i]);                      // slots past N hold -1 and stay there
                md[i + 1] = min_raw(d.y, md[i + 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
                float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                float d = (xx + yy) + zz;
                md[i] = min_raw(d, md[i]);
            }
        }
        // per-thread argmax: the maximum first (v_max3_f32 tree), then the LOWEST slot that holds it -- half the
        // compare/select pairs of a running (max, index) chain, and no dependent chain through the maximum
        float bm = md[0];
#pragma unroll
        for (int i = 1; i < PPT; ++i) bm = __builtin_fmaxf(bm, md[i]);
        int bi = PPT - 1;
#pragma unroll
        for (int i = PPT - 2; i >= 0; --i) bi = md[i] == bm ? i : bi;
        const int bj = t + bi * THREADS;
        unsigned long long key = bm < 0.f ? 0ull
                                          : ((unsigned long long)__float_as_uint(bm) << 32) | (0xFFFFFFFFu - (unsigned)bj);
        key = pn2_wave_max_u64_dpp(key);
        if (NW > 1) {
            // the waves meet in ONE LDS word: lane 0 of each issues a ds_max_u64, everybody reads the word back after the
            // barrier (no per-wave slots to reduce again).  Three words in rotation: the next one is cleared here, two
            // barriers after its last reader.
            unsigned long long *cur = slots + rot;
            rot = rot == 2 ? 0 : rot + 1;
            if (lane == 0) __hip_atomic_fetch_max(cur, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (t == 0) slots[rot] = 0ull;
            __syncthreads();
            key = *cur;
        }
        far = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    }
    if (PIECE && it_end < npoint) {                   // hand the state to the next launch
#pragma unroll
        for (int i = 0; i < PPT; ++i) md_ws[((size_t)b * PPT + i) * THREADS + t] = md[i];
        if (t == 0) far_ws[b] = far;
    }
}

// ---------------------------------------------------------------------------------------------
// Farthest-point sampling with exact spatial pruning (2048 < N <= 28 672, one workgroup per cloud).
//
// The plain kernel above updates ALL N running distances in every one of the npoint dependent iterations; on one CU
// that is its whole cost (the iteration is bound by the instruction count of the waves sharing a SIMD: 0.55 us at
// N = 4096, 2.25 us at 28 672).  But a new sample only lowers the distance of points closer to it than their current
// distance, i.e. inside a ball whose radius shrinks like 1/sqrt(i).  So the points are first grouped spatially -- a
// counting sort by Morton cell in LDS, once per launch -- and every WAVE owns one contiguous run of the sorted order
// (64 * PPT points: a compact region) with its bounding box.  Per iteration a wave whose box is farther from the new
// sample than its own largest running distance skips the update and re-offers its cached candidate; after the first
// few dozen samples one to three of the sixteen waves are active.
//
// EXACT: a skipped update is one that provably changes nothing.  The reference distance is d = ((dx*dx + dy*dy) + dz*dz)
// in fp32 (relative error <= 3 ulp-ish, < 4e-7); the box distance lb bounds the true distance from below and is itself
// computed within 4e-7; a wave is skipped only if lb * (1 - 2e-6) > max(md) >= md[j], hence d_fp32(j) >= md[j] for every
// point j of the wave and min(md[j], d) == md[j].  The running distances are therefore bit-identical to the plain
// kernel's, and the argmax key packs (distance bits, ~ORIGINAL index) as before: lowest original index on ties.
// Two points with equal distance inside one THREAD (duplicates -- 9..23 % of a resampled KITTI cloud) take a slow
// path that compares their original indices through the LDS permutation.
// ---------------------------------------------------------------------------------------------
constexpr int FPS_NC = 4096;                          // Morton cells: 5 + 5 bits of (x, y) interleaved, 2 bits of z

__device__ __forceinline__ int fps_cell(float x, float y, float z, float ox, float oy, float oz, float sx, float sy, float sz) {
    int ix = (int)((x - ox) * sx), iy = (int)((y - oy) * sy), iz = (int)((z - oz) * sz);
    ix = ix < 0 ? 0 : (ix > 31 ? 31 : ix);
    iy = iy < 0 ? 0 : (iy > 31 ? 31 : iy);
    iz = iz < 0 ? 0 : (iz > 3 ? 3 : iz);
    unsigned m = 0;
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) m |= (((unsigned)ix >> bit) & 1u) << (2 * bit) | (((unsigned)iy >> bit) & 1u) << (2 * bit + 1);
    return (int)((m << 2) | (unsigned)iz);
}

template <int THREADS, int PPT, bool XYZ_LDS>
__global__ __launch_bounds__(THREADS) void fps_pruned_kernel(const float *__restrict__ xyz, int N,
                                                             const int64_t *__restrict__ start, int npoint,
                                                             int64_t *__restrict__ out) {
    constexpr int NW = THREADS / 64, WCAP = 64 * PPT, CAP = THREADS * PPT;
    extern __shared__ int fps_p_lds[];
    int *perm = fps_p_lds;                               // [CAP]: sorted position -> original index (-1: empty)
    int *hist = perm + CAP;                              // [FPS_NC] counts -> cursors
    int *wsum = hist + FPS_NC;                           // [NW] + carry
    float *box = reinterpret_cast<float *>(wsum + 32);   // [6] cloud bounding box, then [NW][6] wave boxes at box + 8
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(box + 8 + 6 * NW + 2);   // [3] rotating meeting words
    float4 *cloud = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(slots + 4) + 15) & ~(uintptr_t)15);   // XYZ_LDS: [N] by ORIGINAL index -- the winner's coordinates are
                                                             // one broadcast ds_read_b128 away instead of an L2 round trip

    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *p = xyz + (size_t)b * N * 3;

    // ---- cloud bounding box (the cell grid adapts to whatever scale the coordinates have)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = t; j < N; j += THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = p[3 * j + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int m = 32; m >= 1; m >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, 64)); }
    float *wbox = reinterpret_cast<float *>(perm);       // scratch [NW][6] (perm is not in use yet)
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { wbox[wave * 6 + a] = lo[a]; wbox[wave * 6 + 3 + a] = hi[a]; }
    }
    for (int i = t; i < FPS_NC; i += THREADS) hist[i] = 0;
    if (t < 3) slots[t] = 0ull;
    __syncthreads();
    if (t < 6) {
        float v = wbox[t];
        for (int w = 1; w < NW; ++w) v = t < 3 ? fminf(v, wbox[w * 6 + t]) : fmaxf(v, wbox[w * 6 + t]);
        box[t] = v;
    }
    __syncthreads();
    const float ox = box[0], oy = box[1], oz = box[2];
    // (an axis without extent -- a flat cloud -- gets scale 0: everything in cell 0 of that axis, never 0 * inf)
    const float sx = box[3] > ox ? 32.0f / (box[3] - ox) : 0.f, sy = box[4] > oy ? 32.0f / (box[4] - oy) : 0.f,
                sz = box[5] > oz ? 4.0f / (box[5] - oz) : 0.f;
    __syncthreads();                                      // wbox (aliasing perm) is dead

    // ---- counting sort by cell: histogram, exclusive scan, fill through LDS cursors
    for (int j = t; j < N; j += THREADS)
        atomicAdd(&hist[fps_cell(p[3 * j], p[3 * j + 1], p[3 * j + 2], ox, oy, oz, sx, sy, sz)], 1);
    __syncthreads();
    {
        constexpr int CPT = FPS_NC / THREADS;             // consecutive cells per thread
        int c[CPT], sum = 0;
#pragma unroll
        for (int i = 0; i < CPT; ++i) { c[i] = hist[t * CPT + i]; sum += c[i]; }
        int x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int base = x - sum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int i = 0; i < CPT; ++i) { hist[t * CPT + i] = base; base += c[i]; }
    }
    __syncthreads();
    for (int j = t; j < N; j += THREADS) {
        const float x = p[3 * j], y = p[3 * j + 1], z = p[3 * j + 2];
        const int pos = atomicAdd(&hist[fps_cell(x, y, z, ox, oy, oz, sx, sy, sz)], 1);
        perm[pos] = j;
        if (XYZ_LDS) cloud[j] = make_float4(x, y, z, 0.f);
    }
    for (int i = N + t; i < CAP; i += THREADS) perm[i] = -1;
    __syncthreads();

    // ---- this thread's points: sorted positions wave * WCAP + i * 64 + lane (the wave owns one contiguous run)
    float px[PPT], py[PPT], pz[PPT], md[PPT];
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int pbase = wave * WCAP + lane;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = perm[pbase + i * 64];
        if (j >= 0) {
            px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2];
            md[i] = 1e10f;
            blo[0] = fminf(blo[0], px[i]); bhi[0] = fmaxf(bhi[0], px[i]);
            blo[1] = fminf(blo[1], py[i]); bhi[1] = fmaxf(bhi[1], py[i]);
            blo[2] = fminf(blo[2], pz[i]); bhi[2] = fmaxf(bhi[2], pz[i]);
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            md[i] = -1.f;                                 // never a candidate: distances are >= 0
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int m = 32; m >= 1; m >>= 1) { blo[a] = fminf(blo[a], __shfl_xor(blo[a], m, 64)); bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], m, 64)); }
    float *mybox = box + 8 + 6 * wave;                    // kept in LDS, read back as six broadcast loads per iteration:
    if (lane == 0) {                                      // at 24-28 points per thread every register counts
#pragma unroll
        for (int a = 0; a < 3; ++a) { mybox[a] = blo[a]; mybox[3 + a] = bhi[a]; }
    }
    __syncthreads();

    int rot = 0;
    int far = (int)(start[b] < 0 ? 0 : (start[b] >= N ? N - 1 : start[b]));
    int64_t *o = out + (size_t)b * npoint;
    // The candidate a wave offers while it is being skipped must be the one the plain kernel would compute from the same
    // running distances: initially every point sits at 1e10, i.e. (1e10, lowest original index of the wave); an empty
    // wave offers nothing, ever.
    int minidx = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < PPT; ++i) { const int j = perm[pbase + i * 64]; minidx = (j >= 0 && j < minidx) ? j : minidx; }
    for (int m = 32; m >= 1; m >>= 1) { const int o2 = __shfl_xor(minidx, m, 64); minidx = o2 < minidx ? o2 : minidx; }
    unsigned long long wkey = minidx == 0x7FFFFFFF ? 0ull : ((unsigned long long)__float_as_uint(1e10f) << 32) | (0xFFFFFFFFu - (unsigned)minidx);
    float wbest = minidx == 0x7FFFFFFF ? 0.f : 1e10f;
    for (int it = 0; it < npoint; ++it) {
        if (t == 0) o[it] = far;
        float cx, cy, cz;
        if (XYZ_LDS) {
            float4 c = cloud[far];
            PN2_OPAQUE3(c.x, c.y, c.z);
            cx = c.x; cy = c.y; cz = c.z;
        } else {
            const int f = __builtin_amdgcn_readfirstlane(far);
            cx = p[3 * f]; cy = p[3 * f + 1]; cz = p[3 * f + 2];
        }
        // squared distance from the sample to this wave's box, from below
        const float ex = fmaxf(fmaxf(mybox[0] - cx, cx - mybox[3]), 0.f), ey = fmaxf(fmaxf(mybox[1] - cy, cy - mybox[4]), 0.f),
                    ez = fmaxf(fmaxf(mybox[2] - cz, cz - mybox[5]), 0.f);
        const float lb = (ex * ex + ey * ey) + ez * ez;
        if (!(lb * 0.999998f > wbest)) {                  // wave-uniform: the sample may reach into this wave's region
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float d = (xx + yy) + zz;
                md[i] = min_raw(d, md[i]);                // empty slots hold -1 and stay there
            }
            float bm = md[0];
#pragma unroll
            for (int i = 1; i < PPT; ++i) bm = __builtin_fmaxf(bm, md[i]);
            int bi = PPT - 1, hits = 0;
#pragma unroll
            for (int i = PPT - 1; i >= 0; --i) { const bool h = md[i] == bm; bi = h ? i : bi; hits += h ? 1 : 0; }
            int bj = bm < 0.f ? 0 : perm[pbase + bi * 64];
            if (__any(hits > 1 && bm >= 0.f)) {           // equal distances inside one thread: lowest ORIGINAL index wins
                if (hits > 1 && bm >= 0.f) {
#pragma unroll
                    for (int i = 0; i < PPT; ++i)
                        if (md[i] == bm) { const int j = perm[pbase + i * 64]; bj = j < bj ? j : bj; }
                }
            }
            const unsigned long long key = bm < 0.f ? 0ull : ((unsigned long long)__float_as_uint(bm) << 32) | (0xFFFFFFFFu - (unsigned)bj);
            wkey = pn2_wave_max_u64_dpp(key);
            wbest = __uint_as_float((unsigned)(wkey >> 32));      // 0 for an empty wave: skipped from now on
        }
        unsigned long long *cur = slots + rot;
        rot = rot == 2 ? 0 : rot + 1;
        if (lane == 0) __hip_atomic_fetch_max(cur, wkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t == 0) slots[rot] = 0ull;
        __syncthreads();
        far = (int)(0xFFFFFFFFu - (unsigned)(*cur & 0xFFFFFFFFull));
    }
}

// The same kernel with a second level of pruning for 24+ points per thread (N > 22 528), where the one above runs out of
// registers (24-72 of its words went to scratch) and its per-thread argmax over all PPT points is the iteration: the skip test
// is per ROW of 64 consecutive sorted points (lane i of a wave holds row i's bounding box and largest running distance and
// tests it while the other lanes test theirs), only touched rows are updated, four at a time with their four DPP maxima
// interleaved, and the wave's candidate comes from the row maxima (the original index is looked up in the tied rows only:
// no per-thread argmax, no tie slow path).  Measured, us per iteration: N = 25 000 1.93 -> 1.68, 28 672 2.26 -> 1.87
// (16 384: 1.04 -> 1.42, 20 000: 1.20 -> 1.52 -- the dependent DPP chains of a lone active wave cost more than the arithmetic
// they save there, so the smaller clouds stay with the wave-level kernel; row keys through 64-lane ds_max_u64 instead of DPP:
// 2.85 at 25 000, a same-address LDS atomic takes ~8 cycles per lane).
#ifdef PN2_FPS_DBG
__device__ unsigned long long pn2_fps_dbg[1024 * 16 * 6];   // [iteration][wave][t0, t_test, t_rows, t_tie, t_barrier, groups]
extern "C" __global__ void pn2_fps_dbg_touch() {}
#define FPS_DBG(k, v) if (b == 0 && lane == 0 && it < 1024) pn2_fps_dbg[(it * 16 + wave) * 6 + (k)] = (v);
#else
#define FPS_DBG(k, v)
#endif
// MAP: which rows of the sorted order a wave owns.  0: one contiguous run (a sample's neighbourhood then falls into two or three
// waves, which walk their 6 .. 9 touched rows one group after the other while the others idle: 3.2 active waves, 8.7 rows in
// the slowest, tools/fps_dbg.py).  1: row r belongs to wave r % NW.  2: GROUPS of four consecutive rows (256 sorted points: the
// unit of the interleaved DPP reduction) are dealt round-robin -- a neighbourhood of twenty rows becomes one group in each of
// five waves.  Measured, us per iteration, MAP 0 / 1 / 2: N = 25 000 1.64 / 1.55 / 1.41, 28 672 1.89 / 1.67 / 1.54 (B = 8 x 25 000:
// 1.68 / 1.58 / 1.42); at 16 384 and 20 000 points the wave-level kernel still wins (1.04 and 1.20 against 1.22 and 1.23).
template <int THREADS, int PPT, bool XYZ_LDS, int MAP>
__global__ __launch_bounds__(THREADS) void fps_rows_kernel(const float *__restrict__ xyz, int N,
                                                             const int64_t *__restrict__ start, int npoint,
                                                             int64_t *__restrict__ out) {
    constexpr int NW = THREADS / 64, WCAP = 64 * PPT, CAP = THREADS * PPT;
    extern __shared__ int fps_p_lds[];
    int *perm = fps_p_lds;                               // [CAP]: sorted position -> original index (-1: empty)
    int *hist = perm + CAP;                              // [FPS_NC] counts -> cursors
    int *wsum = hist + FPS_NC;                           // [NW] + carry
    float *box = reinterpret_cast<float *>(wsum + 32);   // [6] cloud bounding box, then [NW][6] wave boxes at box + 8
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(box + 8 + 6 * NW + 2);   // [3] rotating meeting words
    float4 *cloud = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(slots + 4) + 15) & ~(uintptr_t)15);   // XYZ_LDS: [N] by ORIGINAL index -- the winner's coordinates are
                                                             // one broadcast ds_read_b128 away instead of an L2 round trip

    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *p = xyz + (size_t)b * N * 3;

    // ---- cloud bounding box (the cell grid adapts to whatever scale the coordinates have)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = t; j < N; j += THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = p[3 * j + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int m = 32; m >= 1; m >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, 64)); }
    float *wbox = reinterpret_cast<float *>(perm);       // scratch [NW][6] (perm is not in use yet)
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { wbox[wave * 6 + a] = lo[a]; wbox[wave * 6 + 3 + a] = hi[a]; }
    }
    for (int i = t; i < FPS_NC; i += THREADS) hist[i] = 0;
    if (t < 3) slots[t] = 0ull;
    __syncthreads();
    if (t < 6) {
        float v = wbox[t];
        for (int w = 1; w < NW; ++w) v = t < 3 ? fminf(v, wbox[w * 6 + t]) : fmaxf(v, wbox[w * 6 + t]);
        box[t] = v;
    }
    __syncthreads();
    const float ox = box[0], oy = box[1], oz = box[2];
    // (an axis without extent -- a flat cloud -- gets scale 0: everything in cell 0 of that axis, never 0 * inf)
    const float sx = box[3] > ox ? 32.0f / (box[3] - ox) : 0.f, sy = box[4] > oy ? 32.0f / (box[4] - oy) : 0.f,
                sz = box[5] > oz ? 4.0f / (box[5] - oz) : 0.f;
    __syncthreads();                                      // wbox (aliasing perm) is dead

    // ---- counting sort by cell: histogram, exclusive scan, fill through LDS cursors
    for (int j = t; j < N; j += THREADS)
        atomicAdd(&hist[fps_cell(p[3 * j], p[3 * j + 1], p[3 * j + 2], ox, oy, oz, sx, sy, sz)], 1);
    __syncthreads();
    {
        constexpr int CPT = FPS_NC / THREADS;             // consecutive cells per thread
        int c[CPT], sum = 0;
#pragma unroll
        for (int i = 0; i < CPT; ++i) { c[i] = hist[t * CPT + i]; sum += c[i]; }
        int x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int base = x - sum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int i = 0; i < CPT; ++i) { hist[t * CPT + i] = base; base += c[i]; }
    }
    __syncthreads();
    for (int j = t; j < N; j += THREADS) {
        const float x = p[3 * j], y = p[3 * j + 1], z = p[3 * j + 2];
        const int pos = atomicAdd(&hist[fps_cell(x, y, z, ox, oy, oz, sx, sy, sz)], 1);
        perm[pos] = j;
        if (XYZ_LDS) cloud[j] = make_float4(x, y, z, 0.f);
    }
    for (int i = N + t; i < CAP; i += THREADS) perm[i] = -1;
    __syncthreads();

    // ---- this thread's points: sorted positions wave * WCAP + i * 64 + lane.  The wave owns one contiguous run of the sorted
    // order; ROW i of the wave -- the i-th point of its 64 lanes -- is 64 consecutive sorted positions: a compact cluster with
    // its own bounding box and its own largest running distance, both held by LANE i of the wave.
    float px[PPT], py[PPT], pz[PPT], md[PPT];
    auto pos_of = [&](int i) {                            // sorted position of this thread's i-th point (a bijection onto [0, CAP))
        constexpr int FULL = PPT & ~3;
        if (MAP == 0) return wave * WCAP + i * 64 + lane;
        if (MAP == 1) return (i * NW + wave) * 64 + lane;
        return i < FULL ? ((((i >> 2) * NW + wave) * 4 + (i & 3)) * 64 + lane) : (FULL * NW * 64 + ((i - FULL) * NW + wave) * 64 + lane);
    };
    float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};   // lane i: box of row i (empty: never reached)
    unsigned rmax = 0u;                                   // lane i: bits of the largest running distance of row i (0: empty / exhausted)
    int minidx = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = perm[pos_of(i)];
        float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (j >= 0) {
            px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2];
            md[i] = 1e10f;
            blo[0] = bhi[0] = px[i]; blo[1] = bhi[1] = py[i]; blo[2] = bhi[2] = pz[i];
            minidx = j < minidx ? j : minidx;
        } else {
            px[i] = py[i] = pz[i] = 0.f;
            md[i] = -1.f;                                 // never a candidate: distances are >= 0
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int m = 32; m >= 1; m >>= 1) { blo[a] = fminf(blo[a], __shfl_xor(blo[a], m, 64)); bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], m, 64)); }
        if (lane == i) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { rlo[a] = blo[a]; rhi[a] = bhi[a]; }
            rmax = blo[0] <= bhi[0] ? __float_as_uint(1e10f) : 0u;
        }
    }

    int rot = 0;
    int far = (int)(start[b] < 0 ? 0 : (start[b] >= N ? N - 1 : start[b]));
    int64_t *o = out + (size_t)b * npoint;
    // The candidate a wave offers while it is being skipped must be the one the plain kernel would compute from the same
    // running distances: initially every point sits at 1e10, i.e. (1e10, lowest original index of the wave); an empty
    // wave offers nothing, ever.
    for (int m = 32; m >= 1; m >>= 1) { const int o2 = __shfl_xor(minidx, m, 64); minidx = o2 < minidx ? o2 : minidx; }
    unsigned long long wkey = minidx == 0x7FFFFFFF ? 0ull : ((unsigned long long)__float_as_uint(1e10f) << 32) | (0xFFFFFFFFu - (unsigned)minidx);
    __syncthreads();
    for (int it = 0; it < npoint; ++it) {
        FPS_DBG(0, __builtin_readcyclecounter())
        if (t == 0) o[it] = far;
        float cx, cy, cz;
        if (XYZ_LDS) {
            float4 c = cloud[far];
            PN2_OPAQUE3(c.x, c.y, c.z);
            cx = c.x; cy = c.y; cz = c.z;
        } else {
            const int f = __builtin_amdgcn_readfirstlane(far);
            cx = p[3 * f]; cy = p[3 * f + 1]; cz = p[3 * f + 2];
        }
        // every row against the new sample at once (lane i: row i): squared distance to the row's box, from below.  A row
        // farther away than its largest running distance cannot change (EXACT, see the header).
        const float ex = fmaxf(fmaxf(rlo[0] - cx, cx - rhi[0]), 0.f), ey = fmaxf(fmaxf(rlo[1] - cy, cy - rhi[1]), 0.f),
                    ez = fmaxf(fmaxf(rlo[2] - cz, cz - rhi[2]), 0.f);
        const float lb = (ex * ex + ey * ey) + ez * ez;
        const unsigned long long touched = __ballot(!(lb * 0.999998f > __uint_as_float(rmax)));   // (empty rows: lb = inf)
        FPS_DBG(1, __builtin_readcyclecounter())
#ifdef PN2_FPS_DBG
        { int ng = 0; for (int i0 = 0; i0 < PPT; i0 += 4) ng += ((touched >> i0) & 15ull) ? 1 : 0; FPS_DBG(5, (unsigned long long)ng | ((unsigned long long)__popcll(touched) << 32)) }
        FPS_DBG(2, 0ull) FPS_DBG(3, 0ull)
#endif
        if (touched != 0ull) {                            // wave-uniform: the sample may reach into some of this wave's rows
            // rows in groups of four (neighbours in the sorted order: a sample that touches one usually touches the next): the
            // four row maxima go through one interleaved DPP reduction
#pragma unroll
            for (int i0 = 0; i0 < PPT; i0 += 4) {
                if ((touched >> i0) & 15ull) {            // wave-uniform
                    unsigned v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u < PPT ? i0 + u : PPT - 1;
                        if (i0 + u < PPT) {
                            const float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
                            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                            const float d = (xx + yy) + zz;
                            md[i] = min_raw(d, md[i]);    // empty slots hold -1 and stay there
                            v[u] = md[i] < 0.f ? 0u : __float_as_uint(md[i]);
                        } else {
                            v[u] = 0u;
                        }
                    }
                    pn2_wave_max_u32_x4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u < PPT) rmax = lane == i0 + u ? (unsigned)__builtin_amdgcn_readlane((int)v[u], 63) : rmax;
                }
            }
            FPS_DBG(2, __builtin_readcyclecounter())
            // the wave's candidate: the largest row maximum (distances are >= 0: their bit patterns order like unsigned
            // integers), and among the points that attain it the lowest ORIGINAL index -- looked up only in the rows that tie
            const unsigned bv = pn2_wave_max_u32(rmax);
            const unsigned long long tied = __ballot(rmax == bv && lane < PPT);
            unsigned low = 0u;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                if ((tied >> i) & 1ull) {                 // wave-uniform
                    const int j = perm[pos_of(i)];
                    const unsigned c = (md[i] >= 0.f && __float_as_uint(md[i]) == bv) ? 0xFFFFFFFFu - (unsigned)j : 0u;
                    const unsigned cm = pn2_wave_max_u32(c);
                    low = cm > low ? cm : low;
                }
            }
            wkey = low == 0u ? 0ull : ((unsigned long long)bv << 32) | low;      // (no valid point in the wave: nothing to offer)
            FPS_DBG(3, __builtin_readcyclecounter())
        }
        unsigned long long *cur = slots + rot;
        rot = rot == 2 ? 0 : rot + 1;
        if (lane == 0) __hip_atomic_fetch_max(cur, wkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t == 0) slots[rot] = 0ull;
        __syncthreads();
        far = (int)(0xFFFFFFFFu - (unsigned)(*cur & 0xFFFFFFFFull));
        FPS_DBG(4, __builtin_readcyclecounter())
    }
}

template <int THREADS, int PPT>
int launch_fps_pruned(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out, hipStream_t s) {
    // two-level pruning (fps_rows_kernel) where the wave-level kernel spills: 24+ points per thread; PN2_FPS_ROWS_MIN_PPT moves the
    // hand-over, PN2_FPS_ROWMAP picks the row ownership (A/B runs)
    const int rows_min = pn2_opt(PN2_OPT_FPS_ROWS_MIN_PPT);
    const int rowmap = pn2_opt(PN2_OPT_FPS_ROWMAP);
    const size_t fixed = sizeof(int) * ((size_t)THREADS * PPT + FPS_NC + 32 + 8 + 6 * (THREADS / 64) + 2) + 64;
    const bool in_lds = fixed + (size_t)N * 16 <= 160 * 1024;
    typedef void (*kernel_t)(const float *, int, const int64_t *, int, int64_t *);
    kernel_t k_lds = &fps_pruned_kernel<THREADS, PPT, true>, k_mem = &fps_pruned_kernel<THREADS, PPT, false>;
    if constexpr (PPT >= 16) {
        if (PPT >= rows_min) {
            if (rowmap == 0) { k_lds = &fps_rows_kernel<THREADS, PPT, true, 0>; k_mem = &fps_rows_kernel<THREADS, PPT, false, 0>; }
            else if (rowmap == 1) { k_lds = &fps_rows_kernel<THREADS, PPT, true, 1>; k_mem = &fps_rows_kernel<THREADS, PPT, false, 1>; }
            else { k_lds = &fps_rows_kernel<THREADS, PPT, true, 2>; k_mem = &fps_rows_kernel<THREADS, PPT, false, 2>; }
        }
    }
    if (pn2_raise_dynamic_lds_once(reinterpret_cast<const void *>(k_lds)) != PN2_OK ||
        pn2_raise_dynamic_lds_once(reinterpret_cast<const void *>(k_mem)) != PN2_OK)
        return PN2_ELAUNCH;
    if (in_lds)
        hipLaunchKernelGGL(k_lds, dim3(B), dim3(THREADS), fixed + (size_t)N * 16, s, xyz, N, start, npoint, out);
    else
        hipLaunchKernelGGL(k_mem, dim3(B), dim3(THREADS), fixed, s, xyz, N, start, npoint, out);
    return pn2_launch_status();
}

// Any N: the running distance lives in caller scratch (global), the cloud is re-read from
// L2 every iteration.  Used beyond the register-resident range (N > 16384).
__global__ __launch_bounds__(1024) void fps_large_kernel(const float *__restrict__ xyz, int N,
                                                         const int64_t *__restrict__ start, int npoint,
                                                         int64_t *__restrict__ out, float *__restrict__ work) {
    __shared__ unsigned long long slots[2][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    float *md = work + (size_t)b * N;
    for (int j = t; j < N; j += 1024) md[j] = 1e10f;
    // an out-of-range start (only the Python `start=` override can produce one) must not index past the cloud
    int far = (int)(start[b] < 0 ? 0 : (start[b] >= N ? N - 1 : start[b]));
    int64_t *o = out + (size_t)b * npoint;
    for (int it = 0; it < npoint; ++it) {
        if (t == 0) o[it] = far;
        int f = __builtin_amdgcn_readfirstlane(far);
        float cx = p[3 * f], cy = p[3 * f + 1], cz = p[3 * f + 2];
        float bm = -1.0f;
        int bj = 0;
        for (int j = t; j < N; j += 1024) {
            float dx = p[3 * j] - cx, dy = p[3 * j + 1] - cy, dz = p[3 * j + 2] - cz;
            float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            float d = (xx + yy) + zz;
            float m = md[j];
            m = d < m ? d : m;
            md[j] = m;
            if (m > bm) { bm = m; bj = j; }
        }
        unsigned long long key = bm < 0.f ? 0ull
                                          : ((unsigned long long)__float_as_uint(bm) << 32) | (0xFFFFFFFFu - (unsigned)bj);
        key = pn2_wave_max_u64(key);
        if (lane == 0) slots[it & 1][wave] = key;
        __syncthreads();
        key = slots[it & 1][lane & 15];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            unsigned long long other = __shfl_xor(key, m, 64);
            key = other > key ? other : key;
        }
        far = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    }
}

// N > 16384: ONE cloud spread over W workgroups (grid (W, B), 1024 threads, PPT = 8 or 16 points per thread in
// registers, workgroup w owns the points [w * 1024 * PPT, (w + 1) * 1024 * PPT)).  An iteration is the single-
// workgroup iteration plus one exchange: the thread that owns the workgroup's best point publishes
// {key, x, y, z} into the slot (iteration, w) of a zero-initialised table, and wave 0 of every workgroup polls the W
// slots of that iteration until all 4 W words are non-zero (the xyz words carry the iteration number in their
// upper half, the key is never 0).  All accesses are relaxed agent-scope atomics, served at the device's point of
// coherence -- no fences, no second pass over memory; the winner's coordinates travel with the key, so no
// dependent load follows.  The single-workgroup fallback below re-reads the whole cloud and its distance array
// from L2 every iteration: 13 us per iteration at N = 65 536 against ~3 us here.
// All W * B workgroups must be resident at once (they spin on each other): the host caps W * B.
struct FpsSlot { unsigned long long w[4]; };               // key, x | tag, y | tag, z | tag

template <int PPT>
__global__ __launch_bounds__(1024) void fps_coop_kernel(const float *__restrict__ xyz, int N,
                                                        const int64_t *__restrict__ start, int npoint,
                                                        int64_t *__restrict__ out, FpsSlot *__restrict__ table) {
    __shared__ unsigned long long red[3];                 // the waves' ds_max_u64 meeting word, three in rotation
    __shared__ float4 bcast[2];
    const int W = gridDim.x, w = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *p = xyz + (size_t)b * N * 3;
    const int base = w * 1024 * PPT;
    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = base + t + i * 1024;
        if (j < N) { px[i] = p[3 * j]; py[i] = p[3 * j + 1]; pz[i] = p[3 * j + 2]; md[i] = 1e10f; }
        else { px[i] = py[i] = pz[i] = 0.f; md[i] = -1.f; }      // never a candidate: distances are >= 0
    }
    if (t < 3) red[t] = 0ull;
    __syncthreads();
    int rot = 0;
    // an out-of-range start (only the Python `start=` override can produce one) must not index past the cloud
    int far = (int)(start[b] < 0 ? 0 : (start[b] >= N ? N - 1 : start[b]));
    float cx = p[3 * far], cy = p[3 * far + 1], cz = p[3 * far + 2];
    int64_t *o = out + (size_t)b * npoint;
    FpsSlot *tab = table + (size_t)b * npoint * W;
    for (int it = 0; it < npoint; ++it) {
        if (w == 0 && t == 0) o[it] = far;
        if (it == npoint - 1) break;                          // the last sample needs no successor
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float dx = px[i] - cx, dy = py[i] - cy, dz = pz[i] - cz;
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            const float d = (xx + yy) + zz;
            md[i] = min_raw(d, md[i]);                         // slots past N hold -1 and stay there
        }
        float bm = md[0];                                     // maximum first, then the lowest slot that holds it
#pragma unroll
        for (int i = 1; i < PPT; ++i) bm = __builtin_fmaxf(bm, md[i]);
        int bi = PPT - 1;
        float bx = px[PPT - 1], by = py[PPT - 1], bz = pz[PPT - 1];
#pragma unroll
        for (int i = PPT - 2; i >= 0; --i) {
            const bool hit = md[i] == bm;
            bi = hit ? i : bi; bx = hit ? px[i] : bx; by = hit ? py[i] : by; bz = hit ? pz[i] : bz;
        }
        const int bj = base + t + bi * 1024;
        const unsigned long long mine = bm < 0.f ? 0ull
                                                 : ((unsigned long long)__float_as_uint(bm) << 32) | (0xFFFFFFFFu - (unsigned)bj);
        unsigned long long key = pn2_wave_max_u64_dpp(mine);
        unsigned long long *cur = red + rot;
        rot = rot == 2 ? 0 : rot + 1;
        if (lane == 0) __hip_atomic_fetch_max(cur, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (t == 0) red[rot] = 0ull;                          // cleared two barriers after its last reader
        __syncthreads();
        key = *cur;                                           // every thread now holds the workgroup's best
        FpsSlot *row = tab + (size_t)it * W;
        if (mine == key && mine != 0ull) {                    // exactly one thread (indices are unique): publish
            const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
            __hip_atomic_store(&row[w].w[1], tag | __float_as_uint(bx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&row[w].w[2], tag | __float_as_uint(by), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&row[w].w[3], tag | __float_as_uint(bz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&row[w].w[0], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {                                      // lanes 0 .. 4W-1 poll one word each
            const bool act = lane < 4 * W;
            const unsigned long long *src = &row[lane >> 2].w[lane & 3];
            unsigned long long v = 1ull;
            // Bounded: if the W workgroups of a cloud are not all resident (a partitioned device, a CU mask the host
            // could not see) the missing ones never publish.  After ~1 s of polling the kernel aborts: the launch
            // fails loudly (hipErrorLaunchFailure at the next synchronisation) instead of hanging the GPU.
            unsigned spins = 0;
            for (;;) {
                if (act) v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__ballot(v == 0ull) == 0ull) break;
                if (++spins > (1u << 22)) __builtin_trap();
                if (spins > 64) __builtin_amdgcn_s_sleep(2);           // a late partner: stop hammering the fabric
            }
            const unsigned long long best = pn2_wave_max_u64_dpp((act && (lane & 3) == 0) ? v : 0ull);
            const unsigned long long hit = __ballot(act && (lane & 3) == 0 && v == best);
            const int src_lane = __builtin_ctzll(hit);        // lowest lane = lowest w; keys are unique anyway
            const unsigned lo = (unsigned)v;
            const float wx = __uint_as_float((unsigned)__shfl((int)lo, src_lane + 1, 64));
            const float wy = __uint_as_float((unsigned)__shfl((int)lo, src_lane + 2, 64));
            const float wz = __uint_as_float((unsigned)__shfl((int)lo, src_lane + 3, 64));
            if (lane == 0)
                bcast[it & 1] = make_float4(wx, wy, wz, __uint_as_float(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull)));
        }
        __syncthreads();
        float4 c = bcast[it & 1];
        PN2_OPAQUE4(c.x, c.y, c.z, c.w);
        cx = c.x; cy = c.y; cz = c.z;
        far = (int)__float_as_uint(c.w);
    }
}

template <int THREADS, int PPT>
int launch_fps(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out, hipStream_t s) {
    const bool in_lds = (size_t)N * 16 + 32 <= 160 * 1024;  // gfx950: 160 KiB of LDS per workgroup
    size_t lds = (in_lds ? (size_t)N * 16 : 0) + 32;           // + the three rotating reduction words
    if (in_lds) {
        if (lds > 64 * 1024) {   // above the default dynamic-LDS window: opt in once per instantiation
            static Pn2PerDevice raised;
            if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&fps_kernel<THREADS, PPT, true>), raised) != PN2_OK) return PN2_ELAUNCH;
        }
        hipLaunchKernelGGL((fps_kernel<THREADS, PPT, true>), dim3(B), dim3(THREADS), lds, s, xyz, N, start, npoint, out);
    } else
        hipLaunchKernelGGL((fps_kernel<THREADS, PPT, false>), dim3(B), dim3(THREADS), lds, s, xyz, N, start, npoint, out);
    return pn2_launch_status();
}

// Spatial (Morton-cell) ordering of the query centres of one cloud for ball_query_kernel: bounding box, 4096-cell histogram,
// scan and fill in LDS -- one workgroup per cloud, any S (the order array lives in global memory).
template <int THREADS, int PPT>
int launch_fps_pieces(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out, void *work, int piece, hipStream_t s) {
    const size_t lds = sizeof(float4) * (size_t)N + 4 * sizeof(unsigned long long);
    float *md_ws = reinterpret_cast<float *>(work);
    int *far_ws = reinterpret_cast<int *>(md_ws + (size_t)B * PPT * THREADS);
    static Pn2PerDevice raised;
    if (lds > 64 * 1024 && pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&fps_kernel<THREADS, PPT, true, true>), raised) != PN2_OK) return PN2_ELAUNCH;
    for (int it0 = 0; it0 < npoint; it0 += piece)
        hipLaunchKernelGGL((fps_kernel<THREADS, PPT, true, true>), dim3(B), dim3(THREADS), lds, s, xyz, N, start, npoint, out, it0, it0 + piece, md_ws,
                           far_ws);
    return pn2_launch_status();
}

__global__ __launch_bounds__(1024) void bq_order_kernel(const float *__restrict__ new_xyz, int S, int *__restrict__ order) {
    __shared__ int hist[FPS_NC];
    __shared__ int wsum[16];
    __shared__ float wbox[16][6];
    __shared__ float box[6];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *q = new_xyz + (size_t)b * S * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = t; j < S; j += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = q[3 * j + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int m = 32; m >= 1; m >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], m, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m, 64)); }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { wbox[wave][a] = lo[a]; wbox[wave][3 + a] = hi[a]; }
    }
    for (int i = t; i < FPS_NC; i += 1024) hist[i] = 0;
    __syncthreads();
    if (t < 6) {
        float v = wbox[0][t];
        for (int w = 1; w < 16; ++w) v = t < 3 ? fminf(v, wbox[w][t]) : fmaxf(v, wbox[w][t]);
        box[t] = v;
    }
    __syncthreads();
    const float ox = box[0], oy = box[1], oz = box[2];
    const float sx = box[3] > ox ? 32.0f / (box[3] - ox) : 0.f, sy = box[4] > oy ? 32.0f / (box[4] - oy) : 0.f,
                sz = box[5] > oz ? 4.0f / (box[5] - oz) : 0.f;
    for (int j = t; j < S; j += 1024) atomicAdd(&hist[fps_cell(q[3 * j], q[3 * j + 1], q[3 * j + 2], ox, oy, oz, sx, sy, sz)], 1);
    __syncthreads();
    {
        constexpr int CPT = FPS_NC / 1024;
        int c[CPT], sum = 0;
#pragma unroll
        for (int i = 0; i < CPT; ++i) { c[i] = hist[t * CPT + i]; sum += c[i]; }
        int x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int base = x - sum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int i = 0; i < CPT; ++i) { hist[t * CPT + i] = base; base += c[i]; }
    }
    __syncthreads();
    int *o = order + (size_t)b * S;
    for (int j = t; j < S; j += 1024) o[atomicAdd(&hist[fps_cell(q[3 * j], q[3 * j + 1], q[3 * j + 2], ox, oy, oz, sx, sy, sz)], 1)] = j;
}

// ---------------------------------------------------------------------------------------------
// Ball query.  A workgroup of BQ_WAVES waves takes BQ_WAVES * BQ_CPW consecutive query centres of one cloud, every wave
// BQ_CPW of them at once.  The candidates are staged through LDS in tiles of BQ_TILE points as (x, y, z, |p|^2) float4 --
// loaded and normed ONCE per workgroup (coalesced, the next tile's raw floats already in flight in registers while the
// current one is scanned) instead of every wave re-reading the cloud from L2 with 12-byte-stride scalar loads and
// recomputing the norm per pair.  A wave scans the tile in index order, 64 candidates per ds_read_b128 step, and tests
// that candidate against its BQ_CPW centres: one LDS read and one loop trip per 256 pairs.  Per centre a ballot + prefix
// popcount gives each in-radius lane its output slot; a centre stops at nsample slots, the wave when all its centres
// have, the tile loop when every wave has (the reference sorts a full N-row instead, pointnet_util.py:100-106).
// On a KITTI-shaped cloud 8..40 % of the balls never fill (far, sparse points): those centres scan the whole cloud,
// which is what the launch costs -- 3 G pairs at cfg5.
// ---------------------------------------------------------------------------------------------
constexpr int BQ_WAVES = 4, BQ_CPW = 4, BQ_TILE = 1024, BQ_THREADS = BQ_WAVES * 64, BQ_PPT = BQ_TILE / BQ_THREADS;

//
// `order` (optional, [B][S] int32): a spatial ordering of the centres (bq_order_kernel).  A workgroup lives as long as the
// slowest of its BQ_WAVES * BQ_CPW centres scans; taken in sampling order (FPS output: scattered all over the cloud) almost
// every workgroup holds one sparse-region centre that runs to the end of the cloud.  Taken in Morton order the sixteen centres
// of a workgroup are neighbours with similar neighbour density: dense groups stop after a few tiles, and the full scans are
// confined to the workgroups of the sparse regions.  The results are the same rows, written to the centres' own positions.
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float *__restrict__ xyz,
                                                                const float *__restrict__ new_xyz, int N, int S, float r2,
                                                                int nsample, const int *__restrict__ order,
                                                                int64_t *__restrict__ out) {
    __shared__ float4 tile[BQ_TILE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int s0 = (blockIdx.x * BQ_WAVES + wave) * BQ_CPW, b = blockIdx.y;
    const float *p = xyz + (size_t)b * N * 3;
    float qx[BQ_CPW], qy[BQ_CPW], qz[BQ_CPW], nq[BQ_CPW];
    int cnt[BQ_CPW], first[BQ_CPW];
    int64_t *rows[BQ_CPW];
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const bool live = s0 + c < S;
        const int sid = !live ? 0 : (order ? order[(size_t)b * S + s0 + c] : s0 + c);
        const float *q = new_xyz + ((size_t)b * S + sid) * 3;
        qx[c] = q[0]; qy[c] = q[1]; qz[c] = q[2];
        nq[c] = sq_norm3(qx[c], qy[c], qz[c]);
        cnt[c] = live ? 0 : nsample;                     // a centre past the end is "done" from the start
        first[c] = N;
        rows[c] = out + ((size_t)b * S + sid) * nsample;
    }
    float raw[BQ_PPT][3];
    auto fetch = [&](int base) {
#pragma unroll
        for (int i = 0; i < BQ_PPT; ++i) {
            const int j = base + t + i * BQ_THREADS;
            const float *src = p + 3 * (size_t)(j < N ? j : 0);
            raw[i][0] = src[0]; raw[i][1] = src[1]; raw[i][2] = src[2];
        }
    };
    fetch(0);
    for (int base = 0; base < N; base += BQ_TILE) {
        const int tn = min(BQ_TILE, N - base);
#pragma unroll
        for (int i = 0; i < BQ_PPT; ++i)
            tile[t + i * BQ_THREADS] = make_float4(raw[i][0], raw[i][1], raw[i][2], sq_norm3(raw[i][0], raw[i][1], raw[i][2]));
        __syncthreads();
        if (base + BQ_TILE < N) fetch(base + BQ_TILE);           // in flight under the scan
        bool busy = false;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) busy |= cnt[c] < nsample;
        // Branch-free test of one candidate against the wave's centres (a finished centre tests against r2 = -1: never
        // inside), ONE branch per step for the rare "somebody got a hit" path, two steps in flight: a lone full-scan
        // wave -- the isolated far point that never fills its ball decides how long the workgroup lives -- is bound
        // by the latency chain LDS read -> distance -> compare -> branch, not by issue slots.
        for (int k = 0; k < tn && busy; k += 128) {
            const int j0 = k + lane, j1 = k + 64 + lane;
            const float4 v0 = tile[j0 < tn ? j0 : 0], v1 = tile[j1 < tn ? j1 : 0];
            unsigned long long m0[BQ_CPW], m1[BQ_CPW], any = 0ull;
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) {
                const float rc = cnt[c] < nsample ? r2 : -1.0f;
                const float d0 = pair_dist(qx[c], qy[c], qz[c], nq[c], v0.x, v0.y, v0.z, v0.w);
                const float d1 = pair_dist(qx[c], qy[c], qz[c], nq[c], v1.x, v1.y, v1.z, v1.w);
                m0[c] = __ballot(j0 < tn && !(d0 > rc));
                m1[c] = __ballot(j1 < tn && !(d1 > rc));
                any |= m0[c] | m1[c];
            }
            if (any == 0ull) continue;
            busy = false;
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned long long mask = h ? m1[c] : m0[c];
                    if (mask && cnt[c] < nsample) {              // (the second half-step of a centre that just filled up: skipped)
                        const int j = h ? j1 : j0;
                        const int pos = cnt[c] + __popcll(mask & ((1ull << lane) - 1ull));
                        if (((mask >> lane) & 1ull) && pos < nsample) rows[c][pos] = base + j;
                        if (cnt[c] == 0) first[c] = base + k + 64 * h + __ffsll((long long)mask) - 1;
                        cnt[c] += __popcll(mask);
                    }
                }
                busy |= cnt[c] < nsample;
            }
        }
        if (__syncthreads_or(busy) == 0) break;                   // also: the tile may be overwritten
    }
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        if (s0 + c >= S) break;
        const int have = cnt[c] > nsample ? nsample : cnt[c];
        for (int k = have + lane; k < nsample; k += 64) rows[c][k] = first[c];
    }
}

// ---------------------------------------------------------------------------------------------
// 3-NN + inverse-distance weights: one thread per query point, the S candidates are staged
// through LDS as (x, y, z, |p|^2) and broadcast-read; the reference materialises [B,N,S] and
// sorts every row (pointnet_util.py:295-300).
// ---------------------------------------------------------------------------------------------
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2, int N, int S,
                                                       int64_t *__restrict__ idx, float *__restrict__ dist,
                                                       float *__restrict__ weight) {
    __shared__ float4 tile[NN_TILE];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool live = n < N;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const float *q = xyz1 + ((size_t)b * N + n) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    const float nq = sq_norm3(qx, qy, qz);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    const float *c = xyz2 + (size_t)b * S * 3;
    for (int base = 0; base < S; base += NN_TILE) {
        const int cnt = min(NN_TILE, S - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            float x = c[3 * (base + k)], y = c[3 * (base + k) + 1], z = c[3 * (base + k) + 2];
            tile[k] = make_float4(x, y, z, sq_norm3(x, y, z));
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const float4 v = tile[k];
            const float d = pair_dist(qx, qy, qz, nq, v.x, v.y, v.z, v.w);
            const int j = base + k;
            if (d < d2) {
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                    else { d1 = d; i1 = j; }
                } else { d2 = d; i2 = j; }
            }
        }
    }
    if (!live) return;
    const size_t o = ((size_t)b * N + n) * 3;
    idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
    dist[o] = d0; dist[o + 1] = d1; dist[o + 2] = d2;
    float w0 = d0 < 1e-10f ? 1e-10f : d0, w1 = d1 < 1e-10f ? 1e-10f : d1, w2 = d2 < 1e-10f ? 1e-10f : d2;
    w0 = __fdiv_rn(1.0f, w0); w1 = __fdiv_rn(1.0f, w1); w2 = __fdiv_rn(1.0f, w2);
    const float sum = (w0 + w1) + w2;
    weight[o] = __fdiv_rn(w0, sum); weight[o + 1] = __fdiv_rn(w1, sum); weight[o + 2] = __fdiv_rn(w2, sum);
}

__global__ __launch_bounds__(256) void square_distance_kernel(const float *__restrict__ src,
                                                              const float *__restrict__ dst, int S, int N,
                                                              float *__restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float *q = src + ((size_t)b * S + i) * 3;
    const float *p = dst + ((size_t)b * N + j) * 3;
    const float qx = q[0], qy = q[1], qz = q[2], x = p[0], y = p[1], z = p[2];
    out[((size_t)b * S + i) * N + j] = pair_dist(qx, qy, qz, sq_norm3(qx, qy, qz), x, y, z, sq_norm3(x, y, z));
}

}  // namespace

// Workgroups of fps_coop_kernel<PPT> that are certainly co-resident: what the occupancy query admits per CU (1024-thread
// workgroups: at most 2), times the CUs of THIS device (a CPX partition has 32, not 256), halved as a margin -- the
// query over-reports by one block per CU for some register footprints (MI355X_MICROARCH.md, Residency) and a queue may
// run under a CU mask the runtime does not report.  On a full MI355X: 256 * 1 / 2 = 128, the figure measured in round 1.
template <int PPT>
static int fps_coop_capacity() {
    static const int cap = [] {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&fps_coop_kernel<PPT>), 1024, 0) != hipSuccess)
            per_cu = 1;                                   // no device to ask (host-only callers): the planning default
        if (per_cu > 1) per_cu = 1;                       // one 1024-thread workgroup per CU is all the plan ever counts on
        return pn2_num_cus() * per_cu / 2;
    }();
    return cap;
}

extern "C" {

// Cooperative plan for N > 16384: PPT points per thread and W workgroups per cloud, or W = 0 (single-workgroup
// fallback) when the W * B workgroups could not all be resident or a cloud would need more than 16 of them.
// Largest cloud the register-resident single-workgroup kernel takes.  One CU's VALU rate bounds that kernel
// (0.59 us per iteration at N = 4096, 1.48 at 16 384, 1.65 at 20 000, 2.25 at 28 672 = 28 points per thread, the most
// that fits 1024 threads' 128-VGPR budget without spilling); the cooperative kernel's iteration costs 2.2 us up to
// N = 32 768 (its cross-CU exchange), so the hand-over sits at 24 576.  PN2_FPS_SINGLE_MAX moves it for A/B runs.
static int fps_single_max() {
    const int v = pn2_opt(PN2_OPT_FPS_SINGLE_MAX);
    return v < 16384 ? 16384 : (v > 28672 ? 28672 : v);
}

static void fps_coop_plan(int B, int N, int *ppt, int *W) {
    *ppt = 0; *W = 0;
    if (N <= fps_single_max()) return;
    const int enabled = pn2_opt(PN2_OPT_FPS_COOP);
    if (!enabled) return;
    for (int p = 8; p <= 16; p *= 2) {
        const int w = (int)pn2_cdiv(N, 1024 * p);
        const int cap = p == 8 ? fps_coop_capacity<8>() : fps_coop_capacity<16>();
        if (w <= 16 && (int64_t)w * B <= cap) { *ppt = p; *W = w; return; }
    }
}

int64_t pn2_fps_workspace_bytes(int B, int N, int npoint) {
    int ppt, W;
    fps_coop_plan(B, N, &ppt, &W);
    if (W) return (int64_t)B * npoint * W * (int64_t)sizeof(FpsSlot);
    if (N > 2048 && N <= 4096 && pn2_opt(PN2_OPT_FPS_PIECE) > 0) return (int64_t)B * (512 * 8 * 4 + 4);     // the pieced form's state
    return N > fps_single_max() ? (int64_t)B * N * 4 : 0;
}

int pn2_fps(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out_idx, void *work,
            pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz && start && out_idx && B > 0 && N > 0 && npoint > 0);
    hipStream_t s = pn2_s(stream);
    if (N <= 64) return launch_fps<64, 1>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 128) return launch_fps<64, 2>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 256) return launch_fps<64, 4>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 512) return launch_fps<64, 8>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 1024) return launch_fps<128, 8>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 2048) return launch_fps<256, 8>(xyz, B, N, start, npoint, out_idx, s);
    // spatially pruned kernel (exact, see fps_pruned_kernel) where a cloud spans at least eight waves and enough samples are
    // drawn to pay for the one-time sort; PN2_FPS_PRUNE=0 keeps the plain kernels (A/B runs)
    const int prune = pn2_opt(PN2_OPT_FPS_PRUNE);
    // (measured, us per iteration plain -> pruned: N = 4096 0.55 -> 0.85, 8192 0.85 -> 0.92 -- the plain kernel keeps the
    // cloud in LDS and the winner's coordinates are one broadcast read away, the pruned one fetches them from L2 --
    // 16 384 1.39 -> 1.07, 20 000 1.62 -> 1.23, 25 000 2.19 -> 1.81.  PN2_FPS_PRUNE=2 forces it from N > 2048 on: tests.)
    if (prune && N > (prune > 1 ? 2048 : 8192) && N <= 28672 && npoint >= 64) {
        if (N <= 4096) return launch_fps_pruned<512, 8>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 8192) return launch_fps_pruned<1024, 8>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 16384) return launch_fps_pruned<1024, 16>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 20480) return launch_fps_pruned<1024, 20>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 22528) return launch_fps_pruned<1024, 22>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 24576) return launch_fps_pruned<1024, 24>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 25600) return launch_fps_pruned<1024, 25>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 26624) return launch_fps_pruned<1024, 26>(xyz, B, N, start, npoint, out_idx, s);   // (24+: the row-level kernel)
        return launch_fps_pruned<1024, 28>(xyz, B, N, start, npoint, out_idx, s);
    }
    if (N <= 4096) {
        const int piece = pn2_opt(PN2_OPT_FPS_PIECE);
        if (piece > 0 && npoint > piece && work != nullptr) return launch_fps_pieces<512, 8>(xyz, B, N, start, npoint, out_idx, work, piece, s);
        return launch_fps<512, 8>(xyz, B, N, start, npoint, out_idx, s);
    }
    if (N <= 8192) return launch_fps<1024, 8>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= 16384) return launch_fps<1024, 16>(xyz, B, N, start, npoint, out_idx, s);
    if (N <= fps_single_max()) {
        if (N <= 20480) return launch_fps<1024, 20>(xyz, B, N, start, npoint, out_idx, s);
        if (N <= 24576) return launch_fps<1024, 24>(xyz, B, N, start, npoint, out_idx, s);
        return launch_fps<1024, 28>(xyz, B, N, start, npoint, out_idx, s);
    }
    PN2_CHECK_ARG(work != nullptr);
    int ppt, W;
    fps_coop_plan(B, N, &ppt, &W);
    if (W) {
        FpsSlot *table = reinterpret_cast<FpsSlot *>(work);
        pn2_fill_u32(table, 0u, (int64_t)B * npoint * W * (int64_t)(sizeof(FpsSlot) / 4), s);
        if (ppt == 8)
            hipLaunchKernelGGL(fps_coop_kernel<8>, dim3(W, B), dim3(1024), 0, s, xyz, N, start, npoint, out_idx, table);
        else
            hipLaunchKernelGGL(fps_coop_kernel<16>, dim3(W, B), dim3(1024), 0, s, xyz, N, start, npoint, out_idx, table);
        return pn2_launch_status();
    }
    hipLaunchKernelGGL(fps_large_kernel, dim3(B), dim3(1024), 0, s, xyz, N, start, npoint, out_idx, (float *)work);
    return pn2_launch_status();
}

// The centres are put in spatial order first where that pays: large clouds (a full scan is long) with enough centres for the
// ordering to form homogeneous workgroups.  PN2_BQ_ORDER=0: never (A/B runs).
static bool bq_wants_order(int N, int S) {
    const int on = pn2_opt(PN2_OPT_BQ_ORDER);
    return on && N >= 8192 && S >= 1024;
}

#ifdef PN2_FPS_DBG
int pn2_fps_debug_read(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pn2_fps_dbg), sizeof(unsigned long long) * (size_t)n);
}
#endif

int64_t pn2_ball_query_workspace_bytes(int B, int N, int S) {
    return (B > 0 && bq_wants_order(N, S)) ? (int64_t)B * S * (int64_t)sizeof(int) : 0;
}

int pn2_ball_query_ws(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample, int64_t *out_idx,
                      void *work, pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz && new_xyz && out_idx && B > 0 && N > 0 && S > 0 && nsample > 0 && B <= 65535);
    int *order = nullptr;
    if (work != nullptr && bq_wants_order(N, S)) {
        order = static_cast<int *>(work);
        hipLaunchKernelGGL(bq_order_kernel, dim3(B), dim3(1024), 0, pn2_s(stream), new_xyz, S, order);
    }
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)pn2_cdiv(S, BQ_WAVES * BQ_CPW), B), dim3(BQ_THREADS), 0, pn2_s(stream), xyz,
                       new_xyz, N, S, r2, nsample, order, out_idx);
    return pn2_launch_status();
}

int pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample,
                   int64_t *out_idx, pn2_stream_t stream) {
    return pn2_ball_query_ws(xyz, new_xyz, B, N, S, r2, nsample, out_idx, nullptr, stream);
}

int pn2_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, pn2_stream_t stream) {
    PN2_CHECK_ARG(src && dst && out && B > 0 && S > 0 && N > 0 && S <= 65535 && B <= 65535);
    hipLaunchKernelGGL(square_distance_kernel, dim3((unsigned)pn2_cdiv(N, 256), S, B), dim3(256), 0, pn2_s(stream), src,
                       dst, S, N, out);
    return pn2_launch_status();
}

int pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx, float *dist, float *weight,
                 pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz1 && xyz2 && idx && dist && weight && B > 0 && N > 0 && S >= 3 && B <= 65535);
    hipLaunchKernelGGL(three_nn_kernel, dim3((unsigned)pn2_cdiv(N, 256), B), dim3(256), 0, pn2_s(stream), xyz1, xyz2, N, S,
                       idx, dist, weight);
    return pn2_launch_status();
}

}  // extern "C"
