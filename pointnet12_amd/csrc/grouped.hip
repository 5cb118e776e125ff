// Factorised first layer of a set-abstraction MLP.
//
// The grouped input row of position p = (b, s, k) is [f_j, xyz_j - c_s] with j = idx[p], so the first
// 1x1 convolution splits into a per-SOURCE-POINT part and a 3-term geometric part:
//     y[p, :] = (W_f f_j + bias)  +  W_x (xyz_j - c_s)  =  Zf[b, j, :] + W_x (xyz_j - c_s)
// Zf is one small GEMM over the B*N source points instead of the B*S*K grouped positions (K = 32..128
// times fewer rows), the grouped tensor [P, 3+D] is never materialised, and the forward kernel is a row
// gather of Zf plus three FMAs per output with the exactly centred coordinates (the subtraction happens
// before the multiply, as in the reference, pointnet_util.py:128,244 -- no cancellation).
// Replaces, for the first layer only: index_points + cat (pointnet_util.py:127-131, :243-247) and
// nn.Conv2d (pointnet_util.py:197, :254), and their autograd.
#include "pn2_common.h"
#include "bn_tail.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Per-channel sum(y), sum(y*y) for the training-mode BatchNorm are accumulated like in the GEMM epilogue:
// fp32 per 64 rows -> fp64 registers -> LDS -> one fp64 atomic per channel per workgroup.
__global__ __launch_bounds__(256) void group_affine_fwd_kernel(const float *__restrict__ Zf, int ldz,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const int64_t *__restrict__ idx,
                                                               const float *__restrict__ Wx, int ldwx, int N, int S,
                                                               int K, int C, int64_t P, float *__restrict__ Y, int ldy,
                                                               double *__restrict__ stats, FinTail fin) {
    __shared__ double red[256 * 8];
    const int CG = (C + 3) >> 2;                  // float4 column groups per row
    const int RPB = 256 / CG;                     // rows per pass
    const int t = threadIdx.x, cg = t % CG, r = t / CG;
    const bool live = r < RPB;
    const int c = cg * 4;
    float wx[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int a = 0; a < 3; ++a) wx[e][a] = (live && c + e < C) ? Wx[(c + e) * ldwx + a] : 0.f;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int since = 0;
    if (live) {
        // four independent rows per trip: idx -> (xyz, centre, Zf row) is a two-deep dependent chain per row; with
        // the four chains interleaved a trip costs two memory round trips instead of eight
        const int64_t stride = (int64_t)gridDim.x * RPB;
        for (int64_t p0 = (int64_t)blockIdx.x * RPB + r; p0 < P; p0 += 4 * stride) {
            int64_t j[4], g[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                const bool v = p < P;
                g[u] = v ? p / K : 0;
                b[u] = g[u] / S;
                j[u] = v ? idx[p] : 0;
            }
            float4 z[4];
            float d[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *q = xyz + (b[u] * N + j[u]) * 3, *ctr = new_xyz + g[u] * 3;
                d[u][0] = q[0] - ctr[0]; d[u][1] = q[1] - ctr[1]; d[u][2] = q[2] - ctr[2];
                z[u] = *reinterpret_cast<const float4 *>(Zf + (b[u] * N + j[u]) * ldz + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                if (p >= P) break;
                float y[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = __builtin_fmaf(wx[e][0], d[u][0], y[e]);
                    y[e] = __builtin_fmaf(wx[e][1], d[u][1], y[e]);
                    y[e] = __builtin_fmaf(wx[e][2], d[u][2], y[e]);
                    if (c + e >= C) y[e] = 0.f;
                    s0[e] += y[e];
                    s1[e] = __builtin_fmaf(y[e], y[e], s1[e]);
                }
                *reinterpret_cast<float4 *>(Y + p * ldy + c) = make_float4(y[0], y[1], y[2], y[3]);
                if (++since == 64) {              // fold the fp32 partials into fp64 every 64 rows
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st[e] += (double)s0[e]; st[4 + e] += (double)s1[e];
                        s0[e] = 0.f; s1[e] = 0.f;
                    }
                    since = 0;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { st[e] += (double)s0[e]; st[4 + e] += (double)s1[e]; }
    }
    if (stats == nullptr) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[t * 8 + e] = st[e];
    __syncthreads();
    for (int ch = t; ch < C; ch += 256) {
        const int g4 = ch >> 2, e = ch & 3;
        double a0 = 0.0, a1 = 0.0;
        for (int rr = 0; rr < RPB; ++rr) {
            a0 += red[(rr * CG + g4) * 8 + e];
            a1 += red[(rr * CG + g4) * 8 + 4 + e];
        }
        double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * C;
        atomicAdd(rep + ch, a0);
        atomicAdd(rep + C + ch, a1);
    }
    if (fin.ticket != nullptr && tail_is_last_block(fin.ticket, gridDim.x)) run_fin_tail(fin, stats, C, 256);
}

// Backward: dY = c0*dZ + q1*(y-mean) + q0 (BatchNorm backward folded into `coef`, see mlp.hip) is
// scattered back to the source points, G[b, j, :] += dY[p, :], and contracted with the centred
// coordinates, dWx[c, a] += dY[p, c] * (xyz_j - c_s)[a].  The weight / feature gradients then come from
// two small GEMMs over the B*N source points (dW_f = G^T F, dF = G W_f).
__global__ __launch_bounds__(256) void group_affine_bwd_kernel(const float *__restrict__ dZ, int ldz,
                                                               const float *__restrict__ Y, int ldy,
                                                               const float *__restrict__ coef, int ldc,
                                                               const float *__restrict__ xyz,
                                                               const float *__restrict__ new_xyz,
                                                               const int64_t *__restrict__ idx, int N, int S, int K,
                                                               int C, int64_t P, float *__restrict__ G, int ldg,
                                                               float *__restrict__ dWx, int ldwx) {
    // One lane per CHANNEL (not per float4): every atomic wave-instruction then adds 64 consecutive floats of one
    // G row -- the 256-byte contiguous shape the memory-side float atomics run at full rate with; the float4
    // mapping issued four 16-byte-strided instructions per row segment and ran ~4x slower.
    __shared__ float red[256 * 3];
    const int t = threadIdx.x;
    const int CT = C < 256 ? C : 256;             // threads per row
    const int RPB = 256 / CT;                     // rows per pass (1 when C >= 129)
    const int r = t / CT, cl = t % CT;
    const bool live = r < RPB;
    for (int cb = 0; cb < C; cb += CT) {          // C > 256: a thread owns channels cl, cl + 256, ... (uniform trip count)
        const int c = cb + cl;
        float acc[3] = {0.f, 0.f, 0.f};
        if (live && c < C) {
            const float c0 = coef[c], q1 = coef[ldc + c], q0 = coef[2 * ldc + c], mu = coef[3 * ldc + c];
            // four independent rows per trip: the idx -> xyz -> atomic chains of a lane overlap instead of queueing
            const int64_t stride = (int64_t)gridDim.x * RPB;
            for (int64_t p0 = (int64_t)blockIdx.x * RPB + r; p0 < P; p0 += 4 * stride) {
                int64_t j[4], g[4];
                float dzv[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t p = p0 + u * stride;
                    const bool v = p < P;
                    g[u] = v ? p / K : 0;
                    j[u] = v ? idx[p] : 0;
                    dzv[u] = v ? dZ[p * ldz + c] : 0.f;
                    yv[u] = v ? Y[p * ldy + c] : mu;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (p0 + u * stride >= P) break;
                    const int64_t b = g[u] / S;
                    const float *q = xyz + (b * N + j[u]) * 3, *ctr = new_xyz + g[u] * 3;
                    const float dy = __builtin_fmaf(c0, dzv[u], __builtin_fmaf(q1, yv[u] - mu, q0));
                    atomicAdd(G + (b * N + j[u]) * ldg + c, dy);
                    acc[0] = __builtin_fmaf(dy, q[0] - ctr[0], acc[0]);
                    acc[1] = __builtin_fmaf(dy, q[1] - ctr[1], acc[1]);
                    acc[2] = __builtin_fmaf(dy, q[2] - ctr[2], acc[2]);
                }
            }
        }
        red[t * 3] = acc[0]; red[t * 3 + 1] = acc[1]; red[t * 3 + 2] = acc[2];
        __syncthreads();
        if (t < CT && c < C) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int rr = 0; rr < RPB; ++rr) {
                s0 += red[(rr * CT + t) * 3]; s1 += red[(rr * CT + t) * 3 + 1]; s2 += red[(rr * CT + t) * 3 + 2];
            }
            atomicAdd(dWx + c * ldwx, s0);
            atomicAdd(dWx + c * ldwx + 1, s1);
            atomicAdd(dWx + c * ldwx + 2, s2);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Gather + first layer in one launch, for the narrow first layers (3 + D <= 12 input channels, 32 or 64 outputs: the sa1
// stacks).  The stand-alone pair was pn2_group (a gather whose time is the latency of 1 M x 3 dependent row reads: 87 us
// for 50 MB) followed by a GEMM that is pure store bandwidth (268 MB of Y at 1 M x 64): here a lane gathers one grouped
// row (index, centre, xyz, features), keeps it in registers, writes it out once (the backward's weight gradient reads it),
// evaluates the CO outputs as an fma chain against the weight image in LDS, and the 64 x CO tile leaves through LDS as
// 16-byte stores, 256 contiguous bytes per row; the BatchNorm statistics are column sums of the same tile.
// Measured (1 M rows x 64): 102 us alone -- the pair it replaces took 26 + 73 -- of which the stores are 14 and the gather
// nothing (sequential indices: 103): the kernel is bound by the return path of its 192 broadcast ds_read_b128 of weights
// per slab (a broadcast still delivers 64 x 16 bytes).  Inside the step, where the three sa1 branches overlap, one launch
// instead of two is worth 50 us (MSG 6.50 -> 6.45 ms).  An MFMA form (the weight-resident forward with a gathering loader)
// would run at the store bound, ~75 us.  model/pointnet_util.py:127-131 + :197 (first conv), :243-247 + :254.
// ---------------------------------------------------------------------------------------------------------------------
__device__ float pn2_gc_dump[1024 * 256 * 4];       // dump slots of group_conv_fwd_kernel (16 bytes per thread of 1024 workgroups)

template <int CO>
__global__ __launch_bounds__(256) void group_conv_fwd_kernel(const float *__restrict__ xyz, const float *__restrict__ points,
                                                             const float *__restrict__ new_xyz, const int64_t *__restrict__ idx,
                                                             int N, int S, int K, int D, int xyz_first,
                                                             const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                             float *__restrict__ X, int ldx, float *__restrict__ Y, int ldy,
                                                             int64_t slabs, double *__restrict__ stats) {
    // Round 3: the conv runs on the matrix cores.  A lane still gathers one grouped row (12 floats in registers, written once
    // to X for the backward); the rows then pass through this wave's LDS tile to become MFMA operands -- lane (l31, lh) reads
    // quad lh of row l31 (k = 0..7) and quad 2 (k = 8..11; the lh = 1 half multiplies a zero quad) -- against the weights held
    // in REGISTERS (8 per 32-column block).  Before: 768 VALU FMAs per row fed by 192 broadcast ds_read_b128 of weights per
    // slab, which bound the kernel on the LDS return path (1 M x 64: 105 us for 318 MB of stores); now 32 MFMAs and four
    // operand reads per slab.
    constexpr int LT = CO + 4;                                     // output tile pitch: 4 mod 8 floats (conflict-free b128 rows)
    constexpr int LX = 20;                                         // operand tile pitch: 12 used + zero quad at 12..15; 20 = 4 mod 16
    constexpr int NBK = CO / 32;
    __shared__ __attribute__((aligned(16))) float tile[4][64 * LT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int cin = 3 + D;
    float *T = tile[wave], *XT = tile[wave];                       // the operand rows alias the head of the output tile: they are
                                                                   // read into registers before the first output row is written
                                                                   // (two tiles of their own left one workgroup per CU)
    // this lane's weight slice: wreg[j][4 kb + e] = W[32 j + l31][8 kb + 4 lh + e] (zero beyond the 3 + D real columns)
    float wreg[NBK][8], bj[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * kb + 4 * lh + e;
                wreg[j][4 * kb + e] = k < cin ? W[(size_t)(32 * j + l31) * ldw + k] : 0.f;
            }
        bj[j] = bias[32 * j + l31];
    }
    double st[NBK][2];
#pragma unroll
    for (int j = 0; j < NBK; ++j) st[j][0] = st[j][1] = 0.0;
    const int xo = xyz_first ? 0 : D, fo = xyz_first ? 3 : 0;
    // Every wave works on slabs of its own (its own LDS tiles: no workgroup barrier in the loop -- LDS operations of one wave
    // execute in order), and the NEXT slab's index / centre / point reads are in flight while this one is computed and stored.
    const int64_t stride = (int64_t)gridDim.x * 4;
    // Two-stage prefetch.  The gather is a dependent chain (neighbour index -> point address), and gfx9 retires loads and
    // stores through ONE in-order counter: with index and points requested in the same stage, the wait for the index sat
    // behind the previous slab's 19 stores and every slab paid their HBM acknowledgement (the kernel ran at 3 TB/s of stores
    // whatever its arithmetic cost: the MFMA version above was no faster than the VALU one).  Now the index of slab s + 2 and
    // the points of slab s + 1 are requested while slab s is computed, each wait is a counted vmcnt that leaves the stores
    // issued after the request in flight.
    float gx, gy, gz, cxr, cyr, czr, f[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) f[k] = 0.f;
    int64_t jn;                                                    // neighbour index of the slab after next
    // (branch-free on purpose: a uniform branch around a request makes the compiler's vmcnt bookkeeping fall back to
    // vmcnt(0); the host guarantees idx, new_xyz and D >= 1)
    const int dlast = D - 1;
    // (32-bit row arithmetic: P < 2^31 is checked on the host; an int64 division compiles to a branch between a 32-bit and a
    // 64-bit path, which again costs the counted waits)
    // The requests are inline assembly and the waits are counted BY HAND: hipcc's own bookkeeping settles on vmcnt(0..2) in
    // this loop however it is written (measured: branch-free body, pinned request order, equalised loop entries), i.e. every
    // slab waits for the HBM acknowledgement of the previous slab's stores.  Issue order of one slab:
    //     [wait A] use points(s) ... [wait B] 15 point requests(s+1), 1 index request(s+2), 3 X stores, CO/4 Y stores
    // so behind the points of slab s there are 1 + 3 + CO/4 younger operations (wait A) and behind the index of slab s+1 there
    // are 3 + CO/4 (wait B).  The waits take the loaded registers as operands: nothing that reads them can move above.
    // (12-byte requests for the coordinates / centre / feature triples instead of one request per float were tried: the same
    // 91 us at 1 M rows -- the L1 tag lookups do not bound the gather either)
    auto ldg = [](const float *p) { float v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; };
    auto fetch_idx = [&](int64_t slab) {
        const unsigned r = (unsigned)(slab < slabs ? slab : slabs - 1) * 64u + (unsigned)lane;   // grouped row (past the end: a valid dummy)
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(jn) : "v"(idx + r) : "memory");
    };
    auto fetch_pts = [&](int64_t slab, int64_t j) {
        const unsigned r = (unsigned)(slab < slabs ? slab : slabs - 1) * 64u + (unsigned)lane;
        const unsigned g32 = r / (unsigned)K, b32 = g32 / (unsigned)S;
        const int64_t g = g32, bb = b32;
        const float *px = xyz + (bb * N + j) * 3;
        const float *q = new_xyz + g * 3;
        cxr = ldg(q); cyr = ldg(q + 1); czr = ldg(q + 2);            // (subtracted when the row is put together)
        gx = ldg(px); gy = ldg(px + 1); gz = ldg(px + 2);
        const float *pf = points + (bb * N + j) * D;
#pragma unroll
        for (int k = 0; k < 9; ++k) f[k] = ldg(pf + (k < dlast ? k : dlast));     // always a valid address; columns beyond D are
                                                                                   // dropped where x[] is put together
    };
    constexpr int kStores = 3 + CO / 4;
    auto wait_points = [&]() {                                      // wait A
        if (CO == 64) asm volatile("s_waitcnt vmcnt(20)" : "+v"(gx), "+v"(gy), "+v"(gz), "+v"(cxr), "+v"(cyr), "+v"(czr), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" : "+v"(gx), "+v"(gy), "+v"(gz), "+v"(cxr), "+v"(cyr), "+v"(czr), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]) :: "memory");
    };
    auto wait_index = [&]() {                                       // wait B
        if (CO == 64) asm volatile("s_waitcnt vmcnt(19)" : "+v"(jn) :: "memory");
        else asm volatile("s_waitcnt vmcnt(11)" : "+v"(jn) :: "memory");
    };
    static_assert(kStores == 19 || kStores == 11, "the hand-counted waits above");
    int64_t slab = (int64_t)blockIdx.x * 4 + wave;
    fetch_idx(slab);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(jn) :: "memory");
    fetch_pts(slab, jn);
    fetch_idx(slab + stride);
    // The first trip through the loop has fewer younger operations behind its requests than every later one (no stores of
    // a previous slab yet), so the counted waits of the loop would not cover them: everything requested so far is waited for
    // here, explicitly, with the loaded registers as operands (round 3 issued kStores dummy stores instead -- same-address dead
    // stores that hipcc removed, leaving the prologue correct only by the accident of a compiler-placed vmcnt(0)).
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(gx), "+v"(gy), "+v"(gz), "+v"(cxr), "+v"(cyr), "+v"(czr), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(jn) :: "memory");
    for (; slab < slabs; slab += stride) {
        wait_points();
        float x[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float v = 0.f;
            v = k == xo ? gx - cxr : v; v = k == xo + 1 ? gy - cyr : v; v = k == xo + 2 ? gz - czr : v;
#pragma unroll
            for (int e = 0; e < 9; ++e) v = (k == fo + e && e < D) ? f[e] : v;
            x[k] = v;
        }
        wait_index();
        fetch_pts(slab + stride, jn);                              // its index was requested one slab ago
        fetch_idx(slab + 2 * stride);
        asm volatile("" ::: "memory");                             // the requests stay IN FRONT of this slab's stores (the scheduler
                                                                   // otherwise sinks them behind the stores: vmcnt(0) again)
        typedef float v4f __attribute__((ext_vector_type(4)));
        {
            float *xr = X + (slab * 64 + lane) * ldx;
            *reinterpret_cast<float4 *>(&XT[lane * LX + 12]) = make_float4(0.f, 0.f, 0.f, 0.f);     // the zero quad (k = 12..15)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                *reinterpret_cast<float4 *>(&XT[lane * LX + 4 * q]) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
                {   // a quad beyond the row pitch goes to this thread's dump slot (no branch: see fetch_pts)
                    const v4f v = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                    PN2_STREAM_STORE(v, reinterpret_cast<v4f *>(4 * q < ldx ? xr + 4 * q : pn2_gc_dump + 4 * (blockIdx.x % 1024 * 256 + t)));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // y = x W^T on v_mfma_f32_32x32x2_f32: two 32-row blocks x NBK 32-column blocks, k = 0..15 (12..15 zero)
        float4 a0s[2], a1s[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            a0s[rb] = *reinterpret_cast<const float4 *>(&XT[(rb * 32 + l31) * LX + 4 * lh]);          // k = 4 lh + e
            a1s[rb] = *reinterpret_cast<const float4 *>(&XT[(rb * 32 + l31) * LX + 8 + 4 * lh]);      // k = 8 + 4 lh + e
        }
        __builtin_amdgcn_wave_barrier();                           // every operand is in registers: the tile may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float4 a0 = a0s[rb], a1 = a1s[rb];
#pragma unroll
            for (int j = 0; j < NBK; ++j) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wreg[j][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wreg[j][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wreg[j][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wreg[j][3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wreg[j][4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wreg[j][5], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wreg[j][6], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wreg[j][7], acc, 0, 0, 0);
                // bias, statistics straight from the accumulators (column on the lane), rows into the output tile
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = acc[r] + bj[j];
                    T[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LT + 32 * j + l31] = y;
                    s0 += y;
                    s1 = __builtin_fmaf(y, y, s1);
                }
                st[j][0] += (double)s0;
                st[j][1] += (double)s1;
            }
        }
        __builtin_amdgcn_wave_barrier();                           // the tiles are this wave's own: program order is enough
        {
            float *yb = Y + slab * 64 * (int64_t)ldy;
            constexpr int QPR = CO / 4;                            // 16-byte pieces per row
#pragma unroll
            for (int i = 0; i < QPR; ++i) {
                const int q = lane + 64 * i, row = q / QPR, quad = q - row * QPR;
                const float4 v = *reinterpret_cast<const float4 *>(&T[row * LT + 4 * quad]);
                const v4f vv = {v.x, v.y, v.z, v.w};
                *reinterpret_cast<v4f *>(yb + (int64_t)row * ldy + 4 * quad) = vv;   // plain, not streaming: 92 -> 80 us at 1 M rows here
                                                                                 // (the GEMM epilogues measure the opposite: pn2_common.h)
            }
        }
        __builtin_amdgcn_wave_barrier();                           // ... before the next slab overwrites them
    }
    if (stats != nullptr) {
        __shared__ double red[4][CO][2];
#pragma unroll
        for (int j = 0; j < NBK; ++j) {
            double a0 = st[j][0], a1 = st[j][1];
            a0 += __shfl_xor(a0, 32, 64);
            a1 += __shfl_xor(a1, 32, 64);
            if (lh == 0) { red[wave][32 * j + l31][0] = a0; red[wave][32 * j + l31][1] = a1; }
        }
        __syncthreads();
        if (t < CO) {
            double a0 = 0.0, a1 = 0.0;
            for (int w = 0; w < 4; ++w) { a0 += red[w][t][0]; a1 += red[w][t][1]; }
            double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * CO;
            atomicAdd(rep + t, a0);
            atomicAdd(rep + CO + t, a1);
        }
    }
}


}  // namespace

extern "C" {

int pn2_group_conv_fwd(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S,
                       int K, int D, int xyz_first, const float *W, int ldw, const float *bias, float *X, int ldx, float *Y, int ldy,
                       int C_out, double *stats, pn2_stream_t stream) {
    PN2_CHECK_ARG(xyz && W && bias && X && Y && B > 0 && N > 0 && S > 0 && K > 0 && D >= 0 && (D == 0 || points) && (idx || K == N));
    PN2_CHECK_ARG(ldw >= 3 + D && ldx % 4 == 0 && ldx >= ((3 + D + 3) & ~3) && ldy % 4 == 0 && ldy >= C_out);
    const int64_t P = (int64_t)B * S * K;
    if (3 + D > 12 || D > 9 || D < 1 || !idx || !new_xyz || ldx > 12 || !(C_out == 32 || C_out == 64) || P % 64 != 0 || P >= (1LL << 31))
        return PN2_EUNSUPPORTED;                                   // (the kernel is branch-free: it needs an index, centres and features)
    const int64_t slabs = P / 64;
    int64_t grid = pn2_cdiv(slabs, 4);
    const int64_t cap = (int64_t)pn2_num_cus() * 2;                // persistent: two workgroups per CU (LDS: 4 tiles + W each)
    if (grid > cap) grid = cap;
    if (C_out == 64)
        hipLaunchKernelGGL(group_conv_fwd_kernel<64>, dim3((unsigned)grid), dim3(256), 0, pn2_s(stream), xyz, points, new_xyz, idx, N, S, K,
                           D, xyz_first, W, ldw, bias, X, ldx, Y, ldy, slabs, stats);
    else
        hipLaunchKernelGGL(group_conv_fwd_kernel<32>, dim3((unsigned)grid), dim3(256), 0, pn2_s(stream), xyz, points, new_xyz, idx, N, S, K,
                           D, xyz_first, W, ldw, bias, X, ldx, Y, ldy, slabs, stats);
    return pn2_launch_status();
}

int pn2_group_affine_fwd(const float *Zf, int ldz, const float *xyz, const float *new_xyz, const int64_t *idx,
                         const float *Wx, int ldwx, int B, int N, int S, int K, int C, float *Y, int ldy, double *stats,
                         const pn2_bn_finalize_tail *fin, pn2_stream_t stream) {
    PN2_CHECK_ARG(fin == nullptr || (stats && fin->ticket && fin->gamma && fin->beta && fin->affine));
    FinTail ft{};
    if (fin) {
        const int64_t rows = (int64_t)B * S * K;
        ft = FinTail{fin->ticket, fin->gamma, fin->beta, fin->eps, fin->momentum, fin->running_mean, fin->running_var,
                     fin->num_batches_tracked, fin->affine, 1.0 / (double)rows,
                     rows > 1 ? (double)rows / (double)(rows - 1) : 1.0};
    }
    PN2_CHECK_ARG(Zf && xyz && new_xyz && idx && Wx && Y && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 && C <= 1024 &&
                  ldwx >= 3);
    PN2_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && ldz >= ((C + 3) & ~3) && ldy >= ((C + 3) & ~3));
    const int64_t P = (int64_t)B * S * K;
    const int rpb = 256 / ((C + 3) >> 2);
    int64_t blocks = pn2_cdiv(P, (int64_t)rpb * 8);
    if (blocks > 1024) blocks = 1024;    // 2*C same-address fp64 atomics per workgroup at the end
    hipLaunchKernelGGL(group_affine_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), Zf, ldz, xyz, new_xyz, idx,
                       Wx, ldwx, N, S, K, C, P, Y, ldy, stats, ft);
    return pn2_launch_status();
}

int pn2_group_affine_bwd(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *xyz,
                         const float *new_xyz, const int64_t *idx, int B, int N, int S, int K, int C, float *G, int ldg,
                         float *dWx, int ldwx, pn2_stream_t stream) {
    PN2_CHECK_ARG(dZ && Y && coef && xyz && new_xyz && idx && G && dWx && B > 0 && N > 0 && S > 0 && K > 0 && C > 0 &&
                  C <= 1024);
    PN2_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && ldg % 4 == 0 && ldwx >= 3);
    const int64_t P = (int64_t)B * S * K;
    const int rpb = C < 256 ? 256 / C : 1;
    int64_t blocks = pn2_cdiv(P, (int64_t)rpb * 16);
    if (blocks > 768) blocks = 768;      // every workgroup ends with 3*C same-address atomics on dWx: keep that queue short
    hipLaunchKernelGGL(group_affine_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream), dZ, ldz, Y, ldy, coef,
                       (C + 3) & ~3, xyz, new_xyz, idx, N, S, K, C, P, G, ldg, dWx, ldwx);
    return pn2_launch_status();
}

}  // extern "C"
