// Weight-resident shared-MLP kernels for the narrow, long layers (C_in, C_out multiples of 32, <= 128; P ~ 1e5..1e6 rows):
//   pn2_conv1x1_fwd  -> fwd_res_kernel   Y = act(X) W^T + b, BatchNorm statistics            (model/pointnet_util.py:197,254,312)
//   pn2_conv1x1_bwd  -> bwd_res_kernel   dX = dY W (masked by the previous ReLU), dW += dY^T act(X), the previous layer's
//                                        BatchNorm-backward reductions -- dgrad AND wgrad in ONE pass over dZ / Y / Y_prev
//
// Why a second family next to mlp.hip's streamed-weight GEMMs: on these layers the weight matrix is 4..64 KB -- it fits
// in LDS next to a row tile -- while the streamed kernels re-stage it through LDS for every 64-row tile (more LDS
// write traffic than the activations themselves, one barrier per 16-deep k-step), and the backward pass reads dZ, Y and
// Y_prev twice (dgrad, then wgrad) although both form the same dY.  These layers sit at 3..5.5 TB/s in the streamed
// kernels: HBM-bound, so halving the bytes is worth more than any issue-level tuning.
//
//   * W lives in LDS for the lifetime of a persistent workgroup (one per CU), loaded once.
//   * forward: every WAVE owns 32-row slabs end to end (private LDS staging buffer, 32 x N accumulator slab), so the
//     main loop has no workgroup barrier at all; the waves of a CU drift apart and fill each other's load / epilogue
//     phases on the matrix pipe the way separate workgroups would, but share one copy of W.
//   * backward: the 8 waves of a workgroup share one 64-row tile of dY (formed once, in LDS) and of Y_prev; the tile's
//     work -- dX tiles (contraction over C_out) and dW tiles (contraction over the 64 rows; divisible by row halves
//     because dW is accumulated with atomics anyway) -- is dealt to the waves by a host-side LPT plan so every wave
//     issues the same number of MFMAs.  dW accumulators stay in registers across all tiles of the workgroup.
#include "mlp_loaders.h"
#include "split_bf16.h"
#include <algorithm>

// mlp_wide.hip: the register-stationary forward (with the pooling extrema in its epilogue when Kpool > 0)
int pn2_wide_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                 int64_t P, int K, int N, double *stats, LazyBn lz, hipStream_t s, int64_t *rows_done, int Kpool,
                 const float *pool_gamma, float *pool_ws);

namespace {

constexpr int RES_BM = 64;

// ----------------------------------------------------------------------------------------------- backward plan
// Per wave: at most one dX tile (row block rb, column block cj: C_out / 2 MFMAs) and one dW COLUMN UNIT: all C_out rows
// of dW for the input-channel block dw_cj, contracted over the row quarters [dw_q0, dw_q1) of the 64-row tile (a quarter =
// 16 rows = 8 MFMAs per 32 x 32 tile; dW is accumulated with atomics anyway, so splitting its contraction over waves is
// free).  A column unit covers ALL C_out rows because its tiles are INTERLEAVED, tile e = rows {4 i + e} (C_out = 128;
// {2 i + e} for 64): one ds_read_b128 of dY[p][4 i .. 4 i + 3] then feeds four MFMAs that share one X operand -- with
// contiguous 32-row tiles every MFMA of the dW loops waited on two LDS reads of its own (in-kernel stamps: 184 cycles
// per MFMA in the dW-only waves, which set the length of the compute phase).
// In quarter-units (8 * C_out/32 MFMAs) a dX tile costs 2, a dW column 4, a tile 8 * CI_T in all: CI_T per wave.
struct ResPlan {
    signed char dx_rb[8], dx_cj[8];                 // -1: none
    signed char dw_cj[8], dw_q0[8], dw_q1[8];       // dw_q0 == dw_q1: none
};

inline bool make_res_plan(int CI_T, ResPlan *out) {
    ResPlan p;
    int load[8], next_q[4] = {0, 0, 0, 0};
    for (int w = 0; w < 8; ++w) { p.dx_rb[w] = p.dx_cj[w] = -1; p.dw_cj[w] = 0; p.dw_q0[w] = p.dw_q1[w] = 0; load[w] = 0; }
    if (CI_T < 1 || CI_T > 4) return false;
    for (int i = 0; i < 2 * CI_T; ++i) { p.dx_rb[i] = (signed char)(i & 1); p.dx_cj[i] = (signed char)(i >> 1); load[i] = 2; }
    const int target = CI_T > 2 ? CI_T : 2;          // (CI_T = 1: the two dX tiles alone are 2 units each)
    // waves without a dX tile first (they take the long runs of one column), then the rest, each wave ONE column
    for (int pass = 0; pass < 2; ++pass)
        for (int w = 7; w >= 0; --w) {
            if ((pass == 0) != (p.dx_rb[w] < 0) || p.dw_q0[w] != p.dw_q1[w]) continue;
            int cj = -1;
            for (int c = 0; c < CI_T; ++c)
                if (next_q[c] < 4 && (cj < 0 || next_q[c] < next_q[cj])) cj = c;      // the column with most quarters left
            if (cj < 0) break;
            int take = target - load[w];
            if (take < 1) continue;
            if (take > 4 - next_q[cj]) take = 4 - next_q[cj];
            p.dw_cj[w] = (signed char)cj; p.dw_q0[w] = (signed char)next_q[cj]; p.dw_q1[w] = (signed char)(next_q[cj] + take);
            next_q[cj] += take;
            load[w] += take;
        }
    // leftovers (imbalanced shapes): hand them to waves that still have no column, least loaded first
    for (int c = 0; c < CI_T; ++c)
        while (next_q[c] < 4) {
            int w = -1;
            for (int v = 0; v < 8; ++v)
                if (p.dw_q0[v] == p.dw_q1[v] && (w < 0 || load[v] < load[w])) w = v;
            if (w < 0) return false;
            p.dw_cj[w] = (signed char)c; p.dw_q0[w] = (signed char)next_q[c]; p.dw_q1[w] = 4;
            load[w] += 4 - next_q[c];
            next_q[c] = 4;
        }
    *out = p;
    return true;
}

// The same deal for a 4-wave workgroup (two of them per CU: the narrow pairs whose LDS image fits twice): 2 CI_T units per wave,
// every wave the same mix -- CI_T = 2: one dX tile + two quarters of a column each; CI_T = 1: two waves a dX tile, two waves
// two quarters of the column.
inline bool make_res_plan4(int CI_T, ResPlan *out) {
    ResPlan p;
    for (int w = 0; w < 8; ++w) { p.dx_rb[w] = p.dx_cj[w] = -1; p.dw_cj[w] = 0; p.dw_q0[w] = p.dw_q1[w] = 0; }
    if (CI_T == 2) {
        for (int w = 0; w < 4; ++w) {
            p.dx_rb[w] = (signed char)(w & 1); p.dx_cj[w] = (signed char)(w >> 1);
            p.dw_cj[w] = (signed char)(w >> 1); p.dw_q0[w] = (signed char)(2 * (w & 1)); p.dw_q1[w] = (signed char)(2 * (w & 1) + 2);
        }
    } else if (CI_T == 1) {
        for (int w = 0; w < 2; ++w) { p.dx_rb[w] = (signed char)w; p.dx_cj[w] = 0; }
        for (int w = 2; w < 4; ++w) { p.dw_cj[w] = 0; p.dw_q0[w] = (signed char)(2 * (w - 2)); p.dw_q1[w] = (signed char)(2 * (w - 2) + 2); }
    } else {
        return false;
    }
    *out = p;
    return true;
}

extern __shared__ __attribute__((aligned(16))) float res_lds[];

#ifdef PN2_STAMP
// Diagnostic build only (make STAMP=1): per-phase shader-cycle sums of every wave of the first 64 workgroups of the last
// resident-kernel launch, read back with pn2_debug_stamps_res().  Never compiled into the shipped library.
__device__ unsigned long long pn2_res_stamp_buf[64 * 8 * 8];
// (slots 6 / 7: shader cycles and 100 MHz ticks over the whole stamped span -- their ratio is the clock the kernel held)
#define RSTAMP_DECL unsigned long long rst_t = clock64(), rst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long rst_c0 = rst_t, rst_w0 = wall_clock64();
__device__ unsigned long long pn2_res_abs_buf[64 * 8 * 4];     // 100 MHz ticks: kernel entry, loop start, loop end, exit
#define RABS(wv, i) if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) pn2_res_abs_buf[(blockIdx.x * 8 + (wv)) * 4 + (i)] = wall_clock64();
#define RSTAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = clock64(); rst_acc[i] += n_ - rst_t; rst_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define RSTAMP_FLUSH(wv) { rst_acc[6] = clock64() - rst_c0; rst_acc[7] = wall_clock64() - rst_w0; if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) { for (int i_ = 0; i_ < 8; ++i_) pn2_res_stamp_buf[(blockIdx.x * 8 + (wv)) * 8 + i_] = rst_acc[i_]; } }
#else
#define RSTAMP_DECL
#define RABS(wv, i)
#define RSTAMP(i)
#define RSTAMP_FLUSH(wv)
#endif

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }   // 32x32 C/D layout

// Predicated-off stores go here instead of into a branch: every store is then issued on every path, the code stays
// straight-line and the compiler can COUNT the outstanding memory operations (s_waitcnt vmcnt(N)) -- gfx9 retires loads
// and stores through one in-order counter, so a wait it cannot count becomes vmcnt(0) and stalls on the store
// acknowledgements of the previous tile (1-2 us each).
__device__ float pn2_dump_page[64];
// one 256-byte line per wave for the stores of waves that have nothing to write (all waves of all workgroups on ONE line
// cost 2x the whole kernel: same-line stores serialise chip-wide)
__device__ float pn2_dump_lines[1024 * 8 * 64];

__device__ __forceinline__ void store_or_dump(float v, float *dst, bool valid, int lane) {
    PN2_STREAM_STORE(v, valid ? dst : &pn2_dump_page[lane]);
}

// ----------------------------------------------------------------------------------------------- fused backward
// This layer's dY = c0*dZ + q1*(y - mean) + q0 (BatchNorm backward folded into `coef`, see pn2_bn_bwd_coef) with dZ
// dense [P, Co] or implied by the max-pool (POOLED: dZ[g*Kp + kk, c] = dZp[g, c] if kk == arg[g, c]; Kp a power of two
// that divides 64 or is a multiple of it, so the groups of a 64-row tile are the same for every tile).
// Yp: the previous layer's pre-BN output [P, Ci] (MASKED: X = relu(bn(Yp)) with aff_p, dX masked by X > 0 and reduced
// into red_p) or the plain layer input (!MASKED: X = Yp as stored, dX = dY W unmasked, no reductions).
// The kernel takes WHOLE 64-row tiles only (P % 64 == 0; the host hands a ragged tail to the streamed kernels): no row
// predicates anywhere, every address is a per-tile uniform base plus a loop-invariant 32-bit lane offset.
struct ResDy {
    const float *dZ;                                   // dense [P, ld]
    const float *dZp; const int32_t *arg; int ldo; int kshift;   // pooled: [G, ldo], log2(Kp)
    const float *Y; int ld;                            // Y (and dZ) row pitch
    const float *coef;
    LazyCoef lc;                                       // consumer-side BatchNorm backward: `coef` is filled by the prologue
};

// POOL: 0 dense dZ; 1 pooled with Kp a multiple of 64 (the tile lies in ONE group: one (dZp, arg) quad per thread and
// tile); 2 pooled with Kp == 32 (two groups per tile).  DEPTH: register sets of prefetched tiles.
// DBUF: two dY tiles in LDS.  The waves then form tile n + 1's dY (the VALU-heavy part of a tile: BatchNorm-backward
// transform, max-pool select, LDS stores) in the SAME barrier interval in which they multiply tile n, waves 0-3 before
// their MFMAs and waves 4-7 after theirs: the two waves of a SIMD (w and w + 4) sit in opposite phases, so one's
// transform runs in the shadow of the other's matrix work instead of both queueing for the pipe and then both idling
// it (single buffer: matrix pipe 43-57 % busy, the rest the lock-stepped transform / epilogue phases).
template <int CO_T, int CI_T, int POOL, bool MASKED, int DEPTH, bool DBUF, int NTHR = 512>
__global__ __launch_bounds__(NTHR, 2) void bwd_res_kernel(ResDy dy, const float *__restrict__ Yp, int ldp,
                                                         const float *__restrict__ aff_p, const float *__restrict__ W, int ldw,
                                                         int64_t tiles, float *__restrict__ dX, int ldxo,
                                                         double *__restrict__ red_p, float *__restrict__ dW, int lddw,
                                                         ResPlan plan) {
    constexpr int Co = 32 * CO_T, Ci = 32 * CI_T, LDY = Co + 4, LDP = Ci + 4, QD = Co / 4, QP = Ci / 4;
    constexpr int IT_D = RES_BM * QD / NTHR, IT_P = RES_BM * QP / NTHR;
    static_assert(RES_BM * QD % NTHR == 0 && RES_BM * QP % NTHR == 0, "tiles must split evenly over 512 threads");
    constexpr bool POOLED = POOL != 0;
    // 512 % QD == 0 (Co = 32, 64, 128): a thread's quads t + 512 i all sit in the same channel quad, RPI rows apart, so
    // the pooled operands (one row per GROUP) repeat: slot(i) = which of the NZ distinct (dZp, arg) quads quad i uses
    constexpr int RPI = NTHR / QD;
    static_assert(!POOLED || NTHR % QD == 0, "pooled variants need Co in {32, 64, 128}");
    constexpr int NZ = !POOLED ? IT_D : (POOL == 2 && IT_D > 1 ? 2 : 1);
    auto slot = [](int i) { return POOL == 2 && IT_D > 1 ? (RPI * i) / 32 : 0; };
    static_assert(!DBUF || DEPTH == 1, "the double-buffered form keeps one register set");
    float *Wt = res_lds;                                           // [Ci][LDY]: W transposed, co contiguous
    float *dYs = Wt + Ci * LDY;                                    // [DBUF ? 2 : 1][64][LDY]
    float *Yps = dYs + (DBUF ? 2 : 1) * RES_BM * LDY;              // [64][LDP]
    float *tab = Yps + RES_BM * LDP;                               // coefficient rows c0, q1, q0, mean of this layer: 4 * Co

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;

    RABS(wave, 0)
    lazy_coef_prologue(dy.lc);                                     // consumer-side BatchNorm backward (bn_tail.h)
    // ---- one-time: W^T, coefficient table
    for (int i = t; i < Co * Ci; i += NTHR) {
        const int co = i / Ci, ci = i - co * Ci;
        Wt[ci * LDY + co] = W[(int64_t)co * ldw + ci];
    }
    for (int i = t; i < 4 * Co; i += NTHR) tab[i] = dy.coef[i];

    // ---- this wave's share of every tile (fixed for the whole launch)
    const int dx_rb = plan.dx_rb[wave], dx_cj = plan.dx_cj[wave];
    const int dw_cj = plan.dw_cj[wave], dw_q0 = plan.dw_q0[wave], dw_q1 = plan.dw_q1[wave];
    const int dw_b = dw_cj * 32 + l31;                             // this lane's input channel in the dW column unit
    float xmu = 0.f, xsc = 0.f, xbe = 0.f;
    if (MASKED) {
        Affine a(aff_p, Ci);
        xmu = a.mean[dw_b]; xsc = a.scale[dw_b]; xbe = a.beta[dw_b];
    }
    float emu = 0.f, esc = 0.f, ebe = 0.f, eis = 0.f;             // epilogue constants of the dX tile's column
    const int ecol = (dx_cj < 0 ? 0 : dx_cj) * 32 + l31;
    if (MASKED && dx_rb >= 0) {
        Affine a(aff_p, Ci);
        emu = a.mean[ecol]; esc = a.scale[ecol]; ebe = a.beta[ecol]; eis = a.invstd[ecol];
    }

    f32x16 accw[CO_T];                                             // tile e of the column unit: dW rows {CO_T * i + e}
#pragma unroll
    for (int u = 0; u < CO_T; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[u][r] = 0.f;
    double st0 = 0.0, st1 = 0.0;

    // ---- loop-invariant lane offsets (elements): dY quads idx = t + 512 i -> (row, quad); Y_prev quads likewise.
    // Every global address is a kernel-argument base plus a 32-bit element offset = tile term + lane term.
    unsigned od[IT_D], og[NZ], kk[IT_D], op[IT_P];
#pragma unroll
    for (int i = 0; i < IT_D; ++i) {
        const int idx = t + NTHR * i, row = idx / QD, q = idx - row * QD;
        od[i] = (unsigned)row * (unsigned)dy.ld + 4u * q;
        kk[i] = 0;
        if (POOLED) {                                              // group of the row relative to the tile's first group
            const unsigned gsub = POOL == 1 ? 0u : (unsigned)row >> 5;
            og[slot(i)] = gsub * (unsigned)dy.ldo + 4u * q;
            kk[i] = POOL == 1 ? (unsigned)row : (unsigned)row & 31u;     // + (m0 mod Kp) when Kp > 64
        }
    }
#pragma unroll
    for (int i = 0; i < IT_P; ++i) {
        const int idx = t + NTHR * i, row = idx / QP, q = idx - row * QP;
        op[i] = (unsigned)row * (unsigned)ldp + 4u * q;
    }

    // Two register sets: the loads of tile n + 2 are issued as soon as tile n has been written to LDS, so one to two
    // tiles are always in flight -- with a single set the memory pipe idled from "tile landed" to "next fetch issued"
    // and a tile cost T_mem + T_compute (measured: matrix pipe 47 % busy, a third of the wave cycles in s_waitcnt).
    struct Regs { float4 y[IT_D]; float4 z[NZ]; int4 a[POOLED ? NZ : 1]; float4 p[IT_P]; };
    Regs rs[DEPTH];
    auto fetch_dy = [&](Regs &R, int64_t tile) {
        const unsigned tl = (unsigned)(tile < tiles ? tile : tiles - 1);   // past the end: re-read the last tile (never used)
        const unsigned ty = tl * (unsigned)(RES_BM * dy.ld);
        if (POOLED) {
            const unsigned g0 = POOL == 1 ? (tl * RES_BM) >> dy.kshift : tl << 1;
            const unsigned tg = g0 * (unsigned)dy.ldo;
#pragma unroll
            for (int i = 0; i < IT_D; ++i) R.y[i] = ld4(dy.Y + (ty + od[i]));
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                R.z[j] = ld4(dy.dZp + (tg + og[j]));
                R.a[j] = ld4i(dy.arg + (tg + og[j]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < IT_D; ++i) {
                R.y[i] = ld4(dy.Y + (ty + od[i]));
                R.z[i] = ld4(dy.dZ + (ty + od[i]));
            }
        }
    };
    auto fetch_p = [&](Regs &R, int64_t tile) {
        const unsigned tp = (unsigned)(tile < tiles ? tile : tiles - 1) * (unsigned)(RES_BM * ldp);
#pragma unroll
        for (int i = 0; i < IT_P; ++i) R.p[i] = ld4(Yp + (tp + op[i]));
    };
    // registers -> LDS: dY of one tile, formed here once per row.  Where 512 % QD == 0 (C_out = 32, 64, 128) a thread's items
    // all sit in ONE channel quad: its four coefficient quads are read from the table once per launch (DP_HOIST) instead of
    // with every item -- they were 16 of the 23 ds_*_b128 a thread issues per tile of the 128 x 96 pair, all eight waves at
    // the same time with the matrix pipe idle (same-box A/B, round 4: 128 x 96 pooled 603 -> 593 us, 128 x 64 244 -> 236,
    // 64 x 64 162 -> 158).
    constexpr bool DP_HOIST = NTHR % QD == 0;
    DyParams dpk;
    auto finish_dy = [&](Regs &R, float *dst, int64_t tile) {
        const unsigned kbase = POOL == 1 ? (unsigned)(tile * RES_BM) & ((1u << dy.kshift) - 1u) : 0u;
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            const int idx = t + NTHR * i, row = idx / QD, q = idx - row * QD;
            const DyParams dp = DP_HOIST ? dpk : dy_params_tab(tab, Co, 4 * q, true);
            float4 dz = R.z[POOLED ? slot(i) : i];
            if (POOLED) {
                const int4 a = R.a[slot(i)];
                const int k = (int)(kk[i] + kbase);
                dz.x = a.x == k ? dz.x : 0.f; dz.y = a.y == k ? dz.y : 0.f;
                dz.z = a.z == k ? dz.z : 0.f; dz.w = a.w == k ? dz.w : 0.f;
            }
            *reinterpret_cast<float4 *>(&dst[row * LDY + 4 * q]) = dy_from(dz, R.y[i], dp);
        }
    };
    auto write_yp = [&](Regs &R) {
#pragma unroll
        for (int i = 0; i < IT_P; ++i) {
            const int idx = t + NTHR * i, row = idx / QP, q = idx - row * QP;
            *reinterpret_cast<float4 *>(&Yps[row * LDP + 4 * q]) = R.p[i];
        }
    };
    // this wave's share of one tile: dX tile + its sixteen stores, dW units
    auto compute = [&](const float *dYt, int64_t tile) {
        // ---- dX tile: rows rb*32.., columns cj*32.. ; contraction over Co, both operands 4 k-values per ds_read_b128
        float outv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) outv[r] = 0.f;
        if (dx_rb >= 0) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *ap = &dYt[(dx_rb * 32 + l31) * LDY + 4 * lh];
            const float *bp = &Wt[ecol * LDY + 4 * lh];
#pragma unroll
            for (int kb = 0; kb < Co / 8; ++kb) {
                const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * kb);
                const float4 b = *reinterpret_cast<const float4 *>(bp + 8 * kb);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            const float *yq = &Yps[(dx_rb * 32 + 4 * lh) * LDP + ecol];
            float yv[16];
            if (MASKED) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = yq[((r & 3) + 8 * (r >> 2)) * LDP];      // immediate offsets, one wait
            }
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dz = acc[r];
                if (MASKED) {
                    const float y = yv[r];
                    dz = bn_act(y, emu, esc, ebe) > 0.f ? dz : 0.f;
                    s0 += dz;
                    s1 = __builtin_fmaf(dz, (y - emu) * eis, s1);
                }
                outv[r] = dz;
            }
            if (MASKED) { st0 += (double)s0; st1 += (double)s1; }
        }
        // The sixteen stores are issued by EVERY wave (a wave without a dX tile writes zeros to a dump line of its own): the
        // same count of memory operations on every path, so the wait for a prefetched tile stays an exact s_waitcnt
        // vmcnt(N) that leaves the younger loads and these stores in flight -- counted against the shorter path it
        // drained them.  Address = base + one 32-bit offset that changes per tile (nothing to hoist into sixteen
        // address registers).
        {
            float *sb = dx_rb >= 0 ? dX : pn2_dump_lines;
            const unsigned xo = dx_rb >= 0 ? (unsigned)tile * (unsigned)(RES_BM * ldxo) + (unsigned)(dx_rb * 32 + 4 * lh) * (unsigned)ldxo + (unsigned)ecol
                                           : ((blockIdx.x & 1023u) * 8u + (unsigned)wave) * 64u + (unsigned)lane;
            const unsigned xs = dx_rb >= 0 ? (unsigned)ldxo : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) PN2_STREAM_STORE(outv[r], sb + (xo + (unsigned)((r & 3) + 8 * (r >> 2)) * xs));
        }
        // ---- dW column unit: row quarters [q0, q1) of the tile, 8 row pairs each; per pair ONE wide read of dY[p][CO_T i ..]
        // (the interleaved tiles' A operands) and one of Y_prev[p][ci], then CO_T MFMAs
        for (int q = dw_q0; q < dw_q1; ++q) {
            const float *ap = &dYt[(16 * q + lh) * LDY + CO_T * l31];
            const float *bp = &Yps[(16 * q + lh) * LDP + dw_b];
            float a[8][CO_T], bb[8];
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                if (CO_T == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(ap + 2 * pp * LDY);
                    a[pp][0] = v.x; a[pp][1] = v.y; a[pp][2 % CO_T] = v.z; a[pp][3 % CO_T] = v.w;
                } else if (CO_T == 2) {
                    const float2 v = *reinterpret_cast<const float2 *>(ap + 2 * pp * LDY);
                    a[pp][0] = v.x; a[pp][1 % CO_T] = v.y;
                } else {
#pragma unroll
                    for (int e = 0; e < CO_T; ++e) a[pp][e] = ap[2 * pp * LDY + e];
                }
                bb[pp] = bp[2 * pp * LDP];
            }
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                float x = bb[pp];
                if (MASKED) x = fmaxf(bn_act(x, xmu, xsc, xbe), 0.f);
#pragma unroll
                for (int e = 0; e < CO_T; ++e) accw[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[pp][e], x, accw[e], 0, 0, 0);
            }
        }
    };

    int64_t tile = blockIdx.x;
    RSTAMP_DECL
    RABS(wave, 1)
    if (!DBUF) {
        auto step = [&](Regs &R, int64_t tl) {
            finish_dy(R, dYs, tl);
            write_yp(R);
            RSTAMP(0)
            fetch_dy(R, tl + DEPTH * (int64_t)G);                  // this set is free again: DEPTH tiles ahead
            fetch_p(R, tl + DEPTH * (int64_t)G);
            RSTAMP(1)
            __syncthreads();
            RSTAMP(2)
            compute(dYs, tl);
            RSTAMP(3)
            __syncthreads();                                       // tile consumed: the next one may land
            RSTAMP(4)
        };
        fetch_dy(rs[0], tile);
        fetch_p(rs[0], tile);
        if (DEPTH > 1) { fetch_dy(rs[DEPTH - 1], tile + G); fetch_p(rs[DEPTH - 1], tile + G); }
        __syncthreads();                                           // Wt and tab are in place
        if (DP_HOIST) dpk = dy_params_tab(tab, Co, 4 * (t % QD), true);
        while (tile < tiles) {
            step(rs[0], tile);
            tile += G;
            if (DEPTH > 1) {
                if (tile >= tiles) break;
                step(rs[DEPTH - 1], tile);
                tile += G;
            }
        }
    } else {
        Regs &R = rs[0];
        const bool early = (wave & 4) == 0;
        fetch_dy(R, tile);
        fetch_p(R, tile);
        __syncthreads();                                           // Wt and tab are in place
        if (DP_HOIST) dpk = dy_params_tab(tab, Co, 4 * (t % QD), true);
        if (tile < tiles) {
            finish_dy(R, dYs, tile);
            write_yp(R);
            fetch_dy(R, tile + G);
            fetch_p(R, tile + G);
        }
        __syncthreads();
        int cur = 0;
        while (tile < tiles) {
            const bool more = tile + G < tiles;
            const float *dYc = dYs + cur * (RES_BM * LDY);
            float *dYn = dYs + (cur ^ 1) * (RES_BM * LDY);
            if (early) {
                if (more) { finish_dy(R, dYn, tile + G); fetch_dy(R, tile + 2 * (int64_t)G); }
                compute(dYc, tile);
            } else {
                compute(dYc, tile);
                if (more) { finish_dy(R, dYn, tile + G); fetch_dy(R, tile + 2 * (int64_t)G); }
            }
            __syncthreads();                                       // tile n consumed by every wave; dY of tile n + 1 complete
            if (more) { write_yp(R); fetch_p(R, tile + 2 * (int64_t)G); }
            __syncthreads();                                       // Y_prev of tile n + 1 in place
            tile += G;
            cur ^= 1;
        }
    }

    RSTAMP_FLUSH(wave)
    RABS(wave, 2)
    // ---- flush: dW partial tiles (tile e holds the rows {CO_T i + e}: 128 contiguous bytes per half-wave), the dX column's
    // two reductions.  (Summing the waves' partials in LDS first and adding one block per workgroup was measured SLOWER:
    // 128x128 at 65 536 rows 84 -> 106 us.)
    if (dw_q0 < dw_q1) {
#pragma unroll
        for (int e = 0; e < CO_T; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                atomicAdd(dW + (int64_t)(CO_T * acc_row(r, lh) + e) * lddw + dw_b, accw[e][r]);
    }
    if (MASKED && red_p != nullptr && dx_rb >= 0) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (lh == 0) {
            double *rep = red_p + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * Ci;
            atomicAdd(rep + ecol, st0);
            atomicAdd(rep + Ci + ecol, st1);
        }
    }
    RABS(wave, 3)
}

inline size_t bwd_res_lds_bytes(int Co, int Ci, bool dbuf) {
    return sizeof(float) * ((size_t)Ci * (Co + 4) + (dbuf ? 2 : 1) * RES_BM * (Co + 4) + RES_BM * (Ci + 4) + 4 * Co);
}

template <int CO_T, int CI_T, int POOL, bool MASKED, int DEPTH, bool DBUF, int NTHR = 512>
int launch_bwd_res_impl(ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t tiles, float *dX, int ldxo,
                        double *red_p, float *dW, int lddw, hipStream_t s);

template <int CO_T, int CI_T, int POOL, bool MASKED>
int launch_bwd_res(ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t tiles, float *dX, int ldxo,
                   double *red_p, float *dW, int lddw, hipStream_t s) {
    // Single dY buffer, two register sets of prefetched tiles on the light pairs (their tile's MFMA work is shorter than
    // its memory time).  The double-buffered form (DBUF: tile n + 1 transformed inside tile n's barrier interval, waves 0-3
    // and 4-7 in opposite phases) was MEASURED SLOWER on every pair
    // (128x96 pooled 701 vs 676 us, 96x64 408 vs 385, 64x64 156 vs 136 -- profiles/r02_kernel_microbench.txt): the
    // transform did not hide under the partner wave's MFMAs, it only delayed this wave's own.
    constexpr int DEPTH = CO_T * CI_T <= 6 ? 2 : 1;
    // Two 4-wave workgroups per CU instead of one of eight (round 4): the lock-stepped phases of one workgroup -- forming dY,
    // issuing the next tile's requests, the barriers: 25 - 30 % of a tile with the matrix pipe idle (tools/stamp_res.py) -- then
    // run beside the OTHER workgroup's products.  Where the LDS image fits twice and a 4-wave plan exists (C_in <= 64);
    // PN2_RES_HALF=0: the 8-wave form (A/B runs).  Same-box A/B (us): 96 x 64 at 1 M rows 404 -> 385, 64 x 64 at 524 k 153 -> 142,
    // 64 x 32 pooled at 524 k 89 -> 81, 32 x 32 at 524 k 66 -> 63; at 262 k rows the half-size workgroups' tails cost more
    // (52 -> 55, 41 -> 47: not taken there); one register set (a second one: 142 -> 146, 81 -> 86); the step: within noise.
    const int half = pn2_opt(PN2_OPT_RES_HALF);
    if constexpr (CI_T <= 2 && 2 * sizeof(float) * (32 * CI_T * (32 * CO_T + 4) + RES_BM * (32 * CO_T + 4) + RES_BM * (32 * CI_T + 4) + 128 * CO_T) <= 160 * 1024) {
        if (half && tiles >= 32 * (int64_t)pn2_num_cus())           // (from 524 288 rows on: below, the two half-size workgroups' tails cost more)
            return launch_bwd_res_impl<CO_T, CI_T, POOL, MASKED, 1, false, 256>(dy, Yp, ldp, aff_p, W, ldw, tiles, dX, ldxo, red_p, dW, lddw, s);
    }
    return launch_bwd_res_impl<CO_T, CI_T, POOL, MASKED, DEPTH, false>(dy, Yp, ldp, aff_p, W, ldw, tiles, dX, ldxo, red_p, dW, lddw, s);
}

template <int CO_T, int CI_T, int POOL, bool MASKED, int DEPTH, bool DBUF, int NTHR>
int launch_bwd_res_impl(ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t tiles, float *dX, int ldxo,
                        double *red_p, float *dW, int lddw, hipStream_t s) {
    ResPlan plan;
    if (!(NTHR == 256 ? make_res_plan4(CI_T, &plan) : make_res_plan(CI_T, &plan))) return PN2_EINVAL;
    const size_t lds = bwd_res_lds_bytes(32 * CO_T, 32 * CI_T, DBUF);
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&bwd_res_kernel<CO_T, CI_T, POOL, MASKED, DEPTH, DBUF, NTHR>), raised) != PN2_OK)
        return PN2_ELAUNCH;
    const int64_t cap = (int64_t)pn2_num_cus() * (512 / NTHR);     // one 8-wave workgroup per CU, or two of four waves
    PN2_NOTE_KERNEL(bwd_res_kernel<CO_T, CI_T, POOL, MASKED, DEPTH, DBUF, NTHR>);
    hipLaunchKernelGGL((bwd_res_kernel<CO_T, CI_T, POOL, MASKED, DEPTH, DBUF, NTHR>), dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(NTHR), lds, s, dy, Yp,
                       ldp, aff_p, W, ldw, tiles, dX, ldxo, red_p, dW, lddw, plan);
    return pn2_launch_status();
}

#ifndef PN2_SPLIT_RES_DXFREE
#define PN2_SPLIT_RES_DXFREE 1
#endif
// ----------------------------------------------------------------------------------------------- fused backward, bf16 pipe (round 5)
// bwd_res_kernel's job -- dX = dY W masked by the previous ReLU with its BatchNorm-backward sums, dW += dY^T act(Y_prev), ONE pass
// over dZ / Y / Y_prev -- with the fp32 products formed from exact three-way bf16 splits (split_bf16.h; mlp_wide.hip's
// split_nt_kernel has the arithmetic and its measured error): 96 instead of 256 matrix-pipe cycles per 32 x 32 x 16 block.  On
// v_mfma_f32_32x32x2_f32 these layers are matrix-pipe bound (128 x 96 at 1 M rows: 51.5 GFLOP = 330 us at the pipe's peak,
// 540 us measured, against 270 us for its 1.34 GB at 5 TB/s); here they are bound by HBM.
//   * 32-row chunks, double-buffered: dY and X = act(Y_prev) are transformed exactly as the fp32 kernel forms them, split, and
//     stored as three bf16 images each in the dual-use layout of split_tn_kernel (rows of 128 channels, tr_img_off): dX reads
//     dY along the rows (ds_read_b128), dW reads dY and X down the columns (ds_read_b64_tr_b16).  ONE barrier per chunk.
//   * waves 0 .. CI_T - 1 own the chunk's dX tiles (their 32 columns of W pre-split in registers), the 32 x 32 tiles of dW are
//     dealt to the other waves and stay in their accumulators for the whole launch.  The deal is not balanced to the MFMA (a
//     dX tile is C_out / 32 units, a dW tile one) and does not need to be: the heaviest SIMD of 128 x 96 has 7 of the chunk's
//     24 units, 2 700 matrix-pipe cycles per chunk = 165 us per launch, under the memory time.
//   * the raw Y_prev chunk is kept in LDS as well (fp32): the mask and the sums of the dX epilogue need the pre-BatchNorm value.
// FUSE0 (round 6): this layer's INPUT is the output of a FIRST layer whose own input is the 48-byte grouped row X0 (sa1 of the
// segmentation nets: 12 -> 64 -> ...), and nobody needs a gradient with respect to X0.  The masked dX tile of this kernel IS that first
// layer's dZ; the only thing its backward still wants from it (pn2_conv1x1_wgrad_cf: BatchNorm terms in closed form) is
// sum_p dZ[p, c] x0[p, j] -- 12 products per element, formed here straight from the dX registers against the chunk's X0 rows in LDS and
// added into that kernel's scratch.  dX is then NEVER written (nor read back): 0.8 GB and two launches' work per MSG-SemSeg step.
template <int CO_T, int CI_T, bool POOLED, bool MASKED, bool FUSE0 = false>
__global__ __launch_bounds__(512, 1) void split_bwd_res_kernel(ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw,
                                                               int64_t chunks, float *dX, int ldxo, double *red_p, float *dW, int lddw,
                                                               const float *X0 = nullptr, int ld0 = 0, float *part0 = nullptr) {
    constexpr int Co = 32 * CO_T, Ci = 32 * CI_T, BP = 32, NT = 512, QD = Co / 4, QP = Ci / 4, LDP = Ci + 4;
    constexpr int PANEL = BP * 256, BUF = 6 * PANEL;                // one 128-channel panel per piece: dY hi / mid / lo, X hi / mid / lo
    constexpr int KBX = Co / 16;                                    // contraction blocks of a dX tile
    constexpr int NDW = 8 - CI_T, NTILE = CO_T * CI_T, TW = (NTILE + NDW - 1) / NDW;     // dW tiles: waves CI_T .. 7, TW each at most
    // DXFREE (round 6): where only one or two waves own dX tiles and the others hold one weight-gradient tile each (96 x 64, 64 x 64,
    // 32 x 32), the dX waves were the long pole -- their share of the staging, then 6 C_out / 16 MFMAs, then the mask epilogue, while
    // the other waves spent 40 - 57 % of the loop at the barrier (in-kernel stamps, profiles/r05c_stamp_split_bwd_res.txt).  There the
    // dX waves do NO staging: the six or seven other waves stage the whole chunk (the structure split_bwd_cf_kernel was built with).
    // In the serial step (real activations): 96 x 64 at 1 M rows 269.6 -> 242.9 us, 64 x 64 109.9 -> 113.2; MSG step 4.986 -> 4.959 ms
    // over three alternating runs.  (Alone on random operands the same kernels measure 301 -> 317 and 111 -> 117: the microbench's
    // dense random data keeps the staging waves' split arithmetic and the chip's clock elsewhere than the step's ReLU-sparse rows do.)
    constexpr bool DXFREE = CI_T <= 2 && CO_T <= 3 && PN2_SPLIT_RES_DXFREE;
    constexpr int NTS = DXFREE ? NT - 64 * CI_T : NT;               // staging threads
    constexpr int IT_D = (BP * QD + NTS - 1) / NTS, IT_P = (BP * QP + NTS - 1) / NTS;
    static_assert(!POOLED || NTS % QD == 0, "pooled: one channel quad per thread");
    unsigned char *lds_b = reinterpret_cast<unsigned char *>(res_lds);
    float *Yps = res_lds + (2 * BUF) / 4;                           // [2][BP][LDP]: the raw Y_prev chunk
    // SIGN ALTERNATION (round 6).  v_mfma_f32_32x32x16_bf16 does not round its accumulation to nearest: against fp64 every product sum
    // comes out LOW by ~1.4e-8 of its magnitude whatever its sign (tools/exp/split_bias_probe.py: mean error -1.4e-8 |dX| for dX > 0 and
    // for dX < 0 alike; the fp32 MFMA: 1e-10).  Invisible per element, but the BatchNorm-backward sums add dX over 10^5 - 10^7 rows of
    // mixed sign, where a one-sided error grows like P against a sum that grows like sqrt(P): 1.4e-6 of the sum at 2^17 rows, 5e-6 at
    // 2^22.  So every ODD chunk is computed NEGATED -- dY and X are staged with the opposite sign (negated coefficient tables, a median
    // in place of the maximum: no extra instruction; dW = (-dY)^T (-X) is unchanged), the dX tile gets its sign back in the epilogue --
    // and the one-sided error alternates in sign from chunk to chunk: it cancels in every sum over rows.
    float *ctab = Yps + 2 * BP * LDP;                               // [2][4 * Co]: c0, q1, q0, mean of this layer; then -c0, -q1, -q0, mean
    float *xtab = ctab + 8 * Co;                                    // [2][3 * Ci]: mean, scale, beta of the previous BatchNorm; then mean, -scale, -beta
    float *x0s = xtab + 6 * Ci;                                     // FUSE0: [2][BP][12], the chunk's rows of the first layer's input
    static_assert(!FUSE0 || (MASKED && !POOLED && !(CO_T == 4 && CI_T < 4)), "FUSE0: a hidden layer behind a first layer");
    // WL_LDS (C_out = 128 with C_in < 128): a dX wave's `lo` fragments -- the operand of one MFMA in six -- live in LDS, 8 KiB per
    // wave, written and read by that wave alone: 32 registers for the second accumulator (split_nt_kernel does the same at K = 196)
    constexpr bool WL_LDS = CO_T == 4 && CI_T < 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;

    RABS(wave, 0)
    lazy_coef_prologue(dy.lc);
    for (int i = t; i < 4 * Co; i += NT) { const float v = dy.coef[i]; ctab[i] = v; ctab[4 * Co + i] = i < 3 * Co ? -v : v; }
    if (MASKED)
        for (int i = t; i < 3 * Ci; i += NT) { const float v = aff_p[i]; xtab[i] = v; xtab[3 * Ci + i] = i < Ci ? v : -v; }

    // ---- roles (the two kinds of wave run separate instantiations of the chunk loop: a dX wave's 96 fragment registers and a dW
    // wave's 64 accumulator registers then share the register file instead of adding up)
    const bool has_dx = wave < CI_T;
    const int ts = DXFREE ? (has_dx ? 0 : t - 64 * CI_T) : t;       // staging thread index
    const int ecol = (has_dx ? wave : 0) * 32 + l31;                // the dX tile's column of this lane
    // ---- staging: dY item i of thread t = quad q of row `row` (idx = t + 512 i; a thread without an i-th item repeats its last)
    struct Raw { float4 y[IT_D]; float4 z[POOLED ? 1 : IT_D]; int4 a[1]; float4 p[IT_P]; float4 x0; };
    // D2: two register sets, chunk j's requests live in set j & 1 and go out two chunks ahead of their use.  (C_out = 128: one set
    // -- a dX wave holds 96 fragment registers there, and its chunk is longer than a memory latency anyway.)
    constexpr bool D2 = CO_T < 4 && !(FUSE0 && CO_T == 3);          // (96 x 64 with FUSE0: twelve more live registers per dX lane -- one set)
    constexpr bool DX2 = CO_T * CI_T < 16;                          // two alternating dX accumulators (128 x 128: no registers for them, no LDS for WL_LDS)
    Raw raw0, raw1;
    auto d_item = [&](int i, int &row, int &q) { const int idx = (NTS * (i + 1) > BP * QD) ? min(ts + NTS * i, BP * QD - 1) : ts + NTS * i; row = idx / QD; q = idx - row * QD; };
    auto p_item = [&](int i, int &row, int &q) { const int idx = (NTS * (i + 1) > BP * QP) ? min(ts + NTS * i, BP * QP - 1) : ts + NTS * i; row = idx / QP; q = idx - row * QP; };
    auto fetch = [&](Raw &raw, int64_t chunk) {
        const unsigned m0 = (unsigned)(chunk < chunks ? chunk : chunks - 1) * BP;      // past the end: re-read the last chunk (never used)
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            int row, q;
            d_item(i, row, q);
            const unsigned o = (m0 + (unsigned)row) * (unsigned)dy.ld + 4u * (unsigned)q;
            raw.y[i] = ld4(dy.Y + o);
            if (!POOLED) raw.z[POOLED ? 0 : i] = ld4(dy.dZ + o);
        }
        if (POOLED) {                                               // the chunk lies inside ONE group (Kp a multiple of 32)
            const unsigned o = (m0 >> dy.kshift) * (unsigned)dy.ldo + 4u * (unsigned)(ts % QD);
            raw.z[0] = ld4(dy.dZp + o);
            raw.a[0] = ld4i(dy.arg + o);
        }
#pragma unroll
        for (int i = 0; i < IT_P; ++i) {
            int row, q;
            p_item(i, row, q);
            raw.p[i] = ld4(Yp + ((m0 + (unsigned)row) * (unsigned)ldp + 4u * (unsigned)q));
        }
        if (FUSE0) {                                                // threads 0 .. 95: quad t % 3 of row t / 3 (the others repeat the last)
            const unsigned i0 = (unsigned)(ts < 3 * BP ? ts : 3 * BP - 1);
            raw.x0 = ld4(X0 + ((m0 + i0 / 3u) * (unsigned)ld0 + 4u * (i0 % 3u)));
        }
    };
    auto store_split = [&](unsigned char *img, int row, int q, const float4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0, l0);
        split2(v.z, v.w, h1, m1, l1);
        const unsigned o = tr_img_off(row, q >> 1) + 8u * (unsigned)(q & 1);
        *reinterpret_cast<uint2 *>(img + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(img + o + PANEL) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(img + o + 2 * PANEL) = make_uint2(l0, l1);
    };
    auto stage = [&](Raw &raw, int64_t chunk, int buf) {
        unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * PANEL;
        float *yp = Yps + buf * (BP * LDP);
        const unsigned m0 = (unsigned)(chunk < chunks ? chunk : chunks - 1) * BP;
        const int kbase = POOLED ? (int)(m0 & ((1u << dy.kshift) - 1u)) : 0;
        const int par = (int)(chunk & 1);                           // (uniform) odd chunks are staged negated
        const float *ct = ctab + par * (4 * Co), *xt = xtab + par * (3 * Ci);
        const float lim = par ? -INFINITY : INFINITY;               // med3(t, 0, +inf) = max(t, 0);  med3(-t, 0, -inf) = min(-t, 0) = -max(t, 0)
        const unsigned sm = (unsigned)par << 31;
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            int row, q;
            d_item(i, row, q);
            const DyParams dp = dy_params_tab(ct, Co, 4 * q, true);
            float4 dz = raw.z[POOLED ? 0 : i];
            if (POOLED) {
                const int4 a = raw.a[0];
                const int k = kbase + row;
                dz.x = a.x == k ? dz.x : 0.f; dz.y = a.y == k ? dz.y : 0.f; dz.z = a.z == k ? dz.z : 0.f; dz.w = a.w == k ? dz.w : 0.f;
            }
            store_split(ia, row, q, dy_from(dz, raw.y[i], dp));
        }
#pragma unroll
        for (int i = 0; i < IT_P; ++i) {
            int row, q;
            p_item(i, row, q);
            float4 x = raw.p[i];
            *reinterpret_cast<float4 *>(&yp[row * LDP + 4 * q]) = x;
            if (MASKED) {
                const float4 mu = *reinterpret_cast<const float4 *>(&xt[4 * q]), sc = *reinterpret_cast<const float4 *>(&xt[Ci + 4 * q]);
                const float4 be = *reinterpret_cast<const float4 *>(&xt[2 * Ci + 4 * q]);
                x.x = __builtin_amdgcn_fmed3f(bn_act(x.x, mu.x, sc.x, be.x), 0.f, lim); x.y = __builtin_amdgcn_fmed3f(bn_act(x.y, mu.y, sc.y, be.y), 0.f, lim);
                x.z = __builtin_amdgcn_fmed3f(bn_act(x.z, mu.z, sc.z, be.z), 0.f, lim); x.w = __builtin_amdgcn_fmed3f(bn_act(x.w, mu.w, sc.w, be.w), 0.f, lim);
            } else {
                x.x = __uint_as_float(__float_as_uint(x.x) ^ sm); x.y = __uint_as_float(__float_as_uint(x.y) ^ sm);
                x.z = __uint_as_float(__float_as_uint(x.z) ^ sm); x.w = __uint_as_float(__float_as_uint(x.w) ^ sm);
            }
            store_split(ib, row, q, x);
        }
        if (FUSE0 && ts < 3 * BP) *reinterpret_cast<float4 *>(&x0s[buf * (BP * 12) + 4 * ts]) = raw.x0;    // [row][12]: 4 t = 12 (t / 3) + 4 (t % 3)
    };
    auto raw_landed = [&](Raw &raw) {                                       // (see split_nt_kernel: hipcc then waits for the requests, not for the stores)
#pragma unroll
        for (int i = 0; i < IT_D; ++i) {
            asm volatile("" : "+v"(raw.y[i].x), "+v"(raw.y[i].y), "+v"(raw.y[i].z), "+v"(raw.y[i].w));
            if (!POOLED) asm volatile("" : "+v"(raw.z[POOLED ? 0 : i].x), "+v"(raw.z[POOLED ? 0 : i].y), "+v"(raw.z[POOLED ? 0 : i].z), "+v"(raw.z[POOLED ? 0 : i].w));
        }
        if (POOLED) {
            asm volatile("" : "+v"(raw.z[0].x), "+v"(raw.z[0].y), "+v"(raw.z[0].z), "+v"(raw.z[0].w));
            asm volatile("" : "+v"(raw.a[0].x), "+v"(raw.a[0].y), "+v"(raw.a[0].z), "+v"(raw.a[0].w));
        }
#pragma unroll
        for (int i = 0; i < IT_P; ++i) asm volatile("" : "+v"(raw.p[i].x), "+v"(raw.p[i].y), "+v"(raw.p[i].z), "+v"(raw.p[i].w));
        if (FUSE0) asm volatile("" : "+v"(raw.x0.x), "+v"(raw.x0.y), "+v"(raw.x0.z), "+v"(raw.x0.w));
    };
    // transposed fragment (split_tn_kernel's): channels 32 cblk + l31, rows 16 pb + 8 lh + 0 .. 7 of the chunk, of one piece image
    const int g16 = lane >> 4, j16 = lane & 15, tq = j16 >> 2, tp = j16 & 3;
    auto frag_t = [&](const unsigned char *img, int cblk, int pb) {
        const int c0 = (cblk * 32 + 16 * (g16 & 1)) >> 3;
        SplitFrag f;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 16 * pb + 8 * (g16 >> 1) + 4 * r2 + tq;
            const unsigned o = tr_img_off(row, c0 + (tp >> 1)) + 8u * (unsigned)(tp & 1);
            const pn2_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                reinterpret_cast<__attribute__((address_space(3))) pn2_s16x4 *>((__attribute__((address_space(3))) unsigned char *)(img) + o));
            const uint2 u = __builtin_bit_cast(uint2, v);
            f.u[2 * r2] = u.x; f.u[2 * r2 + 1] = u.y;
        }
        return f;
    };

    auto run = [&](auto dx_tag) {
        constexpr bool DXW = decltype(dx_tag)::value;               // this wave owns a dX tile (else: dW tiles)
        SplitFrag wh[DXW ? KBX : 1], wm[DXW ? KBX : 1], wl[DXW && !WL_LDS ? KBX : 1];       // W[co][ecol], co = 16 kb + 8 lh + 0 .. 7
        uint4 *wl_lds = reinterpret_cast<uint4 *>(xtab + 6 * Ci) + (size_t)(DXW ? wave : 0) * KBX * 64 + lane;
        float emu = 0.f, esc = 0.f, ebe = 0.f, eis = 0.f;
        // dW tile (wave - CI_T) + j NDW: rows 32 mb .., columns 32 nb ..
        f32x16 accw[DXW ? 1 : TW];
        double st0 = 0.0, st1 = 0.0;
        float g0[DXW && FUSE0 ? 12 : 1];                            // FUSE0: sum_p dZ[p, ecol] x0[p, j] of this lane's rows
#pragma unroll
        for (int j = 0; j < (DXW && FUSE0 ? 12 : 1); ++j) g0[j] = 0.f;
        if (DXW) {
            float v[KBX][8];
#pragma unroll
            for (int kb = 0; kb < KBX; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[kb][e] = W[(int64_t)(16 * kb + 8 * lh + e) * ldw + ecol];
#pragma unroll
            for (int kb = 0; kb < KBX; ++kb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(v[kb][2 * e], v[kb][2 * e + 1], wh[DXW ? kb : 0].u[e], wm[DXW ? kb : 0].u[e], wl[DXW && !WL_LDS ? kb : 0].u[e]);
                if (WL_LDS) wl_lds[kb * 64] = wl[0].q;
            }
            if (MASKED) {
                Affine a(aff_p, Ci);
                emu = a.mean[ecol]; esc = a.scale[ecol]; ebe = a.beta[ecol]; eis = a.invstd[ecol];
            }
        } else {
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[DXW ? 0 : j][r] = 0.f;
        }
        int64_t chunk = blockIdx.x;
        constexpr bool STAGES = !(DXW && DXFREE);                   // (this instantiation's waves take part in the staging)
        if (STAGES) fetch(raw0, chunk);
        __syncthreads();                                            // the tables are in place
        if (STAGES) {
            stage(raw0, chunk, 0);
            if (D2) {
                fetch(raw1, chunk + G);
                fetch(raw0, chunk + 2 * (int64_t)G);
                raw_landed(raw1);
            } else {
                fetch(raw0, chunk + G);
            }
            raw_landed(raw0);
        }
        int buf = 0;
        RABS(wave, 1)
        RSTAMP_DECL
        // one chunk: stage the next one (its requests went out two chunks ago -- with one chunk of distance the 32 x 32 and 64 x 32
        // pairs, whose chunk is shorter than a memory latency, waited for them: 63 -> 77 us at 524 288 rows), refill its register
        // set, multiply this one
        auto iter = [&](Raw &raw) {
            __syncthreads();                                        // this chunk is staged in `buf`; every wave is done with buf ^ 1
            RSTAMP(2)
            if (STAGES) stage(raw, chunk + G, buf ^ 1);
            RSTAMP(0)
            if (STAGES) fetch(raw, chunk + (D2 ? 3 : 2) * (int64_t)G);
            __builtin_amdgcn_sched_barrier(0);                      // (hipcc sinks the requests below the MFMAs otherwise: ISA of the first version)
            RSTAMP(1)
            const unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * PANEL;
            if (!DXW) {
                // ---- dW tiles of this wave: contraction over the chunk's 32 rows = two blocks.  Two tiles at a time, their MFMAs
                // alternating: back-to-back MFMAs on ONE accumulator each wait for the one before (8 passes + the pipeline; in-kernel
                // stamps of the first version: 41 - 52 cycles per MFMA issued)
                auto frags = [&](int tile, int pb, SplitFrag (&f)[6]) {
                    const int mb = tile / CI_T, nb = tile - mb * CI_T;
                    f[0] = frag_t(ia, mb, pb); f[1] = frag_t(ia + PANEL, mb, pb); f[2] = frag_t(ia + 2 * PANEL, mb, pb);
                    f[3] = frag_t(ib, nb, pb); f[4] = frag_t(ib + PANEL, nb, pb); f[5] = frag_t(ib + 2 * PANEL, nb, pb);
                };
                // the six products of a block, smallest terms first: (A piece, B piece) = (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {3, 5, 4, 3, 4, 3};
                constexpr int JS = TW == 3 ? 1 : 2;                  // (three tiles per wave -- 128 x 96 -- have no registers for a pair's operands)
#pragma unroll
                for (int j = 0; j < TW; j += JS) {
                    const int e0 = (wave - CI_T) + j * NDW, e1 = e0 + NDW;                  // (uniform)
                    const bool two = JS == 2 && j + 1 < TW && e1 < NTILE;
                    if (e0 < NTILE) {
#pragma unroll
                        for (int pb = 0; pb < BP / 16; ++pb) {
                            SplitFrag f0[6], f1[6];
                            frags(e0, pb, f0);
                            if (two) {
                                frags(e1, pb, f1);
#pragma unroll
                                for (int q = 0; q < 6; ++q) {
                                    accw[DXW ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[PA[q]].v, f0[PB[q]].v, accw[DXW ? 0 : j], 0, 0, 0);
                                    accw[DXW || j + 1 >= TW ? 0 : j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[PA[q]].v, f1[PB[q]].v, accw[DXW || j + 1 >= TW ? 0 : j + 1], 0, 0, 0);
                                }
                            } else {
#pragma unroll
                                for (int q = 0; q < 6; ++q)
                                    accw[DXW ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[PA[q]].v, f0[PB[q]].v, accw[DXW ? 0 : j], 0, 0, 0);
                            }
                        }
                    }
                }
                RSTAMP(3)
            } else {
                // ---- dX tile: rows of the chunk x columns 32 wave .., contraction over C_out (two accumulators, alternating: one
                // chain of 6 C_out / 16 dependent MFMAs issued at 41 - 52 cycles apiece, in-kernel stamps of the first versions)
                f32x16 acc, acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
                for (int kb = 0; kb < KBX; ++kb) {
                    const unsigned o = tr_img_off(l31, 2 * kb + lh);
                    SplitFrag ah, am, al;
                    ah.q = *reinterpret_cast<const uint4 *>(ia + o);
                    am.q = *reinterpret_cast<const uint4 *>(ia + o + PANEL);
                    al.q = *reinterpret_cast<const uint4 *>(ia + o + 2 * PANEL);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, wh[DXW ? kb : 0].v, acc, 0, 0, 0);
                    f32x16 &d = DX2 ? acc2 : acc;
                    SplitFrag wlo;
                    if (WL_LDS) wlo.q = wl_lds[kb * 64]; else wlo = wl[DXW && !WL_LDS ? kb : 0];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wlo.v, d, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wm[DXW ? kb : 0].v, acc, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wh[DXW ? kb : 0].v, d, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wm[DXW ? kb : 0].v, acc, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wh[DXW ? kb : 0].v, d, 0, 0, 0);
                }
#ifdef PN2_STAMP
                { float a0_ = acc[0] + acc2[0]; asm volatile("" : "+v"(a0_)); }           // (the stamp then sees the MFMAs complete, not issued)
#endif
                RSTAMP(3)
                // (no use of the request registers here: the stores below are unconditional in this instantiation, hipcc counts
                // them, and the next chunk's staging waits with vmcnt(16) -- for the requests, not for the stores' acknowledgements)
                const float *yq = Yps + buf * (BP * LDP) + (4 * lh) * LDP + ecol;
                float yv[16];
                if (MASKED) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) yv[r] = yq[((r & 3) + 8 * (r >> 2)) * LDP];      // immediate offsets, one wait
                }
                // one 64-bit base per chunk and a running 32-bit offset (sixteen row pointers carried across the loop were 32 registers)
                float *xb = dX + ((size_t)chunk * BP + 4 * lh) * (unsigned)ldxo;
                unsigned offx = (unsigned)ecol;
                asm volatile("" : "+v"(offx));
                float s0 = 0.f, s1 = 0.f;
                const float *x0q = x0s + buf * (BP * 12) + (4 * lh) * 12;     // FUSE0: rows 4 lh + (r & 3) + 8 (r >> 2) of the chunk
                const unsigned sgn = (unsigned)(chunk & 1) << 31;           // an odd chunk was multiplied negated: its sign back
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float dz = __uint_as_float(__float_as_uint(DX2 ? acc[r] + acc2[r] : acc[r]) ^ sgn);
                    if (MASKED) {
                        const float y = yv[r];
                        dz = bn_act(y, emu, esc, ebe) > 0.f ? dz : 0.f;
                        s0 += dz;
                        s1 = __builtin_fmaf(dz, (y - emu) * eis, s1);
                    }
                    if (FUSE0) {
                        // (the same address in every lane of a half-wave: broadcast reads)
                        const float4 *xr = reinterpret_cast<const float4 *>(x0q + ((r & 3) + 8 * (r >> 2)) * 12);
                        const float4 a0 = xr[0], a1 = xr[1], a2 = xr[2];
                        g0[0] = __builtin_fmaf(dz, a0.x, g0[0]); g0[FUSE0 ? 1 : 0] = __builtin_fmaf(dz, a0.y, g0[FUSE0 ? 1 : 0]);
                        g0[FUSE0 ? 2 : 0] = __builtin_fmaf(dz, a0.z, g0[FUSE0 ? 2 : 0]); g0[FUSE0 ? 3 : 0] = __builtin_fmaf(dz, a0.w, g0[FUSE0 ? 3 : 0]);
                        g0[FUSE0 ? 4 : 0] = __builtin_fmaf(dz, a1.x, g0[FUSE0 ? 4 : 0]); g0[FUSE0 ? 5 : 0] = __builtin_fmaf(dz, a1.y, g0[FUSE0 ? 5 : 0]);
                        g0[FUSE0 ? 6 : 0] = __builtin_fmaf(dz, a1.z, g0[FUSE0 ? 6 : 0]); g0[FUSE0 ? 7 : 0] = __builtin_fmaf(dz, a1.w, g0[FUSE0 ? 7 : 0]);
                        g0[FUSE0 ? 8 : 0] = __builtin_fmaf(dz, a2.x, g0[FUSE0 ? 8 : 0]); g0[FUSE0 ? 9 : 0] = __builtin_fmaf(dz, a2.y, g0[FUSE0 ? 9 : 0]);
                        g0[FUSE0 ? 10 : 0] = __builtin_fmaf(dz, a2.z, g0[FUSE0 ? 10 : 0]); g0[FUSE0 ? 11 : 0] = __builtin_fmaf(dz, a2.w, g0[FUSE0 ? 11 : 0]);
                    } else {
#if defined(PN2_EXP_NOSTORE)
                        if (dz == 1.2345e-33f) xb[offx] = dz;
#elif defined(PN2_EXP_PLAINSTORE)
                        xb[offx] = dz;
#else
                        PN2_STREAM_STORE(dz, xb + offx);
#endif
                    }
                    offx += ((r & 3) == 3 ? 5u : 1u) * (unsigned)ldxo;       // rows (r & 3) + 8 (r >> 2): +1 +1 +1 +5
                }
                if (MASKED) { st0 += (double)s0; st1 += (double)s1; }
                RSTAMP(4)
            }
            chunk += G;
            buf ^= 1;
        };
        while (chunk < chunks) {
            iter(D2 ? raw1 : raw0);
            if (chunk >= chunks) break;
            iter(raw0);
        }
        RSTAMP_FLUSH(wave)
        RABS(wave, 2)
        // ---- flush (as bwd_res_kernel): dW tiles, the dX column's two reductions
        if (!DXW) {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int e = (wave - CI_T) + j * NDW;
                if (e < NTILE) {
                    const int mb = e / CI_T, nb = e - mb * CI_T;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        atomicAdd(dW + (int64_t)(mb * 32 + acc_row(r, lh)) * lddw + nb * 32 + l31, accw[DXW ? 0 : j][r]);
                }
            }
        } else if (MASKED && red_p != nullptr) {
            st0 += __shfl_xor(st0, 32, 64);
            st1 += __shfl_xor(st1, 32, 64);
            if (lh == 0) {
                double *rep = red_p + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * Ci;
                atomicAdd(rep + ecol, st0);
                atomicAdd(rep + Ci + ecol, st1);
            }
        }
        if (DXW && FUSE0) {                                         // into pn2_conv1x1_wgrad_cf's scratch: part[replica][channel][16]
#pragma unroll
            for (int j = 0; j < (DXW && FUSE0 ? 12 : 1); ++j) g0[j] += __shfl_xor(g0[j], 32, 64);
            if (lh == 0) {
                float *pr = part0 + ((size_t)(blockIdx.x % PN2_CF_REPL) * Ci + ecol) * 16;
#pragma unroll
                for (int j = 0; j < (DXW && FUSE0 ? 12 : 1); ++j) atomicAdd(pr + j, g0[j]);
            }
        }
    };
    if (has_dx) run(pn2_true{}); else run(pn2_false{});             // (grid <= chunks: every workgroup has a first chunk)
    RABS(wave, 3)
}

template <int CO_T, int CI_T, bool POOLED, bool MASKED, bool FUSE0 = false>
int launch_split_bwd_res(ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W, int ldw, int64_t tiles, float *dX, int ldxo,
                         double *red_p, float *dW, int lddw, hipStream_t s, const float *X0 = nullptr, int ld0 = 0, float *part0 = nullptr) {
    constexpr int Co = 32 * CO_T, Ci = 32 * CI_T;
    constexpr size_t lds = 2 * 6 * (size_t)(32 * 256) + sizeof(float) * (2 * 32 * (Ci + 4) + 8 * Co + 6 * Ci + (FUSE0 ? 2 * 32 * 12 : 0)) +
                           (CO_T == 4 && CI_T < 4 ? (size_t)CI_T * (Co / 16) * 1024 : 0);      // (the kernel's WL_LDS region)
    static_assert(lds <= 160 * 1024, "LDS");
    if ((reinterpret_cast<uintptr_t>(dy.Y) & 15) != 0 || (reinterpret_cast<uintptr_t>(Yp) & 15) != 0) return PN2_EUNSUPPORTED;
    auto kern = split_bwd_res_kernel<CO_T, CI_T, POOLED, MASKED, FUSE0>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    const int64_t chunks = tiles * (RES_BM / 32), cap = pn2_num_cus();
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)(chunks < cap ? chunks : cap)), dim3(512), lds, s, dy, Yp, ldp, aff_p, W, ldw, chunks, dX, ldxo,
                       red_p, dW, lddw, X0, ld0, part0);
    return pn2_launch_status();
}

// ----------------------------------------------------------------------------------------------- pooled last layer from its INPUT (round 6)
// The backward of the LAST layer of a set-abstraction MLP (conv + BN + ReLU + max over groups of Kp rows) without that layer's
// pre-BN output Y -- which the forward then never writes (pn2_conv1x1_fwd_pool with Y = NULL: statistics and pooling extrema come
// out of the GEMM epilogue).  dZ is sparse there (one row per group and channel, D[p, c] = dZp[g, c] where p is the recorded
// arg-max row), and the dense part of dY = c0 D + q1 (y - mean) + q0 is affine in y = W x + b, hence in the layer's input
// x = relu(bn(Y_prev)), which the pass reads anyway:
//     dX = [D | X] W2 + h      W2 = [diag(c0) W ; M],  M = W^T diag(q1) W  [Ci x Ci],  h = (q1 (b - mean) + q0)^T W     (cf_prep_kernel)
//     dW = c0 o (D^T X) + q1 o (W (X^T X) + (b - mean) sx^T) + q0 sx^T,   sx = sum_p x_p                              (cf_finish_kernel)
// split_bwd_res_kernel's chunk loop with these changes: no Y stream (bytes 4 P (2 C_in) instead of 4 P (C_out + 2 C_in)); the D image
// is a select of the group's pre-split dZp quad (no BatchNorm arithmetic, no split per row); waves 0 .. CI_T - 1 own the dX tiles
// and do NOTHING else (contraction over C_out + C_in: their fragment registers fill the file, the stagers' request registers do not
// fit beside them); the other waves stage and own the tiles of T = X^T [D | X] (C_in x (C_out + C_in); Gram tiles below the
// diagonal are mirrored at the flush), dealt so that the four SIMDs carry the same number of MFMAs.  T and sx leave as one slab per
// workgroup (plain stores, summed in fixed order by cf_finish_kernel: no atomics, no zeroed scratch, run-to-run identical).
// Roles by wave: 0 .. CI_T - 1 own the dX tiles; wave CI_T owns the GRAM part of T (all tiles X_i^T X_j, i <= j: the CI_T blocks of X
// are read once per contraction block and serve as A and as B operands); waves CI_T + 1 .. CI_T + CO_T own one D COLUMN of T each
// (tiles X_i^T D_nb, i < CI_T: the B fragments of D_nb are read once) and, with any wave left over, do all the staging.  (First
// version: tiles dealt one by one to all non-dX waves, each tile reading its own six fragment sets -- the T waves sat on the LDS.
// Ablation of that version, us at 1 M rows of 128 x 96: all 497, no T tiles 327, no staging 360, no dX MFMAs 433, no epilogue 460.)
template <int CO_T, int CI_T>
__global__ __launch_bounds__(512, 1) void split_bwd_cf_kernel(const float *dZp, const int32_t *arg, int ldo, int kshift, const float *W2 /* [Co + Ci][Ci], then h[Ci] */,
                                                              const float *Yp, int ldp, const float *aff_p, int64_t chunks, float *dX, int ldxo, double *red_p,
                                                              float *slab) {
    constexpr int Co = 32 * CO_T, Ci = 32 * CI_T, BP = 32, QD = Co / 4, QP = Ci / 4, LDP = Ci + 4;
    constexpr int PANEL = BP * 256, BUF = 6 * PANEL;                // D hi / mid / lo, X hi / mid / lo (one 128-channel panel per piece)
    constexpr int KBX = Co / 16, KBM = Ci / 16, KBT = KBX + KBM;    // contraction blocks of a dX tile: D W', then X M
    constexpr int NS = 8 - CI_T - 1, NTS = 64 * NS;                 // staging waves (CI_T + 1 .. 7) and their threads
    constexpr int NGT = CI_T * (CI_T + 1) / 2, LDT = Co + Ci + 4;   // Gram tiles; slab: T[Ci][LDT], column Co + Ci = sx
    static_assert(NS >= CO_T, "one staging wave per D column");
    constexpr int RPP = NTS / QD, IT_D = (BP + RPP - 1) / RPP, IT_P = (BP * QP + NTS - 1) / NTS;
    static_assert(NTS % QD == 0, "one channel quad of D per staging thread");
    unsigned char *lds_b = reinterpret_cast<unsigned char *>(res_lds);
    float *Yps = res_lds + (2 * BUF) / 4;                           // [2][BP][LDP]: the raw Y_prev chunk
    float *xtab = Yps + 2 * BP * LDP;                               // [2][3 * Ci]: mean, scale, beta of the previous BatchNorm; then mean, -scale, -beta
                                                                    // (odd chunks are computed negated: split_bwd_res_kernel, SIGN ALTERNATION)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    RABS(wave, 0)
    for (int i = t; i < 3 * Ci; i += 512) { const float v = aff_p[i]; xtab[i] = v; xtab[3 * Ci + i] = i < Ci ? v : -v; }

    const bool has_dx = wave < CI_T, is_gram = wave == CI_T;
    const int ecol = (has_dx ? wave : 0) * 32 + l31;
    const int ts = wave <= CI_T ? 0 : t - 64 * (CI_T + 1);          // staging thread index
    struct Raw { float4 z; int4 a; float4 p[IT_P]; };
    Raw raw;
    auto p_item = [&](int i, int &row, int &q) { const int idx = (NTS * (i + 1) > BP * QP) ? min(ts + NTS * i, BP * QP - 1) : ts + NTS * i; row = idx / QP; q = idx - row * QP; };
    auto fetch = [&](int64_t chunk) {
        const unsigned m0 = (unsigned)(chunk < chunks ? chunk : chunks - 1) * BP;      // past the end: re-read the last chunk (never used)
        const unsigned o = (m0 >> kshift) * (unsigned)ldo + 4u * (unsigned)(ts % QD);   // the chunk lies inside ONE group
        raw.z = ld4(dZp + o);
        raw.a = ld4i(arg + o);
#pragma unroll
        for (int i = 0; i < IT_P; ++i) {
            int row, q;
            p_item(i, row, q);
            raw.p[i] = ld4(Yp + ((m0 + (unsigned)row) * (unsigned)ldp + 4u * (unsigned)q));
        }
    };
    auto store_split = [&](unsigned char *img, int row, int q, const float4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0, l0);
        split2(v.z, v.w, h1, m1, l1);
        const unsigned o = tr_img_off(row, q >> 1) + 8u * (unsigned)(q & 1);
        *reinterpret_cast<uint2 *>(img + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(img + o + PANEL) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(img + o + 2 * PANEL) = make_uint2(l0, l1);
    };
    auto stage = [&](int64_t chunk, int buf) {
        unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * PANEL;
        float *yp = Yps + buf * (BP * LDP);
        const unsigned m0 = (unsigned)(chunk < chunks ? chunk : chunks - 1) * BP;
        const int kbase = (int)(m0 & ((1u << kshift) - 1u));
        const int par = (int)(chunk & 1);                           // (uniform) odd chunks are staged negated
        const float *xt = xtab + par * (3 * Ci);
        const float lim = par ? -INFINITY : INFINITY;
        {   // D: the group's dZp quad, split ONCE; a row takes the pieces of the channels whose maximum sits in it, zeros elsewhere
            unsigned h0, m0_, l0, h1, m1, l1;
            split2(raw.z.x, raw.z.y, h0, m0_, l0);
            split2(raw.z.z, raw.z.w, h1, m1, l1);
            const unsigned sm2 = par ? 0x80008000u : 0u;            // (the pieces of -x are the pieces of x negated)
            h0 ^= sm2; m0_ ^= sm2; l0 ^= sm2; h1 ^= sm2; m1 ^= sm2; l1 ^= sm2;
            const int q = ts % QD, r0 = ts / QD;
            const int4 a = raw.a;
#pragma unroll
            for (int i = 0; i < IT_D; ++i) {
                const int row = (RPP * (i + 1) > BP) ? min(r0 + RPP * i, BP - 1) : r0 + RPP * i;
                const int k = kbase + row;
                const unsigned k0 = (a.x == k ? 0x0000ffffu : 0u) | (a.y == k ? 0xffff0000u : 0u);
                const unsigned k1 = (a.z == k ? 0x0000ffffu : 0u) | (a.w == k ? 0xffff0000u : 0u);
                const unsigned o = tr_img_off(row, q >> 1) + 8u * (unsigned)(q & 1);
                *reinterpret_cast<uint2 *>(ia + o) = make_uint2(h0 & k0, h1 & k1);
                *reinterpret_cast<uint2 *>(ia + o + PANEL) = make_uint2(m0_ & k0, m1 & k1);
                *reinterpret_cast<uint2 *>(ia + o + 2 * PANEL) = make_uint2(l0 & k0, l1 & k1);
            }
        }
#pragma unroll
        for (int i = 0; i < IT_P; ++i) {
            int row, q;
            p_item(i, row, q);
            float4 x = raw.p[i];
            *reinterpret_cast<float4 *>(&yp[row * LDP + 4 * q]) = x;
            const float4 mu = *reinterpret_cast<const float4 *>(&xt[4 * q]), sc = *reinterpret_cast<const float4 *>(&xt[Ci + 4 * q]);
            const float4 be = *reinterpret_cast<const float4 *>(&xt[2 * Ci + 4 * q]);
            x.x = __builtin_amdgcn_fmed3f(bn_act(x.x, mu.x, sc.x, be.x), 0.f, lim); x.y = __builtin_amdgcn_fmed3f(bn_act(x.y, mu.y, sc.y, be.y), 0.f, lim);
            x.z = __builtin_amdgcn_fmed3f(bn_act(x.z, mu.z, sc.z, be.z), 0.f, lim); x.w = __builtin_amdgcn_fmed3f(bn_act(x.w, mu.w, sc.w, be.w), 0.f, lim);
            store_split(ib, row, q, x);
        }
    };
    auto raw_landed = [&]() {
        asm volatile("" : "+v"(raw.z.x), "+v"(raw.z.y), "+v"(raw.z.z), "+v"(raw.z.w));
        asm volatile("" : "+v"(raw.a.x), "+v"(raw.a.y), "+v"(raw.a.z), "+v"(raw.a.w));
#pragma unroll
        for (int i = 0; i < IT_P; ++i) asm volatile("" : "+v"(raw.p[i].x), "+v"(raw.p[i].y), "+v"(raw.p[i].z), "+v"(raw.p[i].w));
    };
    const int g16 = lane >> 4, j16 = lane & 15, tq = j16 >> 2, tp = j16 & 3;
    auto frag_t = [&](const unsigned char *img, int cblk, int pb) {
        const int c0 = (cblk * 32 + 16 * (g16 & 1)) >> 3;
        SplitFrag f;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 16 * pb + 8 * (g16 >> 1) + 4 * r2 + tq;
            const unsigned o = tr_img_off(row, c0 + (tp >> 1)) + 8u * (unsigned)(tp & 1);
            const pn2_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                reinterpret_cast<__attribute__((address_space(3))) pn2_s16x4 *>((__attribute__((address_space(3))) unsigned char *)(img) + o));
            const uint2 u = __builtin_bit_cast(uint2, v);
            f.u[2 * r2] = u.x; f.u[2 * r2 + 1] = u.y;
        }
        return f;
    };

    if (has_dx) {
        // ------------------------------------------------------------------------------------------------ dX waves
        // W2[k][ecol], k = 16 kb + 8 lh + 0 .. 7: hi / mid in registers; lo of the D W' blocks in LDS (8 KiB per wave, written and
        // read by this wave alone), lo of the X M blocks in registers
        SplitFrag wh[KBT], wm[KBT], wl[KBM];
        uint4 *wl_lds = reinterpret_cast<uint4 *>(xtab + 6 * Ci) + (size_t)wave * KBX * 64 + lane;
        {
            float v[KBT][8];
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[kb][e] = W2[(16 * kb + 8 * lh + e) * Ci + ecol];
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                SplitFrag lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(v[kb][2 * e], v[kb][2 * e + 1], wh[kb].u[e], wm[kb].u[e], lo.u[e]);
                if (kb < KBX) wl_lds[kb * 64] = lo.q; else wl[kb < KBX ? 0 : kb - KBX] = lo;
            }
        }
        Affine a(aff_p, Ci);
        const float emu = a.mean[ecol], esc = a.scale[ecol], ebe = a.beta[ecol], eis = a.invstd[ecol];
        const float eh = W2[(Co + Ci) * Ci + ecol];
        double st0 = 0.0, st1 = 0.0, st2 = 0.0;
        int64_t chunk = blockIdx.x;
        int buf = 0;
        __syncthreads();                                            // the tables are in place
        RABS(wave, 1)
        RSTAMP_DECL
        while (chunk < chunks) {
            __syncthreads();                                        // this chunk is staged in `buf`
            RSTAMP(2)
            const unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * PANEL;
            f32x16 acc, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#ifndef PN2_X_CF_NODXM
#pragma unroll
            for (int kb = 0; kb < KBT; ++kb) {
                const unsigned char *img = kb < KBX ? ia : ib;
                const unsigned o = tr_img_off(l31, 2 * (kb < KBX ? kb : kb - KBX) + lh);
                SplitFrag ah, am, al, wlo;
                ah.q = *reinterpret_cast<const uint4 *>(img + o);
                am.q = *reinterpret_cast<const uint4 *>(img + o + PANEL);
                al.q = *reinterpret_cast<const uint4 *>(img + o + 2 * PANEL);
                if (kb < KBX) wlo.q = wl_lds[kb * 64]; else wlo = wl[kb < KBX ? 0 : kb - KBX];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, wh[kb].v, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wlo.v, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wm[kb].v, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wh[kb].v, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wm[kb].v, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wh[kb].v, acc2, 0, 0, 0);
            }
#endif
#ifdef PN2_STAMP
            { float a0_ = acc[0] + acc2[0]; asm volatile("" : "+v"(a0_)); }
#endif
            RSTAMP(3)
#ifdef PN2_X_CF_NOEPI
            if (acc[0] == 1.2345e-33f) dX[0] = acc2[1];
            chunk += G; buf ^= 1;
            continue;
#endif
            const float *yq = Yps + buf * (BP * LDP) + (4 * lh) * LDP + ecol;
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = yq[((r & 3) + 8 * (r >> 2)) * LDP];
            float *xb = dX + ((size_t)chunk * BP + 4 * lh) * (unsigned)ldxo;
            unsigned offx = (unsigned)ecol;
            asm volatile("" : "+v"(offx));
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            const unsigned sgn = (unsigned)(chunk & 1) << 31;       // an odd chunk was multiplied negated: its sign back (before h joins)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = yv[r];
                const float x = bn_act(y, emu, esc, ebe);
                float dz = __uint_as_float(__float_as_uint(acc[r] + acc2[r]) ^ sgn) + eh;
                dz = x > 0.f ? dz : 0.f;
                s0 += dz;
                s1 = __builtin_fmaf(dz, (y - emu) * eis, s1);
                s2 += fmaxf(x, 0.f);
                PN2_STREAM_STORE(dz, xb + offx);
                offx += ((r & 3) == 3 ? 5u : 1u) * (unsigned)ldxo;
            }
            st0 += (double)s0; st1 += (double)s1; st2 += (double)s2;
            RSTAMP(4)
            chunk += G;
            buf ^= 1;
        }
        RSTAMP_FLUSH(wave)
        RABS(wave, 2)
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        st2 += __shfl_xor(st2, 32, 64);
        if (lh == 0) {
            if (red_p != nullptr) {
                double *rep = red_p + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * Ci;
                atomicAdd(rep + ecol, st0);
                atomicAdd(rep + Ci + ecol, st1);
            }
            slab[(size_t)blockIdx.x * (Ci * LDT) + (size_t)ecol * LDT + Co + Ci] = (float)st2;
        }
    } else {
        // the six products of a block, smallest terms first: (A piece, B piece) = (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        auto mm2 = [&](f32x16 &c0, const SplitFrag (&a0)[3], const SplitFrag (&b0)[3], f32x16 &c1, const SplitFrag (&a1)[3], const SplitFrag (&b1)[3]) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {                           // two tiles, alternating: back-to-back MFMAs on one accumulator wait for each other
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[q]].v, b0[PB[q]].v, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA[q]].v, b1[PB[q]].v, c1, 0, 0, 0);
            }
        };
        auto mm1 = [&](f32x16 &c0, const SplitFrag (&a0)[3], const SplitFrag (&b0)[3]) {
#pragma unroll
            for (int q = 0; q < 6; ++q) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[PA[q]].v, b0[PB[q]].v, c0, 0, 0, 0);
        };
        auto frag3 = [&](const unsigned char *img, int blk, int pb, SplitFrag (&f)[3]) {
            f[0] = frag_t(img, blk, pb); f[1] = frag_t(img + PANEL, blk, pb); f[2] = frag_t(img + 2 * PANEL, blk, pb);
        };
        float *sl = slab + (size_t)blockIdx.x * (Ci * LDT);
        if (is_gram) {
            // ------------------------------------------------------------------------------------------------ Gram wave
            f32x16 accg[NGT];                                       // tile (i <= j) at index j (j + 1) / 2 + i
#pragma unroll
            for (int j = 0; j < NGT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[j][r] = 0.f;
            int64_t chunk = blockIdx.x;
            int buf = 0;
            __syncthreads();                                        // the tables are in place
            RABS(wave, 1)
            RSTAMP_DECL
            while (chunk < chunks) {
                __syncthreads();
                RSTAMP(2)
#ifndef PN2_X_CF_NOT
                const unsigned char *ib = lds_b + buf * BUF + 3 * PANEL;
#pragma unroll
                for (int pb = 0; pb < BP / 16; ++pb) {
                    SplitFrag fx[CI_T][3];
#pragma unroll
                    for (int i = 0; i < CI_T; ++i) frag3(ib, i, pb, fx[i]);
                    if constexpr (CI_T == 1) {
                        mm1(accg[0], fx[0], fx[0]);
                    } else if constexpr (CI_T == 2) {
                        mm2(accg[0], fx[0], fx[0], accg[1], fx[0], fx[1]);
                        mm1(accg[2], fx[1], fx[1]);
                    } else {
                        static_assert(CI_T <= 3, "Gram tiles");
                        mm2(accg[0], fx[0], fx[0], accg[1], fx[0], fx[1]);
                        mm2(accg[2], fx[1], fx[1], accg[3], fx[0], fx[2]);
                        mm2(accg[4], fx[1], fx[2], accg[5], fx[2], fx[2]);
                    }
                }
#endif
                RSTAMP(3)
                chunk += G;
                buf ^= 1;
            }
            RSTAMP_FLUSH(wave)
            RABS(wave, 2)
#pragma unroll
            for (int j = 0; j < CI_T; ++j)
#pragma unroll
                for (int i = 0; i <= j; ++i) {
                    const f32x16 &acc = accg[j * (j + 1) / 2 + i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) sl[(i * 32 + acc_row(r, lh)) * LDT + Co + j * 32 + l31] = acc[r];
                    if (i != j) {                                   // above the diagonal: its mirror image too
#pragma unroll
                        for (int r = 0; r < 16; ++r) sl[(j * 32 + l31) * LDT + Co + i * 32 + acc_row(r, lh)] = acc[r];
                    }
                }
        } else {
            // ------------------------------------------------------------------------------------------------ staging waves (one D column of T each)
            f32x16 accw[CI_T];
#pragma unroll
            for (int j = 0; j < CI_T; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
            const int nb = wave - CI_T - 1 < CO_T ? wave - CI_T - 1 : -1;             // (uniform) this wave's column of D, -1: staging only
            int64_t chunk = blockIdx.x;
            fetch(chunk);
            __syncthreads();                                        // the tables are in place
            stage(chunk, 0);
            fetch(chunk + G);
            raw_landed();
            int buf = 0;
            RABS(wave, 1)
            RSTAMP_DECL
            while (chunk < chunks) {
                __syncthreads();                                    // this chunk is staged in `buf`; every wave is done with buf ^ 1
                RSTAMP(2)
#ifndef PN2_X_CF_NOSTAGE
                stage(chunk + G, buf ^ 1);
#endif
                RSTAMP(0)
#ifndef PN2_X_CF_NOFETCH
                fetch(chunk + 2 * (int64_t)G);
#endif
                __builtin_amdgcn_sched_barrier(0);
                RSTAMP(1)
#ifndef PN2_X_CF_NOT
                if (nb >= 0) {
                    const unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * PANEL;
#pragma unroll
                    for (int pb = 0; pb < BP / 16; ++pb) {
                        SplitFrag fb[3], fa0[3], fa1[3];
                        frag3(ia, nb, pb, fb);
                        frag3(ib, 0, pb, fa0);
                        if constexpr (CI_T >= 2) {
                            frag3(ib, 1, pb, fa1);
                            mm2(accw[0], fa0, fb, accw[CI_T >= 2 ? 1 : 0], fa1, fb);
                            if constexpr (CI_T == 3) {
                                frag3(ib, 2, pb, fa0);
                                mm1(accw[CI_T == 3 ? 2 : 0], fa0, fb);
                            }
                        } else {
                            mm1(accw[0], fa0, fb);
                        }
                    }
                }
#endif
                RSTAMP(3)
                chunk += G;
                buf ^= 1;
            }
            RSTAMP_FLUSH(wave)
            RABS(wave, 2)
            if (nb >= 0) {
#pragma unroll
                for (int i = 0; i < CI_T; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sl[(i * 32 + acc_row(r, lh)) * LDT + nb * 32 + l31] = accw[i][r];
            }
        }
    }
    RABS(wave, 3)
}

// W2 = [diag(c0) W ; W^T diag(q1) W] and h = (q1 (b - mean) + q0)^T W from the layer's coefficient block (filled here from the
// reductions when the caller hands them over: every block recomputes it, block 0 writes dgamma / dbeta), in fp64.
__global__ __launch_bounds__(128) void cf_prep_kernel(LazyCoef lc, const float *coef, int ldc, const float *__restrict__ W, int ldw,
                                                      const float *__restrict__ bias, int Co, int Ci, float *__restrict__ W2) {
    __shared__ double fac[256];                                     // (Co <= 256)
    lazy_coef_prologue(lc);
    const float *c0 = coef, *q1 = coef + ldc, *q0 = coef + 2 * ldc, *mean = coef + 3 * ldc;
    const int r = blockIdx.x;
    if (r < Co) {
        for (int i = threadIdx.x; i < Ci; i += 128) W2[r * Ci + i] = c0[r] * W[(int64_t)r * ldw + i];
        return;
    }
    // row Co + j of W2: M[j, i] = sum_c (W[c, j] q1[c]) W[c, i];  the last row: h[i] = sum_c (q1 (b - mean) + q0)[c] W[c, i]
    const int j = r - Co;
    for (int c = threadIdx.x; c < Co; c += 128)
        fac[c] = j == Ci ? ((double)q1[c] * ((double)bias[c] - (double)mean[c]) + (double)q0[c]) : (double)W[(int64_t)c * ldw + j] * (double)q1[c];
    __syncthreads();
    for (int i = threadIdx.x; i < Ci; i += 128) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;            // (Co % 4 == 0: the shapes this path takes)
#pragma unroll 4
        for (int c = 0; c < Co; c += 4) {
            s0 += fac[c] * (double)W[(int64_t)c * ldw + i];
            s1 += fac[c + 1] * (double)W[(int64_t)(c + 1) * ldw + i];
            s2 += fac[c + 2] * (double)W[(int64_t)(c + 2) * ldw + i];
            s3 += fac[c + 3] * (double)W[(int64_t)(c + 3) * ldw + i];
        }
        W2[r * Ci + i] = (float)((s0 + s1) + (s2 + s3));
    }
}

// dW[c, i] += c0 (D^T X)[c, i] + q1 (sum_j W[c, j] Gram[j, i] + (b - mean) sx[i]) + q0 sx[i]; block i sums row i of every workgroup's
// slab in a fixed order (fp64) first.  dW is accumulated into (the gradient bucket's contract), this launch is its only writer.
__global__ __launch_bounds__(256) void cf_finish_kernel(const float *__restrict__ slab, int nslab, int Co, int Ci, const float *coef, int ldc,
                                                        const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                        float *__restrict__ dW, int lddw) {
    extern __shared__ double cf_row[];                              // [4][LDT] partial sums, then the row in cf_row[0 .. LDT)
    const int i = blockIdx.x, LDT = Co + Ci + 4, NQ = LDT / 4;      // (LDT <= 256: one float4 column per thread of a 64-thread group)
    const int q = threadIdx.x & 63, sg = threadIdx.x >> 6;          // column quad, slab group (slabs sg, sg + 4, ...)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (q < NQ) {
        const float *p = slab + (size_t)i * LDT + 4 * q;
        const size_t ss = (size_t)Ci * LDT;
        int w = sg;
        for (; w + 28 < nslab; w += 32) {                           // eight requests in flight per thread, summed in slab order
            float4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ld4(p + (size_t)(w + 4 * e) * ss);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0 += (double)v[e].x; a1 += (double)v[e].y; a2 += (double)v[e].z; a3 += (double)v[e].w; }
        }
        for (; w < nslab; w += 4) {
            const float4 v = ld4(p + (size_t)w * ss);
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
        double *d = cf_row + sg * LDT + 4 * q;
        d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < LDT; n += 256) cf_row[n] = (cf_row[n] + cf_row[LDT + n]) + (cf_row[2 * LDT + n] + cf_row[3 * LDT + n]);
    __syncthreads();
    const float *c0 = coef, *q1 = coef + ldc, *q0 = coef + 2 * ldc, *mean = coef + 3 * ldc;
    const double sx = cf_row[Co + Ci];
    for (int c = threadIdx.x; c < Co; c += 256) {
        double dot = 0.0;
        for (int j = 0; j < Ci; ++j) dot += (double)W[(int64_t)c * ldw + j] * cf_row[Co + j];
        const double g = (double)c0[c] * cf_row[c] + (double)q1[c] * (dot + ((double)bias[c] - (double)mean[c]) * sx) + (double)q0[c] * sx;
        dW[(int64_t)c * lddw + i] += (float)g;
    }
}

template <int CO_T, int CI_T>
int launch_split_bwd_cf(const float *dZp, const int32_t *arg, int ldo, int kshift, const float *coef, LazyCoef lc, const float *W, int ldw,
                        const float *bias, const float *Yp, int ldp, const float *aff_p, int64_t P, float *dX, int ldxo, double *red_p,
                        float *dW, int lddw, float *scratch, hipStream_t s) {
    constexpr int Co = 32 * CO_T, Ci = 32 * CI_T, LDT = Co + Ci + 4;
    constexpr size_t lds = 2 * 6 * (size_t)(32 * 256) + sizeof(float) * (2 * 32 * (Ci + 4) + 6 * Ci) + (size_t)CI_T * (Co / 16) * 1024;
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = split_bwd_cf_kernel<CO_T, CI_T>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    float *W2 = scratch, *slab = scratch + (((Co + Ci + 1) * Ci + 63) & ~63);
    const int64_t chunks = P / 32, cap = pn2_num_cus() < 256 ? pn2_num_cus() : 256;
    const int grid = (int)(chunks < cap ? chunks : cap);
    hipLaunchKernelGGL(cf_prep_kernel, dim3(Co + Ci + 1), dim3(128), 0, s, lc, coef, (Co + 3) & ~3, W, ldw, bias, Co, Ci, W2);
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, dZp, arg, ldo, kshift, W2, Yp, ldp, aff_p, chunks, dX, ldxo, red_p, slab);
    hipLaunchKernelGGL(cf_finish_kernel, dim3(Ci), dim3(256), sizeof(double) * 4 * LDT, s, slab, grid, Co, Ci, coef, (Co + 3) & ~3, W, ldw, bias,
                       dW, lddw);
    return pn2_launch_status();
}

// The (C_out, C_in) pairs the networks of the model zoo actually run on long row counts, per variant -- every instantiation
// is a 2 000-line kernel; anything else goes to the streamed dgrad + wgrad pair (the entry point does that by itself).
//   dense, masked  (hidden layers):        32x32, 64x64, 96x64, 128x128
//   dense, plain   (first layer of FP / head MLPs on [P, 128] inputs): 128x128
//   pooled, masked (last layer of SA MLPs): 64x32 (K = 32), 128x64 (K = 32 and K % 64 == 0), 128x96 (K % 64 == 0)
#define PN2_RES_CASE(CO, CI, POOL, MASKED)                                                                                \
    if (Co == CO && Ci == CI)                                                                                              \
        return launch_bwd_res<CO / 32, CI / 32, POOL, MASKED>(dy, Yp, ldp, aff_p, W, ldw, tiles, dX, ldxo, red_p, dW, lddw, s);

// Every workgroup ends with C_out x C_in atomic adds (x the waves sharing a column): a launch must give a workgroup enough
// tiles to pay for them.  128 x 128 (dense: FP stacks, heads, sa2 of MSG at 65 k .. 131 k rows) only ties the streamed pair
// there (70 vs 71 us, 111 vs 123 us): taken from 262 144 rows on (the dense scans of cfg5).
int dispatch_bwd_res(int Kpool, bool masked, int Co, int Ci, ResDy dy, const float *Yp, int ldp, const float *aff_p, const float *W,
                     int ldw, int64_t tiles, float *dX, int ldxo, double *red_p, float *dW, int lddw, hipStream_t s) {
    if (pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_RES)) {
        // the same pairs on the bf16 pipe (pooled: groups of 32 rows or a multiple -- a 32-row chunk lies inside one group)
#define PN2_SPLIT_RES_CASE(CO, CI, POOLED, MASKED)                                                                        \
        if (Co == CO && Ci == CI) {                                                                                        \
            const int rc = launch_split_bwd_res<CO / 32, CI / 32, POOLED, MASKED>(dy, Yp, ldp, aff_p, W, ldw, tiles, dX, ldxo, red_p, dW, lddw, s); \
            if (rc != PN2_EUNSUPPORTED) return rc;                                                                         \
        }
        // (32 x 32 and 64 x 32 stay on the fp32 kernels: two or three waves of eight have matrix work there, the chunk is one
        // memory latency long either way -- same box, us at 524 288 rows: 63 -> 77 and 79 -> 99; PN2_SPLIT_RES=2 takes them too)
        const bool all = pn2_opt(PN2_OPT_SPLIT_RES) >= 2;
        if (Kpool == 0 && masked) {
            if (all) { PN2_SPLIT_RES_CASE(32, 32, false, true) }
            PN2_SPLIT_RES_CASE(64, 64, false, true) PN2_SPLIT_RES_CASE(96, 64, false, true)
            if (tiles >= pn2_opt(PN2_OPT_SPLIT_RES_MIN_TILES_128)) { PN2_SPLIT_RES_CASE(128, 128, false, true) }
        } else if (Kpool == 0) {
            if (tiles >= pn2_opt(PN2_OPT_SPLIT_RES_MIN_TILES_128)) { PN2_SPLIT_RES_CASE(128, 128, false, false) }
        } else if (masked && Kpool % 32 == 0) {
            if (all) { PN2_SPLIT_RES_CASE(64, 32, true, true) }
            PN2_SPLIT_RES_CASE(128, 64, true, true) PN2_SPLIT_RES_CASE(128, 96, true, true)
        }
#undef PN2_SPLIT_RES_CASE
    }
    if (Kpool == 0 && masked) {
        PN2_RES_CASE(32, 32, 0, true) PN2_RES_CASE(64, 64, 0, true) PN2_RES_CASE(96, 64, 0, true)
        if (tiles >= 4096) { PN2_RES_CASE(128, 128, 0, true) }
    } else if (Kpool == 0) {
        if (tiles >= 4096) { PN2_RES_CASE(128, 128, 0, false) }
    } else if (masked && Kpool % 64 == 0) {
        PN2_RES_CASE(128, 64, 1, true) PN2_RES_CASE(128, 96, 1, true)
    } else if (masked && Kpool == 32) {
        PN2_RES_CASE(64, 32, 2, true) PN2_RES_CASE(128, 64, 2, true)
    }
    return PN2_EUNSUPPORTED;
}

// ----------------------------------------------------------------------------------------------- resident forward
// Every wave owns 32-row slabs: global -> registers (in flight under the previous slab's MFMAs) -> BN + ReLU -> its private
// LDS buffer [32][K+4] -> MFMA against the shared W image [N][K+4] -> bias, store, statistics straight from the
// accumulators.  ACT: X = relu(bn(X_raw)) with the affine block `aff` (hidden layers) or X as stored.
//
// POOL: the layer feeds a max over groups of Kp consecutive rows (the last layer of a set-abstraction MLP).  BatchNorm + ReLU
// is monotone in y per channel -- non-decreasing where the folded scale gamma * invstd is >= 0, non-increasing where it is
// negative, and its sign is the sign of gamma, a PARAMETER known before the statistics are -- so
// max_k relu(bn(y_k)) = relu(bn(max_k y_k)) resp. relu(bn(min_k y_k)) EXACTLY.  The statistics the affine map needs are only
// complete when this launch ends: the epilogue therefore records, per group and channel, the extreme y (the largest, or the
// smallest where gamma < 0: the largest of y with its sign bit flipped) with the first row attaining it, and
// pn2_bn_pool_select applies BN + ReLU to it once the affine block exists.  The pass that re-read all of Y to pool it
// (pn2_bn_relu_max: 0.54 GB at 1 M x 128) disappears.  A wave then owns whole groups: Kp / 32 consecutive slabs (Kp = 16: two
// groups per slab), the running extremum stays in its registers.
struct ResPool {
    float2 *rec;              // [G][ldp] records {extreme y, its row (int bits)}
    const float *gamma;       // BatchNorm weight of this layer: its sign picks maximum or minimum
    int ldp, U;               // U = slabs per group (Kp / 32; 1 for Kp = 16: two groups per slab)
    LazyBn lz;                // consumer-side BatchNorm: the input's affine block is filled by the prologue (with or without pooling)
};

template <int K_T, int N_T, bool ACT, int POOL>      // POOL: 0 none, 1 groups of whole slabs (Kp % 32 == 0), 2 Kp == 16
__global__ __launch_bounds__(512, 2) void fwd_res_kernel(const float *__restrict__ X, int ldx, const float *aff /* written by the lazy prologue */,
                                                         const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                         float *__restrict__ Y, int ldy, int64_t slabs, double *__restrict__ stats,
                                                         ResPool pool) {
    constexpr int K = 32 * K_T, N = 32 * N_T, LDA = K + 4, QK = K / 4, IT = 32 * QK / 64;
    const int NW = blockDim.x >> 6;
    float *Ws = res_lds;                                           // [N][LDA]
    float *atab = Ws + N * LDA;                                    // mean, scale, beta rows of the input BatchNorm: 3 * K
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    float *Ab = atab + 3 * K + wave * (32 * LDA);                  // this wave's staging buffer [32][LDA]
    RABS(wave, 0)

    lazy_bn_prologue(pool.lz);                                     // consumer-side BatchNorm (bn_tail.h)
    for (int i = t; i < N * QK; i += blockDim.x) {
        const int n = i / QK, q = i - n * QK;
        const float *src = W + (int64_t)n * ldw + 4 * q;
        float4 v;
        if ((ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) v = ld4(src);
        else v = make_float4(src[0], src[1], src[2], src[3]);
        *reinterpret_cast<float4 *>(&Ws[n * LDA + 4 * q]) = v;
    }
    if (ACT)
        for (int i = t; i < 3 * K; i += blockDim.x) atab[i] = aff[i];          // affine block rows of pitch K (K % 4 == 0)
    float bj[N_T];
#pragma unroll
    for (int j = 0; j < N_T; ++j) bj[j] = bias[32 * j + l31];

    // whole 32-row slabs only (the host hands a ragged tail to the streamed kernel): a per-slab uniform base plus
    // loop-invariant 32-bit lane offsets, no row predicates
    // (the pooling variants are short of registers at 96+ input channels: they recompute the offsets per use -- a division by a
    // constant and a multiply-add each -- instead of holding 2 * IT of them)
    constexpr bool OFFS_LIVE = POOL == 0;
    unsigned ox[OFFS_LIVE ? IT : 1], ol[OFFS_LIVE ? IT : 1];
    int lane_op = lane;           // made opaque once per slab where the offsets are recomputed: loop-invariant code motion would
                                  // put all 2 * IT of them straight back into registers
    auto off_x = [&](int i) { const int idx = lane_op + 64 * i, row = idx / QK, q = idx - row * QK; return (unsigned)row * (unsigned)ldx + 4u * q; };
    auto off_l = [&](int i) { const int idx = lane_op + 64 * i, row = idx / QK, q = idx - row * QK; return (unsigned)(row * LDA + 4 * q); };
    if (OFFS_LIVE) {
#pragma unroll
        for (int i = 0; i < IT; ++i) { ox[OFFS_LIVE ? i : 0] = off_x(i); ol[OFFS_LIVE ? i : 0] = off_l(i); }
    }
    const int64_t stride = (int64_t)gridDim.x * NW;
    float4 rx[IT];
    auto fetch = [&](int64_t slab) {
        const float *xb = X + (slab < slabs ? slab : slabs - 1) * (int64_t)(32 * ldx);
#pragma unroll
        for (int i = 0; i < IT; ++i) rx[i] = ld4(xb + (OFFS_LIVE ? ox[OFFS_LIVE ? i : 0] : off_x(i)));
    };
    // a wave's work items are units of U consecutive slabs (U = 1 without pooling: plain interleaving)
    const int U = POOL ? pool.U : 1;
    int64_t unit = (int64_t)blockIdx.x * NW + wave;
    int sub = 0;
    int64_t slab = unit * U;
    fetch(slab);
    double st[N_T][2];
#pragma unroll
    for (int j = 0; j < N_T; ++j) st[j][0] = st[j][1] = 0.0;
    float vmx[POOL ? N_T : 1];                                     // POOL: the group's running extremum (sign-flipped where gamma < 0)
    int kmx[POOL ? N_T : 1], sgn[POOL ? N_T : 1];                  // its row; the per-column sign mask
    if (POOL) {
#pragma unroll
        for (int j = 0; j < N_T; ++j) sgn[POOL ? j : 0] = pool.gamma[32 * j + l31] < 0.f ? (int)0x80000000 : 0;
    }
    __syncthreads();                                               // W image and table complete (the only barrier before the end)
    // 64 % QK == 0 (K = 32, 64, 128): a lane's items all sit in ONE channel quad; its BatchNorm constants are read from the
    // table once instead of with every item (3 of the 4 ds_*_b128 of the transform)
    constexpr bool AT_HOIST = ACT && 64 % QK == 0 && !(K_T == 4 && N_T == 4);
    float4 hmu, hsc, hbe;
    if (AT_HOIST) {
        const int q4 = 4 * (lane % QK);
        hmu = *reinterpret_cast<const float4 *>(&atab[q4]);
        hsc = *reinterpret_cast<const float4 *>(&atab[K + q4]);
        hbe = *reinterpret_cast<const float4 *>(&atab[2 * K + q4]);
    }

    RSTAMP_DECL
    RABS(wave, 1)
    while (slab < slabs) {
        RSTAMP(4)
        if (!OFFS_LIVE) asm volatile("" : "+v"(lane_op));
        int64_t next_unit = unit;
        int next_sub = sub + 1;
        if (next_sub == U) { next_sub = 0; next_unit += stride; }
        const int64_t next_slab = next_unit * U + next_sub;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            float4 x = rx[i];
            const unsigned oli = OFFS_LIVE ? ol[OFFS_LIVE ? i : 0] : off_l(i);
            if (ACT) {
                const int q4 = AT_HOIST ? 0 : (int)(oli % (unsigned)LDA);
                const float4 mu = AT_HOIST ? hmu : *reinterpret_cast<const float4 *>(&atab[q4]);
                const float4 sc = AT_HOIST ? hsc : *reinterpret_cast<const float4 *>(&atab[K + q4]);
                const float4 be = AT_HOIST ? hbe : *reinterpret_cast<const float4 *>(&atab[2 * K + q4]);
                x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
                x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
                x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
                x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
            }
            *reinterpret_cast<float4 *>(&Ab[oli]) = x;
        }
        RSTAMP(0)
        fetch(next_slab);
        RSTAMP(1)

        f32x16 acc[N_T];
#pragma unroll
        for (int j = 0; j < N_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float *ap = &Ab[l31 * LDA + 4 * lh];
        const float *bp = &Ws[l31 * LDA + 4 * lh];
#pragma unroll
        for (int kb = 0; kb < K / 8; ++kb) {
            const float4 a = *reinterpret_cast<const float4 *>(ap + 8 * kb);
#pragma unroll
            for (int j = 0; j < N_T; ++j) {
                const float4 b = *reinterpret_cast<const float4 *>(bp + j * 32 * LDA + 8 * kb);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
            }
        }
        RSTAMP(2)
        // ---- epilogue, one 32-column block at a time: bias + statistics straight from the accumulators (column on the
        // lane), then through this wave's staging buffer (free: every MFMA of the slab has read it) so that Y leaves as
        // 16-byte stores, 128 contiguous bytes per row -- 16 store instructions per slab instead of 64, straight-line:
        // the next slab's operand wait stays a COUNTED vmcnt (gfx9 retires loads and stores through one in-order
        // counter; a wait the compiler cannot count becomes vmcnt(0) and stalls on the store acknowledgements)
        float *yb = Y + slab * (int64_t)(32 * ldy);
        const unsigned yo = (unsigned)(lane >> 3) * (unsigned)ldy + (unsigned)(lane & 7) * 4u;
#pragma unroll
        for (int j = 0; j < N_T; ++j) {
            float s0 = 0.f, s1 = 0.f;
            // this slab's extremum over the lane's rows: (value, accumulator register), ascending register = ascending row,
            // strict comparisons keep the first; registers 0..7 are rows 0..15, 8..15 rows 16..31 (two groups when Kp = 16)
            float ma = -INFINITY, mb = -INFINITY;
            int mai = 0, mbi = 8;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = acc[j][r] + bj[j];
                Ab[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + l31] = y;
                s0 += y;
                s1 = __builtin_fmaf(y, y, s1);
                if (POOL) {
                    const float yp = __int_as_float(__float_as_int(y) ^ sgn[POOL ? j : 0]);
                    if (POOL == 1 || r < 8) { mai = yp > ma ? r : mai; ma = fmaxf(ma, yp); }
                    else { mbi = yp > mb ? r : mbi; mb = fmaxf(mb, yp); }
                }
            }
            st[j][0] += (double)s0;
            st[j][1] += (double)s1;
            if (POOL) {
                auto rowof = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * lh; };          // row of the slab
                auto meet = [&](float &v, int &k) {                                            // the two half-waves hold disjoint rows
                    const float ov = __shfl_xor(v, 32, 64);
                    const int ok = __shfl_xor(k, 32, 64);
                    const bool take = ov > v || (ov == v && ok < k);
                    v = take ? ov : v; k = take ? ok : k;
                };
                // The records leave through UNCONDITIONAL 8-byte stores, the same number every slab, from a wave-uniform base
                // plus a lane offset (a slab that does not finish a group aims at this wave's dump line; the second half-wave
                // holds a copy after the meeting and writes it too): stores under a branch would make the next slab's operand
                // wait uncountable, i.e. vmcnt(0) behind every store acknowledgement.
                float2 *dump = reinterpret_cast<float2 *>(pn2_dump_lines + ((blockIdx.x & 1023) * 8 + (wave & 7)) * 64);
                const int sg = sgn[POOL ? j : 0];
                if (POOL == 2) {                                    // Kp = 16: groups 2 * slab and 2 * slab + 1 are complete
                    int ka = rowof(mai), kb = rowof(mbi) - 16;
                    meet(ma, ka); meet(mb, kb);
                    float2 *base = pool.rec + (2 * slab) * (int64_t)pool.ldp;
                    base[32 * j + l31] = make_float2(__int_as_float(__float_as_int(ma) ^ sg), __int_as_float(ka));
                    base[pool.ldp + 32 * j + l31] = make_float2(__int_as_float(__float_as_int(mb) ^ sg), __int_as_float(kb));
                } else {
                    // into the group's running extremum (earlier slabs first: ties keep them)
                    const int mk = 32 * sub + rowof(mai);
                    const bool t1 = sub == 0 || ma > vmx[j];
                    vmx[j] = t1 ? ma : vmx[j]; kmx[j] = t1 ? mk : kmx[j];
                    float v1 = vmx[j];
                    int k1 = kmx[j];
                    const bool last = sub == U - 1;                 // wave-uniform
                    if (last) meet(v1, k1);
                    float2 *base = last ? pool.rec + unit * (int64_t)pool.ldp : dump;
                    base[last ? 32 * j + l31 : l31] = make_float2(__int_as_float(__float_as_int(v1) ^ sg), __int_as_float(k1));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = *reinterpret_cast<const float4 *>(&Ab[((lane >> 3) + 8 * i) * 36 + (lane & 7) * 4]);
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f vv = {v.x, v.y, v.z, v.w};
                PN2_STREAM_STORE(vv, reinterpret_cast<v4f *>(yb + (yo + (unsigned)(8 * i) * (unsigned)ldy + 32u * j)));
            }
            if (POOL) __builtin_amdgcn_sched_barrier(0);           // one column block at a time: the scans' temporaries do not pile up
        }
        unit = next_unit; sub = next_sub; slab = next_slab;
    }

    RSTAMP(3)
    RSTAMP_FLUSH(wave)
    RABS(wave, 2)
    if (stats != nullptr) {                                        // fold the waves in LDS (the W image is dead), one atomic per channel
        __syncthreads();
        double *red = reinterpret_cast<double *>(res_lds);        // [NW][N][2]
#pragma unroll
        for (int j = 0; j < N_T; ++j) {
            double a0 = st[j][0], a1 = st[j][1];
            a0 += __shfl_xor(a0, 32, 64);
            a1 += __shfl_xor(a1, 32, 64);
            if (lh == 0) {
                red[(wave * N + 32 * j + l31) * 2] = a0;
                red[(wave * N + 32 * j + l31) * 2 + 1] = a1;
            }
        }
        __syncthreads();
        if (t < N) {
            double a0 = 0.0, a1 = 0.0;
            for (int w = 0; w < NW; ++w) { a0 += red[(w * N + t) * 2]; a1 += red[(w * N + t) * 2 + 1]; }
            double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
            atomicAdd(rep + t, a0);
            atomicAdd(rep + N + t, a1);
        }
    }
    RABS(wave, 3)
}

template <int K_T, int N_T, bool ACT, int POOL>
int launch_fwd_res(const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y, int ldy,
                   int64_t P, double *stats, const ResPool &pool, hipStream_t s) {
    constexpr int K = 32 * K_T, N = 32 * N_T, LDA = K + 4;
    constexpr size_t fixed = sizeof(float) * ((size_t)N * LDA + 3 * K), per_wave = sizeof(float) * 32 * LDA;
    constexpr int nw_fit = (int)((160 * 1024 - fixed) / per_wave);
    // fewer than two waves per SIMD cannot fill each other's phases: W plus eight staging buffers must fit the LDS.  Decided at
    // compile time, so the shapes that do not fit (128 -> 64 / 96 / 128) are not instantiated at all (round 4 shipped them as
    // dead code, one of them with 140 bytes of scratch that tools/check_isa.py had to allow-list)
    if constexpr (nw_fit < 8) return PN2_EUNSUPPORTED;
    else {
    constexpr int nw = 8;
    size_t lds = fixed + nw * per_wave;
    const size_t red = sizeof(double) * 2 * N * nw;                 // the final fold reuses the image
    if (lds < red) lds = red;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(&fwd_res_kernel<K_T, N_T, ACT, POOL>), raised) != PN2_OK) return PN2_ELAUNCH;
    const int64_t slabs = P / 32;                                  // whole slabs; the caller handles P % 32
    int64_t grid = pn2_cdiv(slabs / (POOL ? pool.U : 1), nw);
    if (grid > pn2_num_cus()) grid = pn2_num_cus();
    PN2_NOTE_KERNEL(fwd_res_kernel<K_T, N_T, ACT, POOL>);
    hipLaunchKernelGGL((fwd_res_kernel<K_T, N_T, ACT, POOL>), dim3((unsigned)grid), dim3(64 * nw), lds, s, X, ldx, aff, W, ldw, bias, Y, ldy,
                       slabs, stats, pool);
    return pn2_launch_status();
    }
}

template <int K_T, bool ACT, int POOL>
int dispatch_fwd_res_n(int N, const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y, int ldy,
                       int64_t P, double *stats, const ResPool &pool, hipStream_t s) {
    switch (N / 32) {
        case 1: return launch_fwd_res<K_T, 1, ACT, POOL>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 2: return launch_fwd_res<K_T, 2, ACT, POOL>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 3: return launch_fwd_res<K_T, 3, ACT, POOL>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 4: return launch_fwd_res<K_T, 4, ACT, POOL>(X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
    }
    return PN2_EINVAL;
}

template <bool ACT, int POOL>
int dispatch_fwd_res(int K, int N, const float *X, int ldx, const float *aff, const float *W, int ldw, const float *bias, float *Y,
                     int ldy, int64_t P, double *stats, const ResPool &pool, hipStream_t s) {
    switch (K / 32) {
        case 1: return dispatch_fwd_res_n<1, ACT, POOL>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 2: return dispatch_fwd_res_n<2, ACT, POOL>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 3: return dispatch_fwd_res_n<3, ACT, POOL>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
        case 4: return dispatch_fwd_res_n<4, ACT, POOL>(N, X, ldx, aff, W, ldw, bias, Y, ldy, P, stats, pool, s);
    }
    return PN2_EINVAL;
}

inline int res_min_rows() {
    const int v = pn2_opt(PN2_OPT_RES_MIN_ROWS);
    return v;
}

}  // namespace

// Shapes the weight-resident kernels take: both channel counts multiples of 32 and <= 128, enough rows to give every
// CU a few tiles.  PN2_RES=0 turns the family off (A/B runs against the streamed kernels).
static bool res_shape_ok(int C_out, int C_in) {
    return C_out % 32 == 0 && C_in % 32 == 0 && C_out >= 32 && C_out <= 128 && C_in >= 32 && C_in <= 128;
}

#ifdef PN2_STAMP
extern "C" int pn2_debug_stamps_abs(unsigned long long *host_out, int n) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pn2_res_abs_buf), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -2;
}
extern "C" int pn2_debug_stamps_res(unsigned long long *host_out, int n) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pn2_res_stamp_buf), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int pn2_res_supported(int64_t P, int C_out, int C_in) {
    const int on = pn2_opt(PN2_OPT_RES);
    return on && P >= res_min_rows() && P < (1LL << 31) && res_shape_ok(C_out, C_in);
}

// Whether pn2_conv1x1_bwd runs this layer in the fused kernel (the instantiation list of dispatch_bwd_res) rather than handing
// it to the streamed dgrad + wgrad pair.  Kpool = 0: dense dZ; masked: the layer has a BatchNorm + ReLU input (prev_affine).
extern "C" int pn2_bwd_res_supported(int64_t P, int C_out, int C_in, int Kpool, int masked) {
    if (!pn2_res_supported(P, C_out, C_in)) return 0;
    if (P * (int64_t)std::max(C_out, C_in) >= (1LL << 32)) return 0;     // 32-bit element offsets inside the fused kernel
    const int64_t tiles = P / RES_BM;
    const auto is = [&](int co, int ci) { return C_out == co && C_in == ci; };
    const int64_t min128 = (pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_RES)) ? std::min<int64_t>(4096, pn2_opt(PN2_OPT_SPLIT_RES_MIN_TILES_128)) : 4096;
    if (Kpool == 0 && masked) return is(32, 32) || is(64, 64) || is(96, 64) || (tiles >= min128 && is(128, 128));
    if (Kpool == 0) return tiles >= min128 && is(128, 128);
    if (masked && Kpool % 64 == 0) return is(128, 64) || is(128, 96);
    if (masked && Kpool == 32) return is(64, 32) || is(128, 64);
    return 0;
}

// Called by pn2_conv1x1_fwd (mlp.hip) for supported shapes when no fused BatchNorm tail is requested; P % 32 == 0.
int pn2_fwd_res(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                int64_t P, int K, int N, double *stats, LazyBn lz, hipStream_t s) {
    ResPool none{};
    none.lz = lz;
    if (in_affine) return dispatch_fwd_res<true, 0>(K, N, X, ldx, in_affine, W, ldw, bias, Y, ldy, P, stats, none, s);
    return dispatch_fwd_res<false, 0>(K, N, X, ldx, nullptr, W, ldw, bias, Y, ldy, P, stats, none, s);
}

namespace {
// out[g,c] = relu(bn(v)) with v the recorded maximum where the folded scale is >= 0 and the minimum where it is negative; arg
// = the row that attained it.  Pad columns (c >= C) get the zero pad of the affine block like pn2_bn_relu_max writes them.
__global__ __launch_bounds__(256) void bn_pool_select_kernel(const float2 *__restrict__ rec, int ldp, const float *aff /* written by the prologue */, int lda,
                                                             int64_t G, int C, float *__restrict__ out, int ldo, int32_t *__restrict__ arg,
                                                             LazyBn lz) {
    lazy_bn_prologue(lz);                                          // consumer-side BatchNorm (bounded grid: paid once per workgroup)
    const int qpr = C >> 2;
    const int64_t n = G * qpr;
    Affine a(aff, lda);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / qpr;
        const int c = (int)(i - g * qpr) * 4;
        const float4 mu = ld4(a.mean + c), sc = ld4(a.scale + c), be = ld4(a.beta + c);
        const float4 r01 = ld4(reinterpret_cast<const float *>(rec + g * ldp + c)), r23 = ld4(reinterpret_cast<const float *>(rec + g * ldp + c + 2));
        float4 o;
        o.x = fmaxf(bn_act(r01.x, mu.x, sc.x, be.x), 0.f);
        o.y = fmaxf(bn_act(r01.z, mu.y, sc.y, be.y), 0.f);
        o.z = fmaxf(bn_act(r23.x, mu.z, sc.z, be.z), 0.f);
        o.w = fmaxf(bn_act(r23.z, mu.w, sc.w, be.w), 0.f);
        *reinterpret_cast<float4 *>(out + g * ldo + c) = o;
        // A channel whose BatchNorm weight is exactly 0 has scale 0: every row of the group gives relu(beta), and torch.max
        // (model/pointnet_util.py:199, :256) routes the gradient of an all-equal group to its FIRST row -- not to the row with the
        // largest pre-BN value the epilogue recorded (found by the signed-gamma test of round 5: d gamma of those channels).
        *reinterpret_cast<int4 *>(arg + g * ldo + c) = make_int4(sc.x == 0.f ? 0 : __float_as_int(r01.y), sc.y == 0.f ? 0 : __float_as_int(r01.w),
                                                                 sc.z == 0.f ? 0 : __float_as_int(r23.y),
                                                                 sc.w == 0.f ? 0 : __float_as_int(r23.w));
    }
}
}  // namespace

extern "C" int pn2_conv1x1_fwd_pool(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y,
                                    int ldy, int64_t P, int K, int N, double *stats, int Kpool, const float *gamma, float *pool_ws,
                                    const pn2_bn_lazy *in_lazy, pn2_stream_t stream) {
    // Y == NULL: the pre-BN output is not written at all (statistics and extrema only) -- taken by the bf16-pipe forms of the
    // register-stationary forward alone (PN2_EUNSUPPORTED otherwise); the layer's backward then runs on its input (pn2_conv1x1_bwd_cf)
    PN2_CHECK_ARG(X && in_affine && W && bias && stats && gamma && pool_ws && P > 0 && P < (1LL << 31) && K > 0 && N > 0 && Kpool > 0);
    PN2_CHECK_ARG(lazy_bn_ok(in_lazy, in_affine, K));
    PN2_CHECK_ARG(ldx % 4 == 0 && ldx >= K && ldw >= K && ldy % 4 == 0 && ldy >= N);
    PN2_CHECK_ARG((reinterpret_cast<uintptr_t>(pool_ws) & 15) == 0);
    if (P % Kpool == 0 && ldy == N && N % 32 == 0) {                 // wide last layers (128 / 196 -> 256): the register-stationary forward
        int64_t done = 0;
        const int rc = pn2_wide_fwd(X, ldx, in_affine, W, ldw, bias, Y, ldy, P, K, N, stats, make_lazy_bn(in_lazy), pn2_s(stream), &done,
                                    Kpool, gamma, pool_ws);
        if (rc != PN2_EUNSUPPORTED) return rc;
    }
    if (Y == nullptr) return PN2_EUNSUPPORTED;
    if (!pn2_res_supported(P, N, K) || P % 32 != 0 || P % Kpool != 0 || !(Kpool == 16 || Kpool % 32 == 0)) return PN2_EUNSUPPORTED;
    ResPool pool;
    pool.rec = reinterpret_cast<float2 *>(pool_ws);
    pool.gamma = gamma;
    pool.ldp = N; pool.U = Kpool == 16 ? 1 : Kpool / 32;
    pool.lz = make_lazy_bn(in_lazy);
    if (Kpool == 16) return dispatch_fwd_res<true, 2>(K, N, X, ldx, in_affine, W, ldw, bias, Y, ldy, P, stats, pool, pn2_s(stream));
    return dispatch_fwd_res<true, 1>(K, N, X, ldx, in_affine, W, ldw, bias, Y, ldy, P, stats, pool, pn2_s(stream));
}

extern "C" int pn2_bn_pool_select(const float *pool_ws, const float *affine, int64_t G, int C, float *out, int ldo, int32_t *arg,
                                  const pn2_bn_lazy *lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(lazy_bn_ok(lazy, affine, C));
    // ldo: pitch of out AND of arg (out may be a column slice of a wider matrix: the concatenated MSG output)
    PN2_CHECK_ARG(pool_ws && affine && out && arg && G > 0 && C > 0 && C % 32 == 0 && ldo >= C && ldo % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(pool_ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int64_t n = G * (C >> 2);
    int64_t blocks = pn2_cdiv(n, 256);
    const int64_t cap = (int64_t)pn2_num_cus() * 8;                 // bounded grid (see the kernel)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(bn_pool_select_kernel, dim3((unsigned)blocks), dim3(256), 0, pn2_s(stream),
                       reinterpret_cast<const float2 *>(pool_ws), C, affine, C, G, C, out, ldo, arg, make_lazy_bn(lazy));
    return pn2_launch_status();
}

extern "C" int pn2_conv1x1_bwd(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y,
                               int ldy, const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev,
                               const float *prev_affine, float *dXout, int ldxo, double *prev_red, float *dW, int lddw,
                               int64_t P, int C_out, int C_in, const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && coef && W && prev_Y && dXout && dW && P > 0 && P < (1LL << 31) && res_shape_ok(C_out, C_in));
    PN2_CHECK_ARG(lazy_coef_ok(coef_lazy, coef, C_out));
    PN2_CHECK_ARG(dZ != nullptr || (dZp && arg && Kpool > 0));
    PN2_CHECK_ARG(ldw >= C_in && lddw >= C_in && ldy % 4 == 0 && ldy >= C_out && ld_prev % 4 == 0 && ld_prev >= C_in && ldxo >= C_in);
    PN2_CHECK_ARG(prev_affine != nullptr || prev_red == nullptr);
    hipStream_t s = pn2_s(stream);
    const int kshift = dZ ? 0 : pow2_shift(Kpool);
    PN2_CHECK_ARG(dZ ? (ldz == ldy) : (kshift >= 0 && ldo % 4 == 0 && ldo >= C_out && prev_affine != nullptr && P % Kpool == 0));
    int64_t tiles = P / RES_BM, P_full = tiles * RES_BM;
    int rc = PN2_OK;
    // the fused kernel addresses with 32-bit element offsets (tile base + lane term): past 2^32 elements of any of its
    // matrices (16 GiB: the K = 128 branch of the dense scans at B = 16) the streamed pair below takes all rows
    const int64_t ld_max = std::max(std::max((int64_t)ldy, (int64_t)ld_prev), std::max((int64_t)ldxo, dZ ? (int64_t)ldz : (int64_t)ldo));
    if (P * ld_max >= (1LL << 32)) { tiles = 0; P_full = 0; }
    if (tiles > 0) {
        ResDy dy{dZ, dZp, arg, ldo, kshift, Y, ldy, coef, make_lazy_coef(coef_lazy)};
        rc = dispatch_bwd_res(dZ ? 0 : Kpool, prev_affine != nullptr, C_out, C_in, dy, prev_Y, ld_prev, prev_affine, W, ldw, tiles, dXout, ldxo,
                              prev_red, dW, lddw, s);
        if (rc == PN2_EUNSUPPORTED) { rc = PN2_OK; tiles = 0; P_full = 0; }    // no weight-resident kernel for this pair: all rows below
        else coef_lazy = nullptr;                                              // the block is filled: the tail launches read it
    }
    if (rc != PN2_OK || P_full == P) return rc;
    // ragged tail (< 64 rows), or a pair without a resident kernel: the streamed kernels, on offset pointers; everything
    // they produce is accumulated
    const int64_t tail = P - P_full;
    const int64_t g_off = dZ ? 0 : (P_full / Kpool) * ldo;                     // P_full is a multiple of Kpool here (Kpool | 64)
    const float *dZt = dZ ? dZ + P_full * ldz : nullptr;
    const float *dZpt = dZ ? nullptr : dZp + g_off;
    const int32_t *argt = dZ ? nullptr : arg + g_off;
    rc = pn2_conv1x1_dgrad(dZt, ldz, dZpt, ldo, argt, Kpool, Y + P_full * ldy, ldy, coef, W, ldw, prev_affine ? prev_Y + P_full * ld_prev : nullptr,
                           ld_prev, prev_affine, dXout + P_full * ldxo, ldxo, prev_red, tail, C_out, C_in, nullptr, coef_lazy, stream);
    if (rc != PN2_OK) return rc;
    return pn2_conv1x1_wgrad(dZt, ldz, dZpt, ldo, argt, Kpool, Y + P_full * ldy, ldy, coef, prev_Y + P_full * ld_prev, ld_prev, prev_affine,
                             dW, lddw, nullptr, tail, C_out, C_in, nullptr, stream);
}

// ----------------------------------------------------------------------------------------------- pooled last layer from its input: entry points
static bool cf_shape(int C_out, int C_in) { return C_out == 128 && (C_in == 96 || C_in == 64); }

extern "C" int pn2_conv1x1_bwd_cf_supported(int64_t P, int C_out, int C_in, int Kpool) {
    if (!(pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_RES) && pn2_opt(PN2_OPT_POOL_CF))) return 0;
    if (!pn2_res_supported(P, C_out, C_in) || !cf_shape(C_out, C_in)) return 0;
    // 128 x 64 (POOL_CF >= 2, the default): alone the forward gains little from the missing store and this backward ties the Y-reading
    // one (159 vs 158 us at 524 288 rows, + 18 us of prep / finish); in the step the forward runs 125 -> 99 us and cfg5 MSG gains 0.3 ms
    if (C_in == 64 && pn2_opt(PN2_OPT_POOL_CF) < 2) return 0;
    if (Kpool < 32 || (Kpool & (Kpool - 1)) != 0 || P % Kpool != 0) return 0;
    if (P * (int64_t)std::max(C_out, C_in) >= (1LL << 32)) return 0;           // 32-bit element offsets inside the kernel
    return 1;
}

extern "C" int64_t pn2_conv1x1_bwd_cf_scratch_bytes(int C_out, int C_in) {
    if (!cf_shape(C_out, C_in)) return 0;
    const int64_t w2 = (((int64_t)(C_out + C_in + 1) * C_in + 63) & ~63LL);
    return 4 * (w2 + 256LL * C_in * (C_out + C_in + 4));
}

extern "C" int pn2_conv1x1_bwd_cf(const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *coef, const float *W, int ldw,
                                  const float *bias, const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                                  double *prev_red, float *dW, int lddw, int64_t P, int C_out, int C_in, const pn2_bn_coef_lazy *coef_lazy,
                                  float *scratch, pn2_stream_t stream) {
    PN2_CHECK_ARG(dZp && arg && coef && W && bias && prev_Y && prev_affine && dXout && dW && scratch && P > 0 && P < (1LL << 31));
    PN2_CHECK_ARG(lazy_coef_ok(coef_lazy, coef, C_out));
    PN2_CHECK_ARG(ldw >= C_in && lddw >= C_in && ld_prev % 4 == 0 && ld_prev >= C_in && ldxo >= C_in && ldo % 4 == 0 && ldo >= C_out);
    PN2_CHECK_ARG((reinterpret_cast<uintptr_t>(scratch) & 255) == 0 && (reinterpret_cast<uintptr_t>(prev_Y) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(dZp) & 15) == 0 && (reinterpret_cast<uintptr_t>(arg) & 15) == 0);
    if (!pn2_conv1x1_bwd_cf_supported(P, C_out, C_in, Kpool)) return PN2_EUNSUPPORTED;
    const int64_t ld_max = std::max(std::max((int64_t)ld_prev, (int64_t)ldxo), (int64_t)ldo);
    if (P * ld_max >= (1LL << 32)) return PN2_EUNSUPPORTED;
    const int kshift = pow2_shift(Kpool);
    const LazyCoef lc = make_lazy_coef(coef_lazy);
    hipStream_t s = pn2_s(stream);
    if (C_in == 96) return launch_split_bwd_cf<4, 3>(dZp, arg, ldo, kshift, coef, lc, W, ldw, bias, prev_Y, ld_prev, prev_affine, P, dXout, ldxo, prev_red, dW, lddw, scratch, s);
    return launch_split_bwd_cf<4, 2>(dZp, arg, ldo, kshift, coef, lc, W, ldw, bias, prev_Y, ld_prev, prev_affine, P, dXout, ldxo, prev_red, dW, lddw, scratch, s);
}

// ----------------------------------------------------------------------------------------------- fused backward with the first layer's dZ^T x
extern "C" int pn2_conv1x1_bwd_first_supported(int64_t P, int C_out, int C_in, int N0) {
    if (!(pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_RES) && pn2_opt(PN2_OPT_FUSE_FIRST))) return 0;
    if (!pn2_res_supported(P, C_out, C_in) || P % 64 != 0 || N0 < 1 || N0 > 12) return 0;
    if (P * (int64_t)std::max(C_out, C_in) >= (1LL << 32)) return 0;
    return (C_out == 96 && C_in == 64) || (C_out == 64 && C_in == 64);
}

extern "C" int pn2_conv1x1_bwd_first(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *W, int ldw,
                                     const float *prev_Y, int ld_prev, const float *prev_affine, double *prev_red, float *dW, int lddw,
                                     const float *X0, int ld0, int N0, void *cf_scratch, int64_t P, int C_out, int C_in,
                                     const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(dZ && Y && coef && W && prev_Y && prev_affine && prev_red && dW && X0 && cf_scratch && P > 0 && P < (1LL << 31));
    PN2_CHECK_ARG(lazy_coef_ok(coef_lazy, coef, C_out));
    PN2_CHECK_ARG(ldz == ldy && ldw >= C_in && lddw >= C_in && ldy % 4 == 0 && ldy >= C_out && ld_prev % 4 == 0 && ld_prev >= C_in);
    PN2_CHECK_ARG(ld0 % 4 == 0 && ld0 >= 12 && (reinterpret_cast<uintptr_t>(X0) & 15) == 0 && (reinterpret_cast<uintptr_t>(cf_scratch) & 15) == 0);
    if (!pn2_conv1x1_bwd_first_supported(P, C_out, C_in, N0)) return PN2_EUNSUPPORTED;
    if (P * (int64_t)std::max((int64_t)ldy, std::max((int64_t)ld_prev, (int64_t)ld0)) >= (1LL << 32)) return PN2_EUNSUPPORTED;
    // pn2_conv1x1_wgrad_cf's scratch: moments [PN2_CF_REPL][16][16] doubles, then part [PN2_CF_REPL][128][16] floats (here: [..][C_in][16])
    float *part0 = reinterpret_cast<float *>(reinterpret_cast<double *>(cf_scratch) + PN2_CF_REPL * 256);
    ResDy dy{dZ, nullptr, nullptr, 0, 0, Y, ldy, coef, make_lazy_coef(coef_lazy)};
    const int64_t tiles = P / RES_BM;
    hipStream_t s = pn2_s(stream);
    if (C_out == 96) return launch_split_bwd_res<3, 2, false, true, true>(dy, prev_Y, ld_prev, prev_affine, W, ldw, tiles, nullptr, 0, prev_red, dW, lddw, s, X0, ld0, part0);
    return launch_split_bwd_res<2, 2, false, true, true>(dy, prev_Y, ld_prev, prev_affine, W, ldw, tiles, nullptr, 0, prev_red, dW, lddw, s, X0, ld0, part0);
}
