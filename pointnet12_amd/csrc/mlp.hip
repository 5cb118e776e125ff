// Shared MLP of the set-abstraction / feature-propagation modules on gfx950:
// 1x1 convolution (a [P,K] x [N,K]^T GEMM on v_mfma_f32_32x32x2_f32, exact fp32), training-mode
// BatchNorm (statistics accumulated in the GEMM epilogue, applied on the fly by the consumer),
// ReLU, max over the K neighbours, and the matching backward (dgrad / wgrad / BN reductions).
//
// Activations are position-major: row p = one grouped position, channels contiguous.
// Replaces model/pointnet_util.py:194-199, :251-256, :309-312 and their autograd.
#include "mlp_loaders.h"

namespace {

// ----------------------------------------------------------------------------- epilogues (NT GEMM)
// An epilogue sees the output tile row-wise, 4 consecutive channels at a time (after the accumulators
// have been staged through LDS), so every global access it makes is a coalesced 16-byte one.
// apply() returns the 4 values to store and adds this row's contribution to the two per-channel
// reductions it owns (s0, s1); n is a multiple of 4, n + e >= N lanes must come back as 0.

__device__ __forceinline__ float4 ld4_guard(const float *p, int n, int N) {
    if (n + 3 < N) return ld4(p + n);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) r.x = p[n];
    if (n + 1 < N) r.y = p[n + 1];
    if (n + 2 < N) r.z = p[n + 2];
    return r;
}

struct EpiFwd {             // y = acc + bias -> Y; per-channel sum(y), sum(y*y) -> stats (-> affine block, fused tail)
    float *Y; int ldy; const float *bias; double *stats; FinTail fin;
    static constexpr bool kHasStats = true;
    __host__ __device__ __forceinline__ unsigned *ticket() const { return fin.ticket; }
    __device__ __forceinline__ void tail(int N) const { run_fin_tail(fin, stats, N, NTHREADS); }
    __device__ __forceinline__ bool want_stats() const { return stats != nullptr; }
    __device__ __forceinline__ void prep(int n, int N, float4 (&c)[4]) const { c[0] = ld4_guard(bias, n, N); }
    struct Pre {};
    __device__ __forceinline__ void pre_issue(Pre &, int64_t, int, bool) const {}
    __device__ __forceinline__ void apply(int64_t m, int n, int N, float4 acc, const float4 (&c)[4], const Pre &, float4 &s0,
                                          float4 &s1) const {
        float4 y;
        y.x = acc.x + c[0].x; y.y = acc.y + c[0].y; y.z = acc.z + c[0].z; y.w = acc.w + c[0].w;
        if (n + 3 >= N) {       // zero the pad lanes so the pad columns of Y stay zero
            if (n + 1 >= N) y.y = 0.f;
            if (n + 2 >= N) y.z = 0.f;
            y.w = 0.f;
        }
        {   // streaming store: Y is read again only by later kernels; keeping it out of the L2 leaves the cache to the
            // operand stream and the weights (+2..6 % on the forward GEMMs, tools/bench_kernels.py)
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f yv = {y.x, y.y, y.z, y.w};
            PN2_STREAM_STORE(yv, reinterpret_cast<v4f *>(Y + row_off(m, ldy) + n));
        }
        s0.x += y.x; s0.y += y.y; s0.z += y.z; s0.w += y.w;
        s1.x = __builtin_fmaf(y.x, y.x, s1.x); s1.y = __builtin_fmaf(y.y, y.y, s1.y);
        s1.z = __builtin_fmaf(y.z, y.z, s1.z); s1.w = __builtin_fmaf(y.w, y.w, s1.w);
    }
    __device__ __forceinline__ void flush(int n, int N, double a0, double a1) const {
        double *rep = stats + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;    // spread the same-address queue
        atomicAdd(rep + n, a0);
        atomicAdd(rep + N + n, a1);
    }
};

struct EpiDgradMask {       // dZprev = acc * relu'(prev) -> dXout; sum(dZprev), sum(dZprev*yhat_prev) -> red
    float *dX; int ldx; const float *prevY; int ldp; const float *aff; int lda; double *red; CoefTail ct; const float *zp;
    static constexpr bool kHasStats = true;
    __host__ __device__ __forceinline__ unsigned *ticket() const { return ct.ticket; }
    __device__ __forceinline__ void tail(int N) const { run_coef_tail(ct, red, N, NTHREADS); }
    __device__ __forceinline__ bool want_stats() const { return red != nullptr; }
    __device__ __forceinline__ void prep(int n, int N, float4 (&c)[4]) const {
        Affine a(aff, lda);     // affine blocks are padded to a multiple of 4 with zeros
        c[0] = ld4(a.mean + n); c[1] = ld4(a.scale + n); c[2] = ld4(a.beta + n); c[3] = ld4(a.invstd + n);
    }
    // The previous layer's pre-BN row (for its ReLU mask and BN-backward reduction) is requested before the tile is
    // staged through LDS, so the HBM round trip runs under the two barriers and the LDS image instead of in front
    // of every group of stores.
    struct Pre { float4 y; };
    __device__ __forceinline__ void pre_issue(Pre &q, int64_t m, int n, bool valid) const {
        q.y = ld4(valid ? prevY + row_off(m, ldp) + n : zp);
    }
    __device__ __forceinline__ void apply(int64_t m, int n, int N, float4 acc, const float4 (&c)[4], const Pre &q, float4 &s0,
                                          float4 &s1) const {
        const float4 y = q.y;
        float4 dz;
        dz.x = bn_act(y.x, c[0].x, c[1].x, c[2].x) > 0.f ? acc.x : 0.f;
        dz.y = bn_act(y.y, c[0].y, c[1].y, c[2].y) > 0.f ? acc.y : 0.f;
        dz.z = bn_act(y.z, c[0].z, c[1].z, c[2].z) > 0.f ? acc.z : 0.f;
        dz.w = bn_act(y.w, c[0].w, c[1].w, c[2].w) > 0.f ? acc.w : 0.f;
        {   // streaming store (see EpiFwd); pad lanes: scale = beta = 0 -> 0
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f dv = {dz.x, dz.y, dz.z, dz.w};
            PN2_STREAM_STORE(dv, reinterpret_cast<v4f *>(dX + row_off(m, ldx) + n));
        }
        s0.x += dz.x; s0.y += dz.y; s0.z += dz.z; s0.w += dz.w;
        s1.x = __builtin_fmaf(dz.x, (y.x - c[0].x) * c[3].x, s1.x);
        s1.y = __builtin_fmaf(dz.y, (y.y - c[0].y) * c[3].y, s1.y);
        s1.z = __builtin_fmaf(dz.z, (y.z - c[0].z) * c[3].z, s1.z);
        s1.w = __builtin_fmaf(dz.w, (y.w - c[0].w) * c[3].w, s1.w);
    }
    __device__ __forceinline__ void flush(int n, int N, double a0, double a1) const {
        double *rep = red + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
        atomicAdd(rep + n, a0);
        atomicAdd(rep + N + n, a1);
    }
};

struct EpiStore {           // first layer: dX0 = acc (pad lanes are exact zeros: weight columns n >= N are never fetched)
    float *dX; int ldx;
    static constexpr bool kHasStats = false;
    __host__ __device__ __forceinline__ unsigned *ticket() const { return nullptr; }
    __device__ __forceinline__ void tail(int) const {}
    __device__ __forceinline__ bool want_stats() const { return false; }
    __device__ __forceinline__ void prep(int, int, float4 (&)[4]) const {}
    struct Pre {};
    __device__ __forceinline__ void pre_issue(Pre &, int64_t, int, bool) const {}
    __device__ __forceinline__ void apply(int64_t m, int n, int, float4 acc, const float4 (&)[4], const Pre &, float4 &,
                                          float4 &) const {
        *reinterpret_cast<float4 *>(dX + row_off(m, ldx) + n) = acc;
    }
    __device__ __forceinline__ void flush(int, int, double, double) const {}
};

// ----------------------------------------------------------------------------- NT GEMM core
// C[P,N] = A[P,K] * Bw[N,K]^T for a tall, skinny problem (P ~ 1e5..1e6 rows, K and N <= a few hundred):
// A rows come from a loader functor, Bw is a plain zero-padded matrix, the output goes through an
// epilogue functor.  4 waves as WR x WC, wave tile (BM/WR) x (BN/WC) built from 32x32x2 f32 MFMA tiles.
//
//  * Persistent: gridDim.x <= ~2 workgroups per CU walk the row tiles, so the per-channel reductions
//    stay in registers across tiles and are flushed once per workgroup (LDS combine, then one fp64
//    atomic per channel) -- with one workgroup per tile the 2048-deep same-address atomic queue cost more
//    than the GEMM itself.
//  * Software pipeline over the flattened (row tile, k-step) sequence: the global loads of step s+1 are
//    issued right after the LDS stores of step s and fly under its MFMAs (and under the epilogue at a
//    tile boundary); LDS is double buffered, ONE barrier per k-step.
//  * Epilogue through LDS: the accumulators (column on the lane, rows in registers) are staged into a
//    [BM][BN+4] image aliasing the operand buffers and read back row-wise, so Y / dX / prevY are moved
//    with 16-byte coalesced accesses instead of 64 strided dword stores per lane.
//
// LDS operands are K-contiguous.  One ds_read_b128 gives a lane 4 k-values (k = 8*kb + 4*(lane>>5) + e);
// MFMA e of the group consumes element e from both operands, i.e. the k-order inside an 8-block is
// permuted identically for A and B, which leaves every product pair intact.
#ifdef PN2_STAMP
// Diagnostic build only (make STAMP=1): per-phase shader-cycle sums of wave 0 of every workgroup of the last
// NT GEMM launch, read back with pn2_debug_stamps().  Never compiled into the shipped library.
__device__ unsigned long long pn2_stamp_buf[8 * 2048];
#define STAMP_DECL unsigned long long stamp_t = clock64(), stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = clock64(); stamp_acc[i] += n_ - stamp_t; stamp_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_FLUSH if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048) { for (int i_ = 0; i_ < 8; ++i_) pn2_stamp_buf[blockIdx.x * 8 + i_] = stamp_acc[i_]; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// The weight operand of the NT core.  Rows of pitch `ld` floats; K = contraction length.
//   BNN == false (forward):  p[n * ld + k], n < N rows, k contiguous   (a Conv weight [C_out, C_in] as it is stored)
//   BNN == true  (dgrad):    p[k * ld + n], k < K rows, n contiguous   (the same Conv weight read "down the columns":
//                            dX = dY * W needs W[k = c_out][n = c_in], so no transposed copy is ever made)
// vec != 0: rows are 16-byte aligned (pointer and pitch) and whole float4 reads stay inside a row; otherwise
// the tile is fetched with guarded scalar loads (weights sliced out of a wider matrix, C_in not a multiple of 4).
struct BMat { const float *p; int ld; int K; int vec; const float *zp; };

// A row whose length is not a multiple of 4 still takes the float4 path when its PITCH leaves room for the last quad
// (ld >= round4(len)): the caller then guarantees ZERO pad entries (include/pn2.h) -- the host side makes such a padded copy of
// the 137- / 323- / 515-column weights once per step (pointnet_util._aligned_weight).
inline BMat make_bmat(const float *p, int ld, int K, int contiguous_len) {
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 &&
                     ((contiguous_len & 3) == 0 || ld >= ((contiguous_len + 3) & ~3));
    return BMat{p, ld, K, vec ? 1 : 0, zero_page_dev()};
}

template <int BM, int BN, int BK, bool BNN = false>
struct NtLds {
    static constexpr int kB = BNN ? BK * (BN + 8) : BN * (BK + 4);
    static constexpr int kOperands = 2 * (BM * (BK + 4) + kB);       // floats, double buffered
    static constexpr int kStage = BM * (BN + 8);                      // floats, aliases the operands
    static constexpr int kReduce = NTHREADS * 8 * 2;                  // floats (256 x 8 doubles)
    static constexpr int kFloats = kOperands > kStage ? (kOperands > kReduce ? kOperands : kReduce)
                                                      : (kStage > kReduce ? kStage : kReduce);
};

// Block coordinates as the kernel BODIES see them: the real blockIdx / gridDim for a plain launch, a slice of a one-dimensional
// grid when two bodies share one launch (bwd_pair_kernel below).
struct Bid3 { int x, y, z, gx, gy, gz; };

// ACCS: independent partial accumulators per 32x32 output tile.  The four MFMAs that consume one ds_read_b128 (k-lanes x, y, z,
// w) and the next k block's all accumulate into the SAME tile: one dependent chain per tile, and a dependent
// v_mfma_f32_32x32x2_f32 issues only every ~130 cycles (64 when independent).  Kernels whose waves hold one or two tiles and run
// one or two waves per SIMD (the few-row launches: one workgroup per CU) were bound by exactly that chain (2048 x 1536 -> 256:
// 33 us whether the k-steps were 32 or 64 deep, one or three in flight).  ACCS = 4 gives every k-lane its own accumulator; the
// partials are summed once per tile (48 v_add per tile and wave).
template <int BM, int BN, int BK, int WR, int WC, int MINB, int DEPTH, bool BNN, bool VEC, class ALoad, class Epi, int ACCS = 1, bool ROT = false>
__device__ __forceinline__ void gemm_nt_body(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, int n_lo, const Bid3 bid) {
    constexpr int KS = 4 / (WR * WC);                // waves sharing one wave tile: they split every k-step between them
    static_assert(WR * WC * KS == 4 && (KS == 1 || KS == 2), "four waves");
    static_assert((BK / 8) % KS == 0, "k-step must split evenly over the K-sharing waves");
    constexpr int WTM = BM / WR, WTN = BN / WC;      // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;      // MFMA tiles per wave
    constexpr int LDP = BK + 4;                      // LDS row pitch: 16 consecutive rows cover all 64 banks once
    constexpr int TPR = BK / 4;                      // loader threads per row segment
    constexpr int RPL = NTHREADS / TPR;              // rows per loader pass
    constexpr int A_IT = BM / RPL;
    constexpr int TPRB = BNN ? BN / 4 : TPR;         // B tile in LDS: [BN][BK] (k contiguous) or, BNN, [BK][BN] (n contiguous)
    constexpr int RPLB = NTHREADS / TPRB;            // B rows per loader pass (BN = 96, BNN: 10, sixteen threads idle)
    constexpr int BROWS = BNN ? BK : BN;
    constexpr int B_IT = (BROWS + RPLB - 1) / RPLB;  // the last pass may be partly live
    constexpr int LDBS = BNN ? BN + 8 : LDP;         // LDS pitch of a B row (BNN: lanes 32..63 read 4 rows further down,
                                                     // 4 * (BN + 8) = 32 mod 64 banks -> the two half-waves never collide)
    constexpr int CG = BN / 4;                       // float4 column groups of the output tile
    constexpr int RPP = NTHREADS / CG;               // rows per epilogue pass
    constexpr int LDC = BN + 8;                      // lanes 32..63 of an accumulator store sit 4 rows down: 4 * LDC = 32 mod 64 banks
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 loader threads");

    __shared__ __attribute__((aligned(16))) float lds[NtLds<BM, BN, BK, BNN>::kFloats];
    float *As = lds;                                  // [2][BM*LDP]
    float *Bs = lds + 2 * BM * LDP;                   // [2][BROWS*LDBS]

    extern __shared__ __attribute__((aligned(16))) float nt_tab[];   // ALoad::kTab * K4 floats (may be empty)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kh = wave / (WR * WC);                 // which share of the k-step this wave multiplies (KS == 1: 0)
    const int wr = (wave % (WR * WC)) / WC, wc = wave % WC;
    const int l31 = lane & 31, lh = lane >> 5;
    const int n0 = n_lo + bid.y * BN;           // n_lo > 0: this launch covers the output columns from n_lo on
    const int64_t tiles_m = (P + BM - 1) / BM;
    aload.prologue();                                 // consumer-side BatchNorm: fill the block this loader reads (bn_tail.h)
    if (ALoad::kTab > 0) {                            // per-channel loader constants: global -> LDS once
        const float *src = aload.tab_src();
        for (int i = t * 4; i < ALoad::kTab * K4; i += NTHREADS * 4)
            *reinterpret_cast<float4 *>(nt_tab + i) = ld4(src + i);
        __syncthreads();
    }
    const int nk = (K4 + BK - 1) / BK;
    // Every workgroup walks the k-steps of a tile in its own rotation (sum order differs per workgroup, deterministically):
    // launched together, the workgroups of a few-row product otherwise read the SAME column block of X and W at the same
    // time, whose cache lines -- one per row, a row pitch apart -- sit on very few L2 channels when the pitch is a multiple
    // of a few KB (K = 512, 1536, ...).
    const int krot = ROT ? (int)((bid.x * 7u + bid.y * 3u) % (unsigned)nk) : 0;
    auto kmap = [&](int ks_) { const int v = ks_ + krot; return ROT ? (v >= nk ? v - nk : v) : ks_; };
    const int lrow = t / TPR, lkq = (t % TPR) * 4;    // loader coordinates
    const int brow = t / TPRB, bcq = (t % TPRB) * 4;  // B loader coordinates (row of the LDS layout, first of 4 columns)
    const int ecg = t % CG, erow = t / CG;            // epilogue coordinates
    const int en = n0 + ecg * 4;

    float4 ec[4];
    if (en < ((N + 3) & ~3)) epi.prep(en, N, ec);
    double st[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st[e] = 0.0;

    // Register ring of DEPTH prefetched k-steps: the RAW operands of step s+DEPTH are requested right after the
    // LDS stores of step s; they are only turned into operand values (finish) when their own step comes up, so the
    // requests stay in flight under DEPTH steps of MFMA work (and a tile's epilogue).
    typename ALoad::template Raw<A_IT> ra[DEPTH];
    float4 rb[DEPTH][B_IT];
    auto fetch = [&](typename ALoad::template Raw<A_IT> &qa, float4 (&qb)[B_IT], int64_t tile, int ks_) {
        const int ks = kmap(ks_);
        const int k = ks * BK + lkq;
        const bool kvalid = k < K4 && tile < tiles_m;
        aload.template issue<A_IT>(qa, tile * BM + lrow, RPL, k, P, kvalid);
        const float *zp = bm.zp;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = brow + i * RPLB;
            // (row, col) = global coordinates of the first of this thread's four elements; `lim` bounds the run
            const int row = BNN ? ks * BK + r : n0 + r, col = BNN ? n0 + bcq : ks * BK + bcq;
            const int rlim = BNN ? bm.K : N, lim = BNN ? N : bm.K;
            const bool ok = brow < RPLB && r < BROWS && tile < tiles_m && row < rlim;
            const float *src = bm.p + (int64_t)row * bm.ld + col;
            if (VEC) {
                qb[i] = ld4((ok && col < lim) ? src : zp);
            } else {
                qb[i].x = *((ok && col < lim) ? src : zp);
                qb[i].y = *((ok && col + 1 < lim) ? src + 1 : zp);
                qb[i].z = *((ok && col + 2 < lim) ? src + 2 : zp);
                qb[i].w = *((ok && col + 3 < lim) ? src + 3 : zp);
            }
        }
    };

    int64_t tile = bid.x, ptile = bid.x;   // current step and prefetch cursor over (tile, ks)
    int ks = 0, pks = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        fetch(ra[d], rb[d], ptile, pks);             // beyond the last tile every lane is predicated off (m >= P)
        if (++pks == nk) { pks = 0; ptile += bid.gx; }
    }
    // The first k-step of a tile takes its operands from `fa` / `fb`: already transformed, in registers.  They are
    // produced BEFORE the previous tile's epilogue issues its global stores (gfx9 has one vmcnt for loads and stores
    // alike: waiting for a prefetched operand after the stores means waiting for the stores' HBM acknowledgements --
    // 1-2 us per tile in front of its first MFMA; an ablation build without the stores ran 12 % faster).
    constexpr bool PF = DEPTH == 1;
    float4 fa[A_IT], fb[B_IT];
    auto prefinish = [&](int64_t nt) {
        const int k0 = kmap(0) * BK + lkq;
        const bool kv0 = k0 < K4 && nt < tiles_m;
        const typename ALoad::Params ap0 = aload.params_tab(nt_tab, K4, k0, kv0);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) fa[i] = aload.template finish<A_IT>(ra[0], i, kv0 && (nt * BM + lrow + i * RPL < P), ap0);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) fb[i] = rb[0][i];
    };
    if (PF) prefinish(tile);
    int buf = 0;
    static_assert(ACCS == 1 || ACCS == 2 || ACCS == 4, "partial accumulators");
    f32x16 acc[TM][TN], accp[ACCS > 1 ? ACCS - 1 : 1][TM][TN];     // accp: the extra partials (ACCS > 1)
    STAMP_DECL
    while (tile < tiles_m) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (tile >= tiles_m) break;
            const int64_t m0 = tile * BM;
            if (ks == 0) {
                // keep this a (uniform) branch: if-converted it became one v_cndmask per accumulator register in EVERY
                // k-step, sitting between dependent MFMAs on the same accumulators
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            acc[i][j][r] = 0.f;
#pragma unroll
                            for (int u = 0; u < ACCS - 1; ++u) accp[u][i][j][r] = 0.f;
                        }
            }
            float *Ab = As + buf * (BM * LDP), *Bb = Bs + buf * (BROWS * LDBS);
            if (PF && ks == 0) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i) *reinterpret_cast<float4 *>(&Ab[(lrow + i * RPL) * LDP + lkq]) = fa[i];
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    if ((NTHREADS % TPRB == 0 && BROWS % RPLB == 0) || (brow < RPLB && brow + i * RPLB < BROWS))
                        *reinterpret_cast<float4 *>(&Bb[(brow + i * RPLB) * LDBS + bcq]) = fb[i];
                asm volatile("" ::: "memory");      // keep these stores here: merged with the other branch's they cost 12 v_mov per k-step
            } else {
                const int kc = kmap(ks) * BK + lkq;
                const bool kv = kc < K4;
                const typename ALoad::Params ap = aload.params_tab(nt_tab, K4, kc, kv);
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    *reinterpret_cast<float4 *>(&Ab[(lrow + i * RPL) * LDP + lkq]) =
                        aload.template finish<A_IT>(ra[d], i, kv && (m0 + lrow + i * RPL < P), ap);
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    if ((NTHREADS % TPRB == 0 && BROWS % RPLB == 0) || (brow < RPLB && brow + i * RPLB < BROWS))
                        *reinterpret_cast<float4 *>(&Bb[(brow + i * RPLB) * LDBS + bcq]) = rb[d][i];
            }
            fetch(ra[d], rb[d], ptile, pks);
            if (++pks == nk) { pks = 0; ptile += bid.gx; }
            STAMP(0)
            __syncthreads();
            STAMP(1)
#pragma unroll
            for (int kbi = 0; kbi < BK / 8 / KS; ++kbi) {
                const int kb = kbi * KS + kh;
                float4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&Ab[(wr * WTM + i * 32 + l31) * LDP + kb * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (BNN) {                        // four k-rows of the [BK][BN] image, lanes on consecutive n
                        const float *bp = &Bb[(kb * 8 + lh * 4) * LDBS + wc * WTN + j * 32 + l31];
                        b[j] = make_float4(bp[0], bp[LDBS], bp[2 * LDBS], bp[3 * LDBS]);
                    } else {
                        b[j] = *reinterpret_cast<const float4 *>(&Bb[(wc * WTN + j * 32 + l31) * LDP + kb * 8 + lh * 4]);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (ACCS == 4) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                            accp[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, accp[0][i][j], 0, 0, 0);
                            accp[ACCS > 2 ? 1 : 0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, accp[ACCS > 2 ? 1 : 0][i][j], 0, 0, 0);
                            accp[ACCS > 2 ? 2 : 0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, accp[ACCS > 2 ? 2 : 0][i][j], 0, 0, 0);
                        } else if (ACCS == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                            accp[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, accp[0][i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                            accp[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, accp[0][i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                        }
                    }
            }
            buf ^= 1;
            STAMP(2)
            if (++ks < nk) continue;
            ks = 0;
            if (ACCS > 1) {                                    // fold the partial accumulators (fixed order: deterministic)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[i][j][r];
#pragma unroll
                            for (int u = 0; u < ACCS - 1; ++u) v += accp[u][i][j][r];
                            acc[i][j][r] = v;
                        }
            }

            // ---- epilogue: accumulators -> LDS image [BM][LDC] (aliases the operand buffers) -> rows
            if (PF) prefinish(tile + bid.gx);           // next tile's first operands: consumed before any store goes out
            constexpr int EP_IT = (BM + RPP - 1) / RPP;
            typename Epi::Pre pre[EP_IT];
            const bool ecol = en < ((N + 3) & ~3) && erow < RPP;    // (NTHREADS % CG) threads have no row when BN = 96
#pragma unroll
            for (int i = 0; i < EP_IT; ++i)
                epi.pre_issue(pre[i], m0 + erow + i * RPP, en, ecol && erow + i * RPP < BM && m0 + erow + i * RPP < P);
            __syncthreads();                               // every wave is done reading the operands
            STAMP(3)
            if (KS == 1 || kh == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)       // D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
                            lds[(wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wc * WTN + j * 32 + l31] = acc[i][j][r];
            }
            if (KS > 1) {                                  // the other half of every k-step: add it into the image
                __syncthreads();
                if (kh == 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                lds[(wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wc * WTN + j * 32 + l31] += acc[i][j][r];
                }
            }
            STAMP(4)
            __syncthreads();
            STAMP(5)
            if (ecol) {
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
                for (int i = 0; i < EP_IT; ++i) {
                    const int r = erow + i * RPP;
                    const int64_t m = m0 + r;
                    if (r < BM && m < P)
                        epi.apply(m, en, N, *reinterpret_cast<const float4 *>(&lds[r * LDC + ecg * 4]), ec, pre[i], s0, s1);
                }
                if (Epi::kHasStats) {
                    st[0] += (double)s0.x; st[1] += (double)s0.y; st[2] += (double)s0.z; st[3] += (double)s0.w;
                    st[4] += (double)s1.x; st[5] += (double)s1.y; st[6] += (double)s1.z; st[7] += (double)s1.w;
                }
            }
            STAMP(6)
            __syncthreads();                               // image consumed before the next tile's operands land
            STAMP(7)
            tile += bid.gx;
        }
    }
    STAMP_FLUSH

    if (Epi::kHasStats && epi.want_stats()) {              // combine the RPP row-threads of each column, flush once
        double *red = reinterpret_cast<double *>(lds);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[t * 8 + e] = st[e];
        __syncthreads();
        if (t < BN) {
            const int cg = t >> 2, e = t & 3;
            double a0 = 0.0, a1 = 0.0;
            for (int j = 0; j < RPP; ++j) {
                a0 += red[(j * CG + cg) * 8 + e];
                a1 += red[(j * CG + cg) * 8 + 4 + e];
            }
            if (n0 + t < N) epi.flush(n0 + t, N, a0, a1);
        }
        if (epi.ticket() != nullptr && tail_is_last_block(epi.ticket(), bid.gx * bid.gy)) epi.tail(N);
    }
}

template <int BM, int BN, int BK, int WR, int WC, int MINB, int DEPTH, bool BNN, bool VEC, class ALoad, class Epi, int ACCS = 1, bool ROT = false>
__global__ __launch_bounds__(NTHREADS, MINB) void gemm_nt_kernel(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, int n_lo) {
    gemm_nt_body<BM, BN, BK, WR, WC, MINB, DEPTH, BNN, VEC, ALoad, Epi, ACCS, ROT>(aload, bm, P, K4, N, epi, n_lo,
                                                                                  Bid3{(int)blockIdx.x, (int)blockIdx.y, 0, (int)gridDim.x, (int)gridDim.y, 1});
}

// ----------------------------------------------------------------------------- dgrad + wgrad of one layer in ONE launch
// The few-row and mid-size layers (sa3 / sa4 / FP stacks: P = 1 k .. 64 k rows) run their data gradient and their weight
// gradient as two dependent-looking launches that do not depend on each other, each a ~25 us kernel that leaves half of the chip
// idle (64 .. 256 workgroups of one wave per SIMD).  Forking the weight gradient onto a side stream costs the graph executor
// more than the overlap returns (measured twice: SSG 2.75 -> 2.89 ms).  Here the two kernel BODIES share one launch instead:
// the first blocks of a one-dimensional grid run the NT body (dX), the rest the TN body (dW) -- no fork, no join, one launch
// less on the chain, and the chip is filled by both.  pn2_conv1x1_bwd_pair posts the weight-gradient half as a pending job;
// the dgrad dispatch picks it up in the one leaf that has a pair instantiation (64 x 128 x 16 tiles) and reports back.
struct PairJob {
    bool active, taken;
    const float *X; int ldx; const float *x_aff;
    float *dW; int lddw; int M, N;
};
// (passed down the dispatch explicitly: round 4 posted it in thread-local storage, hidden state behind a C ABI that promises none)

template <int BM, int BN, int BK, int WR, int WC, bool VEC, class ALoad, class Epi>
bool launch_bwd_pair(const PairJob &j, ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, unsigned nt_gx, unsigned nt_gy, int *rc);

template <class A, class B> struct pn2_same { static constexpr bool v = false; };
template <class A> struct pn2_same<A, A> { static constexpr bool v = true; };

template <int BM, int BN, int BK, int WR, int WC, int MINB, int DEPTH, bool BNN, bool VEC, int ACCS = 1, bool ROT = false, class ALoad, class Epi>
int launch_nt(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, int n_lo = 0, int n_hi = 0, PairJob *job = nullptr) {
    int64_t tiles_m = pn2_cdiv(P, BM);
    unsigned tiles_n = (unsigned)pn2_cdiv((n_hi ? n_hi : N) - n_lo, BN);     // output columns [n_lo, n_hi) of N
    int64_t cap = (int64_t)pn2_num_cus() * MINB / tiles_n;  // MINB resident workgroups per CU in total
    if (cap < 1) cap = 1;
    unsigned gx = (unsigned)(tiles_m < cap ? tiles_m : cap);
    if constexpr (BNN && VEC && BM == 64 && BN == 128 && BK == 16 && WR == 2 && WC == 2 && DEPTH == 1 && ACCS == 1 && !ROT &&
                  (pn2_same<ALoad, LoadDyDense>::v || pn2_same<ALoad, LoadDyPooled>::v) &&
                  (pn2_same<Epi, EpiDgradMask>::v || pn2_same<Epi, EpiStore>::v)) {
        if (job != nullptr && job->active && !job->taken && n_lo == 0 && n_hi == 0) {
            int rc = PN2_OK;
            // both halves must be RESIDENT together (two workgroups of 70 KB fit a CU): one per CU for each body -- with the
            // persistent NT grid at its usual three per CU its blocks, dispatched first, took every slot and the TN half ran
            // behind them (first version: SSG 2.65 -> 2.60 ms only)
            int64_t cap1 = (int64_t)pn2_num_cus() / tiles_n;
            if (cap1 < 1) cap1 = 1;
            const unsigned gx1 = (unsigned)(tiles_m < cap1 ? tiles_m : cap1);
            if (launch_bwd_pair<BM, BN, BK, WR, WC, VEC>(*job, aload, bm, P, K4, N, epi, s, gx1, tiles_n, &rc)) {
                job->taken = true;
                return rc;
            }
        }
    }
    PN2_NOTE_KERNEL(gemm_nt_kernel<BM, BN, BK, WR, WC, MINB, DEPTH, BNN, VEC, ALoad, Epi, ACCS, ROT>);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, BK, WR, WC, MINB, DEPTH, BNN, VEC, ALoad, Epi, ACCS, ROT>), dim3(gx, tiles_n), dim3(NTHREADS),
                       (size_t)ALoad::kTab * K4 * sizeof(float), s, aload, bm, P, K4, N, epi, n_lo);
    return pn2_launch_status();
}


// ----------------------------------------------------------------------------- few-row NT GEMM
// The sa3 / sa4 / fp4 / fp3 / fp2 products (P = 2 k .. 8 k rows, K up to 1536) have too few output tiles for the persistent
// core above: it puts one workgroup on a CU, and every k-step of that workgroup is load -> transform -> LDS -> barrier -> MFMA
// with nothing else resident to run under either half (matrix pipe 31 % busy, see dispatch_nt_vec).  This kernel drops the
// shared staging altogether:
//  * one workgroup per 32 x 64 output tile; its four waves SPLIT K (contiguous quarters, 32-deep stages) and never meet
//    until the end -- no barrier, no LDS operand image in the main loop;
//  * a lane fetches its MFMA operands straight from global memory: 16 bytes (k = 8 kb + 4 (lane >> 5) .. + 3) of row
//    lane & 31 of the activation tile and of the two 32-row weight tiles.  The four k blocks of a stage touch the same 128-byte
//    lines back to back (L1 hits); per stage and wave 12 (forward) .. 44 (pooled dgrad) requests against 32 MFMAs;
//  * two register sets: the requests of stage s + 1 are in flight under the MFMAs of stage s (straight-line code, loads
//    only: the compiler's vmcnt bookkeeping stays exact);
//  * addresses are running pointers (32-bit offsets from per-lane bases), no predicates: P % 32, N % 64 and K % 32 must be 0
//    (every few-row layer of the four networks except the 515- and 137-column ones, which stay on the core above);
//  * the four partial tiles are summed in LDS in a fixed order (deterministic) and leave through the same epilogue
//    functors as the core (bias / ReLU mask, per-channel reductions, fused BatchNorm tails).
template <class T, class U> struct fr_same { static constexpr bool v = false; };
template <class T> struct fr_same<T, T> { static constexpr bool v = true; };

// KS: waves sharing one 32 x 64 tile (they split K); a workgroup owns 4 / KS consecutive tiles of the (row tile, column tile) grid,
// column tiles fastest -- its waves then read the same activation rows.  The host picks KS so that about four waves per CU
// have 4+ stages each.
// (one workgroup per CU in the bound: under the two-workgroup cap of 256 registers hipcc spilled 13 registers of the pooled
// data gradient that it does not need when left alone -- the instantiations use 116 .. 246 registers, so two still fit)
template <int KS, bool BNN, class ALoad, class Epi>
__global__ __launch_bounds__(NTHREADS, 1) void fewrow_nt_kernel(ALoad aload, BMat bm, int K4, int N, Epi epi) {
    constexpr bool kPlain = fr_same<ALoad, LoadPlain>::v, kBn = fr_same<ALoad, LoadBnRelu>::v;
    constexpr bool kDense = fr_same<ALoad, LoadDyDense>::v, kPooled = fr_same<ALoad, LoadDyPooled>::v;
    static_assert(kPlain || kBn || kDense || kPooled, "unknown operand loader");
    static_assert(KS == 1 || KS == 2 || KS == 4, "waves per tile");
    constexpr int kTabRows = kPlain ? 0 : kBn ? 3 : 4;
    constexpr int NT = 4 / KS;
    constexpr int BM = 32, BN = 64, LDC = BN + 8, CG = BN / 4, RPP = NTHREADS / CG, EP_IT = BM / RPP;
    __shared__ __attribute__((aligned(16))) float part[4 * BM * LDC];          // 36 KB: one (partial) tile per wave
    extern __shared__ __attribute__((aligned(16))) float fr_tab[];              // kTabRows x K4 per-channel constants
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N >> 6;
    const int tile = blockIdx.x * NT + wave / KS, ks = wave % KS;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    aload.prologue();                                 // consumer-side BatchNorm (bn_tail.h)
    if constexpr (kTabRows > 0) {
        const float *src;
        int pitch;
        if constexpr (kBn) { src = aload.aff; pitch = aload.ldx; } else { src = aload.coef; pitch = aload.ldc; }
        for (int i = t * 4; i < kTabRows * K4; i += NTHREADS * 4) {
            const int r = i / K4, c = i - r * K4;
            *reinterpret_cast<float4 *>(fr_tab + i) = ld4(src + r * pitch + c);
        }
        __syncthreads();
    }
    const int nst = K4 >> 5;
    const int s_lo = (nst * ks) / KS, s_hi = (nst * (ks + 1)) / KS;

    // per-lane bases: everything after this is base + 32-bit stage offset
    const float *a0p, *a1p = nullptr;
    const int32_t *a2p = nullptr;
    int kk = 0;
    if constexpr (kPlain || kBn) a0p = aload.X + row_off(m0 + l31, aload.ldx) + 4 * lh;
    if constexpr (kDense) {
        a0p = aload.Y + row_off(m0 + l31, aload.ldy) + 4 * lh;
        a1p = aload.dZ + row_off(m0 + l31, aload.ldz) + 4 * lh;
    }
    if constexpr (kPooled) {
        const unsigned mi = (unsigned)(m0 + l31);
        const unsigned g = aload.kshift >= 0 ? mi >> aload.kshift : mi / (unsigned)aload.Kp;
        kk = (int)(mi - g * (unsigned)aload.Kp);
        a0p = aload.Y + row_off(mi, aload.ldy) + 4 * lh;
        a1p = aload.dZp + row_off(g, aload.ldo) + 4 * lh;
        a2p = aload.arg + row_off(g, aload.ldo) + 4 * lh;
    }
    const float *bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        bp[j] = BNN ? bm.p + row_off(4 * lh, bm.ld) + n0 + 32 * j + l31 : bm.p + row_off(n0 + 32 * j + l31, bm.ld) + 4 * lh;
    const int ldb = bm.ld;

    struct Stage { float4 a0[4], a1[4]; int4 a2[4]; float4 b[2][4]; };
    auto load = [&](Stage &st, int s) {
        const int k = 32 * s;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st.a0[kb] = ld4(a0p + k + 8 * kb);
            if constexpr (kDense || kPooled) st.a1[kb] = ld4(a1p + k + 8 * kb);
            if constexpr (kPooled) st.a2[kb] = ld4i(a2p + k + 8 * kb);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if constexpr (!BNN) {
                    st.b[j][kb] = ld4(bp[j] + k + 8 * kb);
                } else {
                    const float *q = bp[j] + (unsigned)(k + 8 * kb) * (unsigned)ldb;
                    st.b[j][kb] = make_float4(q[0], q[ldb], q[2 * ldb], q[3 * ldb]);
                }
            }
    };
    // two accumulator chains per 32 x 32 tile (k lanes x, z / y, w): with one or two waves on a SIMD a single chain per tile
    // leaves the matrix pipe waiting on the previous MFMA's result every other issue
    // (forward only: the dgrad loaders' two or three tensors in flight leave no room for the second set, and measured slower with it)
    constexpr int CH = BNN ? 1 : 2;
    f32x16 acc[2][CH];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][c][r] = 0.f;
    auto compute = [&](const Stage &st, int s) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int k = 32 * s + 8 * kb + 4 * lh;
            float4 a;
            if constexpr (kPlain) a = st.a0[kb];
            if constexpr (kBn) {
                const float4 mu = *reinterpret_cast<const float4 *>(fr_tab + k), sc = *reinterpret_cast<const float4 *>(fr_tab + K4 + k),
                             be = *reinterpret_cast<const float4 *>(fr_tab + 2 * K4 + k), x = st.a0[kb];
                a.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
                a.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
                a.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
                a.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
            }
            if constexpr (kDense) a = dy_from(st.a1[kb], st.a0[kb], dy_params_tab(fr_tab, K4, k, true));
            if constexpr (kPooled) {
                const float4 go = st.a1[kb];
                const int4 ar = st.a2[kb];
                float4 dz;
                dz.x = ar.x == kk ? go.x : 0.f;
                dz.y = ar.y == kk ? go.y : 0.f;
                dz.z = ar.z == kk ? go.z : 0.f;
                dz.w = ar.w == kk ? go.w : 0.f;
                a = dy_from(dz, st.a0[kb], dy_params_tab(fr_tab, K4, k, true));
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, st.b[0][kb].x, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, st.b[1][kb].x, acc[1][0], 0, 0, 0);
            acc[0][CH - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, st.b[0][kb].y, acc[0][CH - 1], 0, 0, 0);
            acc[1][CH - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, st.b[1][kb].y, acc[1][CH - 1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, st.b[0][kb].z, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, st.b[1][kb].z, acc[1][0], 0, 0, 0);
            acc[0][CH - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, st.b[0][kb].w, acc[0][CH - 1], 0, 0, 0);
            acc[1][CH - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, st.b[1][kb].w, acc[1][CH - 1], 0, 0, 0);
        }
    };

    // The epilogue's own reads (the previous layer's pre-BN rows, for the ReLU mask): with one or two tiles per workgroup they go out
    // before the main loop (8 .. 16 registers), with four tiles after it (32 registers that the stage sets need until then).
    const int ecg = t % CG, erow = t / CG;
    typename Epi::Pre pre[NT][EP_IT];
    auto issue_pre = [&]() {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const int te = blockIdx.x * NT + tl;
#pragma unroll
            for (int i = 0; i < EP_IT; ++i)
                epi.pre_issue(pre[tl][i], (te / tiles_n) * BM + erow + i * RPP, (te % tiles_n) * BN + ecg * 4, true);
        }
    };
    if constexpr (NT <= 2) issue_pre();

    if (s_lo < s_hi) {
        // The phases are pinned with scheduling barriers: left alone, the machine scheduler sinks every request down to its first
        // use (load, s_waitcnt vmcnt(0), MFMA -- the whole point of the second register set undone).
        Stage sa, sb;
        int s = s_lo;
        load(sa, s);
        __builtin_amdgcn_sched_barrier(0);
        for (; s + 2 <= s_hi; s += 2) {
            load(sb, s + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(sa, s);
            __builtin_amdgcn_sched_barrier(0);
            load(sa, s + 2 < s_hi ? s + 2 : s_hi - 1);            // always issued (clamped): straight-line request counting
            __builtin_amdgcn_sched_barrier(0);
            compute(sb, s + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s < s_hi) compute(sa, s);                             // odd stage count: the clamped request above fetched it
    }

    // ---- epilogue: every tile of the workgroup row-wise by all 256 threads (16-byte coalesced), partials summed in wave order.
    if constexpr (NT > 2) issue_pre();
    float *pw = part + wave * BM * LDC;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)               // D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            pw[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + j * 32 + l31] = CH == 2 ? acc[j][0][r] + acc[j][CH - 1][r] : acc[j][0][r];
    __syncthreads();
    float4 s0[NT], s1[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int te = blockIdx.x * NT + tl;
        const int me = (te / tiles_n) * BM, en = (te % tiles_n) * BN + ecg * 4;
        float4 ec[4];
        epi.prep(en, N, ec);
        s0[tl] = make_float4(0.f, 0.f, 0.f, 0.f);
        s1[tl] = s0[tl];
#pragma unroll
        for (int i = 0; i < EP_IT; ++i) {
            const int r = erow + i * RPP;
            float4 v = *reinterpret_cast<const float4 *>(&part[(tl * KS * BM + r) * LDC + ecg * 4]);
#pragma unroll
            for (int w = 1; w < KS; ++w) {
                const float4 u = *reinterpret_cast<const float4 *>(&part[((tl * KS + w) * BM + r) * LDC + ecg * 4]);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            epi.apply(me + r, en, N, v, ec, pre[tl][i], s0[tl], s1[tl]);
        }
    }
    if (Epi::kHasStats && epi.want_stats()) {              // combine the RPP row-threads of each column, flush once per tile
        double *red = reinterpret_cast<double *>(part);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            __syncthreads();
            red[t * 8 + 0] = (double)s0[tl].x; red[t * 8 + 1] = (double)s0[tl].y; red[t * 8 + 2] = (double)s0[tl].z; red[t * 8 + 3] = (double)s0[tl].w;
            red[t * 8 + 4] = (double)s1[tl].x; red[t * 8 + 5] = (double)s1[tl].y; red[t * 8 + 6] = (double)s1[tl].z; red[t * 8 + 7] = (double)s1[tl].w;
            __syncthreads();
            if (t < BN) {
                const int cg = t >> 2, e = t & 3;
                double a0 = 0.0, a1 = 0.0;
                for (int j = 0; j < RPP; ++j) {
                    a0 += red[(j * CG + cg) * 8 + e];
                    a1 += red[(j * CG + cg) * 8 + 4 + e];
                }
                epi.flush(((blockIdx.x * NT + tl) % tiles_n) * BN + t, N, a0, a1);
            }
        }
        if (epi.ticket() != nullptr && tail_is_last_block(epi.ticket(), gridDim.x)) epi.tail(N);
    }
}

// false: not this shape (the caller goes on to the persistent core)
template <bool BNN, class ALoad, class Epi>
bool launch_fewrow(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, int *rc) {
    // measured against the persistent core (tools/bench_kernels.py, us): forward 2048 x 1536 -> 256 38.7 -> 25.4, 2048 x 512 -> 1024
    // 33.3 -> 32.5, 8192 x 576 -> 256 37.5 -> 36.3; dgrad 2048 x 1024 -> 512 (pooled) 49.1 -> 39.7, 2048 x 512 -> 256 16.9 -> 14;
    // the 8192-row dgrads (dword weight requests down the columns, 1024+ tiles) are 1 .. 3 us slower here and stay on the core
    const int max_tiles = BNN ? pn2_opt(PN2_OPT_FEWROW_MAX_TILES_DGRAD) : pn2_opt(PN2_OPT_FEWROW_MAX_TILES);   // 0: off
    const int force_ks = pn2_opt(PN2_OPT_FEWROW_KS);
    // K = bm.K is the contraction length as the caller states it: a multiple of 32 means there are no pad columns at all
    if (!bm.vec || (P & 31) || (N & 63) || (K4 & 31) || bm.K != K4 || K4 < 128) return false;
    const int64_t tiles = (P / 32) * (N / 64);
    if (tiles > max_tiles || (tiles & 3)) return false;
    constexpr int tab_rows = fr_same<ALoad, LoadPlain>::v ? 0 : fr_same<ALoad, LoadBnRelu>::v ? 3 : 4;
    const size_t dyn = (size_t)tab_rows * K4 * sizeof(float);
    if (dyn + 4 * 32 * 72 * sizeof(float) > 64 * 1024) return false;
    // about four waves per CU with as many stages each as that allows
    int ks = tiles * 4 <= 4 * pn2_num_cus() ? 4 : tiles * 2 <= 4 * pn2_num_cus() ? 2 : 1;
    if (force_ks) ks = force_ks;
    const dim3 grid((unsigned)(tiles * ks / 4));
    if (ks == 4) { PN2_NOTE_KERNEL(fewrow_nt_kernel<4, BNN, ALoad, Epi>); hipLaunchKernelGGL((fewrow_nt_kernel<4, BNN, ALoad, Epi>), grid, dim3(NTHREADS), dyn, s, aload, bm, K4, N, epi); }
    else if (ks == 2) { PN2_NOTE_KERNEL(fewrow_nt_kernel<2, BNN, ALoad, Epi>); hipLaunchKernelGGL((fewrow_nt_kernel<2, BNN, ALoad, Epi>), grid, dim3(NTHREADS), dyn, s, aload, bm, K4, N, epi); }
    else { PN2_NOTE_KERNEL(fewrow_nt_kernel<1, BNN, ALoad, Epi>); hipLaunchKernelGGL((fewrow_nt_kernel<1, BNN, ALoad, Epi>), grid, dim3(NTHREADS), dyn, s, aload, bm, K4, N, epi); }
    *rc = pn2_launch_status();
    return true;
}

template <bool BNN, class ALoad, class Epi>
int dispatch_nt_vec(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, PairJob *job = nullptr) {
    const int cfg = pn2_opt(PN2_OPT_NT_CFG);     // tuning hook (tools/bench_kernels.py)
    {
        int rc = PN2_OK;
        if (launch_fewrow<BNN>(aload, bm, P, K4, N, epi, s, &rc)) return rc;
    }
    // Few rows (the sa3 / fp3 / fp2 stages: P = 2 k .. 8 k): 64x128 tiles would leave most CUs without a workgroup
    // -- 64x64 tiles double the workgroup count (fwd 2048 x 1536 -> 256: 64 workgroups -> 128)
    // -- and 32x64 tiles whose four waves split every k-step in two (summed in the LDS image) double it again.
    // (register budget: three workgroups per CU for the loaders that keep two or three tensors in flight)
    constexpr int SMALL_MINB = ALoad::kRegs >= 8 ? 2 : 3;
    // These few-row launches put ONE workgroup on a CU (one wave per SIMD).  Round 3 looked for what bounds them (2048 x 1536 ->
    // 256: 35 us, 45 TF) with four A/B builds, all selectable here and all measured EQUAL within 1 us: three k-steps in
    // flight (2), four independent accumulator chains per tile (3), a rotated k order per workgroup against L2 channel camping
    // (4), and 64-deep k-steps.  The counters say why (tools/exp/pmc_shape.sh): L2 requests return in ~190 cycles, but a
    // k-step issues 8.4 VALU + 4.5 SALU instructions per MFMA (64-bit addresses, predicates, the BatchNorm transform) in a
    // phase of its own -- with a single wave per SIMD nothing runs under the MFMAs (matrix pipe 31 % busy).  What helps is a
    // second workgroup per CU in another phase, not a deeper ring.
    const int sdepth = pn2_opt(PN2_OPT_NT_SMALL_DEPTH);
    if (N > 32 && cfg != 7 && cfg != 8 && pn2_cdiv(P, 64) * pn2_cdiv(N, 64) * 2 <= pn2_num_cus()) {
        if (sdepth == 4) return launch_nt<32, 64, 32, 1, 2, 2, 1, BNN, true, 1, true>(aload, bm, P, K4, N, epi, s);   // rotated k order
        if (sdepth == 3) return launch_nt<32, 64, 32, 1, 2, 2, 1, BNN, true, 4>(aload, bm, P, K4, N, epi, s);         // four chains per tile
        if (sdepth == 2) return launch_nt<32, 64, 32, 1, 2, 2, 3, BNN, true>(aload, bm, P, K4, N, epi, s);            // three k-steps in flight
        return launch_nt<32, 64, 32, 1, 2, SMALL_MINB, 1, BNN, true>(aload, bm, P, K4, N, epi, s);
    }
    if (N > 32 && cfg != 7 && pn2_cdiv(P, 64) * pn2_cdiv(N, 128) * 2 <= pn2_num_cus()) {
        if (sdepth == 4) return launch_nt<64, 64, 32, 2, 2, 2, 1, BNN, true, 1, true>(aload, bm, P, K4, N, epi, s);
        return launch_nt<64, 64, 32, 2, 2, SMALL_MINB, 1, BNN, true>(aload, bm, P, K4, N, epi, s);
    }
    if (N <= 32) return launch_nt<128, 32, 32, 4, 1, 2, 1, BNN, true>(aload, bm, P, K4, N, epi, s);
    if (N <= 64) return launch_nt<128, 64, 32, 2, 2, 2, 1, BNN, true>(aload, bm, P, K4, N, epi, s);
    // 65..96 output channels (64->96, 128->96 in MSG sa1): an exact 96-wide tile instead of 25 % padding MFMAs
    if (N <= 96 && cfg != 9) {
        if constexpr (ALoad::kRegs >= 8) return launch_nt<128, 96, 16, 4, 1, 2, 1, BNN, true>(aload, bm, P, K4, N, epi, s);   // no spills
        else return launch_nt<128, 96, 16, 4, 1, 3, 1, BNN, true>(aload, bm, P, K4, N, epi, s);
    }
    // 129..224 output channels (128->196, 256->196 of MSG sa2): 128 columns on the 64x128 tile and the remainder on the
    // narrowest tile that holds it, as a second launch, instead of a second 128-wide tile that is 47 % padding at 196.
    // (Not with a fused BatchNorm tail: its ticket counts the workgroups of ONE launch.)
    const int nsplit = pn2_opt(PN2_OPT_NT_NSPLIT);
    if (nsplit && N > 128 && N <= 224 && epi.ticket() == nullptr) {
        int rc;
        if constexpr (ALoad::kRegs >= 8) rc = launch_nt<64, 128, 16, 2, 2, 3, 1, BNN, true>(aload, bm, P, K4, N, epi, s, 0, 128);
        else rc = launch_nt<64, 128, 16, 2, 2, 4, 1, BNN, true>(aload, bm, P, K4, N, epi, s, 0, 128);
        if (rc != PN2_OK) return rc;
        const int rem = N - 128;
        if (rem <= 32) return launch_nt<128, 32, 32, 4, 1, 2, 1, BNN, true>(aload.without_lazy(), bm, P, K4, N, epi, s, 128);
        if (rem <= 64) return launch_nt<128, 64, 32, 2, 2, 2, 1, BNN, true>(aload.without_lazy(), bm, P, K4, N, epi, s, 128);
        if constexpr (ALoad::kRegs >= 8) return launch_nt<128, 96, 16, 4, 1, 2, 1, BNN, true>(aload.without_lazy(), bm, P, K4, N, epi, s, 128);
        else return launch_nt<128, 96, 16, 4, 1, 3, 1, BNN, true>(aload.without_lazy(), bm, P, K4, N, epi, s, 128);
    }
    // 64x128 tiles, 16-deep k-steps: more, smaller workgroups per CU hide the operand stream's latency better than
    // 128x128x32 at two per CU (+10..17 %); deeper register prefetch rings (2..4 k-steps) were measured: no gain.
    // Loaders with a large in-flight register set (two or three tensors per operand row) get a 168-VGPR budget
    // (3 per CU) instead of 128 (4 per CU): no spills.
    if constexpr (ALoad::kRegs >= 8) return launch_nt<64, 128, 16, 2, 2, 3, 1, BNN, true>(aload, bm, P, K4, N, epi, s, 0, 0, job);
    else return launch_nt<64, 128, 16, 2, 2, 4, 1, BNN, true>(aload, bm, P, K4, N, epi, s, 0, 0, job);
}

// Weights whose rows cannot be read as float4 (C_in = 9, 137, ...; column slices of a wider matrix) take guarded
// scalar loads for the weight tile; three tile shapes cover them (these are the first layers: short K or short P).
template <bool BNN, class ALoad, class Epi>
int dispatch_nt(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, PairJob *job = nullptr) {
    if (bm.vec) return dispatch_nt_vec<BNN>(aload, bm, P, K4, N, epi, s, job);
    if (N > 32 && pn2_cdiv(P, 64) * pn2_cdiv(N, 128) * 2 <= pn2_num_cus())
        return launch_nt<64, 64, 32, 2, 2, ALoad::kRegs >= 8 ? 2 : 3, 1, BNN, false>(aload, bm, P, K4, N, epi, s);
    if (N <= 32) return launch_nt<128, 32, 32, 4, 1, 2, 1, BNN, false>(aload, bm, P, K4, N, epi, s);
    if (N <= 64) return launch_nt<128, 64, 32, 2, 2, 2, 1, BNN, false>(aload, bm, P, K4, N, epi, s);
    return launch_nt<64, 128, 16, 2, 2, 3, 1, BNN, false>(aload, bm, P, K4, N, epi, s);
}

// ----------------------------------------------------------------------------- TN GEMM (wgrad)
// dW[M,N] += sum_p dY[p,m] * X[p,n]: both operands arrive position-major and are consumed
// "down the columns" (ds_read_b32, consecutive lanes on consecutive channels: conflict free).
// Split over P across gridDim.z; each workgroup pipelines its position steps exactly like the NT
// core (register prefetch under the MFMAs, double-buffered LDS, one barrier per step) and adds its
// partial tile with fp32 atomics (256 contiguous bytes per wave-instruction).

// KS > 1 (narrow products, M and N <= 64): the tile has fewer than four 32x32 wave tiles, so KS waves share one and
// take every KS-th pair of positions of a stage; each adds its own partial tile (a 32x32 tile is 1024 atomics).
// TD: register ring of prefetched stages (few-row products run ONE short chain of stages per workgroup, one or two workgroups per
// CU: with a single stage in flight that chain runs at global-load latency, see dispatch_nt_vec).
// ACCS: independent partial accumulators per tile (consecutive position pairs rotate over them), see gemm_nt_kernel.
template <int BM, int BN, int WG_BP, int WR, int WC, int MINB, class DyLoad, class XLoad, int KS = 1, int TD = 1, int ACCS = 1>
__device__ __forceinline__ void gemm_tn_body(DyLoad dyload, XLoad xload, int64_t P, int64_t chunk,
                                                              int M, int N, float *__restrict__ dW, int lddw,
                                                              float *__restrict__ dbias, const Bid3 bid) {
    constexpr int WTM = BM / WR, WTN = BN / WC;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_IT = WG_BP * (BM / 4) / NTHREADS, B_IT = WG_BP * (BN / 4) / NTHREADS;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");
    static_assert(WR * WC * KS == 4 && (WG_BP / 2) % KS == 0, "four waves");
    __shared__ __attribute__((aligned(16))) float As[2][WG_BP * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][WG_BP * LDB];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ks = wave / (WR * WC), wt = wave % (WR * WC);
    const int wr = wt / WC, wc = wt % WC;
    const int l31 = lane & 31, lh = lane >> 5;
    const int m0 = bid.x * BM, n0 = bid.y * BN;
    dyload.prologue();                                // consumer-side BatchNorm backward (a layer without a data gradient)
    const int64_t p_begin = (int64_t)bid.z * chunk;
    const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;
    const int arow = t / (BM / 4), acq = (t % (BM / 4)) * 4;      // loader coordinates (fixed per thread)
    const int brow = t / (BN / 4), bcq = (t % (BN / 4)) * 4;
    constexpr int AR = NTHREADS / (BM / 4), BR = NTHREADS / (BN / 4);

    static_assert(ACCS == 1 || ACCS == 2 || ACCS == 4, "partial accumulators");
    static_assert((WG_BP / 2 / KS) % ACCS == 0, "position pairs of a stage rotate evenly over the partial accumulators");
    f32x16 acc[TM][TN], accp[ACCS > 1 ? ACCS - 1 : 1][TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
#pragma unroll
                for (int u = 0; u < ACCS - 1; ++u) accp[u][i][j][r] = 0.f;
            }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    typename DyLoad::template Raw<A_IT> ra[TD];      // raw operands of the next TD stages (see the loader comment)
    typename XLoad::template Raw<B_IT> rb[TD];
    auto fetch = [&](int d, int64_t p0) {
        dyload.template issue<A_IT>(ra[d], p0 + arow, AR, m0 + acq, p_end, m0 + acq < M);
        xload.template issue<B_IT>(rb[d], p0 + brow, BR, n0 + bcq, p_end, n0 + bcq < N);
    };

    // a thread's channel quads are fixed for the whole launch: the per-channel constants are fetched once
    const typename DyLoad::Params dp = dyload.params(m0 + acq, m0 + acq < M);
    const typename XLoad::Params xp = xload.params(n0 + bcq, n0 + bcq < N);
#pragma unroll
    for (int d = 0; d < TD; ++d)
        if (p_begin + (int64_t)d * WG_BP < p_end) fetch(d, p_begin + (int64_t)d * WG_BP);
    int buf = 0;
    for (int64_t p0 = p_begin; p0 < p_end;) {
#pragma unroll
        for (int d = 0; d < TD; ++d) {
            if (p0 >= p_end) break;
            float *Ab = As[buf], *Bb = Bs[buf];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const float4 v = dyload.template finish<A_IT>(ra[d], i, (p0 + arow + i * AR < p_end) && (m0 + acq < M), dp);
                *reinterpret_cast<float4 *>(&Ab[(arow + i * AR) * LDA + acq]) = v;
                bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                *reinterpret_cast<float4 *>(&Bb[(brow + i * BR) * LDB + bcq]) =
                    xload.template finish<B_IT>(rb[d], i, (p0 + brow + i * BR < p_end) && (n0 + bcq < N), xp);
            if (p0 + (int64_t)TD * WG_BP < p_end) fetch(d, p0 + (int64_t)TD * WG_BP);
            __syncthreads();
#pragma unroll
            for (int kq = 0; kq < WG_BP / 2 / KS; ++kq) {
                const int kk = kq * KS + ks;
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ab[(kk * 2 + lh) * LDA + wr * WTM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bb[(kk * 2 + lh) * LDB + wc * WTN + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int u = kq % ACCS;                  // static after unrolling
                        if (u == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                        else accp[ACCS > 1 ? u - 1 : 0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], accp[ACCS > 1 ? u - 1 : 0][i][j], 0, 0, 0);
                    }
            }
            buf ^= 1;
            p0 += WG_BP;
        }
    }
    if (ACCS > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
#pragma unroll
                    for (int u = 0; u < ACCS - 1; ++u) v += accp[u][i][j][r];
                    acc[i][j][r] = v;
                }
    }

    if constexpr (KS > 1) {
        // fold the KS partial tiles in LDS first: dW is tiny here (<= 64x32) and every workgroup adds to the same
        // few cache lines -- KS times the atomics per address cost more than the split gained
        static_assert((KS - 1) * WR * WC * TM * TN * 16 * 64 <= 2 * WG_BP * LDA, "fold buffer");
        __syncthreads();
        float *fold = As[0];
        if (ks > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        fold[((((ks - 1) * (WR * WC) + wt) * TM * TN + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (ks > 0) goto tn_bias;
#pragma unroll
        for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += fold[(((q * (WR * WC) + wt) * TM * TN + i * TN + j) * 16 + r) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * WTN + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < N) atomicAdd(dW + (int64_t)m * lddw + n, acc[i][j][r]);
            }
        }
tn_bias:
    if (dbias != nullptr && bid.y == 0) {      // combine the AR row-threads of each column group in LDS: one atomic per channel
        __syncthreads();
        float *sh = As[0];
        *reinterpret_cast<float4 *>(&sh[t * 4]) = bsum;
        __syncthreads();
        if (arow == 0) {
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < AR; ++r) {
                const float4 v = *reinterpret_cast<const float4 *>(&sh[(r * (BM / 4) + t) * 4]);
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            if (m0 + acq < M) atomicAdd(dbias + m0 + acq, tot.x);
            if (m0 + acq + 1 < M) atomicAdd(dbias + m0 + acq + 1, tot.y);
            if (m0 + acq + 2 < M) atomicAdd(dbias + m0 + acq + 2, tot.z);
            if (m0 + acq + 3 < M) atomicAdd(dbias + m0 + acq + 3, tot.w);
        }
    }
}

template <int BM, int BN, int WG_BP, int WR, int WC, int MINB, class DyLoad, class XLoad, int KS = 1, int TD = 1, int ACCS = 1>
__global__ __launch_bounds__(NTHREADS, MINB) void gemm_tn_kernel(DyLoad dyload, XLoad xload, int64_t P, int64_t chunk,
                                                              int M, int N, float *__restrict__ dW, int lddw,
                                                              float *__restrict__ dbias) {
    gemm_tn_body<BM, BN, WG_BP, WR, WC, MINB, DyLoad, XLoad, KS, TD, ACCS>(dyload, xload, P, chunk, M, N, dW, lddw, dbias,
                                                                           Bid3{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z});
}

// Weight gradient of a FIRST layer, dW[M, N <= 16] += sum_p dY[p, m] * X[p, n] (X = the grouped input rows, 3+D = 9..12
// columns).  Two flops per loaded byte: nothing for the matrix cores to do, the job is streaming dZ and Y once at
// HBM speed.  A lane owns one output channel m (the dZ / Y rows are read as 256-byte segments), the X row of a
// position is one broadcast request per row-slot, and each thread keeps its N accumulators in registers over four
// independent rows per trip.  The 128x32-tile MFMA kernel runs these shapes at 1.1-3.2 TB/s (three quarters of
// its loader threads idle on M = 32..64); this one is bounded by the memory system.
template <int NQ>
__global__ __launch_bounds__(256, 2) void wgrad_skinny_kernel(const float *__restrict__ dZ, int ldz,
                                                              const float *__restrict__ Y, int ldy,
                                                              const float *coef, int ldc,   // (no __restrict__: the prologue writes it)
                                                              const float *__restrict__ X, int ldx, int64_t P, int M, int N,
                                                              int cg_log2, float *__restrict__ dW, int lddw,
                                                              float *__restrict__ dbias, LazyCoef lc) {
    constexpr int NA = NQ * 4;                                // accumulators per channel (+1 for the bias sum)
    __shared__ float red[4 * 64 * 4 * (NA + 1)];              // [wave][column group (<= 64)][4 channels][NA + 1]
    lazy_coef_prologue(lc);                                   // consumer-side BatchNorm backward (bn_tail.h)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int CG = 1 << cg_log2, RS = 256 >> cg_log2;         // float4 column groups per row; rows per workgroup pass
    const int cq = t & (CG - 1), slot = t >> cg_log2;
    const int m = (blockIdx.y * CG + cq) * 4;                 // first of this lane's four output channels
    const bool live = m < M;                                  // (pitches are multiples of 4: a live float4 is in-row)
    const float *zp = reinterpret_cast<const float *>(pn2_zero_page);
    const float4 c0 = ld4(live ? coef + m : zp), q1 = ld4(live ? coef + ldc + m : zp),
                 q0 = ld4(live ? coef + 2 * ldc + m : zp), mu = ld4(live ? coef + 3 * ldc + m : zp);
    const DyParams prm{c0, q1, q0, mu};
    float acc[4][NA + 1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e <= NA; ++e) acc[c][e] = 0.f;
    const int64_t stride = (int64_t)gridDim.x * RS;
    for (int64_t p0 = (int64_t)blockIdx.x * RS + slot; p0 < P; p0 += 4 * stride) {
        float4 dz[4], yv[4], x[4][NQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                         // 4 independent rows: (2 + NQ) x 16 B per lane each
            const int64_t p = p0 + u * stride;
            const bool v = p < P && live;
            dz[u] = ld4(v ? dZ + p * ldz + m : zp);
            yv[u] = ld4(v ? Y + p * ldy + m : zp);
#pragma unroll
            for (int q = 0; q < NQ; ++q) x[u][q] = ld4(v ? X + p * ldx + q * 4 : zp);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool v = p0 + u * stride < P && live;
            const float4 d4 = v ? dy_from(dz[u], yv[u], prm) : kZero4;
            const float dy[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c][NA] += dy[c];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[c][q * 4 + 0] = __builtin_fmaf(dy[c], x[u][q].x, acc[c][q * 4 + 0]);
                    acc[c][q * 4 + 1] = __builtin_fmaf(dy[c], x[u][q].y, acc[c][q * 4 + 1]);
                    acc[c][q * 4 + 2] = __builtin_fmaf(dy[c], x[u][q].z, acc[c][q * 4 + 2]);
                    acc[c][q * 4 + 3] = __builtin_fmaf(dy[c], x[u][q].w, acc[c][q * 4 + 3]);
                }
            }
        }
    }
    // fold the row slots: inside a wave with xor-shuffles (lanes cq, cq + CG, ...), across the four waves through LDS
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e <= NA; ++e) {
            float v = acc[c][e];
            for (int off = CG; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[c][e] = v;
        }
    const int lanes_cg = CG < 64 ? CG : 64;                   // distinct column groups inside one wave
    if (lane < lanes_cg) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e <= NA; ++e) red[((wave * 64 + lane) * 4 + c) * (NA + 1) + e] = acc[c][e];
    }
    __syncthreads();
    // CG <= 64: every wave holds all column groups -> add the 4 waves; CG > 64 cannot happen (M <= 256 per grid.y slab)
    for (int i = t; i < lanes_cg * 4 * (NA + 1); i += 256) {
        const int e = i % (NA + 1), c = (i / (NA + 1)) & 3, g = i / (4 * (NA + 1));
        const int mm = (blockIdx.y * CG + g) * 4 + c;
        if (mm >= M) continue;
        float sum = 0.f;
        for (int w = 0; w < 4; ++w) sum += red[((w * 64 + g) * 4 + c) * (NA + 1) + e];
        if (e < N) atomicAdd(dW + (int64_t)mm * lddw + e, sum);
        else if (e == NA && dbias != nullptr) atomicAdd(dbias + mm, sum);
    }
}

template <int NQ>
int launch_skinny(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, int ldc, const float *X, int ldx,
                  int64_t P, int M, int N, float *dW, int lddw, float *dbias, hipStream_t s, LazyCoef lc) {
    int cg_log2 = 3;                                          // 8 column groups = 32 channels at least
    while ((4 << cg_log2) < M && cg_log2 < 6) ++cg_log2;      // up to 64 groups = 256 channels per grid.y slab
    const int CG = 1 << cg_log2, RS = 256 >> cg_log2;
    const unsigned gy = (unsigned)pn2_cdiv(M, 4 * CG);
    int64_t gx = pn2_cdiv(P, (int64_t)RS * 16);
    const int64_t cap = (int64_t)pn2_num_cus() * 2 / gy;      // two resident workgroups per CU; each ends with M*N atomics
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    PN2_NOTE_KERNEL(wgrad_skinny_kernel<NQ>);
    hipLaunchKernelGGL((wgrad_skinny_kernel<NQ>), dim3((unsigned)gx, gy), dim3(256), 0, s, dZ, ldz, Y, ldy, coef, ldc, X, ldx, P, M, N,
                       cg_log2, dW, lddw, dbias, lc);
    return pn2_launch_status();
}

template <int BM, int BN, int WG_BP, int WR, int WC, int MINB, int KS = 1, int TD = 1, int ACCS = 1, class DyLoad, class XLoad>
int launch_tn(DyLoad dyload, XLoad xload, int64_t P, int M, int N, float *dW, int lddw, float *dbias, hipStream_t s,
              int per_cu = MINB) {
    unsigned tm = (unsigned)pn2_cdiv(M, BM), tn = (unsigned)pn2_cdiv(N, BN);
    const int splitdiv = pn2_opt(PN2_OPT_TN_SPLITDIV);
    int64_t want = (int64_t)pn2_num_cus() * per_cu / ((int64_t)tm * tn) / splitdiv;   // resident workgroups per CU
    if (want < 1) want = 1;
    int64_t max_split = pn2_cdiv(P, 8 * WG_BP);
    int64_t split = want < max_split ? want : max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    int64_t chunk = pn2_cdiv(pn2_cdiv(P, split), WG_BP) * WG_BP;
    split = pn2_cdiv(P, chunk);
    PN2_NOTE_KERNEL(gemm_tn_kernel<BM, BN, WG_BP, WR, WC, MINB, DyLoad, XLoad, KS, TD, ACCS>);
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WG_BP, WR, WC, MINB, DyLoad, XLoad, KS, TD, ACCS>), dim3(tm, tn, (unsigned)split), dim3(NTHREADS), 0, s,
                       dyload, xload, P, chunk, M, N, dW, lddw, dbias);
    return pn2_launch_status();
}

template <int BM, int BN, int BK, int WR, int WC, bool VEC, class ALoad, class Epi, class XLoad>
__global__ __launch_bounds__(NTHREADS, 2) void bwd_pair_kernel(ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, int nt_gx, int nt_gy,
                                                              XLoad xload, int64_t chunk, int tn_M, int tn_N, float *__restrict__ dW,
                                                              int lddw, int tn_gx, int tn_gy, int tn_gz) {
    const int b = (int)blockIdx.x, nt_blocks = nt_gx * nt_gy;
    if (b < nt_blocks) {                                   // (workgroup-uniform: the two bodies never meet)
        gemm_nt_body<BM, BN, BK, WR, WC, 2, 1, true, VEC, ALoad, Epi, 1, false>(aload, bm, P, K4, N, epi, 0,
                                                                              Bid3{b % nt_gx, b / nt_gx, 0, nt_gx, nt_gy, 1});
    } else {
        const int c = b - nt_blocks;
        gemm_tn_body<64, 64, 32, 2, 2, 2, ALoad, XLoad, 1, 2, 4>(aload, xload, P, chunk, tn_M, tn_N, dW, lddw, nullptr,
                                                                 Bid3{c % tn_gx, (c / tn_gx) % tn_gy, c / (tn_gx * tn_gy), tn_gx, tn_gy, tn_gz});
    }
}

template <int BM, int BN, int BK, int WR, int WC, bool VEC, class ALoad, class Epi>
bool launch_bwd_pair(const PairJob &j, ALoad aload, BMat bm, int64_t P, int K4, int N, Epi epi, hipStream_t s, unsigned nt_gx, unsigned nt_gy, int *rc) {
    constexpr bool masked = pn2_same<Epi, EpiDgradMask>::v;
    if (masked != (j.x_aff != nullptr)) return false;      // the weight gradient's X loader follows the data gradient's epilogue
    // the split over P of launch_tn<64, 64, 32, 2, 2, 2, 1, 2, 4> (the few-row weight-gradient configuration)
    constexpr int WG_BP = 32;
    const unsigned tm = (unsigned)pn2_cdiv(j.M, 64), tn = (unsigned)pn2_cdiv(j.N, 64);
    const int splitdiv = pn2_opt(PN2_OPT_TN_SPLITDIV);
    int64_t want = (int64_t)pn2_num_cus() / ((int64_t)tm * tn) / splitdiv;          // one TN workgroup per CU beside one NT workgroup
    if (want < 1) want = 1;
    const int64_t max_split = pn2_cdiv(P, 8 * WG_BP);
    int64_t split = want < max_split ? want : max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    const int64_t chunk = pn2_cdiv(pn2_cdiv(P, split), WG_BP) * WG_BP;
    split = pn2_cdiv(P, chunk);
    const unsigned total = nt_gx * nt_gy + tm * tn * (unsigned)split;
    const size_t dyn = (size_t)ALoad::kTab * K4 * sizeof(float);
    if constexpr (masked) {
        const LoadBnReluFixed xl{j.X, j.ldx, j.x_aff, zero_page_dev()};
        PN2_NOTE_KERNEL(bwd_pair_kernel<BM, BN, BK, WR, WC, VEC, ALoad, Epi, LoadBnReluFixed>);
        hipLaunchKernelGGL((bwd_pair_kernel<BM, BN, BK, WR, WC, VEC, ALoad, Epi, LoadBnReluFixed>), dim3(total), dim3(NTHREADS), dyn, s, aload, bm, P,
                           K4, N, epi, (int)nt_gx, (int)nt_gy, xl, chunk, j.M, j.N, j.dW, j.lddw, (int)tm, (int)tn, (int)split);
    } else {
        const LoadPlain xl{j.X, j.ldx, zero_page_dev()};
        PN2_NOTE_KERNEL(bwd_pair_kernel<BM, BN, BK, WR, WC, VEC, ALoad, Epi, LoadPlain>);
        hipLaunchKernelGGL((bwd_pair_kernel<BM, BN, BK, WR, WC, VEC, ALoad, Epi, LoadPlain>), dim3(total), dim3(NTHREADS), dyn, s, aload, bm, P, K4,
                           N, epi, (int)nt_gx, (int)nt_gy, xl, chunk, j.M, j.N, j.dW, j.lddw, (int)tm, (int)tn, (int)split);
    }
    *rc = pn2_launch_status();
    return true;
}

template <class DyLoad, class XLoad>
int dispatch_tn(DyLoad dyload, XLoad xload, int64_t P, int M, int N, float *dW, int lddw, float *dbias, hipStream_t s) {
    const int cfg = pn2_opt(PN2_OPT_TN_CFG);     // tuning hook (tools/bench_kernels.py)
    // narrow products (the 32-channel layers of sa1): tiles of one or two 32x32 wave tiles, the four waves split the
    // positions of a stage between them (KS) -- no loader thread idles on zero-page columns as in the 128x32 tile:
    // 524 288 x 32 x 32: 78 -> 43 us, 524 288 x 64 x 32 (pooled): 93 -> 57 us.  Every workgroup ends with atomics on
    // the same <= 2048 words, so the 32x32 case asks for two workgroups per CU, not four.
    const int narrow = pn2_opt(PN2_OPT_TN_NARROW);
    constexpr bool heavy_dy = DyLoad::kRegs >= 13;
    if (narrow && N <= 32 && M <= 32)
        return launch_tn<32, 32, 64, 1, 1, heavy_dy ? 3 : 4, 4>(dyload, xload, P, M, N, dW, lddw, dbias, s, 2);
    if (narrow && N <= 32 && M <= 64) return launch_tn<64, 32, 32, 2, 1, heavy_dy ? 3 : 4, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (N <= 32) return launch_tn<128, 32, 32, 4, 1, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (M <= 32) return launch_tn<32, 128, 32, 1, 4, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (M <= 64 && N <= 64) return launch_tn<64, 64, 32, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    const bool narrow_n = N <= 64 || (N > 128 && N < 256 && N % 128 != 0 && N % 128 <= 64);
    // measured (tools/bench_kernels.py wgrad): 16-position stages at 3-4 workgroups per CU beat 32 at 2 by
    // 10-30 % on the long reductions; short ones (P < 128 k) prefer fewer, fatter workgroups.  The pooled dY
    // loader keeps three tensors per row in flight and gets one workgroup per CU less (register budget).
    constexpr bool heavy = DyLoad::kRegs >= 13;
    // few rows (sa3 / fp3 / fp2 stages): the split over P is short, so small tiles -- four times fewer atomics per
    // multiply than 128x128, and enough tiles to fill the chip without a deep split (small-P wgrad 350 -> 259 us/step)
    // (the same holds up to P = 65 536 -- 65 536 x 128 x 128: 52.6 -> 35.5 us, 32 768 x 256 x 320: 90 -> 81 us -- and at
    // P = 131 072 for a 128 x 128 product, 72 -> 65 us, but not for 256 x 128, 118 -> 140 us)
    const int small_p = pn2_opt(PN2_OPT_TN_SMALLP);
    if ((P <= small_p || (P <= 2 * (int64_t)small_p && (int64_t)M * N <= 16384)) && cfg != 3) {
        const int tdepth = pn2_opt(PN2_OPT_TN_SMALL_DEPTH);     // stages in flight (latency-bound chains: see TD)
        if (tdepth >= 3) return launch_tn<64, 64, 32, 2, 2, 2, 1, 2, 1>(dyload, xload, P, M, N, dW, lddw, dbias, s);
        if (tdepth == 2) return launch_tn<64, 64, 32, 2, 2, 2, 1, 2, 4>(dyload, xload, P, M, N, dW, lddw, dbias, s);
        return launch_tn<64, 64, 32, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    }
    if (cfg == 1 || P < 131072) {
        if (narrow_n) return launch_tn<128, 64, 32, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
        if (M <= 64) return launch_tn<64, 128, 32, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
        return launch_tn<128, 128, 16, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    }
    // 128x64 at four workgroups per CU spills six VGPRs into the stage loop (128-register budget): three is faster
    // (wgrad 1M x 96 x 64: 291 -> 240 us)
    if (narrow_n) return launch_tn<128, 64, 16, 2, 2, 3>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    if (M <= 64) return launch_tn<64, 128, 16, 2, 2, heavy ? 3 : 4>(dyload, xload, P, M, N, dW, lddw, dbias, s);
    return launch_tn<128, 128, 16, 2, 2, 2>(dyload, xload, P, M, N, dW, lddw, dbias, s);   // three per CU spills (dense loader)
}

// ----------------------------------------------------------------------------- small kernels

__global__ void bn_finalize_kernel(const double *__restrict__ stats, double inv_p, double unbias, int C, int ld,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float momentum, int training, float *__restrict__ rmean, float *__restrict__ rvar,
                                   int64_t *__restrict__ nbt, float *__restrict__ affine) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt) *nbt += 1;
    if (c >= C) return;
    if (training) {
        double s0 = 0.0, s1 = 0.0;
        for (int r = 0; r < PN2_STAT_REPLICAS; ++r) { s0 += stats[r * 2 * C + c]; s1 += stats[r * 2 * C + C + c]; }
        bn_finalize_channel(s0, s1, c, ld, inv_p, unbias, gamma, beta, eps, momentum, rmean, rvar, affine);
        return;
    }
    const double invstd = 1.0 / sqrt((double)rvar[c] + (double)eps);
    affine[c] = rmean[c];
    affine[ld + c] = (float)((double)gamma[c] * invstd);
    affine[2 * ld + c] = beta[c];
    affine[3 * ld + c] = (float)invstd;
}

// out[g,c] = max_k relu(bn(Y[g*K+k,c])); arg = first k attaining it.
// One wave per group.  A lane owns one float4 column group; LPR (a power of two) lanes cover a row, so one
// wave-wide load instruction fetches 64/LPR consecutive rows of 16 B per lane (1 KiB per instruction, whole
// 64 B segments per row).  Each lane keeps the running (max, first k) of its rows -- it visits k in ascending
// order, so a strict `>` keeps the first -- and the 64/LPR row-lanes of a column are folded with xor-shuffles
// (greater value, or equal value and smaller k).  K == 1 (FeaturePropagation outputs: BN + ReLU only) maps
// the row-lanes to consecutive groups instead.
// ksplit != 0 (few groups, e.g. the 16 group_all rows of sa3 or the 2048 groups of sa2): the four waves of a workgroup
// share ONE group, each takes a quarter of its K rows, and the partial (max, first k) pairs meet in LDS.
template <bool kPooled>
__global__ __launch_bounds__(256) void bn_relu_max_kernel(const float *__restrict__ Y, int ldy,
                                                          const float *aff, int lda, int64_t G, int K,   // (no __restrict__: the prologue writes it)
                                                          int lpr_log2, int ksplit, float *__restrict__ out, int ldo,
                                                          int32_t *__restrict__ arg, LazyBn lz) {
    __shared__ float sh_v[4][64][4];
    __shared__ int sh_k[4][64][4];
    // Round 4: the grid is bounded (a few workgroups per CU, each walking its groups with a stride) so that the consumer-side
    // BatchNorm prologue -- every workgroup turns the producer's sums into the affine block before it reads it -- is paid a few
    // hundred times per launch, not once per group.
    lazy_bn_prologue(lz);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int LPR = 1 << lpr_log2, RPW = 64 >> lpr_log2;
    const int cq = (blockIdx.x * LPR + (lane & (LPR - 1))) * 4;      // first of this lane's four channels
    const int rsub = lane >> lpr_log2;
    const bool colv = cq < lda;
    Affine a(aff, lda);
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), sc = mu, be = mu;
    if (colv) { mu = ld4(a.mean + cq); sc = ld4(a.scale + cq); be = ld4(a.beta + cq); }
    const int64_t wstep = ksplit ? (int64_t)gridDim.y : (int64_t)gridDim.y * 4;
    if (!kPooled) {                               // K == 1: RPW groups per wave, no reduction
        const int64_t waves = (G + RPW - 1) / RPW;
        for (int64_t w = (int64_t)blockIdx.y * 4 + wv; w < waves; w += wstep) {
            const int64_t g = w * RPW + rsub;
            if (!colv || g >= G) continue;
            const float4 y = ld4(Y + g * ldy + cq);
            float4 o;
            o.x = fmaxf(bn_act(y.x, mu.x, sc.x, be.x), 0.f);
            o.y = fmaxf(bn_act(y.y, mu.y, sc.y, be.y), 0.f);
            o.z = fmaxf(bn_act(y.z, mu.z, sc.z, be.z), 0.f);
            o.w = fmaxf(bn_act(y.w, mu.w, sc.w, be.w), 0.f);
            *reinterpret_cast<float4 *>(out + g * ldo + cq) = o;
            if (arg) *reinterpret_cast<int4 *>(arg + g * ldo + cq) = make_int4(0, 0, 0, 0);
        }
        return;
    }
    const int kq = ksplit ? (K + 3) >> 2 : K;     // rows of this wave: [k_lo, k_hi)
    const int k_lo = ksplit ? wv * kq : 0, k_hi = k_lo + kq < K ? k_lo + kq : K;
    // (ksplit: w is uniform over the workgroup, so the barriers inside the loop are met by all four waves)
    for (int64_t w = ksplit ? (int64_t)blockIdx.y : (int64_t)blockIdx.y * 4 + wv; w < G; w += wstep) {
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bk[4] = {0, 0, 0, 0};
        if (colv) {
            const float *y = Y + w * K * ldy + cq;
#pragma unroll 8
            for (int k = k_lo + rsub; k < k_hi; k += RPW) {
                const float4 v = ld4(y + (int64_t)k * ldy);
                const float o[4] = {fmaxf(bn_act(v.x, mu.x, sc.x, be.x), 0.f), fmaxf(bn_act(v.y, mu.y, sc.y, be.y), 0.f),
                                    fmaxf(bn_act(v.z, mu.z, sc.z, be.z), 0.f), fmaxf(bn_act(v.w, mu.w, sc.w, be.w), 0.f)};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (o[e] > best[e]) { best[e] = o[e]; bk[e] = k; }
            }
        }
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ov = __shfl_xor(best[e], off, 64);
                const int ok = __shfl_xor(bk[e], off, 64);
                if (ov > best[e] || (ov == best[e] && ok < bk[e])) { best[e] = ov; bk[e] = ok; }
            }
        }
        if (ksplit) {                             // quarters are in ascending k: "greater, or equal and smaller k" again
            if (rsub == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sh_v[wv][lane][e] = best[e]; sh_k[wv][lane][e] = bk[e]; }
            }
            __syncthreads();
            if (wv == 0) {
#pragma unroll
                for (int q = 1; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ov = sh_v[q][lane & (LPR - 1)][e];
                        const int ok = sh_k[q][lane & (LPR - 1)][e];
                        if (ov > best[e] || (ov == best[e] && ok < bk[e])) { best[e] = ov; bk[e] = ok; }
                    }
            }
            __syncthreads();                      // the slots are free for the next group
        }
        if (colv && rsub == 0 && (!ksplit || wv == 0)) {
            *reinterpret_cast<float4 *>(out + w * ldo + cq) = make_float4(best[0], best[1], best[2], best[3]);
            if (arg) *reinterpret_cast<int4 *>(arg + w * ldo + cq) = make_int4(bk[0], bk[1], bk[2], bk[3]);
        }
    }
}

// red[c] += sum_g dZ, red[C+c] += sum_g dZ*yhat at the pooled positions.
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float *__restrict__ dOut, int ldg, int ldo,
                                                              const float *__restrict__ out,
                                                              const int32_t *__restrict__ arg,
                                                              const float *__restrict__ Y, int ldy,
                                                              const float *__restrict__ aff, int lda, int64_t G, int K,
                                                              int C, float *__restrict__ dZp, double *__restrict__ red,
                                                              CoefTail ct) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < lda) {                                   // pad columns included: dZp's pad lanes must be zero
        Affine a(aff, lda);
        const bool real = c < C;
        const float mu = real ? a.mean[c] : 0.f, is = real ? a.invstd[c] : 0.f;
        // The normalised pre-BN value at the maximum, (y* - mean) * invstd, is recovered from the pooled OUTPUT where that is well
        // conditioned: out = fma(y* - mean, gamma * invstd, beta) > 0, so x-hat = (out - beta) / gamma with an error of
        // eps * |out| / |gamma| -- taken when |gamma| >= (1 + |beta|) / 4 (a fresh BatchNorm has gamma = 1, beta = 0; 3e-7 at worst).
        // That drops the dependent gather Y[(g K + arg) ld + c]: one 64-byte sector per element (sixteen times the bytes) and
        // a second memory round trip per trip -- 346 MB per MSG-SemSeg step by the PMC counters.  Channels with a small gamma
        // keep the gather.
        const float be = real ? a.beta[c] : 0.f, ga = real && is != 0.f ? a.scale[c] / is : 0.f;
        const bool from_out = fabsf(ga) >= 0.25f * (1.f + fabsf(be));
        const float rga = from_out ? 1.f / ga : 0.f;
        // four independent groups per trip: the out / dOut / arg requests of all four go out together and the
        // dependent Y[arg] gathers follow together -- two memory round trips per four groups instead of three per group
        const int64_t stride = (int64_t)gridDim.y * 4;
        for (int64_t g0 = (int64_t)blockIdx.y * 4 + gl; g0 < G; g0 += 4 * stride) {
            float o[4], dz[4], y[4];
            int a4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * stride;
                const bool v = g < G && real;
                o[u] = v ? out[g * ldo + c] : 0.f;
                dz[u] = v ? dOut[g * ldg + c] : 0.f;
                a4[u] = v ? arg[g * ldo + c] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * stride;
                y[u] = (!from_out && g < G && o[u] > 0.f) ? Y[(g * K + a4[u]) * ldy + c] : mu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * stride;
                if (g >= G) break;
                const bool on = o[u] > 0.f;
                dZp[g * ldo + c] = on ? dz[u] : 0.f;
                if (on) {
                    s0 += (double)dz[u];
                    s1 += (double)(dz[u] * (from_out ? (o[u] - be) * rga : (y[u] - mu) * is));
                }
            }
        }
    }
    sh[0][gl][cl] = s0; sh[1][gl][cl] = s1;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        double a1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        double *rep = red + (size_t)(blockIdx.y % PN2_STAT_REPLICAS) * 2 * C;
        atomicAdd(rep + c, a0);
        atomicAdd(rep + C + c, a1);
    }
    if (ct.ticket != nullptr && tail_is_last_block(ct.ticket, gridDim.x * gridDim.y)) run_coef_tail(ct, red, C, 256);
}

// The same for a pooled last layer whose pre-BN output was never written (pn2_conv1x1_fwd_pool with Y = NULL): the value at the
// maximum, y*, is what the forward's epilogue recorded next to the row (rec[g, c] = {y*, row}; the minimum where the folded scale is
// negative) -- a coalesced read instead of the gather, and exact for every gamma.  A channel whose folded scale is exactly 0 routes
// its gradient to row 0 of the group (pn2_bn_pool_select), whose y the record does not hold: recomputed here from the layer's input,
// y = b[c] + W[c, :] . relu(bn(prevY[g K, :])) -- a dot product per (group, channel), on a path only a zero BatchNorm weight takes.
__global__ __launch_bounds__(256) void pool_bwd_reduce_rec_kernel(const float *__restrict__ dOut, int ldg, int ldo, const float *__restrict__ out,
                                                                  const int32_t *__restrict__ arg, const float2 *__restrict__ rec, int ldr,
                                                                  const float *__restrict__ aff, int lda, int64_t G, int K, int C,
                                                                  float *__restrict__ dZp, double *__restrict__ red, const float *__restrict__ W,
                                                                  int ldw, const float *__restrict__ bias, const float *__restrict__ prevY, int ldp,
                                                                  const float *__restrict__ prev_aff, int Ci) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < lda) {                                   // pad columns included: dZp's pad lanes must be zero
        Affine a(aff, lda);
        const bool real = c < C;
        const float mu = real ? a.mean[c] : 0.f, is = real ? a.invstd[c] : 0.f;
        const bool zero_scale = real && a.scale[c] == 0.f;
        const int64_t stride = (int64_t)gridDim.y * 4;
        for (int64_t g0 = (int64_t)blockIdx.y * 4 + gl; g0 < G; g0 += 4 * stride) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * stride;
                if (g >= G) break;
                const float o = real ? out[g * ldo + c] : 0.f;
                const float dz = real ? dOut[g * ldg + c] : 0.f;
                float y = real ? rec[g * ldr + c].x : 0.f;
                const bool on = o > 0.f;
                if (on && zero_scale) {
                    Affine pa(prev_aff, (Ci + 3) & ~3);
                    const float *xr = prevY + (g * K) * ldp;
                    float acc = bias[c];
                    for (int j = 0; j < Ci; ++j)
                        acc = __builtin_fmaf(W[(int64_t)c * ldw + j], fmaxf(bn_act(xr[j], pa.mean[j], pa.scale[j], pa.beta[j]), 0.f), acc);
                    y = acc;
                }
                dZp[g * ldo + c] = on ? dz : 0.f;
                if (on) {
                    s0 += (double)dz;
                    s1 += (double)(dz * ((y - mu) * is));
                }
            }
        }
    }
    sh[0][gl][cl] = s0; sh[1][gl][cl] = s1;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        double a1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        double *rep = red + (size_t)(blockIdx.y % PN2_STAT_REPLICAS) * 2 * C;
        atomicAdd(rep + c, a0);
        atomicAdd(rep + C + c, a1);
    }
}

// Dense last layer (FP): dZ = dOut * (out > 0), same two reductions.
__global__ __launch_bounds__(256) void relu_bwd_reduce_kernel(const float *__restrict__ dOut, int ldo,
                                                              const float *__restrict__ out,
                                                              const float *__restrict__ Y, int ldy,
                                                              const float *__restrict__ aff, int lda, int64_t P, int C,
                                                              float *__restrict__ dZ, int ldz, double *__restrict__ red,
                                                              CoefTail ct) {
    __shared__ double sh[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c >= C && c < lda) {                          // pad columns of dZ are written too (zero): nobody pre-clears it
        for (int64_t p = (int64_t)blockIdx.y * 4 + gl; p < P; p += (int64_t)gridDim.y * 4) dZ[p * ldz + c] = 0.f;
    }
    if (c < C) {
        Affine a(aff, lda);
        const float mu = a.mean[c], is = a.invstd[c];
        // x-hat = (y - mean) * invstd from the OUTPUT where that is well conditioned (see pool_bwd_reduce_kernel): out > 0 is
        // the only case that counts, and there out = fma(y - mean, gamma * invstd, beta).  Y is then not read at all: a quarter
        // of this pass's bytes.  Channels with a small gamma keep reading Y.
        const float be = a.beta[c], ga = is != 0.f ? a.scale[c] / is : 0.f;
        const bool from_out = fabsf(ga) >= 0.25f * (1.f + fabsf(be));
        const float rga = from_out ? 1.f / ga : 0.f;
        const int64_t stride = (int64_t)gridDim.y * 4;
        // Four independent rows in flight, and the NEXT trip's rows are requested before this trip's stores go out: loads and
        // stores retire through one in-order counter, so a load issued behind a store waits for that store's HBM
        // acknowledgement (round 3, found on pn2_group_conv_fwd: 105 -> 92 us there).  Here it measured neutral (65 536 x 128:
        // 24.4 us = 5.5 TB/s either way: the kernel already sits on the memory system).
        float o[4], g4[4], y[4], on_[4], gn[4], yn[4];
        auto fetch = [&](int64_t p0, float (&oo)[4], float (&gg)[4], float (&yy)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                const bool v = p < P;
                const int64_t pc = v ? p : 0;                     // (clamped: always a request, dropped below)
                oo[u] = out[pc * ldo + c];
                gg[u] = dOut[pc * ldo + c];
                yy[u] = from_out ? 0.f : Y[pc * ldy + c];
            }
        };
        int64_t p0 = (int64_t)blockIdx.y * 4 + gl;
        if (p0 < P) fetch(p0, o, g4, y);
        for (; p0 < P; p0 += 4 * stride) {
            fetch(p0 + 4 * stride, on_, gn, yn);                  // past the end: clamped requests, never used
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t p = p0 + u * stride;
                if (p >= P) break;
                const float dz = o[u] > 0.f ? g4[u] : 0.f;
                dZ[p * ldz + c] = dz;
                s0 += (double)dz;
                s1 += (double)(dz * (from_out ? (o[u] - be) * rga : (y[u] - mu) * is));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { o[u] = on_[u]; g4[u] = gn[u]; y[u] = yn[u]; }
        }
    }
    sh[0][gl][cl] = s0; sh[1][gl][cl] = s1;
    __syncthreads();
    if (gl == 0 && c < C) {
        double a0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
        double a1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
        double *rep = red + (size_t)(blockIdx.y % PN2_STAT_REPLICAS) * 2 * C;
        atomicAdd(rep + c, a0);
        atomicAdd(rep + C + c, a1);
    }
    if (ct.ticket != nullptr && tail_is_last_block(ct.ticket, gridDim.x * gridDim.y)) run_coef_tail(ct, red, C, 256);
}

__global__ void bn_bwd_coef_kernel(const double *__restrict__ red, double inv_p, int C, int ld,
                                   const float *__restrict__ gamma, const float *__restrict__ aff, int use_batch,
                                   float *__restrict__ coef, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                   int accumulate) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double r0 = 0.0, r1 = 0.0;
    for (int r = 0; r < PN2_STAT_REPLICAS; ++r) { r0 += red[r * 2 * C + c]; r1 += red[r * 2 * C + C + c]; }
    bn_coef_channel(r0, r1, c, ld, inv_p, gamma, aff, use_batch, coef, dgamma, dbeta, accumulate);
}

inline int round4(int x) { return (x + 3) & ~3; }

}  // namespace

#ifdef PN2_STAMP
extern "C" int pn2_debug_stamps(unsigned long long *host_out, int n) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pn2_stamp_buf), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -2;
}
#endif

namespace {

FinTail make_fin_tail(const pn2_bn_finalize_tail *t, int64_t P) {
    FinTail f{};
    if (t == nullptr) return f;
    f.ticket = t->ticket;
    f.gamma = t->gamma; f.beta = t->beta; f.eps = t->eps; f.momentum = t->momentum;
    f.rmean = t->running_mean; f.rvar = t->running_var; f.nbt = t->num_batches_tracked; f.affine = t->affine;
    f.inv_p = 1.0 / (double)P;
    f.unbias = P > 1 ? (double)P / (double)(P - 1) : 1.0;
    return f;
}

CoefTail make_coef_tail(const pn2_bn_coef_tail *t, int64_t P) {
    CoefTail c{};
    if (t == nullptr) return c;
    c.ticket = t->ticket;
    c.gamma = t->gamma; c.aff = t->affine; c.use_batch = t->use_batch_stats;
    c.coef = t->coef; c.dgamma = t->dgamma; c.dbeta = t->dbeta; c.accumulate = t->accumulate;
    c.inv_p = 1.0 / (double)P;
    return c;
}

bool fin_tail_ok(const pn2_bn_finalize_tail *t, const double *stats) {
    return t == nullptr || (stats && t->ticket && t->gamma && t->beta && t->affine);
}

bool coef_tail_ok(const pn2_bn_coef_tail *t, const double *red) {
    return t == nullptr || (red && t->ticket && t->gamma && t->affine && t->coef);
}

}  // namespace

int pn2_fwd_res(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                int64_t P, int K, int N, double *stats, LazyBn lz, hipStream_t s);      // mlp_res.hip
// mlp_wide.hip: register-stationary kernels for the wide layers; *rows_done = the leading rows they covered (whole tiles)
int pn2_wide_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                 int64_t P, int K, int N, double *stats, LazyBn lz, hipStream_t s, int64_t *rows_done, int Kpool = 0,
                 const float *pool_gamma = nullptr, float *pool_ws = nullptr);
int pn2_wide_dgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy,
                   const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev, const float *prev_affine,
                   float *dXout, int ldxo, double *prev_red, int64_t P, int K, int N, LazyCoef lc, hipStream_t s, int64_t *rows_done);
int pn2_wide_wgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy,
                   const float *coef, const float *X, int ldx, const float *x_affine, float *dW, int lddw, float *dbias,
                   int64_t P, int M, int N, LazyCoef lc, hipStream_t s, float *workspace = nullptr);
int64_t pn2_wide_wgrad_workspace_bytes(int64_t P, int M, int N, int pooled);

namespace {

// ----------------------------------------------------------------------------------------------- first-layer dW, closed form
// The weight gradient of a FIRST layer (C_in = N <= 15: the 3 + D grouped input columns) whose data gradient nobody needs, from
// dZ and the 48-byte input rows alone.  The BatchNorm-backward terms of dY = c0 dZ + q1 (y - mean) + q0 are linear in sums the
// forward already fixed: with s = sum_p x_p, S = sum_p x_p x_p^T and y_p = W x_p + b,
//     dW[c][j] = c0[c] sum_p dZ[p,c] x[p,j]  +  q1[c] ((W S)[c][j] + (b[c] - mean[c]) s[j])  +  q0[c] s[j]
// so Y is not read at all: 872 of the 962 MB per MSG-SemSeg step that the general skinny kernel moved for sa1's three first
// layers were dZ + Y.  Both sums are v_mfma_f32_16x16x4_f32 products straight from the loaded registers (lane = 16 k + n holds
// dZ[p0 + k][c0 + n] resp. x[p0 + k][n]: the A operand of S = X^T X is the SAME register as its B operand), column 15 of the
// row operand is a constant 1, so S[15][j] = s[j]; the S partials leave the fp32 accumulators for fp64 every 64 rows.  The
// closed-form part needs the COMPLETE moments: the workgroup that draws the last ticket adds it, once, in fp64.
typedef float cf_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CF_REPL = PN2_CF_REPL;                              // scratch replicas: 1 / 8 of the workgroups add into each

template <int NB, int U>                                          // 16-channel blocks: M = 16 NB; 4-row groups in flight per wave
__global__ __launch_bounds__(256) void wgrad_first_cf_kernel(const float *__restrict__ dZ, int ldz, const float *coef /* written by the prologue */, int ldc,
                                                             const float *__restrict__ X, int ldx, int64_t P, int N,
                                                             const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                             float *__restrict__ part, double *__restrict__ mom, unsigned *__restrict__ ticket,
                                                             float *__restrict__ dW, int lddw, LazyCoef lc) {
    constexpr int M = 16 * NB;
    __shared__ float fold[4][NB][256];                            // [wave][block][lane * 4 + r]
    __shared__ double sfold[4][256];
    lazy_coef_prologue(lc);                                       // consumer-side BatchNorm backward (bn_tail.h)
    // dZ == NULL (round 6): `part` already holds sum_p dZ[p, c] x[p, j] -- the NEXT layer's fused backward added it straight from
    // its dX tiles (split_bwd_res_kernel, FUSE0: dZ never reached memory) -- and this launch only takes the input's moments
    const bool have_dz = dZ != nullptr;                           // (uniform)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, k = lane >> 4, n = lane & 15;
    const float *zp = reinterpret_cast<const float *>(pn2_zero_page);
    const bool xin = n < N;
    const float fill = n == 15 ? 1.f : 0.f;
    const int nc = xin ? n : 0;                                   // pad lanes re-read column 0 (replaced by `fill`): never past a row
    cf_f32x4 acc[NB], sacc = {0.f, 0.f, 0.f, 0.f};
    double sd[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = cf_f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t groups = (P + 3) >> 2;                          // 4 rows per MFMA
    const int64_t stride = (int64_t)gridDim.x * 4 * U;
    int rows64 = 0;
    // Whole trips (U groups of 4 rows, all inside P) address everything as a WAVE-UNIFORM base (scalar arithmetic) plus one
    // loop-invariant 32-bit lane offset per operand; only the last, ragged trip of the launch checks rows per lane.
    const unsigned lo_z = (unsigned)k * (unsigned)ldz + (unsigned)(NB * n), lo_x = (unsigned)k * (unsigned)ldx + (unsigned)nc;
    const int64_t full_groups = P >> 2;
    auto load_a = [&](const float *zr, float (&av)[NB]) {
        if (NB == 4 || NB == 8) {
#pragma unroll
            for (int h = 0; h < NB / 4; ++h) {
                const float4 q4 = ld4(zr + 4 * h);
                av[4 * h] = q4.x; av[4 * h + 1] = q4.y; av[(4 * h + 2) % NB] = q4.z; av[(4 * h + 3) % NB] = q4.w;
            }
        } else if (NB == 2 || NB == 6) {
#pragma unroll
            for (int h = 0; h < NB / 2; ++h) {
                const float2 q2 = *reinterpret_cast<const float2 *>(zr + 2 * h);
                av[2 * h] = q2.x; av[(2 * h + 1) % NB] = q2.y;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) av[b] = zr[b];
        }
    };
    for (int64_t g0 = ((int64_t)blockIdx.x * 4 + wave) * U; g0 < groups; g0 += stride) {
        float a[U][NB], x[U];
        // the lane's NB CONSECUTIVE channels NB n .. NB n + NB - 1 of row p0 + k (one 4 NB-byte request: whole rows per
        // instruction, not 64-byte fragments of four rows); accumulator tile b then holds the channels NB i + b
        if (g0 + U <= full_groups) {
            const float *zb = dZ + (size_t)(g0 * 4) * (unsigned)ldz, *xb = X + (size_t)(g0 * 4) * (unsigned)ldx;     // uniform
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (have_dz) load_a(zb + (size_t)(4 * u) * (unsigned)ldz + lo_z, a[u]);
                const float xv = xb[(size_t)(4 * u) * (unsigned)ldx + lo_x];
                x[u] = xin ? xv : fill;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t p = (g0 + u) * 4 + k;
                const bool v = p < P;
                if (have_dz) load_a(v ? dZ + p * ldz + NB * n : zp, a[u]);     // (a dead row reads the zero page)
                const float xv = (v ? X + p * ldx + nc : zp)[0];
                x[u] = v ? (xin ? xv : fill) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (have_dz) {
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][b], x[u], acc[b], 0, 0, 0);
            }
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u], x[u], sacc, 0, 0, 0);
        }
        rows64 += 4 * U;
        if (rows64 >= 64) {                                       // the moment partials move to fp64 every 64 rows
            rows64 = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sd[r] += (double)sacc[r]; sacc[r] = 0.f; }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sd[r] += (double)sacc[r];
    // D[i][j] of a 16 x 16 tile sits in lane 16 (i / 4) + j, register i % 4: fold the four waves through LDS, then one set of
    // atomics per workgroup into ITS replica of the scratch (CF_REPL replicas: 1 / 8 of the adders per address)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) fold[wave][b][lane * 4 + r] = acc[b][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) sfold[wave][lane * 4 + r] = sd[r];
    __syncthreads();
    const int rep = blockIdx.x % CF_REPL;
    for (int e = t; e < NB * 256 && have_dz; e += 256) {
        const int b = e >> 8, q = e & 255, l2 = q >> 2, r = q & 3, i = 4 * (l2 >> 4) + r, j = l2 & 15;
        if (j >= N) continue;
        atomicAdd(part + (rep * M + NB * i + b) * 16 + j, fold[0][b][q] + fold[1][b][q] + fold[2][b][q] + fold[3][b][q]);
    }
    {
        const int q = t, l2 = q >> 2, r = q & 3, i = 4 * (l2 >> 4) + r, j = l2 & 15;
        if (j < N && (i < N || i == 15)) atomicAdd(mom + rep * 256 + i * 16 + j, sfold[0][q] + sfold[1][q] + sfold[2][q] + sfold[3][q]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's atomics are performed (tail_is_last_block)
    if (!tail_is_last_block(ticket, gridDim.x)) return;
    // ---- the last workgroup: replicas -> sums, c0 * (sum dZ x) + the closed-form terms, ONE atomic per element of dW
    double *ms = sfold[0];                                        // S[i][j] (i < N) and s[j] = S[15][j], summed over the replicas
    {
        double v = 0.0;
        for (int r2 = 0; r2 < CF_REPL; ++r2) v += ld_f64_device(mom + r2 * 256 + t);
        __syncthreads();
        ms[t] = v;
        __syncthreads();
    }
    for (int e = t; e < M * 16; e += 256) {
        const int c = e >> 4, j = e & 15;
        if (j >= N) continue;
        float a1 = 0.f;
        for (int r2 = 0; r2 < CF_REPL; ++r2) a1 += __hip_atomic_load(part + (r2 * M + c) * 16 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double sj = ms[15 * 16 + j];
        double ws = 0.0;
        for (int i = 0; i < N; ++i) ws += (double)W[(int64_t)c * ldw + i] * ms[i * 16 + j];
        const double c0c = coef[c], q1c = coef[ldc + c], q0c = coef[2 * ldc + c], muc = coef[3 * ldc + c];
        atomicAdd(dW + (int64_t)c * lddw + j, (float)(c0c * (double)a1 + q1c * (ws + ((double)bias[c] - muc) * sj) + q0c * sj));
    }
    if (t == 0) *ticket = 0;
}

}  // namespace

extern "C" {

int pn2_conv1x1_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y,
                    int ldy, int64_t P, int K, int N, double *stats, const pn2_bn_finalize_tail *fin, const pn2_bn_lazy *in_lazy,
                    pn2_stream_t stream) {
    PN2_CHECK_ARG(X && W && bias && Y && P > 0 && P < (1LL << 31) && K > 0 && N > 0 && fin_tail_ok(fin, stats));
    PN2_CHECK_ARG(ldx % 4 == 0 && ldx >= round4(K) && ldw >= K && ldy % 4 == 0 && ldy >= round4(N));
    PN2_CHECK_ARG(lazy_bn_ok(in_lazy, in_affine, K) && (in_lazy == nullptr || in_affine != nullptr));
    LazyBn lz = make_lazy_bn(in_lazy);                                  // realised by the FIRST launch below; later ones read the block
    if (fin == nullptr) {                                               // wide layer: W stays in registers (mlp_wide.hip)
        int64_t done = 0;
        const int rc = pn2_wide_fwd(X, ldx, in_affine, W, ldw, bias, Y, ldy, P, K, N, stats, lz, pn2_s(stream), &done);
        if (rc != PN2_EUNSUPPORTED) {
            if (rc != PN2_OK || done == P) return rc;
            X += done * ldx; Y += done * ldy; P -= done;                // ragged tail: the streamed kernel below
            lz = LazyBn{};
        }
    }
    if (fin == nullptr && pn2_res_supported(P, N, K) && ldy >= N) {      // narrow, long layer: W stays in LDS (mlp_res.hip)
        const int64_t P_full = P & ~(int64_t)31;                        // whole 32-row slabs there, a ragged tail below
        const int rc = pn2_fwd_res(X, ldx, in_affine, W, ldw, bias, Y, ldy, P_full, K, N, stats, lz, pn2_s(stream));
        if (rc != PN2_EUNSUPPORTED) {                                   // (unsupported: W plus eight staging buffers exceed LDS)
            if (rc != PN2_OK || P_full == P) return rc;
            X += P_full * ldx; Y += P_full * ldy; P -= P_full;
            lz = LazyBn{};
        }
    }
    const int K4 = round4(K);
    EpiFwd epi{Y, ldy, bias, stats, make_fin_tail(fin, P)};
    const BMat bm = make_bmat(W, ldw, K, K);
    if (in_affine) return dispatch_nt<false>(LoadBnRelu{X, ldx, in_affine, zero_page_dev(), lz}, bm, P, K4, N, epi, pn2_s(stream));
    return dispatch_nt<false>(LoadPlain{X, ldx, zero_page_dev()}, bm, P, K4, N, epi, pn2_s(stream));
}

int pn2_bn_finalize(const double *stats, int64_t P, int C, const float *gamma, const float *beta, float eps,
                    float momentum, int training, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                    float *affine, pn2_stream_t stream) {
    PN2_CHECK_ARG(gamma && beta && affine && C > 0 && P > 0);
    PN2_CHECK_ARG(training ? stats != nullptr : (running_mean && running_var));
    const int ld = round4(C);
    double unbias = P > 1 ? (double)P / (double)(P - 1) : 1.0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pn2_cdiv(C, 128)), dim3(128), 0, pn2_s(stream), stats, 1.0 / (double)P,
                       unbias, C, ld, gamma, beta, eps, momentum, training, running_mean, running_var,
                       num_batches_tracked, affine);
    return pn2_launch_status();
}

int pn2_bn_relu_max(const float *Y, int ldy, const float *affine, int64_t G, int K, int C, float *out, int ldo,
                    int32_t *arg, const pn2_bn_lazy *lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && affine && out && G > 0 && K > 0 && C > 0 && lazy_bn_ok(lazy, affine, C));
    const int ld = (C + 3) & ~3;
    PN2_CHECK_ARG(ldy % 4 == 0 && ldo % 4 == 0 && ldy >= ld && ldo >= ld);      // float4 rows; pad columns are written (0)
    const LazyBn lz = make_lazy_bn(lazy);
    const int cg = ld / 4;
    int lpr_log2 = 0;
    while ((1 << lpr_log2) < cg && lpr_log2 < 6) ++lpr_log2;
    const int rpw = 64 >> lpr_log2;
    const unsigned gx = (unsigned)pn2_cdiv(cg, 1 << lpr_log2);
    const int64_t waves = K == 1 ? pn2_cdiv(G, rpw) : G;
    // bounded grid: eight workgroups per CU in all (full occupancy for this light kernel), each walking its groups
    int64_t cap = (int64_t)pn2_num_cus() * 8 / gx;
    if (cap < 1) cap = 1;
    if (cap > 65535) cap = 65535;
    if (K >= 32 && G * gx <= 4096 && G <= 65535) {      // too few groups to fill the chip with one wave each: split K
        hipLaunchKernelGGL(bn_relu_max_kernel<true>, dim3(gx, (unsigned)(G < cap ? G : cap)), dim3(256), 0, pn2_s(stream), Y, ldy, affine, ld,
                           G, K, lpr_log2, 1, out, ldo, arg, lz);
        return pn2_launch_status();
    }
    int64_t gy = pn2_cdiv(waves, 4);
    if (gy > cap) gy = cap;
    if (K == 1)
        hipLaunchKernelGGL(bn_relu_max_kernel<false>, dim3(gx, (unsigned)gy), dim3(256), 0, pn2_s(stream), Y, ldy, affine, ld, G, K,
                           lpr_log2, 0, out, ldo, arg, lz);
    else
        hipLaunchKernelGGL(bn_relu_max_kernel<true>, dim3(gx, (unsigned)gy), dim3(256), 0, pn2_s(stream), Y, ldy, affine, ld, G, K,
                           lpr_log2, 0, out, ldo, arg, lz);
    return pn2_launch_status();
}

int pn2_pool_bwd_reduce_ld(const float *dOut, int ld_dout, const float *out, int ldo, const int32_t *arg, const float *Y, int ldy,
                           const float *affine, int64_t G, int K, int C, float *dZp, double *red, const pn2_bn_coef_tail *tail,
                           pn2_stream_t stream) {
    PN2_CHECK_ARG(dOut && out && arg && Y && affine && dZp && red && G > 0 && K > 0 && C > 0 && ldo >= ((C + 3) & ~3) && ld_dout >= C &&
                  coef_tail_ok(tail, red));
    int64_t gy = pn2_cdiv(G, 4 * 4);                    // one trip of four groups per thread where the grid allows
    if (gy > 1024) gy = 1024;                           // (x 8 reduction replicas: same-address queues of <= 128)
    hipLaunchKernelGGL(pool_bwd_reduce_kernel, dim3((unsigned)pn2_cdiv((C + 3) & ~3, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), dOut,
                       ld_dout, ldo, out, arg, Y, ldy, affine, (C + 3) & ~3, G, K, C, dZp, red, make_coef_tail(tail, G * K));
    return pn2_launch_status();
}

int pn2_pool_bwd_reduce_rec(const float *dOut, int ld_dout, const float *out, int ldo, const int32_t *arg, const float *pool_ws,
                            const float *affine, int64_t G, int K, int C, float *dZp, double *red, const float *W, int ldw, const float *bias,
                            const float *prev_Y, int ld_prev, const float *prev_affine, int C_in, pn2_stream_t stream) {
    PN2_CHECK_ARG(dOut && out && arg && pool_ws && affine && dZp && red && W && bias && prev_Y && prev_affine && G > 0 && K > 0 && C > 0 &&
                  C_in > 0 && ldo >= ((C + 3) & ~3) && ld_dout >= C && ldw >= C_in && ld_prev >= C_in);
    int64_t gy = pn2_cdiv(G, 4 * 4);
    if (gy > 1024) gy = 1024;
    hipLaunchKernelGGL(pool_bwd_reduce_rec_kernel, dim3((unsigned)pn2_cdiv((C + 3) & ~3, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), dOut,
                       ld_dout, ldo, out, arg, reinterpret_cast<const float2 *>(pool_ws), C, affine, (C + 3) & ~3, G, K, C, dZp, red, W, ldw, bias,
                       prev_Y, ld_prev, prev_affine, C_in);
    return pn2_launch_status();
}

int pn2_pool_bwd_reduce(const float *dOut, int ldo, const float *out, const int32_t *arg, const float *Y, int ldy,
                        const float *affine, int64_t G, int K, int C, float *dZp, double *red, const pn2_bn_coef_tail *tail,
                        pn2_stream_t stream) {
    return pn2_pool_bwd_reduce_ld(dOut, ldo, out, ldo, arg, Y, ldy, affine, G, K, C, dZp, red, tail, stream);
}

int pn2_relu_bwd_reduce(const float *dOut, int ldo, const float *out, const float *Y, int ldy, const float *affine,
                        int64_t P, int C, float *dZ, int ldz, double *red, const pn2_bn_coef_tail *tail, pn2_stream_t stream) {
    PN2_CHECK_ARG(dOut && out && Y && affine && dZ && red && P > 0 && C > 0 && coef_tail_ok(tail, red));
    int64_t gy = pn2_cdiv(P, 4 * 16);
    if (gy > 512) gy = 512;
    hipLaunchKernelGGL(relu_bwd_reduce_kernel, dim3((unsigned)pn2_cdiv(C, 64), (unsigned)gy), dim3(256), 0, pn2_s(stream), dOut,
                       ldo, out, Y, ldy, affine, (C + 3) & ~3, P, C, dZ, ldz, red, make_coef_tail(tail, P));
    return pn2_launch_status();
}

int pn2_bn_bwd_coef(const double *red, int64_t P, int C, const float *gamma, const float *affine, int use_batch_stats,
                    float *coef, float *dgamma, float *dbeta, int accumulate, pn2_stream_t stream) {
    PN2_CHECK_ARG(red && gamma && affine && coef && P > 0 && C > 0);
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((unsigned)pn2_cdiv(C, 128)), dim3(128), 0, pn2_s(stream), red, 1.0 / (double)P,
                       C, (C + 3) & ~3, gamma, affine, use_batch_stats, coef, dgamma, dbeta, accumulate);
    return pn2_launch_status();
}

// `job`: the weight-gradient half pn2_conv1x1_bwd_pair wants to ride in this launch (null for the plain entry point)
static int conv1x1_dgrad_impl(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy, const float *coef, const float *W, int ldw,
                              const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                              double *prev_red, int64_t P, int K, int N, const pn2_bn_coef_tail *prev_tail, const pn2_bn_coef_lazy *coef_lazy,
                              pn2_stream_t stream, PairJob *job) {
    PN2_CHECK_ARG(Y && coef && W && dXout && P > 0 && P < (1LL << 31) && K > 0 && N > 0 && coef_tail_ok(prev_tail, prev_red) &&
                  (prev_tail == nullptr || prev_Y != nullptr) && lazy_coef_ok(coef_lazy, coef, K));
    LazyCoef lc = make_lazy_coef(coef_lazy);                            // realised by the FIRST launch below
    const CoefTail ct = make_coef_tail(prev_tail, P);
    PN2_CHECK_ARG(dZ != nullptr || (dZp && arg && Kpool > 0 && P < (1LL << 31)));
    PN2_CHECK_ARG(ldw >= N && ldy % 4 == 0 && ldy >= round4(K) && ldxo % 4 == 0 && ldxo >= round4(N));
    const BMat bm = make_bmat(W, ldw, K, N);
    PN2_CHECK_ARG(prev_Y == nullptr || prev_affine != nullptr);
    const int K4 = round4(K), ldc = round4(K);
    hipStream_t s = pn2_s(stream);
    if (prev_tail == nullptr && prev_Y != nullptr) {                    // wide layer: W stays in registers (mlp_wide.hip)
        int64_t done = 0;
        const int rc = pn2_wide_dgrad(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, W, ldw, prev_Y, ld_prev, prev_affine, dXout, ldxo,
                                      prev_red, P, K, N, lc, s, &done);
        if (rc != PN2_EUNSUPPORTED) {
            if (rc != PN2_OK || done == P) return rc;
            lc = LazyCoef{};
            // ragged tail (whole pooling groups: the tile height divides Kpool or is a multiple of it)
            if (dZ) dZ += done * ldz;
            else { dZp += (done / Kpool) * ldo; arg += (done / Kpool) * ldo; }
            Y += done * ldy; prev_Y += done * ld_prev; dXout += done * ldxo; P -= done;
        }
    }
    if (dZ) {
        PN2_CHECK_ARG(ldz % 4 == 0 && ldz >= K4);
        LoadDyDense ld{dZ, ldz, Y, ldy, coef, ldc, zero_page_dev(), lc};
        if (prev_Y)
            return dispatch_nt<true>(ld, bm, P, K4, N,
                                     EpiDgradMask{dXout, ldxo, prev_Y, ld_prev, prev_affine, round4(N), prev_red, ct, zero_page_dev()}, s, job);
        return dispatch_nt<true>(ld, bm, P, K4, N, EpiStore{dXout, ldxo}, s, job);
    }
    PN2_CHECK_ARG(ldo % 4 == 0 && ldo >= K4);
    LoadDyPooled ld{dZp, ldo, arg, Kpool, Y, ldy, coef, ldc, zero_page_dev(), pow2_shift(Kpool), lc};
    if (prev_Y)
        return dispatch_nt<true>(ld, bm, P, K4, N,
                                 EpiDgradMask{dXout, ldxo, prev_Y, ld_prev, prev_affine, round4(N), prev_red, ct, zero_page_dev()}, s, job);
    return dispatch_nt<true>(ld, bm, P, K4, N, EpiStore{dXout, ldxo}, s, job);
}

int pn2_conv1x1_dgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy, const float *coef, const float *W, int ldw,
                      const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                      double *prev_red, int64_t P, int K, int N, const pn2_bn_coef_tail *prev_tail, const pn2_bn_coef_lazy *coef_lazy,
                      pn2_stream_t stream) {
    return conv1x1_dgrad_impl(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, W, ldw, prev_Y, ld_prev, prev_affine, dXout, ldxo, prev_red, P, K, N,
                              prev_tail, coef_lazy, stream, nullptr);
}

int64_t pn2_conv1x1_wgrad_workspace_bytes(int64_t P, int M, int N, int pooled) { return pn2_wide_wgrad_workspace_bytes(P, M, N, pooled); }

int64_t pn2_conv1x1_wgrad_cf_scratch_bytes(void) { return CF_REPL * (128 * 16 * (int64_t)sizeof(float) + 256 * (int64_t)sizeof(double)) + 16; }

int pn2_conv1x1_wgrad_cf(const float *dZ, int ldz, const float *coef, const float *X, int ldx, const float *W, int ldw, const float *bias,
                         void *scratch, float *dW, int lddw, int64_t P, int M, int N, const pn2_bn_coef_lazy *coef_lazy,
                         pn2_stream_t stream) {
    PN2_CHECK_ARG(coef && X && W && bias && scratch && dW && P > 0 && P < (1LL << 31) && M > 0 && N > 0 && lazy_coef_ok(coef_lazy, coef, M));
    if (M % 16 != 0 || M > 128 || N > 15) return PN2_EUNSUPPORTED;
    PN2_CHECK_ARG((dZ == nullptr || (ldz >= M && ldz % 4 == 0 && (reinterpret_cast<uintptr_t>(dZ) & 15) == 0)) && ldx >= N && lddw >= N && ldw >= N &&
                  (reinterpret_cast<uintptr_t>(scratch) & 15) == 0);
    const LazyCoef lc = make_lazy_coef(coef_lazy);
    double *mom = reinterpret_cast<double *>(scratch);                              // [CF_REPL][16][16]
    float *part = reinterpret_cast<float *>(mom + CF_REPL * 256);                   // [CF_REPL][M][16]
    unsigned *ticket = reinterpret_cast<unsigned *>(part + CF_REPL * 128 * 16);
    const int per_cu = pn2_opt(PN2_OPT_CF_WGS_PER_CU);
    constexpr int U = 8;
    int64_t grid = pn2_cdiv(P, 4 * U * 4 * 4);                     // >= 4 trips per wave
    if (grid > (int64_t)per_cu * pn2_num_cus()) grid = (int64_t)per_cu * pn2_num_cus();
    if (grid < 1) grid = 1;
    hipStream_t s = pn2_s(stream);
    const int ldc = round4(M);
#define PN2_CF_CASE(NBV)                                                                                                               \
    case NBV: PN2_NOTE_KERNEL(wgrad_first_cf_kernel<NBV, U>); hipLaunchKernelGGL((wgrad_first_cf_kernel<NBV, U>), dim3((unsigned)grid), dim3(256), 0, s, dZ, ldz, coef, ldc, X, ldx, P, N, \
                                 W, ldw, bias, part, mom, ticket, dW, lddw, lc); break;
    switch (M / 16) {
        PN2_CF_CASE(1) PN2_CF_CASE(2) PN2_CF_CASE(3) PN2_CF_CASE(4) PN2_CF_CASE(5) PN2_CF_CASE(6) PN2_CF_CASE(7) PN2_CF_CASE(8)
    }
#undef PN2_CF_CASE
    return pn2_launch_status();
}

int pn2_conv1x1_wgrad_ws(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                         const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                         const pn2_bn_coef_lazy *coef_lazy, float *workspace, pn2_stream_t stream);

int pn2_conv1x1_wgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                      const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                      const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream) {
    return pn2_conv1x1_wgrad_ws(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, X, ldx, x_affine, dW, lddw, dbias, P, M, N, coef_lazy, nullptr, stream);
}

int pn2_conv1x1_wgrad_ws(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                         const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                         const pn2_bn_coef_lazy *coef_lazy, float *workspace, pn2_stream_t stream) {
    PN2_CHECK_ARG(Y && coef && X && dW && P > 0 && P < (1LL << 31) && M > 0 && N > 0 && lazy_coef_ok(coef_lazy, coef, M));
    const LazyCoef lc = make_lazy_coef(coef_lazy);
    PN2_CHECK_ARG(dZ != nullptr || (dZp && arg && Kpool > 0 && P < (1LL << 31)));
    PN2_CHECK_ARG(ldy % 4 == 0 && ldy >= round4(M) && ldx % 4 == 0 && ldx >= round4(N) && lddw >= N);
    const int ldc = round4(M);
    hipStream_t s = pn2_s(stream);
    PN2_CHECK_ARG(dZ ? (ldz % 4 == 0 && ldz >= round4(M)) : (ldo % 4 == 0 && ldo >= round4(M)));
    {                                                                   // wide layer: all of dW resident in one workgroup (mlp_wide.hip)
        // (the workspace is used only if it is as large as the query said it must be: the caller passes what it was told)
        const int rc = pn2_wide_wgrad(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, X, ldx, x_affine, dW, lddw, dbias, P, M, N, lc, s,
                                      pn2_wide_wgrad_workspace_bytes(P, M, N, dZ == nullptr) > 0 ? workspace : nullptr);
        if (rc != PN2_EUNSUPPORTED) return rc;
    }
    if (dZ) {
        const int skinny = pn2_opt(PN2_OPT_WGRAD_SKINNY);
        if (skinny && x_affine == nullptr && N <= 16 && P >= 4096) {     // first layers: stream dZ / Y once, no MFMA
            switch ((N + 3) / 4) {
                case 1: return launch_skinny<1>(dZ, ldz, Y, ldy, coef, ldc, X, ldx, P, M, N, dW, lddw, dbias, s, lc);
                case 2: return launch_skinny<2>(dZ, ldz, Y, ldy, coef, ldc, X, ldx, P, M, N, dW, lddw, dbias, s, lc);
                case 3: return launch_skinny<3>(dZ, ldz, Y, ldy, coef, ldc, X, ldx, P, M, N, dW, lddw, dbias, s, lc);
                default: return launch_skinny<4>(dZ, ldz, Y, ldy, coef, ldc, X, ldx, P, M, N, dW, lddw, dbias, s, lc);
            }
        }
        LoadDyDense dy{dZ, ldz, Y, ldy, coef, ldc, zero_page_dev(), lc};
        if (x_affine) return dispatch_tn(dy, LoadBnReluFixed{X, ldx, x_affine, zero_page_dev()}, P, M, N, dW, lddw, dbias, s);
        return dispatch_tn(dy, LoadPlain{X, ldx, zero_page_dev()}, P, M, N, dW, lddw, dbias, s);
    }
    PN2_CHECK_ARG(ldo % 4 == 0 && ldo >= round4(M));
    LoadDyPooled dy{dZp, ldo, arg, Kpool, Y, ldy, coef, ldc, zero_page_dev(), pow2_shift(Kpool), lc};
    if (x_affine) return dispatch_tn(dy, LoadBnReluFixed{X, ldx, x_affine, zero_page_dev()}, P, M, N, dW, lddw, dbias, s);
    return dispatch_tn(dy, LoadPlain{X, ldx, zero_page_dev()}, P, M, N, dW, lddw, dbias, s);
}

int pn2_conv1x1_bwd_pair(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy,
                         const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev, const float *prev_affine,
                         float *dXout, int ldxo, double *prev_red, const float *X, int ldx, const float *x_affine, float *dW, int lddw,
                         int64_t P, int C_out, int C_in, const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream) {
    PN2_CHECK_ARG(X && dW && C_out > 0 && C_in > 0 && ldx % 4 == 0 && ldx >= round4(C_in) && lddw >= C_in);
    const int on = pn2_opt(PN2_OPT_BWD_PAIR), small_p = pn2_opt(PN2_OPT_TN_SMALLP);
    const int tn_cfg = pn2_opt(PN2_OPT_TN_CFG), tdepth = pn2_opt(PN2_OPT_TN_SMALL_DEPTH);
    const int M = C_out, N = C_in;
    // the weight-gradient half rides in the data gradient's launch only where dispatch_tn would take its few-row configuration
    const bool tn_small = N > 32 && M > 32 && !(M <= 64 && N <= 64) && tn_cfg == 0 && tdepth == 2 &&
                          (P <= small_p || (P <= 2 * (int64_t)small_p && (int64_t)M * N <= 16384));
    PairJob job{on && tn_small, false, X, ldx, x_affine, dW, lddw, M, N};
    const int rc = conv1x1_dgrad_impl(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, W, ldw, prev_Y, ld_prev, prev_affine, dXout, ldxo, prev_red, P,
                                      C_out, C_in, nullptr, coef_lazy, stream, &job);
    const bool taken = job.taken;
    if (rc != PN2_OK || taken) return rc;
    const int rc2 = pn2_conv1x1_wgrad(dZ, ldz, dZp, ldo, arg, Kpool, Y, ldy, coef, X, ldx, x_affine, dW, lddw, nullptr, P, C_out, C_in, nullptr,
                                      stream);
    return rc2 == PN2_OK ? PN2_OK_SPLIT : rc2;             // done, as two launches (callers that account per launch can tell)
}

}  // extern "C"
