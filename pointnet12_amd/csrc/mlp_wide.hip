// Register-stationary shared-MLP GEMMs for the WIDE layers (C_in or C_out above 128, or both 128: the sa2 / FP / head stacks
// of the segmentation networks -- model/pointnet_util.py:197,254,312 at the widths of model/pointnet2.py:109-114,145-153):
//   pn2_conv1x1_fwd   -> regw_nt_kernel<..., EPI_FWD>     Y  = act(X) W^T + b, BatchNorm statistics
//   pn2_conv1x1_dgrad -> regw_nt_kernel<..., EPI_MASK>    dX = dY W masked by the previous ReLU, its BatchNorm-backward sums
//
// Why a third GEMM family.  The streamed kernels of mlp.hip restage the WEIGHT tile through LDS for every 64-row tile and
// synchronise their four waves every 16 MFMAs; the weight-resident kernels of mlp_res.hip need W in LDS (<= 128 x 128).  On a
// tall, skinny product the weights are the operand that never changes, so here they never move at all:
//   * every wave owns ONE 32-column block of the output for the whole launch and holds its slice of W as the B operand of
//     v_mfma_f32_32x32x2_f32 in REGISTERS (K/2 per lane: 64 .. 128), loaded once;
//   * the activations (the only streamed operand) go through LDS in 64-deep chunks shared by all waves of the workgroup:
//     one barrier per chunk and 64 .. 128 MFMAs per wave between barriers (round 3 measurements, tools/exp/regw_nt.hip: a
//     barrier interval costs 1 000 - 1 500 cycles of matrix-pipe idle when both waves of a SIMD stage at the same time, so the
//     staging of chunk c + 1 is spread between the MFMA groups of chunk c and the intervals are long);
//   * LDS holds nothing but two activation chunks: 70 KB, whatever K and N are.
// Ceiling: 157.3 TF.  (Round 3 read ~2.0 GHz from GRBM_GUI_ACTIVE / 8 / time under these kernels and called 131 TF the
// practical ceiling; round 4's barrier-free pure-MFMA loop on the same pool holds 2.39 GHz and delivers 154.6 TF with or without
// a workgroup barrier every 64 MFMAs -- tools/exp/mfma_peak.hip, profiles/r04_mfma_peak.txt -- so whatever clock these kernels
// hold is a property of their own mix of LDS, memory and matrix work, not a ceiling.)
//
// Whole BM-row tiles only; the entry points of mlp.hip hand a ragged tail to the streamed kernels.
#include "mlp_loaders.h"
#include "split_bf16.h"

namespace {

enum { MODE_PLAIN = 0, MODE_BNRELU = 1, MODE_DYDENSE = 2, MODE_DYPOOLED = 3 };
enum { EPI_FWD = 0, EPI_MASK = 1, EPI_STORE = 2 };

struct RegwArgs {
    // streamed operand: X (forward) or this layer's pre-BN output Y (dgrad), [P, lda]
    const float *A; int lda;
    const float *dZ; int ldz;                                      // MODE_DYDENSE
    const float *dZp; const int32_t *arg; int ldo; int kshift;     // MODE_DYPOOLED: [G, ldo], Kpool = 1 << kshift
    const float *tab;                                              // MODE_BNRELU: affine block of the input (rows mean, scale, beta of
                                                                   // pitch K4); dgrad: coef rows c0, q1, q0, mean of pitch K4
    const float *W; int ldw;                                       // forward: [N, K] rows; dgrad (BNN): [K, N] rows
    const float *bias;                                             // EPI_FWD
    float *Out; int ldout;                                         // Y or dX
    const float *prevY; int ldp; const float *prev_aff;            // EPI_MASK: previous layer's pre-BN output and affine block (pitch N4)
    double *red;                                                   // EPI_FWD: stats; EPI_MASK: prev_red (replicated, may be null)
    int64_t tiles; int K; int N;
    float2 *pool_rec; const float *pool_gamma; int pool_ld;        // EPI_FWD with PKP > 0: per-group extrema records [G, pool_ld]
    LazyBn lz;                                                     // MODE_BNRELU: `tab` is filled by the prologue (consumer-side BatchNorm)
    LazyCoef lc;                                                   // dgrad: `tab` (the coefficient block) likewise
};

extern __shared__ __attribute__((aligned(16))) float wide_lds[];

// K4: row length of the streamed operand in floats (a multiple of 4, >= K).  NCB: 32-column blocks of the output (one per wave
// column), RS: row splits (waves = NCB * RS), TM: 32-row blocks per wave; workgroup tile BM = 32 * TM * RS rows x 32 * NCB.
template <int K4, int NCB, int RS, int TM, int KC, int MODE, int EPI, bool BNN, int PKP, bool ADB>
__global__ __launch_bounds__(64 * NCB * RS) void regw_nt_kernel(const RegwArgs g) {
    constexpr int KP = (K4 + 7) & ~7, NT = 64 * NCB * RS, BM = 32 * TM * RS, LDP = KC + 4, NCH = (KP + KC - 1) / KC;
    constexpr int NTAB = MODE == MODE_PLAIN ? 0 : (MODE == MODE_BNRELU ? 3 : 4);
    constexpr bool DY = MODE == MODE_DYDENSE || MODE == MODE_DYPOOLED;
    static_assert(KC % 8 == 0 && K4 % 4 == 0, "chunk / row geometry");
    // chunk c: k in [c * KC, c * KC + 4 * QT(c)); QV(c) of its QT(c) float4 quads exist in memory, the rest (k >= K4, at most
    // one quad of the last chunk) are written as zeros
    auto QT = [](int c) constexpr { return ((KP - c * KC) < KC ? (KP - c * KC) : KC) / 4; };
    auto QV = [](int c) constexpr { return ((K4 - c * KC) < KC ? (K4 - c * KC) : KC) / 4; };
    constexpr int A_IT = (BM * (KC / 4) + NT - 1) / NT;            // staging items (float4 of one row) per thread, widest chunk
    // Pooled dZ (dZ[g * Kp + kk, c] = dZp[g, c] if kk == arg[g, c]): a thread's items all sit in the same channel quad
    // (NT % (KC / 4) == 0, every chunk full), (NT / (KC / 4)) rows apart, so the (dZp, arg) quads repeat per GROUP:
    // NZ distinct ones per thread and chunk instead of A_IT.
    constexpr bool POOLED = MODE == MODE_DYPOOLED;
    constexpr int RPI = NT / (KC / 4);                              // rows between a thread's consecutive items
    static_assert(!POOLED || (PKP > 0 && (PKP & (PKP - 1)) == 0 && NT % (KC / 4) == 0 && K4 % KC == 0 && (BM <= PKP || PKP % RPI == 0)),
                  "pooled loader geometry");
    constexpr int NZ = !POOLED ? 1 : (BM <= PKP ? 1 : (RPI * (A_IT - 1)) / (PKP > 0 ? PKP : 1) + 1);
    auto zslot = [](int i) constexpr { return BM <= PKP ? 0 : (RPI * i) / (PKP > 0 ? PKP : 1); };

    float *As0 = wide_lds, *As1 = wide_lds + BM * LDP;
    float *tab = wide_lds + 2 * BM * LDP;                           // NTAB rows of KP floats
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int cb = wave % NCB, rs = wave / NCB;
    const int n = cb * 32 + l31;                                    // this lane's output column
    const int N = g.N, K = g.K;
    if (MODE == MODE_BNRELU) lazy_bn_prologue(g.lz);                // consumer-side BatchNorm (bn_tail.h): before `tab` is copied
    if (DY) lazy_coef_prologue(g.lc);

    // ---- one-time: this lane's slice of W, the B operand of every MFMA it issues: w[4 kb + e] = W(n, k = 8 kb + 4 lh + e)
    float w[KP / 2];
    if (BNN) {                                                      // W stored [K, N]: consecutive lanes on consecutive n
#pragma unroll
        for (int kb = 0; kb < KP / 8; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * kb + 4 * lh + e;
                w[kb * 4 + e] = (n < N && k < K) ? g.W[(int64_t)k * g.ldw + n] : 0.f;
            }
    } else {
        // W stored [N, K]: a lane's values lie along a row, 32 lanes on 32 rows.  Read straight from global that is 32 cache
        // lines per instruction and a 4x over-fetch from L2 (measured: 10+ us of fixed cost per launch).  Instead every wave
        // copies its 32 rows chunk by chunk into a private LDS region with coalesced reads and picks its operands from there.
        float *reg = wide_lds + wave * (32 * LDP);                  // NCB * RS * 32 <= 2 * BM rows: inside the chunk buffers
        const bool vec = (g.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(g.W) & 15) == 0 && (K & 3) == 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            constexpr int dummy = 0; (void)dummy;
            const int qt = QT(c);
            if (vec) {
                for (int idx = lane; idx < 32 * qt; idx += 64) {
                    const int row = idx / qt, q = idx - row * qt, k = c * KC + 4 * q, nn = cb * 32 + row;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nn < N && k < K) v = ld4(g.W + (int64_t)nn * g.ldw + k);
                    *reinterpret_cast<float4 *>(&reg[row * LDP + 4 * q]) = v;
                }
            } else {
                for (int idx = lane; idx < 32 * 4 * qt; idx += 64) {
                    const int row = idx / (4 * qt), kk = idx - row * (4 * qt), k = c * KC + kk, nn = cb * 32 + row;
                    reg[row * LDP + kk] = (nn < N && k < K) ? g.W[(int64_t)nn * g.ldw + k] : 0.f;
                }
            }
            // same wave, LDS operations complete in order: no barrier between these writes and reads
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb)
                if (kb < qt / 2) {
                    const float4 v = *reinterpret_cast<const float4 *>(&reg[l31 * LDP + 8 * kb + 4 * lh]);
                    const int wi = (c * (KC / 8) + kb) * 4;
                    w[wi] = v.x; w[wi + 1] = v.y; w[wi + 2] = v.z; w[wi + 3] = v.w;
                }
        }
        __syncthreads();                                            // the regions alias the chunk buffers
    }
    for (int i = t; i < NTAB * KP; i += NT) {
        const int r = i / KP, k = i - r * KP;
        tab[i] = k < K4 ? g.tab[r * K4 + k] : 0.f;
    }

    // ---- per-lane epilogue constants (the lane's column never changes)
    const int N4 = (N + 3) & ~3;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;                   // EPI_FWD: bias; EPI_MASK: mean, scale, beta, invstd of column n
    if (EPI == EPI_FWD) e0 = n < N ? g.bias[n] : 0.f;
    // FPOOL: this forward feeds a max over groups of FPOOL consecutive rows (the last layer of a set-abstraction MLP): the
    // epilogue also records, per group and channel, the extreme pre-BN value -- the largest, or the smallest where this
    // layer's BatchNorm weight is negative (sign bit flipped: e1) -- with the first row attaining it, exactly as the
    // weight-resident forward does (mlp_res.hip, ResPool); pn2_bn_pool_select turns the records into max_k relu(bn(y_k)).
    constexpr int FPOOL = EPI == EPI_FWD ? PKP : 0;
    static_assert(FPOOL == 0 || (RS == 1 && BM % FPOOL == 0 && FPOOL % 32 == 0), "forward pooling: whole groups inside a wave's tile");
    if (FPOOL > 0) e1 = __int_as_float((n < N && g.pool_gamma[n] < 0.f) ? (int)0x80000000 : 0);
    // LATE_E: the 196-deep dense data gradient holds 100 weight registers next to two accumulator tiles and four staging
    // items; with the four epilogue constants resident as well it spilled six loop-invariant registers (28 bytes of scratch,
    // reloaded once per tile).  There the constants are fetched per tile instead, beside the prevY requests of the epilogue
    // (same cache lines for every tile: L1 hits that return under the wait those requests need anyway).
    constexpr bool LATE_E = EPI == EPI_MASK && KP == 200;
    // (re-deriving the offsets in EVERY instantiation frees 12 .. 28 registers but measured 3 - 5 % slower on the forward
    // layers -- 162 -> 170, 231 -> 238, 80.6 -> 83.7 us -- so only the two register-bound cases take it)
    if (EPI == EPI_MASK && !LATE_E && n < N4) {
        Affine a(g.prev_aff, N4);
        e0 = a.mean[n]; e1 = a.scale[n]; e2 = a.beta[n]; e3 = a.invstd[n];
    }
    double st0 = 0.0, st1 = 0.0;

    // ---- streamed operand: raw registers of one chunk in flight, turned into operand values when staged
    struct Raw { float4 y[A_IT]; float4 z[POOLED ? NZ : (DY ? A_IT : 1)]; int4 a[NZ]; };
    Raw raw;
    const int64_t tiles = g.tiles;
    const int G = gridDim.x;
    // item i of chunk c: idx = t + NT * i over BM rows x QT(c) quads; row = idx / QT, q = idx % QT.
    // FULL chunks (QT == QV == KC / 4 =: QF, and NT % QF == 0): a thread's items all sit in ONE channel quad lq = t % QF, RPI rows
    // apart, so every global address is a WAVE-UNIFORM base (tile, item and chunk terms: scalar arithmetic) plus one of three
    // loop-invariant 32-bit lane offsets, and the LDS address one lane offset plus an immediate.  (Round 4 ablation,
    // tools/exp/x_wide.sh: with the per-item index arithmetic -- two divisions, two 64-bit multiply-adds and four 64-bit adds
    // per request, ~25 vector instructions -- the requests alone cost 10 - 25 % of these kernels.)
    constexpr int QF = KC / 4;
    static_assert(NT % QF == 0, "a thread's items of a full chunk share one channel quad");
    int tq = t;                                                     // (LATE_E: re-derived per chunk, see the chunk loop)
    const int lrow = t / QF, lq = t - lrow * QF;
    const unsigned vq4 = 4u * (unsigned)lq;                         // element offsets, lane part
    const unsigned voA = (unsigned)lrow * (unsigned)g.lda + vq4;
    const unsigned voZ = MODE == MODE_DYDENSE ? (unsigned)lrow * (unsigned)g.ldz + vq4 : 0u;
    const unsigned ldsw = (unsigned)lrow * LDP + vq4;
    auto is_full = [&](int c) constexpr { return QT(c) == QF && QV(c) == QF; };
    auto fetch_item = [&](int64_t tile, int c, int i) {
        const int qt = QT(c), qv = QV(c);
        const unsigned tl = (unsigned)(tile < tiles ? tile : tiles - 1);      // past the end: re-read the last tile (never used)
        if (is_full(c)) {
            if (POOLED && (i == A_IT - 1 || zslot(i + 1) != zslot(i))) {
                // The (dZp, arg) quad of a group is shared by all of the thread's items in that group: it is requested behind
                // the LAST of them (the items of the chunk being staged still read the previous one), by every thread.  The group
                // is wave-uniform: a tile lies inside one group (BM <= PKP), or the groups are whole multiples of the row step
                // RPI and the lane's row < RPI never crosses into the next one.
                int i0 = i;
                while (i0 > 0 && zslot(i0 - 1) == zslot(i)) --i0;
                const unsigned grp = (tl * BM + (unsigned)(i0 * RPI)) / (unsigned)(PKP > 0 ? PKP : 1);
                const size_t go = (size_t)grp * (unsigned)g.ldo + (unsigned)(c * KC);
                raw.z[POOLED ? zslot(i) : 0] = ld4(g.dZp + go + vq4);
                raw.a[POOLED ? zslot(i) : 0] = ld4i(g.arg + go + vq4);
            }
            if (NT * i >= BM * QF) return;                                   // static: this chunk has fewer items
            if (NT * (i + 1) > BM * QF && lrow + i * RPI >= BM) return;      // the last, partly filled pass
            const size_t m = (size_t)tl * BM + (unsigned)(i * RPI);          // uniform
            raw.y[i] = ld4(g.A + m * (unsigned)g.lda + (unsigned)(c * KC) + voA);
            if (MODE == MODE_DYDENSE) raw.z[DY ? i : 0] = ld4(g.dZ + m * (unsigned)g.ldz + (unsigned)(c * KC) + voZ);
            return;
        }
        const int idx = tq + NT * i, row = idx / qt, q = idx - row * qt;
        if (NT * i >= BM * qt) return;                                   // static: this chunk has fewer items
        if (NT * (i + 1) > BM * qt && idx >= BM * qt) return;            // the last, partly filled pass
        const unsigned qq = q < qv ? q : qv - 1;                               // a pad quad re-reads the last valid one (zeroed when staged)
        const unsigned m = tl * BM + row;
        const int k = c * KC + 4 * qq;
        raw.y[i] = ld4(g.A + row_off(m, g.lda) + k);
        if (MODE == MODE_DYDENSE) raw.z[DY ? i : 0] = ld4(g.dZ + row_off(m, g.ldz) + k);
    };
    // The per-channel constants of a full chunk (BatchNorm mean / scale / beta, or the four BatchNorm-backward coefficient
    // rows) depend on the chunk alone: fetched from the table once per chunk (HOIST; the memory clobbers that pin the staging
    // to its k block would otherwise make every item read them again), where the register budget has room for 12 / 16 more.
    struct Consts { float4 a, b, c, d; };
    constexpr bool HOIST = MODE != MODE_PLAIN && !(EPI == EPI_MASK && KP == 200);
    auto chunk_consts = [&](int c) {
        Consts o;
        const int k = c * KC + (int)vq4;
        o.a = *reinterpret_cast<const float4 *>(&tab[k]);
        o.b = *reinterpret_cast<const float4 *>(&tab[KP + k]);
        o.c = *reinterpret_cast<const float4 *>(&tab[2 * KP + k]);
        o.d = NTAB > 3 ? *reinterpret_cast<const float4 *>(&tab[(NTAB > 3 ? 3 : 0) * KP + k]) : o.c;
        return o;
    };
    auto stage_item = [&](float *dst, int64_t tile, int c, int i, const Consts &cc) {
        const int qt = QT(c), qv = QV(c);
        const bool full = is_full(c);
        const int idx = tq + NT * i;
        const int row = full ? lrow + i * RPI : idx / qt, q = full ? lq : idx - (idx / qt) * qt;
        if (NT * i >= BM * qt) return;                                   // static: this chunk has fewer items
        if (NT * (i + 1) > BM * qt && (full ? row >= BM : idx >= BM * qt)) return;   // the last, partly filled pass
        const int k = c * KC + 4 * q;
        float4 x = raw.y[i];
        if (MODE == MODE_BNRELU) {
            const bool h = HOIST && full;
            const float4 mu = h ? cc.a : *reinterpret_cast<const float4 *>(&tab[k]);
            const float4 sc = h ? cc.b : *reinterpret_cast<const float4 *>(&tab[KP + k]);
            const float4 be = h ? cc.c : *reinterpret_cast<const float4 *>(&tab[2 * KP + k]);
            x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
            x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
            x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
            x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
        }
        if (DY) {
            DyParams dp;
            if (HOIST && full) { dp.c0 = cc.a; dp.q1 = cc.b; dp.q0 = cc.c; dp.mu = cc.d; }
            else dp = dy_params_tab(tab, KP, k, true);
            float4 dz = raw.z[POOLED ? zslot(i) : (DY ? i : 0)];
            if (POOLED) {
                const int4 a = raw.a[POOLED ? zslot(i) : 0];
                const unsigned tl = (unsigned)(tile < tiles ? tile : tiles - 1);
                const int kk = (int)((tl * BM + (unsigned)(i * RPI) + (unsigned)lrow) & (unsigned)(PKP - 1));
                dz.x = a.x == kk ? dz.x : 0.f; dz.y = a.y == kk ? dz.y : 0.f;
                dz.z = a.z == kk ? dz.z : 0.f; dz.w = a.w == kk ? dz.w : 0.f;
            }
            x = dy_from(dz, x, dp);
        }
        if (qv != qt && q >= qv) x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (full) *reinterpret_cast<float4 *>(&dst[ldsw + (unsigned)(i * RPI * LDP)]) = x;
        else *reinterpret_cast<float4 *>(&dst[row * LDP + 4 * q]) = x;
    };

    int64_t tile = blockIdx.x;
    float *cur = As0, *nxt = As1;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) fetch_item(tile, 0, i);
    __syncthreads();                                                // table complete (and the W regions released)
    {
        const Consts c0 = chunk_consts(0);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) stage_item(cur, tile, 0, i, c0);
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        if (NCH > 1) fetch_item(tile, 1, i); else fetch_item(tile + G, 0, i);
    }

    while (tile < tiles) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            __syncthreads();                                        // chunk c is in `cur`; every wave is done with `nxt`
            // LATE_E (register budget): the lane's staging offsets are re-derived inside every chunk instead of living in
            // registers across the whole tile loop (one of them was spilled)
            if (LATE_E || (FPOOL > 0 && KP == 200)) asm volatile("" : "+v"(tq));
            const int kbs = QT(c) / 2;                              // 8-wide k blocks of this chunk (static after unrolling)
            const bool last = c == NCH - 1;
            const int64_t t1 = last ? tile + G : tile;              // the chunk staged during this one ...
            const int c1 = last ? 0 : c + 1;
            const bool last1 = c1 == NCH - 1;
            const int64_t t2 = last1 ? t1 + G : t1;                 // ... and the one requested
            const int c2 = last1 ? 0 : c1 + 1;
            const float *ap = cur + (rs * TM * 32 + l31) * LDP + 4 * lh;
            // Operand reads run one k block ahead of the MFMAs that consume them (two register sets).  The memory clobbers pin
            // the LDS / global operations to their k block: without them the compiler hoists every read of the chunk to its
            // top (TM x 8 x 4 registers: spills) and sinks the staging behind the last MFMA, where both waves of a SIMD
            // would do it at the same time with the matrix pipe idle.
            Consts cc;
            if (HOIST && is_full(c1)) cc = chunk_consts(c1);       // (the table is read-only after the first barrier)
            // staging item i of the next chunk rides behind k block slot(i): spread over the chunk's k blocks, so that the
            // vector work of one item sits in the shadow of eight MFMAs instead of queueing behind the previous item's
            constexpr int dummy_n = 0; (void)dummy_n;
            const int n1 = (BM * QT(c1) + NT - 1) / NT;            // items of the chunk being staged (static after unrolling)
            const int step = kbs / n1 > 0 ? kbs / n1 : 1;
            float4 a[ADB ? 2 : 1][TM];
            if (ADB) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDP);
            }
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                if (kb < kbs) {
                    asm volatile("" ::: "memory");
                    if (ADB && kb + 1 < kbs) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            a[ADB ? (kb + 1) & 1 : 0][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDP + 8 * (kb + 1));
                    }
                    if (!ADB) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * LDP + 8 * kb);
                    }
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) {
                        const int slot = i * step < kbs ? i * step : kbs - 1;
                        if (slot == kb) {
                            stage_item(nxt, t1, c1, i, cc);
                            fetch_item(t2, c2, i);
                        }
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float4 av = a[ADB ? kb & 1 : 0][i];
                        const int wi = (c * (KC / 8) + kb) * 4;
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w[wi + 0], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w[wi + 1], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w[wi + 2], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w[wi + 3], acc[i], 0, 0, 0);
                    }
                }
            }
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        // ---- epilogue straight from the accumulators: column on the lane, 128 contiguous bytes per half-wave and register.
        // The lane offset is made opaque once per tile: loop-invariant addresses would otherwise be hoisted out of the tile
        // loop into 2 x 16 x TM registers (and spilled).
        {
            const unsigned row0 = (unsigned)tile * BM + rs * TM * 32 + 4 * lh;
            unsigned lo = (unsigned)n;
            asm volatile("" : "+v"(lo));
            float s0 = 0.f, s1 = 0.f;
            if (EPI == EPI_FWD) {
                float *yb = g.Out + row_off(row0, g.ldout);
                unsigned off = lo;
                constexpr int GPT = FPOOL > 0 ? BM / (FPOOL > 0 ? FPOOL : 1) : 1, BPG = FPOOL > 0 ? FPOOL / 32 : TM;
                const int sg = __float_as_int(e1);
                float mv[GPT];
                int mk[GPT];
#pragma unroll
                for (int gq = 0; gq < GPT; ++gq) { mv[gq] = -INFINITY; mk[gq] = 0; }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float y = acc[i][r] + e0;
                        if (n < N4) PN2_STREAM_STORE(y, yb + off);      // pad columns receive exact zeros (w = bias = 0)
                        s0 += y;
                        s1 = __builtin_fmaf(y, y, s1);
                        if (FPOOL > 0) {
                            // ascending block, ascending register = ascending row for this lane: a strict > keeps the first row
                            const int gq = i / BPG;
                            const float yp = __int_as_float(__float_as_int(y) ^ sg);
                            const int row = (i - gq * BPG) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            mk[gq] = yp > mv[gq] ? row : mk[gq];
                            mv[gq] = fmaxf(mv[gq], yp);
                        }
                        off += ((r & 3) == 3 ? (r == 15 ? 5u : 5u) : 1u) * (unsigned)g.ldout;   // rows (r&3) + 8 (r>>2): +1 +1 +1 +5
                    }
                if (FPOOL > 0) {
#pragma unroll
                    for (int gq = 0; gq < GPT; ++gq) {
                        const float ov = __shfl_xor(mv[gq], 32, 64);   // the two half-waves hold disjoint rows of the column
                        const int ok = __shfl_xor(mk[gq], 32, 64);
                        const bool take = ov > mv[gq] || (ov == mv[gq] && ok < mk[gq]);
                        const float v = take ? ov : mv[gq];
                        const int k = take ? ok : mk[gq];
                        // unconditional 8-byte stores, the same number every tile (both half-waves write the same record)
                        if (n < N4)
                            g.pool_rec[(int64_t)((unsigned)tile * GPT + gq) * g.pool_ld + lo] =
                                make_float2(__int_as_float(__float_as_int(v) ^ sg), __int_as_float(k));
                    }
                }
            } else if (EPI == EPI_MASK) {
                const float *pb = g.prevY + row_off(row0, g.ldp);
                float *xb = g.Out + row_off(row0, g.ldout);
                unsigned offp = lo, offx = lo;
                asm volatile("" : "+v"(offx));
                if (LATE_E) {
                    Affine a(g.prev_aff, N4);
                    const unsigned nl = n < N4 ? lo : 0u;                // (opaque per tile: not hoisted back out of the loop)
                    e0 = a.mean[nl]; e1 = a.scale[nl]; e2 = a.beta[nl]; e3 = a.invstd[nl];
                    if (!(n < N4)) e1 = e2 = 0.f;
                }
                constexpr int PVB = 16;                                  // prevY values in flight per batch (8, requested in two dependent
                                                                         // rounds, cost the 196 -> 128 launch 174 -> 213 us: four exposed
                                                                         // round trips per tile instead of two)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int rb = 0; rb < 16; rb += PVB) {
                        float pv[PVB];
#pragma unroll
                        for (int r = rb; r < rb + PVB; ++r) {
                            pv[r - rb] = n < N4 ? pb[offp] : 0.f;
                            offp += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldp;
                        }
#pragma unroll
                        for (int r = rb; r < rb + PVB; ++r) {
                            const float y = pv[r - rb];
                            const float dz = bn_act(y, e0, e1, e2) > 0.f ? acc[i][r] : 0.f;   // pad columns: scale = beta = 0 -> 0
                            if (n < N4) PN2_STREAM_STORE(dz, xb + offx);
                            s0 += dz;
                            s1 = __builtin_fmaf(dz, (y - e0) * e3, s1);
                            offx += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldout;
                        }
                    }
            } else {
                float *xb = g.Out + row_off(row0, g.ldout);
                unsigned off = lo;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (n < N4) xb[off] = acc[i][r];
                        off += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldout;
                    }
            }
            if (EPI != EPI_STORE) { st0 += (double)s0; st1 += (double)s1; }
        }
        tile += G;
    }
    if (EPI != EPI_STORE && g.red != nullptr) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (lh == 0 && n < N) {
            double *rep = g.red + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
            atomicAdd(rep + n, st0);
            atomicAdd(rep + N + n, st1);
        }
    }
}

template <int K4, int NCB, int RS, int TM, int KC, int MODE, int EPI, bool BNN, int PKP = 0, bool ADB = true>
int launch_regw(const RegwArgs &g, hipStream_t s) {
    constexpr int KP = (K4 + 7) & ~7, BM = 32 * TM * RS, LDP = KC + 4;
    constexpr int NTAB = MODE == MODE_PLAIN ? 0 : (MODE == MODE_BNRELU ? 3 : 4);
    constexpr size_t lds = sizeof(float) * (2 * BM * LDP + NTAB * KP);
    static_assert(BNN || NCB * RS * 32 <= 2 * BM, "the W staging regions must fit inside the chunk buffers");
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = regw_nt_kernel<K4, NCB, RS, TM, KC, MODE, EPI, BNN, PKP, ADB>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    const int64_t cap = pn2_num_cus();
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles < cap ? g.tiles : cap)), dim3(64 * NCB * RS), lds, s, g);
    return pn2_launch_status();
}


// ================================================================================================ forward, LDS-DMA ring (round 5)
// The same register-stationary product with the streamed operand moved global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no VGPR destination) instead of through a register set of the staging threads, and with the two waves of every SIMD
// running HALF A STEP APART.  What the in-kernel stamps of the register-staged kernel and of the first ring version said
// (tools/stamp_wide.py, profiles/r05_stamp_ring.txt): the MFMA sections of a tile keep the matrix pipe busy, everything else --
// the epilogue's 64 stores per wave (6 - 12 % of the launch), the barrier skew and the DMA / address phase behind each
// barrier (9 - 14 %) -- happens on both waves of a SIMD at the same time, with the pipe idle.  So:
//   * ring of FOUR slots, step s = (tile, 64-deep chunk); every wave: barrier, DMA of step s + 2 (its own 1-KiB pieces),
//     first half of the step's k blocks with the BatchNorm + ReLU of step s + 1 applied IN PLACE between them (every thread
//     its own float4 items: the staging pass of the kernel above, once per workgroup), s_waitcnt vmcnt(0) lgkmcnt(0), barrier,
//     second half, epilogue after the last chunk of a tile;
//   * waves NW / 2 .. NW - 1 (group B: the SECOND wave of every SIMD) execute one more barrier in front of the loop and the
//     others (group A) one more behind it: identical code, B one barrier interval behind A.  A's epilogue and DMA issue then
//     sit beside B's MFMAs and the other way round.  The fourth slot is what the lag costs: B still reads step s - 1 when
//     A requests step s + 2.  Who needs what when (G(k): the k-th barrier interval; A runs its half steps 2 s, 2 s + 1 in
//     G(2 s), G(2 s + 1), B in G(2 s + 1), G(2 s + 2)):
//       DMA(s + 2)      issued at the head of G(2 s) (A) / G(2 s + 1) (B) into the slot of step s - 2, last read in G(2 s - 2);
//                       waited for by its issuer at the end of that interval (half a step, > 3 us later);
//       transform(s + 1) in the FIRST half step only: A's items in G(2 s), B's in G(2 s + 1); its raw data landed before
//                       barrier 2 s (B's wait at the end of G(2 s - 1)); first read behind barrier 2 s + 2;
//   * no load is visible to the compiler inside the tile loop, so it never waits for one (the register-staged kernel opened
//     every tile with a vmcnt(0) that also waited for the HBM acknowledgement of the previous tile's 64 stores: loads and
//     stores retire through one in-order counter).  Barriers are raw s_barrier: __syncthreads() would drain the DMA.
//   LDS image of a chunk: row r = 64 floats = sixteen 16-byte slots, slot s holds the row's k quad s ^ (r & 15).  One DMA
//   instruction writes 1 KiB = four whole rows (lane L: row 4 p + L / 16, slot L % 16), so the image is lane-linear as the
//   DMA requires and the swizzle sits in the per-lane SOURCE address; the sixteen lanes of every ds_read_b128 service group
//   ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: sixteen distinct r & 15) then read sixteen distinct slots: conflict-free
//   without a padded pitch (SQ_LDS_BANK_CONFLICT 1.6e4 of 1.0e7 LDS cycles).  K = 196: the 49th quad (k = 192 .. 195) of
//   the rows rides as a 16-byte-per-row image behind the last full chunk and is consumed as a ninth k block of that step
//   (the lanes of the upper half-wave multiply it by the zero weights of k = 196 .. 199).
#ifdef PN2_STAMP
// Diagnostic build only (make STAMP=1): per-phase shader-cycle sums of every wave of the first 64 workgroups of the last ring
// kernel launch (pn2_debug_stamps_wide).  Phases: 0 barriers, 1 DMA issue, 2 first half, 3 waits, 4 epilogue, 5 second half.
__device__ unsigned long long pn2_wide_stamp_buf[64 * 8 * 8];
__device__ unsigned long long pn2_wide_abs_buf[256 * 8 * 4];     // 100 MHz ticks: kernel entry, loop start, loop end, exit (every workgroup)
#define WABS(wv, i) if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && blockIdx.y == 0) pn2_wide_abs_buf[(blockIdx.x * 8 + ((wv) & 7)) * 4 + (i)] = wall_clock64();
#define WSTAMP_DECL unsigned long long wst_t = clock64(), wst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long wst_c0 = wst_t, wst_w0 = wall_clock64();
#define WSTAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = clock64(); wst_acc[i] += n_ - wst_t; wst_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define WSTAMP_FLUSH(wv) { wst_acc[6] = clock64() - wst_c0; wst_acc[7] = wall_clock64() - wst_w0; if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) { for (int i_ = 0; i_ < 8; ++i_) pn2_wide_stamp_buf[(blockIdx.x * 8 + (wv)) * 8 + i_] = wst_acc[i_]; } }
#else
#define WABS(wv, i)
#define WSTAMP_DECL
#define WSTAMP(i)
#define WSTAMP_FLUSH(wv)
#endif

// 16 bytes per lane global -> LDS: the lane's bytes land at m0 + 16 lane.  `base` wave-uniform, `voff` the lane's byte offset.
__device__ __forceinline__ void glds16(const float *base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// NX: N is a whole number of 32-column blocks (no column predicate: 64 unconditional stores per tile instead of 64 branches).
// (Measured and removed, profiles/r05_ring_fwd.txt: the wave groups IN step -- same time within 1 %, the barrier waits move, the
// total does not; the k block's MFMAs round robin over the row blocks instead of four dependent products per block -- equal.)
template <int K4, int NCB, int RS, int TM, int MODE, int PKP, bool NX>
__global__ __launch_bounds__(64 * NCB * RS) void ring_fwd_kernel(const RegwArgs g) {
    constexpr int KC = 64, KP = (K4 + 7) & ~7, NW = NCB * RS, NT = 64 * NW, BM = 32 * TM * RS, LDPW = KC + 4;
    constexpr int NCH = K4 / KC;                                    // steps per tile (full chunks)
    constexpr bool TAIL = K4 % KC != 0;                             // one more k quad per row, riding with the last chunk
    static_assert(K4 % KC == 0 || (K4 % KC == 4 && KP == K4 + 4), "a tail is exactly one quad");
    static_assert(MODE == MODE_PLAIN || MODE == MODE_BNRELU, "forward modes");
    constexpr int NTAB = MODE == MODE_BNRELU ? 3 : 0;
    constexpr int NSLOT = 4;
    constexpr int SLOTF = BM * KC + (TAIL ? BM * 4 : 0);            // floats per ring slot
    constexpr int PIECES = BM / 4, PW = (PIECES + NW - 1) / NW;     // 1-KiB DMA pieces of a full chunk, per wave
    constexpr int QI = BM * 16, A_IT = (QI + NT - 1) / NT;          // float4 items of a full chunk, per thread
    constexpr bool HOIST = (NT / 16) % 16 == 0;                     // a thread's items share one k quad
    constexpr int KBF = KC / 8, KBH = KBF / 2;                      // k blocks of a full chunk / of its first half
    static_assert(BM % 64 == 0 && NW * 32 * LDPW <= NSLOT * SLOTF, "tail pieces are 64 rows; the W staging regions alias the ring");
    static_assert(A_IT <= KBH + 1, "the transform fits the first half step");

    float *tab = wide_lds + NSLOT * SLOTF;                          // NTAB rows of KP floats
    float4 *lds4 = reinterpret_cast<float4 *>(wide_lds);
    const unsigned ring_b = (unsigned)(uintptr_t)wide_lds;         // LDS byte address of the ring
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int cb = wave % NCB, rs = wave / NCB;
    const int n = cb * 32 + l31;
    const int N = g.N, K = g.K;
    const bool grp_b = wave >= (NW + 1) / 2;                        // uniform
    if (MODE == MODE_BNRELU) lazy_bn_prologue(g.lz);

    // ---- one-time: this lane's slice of W (as regw_nt_kernel: through a wave-private LDS transposition)
    float w[KP / 2];
    {
        float *reg = wide_lds + wave * (32 * LDPW);
        const bool vec = (g.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(g.W) & 15) == 0 && (K & 3) == 0;
        constexpr int NCW = (KP + KC - 1) / KC;
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            const int qt = ((KP - c * KC) < KC ? (KP - c * KC) : KC) / 4;
            if (vec) {
                for (int idx = lane; idx < 32 * qt; idx += 64) {
                    const int row = idx / qt, q = idx - row * qt, k = c * KC + 4 * q, nn = cb * 32 + row;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nn < N && k < K) v = ld4(g.W + (int64_t)nn * g.ldw + k);
                    *reinterpret_cast<float4 *>(&reg[row * LDPW + 4 * q]) = v;
                }
            } else {
                for (int idx = lane; idx < 32 * 4 * qt; idx += 64) {
                    const int row = idx / (4 * qt), kk = idx - row * (4 * qt), k = c * KC + kk, nn = cb * 32 + row;
                    reg[row * LDPW + kk] = (nn < N && k < K) ? g.W[(int64_t)nn * g.ldw + k] : 0.f;
                }
            }
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb)
                if (kb < qt / 2) {
                    const float4 v = *reinterpret_cast<const float4 *>(&reg[l31 * LDPW + 8 * kb + 4 * lh]);
                    const int wi = (c * (KC / 8) + kb) * 4;
                    w[wi] = v.x; w[wi + 1] = v.y; w[wi + 2] = v.z; w[wi + 3] = v.w;
                }
        }
        __syncthreads();                                            // the regions alias the ring (no DMA in flight yet)
    }
    for (int i = t; i < NTAB * KP; i += NT) {
        const int r = i / KP, k = i - r * KP;
        tab[i] = k < K4 ? g.tab[r * K4 + k] : 0.f;
    }

    const int N4 = (N + 3) & ~3;
    float e0 = n < N ? g.bias[n] : 0.f;
    constexpr int FPOOL = PKP;
    static_assert(FPOOL == 0 || (RS == 1 && BM % FPOOL == 0 && FPOOL % 32 == 0), "forward pooling: whole groups inside a wave's tile");
    float e1 = 0.f;
    if (FPOOL > 0) e1 = __int_as_float((n < N && g.pool_gamma[n] < 0.f) ? (int)0x80000000 : 0);
    double st0 = 0.0, st1 = 0.0;
    // (a use: the compiler waits for these two loads HERE, not at their first use behind the stores of the first epilogue)
    asm volatile("" : "+v"(e0), "+v"(e1));

    const int64_t tiles = g.tiles;
    const int G = gridDim.x;
    // ---- LDS-DMA of one chunk: piece p = rows 4 p .. 4 p + 3; this wave's pieces p = wave + NW i
    const unsigned drow = (unsigned)(lane >> 4), dslot = (unsigned)(lane & 15);
    auto issue_dma = [&](int64_t tile_, int c, unsigned slot) {       // slot: float offset of the ring slot in wide_lds
        const unsigned tl = (unsigned)(tile_ < tiles ? tile_ : tiles - 1);    // past the end: the last tile once more (never used)
        const float *base = g.A + (size_t)tl * BM * (unsigned)g.lda + (unsigned)(c * KC);        // uniform
        const unsigned sb = ring_b + slot * 4u;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int p = wave + NW * i;
            if (PIECES % NW != 0 && p >= PIECES) p = PIECES - 1;    // dummy: the last piece once more (same bytes, same place)
            const unsigned row = 4u * (unsigned)p + drow, q = dslot ^ (row & 15u);
            glds16(base, 4u * (row * (unsigned)g.lda + 4u * q), (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)p * 1024u)));
        }
        if (TAIL && c == NCH - 1 && wave < BM / 64) {               // quad K4 / 4 - 1 of 64 rows per piece
            const unsigned row = 64u * (unsigned)wave + (unsigned)lane;
            glds16(base, 4u * (row * (unsigned)g.lda + (unsigned)KC),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(sb + (unsigned)(BM * KC * 4) + (unsigned)wave * 1024u)));
        }
    };
    // ---- BatchNorm + ReLU of a landed chunk, in place: item i of thread t is the float4 at 4 (t + NT i)
    struct Consts { float4 mu, sc, be; };
    const int xrow = t >> 4, xslot = t & 15;
    auto chunk_consts = [&](int c) {
        Consts o;
        const int k = c * KC + 4 * (xslot ^ (xrow & 15));
        o.mu = *reinterpret_cast<const float4 *>(&tab[k]);
        o.sc = *reinterpret_cast<const float4 *>(&tab[KP + k]);
        o.be = *reinterpret_cast<const float4 *>(&tab[2 * KP + k]);
        return o;
    };
    auto act4 = [&](float4 x, const float4 &mu, const float4 &sc, const float4 &be) {
        x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
        x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
        x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
        x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
        return x;
    };
    auto xform_item = [&](unsigned slot, int c, int i, const Consts &cc) {
        if (MODE != MODE_BNRELU) return;
        if (NT * i >= QI) return;                                    // static
        const int idx = t + NT * i;
        if (NT * (i + 1) > QI && idx >= QI) return;                  // the last, partly filled pass
        float4 *p = &lds4[slot / 4u + (unsigned)idx];
        if (HOIST) *p = act4(*p, cc.mu, cc.sc, cc.be);
        else {
            const int row = idx >> 4, k = c * KC + 4 * ((idx & 15) ^ (row & 15));
            *p = act4(*p, *reinterpret_cast<const float4 *>(&tab[k]), *reinterpret_cast<const float4 *>(&tab[KP + k]),
                      *reinterpret_cast<const float4 *>(&tab[2 * KP + k]));
        }
    };
    auto xform_tail = [&](unsigned slot) {                           // the tail quad of row t (k = K4 - 4 .. K4 - 1)
        if (MODE != MODE_BNRELU || !TAIL) return;
        if (t < BM) {
            float4 *p = &lds4[slot / 4u + (unsigned)(BM * KC / 4 + t)];
            constexpr int k = K4 - 4;
            *p = act4(*p, *reinterpret_cast<const float4 *>(&tab[k]), *reinterpret_cast<const float4 *>(&tab[KP + k]),
                      *reinterpret_cast<const float4 *>(&tab[2 * KP + k]));
        }
    };

    // ---- prologue: steps 0 and 1 requested, step 0 transformed
    unsigned cur = 0u, nxt = SLOTF, nx2 = 2 * SLOTF, nx3 = 3 * SLOTF;
    int64_t tile = blockIdx.x;
    issue_dma(tile, 0, cur);
    if (NCH > 1) issue_dma(tile, 1, nxt); else issue_dma(tile + G, 0, nxt);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's pieces have landed, its table entries are written
    asm volatile("s_barrier" ::: "memory");                         // ... everybody's
    {
        const Consts c0 = (MODE == MODE_BNRELU && HOIST) ? chunk_consts(0) : Consts{};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) xform_item(cur, 0, i, c0);
        if (NCH == 1) xform_tail(cur);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp_b) asm volatile("s_barrier" ::: "memory");              // group B: one barrier interval behind from here on

    // lane part of the operand address in float4 units: row (rs TM 32 + l31), k quad (2 kb + lh) ^ (l31 & 15)
    // (float4 units on purpose: with float offsets hipcc lost the 16-byte alignment through the rotating slot offsets and split
    // every operand read into two ds_read2_b32 -- 2.2e7 bank-conflict cycles per launch)
    const unsigned arow4 = (unsigned)((rs * TM * 32 + l31) * (KC / 4));
    const unsigned ax = (unsigned)(l31 & 15);
    const unsigned atail4 = (unsigned)(BM * KC / 4 + (rs * TM * 32 + l31));

    WSTAMP_DECL
    while (tile < tiles) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool last = c == NCH - 1;
            const int64_t t1 = last ? tile + G : tile;              // step s + 1: transformed during the first half of this step
            const int c1 = last ? 0 : c + 1;
            const bool last1 = c1 == NCH - 1;
            const int64_t t2 = last1 ? t1 + G : t1;                 // step s + 2: requested now
            const int c2 = last1 ? 0 : c1 + 1;
            const int kbs = KBF + ((TAIL && last) ? 1 : 0);         // 8-wide k blocks of this step
            // ((2 kb + lh) ^ ax) = (2 kb) ^ (lh ^ ax): one opaque register per step, so that hipcc re-derives the eight
            // operand addresses of a step with one v_xor each instead of keeping them (and their per-slot sums) in
            // registers across the tile loop (the 128 -> 256 instantiation spilled 34 loop-invariant registers)
            unsigned ay = (unsigned)lh ^ ax;
            asm volatile("" : "+v"(ay));
            auto a_read = [&](int kb, int i) {
                if (kb < KBF) return lds4[cur / 4u + arow4 + (unsigned)(i * 32 * KC / 4) + ((unsigned)(2 * kb) ^ ay)];
                return lds4[cur / 4u + atail4 + (unsigned)(i * 32)];
            };
            float4 a[2][TM];
            auto k_block = [&](int kb, bool xf, const Consts &cc) {
                __builtin_amdgcn_sched_barrier(0);                  // (k blocks stay apart: hipcc otherwise overlaps several and spills)
                asm volatile("" ::: "memory");
                if (kb + 1 < kbs) {                                 // operand reads run one k block ahead (also across the mid-step barrier)
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[(kb + 1) & 1][i] = a_read(kb + 1, i);
                }
                if (xf) {
#pragma unroll
                    for (int i = 0; i < A_IT; ++i)
                        if ((i < KBH ? i : KBH - 1) == kb) xform_item(nxt, c1, i, cc);
                    if (kb == KBH - 1 && TAIL && last1) xform_tail(nxt);
                }
                asm volatile("" ::: "memory");
                const int wi = (c * KBF + kb) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 av = a[kb & 1][i];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w[wi + 0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w[wi + 1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w[wi + 2], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w[wi + 3], acc[i], 0, 0, 0);
                }
            };
            WSTAMP(5)
            // ---- first half: request step s + 2, transform step s + 1
            asm volatile("s_barrier" ::: "memory");
            WSTAMP(0)
            issue_dma(t2, c2, nx2);
            WSTAMP(1)
            {
                const Consts cc = (MODE == MODE_BNRELU && HOIST) ? chunk_consts(c1) : Consts{};
#pragma unroll
                for (int i = 0; i < TM; ++i) a[0][i] = a_read(0, i);
#pragma unroll
                for (int kb = 0; kb < KBH; ++kb) k_block(kb, true, cc);
            }
            WSTAMP(2)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // own DMA pieces of step s + 2 (and older stores); own in-place writes
            WSTAMP(3)
            // ---- second half
            asm volatile("s_barrier" ::: "memory");
            WSTAMP(0)
#pragma unroll
            for (int kb = KBH; kb < KBF + 1; ++kb)
                if (kb < kbs) k_block(kb, false, Consts{});
            const unsigned tp = cur; cur = nxt; nxt = nx2; nx2 = nx3; nx3 = tp;
        }
        WSTAMP(5)
        // ---- epilogue straight from the accumulators: column on the lane, 128 contiguous bytes per half-wave and register;
        // wave-uniform base + one 32-bit lane offset (an SGPR-base store: no 64-bit address arithmetic per store)
        {
            float *yb = g.Out + ((size_t)(unsigned)tile * BM + (unsigned)(rs * TM * 32)) * (unsigned)g.ldout;      // uniform
            unsigned off = (unsigned)(4 * lh) * (unsigned)g.ldout + (unsigned)n;
            asm volatile("" : "+v"(off));                            // (opaque per tile: 64 store addresses are not hoisted into registers)
            const unsigned lo = off;
            float s0 = 0.f, s1 = 0.f;
            constexpr int GPT = FPOOL > 0 ? BM / (FPOOL > 0 ? FPOOL : 1) : 1, BPG = FPOOL > 0 ? FPOOL / 32 : TM;
            const int sg = __float_as_int(e1);
            float mv[GPT];
            int mk[GPT];
#pragma unroll
            for (int gq = 0; gq < GPT; ++gq) { mv[gq] = -INFINITY; mk[gq] = 0; }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = acc[i][r] + e0;
                    if (NX || n < N4) PN2_STREAM_STORE(y, yb + off);      // pad columns receive exact zeros (w = bias = 0)
                    s0 += y;
                    s1 = __builtin_fmaf(y, y, s1);
                    if (FPOOL > 0) {
                        // ascending block, ascending register = ascending row for this lane: a strict > keeps the first row
                        const int gq = i / BPG;
                        const float yp = __int_as_float(__float_as_int(y) ^ sg);
                        const int row = (i - gq * BPG) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        mk[gq] = yp > mv[gq] ? row : mk[gq];
                        mv[gq] = fmaxf(mv[gq], yp);
                    }
                    off += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldout;   // rows (r & 3) + 8 (r >> 2): +1 +1 +1 +5
                }
            if (FPOOL > 0) {
#pragma unroll
                for (int gq = 0; gq < GPT; ++gq) {
                    const float ov = __shfl_xor(mv[gq], 32, 64);   // the two half-waves hold disjoint rows of the column
                    const int ok = __shfl_xor(mk[gq], 32, 64);
                    const bool take = ov > mv[gq] || (ov == mv[gq] && ok < mk[gq]);
                    const float v = take ? ov : mv[gq];
                    const int k = take ? ok : mk[gq];
                    // unconditional 8-byte stores, the same number every tile (both half-waves write the same record)
                    if (NX || n < N4)
                        g.pool_rec[(int64_t)((unsigned)tile * GPT + gq) * g.pool_ld + (unsigned)n] =
                            make_float2(__int_as_float(__float_as_int(v) ^ sg), __int_as_float(k));
                }
            }
            (void)lo;
            st0 += (double)s0; st1 += (double)s1;
        }
        WSTAMP(4)
        tile += G;
    }
    WSTAMP_FLUSH(wave)
    if (!grp_b) asm volatile("s_barrier" ::: "memory");             // group A: the barrier group B is one short of
    if (g.red != nullptr) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (lh == 0 && n < N) {
            double *rep = g.red + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
            atomicAdd(rep + n, st0);
            atomicAdd(rep + N + n, st1);
        }
    }
}

template <int K4, int NCB, int RS, int TM, int MODE, int PKP, bool NX>
int launch_ring_fwd_nx(const RegwArgs &g, hipStream_t s) {
    constexpr int KC = 64, KP = (K4 + 7) & ~7, BM = 32 * TM * RS;
    constexpr int SLOTF = BM * KC + (K4 % KC != 0 ? BM * 4 : 0);
    constexpr int NTAB = MODE == MODE_BNRELU ? 3 : 0;
    constexpr size_t lds = sizeof(float) * (4 * SLOTF + NTAB * KP);
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = ring_fwd_kernel<K4, NCB, RS, TM, MODE, PKP, NX>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    const int64_t cap = pn2_num_cus();
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles < cap ? g.tiles : cap)), dim3(64 * NCB * RS), lds, s, g);
    return pn2_launch_status();
}

template <int K4, int NN, int NCB, int RS, int TM, int MODE, int PKP = 0>
int launch_ring_fwd(const RegwArgs &g, hipStream_t s) {
    static_assert(NN <= 32 * NCB && NN > 32 * (NCB - 1), "column blocks");
    return launch_ring_fwd_nx<K4, NCB, RS, TM, MODE, PKP, NN % 32 == 0>(g, s);
}

// ================================================================================================ fp32 products on the bf16 pipe (round 5)
// Where the fp32 GEMMs stand (profiles/r05_ring_fwd.txt): inside their tile loops the register-stationary kernels keep the matrix
// pipe ~88 % busy at the ~2.07 GHz the chip holds -- v_mfma_f32_32x32x2_f32 (157 TF) IS their limit, and the same pipe runs bf16
// sixteen times faster.  Every fp32 number is EXACTLY hi + mid + lo with three bf16 pieces (8 + 8 + 8 significand bits: hi =
// bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both subtractions are exact in fp32), so
//     a b = a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid) + [three terms <= 2^-24 |a b|, dropped]
// -- six v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact, the sums are taken in the fp32 accumulator) instead of
// eight v_mfma_f32_32x32x2_f32 per 32 x 32 x 16 block: 192 instead of 512 matrix-pipe cycles.  The dropped terms are below one
// rounding of the product.  MEASURED (tools/exp/split_gemm.hip, profiles/r05_split_gemm_probe.txt, against fp64 on 11 - 22 k
// sampled outputs with |y| up to 13): max error 2.9e-6 / 4.3e-6 / 7.8e-6 at K = 128 / 192 / 256 where a sequential fp32 fma
// chain -- what v_mfma_f32_32x32x2_f32 and the reference's sgemm compute -- has 5.3e-6 / 5.9e-6 / 7.3e-6 (rms 3.6e-7 vs 4.7e-7):
// AT LEAST as close to the exact product as the fp32 arithmetic it replaces.  (Two pieces / three products: 4e-5 -- not used.)
// The kernels below are then bound by HBM (128 -> 256 at 131 072 rows: 4.07 TB/s in the probe), which is where this path belongs.
//
// split_nt_kernel: Out[P, N] = op(A)[P, K] * W^T (forward: W [N, K] rows; data gradient: W [K, N], BNN) for K <= 208:
//   * a wave owns one 32-column block for the whole launch, its W slice pre-split into three bf16 fragment sets in registers
//     (4 VGPRs per 16-deep k block and piece: 96 at K = 128, 156 at K = 196), two 32-row blocks per wave (TM = 2);
//   * the streamed operand is staged through registers: BatchNorm + ReLU / the BatchNorm-backward transform exactly as the fp32
//     kernels form it (same expressions, same fp32 value), THEN split, three bf16 images per 64-deep chunk in LDS
//     ([piece][row][128 bytes], the 16-byte slot of (k block, half-wave) XOR-swizzled by (row >> 1) & 7: the sixteen rows of a
//     ds_read_b128 service group cover sixteen distinct bank slots), double-buffered, one barrier per chunk;
//   * epilogues as the fp32 kernels': the accumulator layout of 32x32x16 is that of 32x32x2.
// KS = 2 (contraction lengths whose W slice does not fit one wave's registers: K = 256): TWO waves share a column block, each
// holds the W fragments of two of every chunk's four k blocks and accumulates its half of the contraction; at the end of a tile
// the pair swaps one row block each through LDS, so that each wave finishes (adds, masks, stores, reduces) one of the two.
// gridDim.y column groups of NCB blocks each (N = 196 with K = 256: 4 + 4 blocks; the staging of a tile is then done by two
// workgroups -- its rows come from HBM once and from L2 / MALL the second time).

template <int K4, int NCB, int RS, int TM, int MODE, int EPI, bool BNN, int PKP, bool NX, int KS>
__global__ __launch_bounds__(64 * NCB * RS * KS, (NCB * RS * KS <= 4 ? 2 : 1)) void split_nt_kernel(const RegwArgs g) {
    constexpr int KC = 64, NW = NCB * RS * KS, NT = 64 * NW, BM = 32 * TM * RS;
    static_assert(KS == 1 || (KS == 2 && TM == 2 && K4 % KC == 0), "K split: pairs swap one of two row blocks, whole chunks");
    constexpr int KB = (K4 + 15) / 16, KPAD = 16 * KB, NCH = (KPAD + KC - 1) / KC;    // 16-deep k blocks; 64-deep chunks
    constexpr int IMG = BM * KC * 2;                                // bytes of one bf16 piece image of a chunk
    constexpr int NTAB = MODE == MODE_PLAIN ? 0 : (MODE == MODE_BNRELU ? 3 : 4);
    constexpr bool DY = MODE == MODE_DYDENSE || MODE == MODE_DYPOOLED, POOLED = MODE == MODE_DYPOOLED;
    constexpr int FPOOL = EPI == EPI_FWD ? PKP : 0;
    constexpr int SUB = FPOOL > BM ? FPOOL / BM : 1;                // tiles per pooling group (a group spans SUB consecutive tiles)
    constexpr int KBW = KB / KS;                                    // k blocks whose W fragments this wave holds
    static_assert(K4 % 4 == 0 && KBW <= 13, "W slice: 12 registers per k block (8 from 13 blocks on: WL_LDS)");
    static_assert(FPOOL == 0 || (RS == 1 && (BM % FPOOL == 0 || FPOOL % BM == 0) && FPOOL % 32 == 0), "forward pooling geometry");
    static_assert(!POOLED || (PKP > 0 && (PKP % BM == 0 || BM % PKP == 0)), "pooled dY: groups and tiles nest");
    unsigned char *lds_b = reinterpret_cast<unsigned char *>(wide_lds);
    float *tab = wide_lds + (2 * 3 * IMG) / 4;                      // NTAB rows of KPAD floats (data gradients: twice -- see SIGNED)
    // SIGNED (data gradients; round 6): odd TILES are computed negated -- the coefficient rows c0, q1, q0 of the second table copy carry
    // the opposite sign, the epilogue gives the tile its sign back -- so that the one-sided error of the bf16 MFMA's accumulation
    // (mlp_res.hip, SIGN ALTERNATION) alternates from tile to tile and cancels in the BatchNorm-backward sums over the rows
    constexpr bool SIGNED = DY && EPI == EPI_MASK;
    constexpr int TABF = NTAB * KPAD * (SIGNED ? 2 : 1);
    float *xch = tab + TABF;                                        // KS == 2: one row block of accumulators per wave (16 x 64 floats)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int cb = wave % NCB + (int)blockIdx.y * NCB, rs = (wave / NCB) % RS, ks = wave / (NCB * RS);
    // WL_LDS (13 k blocks: 156 fragment registers next to accumulators, operands and the staging is over a wave's 256): the `lo`
    // fragments -- each is the operand of ONE of a k block's six MFMAs -- live in LDS, 1 KiB per (column block, k block), and are
    // read back beside the operand reads: 52 registers for one more ds_read_b128 per step (the first version kept all three sets
    // in registers and hipcc put one of them in scratch, re-read in every tile)
    constexpr bool WL_LDS = KBW >= 13;
    uint4 *wl_lds = reinterpret_cast<uint4 *>(xch + (KS == 2 ? NW * 16 * 64 : 0)) + ((size_t)(wave % NCB) * KBW) * 64 + lane;
    // PN2_SPLIT_ORDER (build-time, A/B): 0 = every wave stages the next chunk and then multiplies; 1 = the second wave of every SIMD
    // multiplies first; 2 = the items of the next chunk are staged BETWEEN the k blocks of this one (a k block's MFMAs, an item's
    // transform + split + image writes + its next request, ...); 3 = as 2, and inside a step one MFMA alternates with a few of the
    // item's vector instructions (sched_group_barrier)
#ifndef PN2_SPLIT_ORDER
#define PN2_SPLIT_ORDER 2
#endif
#ifndef PN2_SPLIT_PV_EARLY
#define PN2_SPLIT_PV_EARLY 1
#endif
#ifndef PN2_SPLIT_TN_WOVEN
#define PN2_SPLIT_TN_WOVEN 1
#endif
    const bool late_stager = PN2_SPLIT_ORDER == 1 && NW >= 8 && wave >= NW / 2;                             // (uniform)
    // N not a multiple of 32 (196): only the last column block is ragged -- the others store without a column predicate (64
    // predicated stores per tile are 64 branches: the epilogue of 128 -> 196 was 31 % of its loop)
    const bool all_cols = NX || (cb + 1) * 32 <= ((g.N + 3) & ~3);  // (uniform)
    const int n = cb * 32 + l31;
    const int N = g.N, K = g.K;
    WABS(wave, 0)
    if (MODE == MODE_BNRELU) lazy_bn_prologue(g.lz);
    if (DY) lazy_coef_prologue(g.lc);

    for (int i = t; i < NTAB * KPAD; i += NT) {
        const int r = i / KPAD, k = i - r * KPAD;
        const float v = k < K4 ? g.tab[r * K4 + k] : 0.f;
        tab[i] = v;
        if (SIGNED) tab[NTAB * KPAD + i] = r < 3 ? -v : v;          // (-c0, -q1, -q0, mean)
    }

    const int N4 = (N + 3) & ~3;
    float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;                   // EPI_FWD: bias, pooling sign; EPI_MASK: mean, scale, beta, invstd of column n
    if (EPI == EPI_FWD) {
        e0 = n < N ? g.bias[n] : 0.f;
        if (FPOOL > 0) e1 = __int_as_float((n < N && g.pool_gamma[n] < 0.f) ? (int)0x80000000 : 0);
    } else if (EPI == EPI_MASK && n < N4) {
        Affine a(g.prev_aff, N4);
        e0 = a.mean[n]; e1 = a.scale[n]; e2 = a.beta[n]; e3 = a.invstd[n];
    }
    double st0 = 0.0, st1 = 0.0;

    // ---- staging: item i of thread t covers the float4 quad q = idx % 16 of row idx / 16 of a chunk (idx = t + NT i)
    constexpr int QI = BM * 16, A_IT = (QI + NT - 1) / NT;
    // ONEZ (pooled dY, the tile inside one group of the max-pool, all staging before all requests): a thread's items lie in one
    // channel quad (NT is a multiple of 16) -- ONE (dZp, arg) quad per chunk serves them all
    constexpr bool ONEZ = POOLED && KS == 2 && PKP % BM == 0;
    struct Raw { float4 y[A_IT]; float4 z[DY ? (ONEZ ? 1 : A_IT) : 1]; int4 a[POOLED ? (ONEZ ? 1 : A_IT) : 1]; };
    Raw raw;
    const int64_t tiles = g.tiles;
    const int G = gridDim.x;
    auto lds_off = [](int buf, int row, int slot) { return (unsigned)(buf * 3 * IMG + row * 128 + 16 * (slot ^ ((row >> 1) & 7))); };
    auto item_ok = [&](int c, int i, int &row, int &q) {            // false: this thread has no i-th item in chunk c
        const int idx = t + NT * i;
        row = idx >> 4; q = idx & 15;
        if (NT * i >= QI) return false;
        if (NT * (i + 1) > QI && idx >= QI) return false;
        return c * KC + 4 * q < KPAD;                               // the last chunk may be narrower
    };
    auto fetch_item = [&](int64_t tile_, int c, int i) {
        // NO lane-dependent predicate around a request: inside a divergent branch hipcc waits for the load it has just issued
        // (s_waitcnt vmcnt(0) + register moves before the join -- a full memory latency per chunk in the ragged instantiations,
        // ISA of the first version).  Items a thread does not have re-read its last one, pad quads the last quad of the row.
        if (NT * i >= QI) return;
        const unsigned tl = (unsigned)(tile_ < tiles ? tile_ : tiles - 1);
        const int idx = (NT * (i + 1) > QI) ? min(t + NT * i, QI - 1) : t + NT * i;
        const int row = idx >> 4, q = idx & 15;
        const int k = c * KC + 4 * q;
        const unsigned kk = k < K4 ? (unsigned)k : (unsigned)(K4 - 4);            // (zeroed when staged)
        const unsigned m = tl * BM + (unsigned)row;
        raw.y[i] = ld4(g.A + row_off(m, g.lda) + kk);
        if (MODE == MODE_DYDENSE) raw.z[DY && !ONEZ ? i : 0] = ld4(g.dZ + row_off(m, g.ldz) + kk);
        if (POOLED && (!ONEZ || i == 0)) {
            const unsigned grp = m / (unsigned)(PKP > 0 ? PKP : 1);
            raw.z[DY && !ONEZ ? i : 0] = ld4(g.dZp + row_off(grp, g.ldo) + kk);
            raw.a[POOLED && !ONEZ ? i : 0] = ld4i(g.arg + row_off(grp, g.ldo) + kk);
        }
    };
    auto fetch = [&](int64_t tile_, int c) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) fetch_item(tile_, c, i);
    };
    // the table entries of a chunk: a thread's items all lie in the same four columns (NT is a multiple of 16)
    struct Tabs { float4 a, b, c, d; };
    auto stage_tabs = [&](int c, int64_t tile_ = 0) {
        Tabs tb{};
        const int k = c * KC + 4 * (t & 15);
        const int kt = k < KPAD ? k : 0;                            // (pad quads of the last chunk: any entry, zeroed when staged)
        if (MODE != MODE_PLAIN) {
            const float *tp = tab + (SIGNED ? (int)(tile_ & 1) * (NTAB * KPAD) : 0);     // (uniform) odd tiles: the negated coefficients
            tb.a = *reinterpret_cast<const float4 *>(&tp[kt]); tb.b = *reinterpret_cast<const float4 *>(&tp[KPAD + kt]);
            tb.c = *reinterpret_cast<const float4 *>(&tp[2 * KPAD + kt]);
            if (DY) tb.d = *reinterpret_cast<const float4 *>(&tp[3 * KPAD + kt]);
        }
        return tb;
    };
    auto stage_item = [&](int64_t tile_, int c, int buf, const Tabs &tb, int i) {
        // branch-free as fetch_item: a thread without an i-th item stages its last one again (the same value to the same place),
        // and the columns past KPAD of the last chunk's image are written (zeros) and never read -- no basic block boundary
        // inside a chunk, so that this work can sit between the MFMAs
        if (NT * i >= QI) return;
        const unsigned tl = (unsigned)(tile_ < tiles ? tile_ : tiles - 1);
        const int idx = (NT * (i + 1) > QI) ? min(t + NT * i, QI - 1) : t + NT * i;
        const int row = idx >> 4, q = idx & 15;
        const int k = c * KC + 4 * q;
        // The transform of an item must not rise above the barrier into the chunk that REQUESTED it: it is register-only work, and
        // hipcc put the max-pool's select directly behind the loads (saving four registers) -- with an s_waitcnt for requests a
        // few instructions old, a memory latency per chunk (in-kernel stamps of the K = 256 data gradients: "fetch" 23 % of the
        // loop).  A volatile use of the item's registers here keeps it on this side.
        // (the data gradients only: on the forward kernels the same pin is neutral to 16 % slower -- 96 -> 128 pooled, two workgroups per CU)
        if (DY) asm volatile("" : "+v"(raw.y[i].x), "+v"(raw.y[i].y), "+v"(raw.y[i].z), "+v"(raw.y[i].w));
        if (DY) asm volatile("" : "+v"(raw.z[DY && !ONEZ ? i : 0].x), "+v"(raw.z[DY && !ONEZ ? i : 0].y), "+v"(raw.z[DY && !ONEZ ? i : 0].z), "+v"(raw.z[DY && !ONEZ ? i : 0].w));
        if (POOLED) asm volatile("" : "+v"(raw.a[POOLED && !ONEZ ? i : 0].x), "+v"(raw.a[POOLED && !ONEZ ? i : 0].y), "+v"(raw.a[POOLED && !ONEZ ? i : 0].z), "+v"(raw.a[POOLED && !ONEZ ? i : 0].w));
        const float4 x = raw.y[i];
        float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DY) {
            dz = raw.z[DY && !ONEZ ? i : 0];
            if (POOLED) {
                const int4 a = raw.a[POOLED && !ONEZ ? i : 0];
                const int kk = (int)((tl * BM + (unsigned)row) % (unsigned)(PKP > 0 ? PKP : 1));
                dz.x = a.x == kk ? dz.x : 0.f; dz.y = a.y == kk ? dz.y : 0.f; dz.z = a.z == kk ? dz.z : 0.f; dz.w = a.w == kk ? dz.w : 0.f;
            }
        }
        // two channel pairs, one after the other (the expressions are bn_act's / dy_from's, value for value)
        unsigned hh[2], mm[2], ll[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float u0 = h ? x.z : x.x, u1 = h ? x.w : x.y;
            auto row2 = [&](const float4 &held) { return h ? make_float2(held.z, held.w) : make_float2(held.x, held.y); };
            if (MODE == MODE_BNRELU) {
                const float2 mu = row2(tb.a), sc = row2(tb.b), be = row2(tb.c);
                u0 = fmaxf(bn_act(u0, mu.x, sc.x, be.x), 0.f); u1 = fmaxf(bn_act(u1, mu.y, sc.y, be.y), 0.f);
            }
            if (DY) {
                const float2 c0 = row2(tb.a), q1 = row2(tb.b), q0 = row2(tb.c), mu = row2(tb.d);   // (dy_params_tab's rows)
                const float d0 = h ? dz.z : dz.x, d1 = h ? dz.w : dz.y;
                u0 = __builtin_fmaf(c0.x, d0, __builtin_fmaf(q1.x, u0 - mu.x, q0.x));
                u1 = __builtin_fmaf(c0.y, d1, __builtin_fmaf(q1.y, u1 - mu.y, q0.y));
            }
            if (k >= K4) { u0 = 0.f; u1 = 0.f; }
            split2(u0, u1, hh[h], mm[h], ll[h]);
        }
        const unsigned h0 = hh[0], h1 = hh[1], m0 = mm[0], m1 = mm[1], l0 = ll[0], l1 = ll[1];
        const unsigned o = lds_off(buf, row, q >> 1) + 8u * (unsigned)(q & 1);
        *reinterpret_cast<uint2 *>(lds_b + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(lds_b + o + IMG) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(lds_b + o + 2 * IMG) = make_uint2(l0, l1);
    };
    auto stage = [&](int64_t tile_, int c, int buf, const Tabs &tb) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) stage_item(tile_, c, buf, tb, i);
    };
    // the WG's tile sequence: pooling groups that span SUB tiles are walked tile by tile inside one workgroup
    auto tile_of = [&](int64_t seq) { return ((int64_t)blockIdx.x + (seq / SUB) * G) * SUB + (seq % SUB); };
    int64_t seq = 0;
    int64_t tile = tile_of(0);
    if (tile >= tiles) return;
    fetch(tile, 0);                                                 // (in flight under the W slice's loads)
    // ---- this lane's W slice: column n, k = 16 kb + 8 lh + 0 .. 7, three fragment sets (KS == 2: k blocks 2 ks, 2 ks + 1 of every chunk)
    // Every request of the slice is issued BEFORE the first one is used (clamped addresses and a select instead of a branch per
    // load: the first version waited for each of its 13 - 26 loads in turn -- 12 us of a 100 - 180 us launch, in-kernel stamps).
    SplitFrag wh[KBW], wm[KBW], wl[WL_LDS ? 1 : KBW];
    auto kb_of = [&](int kbw) { return KS == 1 ? kbw : (kbw >> 1) * 4 + 2 * ks + (kbw & 1); };
    if (BNN) {
        float v[KBW][8];
#pragma unroll
        for (int kbw = 0; kbw < KBW; ++kbw) {
            const int k0 = 16 * kb_of(kbw) + 8 * lh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = n < N && k0 + e < K;
                v[kbw][e] = g.W[ok ? (int64_t)(k0 + e) * g.ldw + n : 0];
                v[kbw][e] = ok ? v[kbw][e] : 0.f;
            }
        }
#pragma unroll
        for (int kbw = 0; kbw < KBW; ++kbw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) split2(v[kbw][2 * e], v[kbw][2 * e + 1], wh[kbw].u[e], wm[kbw].u[e], wl[WL_LDS ? 0 : kbw].u[e]);
            if (WL_LDS) wl_lds[kbw * 64] = wl[0].q;
        }
    } else {
        // W stored [N, K] (rows of 16-byte aligned quads: the launcher checks): a lane's values lie along a row, 32 lanes on 32
        // rows -- read straight from global that is 32 cache lines per instruction (4x over-fetch).  The wave copies 16 columns of
        // its 32 rows into a private LDS tile with coalesced reads (lane = (row, quad)) and picks its eight values from there.
        float *reg = wide_lds + wave * (32 * 20);                   // 32 rows x 16 floats, pitch 20: inside the operand buffers
        float4 x[KBW][2];
#pragma unroll
        for (int kbw = 0; kbw < KBW; ++kbw)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = lane + 64 * h, row = idx >> 2, q = idx & 3, kk = 16 * kb_of(kbw) + 4 * q, nn = cb * 32 + row;
                const bool ok = nn < N && kk + 3 < K;
                x[kbw][h] = ld4(g.W + (ok ? (int64_t)nn * g.ldw + kk : 0));
                if (!ok) x[kbw][h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int kbw = 0; kbw < KBW; ++kbw) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int idx = lane + 64 * h, row = idx >> 2, q = idx & 3;
                *reinterpret_cast<float4 *>(&reg[row * 20 + 4 * q]) = x[kbw][h];
            }
            // same wave, LDS operations complete in order: no barrier between these writes and reads
            const float4 a = *reinterpret_cast<const float4 *>(&reg[l31 * 20 + 8 * lh]), b = *reinterpret_cast<const float4 *>(&reg[l31 * 20 + 8 * lh + 4]);
            split2(a.x, a.y, wh[kbw].u[0], wm[kbw].u[0], wl[WL_LDS ? 0 : kbw].u[0]);
            split2(a.z, a.w, wh[kbw].u[1], wm[kbw].u[1], wl[WL_LDS ? 0 : kbw].u[1]);
            split2(b.x, b.y, wh[kbw].u[2], wm[kbw].u[2], wl[WL_LDS ? 0 : kbw].u[2]);
            split2(b.z, b.w, wh[kbw].u[3], wm[kbw].u[3], wl[WL_LDS ? 0 : kbw].u[3]);
            if (WL_LDS) wl_lds[kbw * 64] = wl[0].q;                  // (every wave of the column block writes the same values)
        }
    }
    if (KS == 1 && K4 % 16 != 0 && K4 % 16 <= 8) {
        // the last k block holds K4 % 16 contraction indices: the fragment registers past them are zero in EVERY lane -- say so,
        // and hipcc re-creates them where it needs them instead of keeping (196: 6) registers of zeros through the tile loop
#pragma unroll
        for (int e = (K4 % 16) / 2; e < 4; ++e) { wh[KBW - 1].u[e] = 0u; wm[KBW - 1].u[e] = 0u; if (!WL_LDS) wl[KBW - 1].u[e] = 0u; }
    }
    __syncthreads();                                                // the table is in place, the W tiles are read
    stage(tile, 0, 0, stage_tabs(0, tile));
    if (NCH > 1) fetch(tile, 1); else fetch(tile_of(1), 0);
    auto raw_landed = [&]() {                                       // a use of every raw register: hipcc waits for the requests HERE
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            asm volatile("" : "+v"(raw.y[i].x), "+v"(raw.y[i].y), "+v"(raw.y[i].z), "+v"(raw.y[i].w));
            if (DY && (!ONEZ || i == 0)) asm volatile("" : "+v"(raw.z[DY && !ONEZ ? i : 0].x), "+v"(raw.z[DY && !ONEZ ? i : 0].y), "+v"(raw.z[DY && !ONEZ ? i : 0].z), "+v"(raw.z[DY && !ONEZ ? i : 0].w));
            if (POOLED && (!ONEZ || i == 0)) asm volatile("" : "+v"(raw.a[POOLED && !ONEZ ? i : 0].x), "+v"(raw.a[POOLED && !ONEZ ? i : 0].y), "+v"(raw.a[POOLED && !ONEZ ? i : 0].z), "+v"(raw.a[POOLED && !ONEZ ? i : 0].w));
        }
    };
    raw_landed();                                                   // (the loop header then has nothing pending on either edge)
    int buf = 0;
    float mv[FPOOL > 0 ? (SUB > 1 ? 1 : BM / (FPOOL > 0 ? FPOOL : 1)) : 1];
    int mk[FPOOL > 0 ? (SUB > 1 ? 1 : BM / (FPOOL > 0 ? FPOOL : 1)) : 1];

    WABS(wave, 1)
    WSTAMP_DECL
    // (the tile loop is instantiated once per staging order: one body with a uniform branch around its two staging sites joins the
    // two sets of request registers with copies, and hipcc waits for the requests it has just issued to make them)
    auto tile_loop = [&](auto late_tag) {
    constexpr bool LATE = decltype(late_tag)::value;
    while (tile < tiles) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int64_t tile_next = tile_of(seq + 1);
        // PV_EARLY (mask epilogue): the sixteen values of the previous layer's output this lane masks with are requested at the
        // top of the tile's LAST chunk, a chunk of MFMAs ahead of the epilogue that waited for them (a memory latency per tile)
        constexpr bool PV_EARLY = EPI == EPI_MASK && TM / KS == 1 && PN2_SPLIT_PV_EARLY;
        float pvq[PV_EARLY ? 16 : 1];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            WSTAMP(5)
            __syncthreads();                                        // chunk c is staged in `buf`; every wave is done with buf ^ 1
            WSTAMP(0)
            const bool last = c == NCH - 1;
            if (PV_EARLY && last) {
                const unsigned row0 = (unsigned)tile * BM + rs * TM * 32 + 4 * lh;
                const float *pb = g.prevY + row_off(row0, g.ldp);
                const bool colv = NX || n < N4;                      // (no predicate around the request: a pad column re-reads column 0)
                unsigned offp = colv ? (unsigned)n : 0u;
                asm volatile("" : "+v"(offp));
                if (KS == 2) offp += (unsigned)(32 * ks) * (unsigned)g.ldp;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pvq[PV_EARLY ? r : 0] = pb[offp];
                    offp += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldp;
                }
            }
            const int64_t t1 = last ? tile_next : tile;             // staged now (fetched one step ago)
            const int c1 = last ? 0 : c + 1;
            const bool last1 = c1 == NCH - 1;
            const int64_t t2 = last1 ? (last ? tile_of(seq + 2) : tile_next) : t1;       // requested now
            const int c2 = last1 ? 0 : c1 + 1;
            // The two waves of a SIMD (w and w + NW / 2) take the interval's two jobs in OPPOSITE order: the first group stages the
            // next chunk (BatchNorm / dY transform, split, LDS stores: vector work) and then multiplies, the second multiplies first
            // -- one's staging runs beside the other's MFMAs instead of both queueing for the matrix pipe and then both idling it
            // (in-kernel stamps of the in-step version: MFMA phases 36 - 53 % of the loop, staging 16 - 27 %, barrier waits 34 %).
            const Tabs tb = stage_tabs(c1, t1 < tiles ? t1 : tiles - 1);     // (first in program order: its LDS reads lead the chunk)
            // (measured slower woven, in the step: KS == 2 -- two k blocks of twelve MFMAs per chunk and wave -- 291 -> 323 us; the
            // four-wave workgroups, two to a CU: 96 -> 128 pooled 304 -> 328, 64 -> 128 pooled 184 -> 193)
            constexpr bool WOVEN = PN2_SPLIT_ORDER >= 2 && KS == 1 && NW >= 6;
            if (!WOVEN && !LATE) {
                stage(t1, c1, buf ^ 1, tb);
                WSTAMP(1)
                fetch(t2, c2);
                WSTAMP(2)
            }
            constexpr int KBC = KC / 16, STEPS = KBC / KS;
            const int kbs = (KPAD - c * KC) < KC ? (KPAD - c * KC) / 16 : KBC;
#pragma unroll
            for (int kbl = 0; kbl < STEPS; ++kbl) {
                const int kb = KS == 1 ? kbl : 2 * ks + kbl;        // (KS == 2: this wave's two k blocks of the chunk)
                if (KS == 2 || kb < kbs) {
                    const int kg = KS == 1 ? c * KBC + kb : c * 2 + kbl;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = rs * TM * 32 + i * 32 + l31;
                        const unsigned o = lds_off(buf, row, 2 * kb + lh);
                        SplitFrag ah, am, al, wlo;
                        ah.q = *reinterpret_cast<const uint4 *>(lds_b + o);
                        am.q = *reinterpret_cast<const uint4 *>(lds_b + o + IMG);
                        al.q = *reinterpret_cast<const uint4 *>(lds_b + o + 2 * IMG);
                        if (WL_LDS) wlo.q = wl_lds[kg * 64]; else wlo = wl[WL_LDS ? 0 : kg];
                        // smallest terms first
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, wh[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wlo.v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wm[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, wh[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wm[kg].v, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wh[kg].v, acc[i], 0, 0, 0);
                    }
                }
                if (WOVEN) {
                    // The staging of the next chunk, item by item, between this chunk's k blocks: while this wave transforms and
                    // splits (vector pipe) the matrix pipe works through the MFMAs just issued and the other wave of the SIMD's.
                    // With the whole staging before (or after) the whole product the pipe idled through it: in-kernel stamps of
                    // that version had the MFMA phases at 35 - 57 % of the loop.  The image writes follow the operand reads in
                    // program order (hipcc cannot tell the two buffers apart and keeps LDS order), the request of an item goes out
                    // as soon as its registers are free; the scheduling barrier keeps hipcc from undoing the order.
#pragma unroll
                    for (int i = 0; i < A_IT; ++i)
                        if ((i * (kbs < STEPS ? kbs : STEPS)) / A_IT == kbl) {          // (over the k blocks the chunk HAS: a narrow last chunk)
                            stage_item(t1, c1, buf ^ 1, tb, i);
                            fetch_item(t2, c2, i);
                        }
                    if (PN2_SPLIT_ORDER == 3) {
                        constexpr int VEST = (MODE == MODE_PLAIN ? 30 : MODE == MODE_BNRELU ? 45 : MODE == MODE_DYDENSE ? 56 : 66) * ((A_IT + STEPS - 1) / STEPS);
                        constexpr int VPM = (VEST + 6 * TM - 1) / (6 * TM) < 7 ? (VEST + 6 * TM - 1) / (6 * TM) : 7;
#pragma unroll
                        for (int j = 0; j < 6 * TM; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            WSTAMP(3)
            if (!WOVEN && LATE) {
                stage(t1, c1, buf ^ 1, tb);
                WSTAMP(1)
                fetch(t2, c2);
                WSTAMP(2)
            }
            buf ^= 1;
        }
        if (KS == 2) {
            // the pair (ks = 0, 1) of a column block: hand the row block the OTHER wave finishes over, take the partner's share of one's own
            float *mine = xch + (size_t)wave * (16 * 64), *theirs = xch + (size_t)(wave ^ (NCB * RS)) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = ks == 0 ? acc[1][r] : acc[0][r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float o = theirs[r * 64 + lane];
                if (ks == 0) acc[0][r] += o; else acc[1][r] += o;
            }
        }
        // The requests of the next tile's first chunk are in flight (issued an interval ago).  gfx9 retires loads and stores
        // through ONE in-order counter and hipcc does not count across the epilogue's 32 stores: left alone it opens the next tile
        // with s_waitcnt vmcnt(0), i.e. with the HBM acknowledgement of every store of this tile (ISA of the first version).  A
        // use of the raw registers HERE makes it wait for the loads now -- they landed long ago -- and for nothing later.
        raw_landed();
        // ---- epilogue straight from the accumulators (the layout of v_mfma_f32_32x32x2_f32: column on the lane); instantiated
        // twice where N is ragged: the full column blocks take the copy without column predicates
        auto epilogue = [&](auto all_tag, auto store_tag) {
            constexpr bool ALLC = decltype(all_tag)::value;
            // STORE = false (pooled last layer only, g.Out == nullptr): Y never leaves the chip -- the statistics and the pooling
            // extrema come from the accumulators, and the layer's backward runs on its INPUT (split_bwd_res_kernel, CF)
            constexpr bool STORE = decltype(store_tag)::value;
            const unsigned row0 = (unsigned)tile * BM + rs * TM * 32 + 4 * lh;
            unsigned lo = (unsigned)n;
            asm volatile("" : "+v"(lo));
            float s0 = 0.f, s1 = 0.f;
            if (EPI == EPI_FWD) {
                float *yb = g.Out + row_off(row0, g.ldout);
                unsigned off = lo;
                constexpr int GPT = FPOOL > 0 ? (SUB > 1 ? 1 : BM / (FPOOL > 0 ? FPOOL : 1)) : 1;
                constexpr int BPG = FPOOL > 0 ? (SUB > 1 ? TM : FPOOL / 32) : TM;        // row blocks of this tile per group
                const int sg = __float_as_int(e1);
                const int sub = (int)(seq % SUB);
                if (FPOOL > 0 && sub == 0) {
#pragma unroll
                    for (int gq = 0; gq < GPT; ++gq) { mv[gq] = -INFINITY; mk[gq] = 0; }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        static_assert(KS == 1 || EPI != EPI_FWD, "the K split serves the data gradient only");
                        const float y = acc[i][r] + e0;
                        if (STORE && (ALLC || n < N4)) PN2_STREAM_STORE(y, yb + off);      // pad columns receive exact zeros (w = bias = 0)
                        s0 += y;
                        s1 = __builtin_fmaf(y, y, s1);
                        if (FPOOL > 0) {
                            // ascending tile, block, register = ascending row for this lane: a strict > keeps the first row
                            const int gq = SUB > 1 ? 0 : i / BPG;
                            const float yp = __int_as_float(__float_as_int(y) ^ sg);
                            const int row = sub * BM + (i - gq * BPG) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            mk[gq] = yp > mv[gq] ? row : mk[gq];
                            mv[gq] = fmaxf(mv[gq], yp);
                        }
                        off += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldout;   // rows (r & 3) + 8 (r >> 2): +1 +1 +1 +5
                    }
                if (FPOOL > 0 && sub == SUB - 1) {
#pragma unroll
                    for (int gq = 0; gq < GPT; ++gq) {
                        const float ov = __shfl_xor(mv[gq], 32, 64);   // the two half-waves hold disjoint rows of the column
                        const int ok = __shfl_xor(mk[gq], 32, 64);
                        const bool take = ov > mv[gq] || (ov == mv[gq] && ok < mk[gq]);
                        const float v = take ? ov : mv[gq];
                        const int k = take ? ok : mk[gq];
                        const int64_t grp = SUB > 1 ? tile / SUB : (int64_t)tile * GPT + gq;
                        if (ALLC || n < N4)
                            g.pool_rec[grp * g.pool_ld + lo] = make_float2(__int_as_float(__float_as_int(v) ^ sg), __int_as_float(k));
                    }
                }
            } else {
                const float *pb = g.prevY + row_off(row0, g.ldp);
                float *xb = g.Out + row_off(row0, g.ldout);
                unsigned offp = lo, offx = lo;
                asm volatile("" : "+v"(offx));
                if (KS == 2) {                                           // this wave finishes row block ks only
                    offp += (unsigned)(32 * ks) * (unsigned)g.ldp;
                    offx += (unsigned)(32 * ks) * (unsigned)g.ldout;
                }
#pragma unroll
                for (int i = 0; i < TM / KS; ++i) {
                    float pv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (PV_EARLY) pv[r] = (ALLC || n < N4) ? pvq[PV_EARLY ? r : 0] : 0.f;
                        else pv[r] = (ALLC || n < N4) ? pb[offp] : 0.f;
                        offp += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldp;
                    }
                    const unsigned sgn = SIGNED ? (unsigned)(tile & 1) << 31 : 0u;       // an odd tile was multiplied negated: its sign back
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float y = pv[r];
                        const float a = __uint_as_float(__float_as_uint(KS == 1 ? acc[i][r] : (ks == 0 ? acc[0][r] : acc[1][r])) ^ sgn);
                        const float dz = bn_act(y, e0, e1, e2) > 0.f ? a : 0.f;   // pad columns: scale = beta = 0 -> 0
                        if (ALLC || n < N4) PN2_STREAM_STORE(dz, xb + offx);
                        s0 += dz;
                        s1 = __builtin_fmaf(dz, (y - e0) * e3, s1);
                        offx += ((r & 3) == 3 ? 5u : 1u) * (unsigned)g.ldout;
                    }
                }
            }
            st0 += (double)s0; st1 += (double)s1;
        };
        if constexpr (EPI == EPI_FWD && FPOOL > 0) {
            if (g.Out == nullptr) { if (NX || all_cols) epilogue(pn2_true{}, pn2_false{}); else epilogue(pn2_false{}, pn2_false{}); }
            else if (NX || all_cols) epilogue(pn2_true{}, pn2_true{});
            else epilogue(pn2_false{}, pn2_true{});
        } else {
            if (NX || all_cols) epilogue(pn2_true{}, pn2_true{}); else epilogue(pn2_false{}, pn2_true{});
        }
        WSTAMP(4)
        ++seq;
        tile = tile_next;
    }
    };
    if (PN2_SPLIT_ORDER == 1 && late_stager) tile_loop(pn2_true{}); else tile_loop(pn2_false{});
    WSTAMP_FLUSH(wave)
    WABS(wave, 2)
    if (g.red != nullptr) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        // (the half-wave index taken afresh: kept from the top of the kernel it is one register too many across the loop of 196 -> 256)
        const int lh_end = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 5);
        if (lh_end == 0 && n < N) {
            double *rep = g.red + (size_t)(blockIdx.x % PN2_STAT_REPLICAS) * 2 * N;
            atomicAdd(rep + n, st0);
            atomicAdd(rep + N + n, st1);
        }
    }
    WABS(wave, 3)
}

template <int K4, int NN, int NCB, int RS, int TM, int MODE, int EPI, bool BNN, int PKP = 0, int KS = 1, int NG = 1>
int launch_split(const RegwArgs &g, hipStream_t s) {
    static_assert(NN <= 32 * NCB * NG && NN > 32 * (NCB * NG - NCB), "column blocks");
    constexpr int KC = 64, BM = 32 * TM * RS, KPAD = 16 * ((K4 + 15) / 16);
    constexpr int NTAB = MODE == MODE_PLAIN ? 0 : (MODE == MODE_BNRELU ? 3 : 4);
    constexpr int KBW = ((K4 + 15) / 16) / KS;
    constexpr bool SIGNED = (MODE == MODE_DYDENSE || MODE == MODE_DYPOOLED) && EPI == EPI_MASK;     // (two coefficient tables)
    constexpr size_t lds = 2 * 3 * (size_t)(BM * KC * 2) + sizeof(float) * (NTAB * KPAD * (SIGNED ? 2 : 1) + (KS == 2 ? NCB * RS * KS * 16 * 64 : 0)) +
                           (KBW >= 13 ? (size_t)NCB * KBW * 1024 : 0);               // (the kernel's WL_LDS region)
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = split_nt_kernel<K4, NCB, RS, TM, MODE, EPI, BNN, PKP, NN % 32 == 0 && NN == 32 * NCB * NG, KS>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    constexpr int SUB = (EPI == EPI_FWD && PKP > BM) ? PKP / BM : 1;
    // four-wave workgroups fit a CU twice (launch bounds, LDS): option SPLIT_WG2 lets the grid say so (two co-resident workgroups:
    // one's staging / epilogue under the other's MFMAs)
    const int64_t per_cu = (NCB * RS * KS <= 4 && pn2_opt(PN2_OPT_SPLIT_WG2)) ? 2 : 1;
    const int64_t cap = per_cu * pn2_num_cus() / NG, units = g.tiles / SUB;
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)(units < cap ? units : cap), NG), dim3(64 * NCB * RS * KS), lds, s, g);
    return pn2_launch_status();
}

}  // namespace

#ifdef PN2_STAMP
extern "C" int pn2_debug_stamps_wide_abs(unsigned long long *host_out, int n) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pn2_wide_abs_buf), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -2;
}
extern "C" int pn2_debug_stamps_wide(unsigned long long *host_out, int n) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pn2_wide_stamp_buf), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -2;
}
#endif

// ------------------------------------------------------------------------------------------------------------ dispatch
// Shapes with an instantiation (every one is a fully unrolled kernel of its own).  Rows: whole tiles of BM; the caller runs
// the remainder through the streamed kernels.  *bm_out = the tile height the instantiation uses.
//
// forward (K -> N):  128->128, 128->256, 128->196, 196->256         (sa2 of MSG, sa3 of SSG, the FP / head stacks)
// dgrad  (C_out -> C_in):  128->128, 256->128, 196->128, 256->196
#define PN2_WIDE_MIN_ROWS_DEFAULT 65536

int pn2_wide_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y, int ldy,
                 int64_t P, int K, int N, double *stats, LazyBn lz, hipStream_t s, int64_t *rows_done, int Kpool,
                 const float *pool_gamma, float *pool_ws) {
    const int on = pn2_opt(PN2_OPT_WIDE), min_rows = pn2_opt(PN2_OPT_WIDE_MIN_ROWS);
    *rows_done = 0;
    if (!on || P < min_rows || ldx != ((K + 3) & ~3)) return PN2_EUNSUPPORTED;
    // the LDS-DMA ring form (16-byte aligned rows; PN2_RING=0: the register-staged kernel, A/B runs)
    const int ring_on = pn2_opt(PN2_OPT_RING);
    const bool ring = ring_on && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    RegwArgs g{};
    g.lz = lz;
    g.A = X; g.lda = ldx; g.tab = in_affine; g.W = W; g.ldw = ldw; g.bias = bias; g.Out = Y; g.ldout = ldy; g.red = stats;
    g.K = K; g.N = N;
    g.pool_rec = reinterpret_cast<float2 *>(pool_ws); g.pool_gamma = pool_gamma; g.pool_ld = N;
    // fp32 products on the bf16 pipe (split_nt_kernel; option PN2_SPLIT): 64-row tiles (128 where the waves split the rows)
    if (pn2_opt(PN2_OPT_SPLIT) && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 && (ldw & 3) == 0) {
        const bool pool_ok = Kpool == 0 || (in_affine && pool_gamma && pool_ws && pn2_opt(PN2_OPT_WIDE_POOL) && P % 128 == 0);
#define SPLIT_FWD(KK, NN, NCB, RS, TM, PKP)                                                                              \
        if (K == KK && N == NN && Kpool == PKP && pool_ok && (Y != nullptr || PKP > 0)) {                                \
            constexpr int BM = 32 * TM * RS;                                                                             \
            g.tiles = P / BM;                                                                                            \
            if (PKP > 0) g.tiles = (P / (PKP > BM ? PKP : BM)) * ((PKP > BM ? PKP : BM) / BM);                           \
            *rows_done = g.tiles * BM;                                                                                   \
            if (g.tiles > 0) {                                                                                           \
                if (in_affine) return launch_split<((KK + 3) & ~3), NN, NCB, RS, TM, MODE_BNRELU, EPI_FWD, false, PKP>(g, s);  \
                if (PKP == 0) return launch_split<((KK + 3) & ~3), NN, NCB, RS, TM, MODE_PLAIN, EPI_FWD, false, 0>(g, s);    \
            }                                                                                                            \
        }
        if (P >= pn2_opt(PN2_OPT_SPLIT_MIN_ROWS_128)) { SPLIT_FWD(128, 128, 4, 2, 1, 0) }          // (eight waves with one row block each: 64-row tiles)
        SPLIT_FWD(128, 256, 8, 1, 2, 0)
        SPLIT_FWD(128, 196, 7, 1, 2, 0)
        SPLIT_FWD(196, 256, 8, 1, 2, 0)
        SPLIT_FWD(128, 256, 8, 1, 2, 64)
        SPLIT_FWD(128, 256, 8, 1, 2, 128)
        SPLIT_FWD(196, 256, 8, 1, 2, 128)
        if (pn2_opt(PN2_OPT_SPLIT_NARROW)) {
            // the sa1 layers of MSG-SemSeg that the weight-resident kernels served (same box, us: 64 -> 96 at 1 M rows 167 -> 138,
            // 96 -> 128 pooled over 64 270 -> 243, 64 -> 128 pooled over 64 95 -> 90; 64 -> 64 ties and stays where it was).  A last
            // layer is taken in its POOLED form only: its plain form must keep the arithmetic of the pooled forms this kernel does
            // not have (groups of 16 / 32), or the two would differ in the last bit (test_pool_in_gemm_epilogue_...)
            SPLIT_FWD(64, 96, 3, 2, 2, 0)
            SPLIT_FWD(96, 128, 4, 1, 2, 128)
            SPLIT_FWD(96, 128, 4, 1, 2, 64)
            SPLIT_FWD(64, 128, 4, 1, 2, 64)
        }
#undef SPLIT_FWD
        *rows_done = 0;
    }
    if (Y == nullptr) return PN2_EUNSUPPORTED;                      // (only the pooled bf16-pipe forms above run without an output)
    if (Kpool > 0) {
        // the last layer of a pooled MLP: whole tiles only (a pooled launch has no streamed tail), needs an input affine block
        const int pool_on = pn2_opt(PN2_OPT_WIDE_POOL);
        if (!pool_on || !in_affine || !pool_gamma || !pool_ws || P % 128 != 0) return PN2_EUNSUPPORTED;
#define WIDE_FWD_POOL(KK, NN, NCB, TM, PKP)                                                                              \
        if (K == KK && N == NN && Kpool == PKP) {                                                                        \
            static_assert(32 * TM == 128, "pooled forward: 128-row tiles");                                              \
            g.tiles = P / 128;                                                                                           \
            *rows_done = P;                                                                                              \
            if (ring) return launch_ring_fwd<((KK + 3) & ~3), NN, NCB, 1, TM, MODE_BNRELU, PKP>(g, s);                     \
            return launch_regw<((KK + 3) & ~3), NCB, 1, TM, 64, MODE_BNRELU, EPI_FWD, false, PKP>(g, s);                  \
        }
        WIDE_FWD_POOL(128, 256, 8, 4, 64)         // sa2 of MSG (K = 64), PointNet2ClsMsg
        WIDE_FWD_POOL(128, 256, 8, 4, 128)
        WIDE_FWD_POOL(196, 256, 8, 4, 128)        // sa2 of MSG (K = 128)
#undef WIDE_FWD_POOL
        return PN2_EUNSUPPORTED;
    }
#define WIDE_FWD(KK, NN, NCB, RS, TM, MINROWS)                                                                           \
    if (K == KK && N == NN && P >= MINROWS) {                                                                            \
        constexpr int BM = 32 * TM * RS;                                                                                 \
        g.tiles = P / BM;                                                                                                \
        *rows_done = g.tiles * BM;                                                                                       \
        if (ring && in_affine) return launch_ring_fwd<((KK + 3) & ~3), NN, NCB, RS, TM, MODE_BNRELU>(g, s);                \
        if (ring) return launch_ring_fwd<((KK + 3) & ~3), NN, NCB, RS, TM, MODE_PLAIN>(g, s);                              \
        if (in_affine) return launch_regw<((KK + 3) & ~3), NCB, RS, TM, 64, MODE_BNRELU, EPI_FWD, false>(g, s);          \
        return launch_regw<((KK + 3) & ~3), NCB, RS, TM, 64, MODE_PLAIN, EPI_FWD, false>(g, s);                          \
    }
    // (four waves per workgroup, one per SIMD with four row blocks each, measured SLOWER than two per SIMD with two blocks:
    // 128 -> 128 at 131 072 rows 53.2 vs 49.6 us, dgrad 196 -> 128 198 vs 178, 256 -> 128 106 vs 93)
    WIDE_FWD(128, 128, 4, 2, 2, 98304)        // 65 536 rows: 29.9 vs 28.2 us streamed; 131 072: 51 vs 54
    WIDE_FWD(128, 256, 8, 1, 4, 0)
    WIDE_FWD(128, 196, 7, 1, 4, 0)
    WIDE_FWD(196, 256, 8, 1, 4, 0)
#undef WIDE_FWD
    return PN2_EUNSUPPORTED;
}

int pn2_wide_dgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy,
                   const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev, const float *prev_affine,
                   float *dXout, int ldxo, double *prev_red, int64_t P, int K, int N, LazyCoef lc, hipStream_t s, int64_t *rows_done) {
    const int on = pn2_opt(PN2_OPT_WIDE), min_rows = pn2_opt(PN2_OPT_WIDE_MIN_ROWS);
    *rows_done = 0;
    if (!on || P < min_rows || ldy != ((K + 3) & ~3) || prev_Y == nullptr) return PN2_EUNSUPPORTED;
    if (!dZ && (Kpool <= 0 || P % Kpool != 0)) return PN2_EUNSUPPORTED;
    RegwArgs g{};
    g.lc = lc;
    g.A = Y; g.lda = ldy; g.dZ = dZ; g.ldz = ldz; g.dZp = dZp; g.arg = arg; g.ldo = ldo; g.kshift = 0; g.tab = coef;
    g.W = W; g.ldw = ldw; g.Out = dXout; g.ldout = ldxo; g.prevY = prev_Y; g.ldp = ld_prev; g.prev_aff = prev_affine; g.red = prev_red;
    g.K = K; g.N = N;
    if (pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_K256) && dZ == nullptr && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && ldo % 4 == 0) {
        // K = 256 (the pooled last layers of sa2): the contraction split over wave pairs, N = 196 as two column groups
#define SPLIT_DGRAD_P(KK, NN, NCB, NG, PKP)                                                                              \
        if (K == KK && N == NN && Kpool == PKP && P % 64 == 0) {                                                         \
            g.tiles = P / 64;                                                                                            \
            *rows_done = P;                                                                                              \
            return launch_split<KK, NN, NCB, 1, 2, MODE_DYPOOLED, EPI_MASK, true, PKP, 2, NG>(g, s);                     \
        }
        SPLIT_DGRAD_P(256, 128, 4, 1, 64)
        SPLIT_DGRAD_P(256, 128, 4, 1, 128)
        SPLIT_DGRAD_P(256, 196, 4, 2, 128)
        SPLIT_DGRAD_P(256, 196, 4, 2, 64)
#undef SPLIT_DGRAD_P
    }
    if (pn2_opt(PN2_OPT_SPLIT) && dZ != nullptr && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && (reinterpret_cast<uintptr_t>(dZ) & 15) == 0) {
        // fp32 products on the bf16 pipe, contraction lengths that fit the register-held W slice (K <= 208): the dense data gradients
#define SPLIT_DGRAD(KK, NN, NCB, RS, TM, MINROWS)                                                                        \
        if (K == KK && N == NN && P >= MINROWS) {                                                                        \
            constexpr int BM = 32 * TM * RS;                                                                             \
            g.tiles = P / BM;                                                                                            \
            *rows_done = g.tiles * BM;                                                                                   \
            return launch_split<((KK + 3) & ~3), NN, NCB, RS, TM, MODE_DYDENSE, EPI_MASK, true, 0>(g, s);                \
        }
        // (two raw streams per row: eight waves with one row block each keep the in-flight set at 16 registers)
        SPLIT_DGRAD(128, 128, 4, 2, 1, pn2_opt(PN2_OPT_SPLIT_MIN_ROWS_128))
        SPLIT_DGRAD(196, 128, 4, 2, 1, 0)
#undef SPLIT_DGRAD
    }
#define WIDE_DGRAD(KK, NN, NCB, RS, TM, KC, PKP, ADB, MINROWS)                                                           \
    if (K == KK && N == NN && P >= MINROWS && (PKP == 0 ? dZ != nullptr : (dZ == nullptr && Kpool == PKP))) {           \
        constexpr int BM = 32 * TM * RS;                                                                                 \
        g.tiles = P / BM;                                                                                                \
        *rows_done = g.tiles * BM;                                                                                       \
        return launch_regw<((KK + 3) & ~3), NCB, RS, TM, KC, PKP == 0 ? MODE_DYDENSE : MODE_DYPOOLED, EPI_MASK, true, PKP, ADB>(g, s); \
    }
    WIDE_DGRAD(128, 128, 4, 2, 2, 64, 0, true, 98304)      // 65 536 rows: ties the streamed kernel (33.9 vs 33.6 us)
    const int adb196 = pn2_opt(PN2_OPT_WIDE_ADB196);      // A/B: operand reads one k block ahead (two register sets)
    if (adb196) { WIDE_DGRAD(196, 128, 4, 2, 2, 64, 0, true, 0) }
    WIDE_DGRAD(196, 128, 4, 2, 2, 64, 0, false, 0)
    WIDE_DGRAD(256, 128, 4, 2, 1, 128, 64, true, 0)
    WIDE_DGRAD(256, 128, 4, 2, 1, 128, 128, true, 0)
    WIDE_DGRAD(256, 196, 7, 1, 2, 64, 128, true, 0)
    WIDE_DGRAD(256, 196, 7, 1, 2, 64, 64, true, 0)
#undef WIDE_DGRAD
    return PN2_EUNSUPPORTED;
}

// ================================================================================================ weight gradient
// dW[M, N] += sum_p dY[p, m] X[p, n] on the wide layers, with the WHOLE product resident: one workgroup keeps all of dW in
// the accumulator registers of its waves (wave w: output-channel block w x TNW input-channel blocks, 16 registers per 32 x 32
// tile) and streams its share of the P rows through LDS in chunks of 32 positions.  Against the streamed kernel of mlp.hip
// (128 x 128 tiles: dY re-read per N tile and X per M tile, 1.7x the algorithmic HBM traffic by the PMC counters of round 2,
// one barrier per 32 MFMAs):
//   * every operand row is read ONCE per launch (formed into dY / relu(bn(X)) once, by the thread that staged it);
//   * a barrier interval is 16 x TNW MFMAs per wave (112 at 256 x 196), the staging of the next chunk rides between the
//     position pairs of this one (see regw_nt_kernel);
//   * the price is one full set of dW atomics per workgroup (256 x M x N floats at the memory side's ~1.3 TB/s: 39 us at
//     256 x 196) -- it is a tail, not a rate.
namespace {

struct WgradArgs {
    const float *Y; int ldy;                                       // this layer's pre-BN output [P, ldy]
    const float *dZ; int ldz;                                      // dense dZ, or
    const float *dZp; const int32_t *arg; int ldo;                 // pooled: [G, ldo], group size PKP (template)
    const float *coef;                                             // c0, q1, q0, mean rows of pitch M4
    const float *X; int ldx; const float *x_aff;                   // layer input [P, ldx]; its affine block (pitch N4) or null
    float *dW; int lddw; float *dbias;
    int64_t P; int64_t rows_per_wg; int M; int N;
    LazyCoef lc;                                                   // `coef` is filled by the prologue (a layer without a data gradient)
    float *ws;                                                     // two-phase flush: partial slabs [workgroup][MB * 32][NB * 32], or null
};

template <int MM, int NN, int TNW, int DYM, bool XACT, int PKP, int BP, bool BIAS>
__global__ __launch_bounds__(64 * ((MM + 31) / 32) * (((NN + 31) / 32) / TNW)) void wgrad_full_kernel(const WgradArgs g) {
    constexpr int MB = (MM + 31) / 32, NB = (NN + 31) / 32;
    constexpr int WN = NB / TNW, NW = MB * WN, NT = 64 * NW;
    constexpr int LDA = MB * 32 + 4, LDB = NB * 32 + 4;            // LDS row pitches (floats): the two half-waves of a b32
                                                                   // operand read sit one row apart = 4 banks apart
    constexpr bool POOLED = DYM == MODE_DYPOOLED;
    static_assert(NB % TNW == 0 && NW <= 16, "wave grid");
    float *As0 = wide_lds, *As1 = As0 + BP * LDA, *Bs0 = As1 + BP * LDA, *Bs1 = Bs0 + BP * LDB;
    float *ctab = Bs1 + BP * LDB;                                  // 4 rows of MB * 32: c0, q1, q0, mean
    float *xtab = ctab + 4 * MB * 32;                              // 3 rows of NB * 32: mean, scale, beta of the input BatchNorm
    float *dump = xtab + 3 * NB * 32;                              // NT float4: where the dead items of a partly filled pass store
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int mb = wave / WN, nb0 = (wave % WN) * TNW;
    constexpr int M = MM, N = NN, M4 = (M + 3) & ~3, N4 = (N + 3) & ~3;
    constexpr int QA = M4 / 4, QB = N4 / 4;                        // quads per row that exist in memory

    lazy_coef_prologue(g.lc);
    // ---- one-time: tables (zero beyond the real channels), zero pad columns of both chunk buffers
    for (int i = t; i < 4 * MB * 32; i += NT) { const int r = i / (MB * 32), c = i - r * (MB * 32); ctab[i] = c < M4 ? g.coef[r * M4 + c] : 0.f; }
    if (XACT)
        for (int i = t; i < 3 * NB * 32; i += NT) { const int r = i / (NB * 32), c = i - r * (NB * 32); xtab[i] = c < N4 ? g.x_aff[r * N4 + c] : 0.f; }
    for (int i = t; i < 2 * BP * LDA; i += NT) As0[i] = 0.f;       // (As0, As1 contiguous; Bs0, Bs1 contiguous)
    for (int i = t; i < 2 * BP * LDB; i += NT) Bs0[i] = 0.f;

    f32x16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int64_t p_begin = (int64_t)blockIdx.x * g.rows_per_wg;
    const int64_t p_end = p_begin + g.rows_per_wg < g.P ? p_begin + g.rows_per_wg : g.P;
    // staging items: dY quads idx = t + NT i over BP x QA, X quads over BP x QB (run-time QA / QB: the loops below are
    // unrolled to the template bounds and predicated)
    constexpr int ITA = (BP * QA + NT - 1) / NT, ITB = (BP * QB + NT - 1) / NT;
    static_assert(!POOLED || NT % QA == 0, "pooled: a thread's dY items share one channel quad");
    struct Raw { float4 y[ITA]; float4 z[POOLED ? 1 : ITA]; int4 a[1]; float4 x[ITB]; };
    Raw raw;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // Pooled dZ: NT % QA == 0 is required by the host (every item of a thread sits in the same channel quad) and the chunk
    // lies inside one pooling group (BP <= PKP, chunks aligned to BP): one (dZp, arg) quad per thread and chunk.
    //
    // Addresses: a chunk's rows start at a WAVE-UNIFORM base (p0: scalar arithmetic); what a lane adds -- row * pitch + 4 q of
    // its items -- never changes, so it lives in ITA + ITB registers (one each where NT % QA == 0 / NT % QB == 0: the items of a
    // thread are then whole row steps apart and the step folds into the base) instead of being re-derived per request
    // (~25 vector instructions each: round-4 ablation, the requests alone were 11 - 19 % of these kernels).
    constexpr bool A_INV = NT % QA == 0, B_INV = NT % QB == 0;
    constexpr int RPA = NT / QA, RPB = NT / QB;                    // row steps (A_INV / B_INV)
    unsigned offA[A_INV ? 1 : ITA], ldsA[A_INV ? 1 : ITA], q4A[A_INV ? 1 : ITA], offB[B_INV ? 1 : ITB], ldsB[B_INV ? 1 : ITB], q4B[B_INV ? 1 : ITB];
#pragma unroll
    for (int i = 0; i < (A_INV ? 1 : ITA); ++i) {
        const int idx = t + NT * i, row = idx / QA, q = idx - row * QA;
        offA[i] = (unsigned)row * (unsigned)g.ldy + 4u * q;
        ldsA[i] = (unsigned)row * LDA + 4u * q;
        q4A[i] = 4u * q;
    }
#pragma unroll
    for (int i = 0; i < (B_INV ? 1 : ITB); ++i) {
        const int idx = t + NT * i, row = idx / QB, q = idx - row * QB;
        offB[i] = (unsigned)row * (unsigned)g.ldx + 4u * q;
        ldsB[i] = (unsigned)row * LDB + 4u * q;
        q4B[i] = 4u * q;
    }
    const unsigned qa4 = 4u * (unsigned)(t % QA);                  // (A_INV: the thread's channel quad)
    // item i of a chunk: lane part of its global offset / of its LDS offset, and whether its row lies inside `rows` rows
    auto a_off = [&](int i) { return A_INV ? offA[0] + (unsigned)(i * RPA) * (unsigned)g.ldy : offA[i]; };
    auto a_lds = [&](int i) { return A_INV ? ldsA[0] + (unsigned)(i * RPA * LDA) : ldsA[i]; };
    auto a_live = [&](int i, int rows) { return A_INV ? (int)(t / QA) + i * RPA < rows : ldsA[i] < (unsigned)rows * LDA; };
    auto b_off = [&](int i) { return B_INV ? offB[0] + (unsigned)(i * RPB) * (unsigned)g.ldx : offB[i]; };
    auto b_lds = [&](int i) { return B_INV ? ldsB[0] + (unsigned)(i * RPB * LDB) : ldsB[i]; };
    auto b_live = [&](int i, int rows) { return B_INV ? (int)(t / QB) + i * RPB < rows : ldsB[i] < (unsigned)rows * LDB; };
    auto rows_of = [&](int64_t p0) { const int64_t left = p_end - p0; return (int)(left < 0 ? 0 : (left < BP ? left : BP)); };
    auto fetch = [&](int64_t p0) {
        const int rows = rows_of(p0);                              // uniform; < BP only in the last chunk of the last workgroup
        const int64_t pb = rows > 0 ? p0 : p_begin;                // past the end: every item is dead and re-reads a valid row
        const float *yb = g.Y + (size_t)pb * (unsigned)g.ldy, *zb = g.dZ + (size_t)pb * (unsigned)g.ldy;    // (ldz == ldy: host)
        const float *xb = g.X + (size_t)pb * (unsigned)g.ldx;
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            const unsigned o = a_live(i, rows) ? a_off(i) : (A_INV ? qa4 : 0u);    // a dead item reads row 0 of the chunk (zeroed / dumped when staged)
            raw.y[i] = ld4(yb + o);
            if (!POOLED) raw.z[POOLED ? 0 : i] = ld4(zb + o);
        }
        if (POOLED) {
            const size_t go = (size_t)(pb / (PKP > 0 ? PKP : 1)) * (unsigned)g.ldo;
            raw.z[0] = ld4(g.dZp + go + qa4);
            raw.a[0] = ld4i(g.arg + go + qa4);
        }
#pragma unroll
        for (int i = 0; i < ITB; ++i) raw.x[i] = ld4(xb + (b_live(i, rows) ? b_off(i) : 0u));
    };
    // The BatchNorm-backward coefficient rows of the thread's channel quad never change where A_INV holds: read from the table
    // once (behind the first barrier), likewise the input's BatchNorm constants where B_INV holds.
    DyParams dpk;
    float4 xk_mu, xk_sc, xk_be;
    // (straight-line on purpose: an early return or a uniform branch around the staging would put it into basic blocks of
    // its own, and the scheduler could not slide its VALU work between the MFMAs of the position pairs)
    auto stage_a = [&](float *Ad, int64_t p0, int i) {
        const int rows = rows_of(p0);
        const bool live = a_live(i, rows);
        const unsigned lo = a_lds(i);
        const DyParams dp = A_INV ? dpk : dy_params_tab(ctab, MB * 32, (int)q4A[A_INV ? 0 : i], true);
        float4 dz = raw.z[POOLED ? 0 : i];
        if (POOLED) {
            const int4 a = raw.a[0];
            const int kk = (int)((unsigned)p0 & (unsigned)(PKP - 1)) + (int)(t / QA) + i * RPA;    // the chunk lies inside one group
            dz.x = a.x == kk ? dz.x : 0.f; dz.y = a.y == kk ? dz.y : 0.f;
            dz.z = a.z == kk ? dz.z : 0.f; dz.w = a.w == kk ? dz.w : 0.f;
        }
        float4 v = dy_from(dz, raw.y[i], dp);
        if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>((BP * QA % NT == 0 || lo < (unsigned)(BP * LDA)) ? &Ad[lo] : &dump[4 * t]) = v;
        if (BIAS) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
    };
    auto stage_b = [&](float *Bd, int64_t p0, int i) {
        const int rows = rows_of(p0);
        const bool live = b_live(i, rows);
        const unsigned lo = b_lds(i);
        float4 x = raw.x[i];
        if (XACT) {
            const unsigned q4 = q4B[B_INV ? 0 : i];
            const float4 mu = B_INV ? xk_mu : *reinterpret_cast<const float4 *>(&xtab[q4]);
            const float4 sc = B_INV ? xk_sc : *reinterpret_cast<const float4 *>(&xtab[NB * 32 + q4]);
            const float4 be = B_INV ? xk_be : *reinterpret_cast<const float4 *>(&xtab[2 * NB * 32 + q4]);
            x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f);
            x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
            x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f);
            x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
        }
        if (!live) x = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>((BP * QB % NT == 0 || lo < (unsigned)(BP * LDB)) ? &Bd[lo] : &dump[4 * t]) = x;
    };

    float *Ac = As0, *An = As1, *Bc = Bs0, *Bn = Bs1;
    if (p_begin < p_end) {
        fetch(p_begin);
        __syncthreads();                                            // tables and the zeroed buffers
        if (A_INV) dpk = dy_params_tab(ctab, MB * 32, (int)qa4, true);
        if (B_INV && XACT) {
            xk_mu = *reinterpret_cast<const float4 *>(&xtab[q4B[0]]);
            xk_sc = *reinterpret_cast<const float4 *>(&xtab[NB * 32 + q4B[0]]);
            xk_be = *reinterpret_cast<const float4 *>(&xtab[2 * NB * 32 + q4B[0]]);
        }
#pragma unroll
        for (int i = 0; i < ITA; ++i) stage_a(Ac, p_begin, i);
#pragma unroll
        for (int i = 0; i < ITB; ++i) stage_b(Bc, p_begin, i);
        fetch(p_begin + BP);                                        // (past the end: dead items, never staged as data)
    }
    for (int64_t p0 = p_begin; p0 < p_end; p0 += BP) {
        __syncthreads();                                            // chunk p0 is in (Ac, Bc); every wave is done with (An, Bn)
        const float *ap = Ac + lh * LDA + mb * 32 + l31;
        const float *bp = Bc + lh * LDB + nb0 * 32 + l31;
        float a[2], b[2][TNW];
        a[0] = ap[0];
#pragma unroll
        for (int j = 0; j < TNW; ++j) b[0][j] = bp[32 * j];
#pragma unroll
        for (int s = 0; s < BP / 2; ++s) {
            asm volatile("" ::: "memory");
            if (s + 1 < BP / 2) {
                a[(s + 1) & 1] = ap[2 * (s + 1) * LDA];
#pragma unroll
                for (int j = 0; j < TNW; ++j) b[(s + 1) & 1][j] = bp[2 * (s + 1) * LDB + 32 * j];
            }
            // the next chunk's staging rides behind the position pairs: dY items first, then X items, then the request for
            // the chunk after it
            // (always: past the end the items carry zeros, and nobody reads that chunk)
#pragma unroll
            for (int i = 0; i < ITA; ++i)
                if (s == i) stage_a(An, p0 + BP, i);
#pragma unroll
            for (int i = 0; i < ITB; ++i)
                if (s == ITA + i) stage_b(Bn, p0 + BP, i);
            if (s == ITA + ITB) fetch(p0 + 2 * BP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1], b[s & 1][j], acc[j], 0, 0, 0);
        }
        float *tp = Ac; Ac = An; An = tp;
        tp = Bc; Bc = Bn; Bn = tp;
    }
    static_assert(ITA + ITB < BP / 2, "the staging must fit between the position pairs of a chunk");

    // ---- flush: every accumulator register is 2 x 128 contiguous bytes of dW
    const int m_base = mb * 32 + 4 * lh;
    if (g.ws != nullptr) {
        // two-phase form (round 4, built to be measured): the workgroup's whole slab leaves as PLAIN stores (padded tile grid: no
        // predicates), wgrad_reduce_kernel adds the slabs up and issues one atomic per element -- 256x fewer atomics, the same
        // bytes written once and read once
        float *slab = g.ws + (size_t)blockIdx.x * (MB * 32) * (NB * 32);
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = (nb0 + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(m_base + (r & 3) + 8 * (r >> 2)) * (NB * 32) + n] = acc[j][r];
        }
    } else {
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = (nb0 + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + (r & 3) + 8 * (r >> 2);
                if (m < M && n < N) atomicAdd(g.dW + (int64_t)m * g.lddw + n, acc[j][r]);
            }
        }
    }
    if (BIAS && g.dbias != nullptr) {                               // fold the row groups of each channel quad in LDS
        __syncthreads();
        float *sh = wide_lds;
        *reinterpret_cast<float4 *>(&sh[t * 4]) = bsum;
        __syncthreads();
        static_assert(!BIAS || NT % QA == 0, "bias fold: thread t + QA k holds quad t");
        if (t < QA) {
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = t; k < NT; k += QA) {
                const float4 v = *reinterpret_cast<const float4 *>(&sh[k * 4]);
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            const int m = 4 * t;
            if (m < M) atomicAdd(g.dbias + m, tot.x);
            if (m + 1 < M) atomicAdd(g.dbias + m + 1, tot.y);
            if (m + 2 < M) atomicAdd(g.dbias + m + 2, tot.z);
            if (m + 3 < M) atomicAdd(g.dbias + m + 3, tot.w);
        }
    }
}

// Second phase of the two-phase flush: dW[m, n] += sum over the workgroups' slabs.  grid (element blocks, slab ranges): a
// thread owns one float4 of one row and a range of slabs (coalesced 16-byte reads, slab pitch apart), then four atomics.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ ws, int slabs, int rows, int ldn, int M, int N,
                                                           float *__restrict__ dW, int lddw) {
    const int q = blockIdx.x * 256 + threadIdx.x, qpr = ldn >> 2;
    if (q >= rows * qpr) return;
    const int m = q / qpr, n = (q - m * qpr) * 4;
    const int per = (slabs + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = w0 + per < slabs ? w0 + per : slabs;
    const size_t pitch = (size_t)rows * ldn;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    const float *p = ws + (size_t)w0 * pitch + (size_t)m * ldn + n;
    int w = w0;
    for (; w + 1 < w1; w += 2, p += 2 * pitch) {
        const float4 u = ld4(p), v = ld4(p + pitch);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    if (w < w1) { const float4 u = ld4(p); a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    if (m < M) {
        float *d = dW + (int64_t)m * lddw + n;
        if (n < N) atomicAdd(d, a.x + b.x);
        if (n + 1 < N) atomicAdd(d + 1, a.y + b.y);
        if (n + 2 < N) atomicAdd(d + 2, a.z + b.z);
        if (n + 3 < N) atomicAdd(d + 3, a.w + b.w);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, bf16 pipe
// dW[M, N] += sum_p dY[p, m] X[p, n] with fp32 products formed from exact three-way bf16 splits (see split_nt_kernel): the
// whole dW in the accumulators of one workgroup as in wgrad_full_kernel (wave: one 32-row block of dW x TNW 32-column blocks),
// the P rows in chunks of 16 = ONE v_mfma_f32_32x32x16_bf16 contraction block.  Both operands are needed "down the columns"
// (eight consecutive rows p of one channel per lane): the staging pass forms dY / relu(bn(X)) exactly as the fp32 kernel does,
// splits, and stores three bf16 images per operand as plain rows [p][128 channels] per 128-channel panel with the chunk XOR of
// the programming guide's dual-use image (b) -- off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) --
// and the fragments come back through ds_read_b64_tr_b16 (the hardware's 4 x 16 transpose: lane 4 q + p of a 16-lane group
// supplies the address of block row q, elements 4 p .. 4 p + 3; lane i receives column i): two reads per fragment, conflict-free.
template <int MM, int NN, int TNW, int DYM, int PKP, int BP>          // BP: rows per chunk = BP / 16 contraction blocks
__global__ __launch_bounds__(64 * ((MM + 31) / 32) * (((NN + 31) / 32) / TNW)) void split_tn_kernel(const WgradArgs g) {
    constexpr int MB = (MM + 31) / 32, NB = (NN + 31) / 32, WN = NB / TNW, NW = MB * WN, NT = 64 * NW;
    static_assert(BP == 16 || BP == 32, "chunk rows");
    constexpr int M4 = (MM + 3) & ~3, N4 = (NN + 3) & ~3, QA = M4 / 4, QB = N4 / 4;
    constexpr int PA = (MB * 32 + 127) / 128, PB = (NB * 32 + 127) / 128;      // 128-channel panels
    constexpr int PANEL = BP * 256;                                 // bytes of one panel of one piece
    constexpr int IMG_A = PA * PANEL, IMG_B = PB * PANEL, BUF = 3 * (IMG_A + IMG_B);
    constexpr bool POOLED = DYM == MODE_DYPOOLED;
    static_assert(NB % TNW == 0 && NW <= 16, "wave grid");
    static_assert(!POOLED || (NT % QA == 0 && PKP % BP == 0), "pooled: one channel quad per thread, chunks inside a group");
    constexpr int PBLK = BP / 16; (void)PBLK;
    unsigned char *lds_b = reinterpret_cast<unsigned char *>(wide_lds);
    float *ctab = wide_lds + (2 * BUF) / 4;                         // 4 rows of MB * 32: c0, q1, q0, mean
    float *xtab = ctab + 4 * MB * 32;                               // 3 rows of NB * 32: mean, scale, beta of the input BatchNorm
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, lh = lane >> 5;
    const int mb = wave / WN, nb0 = (wave % WN) * TNW;

    lazy_coef_prologue(g.lc);
    for (int i = t; i < 4 * MB * 32; i += NT) { const int r = i / (MB * 32), c = i - r * (MB * 32); ctab[i] = c < M4 ? g.coef[r * M4 + c] : 0.f; }
    for (int i = t; i < 3 * NB * 32; i += NT) { const int r = i / (NB * 32), c = i - r * (NB * 32); xtab[i] = c < N4 ? g.x_aff[r * N4 + c] : 0.f; }
    // zero both buffers once: the channels past M4 / N4 of the last panels are never written and must read as zeros
    for (int i = t; i < 2 * BUF / 16; i += NT) reinterpret_cast<uint4 *>(lds_b)[i] = make_uint4(0u, 0u, 0u, 0u);

    f32x16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int64_t p_begin = (int64_t)blockIdx.x * g.rows_per_wg;
    const int64_t p_end = p_begin + g.rows_per_wg < g.P ? p_begin + g.rows_per_wg : g.P;
    constexpr int ITA = (BP * QA + NT - 1) / NT, ITB = (BP * QB + NT - 1) / NT;
    struct Raw { float4 y[ITA]; float4 z[POOLED ? 1 : ITA]; int4 a[1]; float4 x[ITB]; };
    Raw raw;
    // a float4 of channels 4 q .. 4 q + 3 of row `row`: three 8-byte stores (4 bf16 each) into the piece images
    auto store_split = [&](unsigned char *img, int img_bytes, int row, int q, float4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0, l0);
        split2(v.z, v.w, h1, m1, l1);
        const int col = 4 * q, panel = col >> 7, ch = (col & 127) >> 3;
        const unsigned o = (unsigned)(panel * PANEL) + tr_img_off(row, ch) + 8u * (unsigned)((col >> 2) & 1);
        *reinterpret_cast<uint2 *>(img + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(img + o + img_bytes) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(img + o + 2 * img_bytes) = make_uint2(l0, l1);
    };
    // ---- staging item by item, branch-free (a thread without an i-th item repeats its last: the same value to the same place; no
    // lane-dependent predicate around a request).  In the loop one item of the next chunk -- transform, split, image writes, then
    // its request two chunks ahead -- follows every tile's MFMAs (split_nt_kernel's order: the matrix pipe works through the
    // tile while the wave stages).
    auto a_it = [&](int i, int &row, int &q) { const int idx = (NT * (i + 1) > BP * QA) ? min(t + NT * i, BP * QA - 1) : t + NT * i; row = idx / QA; q = idx - row * QA; };
    auto b_it = [&](int i, int &row, int &q) { const int idx = (NT * (i + 1) > BP * QB) ? min(t + NT * i, BP * QB - 1) : t + NT * i; row = idx / QB; q = idx - row * QB; };
    auto fetch_a = [&](int i, int64_t p0) {
        const int64_t pb = p0 < p_end ? p0 : p_begin;
        int row, q;
        a_it(i, row, q);
        raw.y[i] = ld4(g.Y + (size_t)(pb + row) * (unsigned)g.ldy + 4 * q);
        if (!POOLED) raw.z[POOLED ? 0 : i] = ld4(g.dZ + (size_t)(pb + row) * (unsigned)g.ldy + 4 * q);
        if (POOLED && i == ITA - 1) {                               // (one quad per chunk: behind the LAST item that used the previous one)
            const size_t go = (size_t)(pb / (PKP > 0 ? PKP : 1)) * (unsigned)g.ldo + 4u * (unsigned)(t % QA);
            raw.z[0] = ld4(g.dZp + go);
            raw.a[0] = ld4i(g.arg + go);
        }
    };
    auto fetch_b = [&](int i, int64_t p0) {
        const int64_t pb = p0 < p_end ? p0 : p_begin;
        int row, q;
        b_it(i, row, q);
        raw.x[i] = ld4(g.X + (size_t)(pb + row) * (unsigned)g.ldx + 4 * q);
    };
    auto stage_a = [&](int i, int64_t p0, int buf) {
        unsigned char *ia = lds_b + buf * BUF;
        int row, q;
        a_it(i, row, q);
        const DyParams dp = dy_params_tab(ctab, MB * 32, 4 * q, true);
        float4 dz = raw.z[POOLED ? 0 : i];
        if (POOLED) {
            const int4 a = raw.a[0];
            const int kk = (int)((unsigned)p0 & (unsigned)(PKP - 1)) + row;
            dz.x = a.x == kk ? dz.x : 0.f; dz.y = a.y == kk ? dz.y : 0.f; dz.z = a.z == kk ? dz.z : 0.f; dz.w = a.w == kk ? dz.w : 0.f;
        }
        float4 v = dy_from(dz, raw.y[i], dp);
        if (!(p0 < p_end)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        store_split(ia, IMG_A, row, q, v);
    };
    auto stage_b = [&](int i, int64_t p0, int buf) {
        unsigned char *ib = lds_b + buf * BUF + 3 * IMG_A;
        int row, q;
        b_it(i, row, q);
        const float4 mu = *reinterpret_cast<const float4 *>(&xtab[4 * q]), sc = *reinterpret_cast<const float4 *>(&xtab[NB * 32 + 4 * q]);
        const float4 be = *reinterpret_cast<const float4 *>(&xtab[2 * NB * 32 + 4 * q]);
        float4 x = raw.x[i];
        x.x = fmaxf(bn_act(x.x, mu.x, sc.x, be.x), 0.f); x.y = fmaxf(bn_act(x.y, mu.y, sc.y, be.y), 0.f);
        x.z = fmaxf(bn_act(x.z, mu.z, sc.z, be.z), 0.f); x.w = fmaxf(bn_act(x.w, mu.w, sc.w, be.w), 0.f);
        if (!(p0 < p_end)) x = make_float4(0.f, 0.f, 0.f, 0.f);
        store_split(ib, IMG_B, row, q, x);
    };
    auto fetch = [&](int64_t p0) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) fetch_a(i, p0);
#pragma unroll
        for (int i = 0; i < ITB; ++i) fetch_b(i, p0);
    };
    auto stage = [&](int64_t p0, int buf) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) stage_a(i, p0, buf);
#pragma unroll
        for (int i = 0; i < ITB; ++i) stage_b(i, p0, buf);
    };
    // transposed fragment: channels cblk * 32 + l31, rows 16 pb + 8 lh + 0 .. 7 of the chunk, of one piece image
    const int g16 = lane >> 4, j16 = lane & 15, tq = j16 >> 2, tp = j16 & 3;
    auto frag = [&](const unsigned char *img, int cblk, int pb) {
        const int col0 = cblk * 32 + 16 * (g16 & 1), panel = col0 >> 7, c0 = (col0 & 127) >> 3;
        SplitFrag f;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 16 * pb + 8 * (g16 >> 1) + 4 * r2 + tq;
            const unsigned o = (unsigned)(panel * PANEL) + tr_img_off(row, c0 + (tp >> 1)) + 8u * (unsigned)(tp & 1);
            const pn2_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                reinterpret_cast<__attribute__((address_space(3))) pn2_s16x4 *>((__attribute__((address_space(3))) unsigned char *)(img) + o));
            const uint2 u = __builtin_bit_cast(uint2, v);
            f.u[2 * r2] = u.x; f.u[2 * r2 + 1] = u.y;
        }
        return f;
    };

    WABS(wave, 0)
    if (p_begin < p_end) {
        __syncthreads();                                            // tables, zeroed images
        fetch(p_begin);
        stage(p_begin, 0);
        fetch(p_begin + BP);
        int buf = 0;
        WABS(wave, 1)
        WSTAMP_DECL
        for (int64_t p0 = p_begin; p0 < p_end; p0 += BP) {
            WSTAMP(5)
            __syncthreads();                                        // chunk p0 is in `buf`; every wave is done with buf ^ 1
            WSTAMP(0)
            if (!PN2_SPLIT_TN_WOVEN) {
                stage(p0 + BP, buf ^ 1);
                WSTAMP(1)
                fetch(p0 + 2 * BP);
                WSTAMP(2)
            }
            const unsigned char *ia = lds_b + buf * BUF, *ib = ia + 3 * IMG_A;
#pragma unroll
            for (int pb = 0; pb < BP / 16; ++pb) {
                const SplitFrag ah = frag(ia, mb, pb), am = frag(ia + IMG_A, mb, pb), al = frag(ia + 2 * IMG_A, mb, pb);
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    const SplitFrag bh = frag(ib, nb0 + j, pb), bm = frag(ib + IMG_B, nb0 + j, pb), bl = frag(ib + 2 * IMG_B, nb0 + j, pb);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bm.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am.v, bh.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bm.v, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh.v, acc[j], 0, 0, 0);
                    if (PN2_SPLIT_TN_WOVEN) {
                        constexpr int NIT = ITA + ITB, STEPS = (BP / 16) * TNW;
                        const int step = pb * TNW + j;
#pragma unroll
                        for (int i = 0; i < NIT; ++i)
                            if ((i * STEPS) / NIT == step) {
                                if (i < ITA) { stage_a(i, p0 + BP, buf ^ 1); fetch_a(i, p0 + 2 * BP); }
                                else { stage_b(i - ITA, p0 + BP, buf ^ 1); fetch_b(i - ITA, p0 + 2 * BP); }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            WSTAMP(3)
            buf ^= 1;
        }
        if (wave < 8) { WSTAMP_FLUSH(wave) }
        WABS(wave, 2)
    }
    // ---- flush: every accumulator register is 2 x 128 contiguous bytes of dW -- atomics, or (caller scratch: pn2_conv1x1_wgrad_ws)
    // the workgroup's slab as plain stores for wgrad_reduce_kernel to add up (in-kernel stamps: the loop is 180 us of this
    // kernel's 260 at 256 x 196 -- the rest is 256 workgroups x 50 176 device-scope atomics)
    const int m_base = mb * 32 + 4 * lh;
    if (g.ws != nullptr) {
        float *slab = g.ws + (size_t)blockIdx.x * (MB * 32) * (NB * 32);
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = (nb0 + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(m_base + (r & 3) + 8 * (r >> 2)) * (NB * 32) + n] = acc[j][r];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = (nb0 + j) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + (r & 3) + 8 * (r >> 2);
            if (m < MM && n < NN) atomicAdd(g.dW + (int64_t)m * g.lddw + n, acc[j][r]);
        }
    }
}

template <int MM, int NN, int TNW, int DYM, int PKP, int BP>
int launch_split_tn(WgradArgs g, hipStream_t s) {
    constexpr int MB = (MM + 31) / 32, NB = (NN + 31) / 32, NW = MB * (NB / TNW);
    constexpr int PA = (MB * 32 + 127) / 128, PB = (NB * 32 + 127) / 128, BUF = 3 * (PA + PB) * BP * 256;
    constexpr size_t lds = 2 * (size_t)BUF + sizeof(float) * (4 * MB * 32 + 3 * NB * 32);
    static_assert(lds <= 160 * 1024, "LDS");
    if (g.dbias != nullptr || g.P % BP != 0) return PN2_EUNSUPPORTED;
    auto kern = split_tn_kernel<MM, NN, TNW, DYM, PKP, BP>;
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    int64_t wgs = pn2_num_cus();
    int64_t rows = pn2_cdiv(pn2_cdiv(g.P, wgs), BP) * BP;
    if (PKP > 0 && rows % PKP != 0 && PKP % rows != 0) rows = pn2_cdiv(rows, PKP) * PKP;   // chunks never straddle a group: BP | PKP
    g.rows_per_wg = rows;
    wgs = pn2_cdiv(g.P, rows);
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * NW), lds, s, g);
    if (g.ws != nullptr) {                                          // (second phase as launch_wgrad_full's)
        const int rows_pad = MB * 32, ldn = NB * 32;
        const unsigned gx = (unsigned)pn2_cdiv((int64_t)rows_pad * (ldn / 4), 256);
        unsigned gy = (unsigned)(4 * pn2_num_cus() / gx);
        if (gy < 1) gy = 1;
        if (gy > (unsigned)wgs) gy = (unsigned)wgs;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, gy), dim3(256), 0, s, g.ws, (int)wgs, rows_pad, ldn, g.M, g.N, g.dW, g.lddw);
    }
    return pn2_launch_status();
}

template <int MM, int NN, int TNW, int DYM, bool XACT, int PKP, int BP>
int launch_wgrad_full(WgradArgs g, hipStream_t s) {
    constexpr int MB = (MM + 31) / 32, NB = (NN + 31) / 32;
    constexpr int NW = MB * (NB / TNW), LDA = MB * 32 + 4, LDB = NB * 32 + 4;
    constexpr size_t lds = sizeof(float) * (2 * BP * LDA + 2 * BP * LDB + 4 * MB * 32 + 3 * NB * 32 + 4 * 64 * NW);
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = wgrad_full_kernel<MM, NN, TNW, DYM, XACT, PKP, BP, false>;
    if (g.dbias != nullptr) return PN2_EUNSUPPORTED;              // (eval-mode bias gradients: the streamed kernel)
    static Pn2PerDevice raised;
    if (pn2_raise_dynamic_lds(reinterpret_cast<const void *>(kern), raised) != PN2_OK) return PN2_ELAUNCH;
    int64_t wgs = pn2_num_cus();
    int64_t rows = pn2_cdiv(pn2_cdiv(g.P, wgs), BP) * BP;
    if (PKP > 0 && rows % PKP != 0 && PKP % rows != 0) rows = pn2_cdiv(rows, PKP) * PKP;   // chunks never straddle a group: BP | PKP
    g.rows_per_wg = rows;
    wgs = pn2_cdiv(g.P, rows);
    PN2_NOTE_KERNEL(kern);
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(64 * NW), lds, s, g);
    if (g.ws != nullptr) {
        const int rows_pad = MB * 32, ldn = NB * 32;
        const unsigned gx = (unsigned)pn2_cdiv((int64_t)rows_pad * (ldn / 4), 256);
        unsigned gy = (unsigned)(4 * pn2_num_cus() / gx);          // ~4 workgroups per CU in all
        if (gy < 1) gy = 1;
        if (gy > (unsigned)wgs) gy = (unsigned)wgs;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, gy), dim3(256), 0, s, g.ws, (int)wgs, rows_pad, ldn, g.M, g.N, g.dW, g.lddw);
    }
    return pn2_launch_status();
}

}  // namespace

// wide weight gradients (C_out x C_in): 256 x 196 and 256 x 128 on the max-pool's sparse dZ (groups of 64 or 128), 196 x 128 and
// 128 x 128 dense; the input is always a BatchNorm + ReLU of the previous layer's output here
int pn2_wide_wgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y, int ldy,
                   const float *coef, const float *X, int ldx, const float *x_affine, float *dW, int lddw, float *dbias,
                   int64_t P, int M, int N, LazyCoef lc, hipStream_t s, float *workspace) {
    const int on = pn2_opt(PN2_OPT_WIDE) && pn2_opt(PN2_OPT_WIDE_WGRAD);
    const int min_rows = pn2_opt(PN2_OPT_WIDE_WGRAD_MIN_ROWS);
    if (!on || P < min_rows || x_affine == nullptr || ldy != ((M + 3) & ~3) || ldx != ((N + 3) & ~3)) return PN2_EUNSUPPORTED;
    if (!dZ && (Kpool <= 0 || P % Kpool != 0)) return PN2_EUNSUPPORTED;
    if (dZ && ldz != ldy) return PN2_EUNSUPPORTED;                 // (one lane offset serves Y and dZ)
    WgradArgs g{};
    g.Y = Y; g.ldy = ldy; g.dZ = dZ; g.ldz = ldz; g.dZp = dZp; g.arg = arg; g.ldo = ldo; g.coef = coef; g.X = X; g.ldx = ldx;
    g.x_aff = x_affine; g.dW = dW; g.lddw = lddw; g.dbias = dbias; g.P = P; g.M = M; g.N = N; g.lc = lc; g.ws = workspace;
    if (pn2_opt(PN2_OPT_SPLIT) && pn2_opt(PN2_OPT_SPLIT_WGRAD) && dbias == nullptr && P % 16 == 0) {
#define SPLIT_WGRAD(MM, NN, TNW, PKP, BP)                                                                                \
        if (M == MM && N == NN && P % BP == 0 && (PKP == 0 ? dZ != nullptr : (dZ == nullptr && Kpool == PKP)))          \
            return launch_split_tn<MM, NN, TNW, PKP == 0 ? MODE_DYDENSE : MODE_DYPOOLED, PKP, BP>(g, s);
        // (32-row chunks where the LDS holds them -- four column blocks per wave are 24 MFMAs per 16 rows: too short a barrier
        // interval; same box: 196 x 128 dense 190 (fp32) / 216 (16-row chunks) / 157 us, 256 x 128 pooled 110 / 87 / 84)
        SPLIT_WGRAD(256, 196, 7, 128, 16)
        SPLIT_WGRAD(256, 196, 7, 64, 16)
        SPLIT_WGRAD(256, 128, 4, 64, 32)
        SPLIT_WGRAD(256, 128, 4, 128, 32)
        SPLIT_WGRAD(196, 128, 4, 0, 32)
        SPLIT_WGRAD(256, 128, 4, 64, 16)                            // (row counts that are no multiple of 32)
        SPLIT_WGRAD(256, 128, 4, 128, 16)
        SPLIT_WGRAD(196, 128, 4, 0, 16)
#undef SPLIT_WGRAD
    }
#define WIDE_WGRAD(MM, NN, TNW, PKP, BP)                                                                                 \
    if (M == MM && N == NN && (PKP == 0 ? dZ != nullptr : (dZ == nullptr && Kpool == PKP)))                             \
        return launch_wgrad_full<MM, NN, TNW, PKP == 0 ? MODE_DYDENSE : MODE_DYPOOLED, true, PKP, BP>(g, s);
    WIDE_WGRAD(256, 196, 7, 128, 16)
    WIDE_WGRAD(256, 196, 7, 64, 16)
    WIDE_WGRAD(256, 128, 4, 64, 32)
    WIDE_WGRAD(256, 128, 4, 128, 32)
    WIDE_WGRAD(196, 128, 4, 0, 32)
    // (128 x 128 at 131 072 rows: 64.4 us against 61.8 streamed -- the atomic tail of 256 full copies outweighs the saved
    // re-reads on a product this small; instantiation dropped)
#undef WIDE_WGRAD
    return PN2_EUNSUPPORTED;
}

// Bytes of caller scratch with which pn2_conv1x1_wgrad flushes dW in two phases (0: this shape keeps the atomic flush, or the
// two-phase form is switched off: PN2_WGRAD_TWO_PHASE, default by measurement -- see HISTORY.md section 4).
int64_t pn2_wide_wgrad_workspace_bytes(int64_t P, int M, int N, int pooled) {
    const int on = pn2_opt(PN2_OPT_WIDE) && pn2_opt(PN2_OPT_WIDE_WGRAD) && pn2_opt(PN2_OPT_WGRAD_TWO_PHASE);
    const int min_rows = pn2_opt(PN2_OPT_WIDE_WGRAD_MIN_ROWS);
    if (!on || P < min_rows) return 0;
    // measured alone on the chip (us, atomic flush -> two-phase): 256 x 196 360.8 -> 340.4, 196 x 128 192.2 -> 193.8, 256 x 128
    // 105.9 -> 111.3: only the largest product has a tail long enough to pay for the second launch
    const int all = pn2_opt(PN2_OPT_WGRAD_TWO_PHASE_ALL);
    const bool shape = (M == 256 && N == 196 && pooled) || (all && ((M == 256 && N == 128 && pooled) || (M == 196 && N == 128 && !pooled)));
    if (!shape) return 0;
    return (int64_t)pn2_num_cus() * ((M + 31) / 32 * 32) * ((N + 31) / 32 * 32) * (int64_t)sizeof(float);
}
