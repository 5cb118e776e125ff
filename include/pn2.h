/*
 * pn2.h -- C ABI of libpn2_hip.so: the PointNet++ set-abstraction / feature-propagation
 * hot path as hand-written gfx950 (MI355X) HIP kernels.
 *
 * The reference (Jiang-Muyun/PointNet12) has no FFI: its hot path is 314 lines of ATen
 * calls in model/pointnet_util.py.  This library sits BELOW a Python mirror of that file
 * (pointnet12_amd/pointnet_util.py); every entry point names the reference lines whose
 * ATen op sequence it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add to call these from the original file.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; tensors are dense row-major
 *     with the shapes given; float = IEEE fp32; indices = int64 (torch.long at the API);
 *   - the caller owns all memory (no allocation, no retained pointers); the only process-wide state is the option table
 *     below, changed by nothing but pn2_set_option() -- the library never reads the environment;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - return value: PN2_OK (0) or a negative PN2_E* code (pn2_error_string() names it);
 *     launch errors are reported through hipGetLastError() as PN2_ELAUNCH;
 *   - results: index outputs are bit-identical to the reference's CPU results (the fp32
 *     expression forms are pinned in oracle/pn2_oracle.c); float outputs agree to 1e-5.
 */
#ifndef PN2_H
#define PN2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_ABI_VERSION 11

/* Per-channel fp64 reduction buffers ("stats", "red") are PN2_STAT_REPLICAS interleaved copies of
 * double[2*C] (sum, then second moment): workgroups add into copy (workgroup index % replicas) so the
 * same-address atomic queues stay short; pn2_bn_finalize / pn2_bn_bwd_coef sum the copies.  The caller
 * allocates and zeroes PN2_STAT_REPLICAS * 2 * C doubles. */
#define PN2_STAT_REPLICAS 8
#define PN2_DWX_REPLICAS 32   /* copies of the dWx partial block in pn2_group_affine_bwd_seg's scratch */

#define PN2_OK 0
#define PN2_OK_SPLIT 1            /* pn2_conv1x1_bwd_pair only: done, but issued as two launches (not an error) */
#define PN2_EINVAL (-1)     /* bad argument (null pointer, non-positive size, unsupported shape) */
#define PN2_ELAUNCH (-2)    /* hipLaunch / hipMemsetAsync failed */
#define PN2_EUNSUPPORTED (-3)

typedef void *pn2_stream_t;

/* Fused BatchNorm "tails" (optional; pass NULL for none).  A kernel that finishes a per-channel reduction can do
 * the small per-channel step that depends on it before it ends, instead of leaving it to a launch of its own
 * (pn2_bn_finalize / pn2_bn_bwd_coef: ~5 us hops on the dependency chain, 50 per training step): every
 * workgroup takes a ticket once its sums are visible device-wide and the one that draws the last ticket does the
 * work.  *ticket must be 0 on entry (the kernel leaves it 0), and must not be shared by concurrent launches.
 * The arithmetic is the same code path as the stand-alone entry points. */
typedef struct pn2_bn_finalize_tail {       /* training-mode pn2_bn_finalize of the channels this launch produces */
    unsigned *ticket;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var;      /* may be NULL */
    int64_t *num_batches_tracked;           /* may be NULL */
    float *affine;                          /* out: float[4 * round4(C)] */
} pn2_bn_finalize_tail;

typedef struct pn2_bn_coef_tail {           /* pn2_bn_bwd_coef of the layer whose reductions this launch completes */
    unsigned *ticket;
    const float *gamma, *affine;
    int use_batch_stats;
    float *coef;                            /* out: float[4 * round4(C)] */
    float *dgamma, *dbeta;                  /* may be NULL */
    int accumulate;
} pn2_bn_coef_tail;

/* Consumer-side BatchNorm (ABI 8).  The statistics -> affine block step of training-mode BatchNorm (pn2_bn_finalize) and the
 * reductions -> coefficients step of its backward (pn2_bn_bwd_coef) as a PROLOGUE of the first launch that reads the block,
 * instead of a launch of their own (one fused relu(bn(conv)) per layer in the reference: model/pointnet_util.py:195-197,
 * :252-255, :309-312).  The entry points that take a `const pn2_bn_lazy *` / `const pn2_bn_coef_lazy *` fill `affine` /
 * `coef` (which MUST be the very block passed as their in_affine / affine / coef argument) from the finished sums before
 * they read it; the block is valid for every later launch.  NULL: the block was already written.  Running statistics,
 * num_batches_tracked and dgamma / dbeta are updated exactly once per call. */
typedef struct pn2_bn_lazy {
    const double *stats;          /* replicated sums of the producing launch (PN2_STAT_REPLICAS x 2 x C doubles) */
    const float *gamma, *beta;    /* BatchNorm weight / bias, float[C] */
    float eps, momentum;
    float *running_mean, *running_var;      /* may be NULL */
    int64_t *num_batches_tracked;           /* may be NULL */
    float *affine;                /* float[4 * round4(C)], pad entries zero: filled */
    int64_t count;                /* rows the statistics were taken over */
    int C;
} pn2_bn_lazy;

typedef struct pn2_bn_coef_lazy {
    const double *red;            /* replicated reductions (sum dZ, sum dZ * yhat) of the producing launch */
    const float *gamma;
    const float *affine;          /* this layer's affine block */
    float *coef;                  /* float[4 * round4(C)]: filled */
    float *dgamma, *dbeta;        /* may be NULL */
    int accumulate;               /* != 0: dgamma / dbeta are added to */
    int64_t count;
    int C;
} pn2_bn_coef_lazy;

int pn2_version(void);
const char *pn2_error_string(int code);
/* Diagnostics (ABI 11): the kernel template instantiation the CALLING THREAD's most recent GEMM entry point enqueued, spelled as
 * rocprofv3 prints it ("(anonymous namespace)::split_bwd_res_kernel<4, 3, true, true>"), or NULL.  Thread-local; the launchers
 * store a pointer, nothing in the library reads it: bench.py prices every launch of its instrumented pass per KERNEL with it. */
const char *pn2_last_kernel(void);
void pn2_clear_last_kernel(void);

/* Dispatch / tuning options (ABI 10).  Which kernel family takes a layer, tile overrides, A/B switches of measured
 * experiments: one process-wide table of ints, every default the measured winner (the list with defaults and meanings:
 * PN2_OPTION_LIST in pointnet12_amd/csrc/pn2_common.h).  `name` is
 * the option's name with or without its "PN2_" prefix ("PN2_RING", "WIDE_MIN_ROWS", ...).  An option takes effect with the
 * next call; set options before work is enqueued from several threads.  pn2_option_name(i): name of option i, NULL past the
 * end (enumeration).  Unknown name: PN2_EINVAL.
 *
 * NUMERICS CONTRACT.  Every option but the two groups below changes results by fp32 summation order at most.  The exceptions
 * choose the ARITHMETIC a GEMM layer runs in, and are ON by default:
 *   SPLIT (with SPLIT_WGRAD, SPLIT_K256, SPLIT_NARROW, SPLIT_RES, SPLIT_MIN_ROWS_128, SPLIT_RES_MIN_TILES_128 selecting layers):
 *     1 = the long layers (from 65 536 rows) form every fp32 product from EXACT three-way bf16 splits of both operands
 *     (x = hi + mid + lo, 8 + 8 + 8 significand bits) as six v_mfma_f32_32x32x16_bf16 products accumulated in fp32; the three
 *     dropped cross terms are <= 2^-24 |a b| each.  Error against fp64: <= that of the sequential fp32 fma chain of
 *     v_mfma_f32_32x32x2_f32 it replaces (2.9e-6 vs 5.3e-6 at K = 128, profiles/r05_split_gemm_probe.txt), same 1e-5 contract.
 *     PRECONDITION: finite operands with |x| < 2^127 (bf16 rounds a larger |x| to inf and the residual becomes NaN where the
 *     fp32 pipe gives a finite product), and pieces below 2^-126 are flushed (the lo piece of |x| < 2^-110 is lost: relative
 *     error up to 2^-16 on such operands).  Activations, weights and gradients of a BatchNorm network sit 30 binades inside both.
 *     0 = v_mfma_f32_32x32x2_f32 everywhere (an exact fp32 fma chain per output element).
 *   POOL_CF: 2 (default) = a pooled last layer of 128 x 96 or 128 x 64 (1: 128 x 96 only; 0: off) never writes its pre-BN output; its backward is evaluated from the
 *     layer's input (pn2_conv1x1_bwd_cf: the same function, another rounding order; fp64-checked to 3e-6).
 * A caller that needs one arithmetic across library versions sets these explicitly. */
int pn2_set_option(const char *name, int value);
int pn2_get_option(const char *name, int *value);
const char *pn2_option_name(int index);

/* ------------------------------------------------------------------ geometry (index-exact) */

/* farthest_point_sample, model/pointnet_util.py:63-84 (the npoint-iteration Python loop).
 * xyz [B,N,3]; start [B] = the randint draw of :75 (host code owns the RNG);
 * out_idx [B,npoint].  work: caller scratch of pn2_fps_workspace_bytes(B,N,npoint) bytes (may be
 * NULL when that is 0; contents need not be initialised).  Distance form ((dx*dx+dy*dy)+dz*dz)
 * un-fused, argmax ties to the lowest index.  N <= 24576: one workgroup per cloud, cloud and running
 * distances in registers; larger clouds are spread over up to 16 cooperating workgroups each. */
int64_t pn2_fps_workspace_bytes(int B, int N, int npoint);
int pn2_fps(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out_idx,
            void *work, pn2_stream_t stream);

/* query_ball_point, model/pointnet_util.py:87-107 (dense [B,S,N] distance matrix + sort).
 * xyz [B,N,3], new_xyz [B,S,3], r2 = float32(radius**2); out_idx [B,S,nsample]: the first
 * nsample indices with !(d > r2) in ascending order, padded with the first; N everywhere
 * for an empty ball (the reference's tensor at :107). */
int pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample,
                   int64_t *out_idx, pn2_stream_t stream);
/* The same query with caller scratch (`work`: pn2_ball_query_workspace_bytes(B, N, S) bytes, 4-byte aligned; 0 bytes / NULL:
 * identical to pn2_ball_query): on large clouds the centres are first put in spatial (Morton-cell) order so that the sixteen
 * centres a workgroup scans for have similar neighbour densities -- same out_idx, bit for bit. */
int64_t pn2_ball_query_workspace_bytes(int B, int N, int S);
int pn2_ball_query_ws(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample, int64_t *out_idx,
                      void *work, pn2_stream_t stream);

/* square_distance, model/pointnet_util.py:19-40. src [B,S,3], dst [B,N,3] -> out [B,S,N]. */
int pn2_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, pn2_stream_t stream);

/* 3-NN search + inverse-distance weights, model/pointnet_util.py:295-300 (dense matrix +
 * full sort + clamp 1e-10 + reciprocal + normalise).  xyz1 [B,N,3], xyz2 [B,S,3], S >= 3.
 * idx [B,N,3], dist [B,N,3] (raw, unclamped, ascending), weight [B,N,3]. Ties -> lower index. */
int pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx, float *dist,
                 float *weight, pn2_stream_t stream);

/* ------------------------------------------------------------------ gathers / scatters */

/* index_points, model/pointnet_util.py:43-60.  points [B,N,C], idx [B,M] -> out [B,M,C].
 * err (device int, may be NULL): set to 1 if any index is outside [0,N) (the reference raises
 * IndexError); offending rows are written as zeros. */
int pn2_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M, float *out, int *err,
                    pn2_stream_t stream);
/* backward of the above: grad_points [B,N,C] += scatter of grad_out [B,M,C] (caller zeroes). */
int pn2_gather_rows_bwd(const float *grad_out, const int64_t *idx, int B, int N, int C, int M, float *grad_points,
                        pn2_stream_t stream);

/* Grouping of sample_and_group (:127-133) and of the MSG loop (:243-251): gather K
 * neighbours, subtract the centroid from xyz, concatenate with the D features.
 * xyz [B,N,3], points [B,N,D] or NULL (D = 0), new_xyz [B,S,3], idx [B,S,K].
 * out [B*S*K, ld] position-major rows, ld >= 3+D and a multiple of 4; row = [xyz-c, feat] if xyz_first (SSG, :131)
 * else [feat, xyz-c] (MSG, :247); columns 3+D..ld-1 are zero.  new_xyz == NULL means
 * "do not centre" (sample_and_group_all, :140-157, with idx = arange). */
int pn2_group(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S,
              int K, int D, int xyz_first, int ld, float *out, int *err, pn2_stream_t stream);
/* backward: grad_points [B,N,D] += feature columns of grad_rows [B*S*K, ld] (caller zeroes). */
int pn2_group_bwd(const float *grad_rows, const int64_t *idx, int B, int N, int S, int K, int D, int xyz_first,
                  int ld, float *grad_points, pn2_stream_t stream);

/* pn2_group + the FIRST conv of a shared MLP in one launch, for narrow first layers (3 + D <= 12 input channels, C_out 32
 * or 64, B*S*K a multiple of 64; otherwise PN2_EUNSUPPORTED and nothing is launched): X [B*S*K, ldx] receives the grouped
 * rows exactly as pn2_group writes them (the backward's weight gradient reads them), Y [B*S*K, ldy] = X W^T + bias with
 * W [C_out, 3 + D] as stored (pitch ldw), stats (may be NULL) the per-channel sums like pn2_conv1x1_fwd.
 * model/pointnet_util.py:127-131 + :197, :243-247 + :254. */
int pn2_group_conv_fwd(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S,
                       int K, int D, int xyz_first, const float *W, int ldw, const float *bias, float *X, int ldx, float *Y, int ldy,
                       int C_out, double *stats, pn2_stream_t stream);

/* Factorised first MLP layer of a set-abstraction level (replaces gather + cat + the first 1x1 conv,
 * model/pointnet_util.py:127-131,:197 / :243-247,:254, for that layer only):
 *   Y[p, c] = Zf[b, idx[p], c] + sum_a Wx[c, a] * (xyz[b, idx[p], a] - new_xyz[b, s, a]),  p = (b, s, k)
 * with Zf [B*N, ldz] = W_f f + bias precomputed per SOURCE point (one small pn2_conv1x1_fwd) and
 * Wx [C, 3] (row pitch ldwx >= 3) the xyz columns of the layer's weight -- it may point into the full
 * [C, 3+D] weight.  stats as in pn2_conv1x1_fwd (double[2*C], may be NULL). */
int pn2_group_affine_fwd(const float *Zf, int ldz, const float *xyz, const float *new_xyz, const int64_t *idx,
                         const float *Wx, int ldwx, int B, int N, int S, int K, int C, float *Y, int ldy,
                         double *stats, const pn2_bn_finalize_tail *fin, pn2_stream_t stream);
/* backward: dY = c0*dZ + q1*(y-mean) + q0 (coef from pn2_bn_bwd_coef) is scattered to the source points,
 * G[b*N + idx[p], :] += dY[p, :] (G [B*N, ldg], caller zeroes), and dWx[c, a] += dY[p, c] * (xyz - centre)[a]
 * (dWx [C, 3] with row pitch ldwx >= 3 -- it may point at the xyz columns of the full weight gradient --
 * caller zeroes or accumulates). */
int pn2_group_affine_bwd(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *xyz,
                         const float *new_xyz, const int64_t *idx, int B, int N, int S, int K, int C, float *G, int ldg,
                         float *dWx, int ldwx, pn2_stream_t stream);

/* Inverse-distance interpolation, model/pointnet_util.py:301: out[b,n,col0+c] =
 * ((p2[i0,c]*w0 + p2[i1,c]*w1) + p2[i2,c]*w2).  points2 [B,S,D]; out rows of pitch ld
 * (so the result lands directly inside the concatenated FP input, :305).  zero_tail != 0: the columns
 * col0+D .. ld-1 of every row are cleared as well (the pad lanes of a float4-pitched row; nobody pre-clears it).
 * points1 != NULL ([B,N,col0] contiguous): the same launch copies it into columns 0..col0-1, i.e. the whole
 * cat([points1, interpolated], -1) of :305 is one kernel. */
int pn2_three_interp(const float *points2, const int64_t *idx, const float *weight, int B, int N, int S, int D,
                     float *out, int ld, int col0, int zero_tail, const float *points1, pn2_stream_t stream);
/* backward: grad_points2 [B,S,D] += w_k * grad_out[b,n,col0+c] (caller zeroes).  S == 1 (the repeat branch of :292-293: every
 * index is 0, idx is not read): a deterministic column sum, STORED into grad_points2. */
int pn2_three_interp_bwd(const float *grad_out, int ld, int col0, const int64_t *idx, const float *weight, int B,
                         int N, int S, int D, float *grad_points2, pn2_stream_t stream);

/* Strided 2-D copy: dst[r, dcol0 + c] = src[r, scol0 + c], r < rows, c < cols (the cat of :305,
 * and its backward split). */
int pn2_copy_cols(const float *src, int lds, int scol0, float *dst, int ldd, int dcol0, int64_t rows, int cols,
                  pn2_stream_t stream);

/* ------------------------------------------------------------------ shared MLP (1x1 conv + BN + ReLU [+ max])
 *
 * Position-major activations: one grouped position per row, P = B*S*K rows (B*N for FP).
 * Replaces nn.Conv2d/Conv1d(k=1) + nn.BatchNorm2d/1d + F.relu (+ torch.max over K),
 * model/pointnet_util.py:194-199, :251-256, :309-312, and their autograd.
 *
 * Per-channel "affine" blocks are float[4*C] arrays written by pn2_bn_finalize:
 *   [0..C) mean   [C..2C) scale = gamma*invstd   [2C..3C) beta   [3C..4C) invstd
 * and relu((y-mean)*scale+beta) is applied on the fly wherever a pre-BN tensor is consumed.
 */

/* Y[P,N] = act(X)[P,K] * W[N,K]^T + bias.  X has row pitch ldx (>= round4(K), multiple of 4, pad
 * columns zero).  W is the Conv weight [N = C_out, K = C_in] as nn.Conv1d/2d stores it, row pitch
 * ldw >= K floats: no padding or alignment is required (16-byte aligned rows with K % 4 == 0 are read as
 * float4, anything else -- a C_in of 9 or 137, or the feature columns sliced out of a [C_out, 3+D]
 * weight -- with guarded scalar loads).  A 16-byte aligned W with ldw % 4 == 0 and ldw >= round4(K) is read as float4
 * too, the last quad of a row included: its pad entries [K, round4(K)) MUST then be zero (the host side hands over such a
 * padded copy of the 137-column weight of fp1's first layer: 38.5 -> 30.5 us at 65 536 rows).  Y pitch ldy (multiple of 4, >= round4(N); pad lanes are
 * written as zeros).  in_affine: NULL (X is used as is) or the affine block (4*ldx floats) of
 * the layer that produced X.  stats: NULL or a replicated double[2*N] block (see PN2_STAT_REPLICAS; caller zeroes) receiving
 * sum(y) and sum(y*y) per output channel over the P rows (training-mode BN statistics).  P < 2^31 rows for the three conv1x1 entry points (PN2_EINVAL otherwise). */
int pn2_conv1x1_fwd(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias,
                    float *Y, int ldy, int64_t P, int K, int N, double *stats, const pn2_bn_finalize_tail *fin,
                    const pn2_bn_lazy *in_lazy, pn2_stream_t stream);

/* BatchNorm statistics -> affine block.  training != 0: mean/var (biased) from stats/P,
 * running_mean/var (may be NULL) updated with `momentum` and the unbiased variance,
 * *num_batches_tracked (may be NULL) incremented.  training == 0: running statistics are used. */
int pn2_bn_finalize(const double *stats, int64_t P, int C, const float *gamma, const float *beta, float eps,
                    float momentum, int training, float *running_mean, float *running_var,
                    int64_t *num_batches_tracked, float *affine, pn2_stream_t stream);

/* pn2_conv1x1_fwd for the LAST layer of a pooled shared MLP in training mode (model/pointnet_util.py:194-199 / :251-256: the
 * conv whose BN + ReLU output is max-reduced over the Kpool rows of a group), weight-resident kernel only: besides Y and the
 * statistics it records, per group and channel, the extreme pre-BN value -- the largest, or the smallest where gamma (this
 * layer's BatchNorm weight, float[C_out]) is negative -- and the first row attaining it into pool_ws (16-byte aligned
 * float[2 * (P / Kpool) * C_out]: {value, row as int32} pairs).  BN + ReLU is monotone per channel with the sign of gamma,
 * so pn2_bn_pool_select turns these into max_k relu(bn(y_k)) exactly once the affine block exists, and the pass over Y of
 * pn2_bn_relu_max is not needed.  Returns PN2_EUNSUPPORTED (nothing launched) when the shape is outside the resident kernels
 * (see pn2_res_supported; also needs P % 32 == 0, Kpool == 16 or Kpool % 32 == 0): call pn2_conv1x1_fwd + pn2_bn_relu_max
 * then. */
/* Y == NULL (ABI 11): the pre-BN output is not written at all -- statistics and extrema only.  Taken by the bf16-pipe forms of the
 * forward alone (PN2_EUNSUPPORTED otherwise: call again with a Y); the layer's backward then runs on the layer's INPUT:
 * pn2_pool_bwd_reduce_rec + pn2_conv1x1_bwd_cf below, where pn2_conv1x1_bwd_cf_supported() says so. */
int pn2_conv1x1_fwd_pool(const float *X, int ldx, const float *in_affine, const float *W, int ldw, const float *bias, float *Y,
                         int ldy, int64_t P, int K, int N, double *stats, int Kpool, const float *gamma, float *pool_ws,
                         const pn2_bn_lazy *in_lazy, pn2_stream_t stream);
/* out[g,c] = relu(bn(v)) of the recorded extreme value, arg[g,c] = its row (same outputs as pn2_bn_relu_max up to which of
 * several rows with EQUAL post-BN value is named; a channel whose scale is exactly 0 -- BatchNorm weight 0: every row gives
 * relu(beta) -- names row 0, as torch.max of an all-equal group does).  C % 32 == 0, ldo == C. */
int pn2_bn_pool_select(const float *pool_ws, const float *affine, int64_t G, int C, float *out, int ldo, int32_t *arg,
                       const pn2_bn_lazy *lazy, pn2_stream_t stream);

/* out[g,c] = max_k relu(bn(Y[g*K+k, c])), arg[g,c] = first k attaining it (K = 1: plain
 * BN+ReLU, arg may be NULL).  Y pitch ldy, out / arg pitch ldo: both multiples of 4 and >= round4(C)
 * (rows are moved as float4; the pad columns of out / arg are written too, from the zero pad of `affine`). */
int pn2_bn_relu_max(const float *Y, int ldy, const float *affine, int64_t G, int K, int C, float *out, int ldo,
                    int32_t *arg, const pn2_bn_lazy *lazy, pn2_stream_t stream);

/* Backward, last layer after max-pool: dZp[g,c] = out[g,c] > 0 ? dOut[g,c] : 0 (pitch ldo, pad lanes zero),
 * red[0..C) = sum dZ, red[C..2C) = sum dZ*yhat with dZ[g*K+k,c] = (k == arg[g,c]) ? dZp[g,c] : 0.
 * red is double[2*C], caller zeroes.
 * PRECONDITION on `out` (round 4): it must be the pooled output THIS affine block produced, bit for bit --
 * out[g,c] = max(fma(Y[(g*K + arg[g,c]), c] - mean, scale, beta), 0), what pn2_bn_relu_max / pn2_bn_pool_select write.
 * Where |gamma| >= (1 + |beta|) / 4 the kernels take yhat of a positive output from it, yhat = (out - beta) / gamma (error
 * eps * |out| / |gamma|: no gather of Y, one 64-byte sector per element); elsewhere (small or zero gamma) from Y as before.  An
 * `out` from another path (an eval-mode fold, a post-processed slice) gives wrong d gamma / q1.  Y must be non-NULL either way.
 * Both branches and both signs of gamma are held to an fp64 evaluation by tests/test_mlp_gpu.py
 * (test_shared_mlp_negative_and_zero_gamma). */
int pn2_pool_bwd_reduce(const float *dOut, int ldo, const float *out, const int32_t *arg, const float *Y, int ldy,
                        const float *affine, int64_t G, int K, int C, float *dZp, double *red,
                        const pn2_bn_coef_tail *tail, pn2_stream_t stream);
/* The same with the incoming gradient at a pitch of its own (ld_dout >= C, any alignment): a column slice of a wider gradient
 * matrix -- what autograd hands a branch of a concatenated output -- is read in place instead of being copied out first. */
int pn2_pool_bwd_reduce_ld(const float *dOut, int ld_dout, const float *out, int ldo, const int32_t *arg, const float *Y, int ldy,
                           const float *affine, int64_t G, int K, int C, float *dZp, double *red, const pn2_bn_coef_tail *tail,
                           pn2_stream_t stream);
/* pn2_pool_bwd_reduce_ld for a pooled last layer whose pre-BN output was never written (pn2_conv1x1_fwd_pool with Y == NULL): the
 * value at the recorded row comes from pool_ws (the {value, row} records of that forward; `out` / `arg` from pn2_bn_pool_select of
 * the same records) -- exact for every gamma, no gather.  A channel whose folded scale is exactly 0 routes to row 0 of its group,
 * whose value the record does not hold: recomputed from the layer's input, y = bias[c] + W[c, :] . relu(bn(prev_Y[g K, :]))
 * (W [C, C_in] pitch ldw, prev_Y pitch ld_prev with its affine block of pitch round4(C_in)).  Same outputs as
 * pn2_pool_bwd_reduce_ld. */
int pn2_pool_bwd_reduce_rec(const float *dOut, int ld_dout, const float *out, int ldo, const int32_t *arg, const float *pool_ws,
                            const float *affine, int64_t G, int K, int C, float *dZp, double *red, const float *W, int ldw,
                            const float *bias, const float *prev_Y, int ld_prev, const float *prev_affine, int C_in,
                            pn2_stream_t stream);
/* Backward, dense (FP) last layer: dZ = dOut * (out > 0) written to dZ [P, ldz] (its pad columns
 * C .. round4(C)-1 are written as zeros); same reductions, same precondition on `out` (= max(fma(Y - mean, scale, beta), 0)). */
int pn2_relu_bwd_reduce(const float *dOut, int ldo, const float *out, const float *Y, int ldy, const float *affine,
                        int64_t P, int C, float *dZ, int ldz, double *red, const pn2_bn_coef_tail *tail,
                        pn2_stream_t stream);

/* Per-channel BN-backward coefficients from the reductions: coef float[4*C] =
 * [c0 = gamma*invstd, q1 = -c0*invstd*red1/P, q0 = -c0*red0/P, mean]; dgamma = red1, dbeta = red0.
 * dY = c0*dZ + q1*(y-mean) + q0.  use_batch_stats == 0 (eval-mode BN): q1 = q0 = 0.
 * accumulate != 0: dgamma/dbeta are added to (they alias existing .grad storage) instead of overwritten. */
int pn2_bn_bwd_coef(const double *red, int64_t P, int C, const float *gamma, const float *affine,
                    int use_batch_stats, float *coef, float *dgamma, float *dbeta, int accumulate, pn2_stream_t stream);

/* dgrad: dXact[P,N] = dY[P,K] * W[K,N] with dY formed on the fly from (dZ or the pooled
 * pair dOut/arg, Y, coef); K = C_l, N = C_{l-1}.  W is the SAME Conv weight [C_l, C_{l-1}] the forward
 * read (row pitch ldw >= N, no padding / alignment requirement; a 16-byte aligned W with ldw % 4 == 0 and ldw >= round4(N)
 * is read in whole quads and its pad entries [N, round4(N)) MUST be zero): the kernel reads it "down the columns",
 * no transposed copy exists.  dXout pitch ldxo (multiple of 4, >= round4(N); pad lanes written as zeros).
 *   dZ != NULL: dense dZ [P, ldz];  dZ == NULL: pooled form (dZp [G,ldo] from pn2_pool_bwd_reduce, arg, Kpool).
 * Epilogue, prev_Y != NULL: dZprev = dXact * (bn_relu(prev_Y) > 0) -> dXout, and
 *   prev_red (double[2*N], caller zeroes) += sum dZprev, sum dZprev*yhat_prev;
 * prev_Y == NULL (first layer): dXout = dXact.
 * prev_tail (optional, needs prev_Y): the coefficients / dgamma / dbeta of layer l-1 from the finished prev_red. */
int pn2_conv1x1_dgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg,
                      int Kpool, const float *Y, int ldy, const float *coef, const float *W, int ldw,
                      const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                      double *prev_red, int64_t P, int K, int N, const pn2_bn_coef_tail *prev_tail,
                      const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* wgrad: dW[M,N] (pitch lddw, caller zeroes) += sum_p dY[p,m] * Xact[p,n]; M = C_l, N = C_{l-1}.
 * dY formed as in dgrad; Xact = bn_relu(prev_Y) when prev_affine != NULL, else X as is.
 * dbias (may be NULL, caller zeroes) += sum_p dY[p,m]. */
int pn2_conv1x1_wgrad(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg,
                      int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                      const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                      const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* pn2_conv1x1_wgrad with caller scratch (ABI 8): where pn2_conv1x1_wgrad_workspace_bytes(P, M, N, pooled) is > 0 and `workspace`
 * (16-byte aligned, that many bytes, contents irrelevant) is given, the full-tile kernel stores every workgroup's partial dW as
 * plain stores and a second small launch adds the slabs into dW (one atomic per element instead of one per workgroup and
 * element).  workspace == NULL or a query result of 0: exactly pn2_conv1x1_wgrad. */
int64_t pn2_conv1x1_wgrad_workspace_bytes(int64_t P, int M, int N, int pooled);
int pn2_conv1x1_wgrad_ws(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg,
                         int Kpool, const float *Y, int ldy, const float *coef, const float *X, int ldx,
                         const float *x_affine, float *dW, int lddw, float *dbias, int64_t P, int M, int N,
                         const pn2_bn_coef_lazy *coef_lazy, float *workspace, pn2_stream_t stream);

/* Weight gradient of a FIRST layer (C_in = N <= 15, C_out = M a multiple of 16, <= 128) whose data gradient nobody needs, from dZ
 * and the input rows alone (ABI 9).  The BatchNorm-backward terms of dY = c0*dZ + q1*(y - mean) + q0 (model/pointnet_util.py:195-197,
 * backward) are linear in sums the forward already fixed: with s = sum_p x_p, S = sum_p x_p x_p^T and y_p = W x_p + b,
 *     dW[c][j] += c0[c] * sum_p dZ[p,c] x[p,j] + q1[c] * ((W S)[c][j] + (b[c] - mean[c]) * s[j]) + q0[c] * s[j]
 * -- Y is never read (half the bytes of pn2_conv1x1_wgrad on these layers); s and S are accumulated by the same pass (fp64 across
 * workgroups) and the closed-form part is added once, by the workgroup that finishes last.  `coef` as for pn2_conv1x1_wgrad
 * (rows c0, q1, q0, mean of pitch round4(M); realised from `coef_lazy` when given); W [M, N] (pitch ldw) and bias [M] are the
 * layer's parameters; `scratch`: pn2_conv1x1_wgrad_cf_scratch_bytes() bytes, 16-byte aligned, ZEROED by the caller (left dirty).
 * dZ == NULL (ABI 11): sum_p dZ[p,c] x[p,j] is already in `scratch` -- pn2_conv1x1_bwd_first of the NEXT layer added it there --
 * and this call takes the input's moments and finishes.
 * PN2_EUNSUPPORTED for other shapes: use pn2_conv1x1_wgrad. */
int64_t pn2_conv1x1_wgrad_cf_scratch_bytes(void);
int pn2_conv1x1_wgrad_cf(const float *dZ, int ldz, const float *coef, const float *X, int ldx, const float *W, int ldw,
                         const float *bias, void *scratch, float *dW, int lddw, int64_t P, int M, int N,
                         const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* pn2_conv1x1_dgrad followed by pn2_conv1x1_wgrad of ONE layer (same dZ / pooled pair, Y, coef; X = the layer's input, i.e.
 * prev_Y wherever there is a previous layer, with x_affine = prev_affine) as one call: on the few-row and mid-size layers
 * (sa3 / sa4 / FP stacks, P up to 64 k rows) both kernel bodies share ONE launch -- the first workgroups of the grid compute
 * dX, the rest dW -- so neither leaves half of the chip idle and the chain is one launch shorter; elsewhere the two
 * launches are issued one after the other and the call returns PN2_OK_SPLIT (1) instead of PN2_OK.  Results are those of the
 * two separate calls. */
int pn2_conv1x1_bwd_pair(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *Y,
                         int ldy, const float *coef, const float *W, int ldw, const float *prev_Y, int ld_prev,
                         const float *prev_affine, float *dXout, int ldxo, double *prev_red, const float *X, int ldx,
                         const float *x_affine, float *dW, int lddw, int64_t P, int C_out, int C_in,
                         const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* pn2_conv1x1_bwd of the LAST layer of a pooled shared MLP (model/pointnet_util.py:197-199, :254-256: conv + BN + ReLU + max over
 * Kpool rows) WITHOUT that layer's pre-BN output (ABI 11).  dZ is sparse there (one row per group and channel: dZp / arg as
 * pn2_pool_bwd_reduce_rec wrote them) and the dense part of dY = c0 dZ + q1 (y - mean) + q0 is affine in y = W x + b, hence in the
 * layer's input x = relu(bn(prev_Y)), which the pass reads anyway:
 *     dXout = ([D | X] [diag(c0) W ; W^T diag(q1) W] + (q1 (b - mean) + q0)^T W) masked by the previous ReLU, reductions into prev_red;
 *     dW   += c0 o (D^T X) + q1 o (W (X^T X) + (b - mean) (1^T X)) + q0 (1^T X)
 * -- the same function of the same inputs as pn2_conv1x1_bwd (fp32 rounding differs: tests/test_mlp_gpu.py holds both to fp64),
 * 4 P (2 C_in) bytes instead of 4 P (C_out + 2 C_in), and the forward writes no Y (pn2_conv1x1_fwd_pool with Y == NULL).
 * Partial products leave as one slab per workgroup and are summed in a fixed order: dW is run-to-run identical.
 * bias: the conv bias [C_out].  scratch: pn2_conv1x1_bwd_cf_scratch_bytes(C_out, C_in) bytes, 256-byte aligned, contents need
 * not be initialised.  Training-mode BatchNorm, prev_affine != NULL.  pn2_conv1x1_bwd_cf_supported(): 1 where the call runs
 * (128 x 96 and 128 x 64 with Kpool a power of two >= 32, from pn2_res_supported()'s row count on, library options SPLIT /
 * SPLIT_RES / POOL_CF on); PN2_EUNSUPPORTED otherwise. */
int pn2_conv1x1_bwd_cf_supported(int64_t P, int C_out, int C_in, int Kpool);
int64_t pn2_conv1x1_bwd_cf_scratch_bytes(int C_out, int C_in);
int pn2_conv1x1_bwd_cf(const float *dZp, int ldo, const int32_t *arg, int Kpool, const float *coef, const float *W, int ldw,
                       const float *bias, const float *prev_Y, int ld_prev, const float *prev_affine, float *dXout, int ldxo,
                       double *prev_red, float *dW, int lddw, int64_t P, int C_out, int C_in,
                       const pn2_bn_coef_lazy *coef_lazy, float *scratch, pn2_stream_t stream);

/* pn2_conv1x1_bwd of a layer whose INPUT is the output of a FIRST layer with a narrow input X0 (N0 <= 12 columns: the grouped
 * 3 + D rows of sa1; model/pointnet_util.py:127-131 feeding :195-197 / :252-255) when nobody needs a gradient with respect to X0
 * (ABI 11).  The masked dX of this call IS that first layer's dZ; its backward (pn2_conv1x1_wgrad_cf: BatchNorm terms in closed
 * form) needs only sum_p dZ[p, c] X0[p, j] from it, which this call forms from its dX tiles and adds into THAT call's scratch
 * (cf_scratch: pn2_conv1x1_wgrad_cf_scratch_bytes() bytes, zeroed by the caller, then handed to pn2_conv1x1_wgrad_cf with
 * dZ == NULL) -- dX itself is never written.  dense dZ, prev_affine / prev_red required (training-mode BatchNorm), X0 16-byte
 * aligned with pitch ld0 >= 12, ld0 % 4 == 0 (pad columns zero).  prev_red receives the first layer's two reductions as in
 * pn2_conv1x1_bwd.  _supported(): 1 for 96 x 64 and 64 x 64 with P % 64 == 0 from pn2_res_supported()'s row count on (options
 * SPLIT, SPLIT_RES, FUSE_FIRST on); PN2_EUNSUPPORTED otherwise (call pn2_conv1x1_bwd). */
int pn2_conv1x1_bwd_first_supported(int64_t P, int C_out, int C_in, int N0);
int pn2_conv1x1_bwd_first(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *W, int ldw,
                          const float *prev_Y, int ld_prev, const float *prev_affine, double *prev_red, float *dW, int lddw,
                          const float *X0, int ld0, int N0, void *cf_scratch, int64_t P, int C_out, int C_in,
                          const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* Fused backward of one layer for the narrow, long layers (csrc/mlp_res.hip): dgrad AND wgrad in ONE pass over dZ / Y /
 * prev_Y -- autograd of model/pointnet_util.py:197,254,312 for conv + BatchNorm + ReLU.  dY is formed once per row
 * (as in pn2_conv1x1_dgrad), dXout = (dY W) masked by the previous layer's ReLU with its two BatchNorm-backward
 * reductions added to prev_red, dW (pitch lddw, caller zeroes) += dY^T relu(bn(prev_Y)).  prev_affine == NULL: the
 * layer input is prev_Y as stored, dXout = dY W unmasked, prev_red must be NULL.  The weight matrix stays in LDS for
 * the whole launch.  Only for shapes pn2_res_supported() accepts (C_out, C_in multiples of 32, <= 128); training-mode
 * BatchNorm only (no dbias).  pn2_conv1x1_fwd picks the matching weight-resident forward kernel by itself. */
int pn2_res_supported(int64_t P, int C_out, int C_in);
/* 1 if pn2_conv1x1_bwd runs (P, C_out, C_in) with this dZ form (Kpool = 0: dense) and input (masked: BatchNorm + ReLU of the
 * previous layer) in its fused kernel; 0: it would hand the layer to pn2_conv1x1_dgrad + pn2_conv1x1_wgrad. */
int pn2_bwd_res_supported(int64_t P, int C_out, int C_in, int Kpool, int masked);
int pn2_conv1x1_bwd(const float *dZ, int ldz, const float *dZp, int ldo, const int32_t *arg, int Kpool,
                    const float *Y, int ldy, const float *coef, const float *W, int ldw, const float *prev_Y,
                    int ld_prev, const float *prev_affine, float *dXout, int ldxo, double *prev_red, float *dW,
                    int lddw, int64_t P, int C_out, int C_in, const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* Eval-mode fused module (csrc/eval.hip): rows -> L x (linear + ReLU) -> max over the neighbours in ONE launch, BatchNorm
 * folded into the weights by the caller (W' = diag(gamma / sqrt(var + eps)) W, b' likewise; rows of pitch ldw >=
 * round8(K), zero padded, 16-byte aligned).  model/pointnet_util.py:127-133 / :243-251 (gather, centre, concat) +
 * :194-199 / :251-256 / :309-312 (conv + BN + ReLU, max) under .eval() -- the reference's viewer loop pcdvis.py:118-136.
 * Input: X != NULL: plain rows [P, ldx] with P passed in `B` (FP modules, heads);  X == NULL: grouped rows formed on the
 * fly from idx [B,S,Knb] (xyz [B,N,3] minus new_xyz [B,S,3], cat with points [B,N,D]; xyz_first as pn2_group;
 * idx == NULL with S == 1, Knb == N, new_xyz == NULL: group_all, un-centred).  pool: 0 (out [P, ldo], ReLU applied) or
 * Knb (out [P / Knb, ldo]); Knb must be 16 or a multiple of 32.  L <= 4; activations of a 32-row tile stay in LDS.
 * Deviation from the reference: ReLU is fmaxf(x, 0) and the pooling an integer atomicMax on the bits of the non-negative
 * result, both of which DROP a NaN activation (torch.relu / torch.max propagate it).  A diverged model therefore shows
 * finite pooled features here; the training-mode path (pn2_bn_relu_max) and the un-pooled outputs keep NaNs visible. */
typedef struct {
    const float *W;
    const float *bias;
    int K, N, ldw;
} pn2_eval_layer;
int pn2_fused_eval(const float *X, int ldx, const float *xyz, const float *points, const float *new_xyz,
                   const int64_t *idx, int B, int N, int S, int Knb, int D, int xyz_first,
                   const pn2_eval_layer *layers, int L, int pool, float *out, int ldo, pn2_stream_t stream);

/* ---- scatter-adds of the backward pass as segmented reductions (csrc/scatter.hip) ----------------------------
 * pn2_invert_index: idx [B, M] int64 with values in [0, T) -> members int32 [B, M], owners int32 [B, M]: the
 *   positions m of a cloud sorted by the value they point at (a counting sort per cloud; order inside one value is
 *   unspecified), and that value.  Out-of-range entries are dropped (their slots, at the end of the cloud's array,
 *   hold -1).  scratch: int32 [B, 2T+1]. */
int pn2_invert_index(const int64_t *idx, int B, int M, int T, int32_t *members, int32_t *owners, int32_t *scratch,
                     pn2_stream_t stream);
/* pn2_three_interp_bwd over the target-sorted 3-NN index (idx viewed as [B, 3N], T = S): runs of equal target are
 * summed in registers; a run that lies inside one chunk of the member list is STORED, the (at most two) runs that
 * straddle a chunk's ends are added atomically.  grad_points2 [B,S,D]: the caller MUST zero it (targets without
 * members are never written, and the straddling runs accumulate). */
int pn2_three_interp_bwd_seg(const float *grad_out, int ld, int col0, const int32_t *members, const int32_t *owners,
                             const float *weight, int B, int N, int S, int D, float *grad_points2, pn2_stream_t stream);
/* pn2_group_affine_bwd over the source-sorted ball-query index (idx viewed as [B, S*K], T = N).  G [B*N, ldg]:
 * the caller MUST zero it (same store-versus-atomic rule as pn2_three_interp_bwd_seg); dWx accumulated as in pn2_group_affine_bwd.  C <= 256.  dwx_scratch: float[PN2_DWX_REPLICAS * 3 *
 * round4(C)] zeroed by the caller, or NULL.  With it the per-workgroup dWx partials are added to one of
 * PN2_DWX_REPLICAS copies and a second small launch folds the copies into dWx (all resident workgroups finish
 * together; their 3*C same-word atomics on dWx itself cost up to 4x the rest of the launch). */
int pn2_group_affine_bwd_seg(const float *dZ, int ldz, const float *Y, int ldy, const float *coef, const float *xyz,
                             const float *new_xyz, const int32_t *members, const int32_t *owners, int B, int N, int S,
                             int K, int C, float *G, int ldg, float *dWx, int ldwx, float *dwx_scratch,
                             const pn2_bn_coef_lazy *coef_lazy, pn2_stream_t stream);

/* ---- the loss either side of the path ---------------------------------------
 * Replaces F.nll_loss(pred, target) of semseg.py:143 (weight == NULL) and the class-weighted form of
 * pcdseg.py:179, reduction "mean":
 *     loss = - sum_r w[t_r] * logp[r, t_r] / sum_r w[t_r]     over rows with t_r != ignore_index.
 * logp [R, ld >= C] log-probabilities, target int64[R].  A target outside [0, C) that is not ignore_index
 * turns the loss NaN (ATen raises a device assert).  workspace: pn2_nll_loss_workspace_bytes(R) bytes,
 * zeroed ONCE by the caller (the kernel leaves it reusable); do not share it between concurrent launches.
 * Outputs: *loss and *denom (= sum of the weights, kept for the backward).  The partial sums are fp64 and
 * combined in a fixed order: the result does not depend on scheduling. */
int64_t pn2_nll_loss_workspace_bytes(int64_t R);
int pn2_nll_loss_fwd(const float *logp, int ld, const int64_t *target, const float *weight, int64_t R, int C,
                     int64_t ignore_index, void *workspace, float *loss, float *denom, pn2_stream_t stream);
/* dlogp[r, c] = -(*grad_loss) * w[t_r] / (*denom) at c == t_r (and t_r != ignore_index), 0 elsewhere;
 * every element of dlogp [R, ld] is written. */
int pn2_nll_loss_bwd(const int64_t *target, const float *weight, int64_t R, int C, int64_t ignore_index,
                     const float *grad_loss, const float *denom, float *dlogp, int ld, pn2_stream_t stream);
/* F.log_softmax(x, dim=-1) of the segmentation heads (model/pointnet2.py:175; :46, :103, :138) on rows whose C <= 64 logits
 * are the leading columns of a padded row (pitch ldx, a multiple of 4: the output of pn2_conv1x1_fwd as it stands -- no
 * slice copy): out[r, c] = x[r, c] - max_c x - log sum_c exp(x - max) for c < C, pitch ldo >= C. */
int pn2_log_softmax_fwd(const float *x, int ldx, int64_t R, int C, float *out, int ldo, pn2_stream_t stream);
/* Its backward, grad_x[r, c] = grad_out[r, c] - exp(out[r, c]) * sum_c grad_out[r, c], written at pitch ldgx <= 64 with
 * the pad columns c in [C, ldgx) set to zero: the padded gradient pn2_conv1x1_wgrad / _dgrad read as dZ. */
int pn2_log_softmax_bwd(const float *grad_out, int ldg, const float *out, int ldo, int64_t R, int C, float *grad_x, int ldgx,
                        pn2_stream_t stream);

/* ---- the optimiser step and the loader's per-cloud preparation --------------
 * pn2_adam_step replaces torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-08, weight_decay) of
 * semseg.py:106-111 / pcdseg.py:133-138 (amsgrad off, L2 decay folded into the gradient) over ONE flat fp32 buffer
 * that all parameters alias; grad / exp_avg / exp_avg_sq are flat buffers of the same layout.  One launch:
 *     g += weight_decay * p;  m = lerp(m, g, 1 - beta1);  v = beta2 * v + (1 - beta2) * g * g;
 *     p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * `step` is t (>= 1) of this call.  step_dev != NULL: device int64[2] {steps taken so far, 0}; t = step_dev[0] + 1
 * is read on the device and step_dev[0] advanced by the launch itself (hipGraph replay needs no new arguments);
 * `step` is then ignored.  lr_dev != NULL: the learning rate is read from device memory instead of `lr`
 * (pcdseg.py:159-163 rewrites param_group['lr'] every epoch).  zero_grad != 0 also clears grad (the next
 * optimizer.zero_grad(), semseg.py:137) in the same pass. */
int pn2_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, const float *lr_dev, int64_t *step_dev,
                  int zero_grad, pn2_stream_t stream);
/* pn2_prepare_clouds replaces SemKITTI_Loader.__getitem__ (data_utils/SemKITTI_Loader.py:93-113) for a batch:
 * pcd_normalize (:23-30: x/70, y/70, z/3, (i-0.5)*2, clip to [-1,1]), pcd_jitter (:17-21: += noise) and
 * `pcd[choice]`, `label[choice]` (:110-113).
 *   raw        [rows, 4] fp32 x,y,z,intensity (the .bin row format, kitti_utils.py:200) of any number of scans kept
 *              resident in HBM; cloud b of the batch is rows [row_begin[b], row_begin[b] + row_count[b])
 *              (int64[B] each, device memory).
 *   raw_label  int32[rows] class per raw point, or NULL.
 *   noise      [*, 4] fp32 clipped jitter rows, one per RAW point (the reference jitters before it resamples, so
 *              duplicates share their noise); cloud b's rows start at noise_begin[b] (int64[B], device; NULL: the
 *              same rows as raw).  noise == NULL: no jitter (evaluation).
 *   choice     int64[B, N] row numbers inside each cloud (np.random.choice(M, N, replace=True)).
 * Outputs points [B, N, 4] fp32 and labels int64[B, N] (or NULL).  A choice outside [0, row_count[b]) sets
 * *bad_index (device int, caller zeroes, may be NULL) and reads row 0 (numpy raises IndexError). */
int pn2_prepare_clouds(const float *raw, const int64_t *row_begin, const int64_t *row_count, const int32_t *raw_label,
                       const float *noise, const int64_t *noise_begin, const int64_t *choice, int B, int N,
                       float *points, int64_t *labels, int *bad_index, pn2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_H */
