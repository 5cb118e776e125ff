/*
 * pn2_oracle.c -- CPU restatement of the PointNet++ geometry primitives.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in
 * pointnet12_amd/ may import, link or call it: the product path is the HIP
 * library and fails loudly when that library is missing.
 *
 * Each function restates, in scalar C with the rounding order spelled out, what
 * the reference computes with ATen ops in model/pointnet_util.py (cited per
 * function as file:line relative to the reference checkout).  The fp32
 * expression forms were pinned against the reference imported on CPU
 * (tools/make_golden.py asserts bit equality every time fixtures are made):
 *
 *   FPS distance      d = ((dx*dx + dy*dy) + dz*dz), no contraction   (:80)
 *   pair distance     dot = fma(az,bz, fma(ay,by, ax*bx))             (:37)
 *                     n(p) = ((x*x + y*y) + z*z), no contraction      (:38-39)
 *                     d   = ((-2*dot) + n(a)) + n(b)
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off is load-bearing; fused ops appear only as __builtin_fmaf.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

static inline float sq_norm3(const float *p) {
    /* torch.sum(p ** 2, -1) over 3 contiguous values: ((x*x + y*y) + z*z). :38-39 */
    float xx = p[0] * p[0];
    float yy = p[1] * p[1];
    float zz = p[2] * p[2];
    return (xx + yy) + zz;
}

static inline float pair_dist(const float *a, float na, const float *b, float nb) {
    /* square_distance, :37-39.  matmul with K=3 accumulates as an fma chain. */
    float dot = a[0] * b[0];
    dot = __builtin_fmaf(a[1], b[1], dot);
    dot = __builtin_fmaf(a[2], b[2], dot);
    float d = -2.0f * dot;
    d = d + na;
    d = d + nb;
    return d;
}

/* square_distance(src, dst) for one cloud.  pointnet_util.py:19-40. */
ORC_API void orc_square_distance(const float *src, const float *dst, int64_t S, int64_t N, float *out) {
    float *nd = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int64_t j = 0; j < N; ++j) nd[j] = sq_norm3(dst + 3 * j);
    for (int64_t i = 0; i < S; ++i) {
        const float *a = src + 3 * i;
        float na = sq_norm3(a);
        for (int64_t j = 0; j < N; ++j) out[i * N + j] = pair_dist(a, na, dst + 3 * j, nd[j]);
    }
    free(nd);
}

/*
 * farthest_point_sample for B clouds.  pointnet_util.py:63-84.
 * start[b] is the randint draw of :75 (the caller owns the RNG).
 * distance starts at 1e10 (:74); update is "dist < distance" (:81-82);
 * argmax takes the lowest index among equal maxima (:83).
 */
ORC_API void orc_fps(const float *xyz, int64_t B, int64_t N, const int64_t *start, int64_t npoint, int64_t *out) {
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        float *dist = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
        const float *p = xyz + b * N * 3;
        int64_t far = start[b];
        for (int64_t j = 0; j < N; ++j) dist[j] = 1e10f;
        for (int64_t i = 0; i < npoint; ++i) {
            out[b * npoint + i] = far;
            float cx = p[3 * far], cy = p[3 * far + 1], cz = p[3 * far + 2];
            float best = -1.0f;
            int64_t besti = 0;
            for (int64_t j = 0; j < N; ++j) {
                float dx = p[3 * j] - cx, dy = p[3 * j + 1] - cy, dz = p[3 * j + 2] - cz;
                float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                float d = (xx + yy) + zz;
                if (d < dist[j]) dist[j] = d;
                if (dist[j] > best) { best = dist[j]; besti = j; }
            }
            far = besti;
        }
        free(dist);
    }
}

/*
 * query_ball_point for B clouds.  pointnet_util.py:87-107.
 * Candidates with sqrdist > r2 are dropped (:102; d == r2 stays in), the
 * first nsample survivors in ascending index order are kept (:103) and the
 * remaining slots repeat the first survivor (:104-106).  An empty ball gives
 * N in every slot, exactly what the reference's tensor holds at :107.
 * r2 is float32(radius ** 2) -- the reference compares fp32 against a Python
 * double; tools/make_golden.py checks the two compares agree for every radius
 * the model zoo uses.
 */
ORC_API void orc_ball_query(const float *xyz, const float *new_xyz, int64_t B, int64_t N, int64_t S,
                            float r2, int64_t nsample, int64_t *out) {
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        float *nd = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
        const float *p = xyz + b * N * 3;
        for (int64_t j = 0; j < N; ++j) nd[j] = sq_norm3(p + 3 * j);
        for (int64_t s = 0; s < S; ++s) {
            const float *q = new_xyz + (b * S + s) * 3;
            float nq = sq_norm3(q);
            int64_t *row = out + (b * S + s) * nsample;
            int64_t cnt = 0;
            for (int64_t j = 0; j < N && cnt < nsample; ++j) {
                float d = pair_dist(q, nq, p + 3 * j, nd[j]);
                if (!(d > r2)) row[cnt++] = j;
            }
            int64_t first = cnt > 0 ? row[0] : N;
            for (int64_t k = cnt; k < nsample; ++k) row[k] = first;
        }
        free(nd);
    }
}

/*
 * The 3-NN search of PointNetFeaturePropagation.forward.  pointnet_util.py:295-297.
 * dists = square_distance(xyz1, xyz2); sort ascending; keep 3.  The reference's
 * sort is not stable for long rows, so equal distances may come back in either
 * order there; here ties go to the lower index and tests compare interpolated
 * values, never raw indices, wherever ties occur.  S >= 3 required (S == 2
 * raises in the reference, S == 1 takes the repeat branch at :292-293).
 */
ORC_API void orc_three_nn(const float *xyz1, const float *xyz2, int64_t B, int64_t N, int64_t S,
                          int64_t *idx, float *dist) {
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        float *nd = (float *)malloc(sizeof(float) * (size_t)(S > 0 ? S : 1));
        const float *p2 = xyz2 + b * S * 3;
        for (int64_t j = 0; j < S; ++j) nd[j] = sq_norm3(p2 + 3 * j);
        for (int64_t i = 0; i < N; ++i) {
            const float *a = xyz1 + (b * N + i) * 3;
            float na = sq_norm3(a);
            float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
            int64_t i0 = 0, i1 = 0, i2 = 0;
            for (int64_t j = 0; j < S; ++j) {
                float d = pair_dist(a, na, p2 + 3 * j, nd[j]);
                if (d < d0) { d2 = d1; i2 = i1; d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = j; }
                else if (d < d2) { d2 = d; i2 = j; }
            }
            int64_t o = (b * N + i) * 3;
            idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
            dist[o] = d0; dist[o + 1] = d1; dist[o + 2] = d2;
        }
        free(nd);
    }
}

/*
 * Inverse-distance weights of :298-300: clamp below 1e-10, reciprocal,
 * normalise by the sum taken as ((w0 + w1) + w2).
 */
ORC_API void orc_three_weights(const float *dist, int64_t rows, float *w) {
    for (int64_t r = 0; r < rows; ++r) {
        float w0 = dist[3 * r], w1 = dist[3 * r + 1], w2 = dist[3 * r + 2];
        if (w0 < 1e-10f) w0 = 1e-10f;
        if (w1 < 1e-10f) w1 = 1e-10f;
        if (w2 < 1e-10f) w2 = 1e-10f;
        w0 = 1.0f / w0; w1 = 1.0f / w1; w2 = 1.0f / w2;
        float s = (w0 + w1) + w2;
        w[3 * r] = w0 / s; w[3 * r + 1] = w1 / s; w[3 * r + 2] = w2 / s;
    }
}

/* Weighted 3-point interpolation of :301: sum over k of points2[idx_k] * w_k, k ascending. */
ORC_API void orc_three_interp(const float *points2, const int64_t *idx, const float *w,
                              int64_t B, int64_t N, int64_t S, int64_t D, float *out) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t i = 0; i < N; ++i) {
            int64_t o = (b * N + i) * 3;
            const float *r0 = points2 + (b * S + idx[o]) * D;
            const float *r1 = points2 + (b * S + idx[o + 1]) * D;
            const float *r2 = points2 + (b * S + idx[o + 2]) * D;
            float *dst = out + (b * N + i) * D;
            for (int64_t c = 0; c < D; ++c) {
                float t0 = r0[c] * w[o], t1 = r1[c] * w[o + 1], t2 = r2[c] * w[o + 2];
                dst[c] = (t0 + t1) + t2;
            }
        }
}

/* index_points with a rank-2 or flattened rank-3 index.  pointnet_util.py:43-60. Returns -1 on a bad index. */
ORC_API int orc_gather_rows(const float *points, const int64_t *idx, int64_t B, int64_t N, int64_t C,
                            int64_t M, float *out) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t m = 0; m < M; ++m) {
            int64_t j = idx[b * M + m];
            if (j < 0 || j >= N) return -1;
            memcpy(out + (b * M + m) * C, points + (b * N + j) * C, sizeof(float) * (size_t)C);
        }
    return 0;
}

/*
 * Grouping of sample_and_group / the MSG loop.  pointnet_util.py:127-133, :243-251.
 * out[b,s,k,:] = [xyz[idx]-new_xyz[s], points[idx]] when xyz_first (SSG, :131)
 *             = [points[idx], xyz[idx]-new_xyz[s]] otherwise   (MSG, :247).
 * points may be NULL (D == 0).  ld >= 3+D is the row pitch of out; pad lanes are zeroed.
 */
ORC_API int orc_group(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx,
                      int64_t B, int64_t N, int64_t S, int64_t K, int64_t D, int xyz_first,
                      int64_t ld, float *out) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t s = 0; s < S; ++s)
            for (int64_t k = 0; k < K; ++k) {
                int64_t j = idx[(b * S + s) * K + k];
                if (j < 0 || j >= N) return -1;
                float *row = out + ((b * S + s) * K + k) * ld;
                const float *p = xyz + (b * N + j) * 3;
                const float *c = new_xyz + (b * S + s) * 3;
                float *gx = xyz_first ? row : row + D;
                float *gp = xyz_first ? row + 3 : row;
                gx[0] = p[0] - c[0]; gx[1] = p[1] - c[1]; gx[2] = p[2] - c[2];
                if (D > 0) memcpy(gp, points + (b * N + j) * D, sizeof(float) * (size_t)D);
                for (int64_t t = 3 + D; t < ld; ++t) row[t] = 0.0f;
            }
    return 0;
}

ORC_API int orc_version(void) { return 1; }
