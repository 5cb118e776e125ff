/*
 * pn2_host.c -- the geometry entry points of include/pn2.h as a HOST (CPU) library, same symbols, same signatures.
 *
 * TEST INFRASTRUCTURE ONLY (see pn2_oracle.c).  SURVEY.md section 8(b): "the same symbols exist in a host (CPU) build
 * of the restatement for parity runs" -- a harness that drives the C ABI of the HIP library can be pointed at this
 * library instead (host pointers where the HIP library takes device pointers, the stream argument ignored) and gets
 * the restatement's answers: pn2_fps, pn2_ball_query, pn2_square_distance, pn2_three_nn, pn2_gather_rows, pn2_group,
 * pn2_three_interp.  Each is a thin adapter over the scalar restatement in pn2_oracle.c (compiled into this library),
 * which cites the reference lines it follows.  Never linked into or loaded by pointnet12_amd/.
 */
#include "pn2_oracle.c"
#include "../include/pn2.h"

#define HOST_API __attribute__((visibility("default")))

HOST_API int pn2_version(void) { return PN2_ABI_VERSION; }

HOST_API int64_t pn2_fps_workspace_bytes(int B, int N, int npoint) { (void)B; (void)N; (void)npoint; return 0; }

HOST_API int pn2_fps(const float *xyz, int B, int N, const int64_t *start, int npoint, int64_t *out_idx, void *work,
                     pn2_stream_t stream) {
    (void)work; (void)stream;
    if (!xyz || !start || !out_idx || B <= 0 || N <= 0 || npoint <= 0) return PN2_EINVAL;
    orc_fps(xyz, B, N, start, npoint, out_idx);
    return PN2_OK;
}

HOST_API int pn2_ball_query(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample,
                            int64_t *out_idx, pn2_stream_t stream) {
    (void)stream;
    if (!xyz || !new_xyz || !out_idx || B <= 0 || N <= 0 || S <= 0 || nsample <= 0) return PN2_EINVAL;
    orc_ball_query(xyz, new_xyz, B, N, S, r2, nsample, out_idx);
    return PN2_OK;
}

HOST_API int64_t pn2_ball_query_workspace_bytes(int B, int N, int S) { (void)B; (void)N; (void)S; return 0; }

HOST_API int pn2_ball_query_ws(const float *xyz, const float *new_xyz, int B, int N, int S, float r2, int nsample, int64_t *out_idx,
                               void *work, pn2_stream_t stream) {
    (void)work;
    return pn2_ball_query(xyz, new_xyz, B, N, S, r2, nsample, out_idx, stream);
}

HOST_API int pn2_square_distance(const float *src, const float *dst, int B, int S, int N, float *out, pn2_stream_t stream) {
    (void)stream;
    if (!src || !dst || !out || B <= 0 || S <= 0 || N <= 0) return PN2_EINVAL;
    for (int b = 0; b < B; ++b)
        orc_square_distance(src + (int64_t)b * S * 3, dst + (int64_t)b * N * 3, S, N, out + (int64_t)b * S * N);
    return PN2_OK;
}

HOST_API int pn2_three_nn(const float *xyz1, const float *xyz2, int B, int N, int S, int64_t *idx, float *dist, float *weight,
                          pn2_stream_t stream) {
    (void)stream;
    if (!xyz1 || !xyz2 || !idx || !dist || !weight || B <= 0 || N <= 0 || S < 3) return PN2_EINVAL;
    orc_three_nn(xyz1, xyz2, B, N, S, idx, dist);
    orc_three_weights(dist, (int64_t)B * N, weight);
    return PN2_OK;
}

HOST_API int pn2_gather_rows(const float *points, const int64_t *idx, int B, int N, int C, int M, float *out, int *err,
                             pn2_stream_t stream) {
    (void)stream;
    if (!points || !idx || !out) return PN2_EINVAL;
    const int rc = orc_gather_rows(points, idx, B, N, C, M, out);
    if (err) *err = rc != 0;
    return PN2_OK;
}

HOST_API int pn2_group(const float *xyz, const float *points, const float *new_xyz, const int64_t *idx, int B, int N, int S, int K,
                       int D, int xyz_first, int ld, float *out, int *err, pn2_stream_t stream) {
    (void)stream;
    if (!xyz || !new_xyz || !idx || !out || ld < 3 + D) return PN2_EINVAL;     /* (the un-centred group_all form is not adapted) */
    const int rc = orc_group(xyz, points, new_xyz, idx, B, N, S, K, D, xyz_first, ld, out);
    if (err) *err = rc != 0;
    return PN2_OK;
}

HOST_API int pn2_three_interp(const float *points2, const int64_t *idx, const float *weight, int B, int N, int S, int D, float *out,
                              int ld, int col0, int zero_tail, const float *points1, pn2_stream_t stream) {
    (void)stream;
    if (!points2 || !idx || !weight || !out || ld < col0 + D) return PN2_EINVAL;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)B * N * D);
    if (!tmp) return PN2_EINVAL;
    orc_three_interp(points2, idx, weight, B, N, S, D, tmp);
    for (int64_t r = 0; r < (int64_t)B * N; ++r) {
        float *row = out + r * ld;
        if (points1) memcpy(row, points1 + r * col0, sizeof(float) * (size_t)col0);
        memcpy(row + col0, tmp + r * D, sizeof(float) * (size_t)D);
        if (zero_tail)
            for (int c = col0 + D; c < ld; ++c) row[c] = 0.0f;
    }
    free(tmp);
    return PN2_OK;
}
