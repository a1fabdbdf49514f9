/* CPU oracle, stereo front end: left/right key-point matching inside an
 * epipolar band with cross-check, disparity -> depth, and the sparse depth
 * plane FrameProcessor::backprojectAndFilter can read
 * (reference src/frame_processor.cpp:230-308 samples depth at the rounded key
 * point).  TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
 *
 * The reference has no stereo matcher: its depth comes from the ZED SDK
 * (reference src/zed_interface.cpp:84-85).  The semantics below are this
 * package's own definition, so there is nothing of the reference to be pinned
 * against; the GPU path has to reproduce them bit for bit.
 *
 *   candidates of left key point i: right key points j with
 *       |yL_i - yR_j| <= band   and   min_disp <= xL_i - xR_j <= max_disp
 *   best = smallest Hamming-256 distance, ties to the lowest j; the Lowe test
 *   d0 < ratio * d1 (float32) applies when a second candidate exists;
 *   the same search runs from every right key point over the left ones, and a
 *   pair survives only if each is the other's best (cross-check);
 *   disparity = xL - xR (float32), depth = (float)(fx * baseline / disparity).
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int hamming256(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
}

/* best and second-best candidate of every query; sign = +1 when the queries
 * are left key points (disparity = xq - xt), -1 when they are right ones */
static void band_knn2(const ora_kp *qk, const uint8_t *qd, int nq, const ora_kp *tk, const uint8_t *td, int nt,
                      float band, float min_disp, float max_disp, int sign, ora_knn2 *out)
{
    for (int i = 0; i < nq; i++) {
        ora_knn2 r = {-1, 0x7fff, -1, 0x7fff};
        for (int j = 0; j < nt; j++) {
            const float dy = qk[i].y - tk[j].y;
            const float dx = sign > 0 ? qk[i].x - tk[j].x : tk[j].x - qk[i].x;
            if (!(fabsf(dy) <= band) || !(dx >= min_disp) || !(dx <= max_disp)) continue;
            const int d = hamming256(qd + 32 * (size_t)i, td + 32 * (size_t)j);
            if (d < r.dist0) { r.idx1 = r.idx0; r.dist1 = r.dist0; r.idx0 = j; r.dist0 = d; }
            else if (d < r.dist1) { r.idx1 = j; r.dist1 = d; }
        }
        out[i] = r;
    }
}

/* right_idx (nl): matched right key point or -1; disparity (nl): xL - xR or
 * -1; depth (nl): fx * baseline / disparity or NaN.  Returns the number of
 * matched left key points. */
int ora_stereo_match(const ora_kp *lk, const uint8_t *ld, int nl, const ora_kp *rk, const uint8_t *rd, int nr,
                     float band, float min_disp, float max_disp, float ratio, double fx, double baseline,
                     int32_t *right_idx, float *disparity, float *depth)
{
    ora_knn2 *fwd = (ora_knn2 *)malloc(sizeof(ora_knn2) * (size_t)(nl > 0 ? nl : 1));
    ora_knn2 *bwd = (ora_knn2 *)malloc(sizeof(ora_knn2) * (size_t)(nr > 0 ? nr : 1));
    band_knn2(lk, ld, nl, rk, rd, nr, band, min_disp, max_disp, +1, fwd);
    band_knn2(rk, rd, nr, lk, ld, nl, band, min_disp, max_disp, -1, bwd);
    int n = 0;
    for (int i = 0; i < nl; i++) {
        right_idx[i] = -1;
        disparity[i] = -1.0f;
        depth[i] = NAN;
        const int j = fwd[i].idx0;
        if (j < 0) continue;
        if (fwd[i].idx1 >= 0 && !((float)fwd[i].dist0 < ratio * (float)fwd[i].dist1)) continue;
        if (bwd[j].idx0 != i) continue;
        const float d = lk[i].x - rk[j].x;
        right_idx[i] = j;
        disparity[i] = d;
        depth[i] = (float)(fx * baseline / (double)d);
        n++;
    }
    free(fwd); free(bwd);
    return n;
}

/* Sparse depth plane: NaN everywhere except at (cvRound(x), cvRound(y)) of the
 * matched left key points; where several share a pixel the nearest depth wins. */
void ora_stereo_depth_plane(const ora_kp *lk, const float *depth, int nl, int w, int h, float *plane, int stride)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) plane[(size_t)y * stride + x] = NAN;
    for (int i = 0; i < nl; i++) {
        if (isnan(depth[i])) continue;
        const int u = (int)roundf(lk[i].x), v = (int)roundf(lk[i].y);  /* std::round, as the back-projection samples */
        if (u < 0 || u >= w || v < 0 || v >= h) continue;
        float *p = &plane[(size_t)v * stride + u];
        if (isnan(*p) || depth[i] < *p) *p = depth[i];
    }
}
