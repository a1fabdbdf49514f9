/* CPU oracle, pose from 3-D/2-D correspondences.  TEST INFRASTRUCTURE ONLY --
 * see vo_oracle.h.  PARITY UNPINNED (no OpenCV here, no reference vectors).
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* [OCV-recall] cv::Rodrigues, vector -> matrix */
void ora_rodrigues_vec2mat(const double *r, double *R)
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c;
    double itheta = 1. / theta;
    double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

/* [OCV-recall] cv::Rodrigues, matrix -> vector (input assumed orthonormal) */
void ora_rodrigues_mat2vec(const double *R, double *r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; rx = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; ry = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5; rz = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* The PnP solvers are filled in below (EPnP minimal solver, iterative
 * refinement, RANSAC driver). */
int ora_epnp(const double *obj, const double *img_norm, int n, double *R, double *t)
{
    (void)obj; (void)img_norm; (void)n; (void)R; (void)t;
    return 0;
}

int ora_pnp_iterative(const double *obj, const double *img, int n, double fx,
                      double fy, double cx, double cy, double *rvec, double *tvec)
{
    (void)obj; (void)img; (void)n; (void)fx; (void)fy; (void)cx; (void)cy;
    (void)rvec; (void)tvec;
    return 0;
}

int ora_solve_pnp_ransac(const float *obj, const float *img, int n, double fx,
                         double fy, double cx, double cy, int iters,
                         double thr, double conf, double *rvec, double *tvec,
                         int32_t *inliers, int *n_inliers)
{
    (void)obj; (void)img; (void)n; (void)fx; (void)fy; (void)cx; (void)cy;
    (void)iters; (void)thr; (void)conf; (void)rvec; (void)tvec; (void)inliers;
    if (n_inliers) *n_inliers = 0;
    return 0;
}
