/* CPU oracle, robust estimation: cv::RNG, 7-point fundamental RANSAC and the
 * FeatureMatcher filter chain.  TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
 * PARITY UNPINNED (no OpenCV here, no reference test vectors).
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int ora_cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------- RNG */

/* [OCV-recall] cv::RNG: multiply-with-carry, coefficient 4164903690 */
uint32_t ora_rng_next(ora_rng *r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}

int ora_rng_uniform(ora_rng *r, int a, int b)
{
    return a == b ? a : (int)(ora_rng_next(r) % (uint32_t)(b - a) + a);
}

/* [OCV-recall] RANSACUpdateNumIters (ptsetreg.cpp) */
static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : ora_cv_round(num / denom);
}

/* ----------------------------------------------------- 7-point F solver */

/* [OCV-recall] cv::solveCubic, 4-coefficient form; returns root count */
static int solve_cubic(const double c[4], double r[3])
{
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0, x1 = 0, x2 = 0;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * M_PI / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * M_PI / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * pow(R, 1. / 3) - a1 / 3; x1 = pow(R, 1. / 3) - a1 / 3; }
            else { x0 = 2 * pow(-R, 1. / 3) - a1 / 3; x1 = -pow(-R, 1. / 3) - a1 / 3; }
            n = 2;
        } else {
            d = sqrt(-d);
            double e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    r[0] = x0; r[1] = x1; r[2] = x2;
    return n;
}

/* Orthonormal basis (f1, f2) of the null space of the 7x9 matrix A.
 * OpenCV takes the last two right singular vectors of A; any orthonormal
 * basis of that plane yields the same pencil of candidate matrices, so the
 * basis is obtained here from a Householder QR of A^T (the last two columns
 * of Q), which needs no iteration and behaves the same on CPU and GPU. */
static void null_space_7x9(const double A[7][9], double f1[9], double f2[9])
{
    double M[9][7]; /* A^T, reduced in place */
    double V[7][9]; /* Householder vectors */
    double beta[7];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 7; j++) M[i][j] = A[j][i];
    for (int k = 0; k < 7; k++) {
        double s = 0;
        for (int i = k; i < 9; i++) s += M[i][k] * M[i][k];
        double nrm = sqrt(s);
        double alpha = M[k][k] > 0 ? -nrm : nrm;
        for (int i = 0; i < 9; i++) V[k][i] = 0;
        for (int i = k; i < 9; i++) V[k][i] = M[i][k];
        V[k][k] -= alpha;
        double vn = 0;
        for (int i = k; i < 9; i++) vn += V[k][i] * V[k][i];
        beta[k] = vn > 0 ? 2.0 / vn : 0.0;
        for (int j = k; j < 7; j++) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * M[i][j];
            dot *= beta[k];
            for (int i = k; i < 9; i++) M[i][j] -= dot * V[k][i];
        }
    }
    /* Q e_7 and Q e_8 with Q = H_0 H_1 ... H_6 */
    for (int c = 0; c < 2; c++) {
        double q[9] = {0};
        q[7 + c] = 1.0;
        for (int k = 6; k >= 0; k--) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * q[i];
            dot *= beta[k];
            for (int i = k; i < 9; i++) q[i] -= dot * V[k][i];
        }
        memcpy(c == 0 ? f1 : f2, q, sizeof(q));
    }
}

/* [OCV-recall] run7Point (fundam.cpp) with Hartley normalisation of both
 * point sets; returns 0..3 matrices scaled so that F[8] == 1. */
int ora_f7(const float *m1, const float *m2, double *Fout)
{
    double A[7][9], f1[9], f2[9], c[4], r[3];
    double m1cx = 0, m1cy = 0, m2cx = 0, m2cy = 0, scale1 = 0, scale2 = 0;
    const double t = 1. / 7;
    for (int i = 0; i < 7; i++) {
        m1cx += m1[2 * i]; m1cy += m1[2 * i + 1];
        m2cx += m2[2 * i]; m2cy += m2[2 * i + 1];
    }
    m1cx *= t; m1cy *= t; m2cx *= t; m2cy *= t;
    for (int i = 0; i < 7; i++) {
        double dx = m1[2 * i] - m1cx, dy = m1[2 * i + 1] - m1cy;
        scale1 += sqrt(dx * dx + dy * dy);
        dx = m2[2 * i] - m2cx; dy = m2[2 * i + 1] - m2cy;
        scale2 += sqrt(dx * dx + dy * dy);
    }
    scale1 *= t; scale2 *= t;
    if (scale1 < FLT_EPSILON || scale2 < FLT_EPSILON) return 0;
    scale1 = sqrt(2.) / scale1;
    scale2 = sqrt(2.) / scale2;

    for (int i = 0; i < 7; i++) {
        double x0 = (m1[2 * i] - m1cx) * scale1, y0 = (m1[2 * i + 1] - m1cy) * scale1;
        double x1 = (m2[2 * i] - m2cx) * scale2, y1 = (m2[2 * i + 1] - m2cy) * scale2;
        A[i][0] = x1 * x0; A[i][1] = x1 * y0; A[i][2] = x1;
        A[i][3] = y1 * x0; A[i][4] = y1 * y0; A[i][5] = y1;
        A[i][6] = x0; A[i][7] = y0; A[i][8] = 1;
    }
    null_space_7x9(A, f1, f2);

    /* det(lambda*f1 + (1-lambda)*f2) = 0 */
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7];
    double t1 = f2[3] * f2[8] - f2[5] * f2[6];
    double t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) +
           f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) -
           f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7];
    t1 = f1[3] * f1[8] - f1[5] * f1[6];
    t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) +
           f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) -
           f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);

    int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n < 0 ? 0 : n;

    for (int k = 0; k < n; k++) {
        double Fn[9], F[9];
        double lambda = r[k], mu = 1.;
        double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; Fn[8] = 1.; }
        else Fn[8] = 0.;
        for (int i = 0; i < 8; i++) Fn[i] = f1[i] * lambda + f2[i] * mu;
        /* de-normalise: F = T2^T Fn T1, T = [s 0 -s*cx; 0 s -s*cy; 0 0 1] */
        double T1[9] = {scale1, 0, -scale1 * m1cx, 0, scale1, -scale1 * m1cy, 0, 0, 1};
        double T2[9] = {scale2, 0, -scale2 * m2cx, 0, scale2, -scale2 * m2cy, 0, 0, 1};
        double tmp[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int l = 0; l < 3; l++) acc += T2[l * 3 + i] * Fn[l * 3 + j];
                tmp[i * 3 + j] = acc;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int l = 0; l < 3; l++) acc += tmp[i * 3 + l] * T1[l * 3 + j];
                F[i * 3 + j] = acc;
            }
        if (fabs(F[8]) > FLT_EPSILON) {
            double inv = 1. / F[8];
            for (int i = 0; i < 9; i++) F[i] *= inv;
        }
        memcpy(Fout + 9 * k, F, sizeof(F));
    }
    return n;
}

/* [OCV-recall] FMEstimatorCallback::computeError: larger of the two squared
 * point-to-epipolar-line distances, computed in double, stored as float. */
void ora_f_error(const float *p1, const float *p2, int n, const double *F,
                 float *err)
{
    for (int i = 0; i < n; i++) {
        double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double a = F[0] * x1 + F[1] * y1 + F[2];
        double b = F[3] * x1 + F[4] * y1 + F[5];
        double c = F[6] * x1 + F[7] * y1 + F[8];
        double s2 = 1. / (a * a + b * b);
        double d2 = x2 * a + y2 * b + c;
        a = F[0] * x2 + F[3] * y2 + F[6];
        b = F[1] * x2 + F[4] * y2 + F[7];
        c = F[2] * x2 + F[5] * y2 + F[8];
        double s1 = 1. / (a * a + b * b);
        double d1 = x1 * a + y1 * b + c;
        double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
        err[i] = (float)(e1 > e2 ? e1 : e2);
    }
}

/* [OCV-recall] haveCollinearPoints: last point against every earlier pair */
static int have_collinear(const float *m, int count)
{
    int i = count - 1;
    for (int j = 0; j < i; j++) {
        double dx1 = m[2 * j] - m[2 * i], dy1 = m[2 * j + 1] - m[2 * i + 1];
        for (int k = 0; k < j; k++) {
            double dx2 = m[2 * k] - m[2 * i], dy2 = m[2 * k + 1] - m[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <=
                FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return 1;
        }
    }
    return 0;
}

/* [OCV-recall] RANSACPointSetRegistrator::getSubset for 2-D/2-D data */
static int get_subset_2d(const float *p1, const float *p2, int count, int m,
                         float *ms1, float *ms2, ora_rng *rng, int max_attempts,
                         int check_collinear)
{
    int idx[16];
    for (int iters = 0; iters < max_attempts; iters++) {
        int i;
        for (i = 0; i < m; i++) {
            int idx_i;
            for (;;) {
                idx_i = ora_rng_uniform(rng, 0, count);
                int dup = 0;
                for (int j = 0; j < i; j++) if (idx[j] == idx_i) { dup = 1; break; }
                if (!dup) break;
            }
            idx[i] = idx_i;
            ms1[2 * i] = p1[2 * idx_i]; ms1[2 * i + 1] = p1[2 * idx_i + 1];
            ms2[2 * i] = p2[2 * idx_i]; ms2[2 * i + 1] = p2[2 * idx_i + 1];
        }
        if (!check_collinear || (!have_collinear(ms1, m) && !have_collinear(ms2, m)))
            return 1;
    }
    return 0;
}

static int cmp_float_asc(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* [OCV-recall] LMeDSPointSetRegistrator::run (ptsetreg.cpp), what
 * findFundamentalMat(FM_RANSAC) runs below 15 points: a fixed number of
 * iterations for 45 % outliers, the model with the smallest median error
 * (element count/2 of the sorted errors) wins, inliers are the points within
 * sigma = 2.5 * 1.4826 * (1 + 5 / (count - 7)) * sqrt(median).  The mask is
 * written even when fewer than 7 inliers remain (the call then reports
 * failure), as OpenCV does. */
static int find_fundamental_lmeds(const float *p1, const float *p2, int n, double conf, int max_iters,
                                  uint8_t *mask, double *F, int *iters_run)
{
    const int model_points = 7;
    ora_rng rng = {(uint64_t)-1};
    int niters = ransac_update_num_iters(conf, 0.45, model_points, max_iters);
    if (niters < 3) niters = 3;
    double min_median = DBL_MAX, best[9] = {0};
    float err[16], sorted[16], ms1[14], ms2[14];
    double models[27];
    int iter;
    for (iter = 0; iter < niters; iter++) {
        if (!get_subset_2d(p1, p2, n, model_points, ms1, ms2, &rng, 10000, 1)) {
            if (iter == 0) return 0;
            break;
        }
        const int nmodels = ora_f7(ms1, ms2, models);
        if (nmodels <= 0) continue;
        for (int k = 0; k < nmodels; k++) {
            ora_f_error(p1, p2, n, models + 9 * k, err);
            memcpy(sorted, err, sizeof(float) * (size_t)n);
            qsort(sorted, (size_t)n, sizeof(float), cmp_float_asc);
            const double median = sorted[n / 2];
            if (median < min_median) {
                min_median = median;
                memcpy(best, models + 9 * k, sizeof(best));
            }
        }
    }
    if (iters_run) *iters_run = iter;
    if (!(min_median < DBL_MAX)) return 0;
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - model_points)) * sqrt(min_median);
    if (sigma < 0.001) sigma = 0.001;
    const float t = (float)(sigma * sigma);
    ora_f_error(p1, p2, n, best, err);
    int good = 0;
    for (int i = 0; i < n; i++) { mask[i] = (uint8_t)(err[i] <= t); good += mask[i]; }
    memcpy(F, best, sizeof(best));
    return good >= model_points;
}

/* [OCV-recall] findFundamentalMat(FM_RANSAC) -> RANSACPointSetRegistrator::run
 * with modelPoints 7, seed (uint64)-1, threshold compared squared in float;
 * with 8..14 points the LMedS estimator above (the reference never passes
 * fewer than 8, src/feature_matcher.cpp:52). */
int ora_find_fundamental_ransac(const float *p1, const float *p2, int n,
                                double thr, double conf, int max_iters,
                                uint8_t *mask, double *F, int *iters_run)
{
    const int model_points = 7;
    if (iters_run) *iters_run = 0;
    memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < 8) return 0;
    if (thr <= 0) thr = 3;
    if (conf < DBL_EPSILON || conf > 1 - DBL_EPSILON) conf = 0.99;
    if (n < 15) return find_fundamental_lmeds(p1, p2, n, conf, 1000, mask, F, iters_run);

    ora_rng rng = {(uint64_t)-1};
    int niters = max_iters > 1 ? max_iters : 1;
    int max_good = 0;
    float *err = (float *)malloc(sizeof(float) * (size_t)n);
    uint8_t *cur = (uint8_t *)malloc((size_t)n);
    const float t = (float)(thr * thr);
    float ms1[14], ms2[14];
    double models[27];
    int iter;
    for (iter = 0; iter < niters; iter++) {
        if (!get_subset_2d(p1, p2, n, model_points, ms1, ms2, &rng, 10000, 1)) {
            if (iter == 0) { free(err); free(cur); return 0; }
            break;
        }
        int nmodels = ora_f7(ms1, ms2, models);
        if (nmodels <= 0) continue;
        for (int k = 0; k < nmodels; k++) {
            ora_f_error(p1, p2, n, models + 9 * k, err);
            int good = 0;
            for (int i = 0; i < n; i++) { cur[i] = (uint8_t)(err[i] <= t); good += cur[i]; }
            if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
                memcpy(mask, cur, (size_t)n);
                memcpy(F, models + 9 * k, sizeof(double) * 9);
                max_good = good;
                niters = ransac_update_num_iters(conf, (double)(n - good) / n, model_points, niters);
            }
        }
    }
    if (iters_run) *iters_run = iter;
    free(err); free(cur);
    return max_good > 0;
}

/* reference src/feature_matcher.cpp:34-103 */
int ora_match_filter(const ora_knn2 *knn, int nq, int nt, const ora_kp *qk,
                     const ora_kp *tk, float ratio, ora_match *out,
                     uint8_t *ransac_mask, int *n_ratio)
{
    return ora_match_filter_x(knn, 0, nq, nt, qk, tk, ratio, out, ransac_mask, n_ratio);
}

/* the same with an optional mutual-nearest cross-check between the ratio
 * test and RANSAC (knn_rev = kNN of the swapped pair, or NULL) */
int ora_match_filter_x(const ora_knn2 *knn, const ora_knn2 *knn_rev, int nq, int nt,
                       const ora_kp *qk, const ora_kp *tk, float ratio, ora_match *out,
                       uint8_t *ransac_mask, int *n_ratio)
{
    ora_match *good = (ora_match *)malloc(sizeof(ora_match) * (size_t)(nq > 0 ? nq : 1));
    int ng = ora_ratio_test(knn, nq, nt, ratio, qk, tk, good);
    if (knn_rev) {
        int k = 0;
        for (int i = 0; i < ng; i++)
            if (knn_rev[good[i].train].idx0 == good[i].query) good[k++] = good[i];
        ng = k;
    }
    if (n_ratio) *n_ratio = ng;
    int nout = 0;
    if (ng >= 8) {
        float *p1 = (float *)malloc(sizeof(float) * 2 * (size_t)ng);
        float *p2 = (float *)malloc(sizeof(float) * 2 * (size_t)ng);
        double F[9];
        for (int i = 0; i < ng; i++) {
            p1[2 * i] = good[i].px; p1[2 * i + 1] = good[i].py;
            p2[2 * i] = good[i].cx; p2[2 * i + 1] = good[i].cy;
        }
        ora_find_fundamental_ransac(p1, p2, ng, 3.0, 0.99, 1000, ransac_mask, F, 0);
        const float max_movement = 100.0f;
        for (int i = 0; i < ng; i++) {
            if (!ransac_mask[i]) continue;
            float dx = good[i].cx - good[i].px, dy = good[i].cy - good[i].py;
            float distance = sqrtf(dx * dx + dy * dy);
            if (distance < max_movement) out[nout++] = good[i];
        }
        free(p1); free(p2);
    }
    free(good);
    return nout;
}
