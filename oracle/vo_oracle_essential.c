/* CPU oracle, essential matrix: five-point minimal solver, Sampson error and
 * the RANSAC loop of cv::findEssentialMat(p1, p2, K, RANSAC, prob, threshold,
 * maxIters, mask).  TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
 *
 * The reference never estimates an essential matrix (its pose comes from
 * cv::solvePnPRansac, reference src/frame_processor.cpp:176-196); this is the
 * monocular-pose estimator asked for on top of the reference.  PARITY UNPINNED:
 * the RANSAC loop, the Sampson error and the threshold scaling are [OCV-recall]
 * of modules/calib3d/src/five-point.cpp and ptsetreg.cpp; the minimal solver
 * restates Nister's published algorithm (null space of the 5x9 epipolar
 * system, the ten cubic constraints det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0,
 * Gauss-Jordan elimination, a 3x3 polynomial matrix in z whose determinant is
 * the tenth-degree polynomial) with a deterministic real-root finder instead of
 * OpenCV's solvePoly, so solutions agree with OpenCV's as a set, not bitwise.
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- polynomials in (x, y, z) of total degree <= 3 -------------------- */

/* monomial order of the 10 x 20 constraint matrix:
 *  0 x^3  1 y^3  2 x^2y 3 xy^2 4 x^2z 5 x^2  6 y^2z 7 y^2  8 xyz  9 xy
 * 10 xz^2 11 xz  12 x   13 yz^2 14 yz  15 y   16 z^3  17 z^2 18 z   19 1   */
static const signed char MONO[20][3] = {
    {3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
    {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

static int mono_index(int a, int b, int c)
{
    for (int i = 0; i < 20; i++)
        if (MONO[i][0] == a && MONO[i][1] == b && MONO[i][2] == c) return i;
    return -1;
}

/* linear forms are (cx, cy, cz, c1); quadratics use the monomials
 * 0 x^2 1 y^2 2 z^2 3 xy 4 xz 5 yz 6 x 7 y 8 z 9 1 */
static const signed char LIN[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
static const signed char QUAD[10][3] = {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 1, 0}, {1, 0, 1},
                                        {0, 1, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
static int lin_lin_map[4][4], quad_lin_map[10][4], maps_ready;

static void build_maps(void)
{
    if (maps_ready) return;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int q = 0; q < 10; q++)
                if (QUAD[q][0] == LIN[i][0] + LIN[j][0] && QUAD[q][1] == LIN[i][1] + LIN[j][1] &&
                    QUAD[q][2] == LIN[i][2] + LIN[j][2])
                    lin_lin_map[i][j] = q;
    for (int q = 0; q < 10; q++)
        for (int j = 0; j < 4; j++)
            quad_lin_map[q][j] = mono_index(QUAD[q][0] + LIN[j][0], QUAD[q][1] + LIN[j][1], QUAD[q][2] + LIN[j][2]);
    maps_ready = 1;
}

/* q (10 coefficients) += a * b for two linear forms */
static void lin_lin(const double *a, const double *b, double *q)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) q[lin_lin_map[i][j]] += a[i] * b[j];
}

/* c (20 coefficients) += s * q * l for a quadratic and a linear form */
static void quad_lin(const double *q, const double *l, double s, double *c)
{
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) c[quad_lin_map[i][j]] += s * (q[i] * l[j]);
}

/* ---- null space of the 5 x 9 epipolar system ---------------------------- */

/* Orthonormal basis of null(A), A 5x9: the last four columns of Q in a
 * Householder QR of A^T (same construction as the 7-point solver uses). */
static void null_space_5x9(const double A[5][9], double N[4][9])
{
    double M[9][5], V[5][9], beta[5];
    for (int i = 0; i < 9; i++)
        for (int j = 0; j < 5; j++) M[i][j] = A[j][i];
    for (int k = 0; k < 5; k++) {
        double s = 0;
        for (int i = k; i < 9; i++) s += M[i][k] * M[i][k];
        const double nrm = sqrt(s);
        const double alpha = M[k][k] > 0 ? -nrm : nrm;
        for (int i = 0; i < 9; i++) V[k][i] = 0;
        for (int i = k; i < 9; i++) V[k][i] = M[i][k];
        V[k][k] -= alpha;
        double vn = 0;
        for (int i = k; i < 9; i++) vn += V[k][i] * V[k][i];
        beta[k] = vn > 0 ? 2.0 / vn : 0.0;
        for (int j = k; j < 5; j++) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * M[i][j];
            dot *= beta[k];
            for (int i = k; i < 9; i++) M[i][j] -= dot * V[k][i];
        }
    }
    for (int c = 0; c < 4; c++) {
        double q[9] = {0};
        q[5 + c] = 1.0;
        for (int k = 4; k >= 0; k--) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * q[i];
            dot *= beta[k];
            for (int i = k; i < 9; i++) q[i] -= dot * V[k][i];
        }
        memcpy(N[c], q, sizeof(q));
    }
}

/* ---- real roots of a polynomial ------------------------------------------ */

static double horner(const double *p, int deg, double x)
{
    double v = p[deg];
    for (int i = deg - 1; i >= 0; i--) v = v * x + p[i];
    return v;
}

/* root of p in [lo, hi] with p(lo), p(hi) of opposite sign: 200 bisection
 * steps at most, stopping when the interval no longer shrinks */
static double bisect(const double *p, int deg, double lo, double hi, double flo)
{
    for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (mid == lo || mid == hi) break;
        const double fm = horner(p, deg, mid);
        if (fm == 0) return mid;
        if ((fm < 0) == (flo < 0)) { lo = mid; flo = fm; }
        else hi = mid;
    }
    return 0.5 * (lo + hi);
}

/* All real roots of p (coefficients p[0..deg], p[deg] != 0), ascending: the
 * roots of the derivative split [-B, B] (B = Cauchy bound) into monotone
 * pieces, each bracketed piece is bisected.  Roots of even multiplicity (no
 * sign change) are not reported. */
static int real_roots(const double *p, int deg, double *roots)
{
    while (deg > 0 && p[deg] == 0) deg--;
    if (deg <= 0) return 0;
    if (deg == 1) { roots[0] = -p[0] / p[1]; return 1; }
    double d[16], crit[16];
    for (int i = 1; i <= deg; i++) d[i - 1] = p[i] * i;
    const int nc = real_roots(d, deg - 1, crit);
    double bound = 0;
    for (int i = 0; i < deg; i++) {
        const double r = fabs(p[i] / p[deg]);
        if (r > bound) bound = r;
    }
    bound += 1.0;
    int n = 0;
    double lo = -bound, flo = horner(p, deg, lo);
    for (int k = 0; k <= nc; k++) {
        double hi = k < nc ? crit[k] : bound;
        if (hi <= lo) continue;
        if (hi > bound) hi = bound;
        const double fhi = horner(p, deg, hi);
        if (flo == 0) { roots[n++] = lo; }
        else if (fhi != 0 && (flo < 0) != (fhi < 0)) roots[n++] = bisect(p, deg, lo, hi, flo);
        lo = hi;
        flo = fhi;
    }
    if (flo == 0 && n < deg) roots[n++] = lo;
    return n;
}

/* ---- the five-point solver ------------------------------------------------- */

static void conv(const double *a, int da, const double *b, int db, double *out)
{
    for (int i = 0; i <= da + db; i++) out[i] = 0;
    for (int i = 0; i <= da; i++)
        for (int j = 0; j <= db; j++) out[i + j] += a[i] * b[j];
}

/* q1, q2: 5 x 2 normalised image points (x = K^-1 p).  Writes up to 10
 * matrices (row-major, Frobenius norm 1, ascending z-root order) with
 * q2^T E q1 = 0; returns their number. */
int ora_e5(const double *q1, const double *q2, double *Eout)
{
    double A[5][9], N[4][9];
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2;
        A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1; A[i][7] = y1; A[i][8] = 1;
    }
    null_space_5x9(A, N);

    /* E = x N0 + y N1 + z N2 + N3: nine linear forms */
    double lin[9][4];
    for (int e = 0; e < 9; e++)
        for (int k = 0; k < 4; k++) lin[e][k] = N[k][e];

    /* EEt = E E^T (quadratic), its trace, then the ten cubics:
     * rows 0..8 = 2 (E E^T) E - tr(E E^T) E, row 9 = det(E) */
    build_maps();
    double EEt[3][3][10], tr[10], Cc[10][20];
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            for (int m = 0; m < 10; m++) EEt[i][j][m] = 0;
            for (int k = 0; k < 3; k++) lin_lin(lin[3 * i + k], lin[3 * j + k], EEt[i][j]);
            if (j != i)
                for (int m = 0; m < 10; m++) EEt[j][i][m] = EEt[i][j][m];
        }
    for (int m = 0; m < 10; m++) tr[m] = EEt[0][0][m] + EEt[1][1][m] + EEt[2][2][m];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double *c = Cc[3 * i + j];
            for (int m = 0; m < 20; m++) c[m] = 0;
            for (int k = 0; k < 3; k++) quad_lin(EEt[i][k], lin[3 * k + j], 2.0, c);
            quad_lin(tr, lin[3 * i + j], -1.0, c);
        }
    {
        double *c = Cc[9];
        for (int m = 0; m < 20; m++) c[m] = 0;
        static const int perm[6][3] = {{0, 4, 8}, {1, 5, 6}, {2, 3, 7}, {2, 4, 6}, {1, 3, 8}, {0, 5, 7}};
        for (int t = 0; t < 6; t++) {
            double ab[10] = {0};
            lin_lin(lin[perm[t][0]], lin[perm[t][1]], ab);
            quad_lin(ab, lin[perm[t][2]], t < 3 ? 1.0 : -1.0, c);
        }
    }

    /* Gauss-Jordan on the first ten columns, partial pivoting */
    double M[10][20];
    for (int r = 0; r < 10; r++)
        for (int c = 0; c < 20; c++) M[r][c] = Cc[r][c];
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (fabs(M[piv][col]) < 1e-300) return 0;
        if (piv != col)
            for (int c = 0; c < 20; c++) { const double t = M[piv][c]; M[piv][c] = M[col][c]; M[col][c] = t; }
        const double inv = 1.0 / M[col][col];
        for (int c = col; c < 20; c++) M[col][c] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            const double f = M[r][col];
            if (f == 0) continue;
            for (int c = col; c < 20; c++) M[r][c] -= f * M[col][c];
        }
    }

    /* rows 4..9 give x^2z, x^2, y^2z, y^2, xyz, xy in the trailing monomials
     * [xz^2 xz x | yz^2 yz y | z^3 z^2 z 1].  k = row(x^2z) - z row(x^2) etc.:
     * B(z) [x y 1]^T = 0 with entries of degree 3, 3, 4 in z (ascending powers) */
    double B[3][3][5];
    for (int r = 0; r < 3; r++) {
        const double *hi = M[4 + 2 * r], *lo = M[5 + 2 * r];
        /* x part: hi has (xz^2, xz, x) at 10,11,12; minus z * lo */
        B[r][0][0] = hi[12];
        B[r][0][1] = hi[11] - lo[12];
        B[r][0][2] = hi[10] - lo[11];
        B[r][0][3] = -lo[10];
        B[r][0][4] = 0;
        B[r][1][0] = hi[15];
        B[r][1][1] = hi[14] - lo[15];
        B[r][1][2] = hi[13] - lo[14];
        B[r][1][3] = -lo[13];
        B[r][1][4] = 0;
        B[r][2][0] = hi[19];
        B[r][2][1] = hi[18] - lo[19];
        B[r][2][2] = hi[17] - lo[18];
        B[r][2][3] = hi[16] - lo[17];
        B[r][2][4] = -lo[16];
    }
    /* det B(z): degree 10 */
    double p[11] = {0}, t1[8], t2[8], m[12];
    static const int cof[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    for (int c = 0; c < 3; c++) {
        /* minor of row 0, column c, cyclic so that all signs are + */
        const int a = cof[c][0], b = cof[c][1];
        const int da = a == 2 ? 4 : 3, db = b == 2 ? 4 : 3, dc = c == 2 ? 4 : 3;
        conv(B[1][a], da, B[2][b], db, t1);
        conv(B[1][b], db, B[2][a], da, t2);
        for (int i = 0; i <= da + db; i++) t1[i] -= t2[i];
        conv(B[0][c], dc, t1, da + db, m);
        for (int i = 0; i <= 10 && i <= dc + da + db; i++) p[i] += m[i];
    }

    double roots[10];
    const int nr = real_roots(p, 10, roots);
    int n_out = 0;
    for (int k = 0; k < nr; k++) {
        const double z = roots[k];
        double r0[3], r1[3];
        for (int c = 0; c < 3; c++) {
            r0[c] = horner(B[0][c], c == 2 ? 4 : 3, z);
            r1[c] = horner(B[1][c], c == 2 ? 4 : 3, z);
        }
        const double cx = r0[1] * r1[2] - r0[2] * r1[1];
        const double cy = r0[2] * r1[0] - r0[0] * r1[2];
        const double cw = r0[0] * r1[1] - r0[1] * r1[0];
        if (fabs(cw) < 1e-300) continue;
        const double x = cx / cw, y = cy / cw;
        double E[9], nrm = 0;
        for (int e = 0; e < 9; e++) {
            E[e] = x * N[0][e] + y * N[1][e] + z * N[2][e] + N[3][e];
            nrm += E[e] * E[e];
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0) || !isfinite(nrm)) continue;
        for (int e = 0; e < 9; e++) Eout[9 * n_out + e] = E[e] / nrm;
        n_out++;
    }
    return n_out;
}

/* [OCV-recall] EMEstimatorCallback::computeError: Sampson distance
 * (x2^T E x1)^2 / (Ex1_0^2 + Ex1_1^2 + Etx2_0^2 + Etx2_1^2), float result */
void ora_e_error(const double *q1, const double *q2, int n, const double *E, float *err)
{
    for (int i = 0; i < n; i++) {
        const double x1[3] = {q1[2 * i], q1[2 * i + 1], 1.0}, x2[3] = {q2[2 * i], q2[2 * i + 1], 1.0};
        const double Ex1[3] = {E[0] * x1[0] + E[1] * x1[1] + E[2], E[3] * x1[0] + E[4] * x1[1] + E[5],
                               E[6] * x1[0] + E[7] * x1[1] + E[8]};
        const double Etx2[2] = {E[0] * x2[0] + E[3] * x2[1] + E[6], E[1] * x2[0] + E[4] * x2[1] + E[7]};
        const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + Ex1[2];
        err[i] = (float)(x2tEx1 * x2tEx1 /
                         (Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + Etx2[0] * Etx2[0] + Etx2[1] * Etx2[1]));
    }
}

/* Hypotheses from a fixed index table (H x 5 point indices): for each sample
 * the five-point solutions and, per solution, the number of points whose
 * Sampson error is <= thr^2 (thr in normalised units).  E: H x 10 x 9,
 * n_models: H, counts: H x 10. */
void ora_e_hypotheses(const double *q1, const double *q2, int n, const int32_t *idx, int H, double thr, double *E,
                      int32_t *n_models, int32_t *counts)
{
    float *err = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    const float t = (float)(thr * thr);
    for (int h = 0; h < H; h++) {
        double s1[10], s2[10];
        for (int i = 0; i < 5; i++) {
            const int j = idx[5 * h + i];
            s1[2 * i] = q1[2 * j]; s1[2 * i + 1] = q1[2 * j + 1];
            s2[2 * i] = q2[2 * j]; s2[2 * i + 1] = q2[2 * j + 1];
        }
        const int nm = ora_e5(s1, s2, E + (size_t)h * 90);
        n_models[h] = nm;
        for (int m = 0; m < 10; m++) {
            int good = 0;
            if (m < nm) {
                ora_e_error(q1, q2, n, E + (size_t)h * 90 + 9 * m, err);
                for (int i = 0; i < n; i++) good += err[i] <= t;
            }
            counts[10 * h + m] = good;
        }
    }
    free(err);
}

static int e_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* [OCV-recall] cv::findEssentialMat(p1, p2, K, RANSAC, prob, threshold,
 * maxIters, mask): pixel points are normalised with K, the threshold is divided
 * by (fx + fy) / 2, then RANSACPointSetRegistrator::run with modelPoints 5,
 * seed (uint64)-1, no subset check.  p1/p2: n x 2 float32 pixels.  Returns 1
 * and fills E (Frobenius norm 1) and mask, or 0. */
int ora_find_essential_ransac(const float *p1, const float *p2, int n, double fx, double fy, double cx, double cy,
                              double prob, double threshold, int max_iters, uint8_t *mask, double *E,
                              int *iters_run)
{
    const int model_points = 5;
    if (iters_run) *iters_run = 0;
    memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < model_points) return 0;
    double *q1 = (double *)malloc(sizeof(double) * 2 * (size_t)n), *q2 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        q1[2 * i] = ((double)p1[2 * i] - cx) / fx; q1[2 * i + 1] = ((double)p1[2 * i + 1] - cy) / fy;
        q2[2 * i] = ((double)p2[2 * i] - cx) / fx; q2[2 * i + 1] = ((double)p2[2 * i + 1] - cy) / fy;
    }
    const double thr = threshold / ((fx + fy) / 2);
    const float t = (float)(thr * thr);
    ora_rng rng = {(uint64_t)-1};
    int niters = max_iters > 1 ? max_iters : 1, max_good = 0, iter;
    float *err = (float *)malloc(sizeof(float) * (size_t)n);
    uint8_t *cur = (uint8_t *)malloc((size_t)n);
    double models[90];
    for (iter = 0; iter < niters; iter++) {
        int idx[5];
        double s1[10], s2[10];
        if (n == model_points) {
            for (int i = 0; i < 5; i++) idx[i] = i;
        } else {
            for (int i = 0; i < 5; i++) {
                for (;;) {
                    const int v = ora_rng_uniform(&rng, 0, n);
                    int dup = 0;
                    for (int j = 0; j < i; j++) dup |= idx[j] == v;
                    if (!dup) { idx[i] = v; break; }
                }
            }
        }
        for (int i = 0; i < 5; i++) {
            s1[2 * i] = q1[2 * idx[i]]; s1[2 * i + 1] = q1[2 * idx[i] + 1];
            s2[2 * i] = q2[2 * idx[i]]; s2[2 * i + 1] = q2[2 * idx[i] + 1];
        }
        const int nm = ora_e5(s1, s2, models);
        for (int k = 0; k < nm; k++) {
            ora_e_error(q1, q2, n, models + 9 * k, err);
            int good = 0;
            for (int i = 0; i < n; i++) { cur[i] = (uint8_t)(err[i] <= t); good += cur[i]; }
            if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
                memcpy(mask, cur, (size_t)n);
                memcpy(E, models + 9 * k, sizeof(double) * 9);
                max_good = good;
                niters = e_update_num_iters(prob, (double)(n - good) / n, model_points, niters);
            }
        }
    }
    if (iters_run) *iters_run = iter;
    free(q1); free(q2); free(err); free(cur);
    return max_good > 0;
}

/* ---- pose from an essential matrix ------------------------------------------ */

/* Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 4),
 * row-major; on return A is diagonal and column j of V the eigenvector of
 * A[j][j].  The GPU kernel performs the same operations in the same order. */
static void jacobi_sym(double *A, double *V, int n)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                if (i != j) off += A[i * n + j] * A[i * n + j];
                else diag += A[i * n + j] * A[i * n + j];
            }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

static double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* [OCV-recall] cv::decomposeEssentialMat (five-point.cpp): E = U D Vt, U and
 * Vt negated when their determinant is negative, W = [0 1 0; -1 0 0; 0 0 1],
 * R1 = U W Vt, R2 = U W^T Vt, t = third column of U.  The SVD here comes from
 * the Jacobi eigen-decomposition of E^T E (singular values descending, ties by
 * index): u_i = E v_i / |E v_i| for the two leading pairs, u_2 = u_0 x u_1.
 * A different SVD routine can exchange R1 and R2 or negate t; the set
 * {R1, R2} x {t, -t} that ora_recover_pose chooses from is the same. */
void ora_decompose_essential(const double *E, double *R1, double *R2, double *t)
{
    double A[9], V[9], U[9], Vt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += E[k * 3 + i] * E[k * 3 + j];
            A[i * 3 + j] = s;
        }
    jacobi_sym(A, V, 3);
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++)
            if (A[order[j] * 4] > A[order[i] * 4]) { const int o = order[i]; order[i] = order[j]; order[j] = o; }
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Vt[i * 3 + k] = V[k * 3 + order[i]];
    for (int c = 0; c < 2; c++) {
        double u[3], nrm = 0;
        for (int i = 0; i < 3; i++) {
            u[i] = E[i * 3 + 0] * Vt[c * 3 + 0] + E[i * 3 + 1] * Vt[c * 3 + 1] + E[i * 3 + 2] * Vt[c * 3 + 2];
            nrm += u[i] * u[i];
        }
        nrm = sqrt(nrm);
        for (int i = 0; i < 3; i++) U[i * 3 + c] = u[i] / nrm;
    }
    U[2] = U[3] * U[7] - U[6] * U[4]; /* u0 x u1: det(U) = +1 */
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; i++) Vt[i] = -Vt[i];
    /* U W = [-u1 u0 u2], U W^T = [u1 -u0 u2] (columns) */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double a = U[i * 3 + 1] * Vt[0 * 3 + j], b = U[i * 3 + 0] * Vt[1 * 3 + j], c = U[i * 3 + 2] * Vt[2 * 3 + j];
            R1[i * 3 + j] = -a + b + c;
            R2[i * 3 + j] = a - b + c;
        }
    for (int i = 0; i < 3; i++) t[i] = U[i * 3 + 2];
}

/* [OCV-recall] cv::triangulatePoints for one correspondence between
 * P0 = [I | 0] and P = [R | t]: the right singular vector of the smallest
 * singular value of the 4 x 4 system {x P[2] - P[0], y P[2] - P[1]} of both
 * views, here the eigenvector of the smallest eigenvalue of A^T A (first
 * minimum).  Its sign is arbitrary; the cheirality tests do not depend on it. */
static void triangulate_one(const double *R, const double *t, double x1, double y1, double x2, double y2, double *Q)
{
    double A[4][4] = {{-1, 0, x1, 0}, {0, -1, y1, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int k = 0; k < 3; k++) {
        A[2][k] = x2 * R[6 + k] - R[k];
        A[3][k] = y2 * R[6 + k] - R[3 + k];
    }
    A[2][3] = x2 * t[2] - t[0];
    A[3][3] = y2 * t[2] - t[1];
    double M[16], V[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[k][i] * A[k][j];
            M[i * 4 + j] = s;
        }
    jacobi_sym(M, V, 4);
    int m = 0;
    for (int i = 1; i < 4; i++)
        if (M[i * 5] < M[m * 5]) m = i;
    for (int k = 0; k < 4; k++) Q[k] = V[k * 4 + m];
}

/* cheirality test of cv::recoverPose for one triangulated point */
static int cheirality_ok(const double *R, const double *t, const double *Q, double dist)
{
    int ok = Q[2] * Q[3] > 0;
    const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
    ok &= Z < dist;
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    ok &= z2 > 0;
    ok &= z2 < dist;
    return ok;
}

/* [OCV-recall] cv::recoverPose(E, p1, p2, K, R, t, mask) with
 * distanceThresh = 50: the four (R, t) candidates of decomposeEssentialMat,
 * each point triangulated under each, a candidate's votes = points in front of
 * both cameras and nearer than the threshold (AND the incoming mask when
 * use_mask), first candidate with the most votes wins in the order
 * (R1,t) (R2,t) (R1,-t) (R2,-t).  p1/p2: n x 2 float32 pixels.  mask (n bytes)
 * is read when use_mask and always written with the winner's 0/1 votes.
 * Returns the number of votes. */
int ora_recover_pose(const double *E, const float *p1, const float *p2, int n, double fx, double fy, double cx,
                     double cy, double dist, int use_mask, uint8_t *mask, double *R, double *t)
{
    double R1[9], R2[9], tt[3], tn[3];
    ora_decompose_essential(E, R1, R2, tt);
    for (int i = 0; i < 3; i++) tn[i] = -tt[i];
    const double *Rs[4] = {R1, R2, R1, R2};
    const double *ts[4] = {tt, tt, tn, tn};
    int good[4] = {0, 0, 0, 0};
    uint8_t *votes = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        const double x1 = ((double)p1[2 * i] - cx) / fx, y1 = ((double)p1[2 * i + 1] - cy) / fy;
        const double x2 = ((double)p2[2 * i] - cx) / fx, y2 = ((double)p2[2 * i + 1] - cy) / fy;
        uint8_t v = 0;
        for (int k = 0; k < 4; k++) {
            double Q[4];
            triangulate_one(Rs[k], ts[k], x1, y1, x2, y2, Q);
            int ok = cheirality_ok(Rs[k], ts[k], Q, dist);
            if (use_mask && !mask[i]) ok = 0;
            v |= (uint8_t)(ok << k);
            good[k] += ok;
        }
        votes[i] = v;
    }
    int best = 3;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) best = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) best = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) best = 2;
    memcpy(R, Rs[best], sizeof(double) * 9);
    memcpy(t, ts[best], sizeof(double) * 3);
    for (int i = 0; i < n; i++) mask[i] = (votes[i] >> best) & 1;
    free(votes);
    return good[best];
}
