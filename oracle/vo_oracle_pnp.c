/* CPU oracle, pose from 3-D/2-D correspondences: EPnP minimal solver,
 * iterative (DLT + Levenberg-Marquardt) refinement and the RANSAC driver of
 * cv::solvePnPRansac.  TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
 * PARITY UNPINNED (no OpenCV here, no reference vectors); OpenCV's SVD-based
 * linear algebra is replaced by Jacobi eigen/SVD and QR least squares, which
 * agree with it up to rounding on well-conditioned input.
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int ora_cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------ Rodrigues */

/* [OCV-recall] cv::Rodrigues, vector -> matrix; J (optional) is the 3x9
 * derivative dR(row-major)/dr laid out as OpenCV's jacobian. */
static void rodrigues_v2m(const double *r, double *R, double *J)
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (J) {
            memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c;
    double itheta = 1. / theta;
    double n[3] = {r[0] * itheta, r[1] * itheta, r[2] * itheta};
    double rrt[9] = {n[0] * n[0], n[0] * n[1], n[0] * n[2], n[0] * n[1], n[1] * n[1],
                     n[1] * n[2], n[0] * n[2], n[1] * n[2], n[2] * n[2]};
    double r_x[9] = {0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {n[0] + n[0], n[1], n[2], n[1], 0, 0, n[2], 0, 0,
                           0, n[0], 0, n[0], n[1] + n[1], n[2], 0, n[2], 0,
                           0, 0, n[0], 0, 0, n[1], n[0], n[1], n[2] + n[2]};
        static const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                         0, 0, 1, 0, 0, 0, -1, 0, 0,
                                         0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = n[i];
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] +
                               a4 * d_r_x[i * 9 + k];
        }
    }
}

void ora_rodrigues_vec2mat(const double *r, double *R) { rodrigues_v2m(r, R, 0); }

/* ------------------------------------------------- small dense algebra */

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 12).
 * On return w holds eigenvalues in descending order and row i of Vt the
 * matching unit eigenvector (the layout cvSVD(..., U_T) gives for a symmetric
 * positive semi-definite input). */
static void jacobi_eig_desc(const double *Ain, int n, double *w, double *Vt)
{
    double A[144], V[144];
    memcpy(A, Ain, sizeof(double) * (size_t)(n * n));
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
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                /* tan and cos of the rotation angle as in svd12_onesided: d = a_qq - a_pp, h = 2 a_pq,
                 * r = sqrt(d^2 + h^2), tan = sign(d) h / (|d| + r), cos^2 = (|d| + r) / (2 r), both from the one
                 * reciprocal 1 / (2 r (|d| + r)) */
                double d = A[q * n + q] - A[p * n + p], h = 2 * apq;
                double x = d * d + h * h;
                if (!(x > 0)) continue;  /* (both underflowed) */
                double rr = sqrt(x);
                double den = fabs(d) + rr, r2 = 2 * rr;
                double wr = 1.0 / (r2 * den);
                double sh = d < 0 ? -h : h;
                double t = (sh * r2) * wr;
                double c = sqrt((den * den) * wr), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    int order[12];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++)
            if (A[order[j] * n + order[j]] > A[order[i] * n + order[i]]) {
                int t = order[i]; order[i] = order[j]; order[j] = t;
            }
    for (int i = 0; i < n; i++) {
        w[i] = A[order[i] * n + order[i]];
        for (int k = 0; k < n; k++) Vt[i * n + k] = V[k * n + order[i]];
    }
}

/* Right singular vectors of a rows x 12 matrix G by one-sided (Hestenes)
 * Jacobi -- the family of method cv::SVD uses -- with the column pairs of
 * every sweep taken in round-robin (tournament) order so that the six pairs
 * of a round are disjoint.  w receives the squared singular values in
 * descending order (ties by column index), row k of Vt the matching vector;
 * for a symmetric positive semi-definite G these are its eigen-pairs.
 * G is column-major here: G[12*j + i] is element (i, j). */
static int rr_partner(int x, int r)
{
    if (x == 11) return r;
    if (x == r) return 11;
    int y = (2 * r - x) % 11;
    return y < 0 ? y + 11 : y;
}

static void svd12_onesided(const double *Gcm, int rows, double *w, double *Vt)
{
    double G[12][12], V[12][12];
    double frob = 0;
    for (int j = 0; j < 12; j++)
        for (int i = 0; i < 12; i++) {
            G[j][i] = i < rows ? Gcm[12 * j + i] : 0.0;
            V[j][i] = i == j;
        }
    for (int j = 0; j < 12; j++)
        for (int i = 0; i < rows; i++) frob += G[j][i] * G[j][i];
    const double eps = 1e-15;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
        for (int r = 0; r < 11; r++)
            for (int x = 0; x < 12; x++) {
                int y = rr_partner(x, r);
                if (x > y) continue;
                double *gp = G[x], *gq = G[y];
                double alpha = 0, beta = 0, gamma = 0;
                /* the three sums and the rotation below are chains of fused multiply-adds (one rounding per term) */
                for (int i = 0; i < rows; i++) alpha = fma(gp[i], gp[i], alpha);
                for (int i = 0; i < rows; i++) beta = fma(gq[i], gq[i], beta);
                for (int i = 0; i < rows; i++) gamma = fma(gp[i], gq[i], gamma);
                /* |gamma| <= eps sqrt(alpha beta), squared */
                if (gamma * gamma <= (eps * eps) * (alpha * beta) || fabs(gamma) <= 1e-26 * frob) continue;
                rotated = 1;
                /* tan and cos of the angle: d = beta - alpha, h = 2 gamma, r = sqrt(d^2 + h^2),
                 * tan = sign(d) h / (|d| + r), cos^2 = (|d| + r) / (2 r), both as multiples of the one reciprocal
                 * 1 / (2 r (|d| + r)); at d = 0 either sign of tan annihilates gamma and + h is taken */
                double d = beta - alpha, h = 2 * gamma;
                double rr = sqrt(d * d + h * h);
                double den = fabs(d) + rr, r2 = 2 * rr;
                double w = 1.0 / (r2 * den);
                double sh = d < 0 ? -h : h;
                double t = (sh * r2) * w;
                double c = sqrt((den * den) * w), sn = c * t;
                for (int i = 0; i < rows; i++) {
                    double a = gp[i], b = gq[i];
                    gp[i] = fma(-sn, b, c * a);
                    gq[i] = fma(sn, a, c * b);
                }
                double *vp = V[x], *vq = V[y];
                for (int i = 0; i < 12; i++) {
                    double a = vp[i], b = vq[i];
                    vp[i] = fma(-sn, b, c * a);
                    vq[i] = fma(sn, a, c * b);
                }
            }
        if (!rotated) break;
    }
    double nrm[12];
    int order[12];
    for (int j = 0; j < 12; j++) {
        double a = 0;
        for (int i = 0; i < rows; i++) a += G[j][i] * G[j][i];
        nrm[j] = a;
        order[j] = j;
    }
    for (int i = 1; i < 12; i++) { /* stable insertion sort, descending */
        int o = order[i], k = i;
        while (k > 0 && nrm[order[k - 1]] < nrm[o]) { order[k] = order[k - 1]; k--; }
        order[k] = o;
    }
    for (int k = 0; k < 12; k++) {
        w[k] = nrm[order[k]];
        for (int i = 0; i < 12; i++) Vt[k * 12 + i] = V[order[k]][i];
    }
}

/* Unit eigenvector of the smallest eigenvalue of a symmetric positive semi-definite n x n matrix (row-major,
 * n <= 12) by inverse iteration on its L D L^T factorisation: twelve solves from the all-ones start.  The DLT
 * system of the pose refinement (n = 12) and the normalised DLT of the homography (n = 9) have one eigenvalue far
 * below the others (the solution, at the noise level; exactly zero for four points), so the iteration converges
 * in a few steps; what cv::SVD would return as the last row of V^T, up to sign.  Divisions by a pivot or by the
 * norm are multiplications by its reciprocal (one division per pivot and per step). */
static void smallest_eigvec_n(const double *A, int n, double *x)
{
    /* L D L^T without pivoting; a pivot that rounding has driven to zero or below is replaced by a tiny positive
     * number: for inverse iteration that only steepens the amplification of the wanted direction */
    double Lf[144], D[12], invD[12];
    double dmax = 0;
    for (int i = 0; i < n; i++) dmax = fmax(dmax, fabs(A[i * n + i]));
    const double tiny = dmax * 1e-30 + 1e-300;
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= Lf[j * n + k] * Lf[j * n + k] * D[k];
        if (!(d > tiny)) d = tiny;
        D[j] = d;
        invD[j] = 1.0 / d;
        for (int i = j + 1; i < n; i++) {
            double v = A[i * n + j];
            for (int k = 0; k < j; k++) v -= Lf[i * n + k] * Lf[j * n + k] * D[k];
            Lf[i * n + j] = v * invD[j];
        }
    }
    for (int i = 0; i < n; i++) x[i] = 1.0 / sqrt((double)n);
    for (int it = 0; it < 12; it++) {
        double y[12];
        for (int i = 0; i < n; i++) {          /* L z = x */
            double v = x[i];
            for (int k = 0; k < i; k++) v -= Lf[i * n + k] * y[k];
            y[i] = v;
        }
        for (int i = 0; i < n; i++) y[i] *= invD[i];
        for (int i = n - 1; i >= 0; i--) {     /* L^T y = z; terms taken from the last row up */
            double v = y[i];
            for (int k = n - 1; k > i; k--) v -= Lf[k * n + i] * y[k];
            y[i] = v;
        }
        double nrm = 0;
        for (int i = 0; i < n; i++) nrm += y[i] * y[i];
        nrm = sqrt(nrm);
        if (!(nrm > 0) || !isfinite(nrm)) break;
        const double inrm = 1.0 / nrm;
        for (int i = 0; i < n; i++) x[i] = y[i] * inrm;
    }
}
static void smallest_eigvec12(const double *A, double *x) { smallest_eigvec_n(A, 12, x); }

/* nearest rotation to a 3x3 matrix A (U V^T of its SVD), via the
 * eigen-decomposition of A^T A; a vanishing third singular value is
 * completed with a cross product. */
static void polar_rotation3(const double *A, double *R)
{
    double AtA[9], w[3], Vt[9], U[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = s;
        }
    jacobi_eig_desc(AtA, 3, w, Vt);
    for (int c = 0; c < 3; c++) {
        double u[3], nrm = 0;
        for (int i = 0; i < 3; i++) {
            u[i] = A[i * 3 + 0] * Vt[c * 3 + 0] + A[i * 3 + 1] * Vt[c * 3 + 1] + A[i * 3 + 2] * Vt[c * 3 + 2];
            nrm += u[i] * u[i];
        }
        nrm = sqrt(nrm);
        if (c == 2 && nrm <= 1e-12 * sqrt(w[0] > 0 ? w[0] : 1)) {
            u[0] = U[3] * U[7] - U[6] * U[4];   /* u0 x u1, columns of U */
            u[1] = U[6] * U[1] - U[0] * U[7];
            u[2] = U[0] * U[4] - U[3] * U[1];
            nrm = 1;
        }
        for (int i = 0; i < 3; i++) U[i * 3 + c] = u[i] / nrm;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = U[i * 3 + 0] * Vt[0 * 3 + j] + U[i * 3 + 1] * Vt[1 * 3 + j] + U[i * 3 + 2] * Vt[2 * 3 + j];
}

/* least squares min |A x - b| for an m x n system (m >= n, n <= 6) by
 * Householder QR */
static void lstsq_qr(const double *Ain, const double *bin, int m, int n, double *x)
{
    double A[6 * 6 + 12], b[8];
    memcpy(A, Ain, sizeof(double) * (size_t)(m * n));
    memcpy(b, bin, sizeof(double) * (size_t)m);
    for (int k = 0; k < n; k++) {
        double s = 0;
        for (int i = k; i < m; i++) s += A[i * n + k] * A[i * n + k];
        double nrm = sqrt(s);
        if (nrm == 0) continue;
        double alpha = A[k * n + k] > 0 ? -nrm : nrm;
        double v[8];
        for (int i = k; i < m; i++) v[i] = A[i * n + k];
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < m; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        double beta = 2.0 / vn;
        for (int j = k; j < n; j++) {
            double dot = 0;
            for (int i = k; i < m; i++) dot += v[i] * A[i * n + j];
            dot *= beta;
            for (int i = k; i < m; i++) A[i * n + j] -= dot * v[i];
        }
        double dot = 0;
        for (int i = k; i < m; i++) dot += v[i] * b[i];
        dot *= beta;
        for (int i = k; i < m; i++) b[i] -= dot * v[i];
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = s / A[k * n + k];
    }
}

/* solve the n x n system A x = b (n <= 6) by Gaussian elimination with
 * partial pivoting; returns 0 when singular */
static int solve_gauss(const double *Ain, const double *bin, int n, double *x)
{
    double A[36], b[6];
    memcpy(A, Ain, sizeof(double) * (size_t)(n * n));
    memcpy(b, bin, sizeof(double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[piv * n + k])) piv = i;
        if (A[piv * n + k] == 0) return 0;
        if (piv != k) {
            for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = s / A[k * n + k];
    }
    return 1;
}

static int invert3(const double *m, double *inv)
{
    double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                 m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (det == 0) return 0;
    double id = 1.0 / det;
    inv[0] = (m[4] * m[8] - m[5] * m[7]) * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = (m[5] * m[6] - m[3] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return 1;
}

/* ----------------------------------------------------------------- EPnP */

typedef struct {
    int n;
    double fu, fv, uc, vc;
    const double *pws; /* n x 3 */
    const double *us;  /* n x 2 (pixels) */
    double *alphas;    /* n x 4 */
    double *pcs;       /* n x 3 */
    double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

/* [OCV-recall] epnp::choose_control_points */
static void epnp_control_points(epnp_t *e)
{
    int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; j++) e->cws[0][j] /= n;
    double ptp[9] = {0}, dc[3], uct[9];
    for (int i = 0; i < n; i++) {
        double d[3];
        for (int j = 0; j < 3; j++) d[j] = e->pws[3 * i + j] - e->cws[0][j];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) ptp[a * 3 + b] += d[a] * d[b];
    }
    jacobi_eig_desc(ptp, 3, dc, uct);
    for (int i = 1; i < 4; i++) {
        double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0) / n);
        for (int j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
    }
}

/* [OCV-recall] epnp::compute_barycentric_coordinates */
static int epnp_barycentric(epnp_t *e)
{
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    if (!invert3(cc, ci)) return 0;
    for (int i = 0; i < e->n; i++) {
        const double *pi = e->pws + 3 * i;
        double *a = e->alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    return 1;
}

static void epnp_compute_ccs_pcs(epnp_t *e, const double *betas, const double *ut)
{
    for (int i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0;
    for (int i = 0; i < 4; i++) {
        const double *v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < e->n; i++) {
        const double *a = e->alphas + 4 * i;
        double *pc = e->pcs + 3 * i;
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
    if (e->pcs[2] < 0.0) { /* solve_for_sign */
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < 3 * e->n; i++) e->pcs[i] = -e->pcs[i];
    }
}

/* [OCV-recall] epnp::estimate_R_and_t + reprojection_error */
static double epnp_R_and_t(epnp_t *e, const double *ut, const double *betas, double R[9], double t[3])
{
    int n = e->n;
    epnp_compute_ccs_pcs(e, betas, ut);
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0};
    for (int i = 0; i < n; i++) {
        const double *pc = e->pcs + 3 * i, *pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    polar_rotation3(abt, R);
    double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] -
                 R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - dot3(R, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    double sum2 = 0;
    for (int i = 0; i < n; i++) {
        const double *pw = e->pws + 3 * i;
        double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

/* [OCV-recall] epnp::gauss_newton (5 iterations, 6x4 least squares) */
static void epnp_gauss_newton(const double *L, const double *rho, double betas[4])
{
    for (int k = 0; k < 5; k++) {
        double A[24], b[6], x[4];
        for (int i = 0; i < 6; i++) {
            const double *rowL = L + i * 10;
            double *rowA = A + i * 4;
            rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
            rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
            rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
            rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
            b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] +
                             rowL[2] * betas[1] * betas[1] + rowL[3] * betas[0] * betas[2] +
                             rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                             rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] +
                             rowL[8] * betas[2] * betas[3] + rowL[9] * betas[3] * betas[3]);
        }
        lstsq_qr(A, b, 6, 4, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

/* [OCV-recall] epnp::compute_pose.  obj: n x 3, img: n x 2 in pixels. */
static int epnp_solve(const double *obj, const double *img, int n, double fu, double fv, double uc,
                      double vc, double R[9], double t[3])
{
    epnp_t e;
    e.n = n; e.fu = fu; e.fv = fv; e.uc = uc; e.vc = vc; e.pws = obj; e.us = img;
    e.alphas = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    e.pcs = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    int ok = 0;
    epnp_control_points(&e);
    if (epnp_barycentric(&e)) {
        /* cv: SVD of M^T M (12x12); here the right singular vectors of the
         * 2n x 12 matrix M itself (same vectors, better conditioned) */
        double d[12], ut[144];
        if (n == 5) {
            double Mcm[144];
            memset(Mcm, 0, sizeof(Mcm));
            for (int i = 0; i < n; i++) {
                const double *as = e.alphas + 4 * i;
                for (int k = 0; k < 4; k++) {
                    Mcm[12 * (3 * k) + 2 * i] = as[k] * fu;
                    Mcm[12 * (3 * k + 1) + 2 * i + 1] = as[k] * fv;
                    Mcm[12 * (3 * k + 2) + 2 * i] = as[k] * (uc - img[2 * i]);
                    Mcm[12 * (3 * k + 2) + 2 * i + 1] = as[k] * (vc - img[2 * i + 1]);
                }
            }
            svd12_onesided(Mcm, 10, d, ut);
        } else {
            double mtm[144] = {0};
            for (int i = 0; i < n; i++) {
                const double *as = e.alphas + 4 * i;
                double m1[12], m2[12];
                for (int k = 0; k < 4; k++) {
                    m1[3 * k] = as[k] * fu; m1[3 * k + 1] = 0.0; m1[3 * k + 2] = as[k] * (uc - img[2 * i]);
                    m2[3 * k] = 0.0; m2[3 * k + 1] = as[k] * fv; m2[3 * k + 2] = as[k] * (vc - img[2 * i + 1]);
                }
                for (int a = 0; a < 12; a++)
                    for (int b = 0; b < 12; b++) mtm[a * 12 + b] += m1[a] * m1[b] + m2[a] * m2[b];
            }
            double mcm[144];
            for (int a = 0; a < 12; a++)
                for (int b = 0; b < 12; b++) mcm[12 * b + a] = mtm[a * 12 + b];
            svd12_onesided(mcm, 12, d, ut);
        }

        /* compute_L_6x10 / compute_rho */
        double L[60], rho[6];
        const double *v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
                dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
                dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; i++) {
            double *row = L + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
        rho[0] = dist2(e.cws[0], e.cws[1]); rho[1] = dist2(e.cws[0], e.cws[2]);
        rho[2] = dist2(e.cws[0], e.cws[3]); rho[3] = dist2(e.cws[1], e.cws[2]);
        rho[4] = dist2(e.cws[1], e.cws[3]); rho[5] = dist2(e.cws[2], e.cws[3]);

        double Betas[4][4] = {{0}}, rep[4] = {0}, Rs[4][9], ts[4][3];
        /* approx 1: [B11 B12 B13 B14] */
        {
            double l[24], b4[4];
            for (int i = 0; i < 6; i++) {
                l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1];
                l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6];
            }
            lstsq_qr(l, rho, 6, 4, b4);
            double *be = Betas[1];
            if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
            else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
        }
        epnp_gauss_newton(L, rho, Betas[1]);
        rep[1] = epnp_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
        /* approx 2: [B11 B12 B22] */
        {
            double l[18], b3[3];
            for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
            lstsq_qr(l, rho, 6, 3, b3);
            double *be = Betas[2];
            if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
            else { be[0] = sqrt(b3[0]); be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
            if (b3[1] < 0) be[0] = -be[0];
            be[2] = 0.0; be[3] = 0.0;
        }
        epnp_gauss_newton(L, rho, Betas[2]);
        rep[2] = epnp_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
        /* approx 3: [B11 B12 B22 B13 B23] */
        {
            double l[30], b5[5];
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 5; j++) l[i * 5 + j] = L[i * 10 + j];
            lstsq_qr(l, rho, 6, 5, b5);
            double *be = Betas[3];
            if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
            else { be[0] = sqrt(b5[0]); be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) be[0] = -be[0];
            be[2] = b5[3] / be[0];
            be[3] = 0.0;
        }
        epnp_gauss_newton(L, rho, Betas[3]);
        rep[3] = epnp_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);

        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        memcpy(R, Rs[N], sizeof(double) * 9);
        memcpy(t, ts[N], sizeof(double) * 3);
        ok = 1;
        for (int i = 0; i < 9; i++) if (!isfinite(R[i])) ok = 0;
        for (int i = 0; i < 3; i++) if (!isfinite(t[i])) ok = 0;
    }
    free(e.alphas); free(e.pcs);
    return ok;
}

int ora_epnp(const double *obj, const double *img_px, int n, double *R, double *t)
{
    /* img_px are pixels of a camera with fu = fv = 1, uc = vc = 0 (normalised) */
    return epnp_solve(obj, img_px, n, 1.0, 1.0, 0.0, 0.0, R, t);
}

/* [OCV-recall] cv::Rodrigues, matrix -> vector.  OpenCV first projects the
 * input on SO(3) (SVD); the solvers here only pass rotations. */
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

/* ---------------------------------------------- iterative refinement */

/* residuals (proj - measured) and optionally the 2n x 6 Jacobian w.r.t.
 * (rvec, tvec); [OCV-recall] cvProjectPoints2 without distortion */
static void project_residuals(const double *obj, const double *img, int n, double fx, double fy,
                              double cx, double cy, const double *param, double *err, double *J)
{
    double R[9], dRdr[27];
    rodrigues_v2m(param, R, J ? dRdr : 0);
    const double *t = param + 3;
    for (int i = 0; i < n; i++) {
        const double *M = obj + 3 * i;
        double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
        double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
        double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
        double z = Z ? 1. / Z : 1;
        double x = X * z, y = Y * z;
        err[2 * i] = x * fx + cx - img[2 * i];
        err[2 * i + 1] = y * fy + cy - img[2 * i + 1];
        if (J) {
            double *j0 = J + (size_t)(2 * i) * 6, *j1 = j0 + 6;
            /* d/dt */
            j0[3] = fx * z; j0[4] = 0; j0[5] = -fx * x * z;
            j1[3] = 0; j1[4] = fy * z; j1[5] = -fy * y * z;
            /* d/dr: dX/dr_k = dR/dr_k * M */
            for (int k = 0; k < 3; k++) {
                const double *d = dRdr + 9 * k;
                double dX = d[0] * M[0] + d[1] * M[1] + d[2] * M[2];
                double dY = d[3] * M[0] + d[4] * M[1] + d[5] * M[2];
                double dZ = d[6] * M[0] + d[7] * M[1] + d[8] * M[2];
                j0[k] = fx * (dX - x * dZ) * z;
                j1[k] = fy * (dY - y * dZ) * z;
            }
        }
    }
}

/* ------------------------------------------------ homography (planar init) */

/* general n x n solve by Gaussian elimination with partial pivoting (n <= 8);
 * returns 0 when singular */
static int solve_gauss_n(const double *Ain, const double *bin, int n, double *x)
{
    double A[64], b[8];
    memcpy(A, Ain, sizeof(double) * (size_t)(n * n));
    memcpy(b, bin, sizeof(double) * (size_t)n);
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[piv * n + k])) piv = i;
        if (A[piv * n + k] == 0) return 0;
        if (piv != k) {
            for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = s / A[k * n + k];
    }
    return 1;
}

/* projection residuals of the 8-parameter homography (h[8] = 1) and the
 * accumulated normal equations; [OCV-recall] HomographyRefineCallback */
static double homography_normal_eq(const float *M, const float *m, int n, const double *h, double *JtJ, double *Jtr)
{
    double S = 0;
    if (JtJ) { memset(JtJ, 0, sizeof(double) * 64); memset(Jtr, 0, sizeof(double) * 8); }
    for (int i = 0; i < n; i++) {
        double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        double e0 = xi - m[2 * i], e1 = yi - m[2 * i + 1];
        S += e0 * e0 + e1 * e1;
        if (JtJ) {
            double j0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            double j1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            for (int a = 0; a < 8; a++) {
                Jtr[a] += j0[a] * e0 + j1[a] * e1;
                for (int b = 0; b < 8; b++) JtJ[a * 8 + b] += j0[a] * j0[b] + j1[a] * j1[b];
            }
        }
    }
    return S;
}

/* [OCV-recall] cv::findHomography(M, m, method = 0) on float32 points: the
 * normalised DLT of HomographyEstimatorCallback::runKernel (translation to the
 * centroid, per-axis scale count / sum |d|, 9x9 normal matrix, eigenvector of
 * the smallest eigenvalue, H / H[8]) followed, for more than four points, by
 * the Levenberg-Marquardt polish of LMSolver (10 iterations, eps FLT_EPSILON;
 * its 8x8 solves are done here by Gaussian elimination instead of DECOMP_EIG).
 * Returns 0 when the point set is degenerate. */
static int find_homography_ls(const float *M, const float *m, int n, double *H)
{
    double cMx = 0, cMy = 0, cmx = 0, cmy = 0, sMx = 0, sMy = 0, smx = 0, smy = 0;
    for (int i = 0; i < n; i++) { cmx += m[2 * i]; cmy += m[2 * i + 1]; cMx += M[2 * i]; cMy += M[2 * i + 1]; }
    cmx /= n; cmy /= n; cMx /= n; cMy /= n;
    for (int i = 0; i < n; i++) {
        smx += fabs(m[2 * i] - cmx); smy += fabs(m[2 * i + 1] - cmy);
        sMx += fabs(M[2 * i] - cMx); sMy += fabs(M[2 * i + 1] - cMy);
    }
    if (fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON || fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON)
        return 0;
    smx = n / smx; smy = n / smy; sMx = n / sMx; sMy = n / sMy;
    double LtL[81] = {0}, H0[9];
    for (int i = 0; i < n; i++) {
        double x = (m[2 * i] - cmx) * smx, y = (m[2 * i + 1] - cmy) * smy;
        double X = (M[2 * i] - cMx) * sMx, Y = (M[2 * i + 1] - cMy) * sMy;
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    for (int j = 0; j < 9; j++)
        for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
    /* [OCV-recall] cv::eigen(LtL) and its last eigenvector; here the same vector by inverse iteration (the
     * result is divided by its last entry below, so sign and scale do not matter) */
    smallest_eigvec_n(LtL, 9, H0);
    double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
    double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[i * 3 + j] = invHnorm[i * 3] * H0[j] + invHnorm[i * 3 + 1] * H0[3 + j] + invHnorm[i * 3 + 2] * H0[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            H[i * 3 + j] = T[i * 3] * Hnorm2[j] + T[i * 3 + 1] * Hnorm2[3 + j] + T[i * 3 + 2] * Hnorm2[6 + j];
    double sc = 1. / H[8];
    for (int i = 0; i < 9; i++) H[i] *= sc;
    if (n <= 4) return 1;

    /* [OCV-recall] LMSolver run: damping lambda * diag(J^T J) fixed at the start, gain-ratio update */
    double x[8], xd[8], A[64], v[8], D[8], Ap[64], d[8];
    memcpy(x, H, sizeof(x));
    double S = homography_normal_eq(M, m, n, x, A, v);
    for (int i = 0; i < 8; i++) D[i] = A[i * 8 + i];
    const double Rlo = 0.25, Rhi = 0.75, epsx = FLT_EPSILON, epsf = FLT_EPSILON;
    double lambda = 1, lc = 0.75;
    double rinf = 0;
    for (int iter = 0;;) {
        memcpy(Ap, A, sizeof(Ap));
        for (int i = 0; i < 8; i++) Ap[i * 8 + i] += lambda * D[i];
        if (!solve_gauss_n(Ap, v, 8, d)) memset(d, 0, sizeof(d));
        for (int i = 0; i < 8; i++) xd[i] = x[i] - d[i];
        double Sd = homography_normal_eq(M, m, n, xd, 0, 0);
        double dS = 0, t = 0;
        for (int i = 0; i < 8; i++) {
            double td = 2 * v[i];
            for (int j = 0; j < 8; j++) td -= A[i * 8 + j] * d[j];
            dS += d[i] * td;
            t += d[i] * v[i];
        }
        double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (Rr > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (Rr < Rlo) {
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
            if (lambda == 0) {
                /* 1 / largest diagonal entry of A^-1 */
                double maxval = DBL_EPSILON;
                for (int c = 0; c < 8; c++) {
                    double e[8] = {0}, col[8];
                    e[c] = 1;
                    if (solve_gauss_n(A, e, 8, col) && fabs(col[c]) > maxval) maxval = fabs(col[c]);
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof(x));
            S = homography_normal_eq(M, m, n, x, A, v);
        }
        /* |r|_inf of the current residual vector */
        rinf = 0;
        for (int i = 0; i < n; i++) {
            double Mx = M[2 * i], My = M[2 * i + 1];
            double ww = x[6] * Mx + x[7] * My + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            double e0 = fabs((x[0] * Mx + x[1] * My + x[2]) * ww - m[2 * i]);
            double e1 = fabs((x[3] * Mx + x[4] * My + x[5]) * ww - m[2 * i + 1]);
            if (e0 > rinf) rinf = e0;
            if (e1 > rinf) rinf = e1;
        }
        double dinf = 0;
        for (int i = 0; i < 8; i++) if (fabs(d[i]) > dinf) dinf = fabs(d[i]);
        iter++;
        if (!(iter < 10 && dinf >= epsx && rinf >= epsf)) break;
    }
    memcpy(H, x, sizeof(x));
    H[8] = 1;
    return 1;
}

/* [OCV-recall] the planar branch of cvFindExtrinsicCameraParams2: the object
 * points are rotated into their own plane (rows of Vt = principal axes, Mc the
 * centroid), a homography to the normalised image points is fitted and split
 * into a rotation and a translation.  param receives rvec, tvec. */
static void planar_pose_init(const double *obj, const double *img, int n, double fx, double fy, double cx,
                             double cy, const double *Mc, const double *Vt_in, double *param)
{
    double Rt[9], T[3];
    memcpy(Rt, Vt_in, sizeof(Rt));
    if (Rt[2] * Rt[2] + Rt[5] * Rt[5] < 1e-10) {
        for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    double det = Rt[0] * (Rt[4] * Rt[8] - Rt[5] * Rt[7]) - Rt[1] * (Rt[3] * Rt[8] - Rt[5] * Rt[6]) +
                 Rt[2] * (Rt[3] * Rt[7] - Rt[4] * Rt[6]);
    if (det < 0)
        for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
    for (int i = 0; i < 3; i++) T[i] = -(Rt[3 * i] * Mc[0] + Rt[3 * i + 1] * Mc[1] + Rt[3 * i + 2] * Mc[2]);
    float *Mxy = (float *)malloc(sizeof(float) * 2 * (size_t)n), *mn = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        const double *s = obj + 3 * i;
        /* cv::findHomography converts its inputs to float32 */
        Mxy[2 * i] = (float)(Rt[0] * s[0] + Rt[1] * s[1] + Rt[2] * s[2] + T[0]);
        Mxy[2 * i + 1] = (float)(Rt[3] * s[0] + Rt[4] * s[1] + Rt[5] * s[2] + T[1]);
        mn[2 * i] = (float)((img[2 * i] - cx) / fx);
        mn[2 * i + 1] = (float)((img[2 * i + 1] - cy) / fy);
    }
    double h[9], R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    int okH = find_homography_ls(Mxy, mn, n, h);
    for (int i = 0; i < 9 && okH; i++) if (!isfinite(h[i])) okH = 0;
    if (okH) {
        double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
        double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
        double s1 = 1. / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), s2 = 1. / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
        double s3 = 2. / (h1n + h2n > DBL_EPSILON ? h1n + h2n : DBL_EPSILON);
        double a[3] = {h[0] * s1, h[3] * s1, h[6] * s1}, b[3] = {h[1] * s2, h[4] * s2, h[7] * s2};
        double tt[3] = {h[2] * s3, h[5] * s3, h[8] * s3};
        double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        double Hm[9] = {a[0], b[0], c[0], a[1], b[1], c[1], a[2], b[2], c[2]}, Hr[9], rv[3];
        /* Rodrigues there and back: the nearest rotation */
        polar_rotation3(Hm, Hr);
        ora_rodrigues_mat2vec(Hr, rv);
        rodrigues_v2m(rv, Hr, 0);
        for (int i = 0; i < 3; i++) t[i] = Hr[3 * i] * T[0] + Hr[3 * i + 1] * T[1] + Hr[3 * i + 2] * T[2] + tt[i];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                R[i * 3 + j] = Hr[i * 3] * Rt[j] + Hr[i * 3 + 1] * Rt[3 + j] + Hr[i * 3 + 2] * Rt[6 + j];
    }
    ora_rodrigues_mat2vec(R, param);
    for (int i = 0; i < 3; i++) param[3 + i] = t[i];
    free(Mxy); free(mn);
}

/* [OCV-recall] cvFindExtrinsicCameraParams2 without extrinsic guess:
 * initialisation from a homography when the object points are (nearly) planar
 * (>= 4 points), else by DLT (>= 6 points; OpenCV throws below that and the
 * function returns 0), then CvLevMarq (<= 20 iterations, eps FLT_EPSILON).
 * obj n x 3, img n x 2 pixels. */
int ora_pnp_iterative(const double *obj, const double *img, int n, double fx, double fy, double cx,
                      double cy, double *rvec, double *tvec)
{
    if (n < 4) return 0;
    /* planarity test on the object points */
    double Mc[3] = {0, 0, 0}, MM[9] = {0}, W[3], Vt[9];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) Mc[j] += obj[3 * i + j];
    for (int j = 0; j < 3; j++) Mc[j] /= n;
    for (int i = 0; i < n; i++) {
        double d[3] = {obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2]};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += d[a] * d[b];
    }
    jacobi_eig_desc(MM, 3, W, Vt);
    double param[6], prev[6];
    if (W[2] / W[1] < 1e-3) {
        planar_pose_init(obj, img, n, fx, fy, cx, cy, Mc, Vt, param);
    } else {
    if (n < 6) return 0; /* "DLT algorithm needs at least 6 points" */
    /* DLT on normalised image points */
    double LL[144] = {0};
    for (int i = 0; i < n; i++) {
        const double *M = obj + 3 * i;
        double x = -(img[2 * i] - cx) / fx, y = -(img[2 * i + 1] - cy) / fy;
        double r0[12] = {M[0], M[1], M[2], 1, 0, 0, 0, 0, x * M[0], x * M[1], x * M[2], x};
        double r1[12] = {0, 0, 0, 0, M[0], M[1], M[2], 1, y * M[0], y * M[1], y * M[2], y};
        for (int a = 0; a < 12; a++)
            for (int b = 0; b < 12; b++) LL[a * 12 + b] += r0[a] * r0[b] + r1[a] * r1[b];
    }
    double RRt[12];
    smallest_eigvec12(LL, RRt);
    double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
    double tt[3] = {RRt[3], RRt[7], RRt[11]};
    double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
                 RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
    if (det < 0) {
        for (int i = 0; i < 9; i++) RR[i] = -RR[i];
        for (int i = 0; i < 3; i++) tt[i] = -tt[i];
    }
    double sc = 0;
    for (int i = 0; i < 9; i++) sc += RR[i] * RR[i];
    sc = sqrt(sc);
    if (fabs(sc) <= DBL_EPSILON) return 0;
    double R[9];
    polar_rotation3(RR, R);
    double nR = 0;
    for (int i = 0; i < 9; i++) nR += R[i] * R[i];
    nR = sqrt(nR);
    ora_rodrigues_mat2vec(R, param);
    for (int i = 0; i < 3; i++) param[3 + i] = tt[i] * nR / sc;
    }

    /* CvLevMarq */
    const int max_iter = 20;
    const double eps = FLT_EPSILON;
    double *err = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *J = (double *)malloc(sizeof(double) * 12 * (size_t)n);
    double JtJ[36], JtErr[6];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0;
    const double LOG10 = log(10.);
    for (;;) {
        /* CALC_J */
        project_residuals(obj, img, n, fx, fy, cx, cy, param, err, J);
        memset(JtJ, 0, sizeof(JtJ));
        memset(JtErr, 0, sizeof(JtErr));
        double e2 = 0;
        for (int i = 0; i < 2 * n; i++) {
            const double *j = J + (size_t)i * 6;
            for (int a = 0; a < 6; a++) {
                JtErr[a] += j[a] * err[i];
                for (int b = 0; b < 6; b++) JtJ[a * 6 + b] += j[a] * j[b];
            }
            e2 += err[i] * err[i];
        }
        memcpy(prev, param, sizeof(prev));
        if (iters == 0) prevErrNorm = sqrt(e2);
        int accepted = 0;
        for (;;) {
            /* step() */
            double lambda = exp(lambdaLg10 * LOG10);
            double A[36], dx[6];
            memcpy(A, JtJ, sizeof(A));
            for (int a = 0; a < 6; a++) A[a * 6 + a] *= 1. + lambda;
            if (!solve_gauss(A, JtErr, 6, dx)) memset(dx, 0, sizeof(dx));
            for (int a = 0; a < 6; a++) param[a] = prev[a] - dx[a];
            /* CHECK_ERR */
            project_residuals(obj, img, n, fx, fy, cx, cy, param, err, 0);
            e2 = 0;
            for (int i = 0; i < 2 * n; i++) e2 += err[i] * err[i];
            errNorm = sqrt(e2);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) continue;
            }
            accepted = 1;
            break;
        }
        (void)accepted;
        lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
        double dn = 0, pn = 0;
        for (int a = 0; a < 6; a++) { dn += (param[a] - prev[a]) * (param[a] - prev[a]); pn += prev[a] * prev[a]; }
        if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < eps) break;
        prevErrNorm = errNorm;
    }
    free(err); free(J);
    memcpy(rvec, param, sizeof(double) * 3);
    memcpy(tvec, param + 3, sizeof(double) * 3);
    return 1;
}

/* ------------------------------------------------------------------ P3P */

static double p3p_horner(const double *p, int deg, double x)
{
    double v = p[deg];
    for (int i = deg - 1; i >= 0; i--) v = v * x + p[i];
    return v;
}

/* ascending real roots (with sign change) of a polynomial of degree <= 4:
 * the roots of each derivative split [-B, B] (Cauchy bound) into monotone
 * pieces that are bisected until the interval stops shrinking */
static int p3p_real_roots4(const double *p_in, double *roots)
{
    int deg = 4;
    while (deg > 0 && p_in[deg] == 0) deg--;
    if (deg <= 0) return 0;
    double P[5][5];
    for (int i = 0; i <= deg; i++) P[deg][i] = p_in[i];
    for (int d = deg; d > 1; d--)
        for (int i = 1; i <= d; i++) P[d - 1][i - 1] = P[d][i] * i;
    double crit[4], cur[4];
    int nc = 1;
    crit[0] = -P[1][0] / P[1][1];
    for (int d = 2; d <= deg; d++) {
        const double *p = P[d];
        double bound = 0;
        for (int i = 0; i < d; i++) bound = fmax(bound, fabs(p[i] / p[d]));
        bound += 1.0;
        int n = 0;
        double lo = -bound, flo = p3p_horner(p, d, lo);
        for (int k = 0; k <= nc; k++) {
            double hi = k < nc ? crit[k] : bound;
            if (hi <= lo) continue;
            if (hi > bound) hi = bound;
            const double fhi = p3p_horner(p, d, hi);
            if (flo == 0) {
                cur[n++] = lo;
            } else if (fhi != 0 && (flo < 0) != (fhi < 0)) {
                double a = lo, b = hi, fa = flo;
                int exact = 0;
                for (int it = 0; it < 200; it++) {
                    const double mid = 0.5 * (a + b);
                    if (mid == a || mid == b) break;
                    const double fm = p3p_horner(p, d, mid);
                    if (fm == 0) { a = b = mid; exact = 1; break; }
                    if ((fm < 0) == (fa < 0)) { a = mid; fa = fm; }
                    else b = mid;
                }
                cur[n++] = exact ? a : 0.5 * (a + b);
            }
            lo = hi;
            flo = fhi;
        }
        if (flo == 0 && n < d) cur[n++] = lo;
        nc = n;
        for (int i = 0; i < n; i++) crit[i] = cur[i];
    }
    for (int i = 0; i < nc; i++) roots[i] = crit[i];
    return nc;
}

/* orthonormal frame of a triangle: e1 along p1 - p0, e3 the normal, e2 = e3 x e1 (rows of F) */
static int p3p_triangle_frame(const double *p0, const double *p1, const double *p2, double *F)
{
    double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    double na = sqrt(dot3(a, a));
    if (na == 0) return 0;
    for (int i = 0; i < 3; i++) a[i] /= na;
    double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    double nc = sqrt(dot3(c, c));
    if (nc == 0) return 0;
    for (int i = 0; i < 3; i++) c[i] /= nc;
    double e2[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
    for (int i = 0; i < 3; i++) { F[i] = a[i]; F[3 + i] = e2[i]; F[6 + i] = c[i]; }
    return 1;
}

/* cv::solvePnP(SOLVEPNP_P3P) on exactly four correspondences, as
 * cv::solvePnPRansac runs it when it is handed four points: the (up to four)
 * poses consistent with the first three points, of which the one that
 * reprojects the fourth point best is kept.  OpenCV's p3p.cpp follows Gao et
 * al.; the same solution set is computed here from Grunert's quartic in
 * v = s3 / s1 (si: distance of point i from the camera), its real roots by
 * bisection, and the rigid motion between the two triangles.
 * obj: 4 x 3, img: 4 x 2 pixels.  Returns 1 and R (row-major), t on success. */
static int p3p_solve4(const double *obj, const double *img, double fx, double fy, double cx, double cy, double *R,
                      double *t)
{
    double j[3][3];
    for (int i = 0; i < 3; i++) {
        double x = (img[2 * i] - cx) / fx, y = (img[2 * i + 1] - cy) / fy;
        double nrm = sqrt(x * x + y * y + 1.0);
        j[i][0] = x / nrm; j[i][1] = y / nrm; j[i][2] = 1.0 / nrm;
    }
    const double a2 = dist2(obj + 3, obj + 6), b2 = dist2(obj, obj + 6), c2 = dist2(obj, obj + 3);
    if (a2 == 0 || b2 == 0 || c2 == 0) return 0;
    const double ca = dot3(j[1], j[2]), cb = dot3(j[0], j[2]), cg = dot3(j[0], j[1]);
    const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2;
    double q[5];
    q[4] = (k1 - 1) * (k1 - 1) - 4 * c2 / b2 * ca * ca;
    q[3] = 4 * (k1 * (1 - k1) * cb - (1 - k2) * ca * cg + 2 * c2 / b2 * ca * ca * cb);
    q[2] = 2 * (k1 * k1 - 1 + 2 * k1 * k1 * cb * cb + 2 * (b2 - c2) / b2 * ca * ca - 4 * k2 * ca * cb * cg +
                2 * (b2 - a2) / b2 * cg * cg);
    q[1] = 4 * (-k1 * (1 + k1) * cb + 2 * a2 / b2 * cg * cg * cb - (1 - k2) * ca * cg);
    q[0] = (1 + k1) * (1 + k1) - 4 * a2 / b2 * cg * cg;
    double roots[4];
    const int nr = p3p_real_roots4(q, roots);
    double FX[9];
    if (!p3p_triangle_frame(obj, obj + 3, obj + 6, FX)) return 0;
    const double Xc[3] = {(obj[0] + obj[3] + obj[6]) / 3, (obj[1] + obj[4] + obj[7]) / 3, (obj[2] + obj[5] + obj[8]) / 3};
    int found = 0;
    double best = 0;
    for (int r = 0; r < nr; r++) {
        const double v = roots[r];
        if (!(v > 0)) continue;
        const double den = 2 * (cg - v * ca);
        if (den == 0) continue;
        const double u = ((k1 - 1) * v * v - 2 * k1 * cb * v + 1 + k1) / den;
        const double w = 1 + v * v - 2 * v * cb;
        if (!(u > 0) || !(w > 0)) continue;
        const double s1 = sqrt(b2 / w), s2 = u * s1, s3 = v * s1;
        double Y[9];
        for (int i = 0; i < 3; i++) { Y[i] = s1 * j[0][i]; Y[3 + i] = s2 * j[1][i]; Y[6 + i] = s3 * j[2][i]; }
        double FY[9];
        if (!p3p_triangle_frame(Y, Y + 3, Y + 6, FY)) continue;
        double Rc[9], tc[3];
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) Rc[i * 3 + k] = FY[i] * FX[k] + FY[3 + i] * FX[3 + k] + FY[6 + i] * FX[6 + k];
        for (int i = 0; i < 3; i++) tc[i] = (Y[i] + Y[3 + i] + Y[6 + i]) / 3 - dot3(Rc + 3 * i, Xc);
        /* the fourth point decides */
        const double *X3 = obj + 9;
        const double Xp = dot3(Rc, X3) + tc[0], Yp = dot3(Rc + 3, X3) + tc[1], Zp = dot3(Rc + 6, X3) + tc[2];
        const double du = cx + fx * Xp / Zp - img[6], dv = cy + fy * Yp / Zp - img[7];
        const double e = du * du + dv * dv;
        if (!isfinite(e)) continue;
        if (!found || e < best) {
            found = 1;
            best = e;
            memcpy(R, Rc, sizeof(Rc));
            memcpy(t, tc, sizeof(tc));
        }
    }
    return found;
}

int ora_p3p(const double *obj, const double *img, double fx, double fy, double cx, double cy, double *R, double *t)
{
    return p3p_solve4(obj, img, fx, fy, cx, cy, R, t);
}

/* ------------------------------------------------------- RANSAC driver */

/* [OCV-recall] RANSACUpdateNumIters */
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

/* [OCV-recall] PnPRansacCallback::computeError: projectPoints in double,
 * stored as float, squared pixel distance */
static void pnp_errors(const float *obj, const float *img, int n, double fx, double fy, double cx,
                       double cy, const double *rvec, const double *tvec, float *err)
{
    double R[9];
    rodrigues_v2m(rvec, R, 0);
    for (int i = 0; i < n; i++) {
        double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
        float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
        err[i] = (float)((double)dx * dx + (double)dy * dy);
    }
}

/* [OCV-recall] cv::solvePnPRansac, EPnP on 5-point samples (n > 5), final
 * iterative refinement on the inliers.  With n == 5 OpenCV runs the minimal
 * solver once, with n == 4 its P3P solver. */
int ora_solve_pnp_ransac(const float *obj, const float *img, int n, double fx, double fy, double cx,
                         double cy, int iters, double thr, double conf, double *rvec, double *tvec,
                         int32_t *inliers, int *n_inliers)
{
    const int model_points = 5;
    if (n_inliers) *n_inliers = 0;
    for (int i = 0; i < 3; i++) rvec[i] = tvec[i] = 0;
    double so[15], si[10], R[9], t[3];
    if (n == 4) {
        /* four points: no RANSAC, one P3P solve, every point reported as inlier */
        for (int i = 0; i < 4; i++) {
            for (int j = 0; j < 3; j++) so[3 * i + j] = obj[3 * i + j];
            si[2 * i] = img[2 * i]; si[2 * i + 1] = img[2 * i + 1];
        }
        if (!p3p_solve4(so, si, fx, fy, cx, cy, R, t)) return 0;
        ora_rodrigues_mat2vec(R, rvec);
        memcpy(tvec, t, sizeof(t));
        for (int i = 0; i < 4; i++) inliers[i] = i;
        *n_inliers = 4;
        return 1;
    }
    if (n < model_points) return 0;

    if (n == model_points) {
        for (int i = 0; i < 5; i++) {
            for (int j = 0; j < 3; j++) so[3 * i + j] = obj[3 * i + j];
            si[2 * i] = img[2 * i]; si[2 * i + 1] = img[2 * i + 1];
        }
        if (!epnp_solve(so, si, 5, fx, fy, cx, cy, R, t)) return 0;
        ora_rodrigues_mat2vec(R, rvec);
        memcpy(tvec, t, sizeof(t));
        for (int i = 0; i < n; i++) inliers[i] = i;
        *n_inliers = n;
        return 1;
    }

    ora_rng rng = {(uint64_t)-1};
    int niters = iters > 1 ? iters : 1, max_good = 0;
    float *err = (float *)malloc(sizeof(float) * (size_t)n);
    uint8_t *mask = (uint8_t *)calloc((size_t)n, 1), *cur = (uint8_t *)malloc((size_t)n);
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    const float tt = (float)(thr * thr);
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        for (int i = 0; i < model_points; i++) {
            int v;
            for (;;) {
                v = ora_rng_uniform(&rng, 0, n);
                int dup = 0;
                for (int j = 0; j < i; j++) if (idx[j] == v) { dup = 1; break; }
                if (!dup) break;
            }
            idx[i] = v;
            for (int j = 0; j < 3; j++) so[3 * i + j] = obj[3 * v + j];
            si[2 * i] = img[2 * v]; si[2 * i + 1] = img[2 * v + 1];
        }
        if (!epnp_solve(so, si, 5, fx, fy, cx, cy, R, t)) continue;
        double rv[3];
        ora_rodrigues_mat2vec(R, rv);
        pnp_errors(obj, img, n, fx, fy, cx, cy, rv, t, err);
        int good = 0;
        for (int i = 0; i < n; i++) { cur[i] = (uint8_t)(err[i] <= tt); good += cur[i]; }
        if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
            memcpy(mask, cur, (size_t)n);
            memcpy(best_r, rv, sizeof(rv));
            memcpy(best_t, t, sizeof(t));
            max_good = good;
            niters = ransac_update_num_iters(conf, (double)(n - good) / n, model_points, niters);
        }
    }
    int ok = 0;
    if (max_good > 0) {
        /* refine on the inliers */
        double *io = (double *)malloc(sizeof(double) * 3 * (size_t)max_good);
        double *ii = (double *)malloc(sizeof(double) * 2 * (size_t)max_good);
        int k = 0;
        for (int i = 0; i < n; i++)
            if (mask[i]) {
                for (int j = 0; j < 3; j++) io[3 * k + j] = obj[3 * i + j];
                ii[2 * k] = img[2 * i]; ii[2 * k + 1] = img[2 * i + 1];
                k++;
            }
        int refined = ora_pnp_iterative(io, ii, k, fx, fy, cx, cy, rvec, tvec);
        if (!refined) {
            /* five non-planar inliers: the DLT initialisation throws "needs at least 6 points"; OpenCV >= 4.5.2
             * catches exactly that case and keeps the pose of the minimal sample set.  Anything else fails. */
            memcpy(rvec, best_r, sizeof(best_r));
            memcpy(tvec, best_t, sizeof(best_t));
            refined = k == 5;
        }
        if (refined) {
            ok = 1;
            k = 0;
            for (int i = 0; i < n; i++) if (mask[i]) inliers[k++] = i;
            *n_inliers = k;
        }
        free(io); free(ii);
    }
    free(err); free(mask); free(cur);
    return ok;
}
