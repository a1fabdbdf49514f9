// cv::solvePnPRansac for gfx950, one workgroup per frame
// (reference src/frame_processor.cpp:176-196): EPnP hypotheses from 5-point
// samples drawn with cv::RNG's stream, reprojection scoring, adaptive
// iteration bound, then solvePnP(SOLVEPNP_ITERATIVE) on the inliers (DLT
// initialisation + Levenberg-Marquardt).
//
// Layout of the work inside the block (128 threads = 2 waves; with 256 threads and sixteen hypotheses per round
// the frames that need many rounds finish sooner -- 1.05 against 1.45 ms per 512 frames when a third of the
// matches are outliers -- but every other phase is then executed by twice the waves and the common case, one
// round, goes from 0.31 to 0.48 ms):
//   * a hypothesis is solved by a group of 16 lanes: the 12x12 singular
//     vectors EPnP needs come from a one-sided Jacobi sweep in which lane j
//     owns column j (registers only, columns exchanged with wave shuffles,
//     the six disjoint column pairs of a round-robin round rotate at once);
//     8 hypotheses are in flight per round;
//   * every wave scores hypotheses against all points (fp64 projection,
//     float32 compare, as OpenCV does);
//   * one lane replays the sequential acceptance / adaptive-iteration logic;
//   * the refinement accumulates its normal equations with all threads and
//     a deterministic wave-shuffle + LDS reduction.
// fp64 throughout, evaluated with contraction disabled.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "vo_internal.h"

namespace vo {

constexpr int PR_THREADS = 128;
constexpr int PR_WAVES = PR_THREADS / kWave;
constexpr int PR_GROUP = 12;                      // lanes per hypothesis: one per column of EPnP's 10 x 12 matrix
constexpr int PR_PER_WAVE = kWave / PR_GROUP;     // hypotheses per wave (the last four lanes of a wave stay idle)
constexpr int PR_CHUNK = PR_WAVES * PR_PER_WAVE;  // hypotheses per round
constexpr int PR_FIRST = 8;                       // hypotheses of the first round
constexpr int PR_WIN = PR_CHUNK * 5 + 96;         // RNG words examined per round
constexpr int PR_NRED = 45;                       // 36 JtJ + 6 JtErr + |err|^2 (pose); 36 + 8 + 1 (homography)
constexpr int kPnpSpecMax = 128;                  // hypotheses per frame the second launch can hold
constexpr int kPnpItemsPerFrame = 16;             // work items (runs of up to PR_CHUNK hypotheses) per frame
constexpr int PR_WIN_ALL = kPnpSpecMax * 5 + 128;  // RNG words examined at once when a frame is handed over

// ---------------------------------------------------------------------
// small dense algebra, register resident (all indices static after unrolling)
// ---------------------------------------------------------------------

__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double dist2(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// sqrt(x) for x well inside the normal range (the callers: sums of squares that a test has found to be far from
// zero, or a squared cosine).  The reciprocal-square-root estimate refined by the two coupled Newton steps and the
// residual corrections that yield the correctly rounded root -- the compiler's own expansion of sqrt() without the
// rescaling of tiny arguments and the pass-through of 0 and infinity, which cannot occur here.
__device__ __forceinline__ double sqrt_normal(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, x);
    g = fma(e, h, g);
    e = fma(-g, g, x);
    return fma(e, h, g);
}

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix; eigenvalues
// descending in w, row i of Vt the matching eigenvector
__device__ __forceinline__ void eig3_desc(const double *Ain, double *w, double *Vt)
{
    constexpr int N = 3;
    double A[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < N; j++) {
                if (i != j) off += A[i * N + j] * A[i * N + j];
                else diag += A[i * N + j] * A[i * N + j];
            }
        if (off <= 1e-30 * diag || off == 0) break;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (fabs(apq) < 1e-300) continue;
                // tan and cos of the rotation angle as in group_svd12: d = a_qq - a_pp, h = 2 a_pq,
                // r = sqrt(d^2 + h^2), tan = sign(d) h / (|d| + r), cos^2 = (|d| + r) / (2 r), both from the one
                // reciprocal 1 / (2 r (|d| + r)) -- square root, reciprocal, square root in a row instead of two
                // divisions and a square root more
                const double d = A[q * N + q] - A[p * N + p], h = 2 * apq;
                const double x = d * d + h * h;
                if (!(x > 0)) continue;  // (both underflowed)
                const double rr = x > 1e-200 ? sqrt_normal(x) : sqrt(x);
                const double den = fabs(d) + rr, r2 = 2 * rr;
                const double wr = 1.0 / (r2 * den);
                const double sh = d < 0 ? -h : h;
                const double t = (sh * r2) * wr;
                const double c = sqrt_normal((den * den) * wr), s = t * c;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
    // exchange sort, descending, carrying the eigenvector columns along
    double ev[3] = {A[0], A[4], A[8]};
    double vc[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) vc[i][k] = V[k * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i + 1; j < 3; j++)
            if (ev[j] > ev[i]) {
                const double t = ev[i]; ev[i] = ev[j]; ev[j] = t;
#pragma unroll
                for (int k = 0; k < 3; k++) { const double u = vc[i][k]; vc[i][k] = vc[j][k]; vc[j][k] = u; }
            }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        w[i] = ev[i];
#pragma unroll
        for (int k = 0; k < 3; k++) Vt[i * 3 + k] = vc[i][k];
    }
}

// nearest rotation (U V^T of the SVD) of a 3x3 matrix
__device__ __forceinline__ void polar_rotation3(const double *A, double *R)
{
    double AtA[9], w[3], Vt[9], U[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = s;
        }
    eig3_desc(AtA, w, Vt);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double u[3], nrm = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = A[i * 3 + 0] * Vt[c * 3 + 0] + A[i * 3 + 1] * Vt[c * 3 + 1] + A[i * 3 + 2] * Vt[c * 3 + 2];
            nrm += u[i] * u[i];
        }
        nrm = sqrt(nrm);
        if (c == 2 && nrm <= 1e-12 * sqrt(w[0] > 0 ? w[0] : 1)) {
            u[0] = U[3] * U[7] - U[6] * U[4];
            u[1] = U[6] * U[1] - U[0] * U[7];
            u[2] = U[0] * U[4] - U[3] * U[1];
            nrm = 1;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) U[i * 3 + c] = u[i] / nrm;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = U[i * 3 + 0] * Vt[0 * 3 + j] + U[i * 3 + 1] * Vt[1 * 3 + j] + U[i * 3 + 2] * Vt[2 * 3 + j];
}

// least squares of a 6 x N system (N <= 5) by Householder QR
template <int N>
__device__ __forceinline__ void lstsq6(const double *Ain, const double *bin, double *x)
{
    constexpr int M = 6;
    double A[M * N], b[M];
#pragma unroll
    for (int i = 0; i < M * N; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < M; i++) b[i] = bin[i];
#pragma unroll
    for (int k = 0; k < N; k++) {
        double s = 0;
#pragma unroll
        for (int i = k; i < M; i++) s += A[i * N + k] * A[i * N + k];
        const double nrm = sqrt(s);
        if (nrm == 0) continue;
        const double alpha = A[k * N + k] > 0 ? -nrm : nrm;
        double v[M];
#pragma unroll
        for (int i = k; i < M; i++) v[i] = A[i * N + k];
        v[k] -= alpha;
        double vn = 0;
#pragma unroll
        for (int i = k; i < M; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        const double beta = 2.0 / vn;
#pragma unroll
        for (int j = k; j < N; j++) {
            double dot = 0;
#pragma unroll
            for (int i = k; i < M; i++) dot += v[i] * A[i * N + j];
            dot *= beta;
#pragma unroll
            for (int i = k; i < M; i++) A[i * N + j] -= dot * v[i];
        }
        double dot = 0;
#pragma unroll
        for (int i = k; i < M; i++) dot += v[i] * b[i];
        dot *= beta;
#pragma unroll
        for (int i = k; i < M; i++) b[i] -= dot * v[i];
    }
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < N; j++) s -= A[k * N + j] * x[j];
        x[k] = s / A[k * N + k];
    }
}

// lstsq6<5> on a 6 x 5 matrix whose columns >= ncols are zero: the Householder
// steps of the zero columns are no-ops, so the first ncols unknowns come out
// exactly as lstsq6<ncols> would give them; the others are set to 0.
__device__ __forceinline__ void lstsq6_pad5(const double *Ain, const double *bin, double *x, int ncols)
{
    constexpr int M = 6, N = 5;
    double A[M * N], b[M];
#pragma unroll
    for (int i = 0; i < M * N; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < M; i++) b[i] = bin[i];
#pragma unroll
    for (int k = 0; k < N; k++) {
        double s = 0;
#pragma unroll
        for (int i = k; i < M; i++) s += A[i * N + k] * A[i * N + k];
        const double nrm = sqrt(s);
        if (nrm == 0) continue;
        const double alpha = A[k * N + k] > 0 ? -nrm : nrm;
        double v[M];
#pragma unroll
        for (int i = k; i < M; i++) v[i] = A[i * N + k];
        v[k] -= alpha;
        double vn = 0;
#pragma unroll
        for (int i = k; i < M; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        const double beta = 2.0 / vn;
#pragma unroll
        for (int j = k; j < N; j++) {
            double dot = 0;
#pragma unroll
            for (int i = k; i < M; i++) dot += v[i] * A[i * N + j];
            dot *= beta;
#pragma unroll
            for (int i = k; i < M; i++) A[i * N + j] -= dot * v[i];
        }
        double dot = 0;
#pragma unroll
        for (int i = k; i < M; i++) dot += v[i] * b[i];
        dot *= beta;
#pragma unroll
        for (int i = k; i < M; i++) b[i] -= dot * v[i];
    }
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < N; j++) s -= A[k * N + j] * x[j];
        x[k] = k < ncols ? s / A[k * N + k] : 0.0;
    }
}

// 6x6 symmetric positive definite solve (Gaussian elimination, no pivoting).  One division per pivot: rows are
// scaled by the pivot's reciprocal.  (Only the pose refinement uses it, which is compared to a tolerance.)
__device__ __forceinline__ bool solve_spd6(const double *Ain, const double *bin, double *x)
{
#pragma clang fp contract(fast)
    constexpr int n = 6;
    double A[36], b[6], ip[6];
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] = bin[i];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < n; k++) {
        ok = ok && (A[k * n + k] != 0);
        ip[k] = 1.0 / A[k * n + k];
#pragma unroll
        for (int i = k + 1; i < n; i++) {
            const double f = A[i * n + k] * ip[k];
#pragma unroll
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
#pragma unroll
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = s * ip[k];
    }
    return ok;
}

__device__ __forceinline__ bool invert3(const double *m, double *inv)
{
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                       m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (det == 0) return false;
    const double id = 1.0 / det;
    inv[0] = (m[4] * m[8] - m[5] * m[7]) * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = (m[5] * m[6] - m[3] * m[8]) * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

// cv::Rodrigues vector -> matrix (+ optional 3x9 derivative)
__device__ void rodrigues_v2m(const double *r, double *R, double *J)
{
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = I[i];
        if (J) {
            for (int i = 0; i < 27; i++) J[i] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c;
    const double itheta = 1. / theta;
    const double n[3] = {r[0] * itheta, r[1] * itheta, r[2] * itheta};
    const double rrt[9] = {n[0] * n[0], n[0] * n[1], n[0] * n[2], n[0] * n[1], n[1] * n[1],
                           n[1] * n[2], n[0] * n[2], n[1] * n[2], n[2] * n[2]};
    const double r_x[9] = {0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        const double drrt[27] = {n[0] + n[0], n[1], n[2], n[1], 0, 0, n[2], 0, 0,
                                 0, n[0], 0, n[0], n[1] + n[1], n[2], 0, n[2], 0,
                                 0, 0, n[0], 0, 0, n[1], n[0], n[1], n[2] + n[2]};
        const double d_r_x[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            const double ri = n[i];
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x[i * 9 + k];
        }
    }
}

// cv::Rodrigues matrix -> vector for a rotation matrix
__device__ void rodrigues_m2v(const double *R, double *r)
{
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
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

// ---------------------------------------------------------------------
// one-sided Jacobi over a group of PR_GROUP lanes
// ---------------------------------------------------------------------

// partner of column x in round r of the 12-player round-robin tournament
__device__ __forceinline__ int rr_partner(int x, int r)
{
    if (x == 11) return r;
    if (x == r) return 11;
    int y = (2 * r - x) % 11;
    return y < 0 ? y + 11 : y;
}

// Right singular vectors of a ROWS x 12 matrix whose column gl lives in
// g[] of lane gl of each PR_GROUP-lane group (lanes past the last whole group idle).  On return v[] is the group's column
// gl of V and *norm2 the squared norm of the rotated column gl (a squared
// singular value); *rank is that value's position in descending order (ties
// by column index).  Every lane of the wave must call this.
template <int ROWS>
__device__ void group_svd12(double (&g)[ROWS], double (&v)[12], int gl, double *norm2, int *rank)
{
    const int lane = threadIdx.x & 63;
    const int grp = lane / PR_GROUP, base = grp * PR_GROUP;
    const bool col = gl < 12 && grp < PR_PER_WAVE;
#pragma unroll
    for (int i = 0; i < 12; i++) v[i] = (i == gl) ? 1.0 : 0.0;
    double own = 0;
#pragma unroll
    for (int i = 0; i < ROWS; i++) own += g[i] * g[i];
    double frob = 0;
    for (int j = 0; j < 12; j++) frob += __shfl(own, base + j);
    const double eps = 1e-15;

    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int r = 0; r < 11; r++) {
            const int y = col ? rr_partner(gl, r) : gl;
            const int src = base + y;
            double pg[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; i++) pg[i] = __shfl(g[i], src);
            const bool am_p = gl < y;
            // both squared norms are summed here, the partner's from the copy just fetched: it is the sum the partner
            // lane forms of its own column, term by term, and costs no further exchange
            double so = 0, sp = 0, gamma = 0;
#pragma unroll
            for (int i = 0; i < ROWS; i++) so = fma(g[i], g[i], so);
#pragma unroll
            for (int i = 0; i < ROWS; i++) sp = fma(pg[i], pg[i], sp);
#pragma unroll
            for (int i = 0; i < ROWS; i++) gamma = fma(g[i], pg[i], gamma);
            const double alpha = am_p ? so : sp, beta = am_p ? sp : so;
            // |gamma| <= eps sqrt(alpha beta), squared: the test then waits for no square root
            const bool skip = !col || gamma * gamma <= (eps * eps) * (alpha * beta) || fabs(gamma) <= 1e-26 * frob;
            if (!skip) {
                // (both lanes of a pair see the same alpha, beta and gamma -- a product does not depend on the order
                // of its factors -- so they are here together and can exchange their columns of V)
                double pv[12];
#pragma unroll
                for (int i = 0; i < 12; i++) pv[i] = __shfl(v[i], src);
                rotated = true;
                // tan and cos of the rotation angle: with d = beta - alpha, h = 2 gamma and r = sqrt(d^2 + h^2),
                // tan = sign(d) h / (|d| + r) and cos^2 = (|d| + r) / (2 r), both as multiples of the one reciprocal
                // 1 / (2 r (|d| + r)).  (At d = 0 either sign of tan annihilates gamma; + h is taken.)
                const double d = beta - alpha, h = 2 * gamma;
                const double rr = sqrt_normal(d * d + h * h);
                const double den = fabs(d) + rr, r2 = 2 * rr;
                const double w = 1.0 / (r2 * den);
                const double sh = d < 0 ? -h : h;
                const double t = (sh * r2) * w;
                const double c = sqrt_normal((den * den) * w), sn = c * t;
                // the lower column of the pair becomes c p - s q, the upper one c q + s p, each as the partner's
                // term fused onto the rounded product of the own one: one expression for both
                // ((-s) q and -(s q) are the same number)
                const double s2 = am_p ? -sn : sn;
#pragma unroll
                for (int i = 0; i < ROWS; i++) g[i] = fma(s2, pg[i], c * g[i]);
#pragma unroll
                for (int i = 0; i < 12; i++) v[i] = fma(s2, pv[i], c * v[i]);
            }
        }
        if (!__any(rotated)) break;
    }
    double n2 = 0;
#pragma unroll
    for (int i = 0; i < ROWS; i++) n2 += g[i] * g[i];
    int rk = 0;
    for (int j = 0; j < 12; j++) {
        const double o = __shfl(n2, base + j);
        rk += (o > n2 || (o == n2 && j < gl)) ? 1 : 0;
    }
    *norm2 = n2;
    *rank = rk;
}

// ---------------------------------------------------------------------
// EPnP on five correspondences
// ---------------------------------------------------------------------

struct Epnp5 {
    double fu, fv, uc, vc;
    double pws[15], us[10], alphas[20];
    double cws[4][3];
};

// control points and barycentric coordinates (epnp::choose_control_points,
// compute_barycentric_coordinates)
__device__ bool epnp5_prepare(Epnp5 &e)
{
    constexpr int n = 5;
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
#pragma unroll
    for (int j = 0; j < 3; j++) e.cws[0][j] /= n;
    double ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], uct[9];
#pragma unroll
    for (int i = 0; i < n; i++) {
        double d[3];
#pragma unroll
        for (int j = 0; j < 3; j++) d[j] = e.pws[3 * i + j] - e.cws[0][j];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) ptp[a * 3 + b] += d[a] * d[b];
    }
    eig3_desc(ptp, dc, uct);
#pragma unroll
    for (int i = 1; i < 4; i++) {
        const double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0) / n);
#pragma unroll
        for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * uct[3 * (i - 1) + j];
    }
    double cc[9], ci[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    if (!invert3(cc, ci)) return false;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pi = e.pws + 3 * i;
        double *a = e.alphas + 4 * i;
#pragma unroll
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    return true;
}

// epnp::compute_R_and_t: camera-frame control points from betas and the four
// singular vectors vv[0..3] (smallest first), absolute orientation, mean
// reprojection error
__device__ double epnp_R_and_t(const Epnp5 &e, const double *vv, const double *betas, double *R, double *t)
{
    constexpr int n = 5;
    double ccs[4][3], pcs[15];
#pragma unroll
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double *v = vv + 12 * i;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *a = e.alphas + 4 * i;
#pragma unroll
        for (int j = 0; j < 3; j++)
            pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[2] < 0.0) {
#pragma unroll
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
#pragma unroll
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pc = pcs + 3 * i, *pw = e.pws + 3 * i;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    polar_rotation3(abt, R);
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] -
                       R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - dot3(R, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    double sum2 = 0;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pw = e.pws + 3 * i;
        const double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        const double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
        const double u = e.us[2 * i], v = e.us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

__device__ void epnp_gauss_newton(const double *L, const double *rho, double *betas)
{
    for (int k = 0; k < 5; k++) {
        double A[24], b[6], x[4];
#pragma unroll
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
        lstsq6<4>(A, b, x);
#pragma unroll
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// epnp::compute_L_6x10 and compute_rho from the four singular vectors of
// smallest singular value vv[0..3] (smallest first, 12 doubles each)
__device__ void epnp5_L_rho(const Epnp5 &e, const double *vv, double *L, double *rho)
{
    double dv[4][6][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double *v = vv + 12 * i;
        int a = 0, b = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[3 * a] - v[3 * b];
            dv[i][j][1] = v[3 * a + 1] - v[3 * b + 1];
            dv[i][j][2] = v[3 * a + 2] - v[3 * b + 2];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
#pragma unroll
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
}

// One of the three beta approximations of epnp::compute_pose (br = 0: [B11 B12
// B13 B14], 1: [B11 B12 B22], 2: [B11 B12 B22 B13 B23]), refined by Gauss-Newton
// and turned into R, t; returns the mean reprojection error.  The three are
// independent, so three lanes of a hypothesis' group run one each.
__device__ double epnp5_branch(const Epnp5 &e, const double *vv, const double *L, const double *rho, int br,
                               double *R, double *t)
{
    double l[30], b5[5], be[4];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        l[i * 5] = L[i * 10];
        l[i * 5 + 1] = L[i * 10 + 1];
        l[i * 5 + 2] = br == 0 ? L[i * 10 + 3] : L[i * 10 + 2];
        l[i * 5 + 3] = br == 0 ? L[i * 10 + 6] : (br == 2 ? L[i * 10 + 3] : 0.0);
        l[i * 5 + 4] = br == 2 ? L[i * 10 + 4] : 0.0;
    }
    lstsq6_pad5(l, rho, b5, br == 0 ? 4 : (br == 1 ? 3 : 5));
    if (br == 0) {
        if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = -b5[1] / be[0]; be[2] = -b5[2] / be[0]; be[3] = -b5[3] / be[0]; }
        else { be[0] = sqrt(b5[0]); be[1] = b5[1] / be[0]; be[2] = b5[2] / be[0]; be[3] = b5[3] / be[0]; }
    } else {
        if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
        else { be[0] = sqrt(b5[0]); be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) be[0] = -be[0];
        be[2] = br == 2 ? b5[3] / be[0] : 0.0;
        be[3] = 0.0;
    }
    epnp_gauss_newton(L, rho, be);
    return epnp_R_and_t(e, vv, be, R, t);
}

// ---------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------

struct PrShared {
    double vv[PR_CHUNK][4][12];  // four smallest singular vectors per hypothesis
    double model[PR_CHUNK][6];   // rvec, tvec per hypothesis
    double model_R[PR_CHUNK][9];  // the rotation matrix of that rvec (what the scoring projects with)
    double best[6];
    double red[PR_WAVES][PR_NRED];
    double sums[PR_NRED];
    double param[6], prev[6];
    double null12[12];
    int ok[PR_CHUNK];
    int counts[PR_CHUNK];
    int start[PR_CHUNK];
    uint16_t r[PR_WIN];
    uint8_t cons[PR_WIN];
    uint16_t r_all[PR_WIN_ALL];   // the same for all hypotheses of a frame that is handed over
    uint8_t cons_all[PR_WIN_ALL];
    int item_start[kPnpItemsPerFrame], item_first[kPnpItemsPerFrame], item_count[kPnpItemsPerFrame], n_new_items;
    int wave_tot[PR_WAVES];
    int nchunk, per_wave, iter, niters, max_good, done;
    long long pos;
    int scan_base;
    int lm_ok;
    int cand, claim_item, claim_n, claim_hint, last;   // claiming published work items
    unsigned long long acc_mask[PR_WAVES];  // accepted hypotheses of a replay
    // planar initialisation (homography) of the refinement
    int init_mode;        // 0 DLT, 1 homography (planar inliers), 2 five non-planar inliers: keep the RANSAC pose
    int h_flag;           // block-uniform decisions of the homography polish
    double pl[12];        // rotation into the plane (row-major 3x3) and translation
    double hn[8];         // centroids and scales of the normalised DLT: cMx cMy cmx cmy sMx sMy smx smy
    double hA[64], hv[8], hD[8], hx[8], hxd[8], hd[8];
    double hS, hlambda, hlc, hrinf;
    double wmax[PR_WAVES];
};

__device__ __forceinline__ int draw_subset5(const uint16_t *r, int o, int (&idx)[5], int win = PR_WIN)
{
    int p = o;
    for (int i = 0; i < 5; i++) {
        for (;;) {
            if (p >= win) return 0;
            const int v = r[p++];
            bool dup = false;
            for (int j = 0; j < i; j++) dup = dup || (idx[j] == v);
            if (!dup) { idx[i] = v; break; }
        }
    }
    return p - o;
}

__device__ int pnp_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// squared reprojection error as cv::solvePnPRansac scores it: projection in
// double, rounded to float, pixel difference in float
__device__ __forceinline__ float pnp_error(const float *o, const float *im, const double *R, const double *t,
                                           double fx, double fy, double cx, double cy)
{
    const double X = o[0], Y = o[1], Z = o[2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
    const float dx = im[0] - px, dy = im[1] - py;
    return (float)((double)dx * dx + (double)dy * dy);
}

// block-wide deterministic sum of NV per-thread doubles -> S.sums[0..NV).
// Inside a wave the values are reduced "reduce-scatter" style: at every step a lane hands half of the values it
// still holds to its partner and adds the partner's other half, so the exchange volume halves each step (V - 1
// + log2(64 / V) shuffles for V = NV rounded up to a power of two, against 6 * NV for value-by-value trees).
// The pairing is fixed, hence the sums are bitwise reproducible.
template <int NV>
__device__ void block_reduce(PrShared &S, double (&v)[NV])
{
    constexpr int V = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;
    static_assert(NV <= 64 && NV <= PR_NRED, "at most one value per lane at the end, and room in the LDS staging");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a[V];
#pragma unroll
    for (int k = 0; k < V; k++) a[k] = k < NV ? v[k] : 0.0;
    int idx = 0;        // index of the value this lane ends up with
    int plain = 0;      // lane bits over which the final value was summed by plain exchanges
#pragma unroll
    for (int step = 0; step < 6; step++) {
        const int off = 32 >> step;
        const int cnt = (V >> step) > 0 ? (V >> step) : 1;  // values held before this step
        const bool up = (lane & off) != 0;
        if (cnt > 1) {
            const int h = cnt / 2;
#pragma unroll
            for (int i = 0; i < h; i++) {
                const double send = up ? a[i] : a[i + h];
                const double keep = up ? a[i + h] : a[i];
                a[i] = keep + __shfl_xor(send, off);
            }
            idx += up ? h : 0;
        } else {
            a[0] += __shfl_xor(a[0], off);
            plain |= off;
        }
    }
    if ((lane & plain) == 0 && idx < NV) S.red[wave][idx] = a[0];
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        double x = 0;
        for (int w = 0; w < PR_WAVES; w++) x += S.red[w][threadIdx.x];
        S.sums[threadIdx.x] = x;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------
// planar inlier sets: the homography initialisation of
// cvFindExtrinsicCameraParams2 (cv::solvePnP, SOLVEPNP_ITERATIVE)
// ---------------------------------------------------------------------

// Unit eigenvector of the smallest eigenvalue of a symmetric positive semi-definite N x N matrix (row-major, in
// LDS) by inverse iteration on its L D L^T factorisation: twelve solves from the all-ones start.  The DLT system
// of the pose refinement (N = 12) and the normalised DLT of the homography (N = 9) have one eigenvalue far below
// the others (the solution, at the noise level; exactly zero for four points), so the iteration converges in a few
// steps.  Divisions by a pivot or by the norm are multiplications by its reciprocal.
//
// A whole wave calls; lane i < N works on row i (the other lanes follow along on copies).  The factorisation and
// both substitutions run column by column: the value a step has just finished is broadcast from its lane and
// every later row takes its term at once, so a step costs one multiply-subtract instead of a row's worth.  Each
// row still subtracts its terms in the order of the one-lane loop (forward: k ascending; backward: k descending)
// -- entries outside a row's triangle are kept at zero, and subtracting a zero product leaves a number as it is.
// tr: N x N doubles of LDS for transposing L; x: N doubles (LDS) for the result.
__device__ __forceinline__ double lane_value(double v, int lane)  // lane: a constant once the loops are unrolled
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int N>
__device__ __forceinline__ void smallest_eigvec(const double *A, double *tr, double *x)
{
    const int i = threadIdx.x & 63;
    const int row = i < N ? i : N - 1;
    double a[N];
#pragma unroll
    for (int k = 0; k < N; k++) a[k] = A[row * N + k];
    // L D L^T without pivoting; a pivot that rounding has driven to zero or below is replaced by a tiny positive
    // number: for inverse iteration that only steepens the amplification of the wanted direction
    double dmax = 0;
#pragma unroll
    for (int k = 0; k < N; k++) dmax = fmax(dmax, fabs(lane_value(a[k], k)));
    const double tiny = dmax * 1e-30 + 1e-300;
    double L[N], D[N], invDi = 0;  // L[k] = L(i, k) for k < i, else 0
#pragma unroll
    for (int k = 0; k < N; k++) L[k] = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double v = a[j];  // row j: the pivot; rows below: column j of L before scaling
#pragma unroll
        for (int k = 0; k < j; k++) v -= L[k] * lane_value(L[k], j) * D[k];
        double d = lane_value(v, j);
        if (!(d > tiny)) d = tiny;
        D[j] = d;
        const double inv = 1.0 / d;
        if (i == j) invDi = inv;
        L[j] = i > j ? v * inv : 0.0;
    }
    // U[k] = L(k, i) for k > i, else 0: column i of L, through LDS
    if (i < N) {
#pragma unroll
        for (int k = 0; k < N; k++) tr[i * N + k] = L[k];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    double U[N];
#pragma unroll
    for (int k = 0; k < N; k++) U[k] = (k > i) ? tr[k * N + row] : 0.0;

    double xi = 1.0 / sqrt((double)N);
    for (int it = 0; it < 12; it++) {
        double v = xi;
#pragma unroll
        for (int k = 0; k < N - 1; k++) v -= L[k] * lane_value(v, k);  // L z = x
        v *= invDi;
#pragma unroll
        for (int k = N - 1; k > 0; k--) v -= U[k] * lane_value(v, k);  // L^T y = z
        double nrm = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double yk = lane_value(v, k);
            nrm += yk * yk;
        }
        nrm = sqrt(nrm);
        if (!(nrm > 0) || !isfinite(nrm)) break;
        const double inrm = 1.0 / nrm;
        xi = v * inrm;
    }
    if (i < N) x[i] = xi;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
}
__device__ __noinline__ void smallest_eigvec12(const double *A, double *tr, double *x) { smallest_eigvec<12>(A, tr, x); }
__device__ __noinline__ void smallest_eigvec9(const double *A, double *tr, double *x) { smallest_eigvec<9>(A, tr, x); }

// 8x8 solve, Gaussian elimination with partial pivoting (the first row of largest magnitude); false when
// singular.  One lane; every index is a compile-time constant (the row exchange is a chain of predicated
// swaps), so the matrix lives in registers, not in scratch memory.
__device__ __noinline__ bool solve_gauss8(const double *Ain, const double *bin, double *x)
{
    constexpr int n = 8;
    double A[64], b[8];
#pragma unroll
    for (int i = 0; i < 64; i++) A[i] = Ain[i];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = bin[i];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = fabs(A[k * n + k]);
#pragma unroll
        for (int i = k + 1; i < n; i++) {
            const double v = fabs(A[i * n + k]);
            if (v > best) { best = v; piv = i; }
        }
        if (best == 0) ok = false;
#pragma unroll
        for (int i = k + 1; i < n; i++) {
            const bool sw = piv == i;
#pragma unroll
            for (int j = k; j < n; j++) {
                const double u = A[k * n + j], w = A[i * n + j];
                A[k * n + j] = sw ? w : u;
                A[i * n + j] = sw ? u : w;
            }
            const double u = b[k], w = b[i];
            b[k] = sw ? w : u;
            b[i] = sw ? u : w;
        }
        const double pivot = ok ? A[k * n + k] : 1.0;
#pragma unroll
        for (int i = k + 1; i < n; i++) {
            const double f = A[i * n + k] / pivot;
#pragma unroll
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    if (!ok) return false;
    double xs[8];
#pragma unroll
    for (int k = n - 1; k >= 0; k--) {
        double sm = b[k];
#pragma unroll
        for (int j = k + 1; j < n; j++) sm -= A[k * n + j] * xs[j];
        xs[k] = sm / A[k * n + k];
    }
#pragma unroll
    for (int k = 0; k < n; k++) x[k] = xs[k];
    return true;
}

// block-wide maximum of one non-negative value per thread
__device__ double block_max(PrShared &S, double v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    if (lane == 0) S.wmax[wave] = v;
    __syncthreads();
    double m = 0;
    for (int w = 0; w < PR_WAVES; w++) m = fmax(m, S.wmax[w]);
    __syncthreads();
    return m;
}

// the inlier's point in its plane and its normalised image point, both rounded
// to float32 as cv::findHomography does with its inputs
__device__ __forceinline__ void planar_pair(const PrShared &S, const float *o, const float *im, double fx, double fy,
                                            double cx, double cy, float *M, float *m)
{
    const double s0 = o[0], s1 = o[1], s2 = o[2];
    M[0] = (float)(S.pl[0] * s0 + S.pl[1] * s1 + S.pl[2] * s2 + S.pl[9]);
    M[1] = (float)(S.pl[3] * s0 + S.pl[4] * s1 + S.pl[5] * s2 + S.pl[10]);
    m[0] = (float)(((double)im[0] - cx) / fx);
    m[1] = (float)(((double)im[1] - cy) / fy);
}

// Normal equations (upper triangle of J^T J, J^T r), squared error and largest
// residual of the 8-parameter homography x over the inliers -> S.hA, S.hv,
// S.hS, S.hrinf (HomographyRefineCallback).  with_j = false: squared error only,
// returned to every thread.
__device__ double homography_pass(PrShared &S, const float *obj, const float *img, const uint8_t *msk, int n,
                                  const double *xs, bool with_j, double fx, double fy, double cx, double cy)
{
    double h[8];
    for (int j = 0; j < 8; j++) h[j] = xs[j];
    double acc[PR_NRED];
#pragma unroll
    for (int j = 0; j < PR_NRED; j++) acc[j] = 0;
    double rmax = 0;
    for (int i = threadIdx.x; i < n; i += PR_THREADS) {
        if (!msk[i]) continue;
        float Mf[2], mf[2];
        planar_pair(S, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, Mf, mf);
        const double Mx = Mf[0], My = Mf[1];
        double ww = h[6] * Mx + h[7] * My + 1.;
        ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
        const double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        const double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        const double e0 = xi - (double)mf[0], e1 = yi - (double)mf[1];
        acc[44] += e0 * e0 + e1 * e1;
        if (with_j) {
            rmax = fmax(rmax, fmax(fabs(e0), fabs(e1)));
            const double j0[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            const double j1[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            int q = 0;
#pragma unroll
            for (int a = 0; a < 8; a++) {
#pragma unroll
                for (int bb = a; bb < 8; bb++, q++) acc[q] += j0[a] * j0[bb] + j1[a] * j1[bb];
                acc[36 + a] += j0[a] * e0 + j1[a] * e1;
            }
        }
    }
    block_reduce<PR_NRED>(S, acc);
    const double sse = S.sums[44];
    if (with_j) {
        const double rinf = block_max(S, rmax);
        if (threadIdx.x == 0) {
            int q = 0;
            for (int a = 0; a < 8; a++)
                for (int bb = a; bb < 8; bb++, q++) S.hA[a * 8 + bb] = S.hA[bb * 8 + a] = S.sums[q];
            for (int a = 0; a < 8; a++) S.hv[a] = S.sums[36 + a];
            S.hS = sse;
            S.hrinf = rinf;
        }
    }
    __syncthreads();
    return sse;
}

// Pose initialisation from a homography for a planar inlier set; S.pl holds the
// rotation into the plane and the translation.  Leaves rvec, tvec in S.param.
// Every thread of the block calls it.
__device__ __noinline__ void planar_pose_init(PrShared &S, const float *obj, const float *img, const uint8_t *msk, int n,
                                              int k, double fx, double fy, double cx, double cy)
{
    const int tid = threadIdx.x;
    // --- cv::findHomography, method 0: normalised DLT
    {
        double c4[4] = {0, 0, 0, 0};
        for (int i = tid; i < n; i += PR_THREADS) {
            if (!msk[i]) continue;
            float Mf[2], mf[2];
            planar_pair(S, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, Mf, mf);
            c4[0] += Mf[0]; c4[1] += Mf[1]; c4[2] += mf[0]; c4[3] += mf[1];
        }
        block_reduce<4>(S, c4);
        if (tid < 4) S.hn[tid] = S.sums[tid] / k;
        __syncthreads();
        double s4[4] = {0, 0, 0, 0};
        for (int i = tid; i < n; i += PR_THREADS) {
            if (!msk[i]) continue;
            float Mf[2], mf[2];
            planar_pair(S, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, Mf, mf);
            s4[0] += fabs((double)Mf[0] - S.hn[0]); s4[1] += fabs((double)Mf[1] - S.hn[1]);
            s4[2] += fabs((double)mf[0] - S.hn[2]); s4[3] += fabs((double)mf[1] - S.hn[3]);
        }
        block_reduce<4>(S, s4);
        if (tid == 0) {
            bool ok = true;
            for (int j = 0; j < 4; j++) ok = ok && !(fabs(S.sums[j]) < DBL_EPSILON);
            for (int j = 0; j < 4; j++) S.hn[4 + j] = ok ? k / S.sums[j] : 0.0;
            S.h_flag = ok ? 1 : 0;
        }
        __syncthreads();
    }
    bool okH = S.h_flag != 0;
    if (okH) {
        double acc[PR_NRED];
#pragma unroll
        for (int j = 0; j < PR_NRED; j++) acc[j] = 0;
        const double cMx = S.hn[0], cMy = S.hn[1], cmx = S.hn[2], cmy = S.hn[3];
        const double sMx = S.hn[4], sMy = S.hn[5], smx = S.hn[6], smy = S.hn[7];
        for (int i = tid; i < n; i += PR_THREADS) {
            if (!msk[i]) continue;
            float Mf[2], mf[2];
            planar_pair(S, obj + 3 * i, img + 2 * i, fx, fy, cx, cy, Mf, mf);
            const double x = ((double)mf[0] - cmx) * smx, y = ((double)mf[1] - cmy) * smy;
            const double X = ((double)Mf[0] - cMx) * sMx, Y = ((double)Mf[1] - cMy) * sMy;
            const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
            const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
            int q = 0;
#pragma unroll
            for (int a = 0; a < 9; a++)
#pragma unroll
                for (int bb = a; bb < 9; bb++, q++) acc[q] += Lx[a] * Lx[bb] + Ly[a] * Ly[bb];
        }
        block_reduce<PR_NRED>(S, acc);
        if (tid < kWave) {  // (the hypotheses' singular vectors are not needed any more: S.vv is workspace)
            double *LtL = &S.vv[0][0][0];
            if (tid < 9)
                for (int bb = 0; bb < 9; bb++) {
                    const int lo = tid < bb ? tid : bb, hi = tid < bb ? bb : tid;
                    LtL[tid * 9 + bb] = S.sums[lo * 9 - (lo * (lo - 1)) / 2 + (hi - lo)];
                }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            smallest_eigvec9(LtL, LtL + 81, S.null12);
        }
        if (tid == 0) {
            double H0[9], T[9], H[9];
            for (int j = 0; j < 9; j++) H0[j] = S.null12[j];
            const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
            const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    T[i * 3 + j] = invHnorm[i * 3] * H0[j] + invHnorm[i * 3 + 1] * H0[3 + j] + invHnorm[i * 3 + 2] * H0[6 + j];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    H[i * 3 + j] = T[i * 3] * Hnorm2[j] + T[i * 3 + 1] * Hnorm2[3 + j] + T[i * 3 + 2] * Hnorm2[6 + j];
            const double sc = 1. / H[8];
            for (int j = 0; j < 8; j++) S.hx[j] = H[j] * sc;
        }
        __syncthreads();
        if (k > 4) {
            // --- the Levenberg-Marquardt polish of LMSolver: 10 iterations, eps FLT_EPSILON
            homography_pass(S, obj, img, msk, n, S.hx, true, fx, fy, cx, cy);
            if (tid == 0) {
                for (int j = 0; j < 8; j++) S.hD[j] = S.hA[j * 8 + j];
                S.hlambda = 1;
                S.hlc = 0.75;
            }
            __syncthreads();
            for (int iter = 0;;) {
                if (tid == 0) {
                    double Ap[64], d[8];
                    for (int j = 0; j < 64; j++) Ap[j] = S.hA[j];
                    for (int j = 0; j < 8; j++) Ap[j * 8 + j] += S.hlambda * S.hD[j];
                    if (!solve_gauss8(Ap, S.hv, d)) {
                        for (int j = 0; j < 8; j++) d[j] = 0;
                    }
                    for (int j = 0; j < 8; j++) { S.hd[j] = d[j]; S.hxd[j] = S.hx[j] - d[j]; }
                }
                __syncthreads();
                const double Sd = homography_pass(S, obj, img, msk, n, S.hxd, false, fx, fy, cx, cy);
                if (tid == 0) {
                    const double Sc = S.hS;
                    double dS = 0, t = 0;
                    for (int i = 0; i < 8; i++) {
                        double td = 2 * S.hv[i];
                        for (int j = 0; j < 8; j++) td -= S.hA[i * 8 + j] * S.hd[j];
                        dS += S.hd[i] * td;
                        t += S.hd[i] * S.hv[i];
                    }
                    const double Rr = (Sc - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
                    if (Rr > 0.75) {
                        S.hlambda *= 0.5;
                        if (S.hlambda < S.hlc) S.hlambda = 0;
                    } else if (Rr < 0.25) {
                        double nu = (Sd - Sc) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
                        nu = nu < 2. ? 2. : (nu > 10. ? 10. : nu);
                        if (S.hlambda == 0) {
                            // 1 / largest diagonal entry of A^-1
                            double maxval = DBL_EPSILON;
                            for (int c = 0; c < 8; c++) {
                                double e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, col[8];
                                e[c] = 1;
                                if (solve_gauss8(S.hA, e, col) && fabs(col[c]) > maxval) maxval = fabs(col[c]);
                            }
                            S.hlambda = S.hlc = 1. / maxval;
                            nu *= 0.5;
                        }
                        S.hlambda *= nu;
                    }
                    S.h_flag = Sd < Sc ? 1 : 0;
                    if (Sd < Sc)
                        for (int j = 0; j < 8; j++) S.hx[j] = S.hxd[j];
                }
                __syncthreads();
                if (S.h_flag) homography_pass(S, obj, img, msk, n, S.hx, true, fx, fy, cx, cy);
                double dinf = 0;
                for (int j = 0; j < 8; j++) dinf = fmax(dinf, fabs(S.hd[j]));
                iter++;
                const bool proceed = iter < 10 && dinf >= (double)FLT_EPSILON && S.hrinf >= (double)FLT_EPSILON;
                __syncthreads();
                if (!proceed) break;
            }
        }
    }
    if (tid == 0) {
        double h[9];
        for (int j = 0; j < 8; j++) h[j] = S.hx[j];
        h[8] = 1;
        for (int j = 0; j < 8 && okH; j++) okH = isfinite(h[j]);
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
        if (okH) {
            const double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
            const double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / (h1n > DBL_EPSILON ? h1n : DBL_EPSILON), s2 = 1. / (h2n > DBL_EPSILON ? h2n : DBL_EPSILON);
            const double s3 = 2. / (h1n + h2n > DBL_EPSILON ? h1n + h2n : DBL_EPSILON);
            const double a[3] = {h[0] * s1, h[3] * s1, h[6] * s1}, b[3] = {h[1] * s2, h[4] * s2, h[7] * s2};
            const double tt[3] = {h[2] * s3, h[5] * s3, h[8] * s3};
            const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const double Hm[9] = {a[0], b[0], c[0], a[1], b[1], c[1], a[2], b[2], c[2]};
            double Hr[9], rv[3];
            // cv::Rodrigues there and back: the nearest rotation
            polar_rotation3(Hm, Hr);
            rodrigues_m2v(Hr, rv);
            rodrigues_v2m(rv, Hr, nullptr);
            for (int i = 0; i < 3; i++)
                t[i] = Hr[3 * i] * S.pl[9] + Hr[3 * i + 1] * S.pl[10] + Hr[3 * i + 2] * S.pl[11] + tt[i];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    R[i * 3 + j] = Hr[i * 3] * S.pl[j] + Hr[i * 3 + 1] * S.pl[3 + j] + Hr[i * 3 + 2] * S.pl[6 + j];
        }
        double rv[3];
        rodrigues_m2v(R, rv);
        for (int j = 0; j < 3; j++) { S.param[j] = rv[j]; S.param[3 + j] = t[j]; }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------
// four points: cv::solvePnP(SOLVEPNP_P3P)
// ---------------------------------------------------------------------

__device__ __forceinline__ double p3p_horner(const double *p, int deg, double x)
{
    double v = p[deg];
    for (int i = deg - 1; i >= 0; i--) v = v * x + p[i];
    return v;
}

// ascending real roots (with sign change) of a polynomial of degree <= 4: the
// roots of each derivative split [-B, B] (Cauchy bound) into monotone pieces
// that are bisected until the interval stops shrinking
__device__ int p3p_real_roots4(const double *p_in, double *roots)
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
                bool exact = false;
                for (int it = 0; it < 200; it++) {
                    const double mid = 0.5 * (a + b);
                    if (mid == a || mid == b) break;
                    const double fm = p3p_horner(p, d, mid);
                    if (fm == 0) { a = b = mid; exact = true; break; }
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

// orthonormal frame of a triangle: e1 along p1 - p0, e3 the normal, e2 = e3 x e1 (rows of F)
__device__ bool p3p_triangle_frame(const double *p0, const double *p1, const double *p2, double *F)
{
    double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    const double b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    const double na = sqrt(dot3(a, a));
    if (na == 0) return false;
    for (int i = 0; i < 3; i++) a[i] /= na;
    double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const double nc = sqrt(dot3(c, c));
    if (nc == 0) return false;
    for (int i = 0; i < 3; i++) c[i] /= nc;
    const double e2[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
    for (int i = 0; i < 3; i++) { F[i] = a[i]; F[3 + i] = e2[i]; F[6 + i] = c[i]; }
    return true;
}

// The (up to four) poses consistent with the first three correspondences, from
// Grunert's quartic in v = s3 / s1 (si: distance of point i from the camera);
// the one that reprojects the fourth point best is returned (OpenCV's p3p.cpp
// reaches the same solution set along Gao's derivation).  One lane.
__device__ __noinline__ bool p3p_solve4(const float *objf, const float *imgf, double fx, double fy, double cx, double cy,
                                        double *R, double *t)
{
    double obj[12], img[8];
    for (int i = 0; i < 12; i++) obj[i] = objf[i];
    for (int i = 0; i < 8; i++) img[i] = imgf[i];
    double j[3][3];
    for (int i = 0; i < 3; i++) {
        const double x = (img[2 * i] - cx) / fx, y = (img[2 * i + 1] - cy) / fy;
        const double nrm = sqrt(x * x + y * y + 1.0);
        j[i][0] = x / nrm; j[i][1] = y / nrm; j[i][2] = 1.0 / nrm;
    }
    const double a2 = dist2(obj + 3, obj + 6), b2 = dist2(obj, obj + 6), c2 = dist2(obj, obj + 3);
    if (a2 == 0 || b2 == 0 || c2 == 0) return false;
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
    if (!p3p_triangle_frame(obj, obj + 3, obj + 6, FX)) return false;
    const double Xc[3] = {(obj[0] + obj[3] + obj[6]) / 3, (obj[1] + obj[4] + obj[7]) / 3, (obj[2] + obj[5] + obj[8]) / 3};
    bool found = false;
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
        const double *X3 = obj + 9;
        const double Xp = dot3(Rc, X3) + tc[0], Yp = dot3(Rc + 3, X3) + tc[1], Zp = dot3(Rc + 6, X3) + tc[2];
        const double du = cx + fx * Xp / Zp - img[6], dv = cy + fy * Yp / Zp - img[7];
        const double e = du * du + dv * dv;
        if (!isfinite(e)) continue;
        if (!found || e < best) {
            found = true;
            best = e;
            for (int i = 0; i < 9; i++) R[i] = Rc[i];
            for (int i = 0; i < 3; i++) t[i] = tc[i];
        }
    }
    return found;
}

#ifdef VO_FE_STAMPS
// Diagnostic build only: cycle stamps of thread 0 of the first 256 blocks.
__device__ unsigned long long g_pnp_stamps[256 * 16];
#define PNP_STAMP(i)                                                                     \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 256)                                        \
            g_pnp_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();          \
    } while (0)
extern "C" int vo_debug_pnp_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pnp_stamps), sizeof(unsigned long long) * 256 * 16);
}
// per block of the first 1024: start, own frame done, block done (cycles); handed over, items solved, items skipped,
// frames resumed
__device__ unsigned long long g_pnp_blocks[1024 * 8];
#define PNP_BLOCK(i, v)                                                                  \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_pnp_blocks[blockIdx.x * 8 + (i)] = (v); \
    } while (0)
#define PNP_BLOCK_ADD(i, v)                                                              \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_pnp_blocks[blockIdx.x * 8 + (i)] += (v); \
    } while (0)
extern "C" int vo_debug_pnp_blocks(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pnp_blocks), sizeof(unsigned long long) * 1024 * 8);
}
// Levenberg-Marquardt loop of the first 256 blocks: iterations, then cycles spent in the normal equations' sums,
// their reduction, the 6 x 6 solves, the error evaluations; inner (lambda) steps
__device__ unsigned long long g_pnp_lm[256 * 8];
#define PNP_LM_T0() unsigned long long lm_t0 = __builtin_amdgcn_s_memtime()
#define PNP_LM_LAP(i)                                                                    \
    do {                                                                                 \
        const unsigned long long lm_t1 = __builtin_amdgcn_s_memtime();                   \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_pnp_lm[blockIdx.x * 8 + (i)] += lm_t1 - lm_t0; \
        lm_t0 = lm_t1;                                                                   \
    } while (0)
#define PNP_LM_SET(i, v)                                                                 \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_pnp_lm[blockIdx.x * 8 + (i)] = (v); \
    } while (0)
#define PNP_LM_COUNT(i)                                                                  \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_pnp_lm[blockIdx.x * 8 + (i)] += 1;  \
    } while (0)
extern "C" int vo_debug_pnp_lm(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pnp_lm), sizeof(unsigned long long) * 256 * 8);
}
#else
#define PNP_LM_T0() do { } while (0)
#define PNP_LM_LAP(i) do { } while (0)
#define PNP_LM_SET(i, v) do { } while (0)
#define PNP_LM_COUNT(i) do { } while (0)
#define PNP_STAMP(i) do { } while (0)
#define PNP_BLOCK(i, v) do { } while (0)
#define PNP_BLOCK_ADD(i, v) do { } while (0)
#endif

// solve hypotheses first_h .. first_h + per_wave - 1 of the round (those below nc), one per group of PR_GROUP
// lanes; all lanes of the wave call.  A function of its own on purpose: it needs nearly every register, and inlined
// into the kernel its loops pick up spills.  (The kernel carries disable_tail_calls for its sake: see there.)
static __device__ __noinline__ void solve_hypothesis(PrShared &S, const float *obj, const float *img, int first_h,
                                                     int per_wave, int nc, double fx, double fy, double cx, double cy)
{
    const int lane = threadIdx.x & 63, grp = lane / PR_GROUP, gl = lane - grp * PR_GROUP;
    const bool member = grp < per_wave;  // the other lanes of the wave follow along idle
    const int h = member ? first_h + grp : first_h;
    const bool active = member && h < nc;
    Epnp5 e;
    e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
    int idx[5] = {0, 1, 2, 3, 4};
    if (active) draw_subset5(S.r, S.start[h], idx);
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) e.pws[3 * i + j] = obj[3 * idx[i] + j];
        e.us[2 * i] = img[2 * idx[i]];
        e.us[2 * i + 1] = img[2 * idx[i] + 1];
    }
    PNP_STAMP(11);
    bool ok = epnp5_prepare(e);
    PNP_STAMP(12);

    // column gl of the 10 x 12 matrix M (epnp::fill_M): gl = 3k + c
    double g[10], v[12];
    {
        const int k = gl / 3, c = gl - 3 * k;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const double a = k == 0 ? e.alphas[4 * i] : (k == 1 ? e.alphas[4 * i + 1] : (k == 2 ? e.alphas[4 * i + 2] : e.alphas[4 * i + 3]));
            g[2 * i] = c == 0 ? a * e.fu : (c == 2 ? a * (e.uc - e.us[2 * i]) : 0.0);
            g[2 * i + 1] = c == 1 ? a * e.fv : (c == 2 ? a * (e.vc - e.us[2 * i + 1]) : 0.0);
        }
        if (!active || !ok) {  // (a zero matrix: the decomposition has nothing to rotate)
#pragma unroll
            for (int i = 0; i < 10; i++) g[i] = 0;
        }
    }
    double n2;
    int rank;
    group_svd12<10>(g, v, gl, &n2, &rank);
    PNP_STAMP(13);
    // ranks 11, 10, 9, 8 are the four smallest singular values: vv[0..3]
    if (active && rank >= 8) {
#pragma unroll
        for (int i = 0; i < 12; i++) S.vv[h][11 - rank][i] = v[i];
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // the three beta approximations on lanes 0..2 of the group
    double R[9], t[3];
    double rep = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = 0;
    t[0] = t[1] = t[2] = 0;
    const bool run = active && ok;
    if (gl < 3 && run) {
        double vv[48], L[60], rho[6];
#pragma unroll
        for (int i = 0; i < 48; i++) vv[i] = S.vv[h][i / 12][i % 12];
        epnp5_L_rho(e, vv, L, rho);
        rep = epnp5_branch(e, vv, L, rho, gl, R, t);
    }
    PNP_STAMP(14);
    // epnp::compute_pose keeps approximation 1 unless 2 reprojects better, then 3 unless it beats that
    const int base = grp * PR_GROUP;
    const double rep1 = __shfl(rep, base), rep2 = __shfl(rep, base + 1), rep3 = __shfl(rep, base + 2);
    int win = 0;
    double best_rep = rep1;
    if (rep2 < best_rep) { win = 1; best_rep = rep2; }
    if (rep3 < best_rep) win = 2;
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = __shfl(R[i], base + win);
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = __shfl(t[i], base + win);
    if (gl == 0 && active) {
        bool good = run;
#pragma unroll
        for (int i = 0; i < 9; i++) good = good && isfinite(R[i]);
#pragma unroll
        for (int i = 0; i < 3; i++) good = good && isfinite(t[i]);
        if (good) {
            double rv[3];
            rodrigues_m2v(R, rv);
#pragma unroll
            for (int i = 0; i < 3; i++) { S.model[h][i] = rv[i]; S.model[h][3 + i] = t[i]; }
            // the scoring projects with the matrix of the rotation vector, not with R itself: formed here, where the
            // hypotheses of a wave do it side by side, rather than one after the other in the scoring loop
            double Rv[9];
            rodrigues_v2m(rv, Rv, nullptr);
#pragma unroll
            for (int i = 0; i < 9; i++) S.model_R[h][i] = Rv[i];
        }
        S.ok[h] = good ? 1 : 0;
    }
}

// ---------------------------------------------------------------------
// Frames that need many hypotheses are spread over the chip.
//
// One block per frame runs the first round of hypotheses (most frames need no more) and, when at most one more
// round is left, that one too.  A frame that still has more than PR_CHUNK iterations to go would keep its block --
// and with it the whole launch -- busy for up to ten more sequential rounds on one compute unit.  Such a frame is
// handed over instead: its block chains the RNG stream into the start positions of every hypothesis the current
// iteration bound still admits (the bound only ever shrinks), publishes them as work items of up to PR_CHUNK
// hypotheses, and looks for work itself.  Every block whose own frame is finished claims published items (of any
// frame), solves and scores their hypotheses and stores (model, inlier count) records.  Whenever an item is
// completed, its block looks at the items before it: once a gap-free run of items from the first one on is complete,
// the sequential acceptance rule can be replayed over their records in RNG order; if the replay shows that the
// sequential loop stops inside them (or every item is complete), that block takes the frame back -- exactly one
// does -- stops where the
// sequential algorithm stops, and refines the pose.  Blocks never wait for each other: a block that finds nothing
// to claim exits.  Nothing is left behind that way: a block looks for work after publishing its own items, so it
// cannot leave while one of them is unclaimed, and whoever claims an item finishes it.  Small batches (the
// single-stream path: one block per launch) have nobody to share with; there the frames' blocks only publish, and
// a second launch of the same kernel with many blocks and no frames of its own takes the items side by side.
// The hypotheses, their order and the acceptance logic are those of the
// one-block loop, so the results are the same bit for bit; only hypotheses past the final iteration bound are
// computed in vain, and few of those: every completed item lowers a per-frame bound from its best inlier count
// (the sequential loop's bound can only be lower), and items that start at or behind it, or belong to a frame
// that has been taken back, are dropped unsolved when they are claimed.
// ---------------------------------------------------------------------

// hypothesis records of a handed-over frame: kPnpSpecMax per frame
struct PnpHyp {
    double model[6];
    int count;
    int ok;  // 1 usable, 0 the solver failed
};
// what a handed-over frame needs to go on
struct PnpResume {
    double best[6];
    long long pos;  // RNG position after the last hypothesis of the work items
    int iter, niters, max_good, n_spec;
};
struct PnpItem {
    long long pos;  // RNG position of the item's first hypothesis
    int first, count;
};

struct PnpArgs {
    const vo_match *valid;
    const float *pts3;
    const int *bp_info;
    int kp_stride;
    const uint32_t *rng_raw;
    int rng_len;
    double fx, fy, cx, cy;
    int max_iters;
    double threshold, confidence;
    int *inliers, *pose;
    double *rt;
    int batch;
    int steal;  // blocks of the first launch claim published work once their own frame is done
    // hand-over (all null: every frame finishes in its own block).  Per frame: state = number of published items
    // (0 until the frame's items, resume record and everything else are in memory), next = items claimed so far
    // (may overshoot), done = bit i set: item i is complete.  Zero before the first launch.
    int *state, *next, *done;
    // per frame, set by the publisher (no clearing needed): hint = an upper bound of the iteration count at which the
    // sequential loop stops, lowered by every finished item; resumed = a block has taken the frame back
    int *hint, *resumed;
    PnpItem *items;  // kPnpItemsPerFrame per frame
    PnpResume *resume;
    PnpHyp *hyp;
};

__device__ __forceinline__ void pnp_lds_views(unsigned char *s_dyn, int kp_stride, float *&obj, float *&img,
                                              uint8_t *&msk)
{
    unsigned char *dyn = s_dyn + ((sizeof(PrShared) + 15) & ~size_t(15));
    obj = reinterpret_cast<float *>(dyn);                              // n x 3
    img = obj + 3 * (size_t)kp_stride;                                 // n x 2
    msk = reinterpret_cast<uint8_t *>(img + 2 * (size_t)kp_stride);    // n
}

__device__ __forceinline__ void pnp_load_points(const vo_match *vm, const float *p3, int n, float *obj, float *img,
                                                uint8_t *msk)
{
    for (int i = threadIdx.x; i < n; i += PR_THREADS) {
        obj[3 * i] = p3[3 * i]; obj[3 * i + 1] = p3[3 * i + 1]; obj[3 * i + 2] = p3[3 * i + 2];
        img[2 * i] = vm[i].px; img[2 * i + 1] = vm[i].py;
        msk[i] = 0;
    }
}

// One window of the RNG stream from S.pos on: every offset simulated as the start of a subset draw, then one lane
// chains offsets into up to `want` hypotheses (S.start, S.nchunk) and moves S.pos behind the last of them.
// Every thread of the block calls it; ends with a barrier.
__device__ __forceinline__ void pnp_chain_round(PrShared &S, int n, const uint32_t *__restrict__ rng_raw, int rng_len, int want)
{
    const int tid = threadIdx.x;
    const long long pos = S.pos;
    for (int k = tid; k < PR_WIN; k += PR_THREADS) {
        const long long a = pos + k;
        S.r[k] = a < rng_len ? (uint16_t)(rng_raw[a] % (uint32_t)n) : (uint16_t)0;
    }
    __syncthreads();
    for (int o = tid; o < PR_WIN; o += PR_THREADS) {
        int idx[5];
        int cons = draw_subset5(S.r, o, idx);
        if (pos + o + cons > rng_len) cons = 0;
        S.cons[o] = (uint8_t)cons;
    }
    __syncthreads();
    if (tid == 0) {
        int w = 0, nc = 0;
        while (nc < want && w < PR_WIN && S.cons[w] != 0) { S.start[nc++] = w; w += S.cons[w]; }
        S.nchunk = nc;
        S.pos = pos + w;
    }
    __syncthreads();
}

// solve the S.nchunk hypotheses of the round (per_wave of them per wave) and count their inliers; ends with a barrier
__device__ __forceinline__ void pnp_solve_score(PrShared &S, const float *obj, const float *img, int n, int per_wave, double fx,
                                double fy, double cx, double cy, float thr2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = S.nchunk;
    // a wave takes part when any of its hypotheses exists
    if (wave * per_wave < nc) solve_hypothesis(S, obj, img, wave * per_wave, per_wave, nc, fx, fy, cx, cy);
    __syncthreads();
    PNP_STAMP(3);
    // a wave scores hypotheses wave, wave + PR_WAVES, ...: four at a time against each point (the projection of a
    // point is one long dependent chain, its division above all; four of them side by side fill each other's waits)
    for (int h0 = wave; h0 < nc; h0 += 4 * PR_WAVES) {
        double R[4][9], t[4][3];
        int cnt[4];
        bool use[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int h = h0 + j * PR_WAVES;
            use[j] = h < nc && S.ok[h];
            const int hh = h < nc ? h : h0;
#pragma unroll
            for (int i = 0; i < 9; i++) R[j][i] = S.model_R[hh][i];
#pragma unroll
            for (int i = 0; i < 3; i++) t[j][i] = S.model[hh][3 + i];
            cnt[j] = 0;
        }
        if (use[1] || use[2] || use[3]) {
            for (int i = lane; i < n; i += kWave) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    cnt[j] += (pnp_error(obj + 3 * i, img + 2 * i, R[j], t[j], fx, fy, cx, cy) <= thr2);
            }
        } else if (use[0]) {
            for (int i = lane; i < n; i += kWave)
                cnt[0] += (pnp_error(obj + 3 * i, img + 2 * i, R[0], t[0], fx, fy, cx, cy) <= thr2);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int c = cnt[j];
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
            if (lane == 0 && use[j]) S.counts[h0 + j * PR_WAVES] = c;
        }
    }
    __syncthreads();
}

// the sequential acceptance rule of cv::RANSACPointSetRegistrator::run for one hypothesis (one lane)
__device__ __forceinline__ bool pnp_accept(PrShared &S, int ok, int goodc, int n, double confidence)
{
    const bool take = ok && goodc > max(S.max_good, 4);  // the caller copies the model into S.best
    if (take) {
        S.max_good = goodc;
        S.niters = pnp_update_num_iters(confidence, (double)(n - goodc) / n, 5, S.niters);
    }
    S.iter++;
    return take;
}

// RANSAC rounds from the state in S (iter, niters, max_good, best, pos) until the iteration bound is reached
// (S.done = 1) or, with `handover`, until a frame with more than one round to go should leave its block (S.done = 2)
__device__ __forceinline__ void pnp_ransac_rounds(PrShared &S, const float *obj, const float *img, int n, const PnpArgs &a, float thr2,
                                  bool handover)
{
    const int tid = threadIdx.x;
    while (true) {
        // the first round is the only one of most frames: four hypotheses per wave (a wave's decomposition
        // takes as many sweeps as its slowest group needs); later rounds fill all five groups of a wave
        const int pw = S.iter == 0 ? PR_FIRST / PR_WAVES : PR_PER_WAVE;
        const int want = min(pw * PR_WAVES, S.niters - S.iter);
        pnp_chain_round(S, n, a.rng_raw, a.rng_len, want);
        PNP_STAMP(2);
        const int nc = S.nchunk;
        pnp_solve_score(S, obj, img, n, pw, a.fx, a.fy, a.cx, a.cy, thr2);
        if (tid == 0) {
            if (nc == 0) S.done = 1;  // RNG table exhausted
            for (int h = 0; h < nc; h++) {
                if (pnp_accept(S, S.ok[h], S.counts[h], n, a.confidence))
                    for (int i = 0; i < 6; i++) S.best[i] = S.model[h][i];
                if (S.iter >= S.niters) break;
            }
            if (S.iter >= S.niters) S.done = 1;
            else if (handover && !S.done && S.niters - S.iter > PR_CHUNK && S.niters - S.iter <= kPnpSpecMax)
                S.done = 2;
        }
        __syncthreads();
        PNP_STAMP(4);
        if (S.done) break;
    }
}

// after the RANSAC rounds: inlier list of the best hypothesis, solvePnP(SOLVEPNP_ITERATIVE) on the inliers, outputs
__device__ __forceinline__ void pnp_finish(PrShared &S, const float *obj, const float *img, uint8_t *msk, int n, double fx, double fy,
                           double cx, double cy, float thr2, int *inl, int *po, double *rto)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool found = S.max_good > 0;
    if (!found) {
        if (tid == 0) { po[0] = 0; po[1] = 0; po[2] = S.iter; po[3] = n; }
        return;
    }

    // inlier mask of the best hypothesis, ascending index list
    {
        double R[9];
        rodrigues_v2m(S.best, R, nullptr);
        const double t[3] = {S.best[3], S.best[4], S.best[5]};
        if (tid == 0) S.scan_base = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += PR_THREADS) {
            const int i = i0 + tid;
            bool keep = false;
            if (i < n) {
                keep = pnp_error(obj + 3 * i, img + 2 * i, R, t, fx, fy, cx, cy) <= thr2;
                msk[i] = (uint8_t)keep;
            }
            const uint64_t bm = __ballot(keep);
            if (lane == 0) S.wave_tot[wave] = __popcll(bm);
            __syncthreads();
            int off = S.scan_base;
            for (int w = 0; w < wave; w++) off += S.wave_tot[w];
            if (keep) inl[off + __popcll(bm & ((1ull << lane) - 1ull))] = i;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < PR_WAVES; w++) tot += S.wave_tot[w];
                S.scan_base += tot;
            }
            __syncthreads();
        }
    }
    const int k = S.scan_base;  // inlier count
    PNP_STAMP(5);

    // ---------------- refinement: solvePnP(SOLVEPNP_ITERATIVE) on inliers ----
    bool init_ok = true;
    {
        // planarity test of the object points (cvFindExtrinsicCameraParams2)
        double c3[3] = {0, 0, 0};
        for (int i = tid; i < n; i += PR_THREADS)
            if (msk[i]) { c3[0] += obj[3 * i]; c3[1] += obj[3 * i + 1]; c3[2] += obj[3 * i + 2]; }
        block_reduce<3>(S, c3);
        const double Mc[3] = {S.sums[0] / k, S.sums[1] / k, S.sums[2] / k};
        __syncthreads();
        double mm[9];
#pragma unroll
        for (int j = 0; j < 9; j++) mm[j] = 0;
        for (int i = tid; i < n; i += PR_THREADS)
            if (msk[i]) {
                const double d[3] = {obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2]};
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int bb = 0; bb < 3; bb++) mm[a * 3 + bb] += d[a] * d[bb];
            }
        block_reduce<9>(S, mm);
        if (tid == 0) {
            double MM[9], W[3], Vt[9];
            for (int j = 0; j < 9; j++) MM[j] = S.sums[j];
            eig3_desc(MM, W, Vt);
            S.lm_ok = 1;
            if (W[2] / W[1] < 1e-3) {
                // a planar structure: rotate the points into their plane (rows of Vt = principal axes)
                S.init_mode = 1;
                if (Vt[2] * Vt[2] + Vt[5] * Vt[5] < 1e-10) {
                    for (int j = 0; j < 9; j++) Vt[j] = (j % 4 == 0) ? 1.0 : 0.0;
                }
                const double det = Vt[0] * (Vt[4] * Vt[8] - Vt[5] * Vt[7]) - Vt[1] * (Vt[3] * Vt[8] - Vt[5] * Vt[6]) +
                                   Vt[2] * (Vt[3] * Vt[7] - Vt[4] * Vt[6]);
                for (int j = 0; j < 9; j++) S.pl[j] = det < 0 ? -Vt[j] : Vt[j];
                for (int j = 0; j < 3; j++)
                    S.pl[9 + j] = -(S.pl[3 * j] * Mc[0] + S.pl[3 * j + 1] * Mc[1] + S.pl[3 * j + 2] * Mc[2]);
            } else {
                // the DLT needs six points; with five OpenCV (>= 4.5.2) catches the exception of its own solvePnP
                // call and keeps the pose of the minimal sample
                S.init_mode = k >= 6 ? 0 : 2;
            }
        }
        __syncthreads();
    }
    const int init_mode = S.init_mode;
    PNP_STAMP(6);
    if (init_mode == 1) planar_pose_init(S, obj, img, msk, n, k, fx, fy, cx, cy);
    if (init_mode == 0) {
        // DLT: L^T L has the block form [[P, 0, Px], [0, P, Py], [Px, Py, Pxy]]
        // with P = sum m m^T, m = (X, Y, Z, 1); x, y = minus the normalised image point
        double acc[40];
#pragma unroll
        for (int j = 0; j < 40; j++) acc[j] = 0;
        for (int i = tid; i < n; i += PR_THREADS) {
            if (!msk[i]) continue;
            const double m[4] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], 1.0};
            const double x = -((double)img[2 * i] - cx) / fx, y = -((double)img[2 * i + 1] - cy) / fy;
            int q = 0;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int bb = a; bb < 4; bb++, q++) {
                    const double mab = m[a] * m[bb];
                    acc[q] += mab;
                    acc[10 + q] += (x * m[a]) * m[bb];
                    acc[20 + q] += (y * m[a]) * m[bb];
                    acc[30 + q] += (x * m[a]) * (x * m[bb]) + (y * m[a]) * (y * m[bb]);
                }
        }
        block_reduce<40>(S, acc);
        PNP_STAMP(7);
        if (tid < kWave) {
            // the symmetric 12 x 12 matrix from its blocks, then the eigenvector of its smallest eigenvalue
            // (the hypotheses' singular vectors are not needed any more: S.vv is workspace)
            double *LL = &S.vv[0][0][0];
            if (tid < 12) {
                const int row = tid;
                for (int cc = 0; cc < 12; cc++) {
                    const int br = row >> 2, bc = cc >> 2, a = row & 3, c2 = cc & 3;
                    const int lo = a < c2 ? a : c2, hi = a < c2 ? c2 : a;
                    const int q = lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo);  // index in the upper triangle of a 4x4
                    double val = 0;
                    if (br == bc) val = (br == 2) ? S.sums[30 + q] : S.sums[q];
                    else if (br + bc == 2) val = S.sums[10 + q];  // blocks (0,2), (2,0)
                    else if (br + bc == 3) val = S.sums[20 + q];  // blocks (1,2), (2,1)
                    LL[row * 12 + cc] = val;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            smallest_eigvec12(LL, LL + 144, S.null12);
        }
        __syncthreads();
        PNP_STAMP(8);
        if (tid == 0) {
            const double *RRt = S.null12;
            double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
            double tt[3] = {RRt[3], RRt[7], RRt[11]};
            const double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) +
                               RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
            if (det < 0) {
                for (int j = 0; j < 9; j++) RR[j] = -RR[j];
                for (int j = 0; j < 3; j++) tt[j] = -tt[j];
            }
            double sc = 0;
            for (int j = 0; j < 9; j++) sc += RR[j] * RR[j];
            sc = sqrt(sc);
            if (fabs(sc) <= DBL_EPSILON) {
                S.lm_ok = 0;
            } else {
                double R[9], nR = 0, rv[3];
                polar_rotation3(RR, R);
                for (int j = 0; j < 9; j++) nR += R[j] * R[j];
                nR = sqrt(nR);
                rodrigues_m2v(R, rv);
                for (int j = 0; j < 3; j++) { S.param[j] = rv[j]; S.param[3 + j] = tt[j] * nR / sc; }
            }
        }
        __syncthreads();
        init_ok = S.lm_ok != 0;
    }

    PNP_STAMP(9);
    if (init_ok && init_mode != 2) {
        // CvLevMarq: <= 20 iterations, epsilon FLT_EPSILON, lambda = 10^lambdaLg10
        const int max_iter = 20;
        const double eps = FLT_EPSILON;
        const double LOG10 = log(10.);
        double prevErrNorm = DBL_MAX, errNorm = 0;
        int lambdaLg10 = -3, iters = 0;
        for (int j = 0; j < 8; j++) PNP_LM_SET(j, 0);
        PNP_LM_T0();
        for (;;) {
            // J^T J (upper triangle), J^T e, |e|^2 at S.param
            constexpr int NLM = 21 + 6 + 1;
            double acc[NLM];
#pragma unroll
            for (int j = 0; j < NLM; j++) acc[j] = 0;
            {
                double R[9], dRdr[27], p[6];
                for (int j = 0; j < 6; j++) p[j] = S.param[j];
                rodrigues_v2m(p, R, dRdr);
                for (int i = tid; i < n; i += PR_THREADS) {
                    // (the refinement is compared with its CPU counterpart to a tolerance, not bit by bit: the block's
                    // sums are paired differently anyway.  Its multiply-adds may fuse.)
#pragma clang fp contract(fast)
                    if (!msk[i]) continue;
                    const double M[3] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]};
                    const double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + p[3];
                    const double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + p[4];
                    const double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + p[5];
                    const double z = Z ? 1. / Z : 1;
                    const double x = X * z, y = Y * z;
                    const double e0 = x * fx + cx - (double)img[2 * i], e1 = y * fy + cy - (double)img[2 * i + 1];
                    double j0[6], j1[6];
                    j0[3] = fx * z; j0[4] = 0; j0[5] = -fx * x * z;
                    j1[3] = 0; j1[4] = fy * z; j1[5] = -fy * y * z;
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const double *d = dRdr + 9 * q;
                        const double dX = d[0] * M[0] + d[1] * M[1] + d[2] * M[2];
                        const double dY = d[3] * M[0] + d[4] * M[1] + d[5] * M[2];
                        const double dZ = d[6] * M[0] + d[7] * M[1] + d[8] * M[2];
                        j0[q] = fx * (dX - x * dZ) * z;
                        j1[q] = fy * (dY - y * dZ) * z;
                    }
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
#pragma unroll
                        for (int bb = a; bb < 6; bb++, q++) acc[q] += j0[a] * j0[bb] + j1[a] * j1[bb];
                        acc[21 + a] += j0[a] * e0 + j1[a] * e1;
                    }
                    acc[27] += e0 * e0 + e1 * e1;
                }
            }
            PNP_LM_LAP(1);
            block_reduce<NLM>(S, acc);
            PNP_LM_LAP(2);
            if (tid < 6) S.prev[tid] = S.param[tid];
            if (iters == 0) prevErrNorm = sqrt(S.sums[27]);
            double JtJ[36], JtErr[6];
            {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int bb = a; bb < 6; bb++, q++) JtJ[a * 6 + bb] = JtJ[bb * 6 + a] = S.sums[q];
            }
#pragma unroll
            for (int j = 0; j < 6; j++) JtErr[j] = S.sums[21 + j];
            __syncthreads();
            for (;;) {
                if (tid == 0) {
                    const double lambda = exp(lambdaLg10 * LOG10);
                    double A[36], dx[6];
#pragma unroll
                    for (int j = 0; j < 36; j++) A[j] = JtJ[j];
#pragma unroll
                    for (int a = 0; a < 6; a++) A[a * 6 + a] *= 1. + lambda;
                    if (!solve_spd6(A, JtErr, dx)) {
#pragma unroll
                        for (int a = 0; a < 6; a++) dx[a] = 0;
                    }
#pragma unroll
                    for (int a = 0; a < 6; a++) S.param[a] = S.prev[a] - dx[a];
                }
                __syncthreads();
                PNP_LM_LAP(3);
                PNP_LM_COUNT(5);
                double e2[1] = {0};
                {
                    double R[9], p[6];
                    for (int j = 0; j < 6; j++) p[j] = S.param[j];
                    rodrigues_v2m(p, R, nullptr);
                    for (int i = tid; i < n; i += PR_THREADS) {
#pragma clang fp contract(fast)
                        if (!msk[i]) continue;
                        const double M[3] = {obj[3 * i], obj[3 * i + 1], obj[3 * i + 2]};
                        const double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + p[3];
                        const double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + p[4];
                        const double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + p[5];
                        const double z = Z ? 1. / Z : 1;
                        const double e0 = X * z * fx + cx - (double)img[2 * i];
                        const double e1 = Y * z * fy + cy - (double)img[2 * i + 1];
                        e2[0] += e0 * e0 + e1 * e1;
                    }
                }
                block_reduce<1>(S, e2);
                errNorm = sqrt(S.sums[0]);
                __syncthreads();
                PNP_LM_LAP(4);
                if (errNorm > prevErrNorm) {
                    if (++lambdaLg10 <= 16) continue;
                }
                break;
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            double dn = 0, pn = 0;
            for (int a = 0; a < 6; a++) {
                dn += (S.param[a] - S.prev[a]) * (S.param[a] - S.prev[a]);
                pn += S.prev[a] * S.prev[a];
            }
            PNP_LM_COUNT(0);
            if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < eps) break;
            prevErrNorm = errNorm;
            __syncthreads();
        }
    }
    __syncthreads();
    PNP_STAMP(10);
    if (tid == 0) {
        // no usable DLT solution: OpenCV reports failure and hands back the best RANSAC hypothesis; five
        // non-planar inliers: that hypothesis is the result
        for (int j = 0; j < 6; j++) rto[j] = (init_ok && init_mode != 2) ? S.param[j] : S.best[j];
        po[0] = init_ok ? 1 : 0;
        po[1] = init_ok ? k : 0;
        po[2] = S.iter;
        po[3] = n;
    }
}

// Hand the frame in S over: chain the RNG stream into work items of up to PR_CHUNK hypotheses covering the
// iterations the current bound still admits (a window that ends early yields a shorter item) and publish them.
// False when not a single hypothesis could be chained (RNG table exhausted): the caller finishes the frame itself.
__device__ __forceinline__ bool pnp_publish_items(PrShared &S, const PnpArgs &a, int b, int n)
{
    const int tid = threadIdx.x;
    const int want_all = S.niters - S.iter;
    int covered = 0, n_items = 0;
    PnpItem *mine = a.items + (size_t)b * kPnpItemsPerFrame;
    // One window for all of them (the claimers chain their own ten again from the item's position): every offset
    // simulated as the start of a subset draw by all threads, then one lane walks the chain and notes where every
    // PR_CHUNK-th hypothesis starts.  Chaining item by item cost a handed-over frame 150 k - 300 k cycles before
    // its first item was on offer.
    while (covered < want_all && n_items < kPnpItemsPerFrame) {
        const long long pos = S.pos;
        for (int k = tid; k < PR_WIN_ALL; k += PR_THREADS) {
            const long long w = pos + k;
            S.r_all[k] = w < a.rng_len ? (uint16_t)(a.rng_raw[w] % (uint32_t)n) : (uint16_t)0;
        }
        __syncthreads();
        for (int o = tid; o < PR_WIN_ALL; o += PR_THREADS) {
            int idx[5];
            int cons = draw_subset5(S.r_all, o, idx, PR_WIN_ALL);
            if (pos + o + cons > a.rng_len) cons = 0;
            S.cons_all[o] = (uint8_t)cons;
        }
        __syncthreads();
        if (tid == 0) {
            const int want = want_all - covered, room = kPnpItemsPerFrame - n_items;
            int w = 0, nc = 0, ni = 0;
            while (nc < want && w < PR_WIN_ALL && S.cons_all[w] != 0) {
                const int c = S.cons_all[w];
                // a new item after PR_CHUNK hypotheses, or when this one would end outside the window of PR_WIN
                // words in which the item's claimer chains them again (many redraws: very few points)
                if (ni == 0 || S.item_count[ni - 1] == PR_CHUNK || w + c - S.item_start[ni - 1] > PR_WIN) {
                    if (ni == room) break;
                    S.item_start[ni] = w;
                    S.item_first[ni] = nc;
                    S.item_count[ni] = 0;
                    ni++;
                }
                S.item_count[ni - 1]++;
                nc++;
                w += c;
            }
            for (int j = 0; j < ni; j++)
                mine[n_items + j] = PnpItem{pos + S.item_start[j], covered + S.item_first[j], S.item_count[j]};
            S.nchunk = nc;
            S.n_new_items = ni;
            S.pos = pos + w;
        }
        __syncthreads();
        if (S.nchunk == 0) break;
        covered += S.nchunk;
        n_items += S.n_new_items;
        __syncthreads();  // (S.nchunk is written again in the next pass)
    }
    if (n_items == 0) return false;
    if (tid == 0) {
        PnpResume &r = a.resume[b];
        for (int i = 0; i < 6; i++) r.best[i] = S.best[i];
        r.pos = S.pos; r.iter = S.iter; r.niters = S.niters; r.max_good = S.max_good; r.n_spec = covered;
        __hip_atomic_store(&a.hint[b], r.niters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.resumed[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();  // items, resume record and hint before the word that announces them
        __hip_atomic_store(&a.state[b], n_items, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// Claim one published work item: the frame nearest to this block's own (in cyclic order) that still has unclaimed
// items.  Returns the frame (item index and item count of the frame in S.claim_item / S.claim_n) or -1 when nothing
// is on offer.  Every thread calls it.
__device__ __forceinline__ int pnp_claim(PrShared &S, const PnpArgs &a)
{
    const int tid = threadIdx.x;
    const int first = (int)(blockIdx.x % (unsigned)a.batch);  // (a sweep launch may have more blocks than frames)
    for (;;) {
        if (tid == 0) S.cand = 0x7fffffff;
        __syncthreads();
        for (int k = tid; k < a.batch; k += PR_THREADS) {
            int bb = first + k;
            if (bb >= a.batch) bb -= a.batch;
            const int st = __hip_atomic_load(&a.state[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st > 0 && __hip_atomic_load(&a.next[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < st)
                atomicMin(&S.cand, k);
        }
        __syncthreads();
        const int k = S.cand;
        if (k == 0x7fffffff) return -1;
        int bb = first + k;
        if (bb >= a.batch) bb -= a.batch;
        if (tid == 0) {
            const int st = __hip_atomic_load(&a.state[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int i = atomicAdd(&a.next[bb], 1);
            S.claim_item = i < st ? i : -1;  // somebody else took the last one in between: look again
            S.claim_n = st;
            __threadfence();  // what the publisher wrote before `state` is read after this
            // (items are claimed in ascending order: every item finished by now lies before this one, so the bound
            // they have set holds at this item's first hypothesis)
            S.claim_hint = __hip_atomic_load(&a.hint[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a frame that has been taken back needs none of its remaining items)
            if (__hip_atomic_load(&a.resumed[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) S.claim_hint = 0;
        }
        __syncthreads();
        if (S.claim_item >= 0) return bb;
    }
}

// Replay of cv::RANSACPointSetRegistrator::run's acceptance rule over the first n_hyps stored hypotheses of frame
// b, in RNG order: a hypothesis is accepted when its count beats every earlier one (a prefix maximum, found by all
// threads at once); one lane then walks the accepted ones, moving the iteration bound, and finds the index at
// which the sequential loop would have stopped.  Leaves S as that loop would (S.done = 1 when it stopped inside the
// stored hypotheses).  Every thread calls it; ends with a barrier.
__device__ __forceinline__ void pnp_replay(PrShared &S, const PnpArgs &a, int b, int n, int n_hyps)
{
    static_assert(kPnpSpecMax <= PR_THREADS, "one stored hypothesis per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PnpResume r = a.resume[b];
    const PnpHyp *hy = a.hyp + (size_t)b * kPnpSpecMax;
    int c = -1;
    if (tid < n_hyps && hy[tid].ok > 0) c = hy[tid].count;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(incl, off);
        if (lane >= off) incl = max(incl, y);
    }
    if (lane == 63) S.wave_tot[wave] = incl;
    int excl = __shfl_up(incl, 1);
    if (lane == 0) excl = -1;
    __syncthreads();
    for (int w = 0; w < wave; w++) excl = max(excl, S.wave_tot[w]);
    const bool acc = c > max(excl, max(r.max_good, 4));
    const unsigned long long bm = __ballot(acc);
    if (lane == 0) S.acc_mask[wave] = bm;
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < 6; i++) S.best[i] = r.best[i];
        int N = r.niters, max_good = r.max_good;
        int prev = 0, final_iter = r.iter + n_hyps;
        bool done = false;
        // between two accepted hypotheses the bound N is constant: the loop stops at the first index j of the
        // stretch with iter + 1 = r.iter + j + 1 >= N
        for (int w = 0; w < PR_WAVES && !done; w++) {
            unsigned long long m = S.acc_mask[w];
            while (m) {
                const int h = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const int j = max(prev, N - r.iter - 1);
                if (j <= h - 1) { done = true; final_iter = r.iter + j + 1; break; }
                for (int i = 0; i < 6; i++) S.best[i] = hy[h].model[i];
                max_good = hy[h].count;
                N = pnp_update_num_iters(a.confidence, (double)(n - max_good) / n, 5, N);
                prev = h;
            }
        }
        if (!done) {
            const int j = max(prev, N - r.iter - 1);
            if (j <= n_hyps - 1) { done = true; final_iter = r.iter + j + 1; }
        }
        S.pos = r.pos; S.iter = final_iter; S.niters = N; S.max_good = max_good; S.done = done ? 1 : 0;
    }
    __syncthreads();
}

// out: pose[b] = {success, n_inliers, iters_run, n_points}, rt[b] = rvec, tvec.
// sweep = 0: block b owns frame b, then helps with published work items; sweep = 1: published work items only.
// disable_tail_calls: a call the optimiser has marked as a possible tail call keeps its callee on the standard
// calling convention, under which solve_hypothesis would save and restore every callee-saved register it touches --
// some 330 stores and 330 loads of a wave's private memory per call, about 19 us.  Without the mark the compiler sees
// that the function is local and all its callers known, and saves only what the caller has live across the call.
__global__ __launch_bounds__(PR_THREADS) __attribute__((disable_tail_calls)) void k_pnp_ransac(PnpArgs a, int sweep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    PrShared &S = *reinterpret_cast<PrShared *>(s_dyn);
    float *obj, *img;
    uint8_t *msk;
    pnp_lds_views(s_dyn, a.kp_stride, obj, img, msk);

    const int tid = threadIdx.x, wave = tid >> 6;
    const int kp_stride = a.kp_stride;
    const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
    const float thr2 = (float)(a.threshold * a.threshold);
    const bool share = a.state != nullptr;
    int b = blockIdx.x;  // the frame whose points are in LDS and whose RANSAC state is in S
    int n = 0;
    bool run = false;    // that frame is to be driven to its end by this block
    bool own = false;    // ... and it is this block's own frame (it may still be handed over)

    PNP_STAMP(0);
    PNP_BLOCK(0, __builtin_amdgcn_s_memtime());
    for (int i = 3; i < 7; i++) PNP_BLOCK(i, 0ull);
    if (!sweep) {
        n = min(a.bp_info[b * 2 + 1], kp_stride);
        const vo_match *vm = a.valid + (size_t)b * kp_stride;
        const float *p3 = a.pts3 + (size_t)b * kp_stride * 3;
        int *inl = a.inliers + (size_t)b * kp_stride;
        int *po = a.pose + b * 4;
        double *rto = a.rt + (size_t)b * 6;
        if (tid < 6) rto[tid] = 0;
        // reference src/frame_processor.cpp:176-177 asks for >= 4 points; with exactly 4 OpenCV runs its P3P
        // solver once, without RANSAC, and reports all four as inliers
        if (n < 5) {
            __syncthreads();  // the zeros above land before the pose
            if (tid == 0) {
                int ok = 0;
                if (n == 4) {
                    float o4[12], i4[8];
                    for (int i = 0; i < 4; i++) {
                        o4[3 * i] = p3[3 * i]; o4[3 * i + 1] = p3[3 * i + 1]; o4[3 * i + 2] = p3[3 * i + 2];
                        i4[2 * i] = vm[i].px; i4[2 * i + 1] = vm[i].py;
                    }
                    double R[9], t[3], rv[3];
                    if (p3p_solve4(o4, i4, fx, fy, cx, cy, R, t)) {
                        rodrigues_m2v(R, rv);
                        for (int i = 0; i < 3; i++) { rto[i] = rv[i]; rto[3 + i] = t[i]; }
                        for (int i = 0; i < 4; i++) inl[i] = i;
                        ok = 1;
                    }
                }
                po[0] = ok; po[1] = ok ? 4 : 0; po[2] = 0; po[3] = n;
            }
            return;
        }
        pnp_load_points(vm, p3, n, obj, img, msk);
        if (tid == 0) {
            S.iter = 0; S.niters = a.max_iters > 1 ? a.max_iters : 1; S.max_good = 0; S.done = 0; S.pos = 0;
        }
        __syncthreads();

        if (n == 5) {
            // OpenCV runs the minimal solver once and reports every point as inlier
            if (tid < 5) S.r[tid] = (uint16_t)tid;
            if (tid == 0) S.start[0] = 0;
            __syncthreads();
            if (wave == 0) solve_hypothesis(S, obj, img, 0, 1, 1, fx, fy, cx, cy);
            __syncthreads();
            if (tid == 0) {
                const int ok = S.ok[0];
                if (ok) {
                    for (int i = 0; i < 6; i++) rto[i] = S.model[0][i];
                    for (int i = 0; i < 5; i++) inl[i] = i;
                }
                po[0] = ok; po[1] = ok ? 5 : 0; po[2] = 0; po[3] = n;
            }
            return;
        }
        run = own = true;
    }

    PNP_STAMP(1);
    for (;;) {
        if (run) {
            if (!S.done) pnp_ransac_rounds(S, obj, img, n, a, thr2, own && share);
            bool handed = false;
            if (S.done == 2) handed = pnp_publish_items(S, a, b, n);
            if (handed) PNP_BLOCK(3, 1ull);
            if (!own) PNP_BLOCK_ADD(6, 1ull);
            if (!handed)
                pnp_finish(S, obj, img, msk, n, fx, fy, cx, cy, thr2, a.inliers + (size_t)b * kp_stride,
                           a.pose + b * 4, a.rt + (size_t)b * 6);
            if (own) PNP_BLOCK(1, __builtin_amdgcn_s_memtime());
            run = own = false;
        }
        PNP_BLOCK(2, __builtin_amdgcn_s_memtime());
        if (!share || (!sweep && !a.steal)) break;
        const int bb = pnp_claim(S, a);
        if (bb < 0) break;
        b = bb;
        n = min(a.bp_info[b * 2 + 1], kp_stride);
        const int item = S.claim_item;
        const PnpItem it = a.items[(size_t)b * kPnpItemsPerFrame + item];
        const int n_items = S.claim_n;
        const PnpResume rs = a.resume[b];
        // The sequential loop stops at an iteration count no later than the bound the finished items have set.  An
        // item whose first hypothesis lies at or behind it is never looked at by the replay -- nor is any later
        // item: the block takes all that are still unclaimed and records them as empty.
        const bool skip = rs.iter + it.first >= S.claim_hint;
        int bits = 1 << item;  // items this block completes
        __syncthreads();  // the previous frame's points and S.claim_* are no longer read
        if (skip) PNP_BLOCK_ADD(5, 1ull); else PNP_BLOCK_ADD(4, 1ull);
        if (skip) {
            if (tid == 0) S.cand = atomicAdd(&a.next[b], kPnpItemsPerFrame + 1);  // (>= n_items when others took the rest)
            __syncthreads();
            PnpHyp *hy = a.hyp + (size_t)b * kPnpSpecMax;
            // hypotheses of this item, and from the first unclaimed item on (contiguous up to n_spec)
            for (int h = it.first + tid; h < it.first + it.count; h += PR_THREADS) { hy[h].ok = 0; hy[h].count = 0; }
            if (S.cand < n_items) {
                bits |= ((1 << n_items) - 1) & ~((1 << S.cand) - 1);
                const int from = a.items[(size_t)b * kPnpItemsPerFrame + S.cand].first;
                for (int h = from + tid; h < rs.n_spec; h += PR_THREADS) { hy[h].ok = 0; hy[h].count = 0; }
            }
        } else {
            pnp_load_points(a.valid + (size_t)b * kp_stride, a.pts3 + (size_t)b * kp_stride * 3, n, obj, img, msk);
            if (tid == 0) S.pos = it.pos;
            __syncthreads();
            pnp_chain_round(S, n, a.rng_raw, a.rng_len, it.count);
            pnp_solve_score(S, obj, img, n, PR_PER_WAVE, fx, fy, cx, cy, thr2);
            const int nc = S.nchunk;  // == it.count: the same window, the same chain
            PnpHyp *hy = a.hyp + (size_t)b * kPnpSpecMax + it.first;
            if (tid < nc) {
                const int ok = S.ok[tid];
                for (int i = 0; i < 6; i++) hy[tid].model[i] = ok ? S.model[tid][i] : 0.0;
                hy[tid].count = ok ? S.counts[tid] : 0;
                hy[tid].ok = ok ? 1 : 0;
            }
            if (tid == 0) {
                // the best count of this item bounds the iteration count from here on (the running best of the
                // sequential loop is at least this, and the bound only falls with the count)
                int cmax = 0;
                for (int h = 0; h < nc; h++)
                    if (S.ok[h]) cmax = max(cmax, S.counts[h]);
                if (cmax > max(rs.max_good, 4))
                    atomicMin(&a.hint[b], pnp_update_num_iters(a.confidence, (double)(n - cmax) / n, 5, rs.niters));
            }
        }
        __threadfence();  // the records before the completion bits
        __syncthreads();
        if (tid == 0) {
            const int m = atomicOr(&a.done[b], bits) | bits;
            __threadfence();  // the other items' records (completed before these) are read after this
            // items 0 .. p-1 are complete: their hypotheses can be replayed, whatever the later items are doing
            S.claim_n = __builtin_ctz(~m);
        }
        __syncthreads();
        const int p = S.claim_n;
        if (p > 0 && !__hip_atomic_load(&a.resumed[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const PnpItem lastp = a.items[(size_t)b * kPnpItemsPerFrame + p - 1];
            pnp_replay(S, a, b, n, p == n_items ? rs.n_spec : lastp.first + lastp.count);
            // The frame goes on as soon as the sequential loop is known to stop inside the complete items (or all
            // items are complete): one block takes it back, the items still out are of no interest any more.
            if (S.done || p == n_items) {
                if (tid == 0) S.last = atomicCAS(&a.resumed[b], 0, 1) == 0;
                __syncthreads();
                if (S.last) {
                    if (skip) {
                        pnp_load_points(a.valid + (size_t)b * kp_stride, a.pts3 + (size_t)b * kp_stride * 3, n, obj, img,
                                        msk);
                        __syncthreads();
                    }
                    run = true;
                }
            }
        }
    }
}

size_t pnp_phase2_bytes(int batch, size_t off[4])
{
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~size_t(255); return at; };
    off[0] = take(5 * sizeof(int) * (size_t)batch);                                // state, next, done, hint, resumed
    off[1] = take(sizeof(PnpItem) * (size_t)batch * kPnpItemsPerFrame);            // items
    off[2] = take(sizeof(PnpResume) * (size_t)batch);                              // resume
    off[3] = take(sizeof(PnpHyp) * (size_t)batch * kPnpSpecMax);                   // hyp
    return o;
}

void launch_pnp_ransac(const vo_match *valid, const float *pts3, const int *bp_info, int kp_stride,
                       const uint32_t *rng_raw, int rng_len, double fx, double fy, double cx, double cy,
                       int max_iters, double threshold, double confidence, int *inliers, int *pose,
                       double *rt, int batch, void *phase2, hipStream_t s, bool zeroed)
{
    const size_t lds = ((sizeof(PrShared) + 15) & ~size_t(15)) + (size_t)kp_stride * (5 * sizeof(float) + 1) + 16;
    PnpArgs a{valid, pts3, bp_info, kp_stride, rng_raw, rng_len, fx, fy, cx, cy, max_iters, threshold, confidence,
              inliers, pose, rt, batch, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // handing frames over pays when hypotheses can outnumber two rounds
    // (VO_PNP_SHARE=0: every frame stays in its own block, for comparisons)
    static const bool share_ok = !(getenv("VO_PNP_SHARE") && getenv("VO_PNP_SHARE")[0] == '0');
    const bool share = share_ok && phase2 != nullptr && max_iters > PR_FIRST + PR_CHUNK;
    if (share) {
        size_t off[4];
        (void)pnp_phase2_bytes(batch, off);
        unsigned char *p = static_cast<unsigned char *>(phase2);
        a.state = reinterpret_cast<int *>(p + off[0]);
        a.next = a.state + batch;
        a.done = a.next + batch;
        a.hint = a.done + batch;
        a.resumed = a.hint + batch;
        a.items = reinterpret_cast<PnpItem *>(p + off[1]);
        a.resume = reinterpret_cast<PnpResume *>(p + off[2]);
        a.hyp = reinterpret_cast<PnpHyp *>(p + off[3]);
        if (!zeroed) (void)hipMemsetAsync(a.state, 0, 3 * sizeof(int) * (size_t)batch, s);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pnp_ransac),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // few frames: too few blocks to share the published work among, a second launch brings them
    const bool few = batch < 64;
    a.steal = few ? 0 : 1;
    hipLaunchKernelGGL(k_pnp_ransac, dim3(batch), dim3(PR_THREADS), lds, s, a, 0);
    if (share && few)
        hipLaunchKernelGGL(k_pnp_ransac, dim3(min(batch * kPnpItemsPerFrame, 512)), dim3(PR_THREADS), lds, s, a, 1);
}

}  // namespace vo
