// Essential-matrix estimation for gfx950: batched five-point hypotheses,
// Sampson-error scoring and the RANSAC loop of cv::findEssentialMat(p1, p2, K,
// RANSAC, prob, threshold, maxIters, mask).
//
// The reference takes its pose from cv::solvePnPRansac on depth-backprojected
// points (reference src/frame_processor.cpp:176-196) and never estimates an
// essential matrix; this is the monocular estimator on top of it.  The minimal
// solver follows Nister's construction: null space of the 5x9 epipolar system,
// the ten cubic constraints det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0 over the
// 20 monomials of degree <= 3 in (x, y, z), Gauss-Jordan elimination, a 3x3
// polynomial matrix in z whose determinant is the tenth-degree polynomial; its
// real roots come from a deterministic derivative-chain bisection.
//
// One lane solves one hypothesis (fp64, the working set of ~300 doubles lives
// in per-lane scratch that stays in L2); scoring uses one wave per candidate
// matrix.  No MFMA: the contractions are 3x3 and 10x20.
#include <float.h>

#include "vo_internal.h"

namespace vo {

namespace {

// ---- polynomial bookkeeping ------------------------------------------------
// cubic monomials, the column order of the constraint matrix:
//  0 x^3  1 y^3  2 x^2y 3 xy^2 4 x^2z 5 x^2  6 y^2z 7 y^2  8 xyz  9 xy
// 10 xz^2 11 xz  12 x   13 yz^2 14 yz  15 y   16 z^3  17 z^2 18 z   19 1
// quadratic monomials: 0 x^2 1 y^2 2 z^2 3 xy 4 xz 5 yz 6 x 7 y 8 z 9 1
// linear forms: (x, y, z, 1)
__device__ constexpr int kLinLin[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
__device__ constexpr int kQuadLin[10][4] = {
    {0, 2, 4, 5},     // x^2 * (x, y, z, 1)
    {3, 1, 6, 7},     // y^2
    {10, 13, 16, 17}, // z^2
    {2, 3, 8, 9},     // xy
    {4, 8, 10, 11},   // xz
    {8, 6, 13, 14},   // yz
    {5, 9, 11, 12},   // x
    {9, 7, 14, 15},   // y
    {11, 14, 17, 18}, // z
    {12, 15, 18, 19}, // 1
};

__device__ __forceinline__ void lin_lin(const double *a, const double *b, double *q /*10, accumulated*/)
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) q[kLinLin[i][j]] += a[i] * b[j];
}

__device__ __forceinline__ void quad_lin(const double *q, const double *l, double s, double *c /*20, accumulated*/)
{
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) c[kQuadLin[i][j]] += s * (q[i] * l[j]);
}

__device__ __forceinline__ double horner(const double *p, int deg, double x)
{
    double v = p[deg];
    for (int i = deg - 1; i >= 0; i--) v = v * x + p[i];
    return v;
}

// ascending real roots of p (degree <= 10) with sign change: the roots of each
// derivative split [-B, B] (Cauchy bound) into monotone pieces that are bisected
__device__ int real_roots10(const double *p_in, double *roots)
{
    int deg = 10;
    while (deg > 0 && p_in[deg] == 0) deg--;
    if (deg <= 0) return 0;
    double P[11][11];  // P[d]: the (deg-d)-th derivative, degree d
    for (int i = 0; i <= deg; i++) P[deg][i] = p_in[i];
    for (int d = deg; d > 1; d--)
        for (int i = 1; i <= d; i++) P[d - 1][i - 1] = P[d][i] * i;
    double crit[10], cur[10];
    int nc = 1;
    crit[0] = -P[1][0] / P[1][1];
    for (int d = 2; d <= deg; d++) {
        const double *p = P[d];
        double bound = 0;
        for (int i = 0; i < d; i++) bound = fmax(bound, fabs(p[i] / p[d]));
        bound += 1.0;
        int n = 0;
        double lo = -bound, flo = horner(p, d, lo);
        for (int k = 0; k <= nc; k++) {
            double hi = k < nc ? crit[k] : bound;
            if (hi <= lo) continue;
            if (hi > bound) hi = bound;
            const double fhi = horner(p, d, hi);
            if (flo == 0) {
                cur[n++] = lo;
            } else if (fhi != 0 && (flo < 0) != (fhi < 0)) {
                double a = lo, b = hi, fa = flo, r;
                bool exact = false;
                for (int it = 0; it < 200; it++) {
                    const double mid = 0.5 * (a + b);
                    if (mid == a || mid == b) break;
                    const double fm = horner(p, d, mid);
                    if (fm == 0) { a = b = mid; exact = true; break; }
                    if ((fm < 0) == (fa < 0)) { a = mid; fa = fm; }
                    else b = mid;
                }
                r = exact ? a : 0.5 * (a + b);
                cur[n++] = r;
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

// q1, q2: five normalised correspondences.  Up to ten matrices (row-major,
// Frobenius norm 1, ascending root order) with q2^T E q1 = 0.
__device__ int solve_e5(const double (&q1)[5][2], const double (&q2)[5][2], double *Eout /*90*/)
{
    // ---- null space of the 5x9 system: Householder QR of its transpose
    double N[4][9];
    {
        double M[9][5], V[5][9], beta[5];
        for (int i = 0; i < 5; i++) {
            const double x1 = q1[i][0], y1 = q1[i][1], x2 = q2[i][0], y2 = q2[i][1];
            M[0][i] = x2 * x1; M[1][i] = x2 * y1; M[2][i] = x2;
            M[3][i] = y2 * x1; M[4][i] = y2 * y1; M[5][i] = y2;
            M[6][i] = x1; M[7][i] = y1; M[8][i] = 1.0;
        }
        for (int k = 0; k < 5; k++) {
            double s = 0;
            for (int i = k; i < 9; i++) s += M[i][k] * M[i][k];
            const double nrm = sqrt(s);
            const double alpha = M[k][k] > 0 ? -nrm : nrm;
            for (int i = 0; i < 9; i++) V[k][i] = i >= k ? M[i][k] : 0.0;
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
            double q[9];
            for (int i = 0; i < 9; i++) q[i] = i == 5 + c ? 1.0 : 0.0;
            for (int k = 4; k >= 0; k--) {
                double dot = 0;
                for (int i = k; i < 9; i++) dot += V[k][i] * q[i];
                dot *= beta[k];
                for (int i = k; i < 9; i++) q[i] -= dot * V[k][i];
            }
            for (int i = 0; i < 9; i++) N[c][i] = q[i];
        }
    }

    // ---- the ten cubic constraints on E = x N0 + y N1 + z N2 + N3
    double C[10][20];
    {
        double lin[9][4];
        for (int e = 0; e < 9; e++)
            for (int k = 0; k < 4; k++) lin[e][k] = N[k][e];
        double EEt[3][3][10], tr[10];
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
                double *c = C[3 * i + j];
                for (int m = 0; m < 20; m++) c[m] = 0;
                for (int k = 0; k < 3; k++) quad_lin(EEt[i][k], lin[3 * k + j], 2.0, c);
                quad_lin(tr, lin[3 * i + j], -1.0, c);
            }
        double *c = C[9];
        for (int m = 0; m < 20; m++) c[m] = 0;
        const int perm[6][3] = {{0, 4, 8}, {1, 5, 6}, {2, 3, 7}, {2, 4, 6}, {1, 3, 8}, {0, 5, 7}};
        for (int t = 0; t < 6; t++) {
            double ab[10];
            for (int m = 0; m < 10; m++) ab[m] = 0;
            lin_lin(lin[perm[t][0]], lin[perm[t][1]], ab);
            quad_lin(ab, lin[perm[t][2]], t < 3 ? 1.0 : -1.0, c);
        }
    }

    // ---- Gauss-Jordan on the first ten columns, partial pivoting
    for (int col = 0; col < 10; col++) {
        int piv = col;
        for (int r = col + 1; r < 10; r++)
            if (fabs(C[r][col]) > fabs(C[piv][col])) piv = r;
        if (fabs(C[piv][col]) < 1e-300) return 0;
        if (piv != col)
            for (int m = 0; m < 20; m++) { const double t = C[piv][m]; C[piv][m] = C[col][m]; C[col][m] = t; }
        const double inv = 1.0 / C[col][col];
        for (int m = col; m < 20; m++) C[col][m] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            const double f = C[r][col];
            if (f == 0) continue;
            for (int m = col; m < 20; m++) C[r][m] -= f * C[col][m];
        }
    }

    // ---- B(z) [x y 1]^T = 0 from (x^2z, x^2), (y^2z, y^2), (xyz, xy); ascending powers of z
    double B[3][3][5];
    for (int r = 0; r < 3; r++) {
        const double *hi = C[4 + 2 * r], *lo = C[5 + 2 * r];
        for (int v = 0; v < 2; v++) {  // x part (columns 10..12), y part (13..15)
            const int o = 10 + 3 * v;
            B[r][v][0] = hi[o + 2];
            B[r][v][1] = hi[o + 1] - lo[o + 2];
            B[r][v][2] = hi[o] - lo[o + 1];
            B[r][v][3] = -lo[o];
            B[r][v][4] = 0;
        }
        B[r][2][0] = hi[19];
        B[r][2][1] = hi[18] - lo[19];
        B[r][2][2] = hi[17] - lo[18];
        B[r][2][3] = hi[16] - lo[17];
        B[r][2][4] = -lo[16];
    }
    double p[11];
    for (int i = 0; i <= 10; i++) p[i] = 0;
    for (int c = 0; c < 3; c++) {
        const int a = (c + 1) % 3, b = (c + 2) % 3;  // cyclic minors: every sign is +
        const int da = a == 2 ? 4 : 3, db = b == 2 ? 4 : 3, dc = c == 2 ? 4 : 3;
        double t1[8], t2[8], m[12];
        for (int i = 0; i <= da + db; i++) t1[i] = t2[i] = 0;
        for (int i = 0; i <= da; i++)
            for (int j = 0; j <= db; j++) t1[i + j] += B[1][a][i] * B[2][b][j];
        for (int i = 0; i <= db; i++)
            for (int j = 0; j <= da; j++) t2[i + j] += B[1][b][i] * B[2][a][j];
        for (int i = 0; i <= da + db; i++) t1[i] -= t2[i];
        for (int i = 0; i <= dc + da + db; i++) m[i] = 0;
        for (int i = 0; i <= dc; i++)
            for (int j = 0; j <= da + db; j++) m[i + j] += B[0][c][i] * t1[j];
        for (int i = 0; i <= 10 && i <= dc + da + db; i++) p[i] += m[i];
    }

    double roots[10];
    const int nr = real_roots10(p, roots);
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

// Sampson distance (x2^T E x1)^2 / (Ex1_0^2 + Ex1_1^2 + Etx2_0^2 + Etx2_1^2), as float
__device__ __forceinline__ float e_error(double x1, double y1, double x2, double y2, const double *E)
{
    const double a0 = E[0] * x1 + E[1] * y1 + E[2], a1 = E[3] * x1 + E[4] * y1 + E[5],
                 a2 = E[6] * x1 + E[7] * y1 + E[8];
    const double b0 = E[0] * x2 + E[3] * y2 + E[6], b1 = E[1] * x2 + E[4] * y2 + E[7];
    const double d = x2 * a0 + y2 * a1 + a2;
    return (float)(d * d / (a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1));
}

constexpr int EH_THREADS = 64;

// one lane per hypothesis of the index table
__global__ __launch_bounds__(EH_THREADS) void k_e_hypotheses(const double *__restrict__ q, int n,
                                                            const int *__restrict__ idx, int H,
                                                            double *__restrict__ E, int *__restrict__ n_models)
{
    const int h = blockIdx.x * EH_THREADS + threadIdx.x;
    if (h >= H) return;
    double q1[5][2], q2[5][2];
    bool ok = true;
    for (int i = 0; i < 5; i++) {
        const int j = idx[5 * h + i];
        ok = ok && j >= 0 && j < n;
        const int jj = ok ? j : 0;
        q1[i][0] = q[4 * jj]; q1[i][1] = q[4 * jj + 1]; q2[i][0] = q[4 * jj + 2]; q2[i][1] = q[4 * jj + 3];
    }
    double out[90];
    const int nm = ok ? solve_e5(q1, q2, out) : 0;
    n_models[h] = nm;
    for (int i = 0; i < 9 * nm; i++) E[(size_t)h * 90 + i] = out[i];
}

// one wave per candidate matrix: number of points with Sampson error <= thr2
__global__ __launch_bounds__(256) void k_e_score(const double *__restrict__ q, int n, const double *__restrict__ E,
                                                 const int *__restrict__ n_models, int H, float thr2,
                                                 int *__restrict__ counts)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= H * 10) return;
    const int h = w / 10, m = w - 10 * h;
    int cnt = 0;
    if (m < n_models[h]) {
        double e[9];
        for (int i = 0; i < 9; i++) e[i] = E[(size_t)h * 90 + 9 * m + i];
        for (int i = lane; i < n; i += 64) {
            const double4 p = reinterpret_cast<const double4 *>(q)[i];
            cnt += e_error(p.x, p.y, p.z, p.w, e) <= thr2;
        }
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) counts[w] = cnt;
}

// ---- RANSAC (one block per problem) ------------------------------------------

constexpr int ER_THREADS = 256;
constexpr int ER_CHUNK = 64;              // hypotheses per round (one wave solves them)
constexpr int ER_WIN = ER_CHUNK * 5 + 96;  // RNG words examined per round

struct ErShared {
    double models[ER_CHUNK][10][9];
    double best[9];
    int nmodels[ER_CHUNK];
    int counts[ER_CHUNK][10];
    int start[ER_CHUNK];
    uint16_t r[ER_WIN];
    uint8_t cons[ER_WIN];
    int wave_tot[ER_THREADS / 64];
    int nchunk, iter, niters, max_good, done;
    long long pos;
};

__device__ __forceinline__ int draw5(const uint16_t *r, int o, int (&idx)[5])
{
    int p = o;
    for (int i = 0; i < 5; i++) {
        for (;;) {
            if (p >= ER_WIN) return 0;
            const int v = r[p++];
            bool dup = false;
            for (int j = 0; j < i; j++) dup = dup || (idx[j] == v);
            if (!dup) { idx[i] = v; break; }
        }
    }
    return p - o;
}

__device__ int e_update_num_iters(double p, double ep, int model_points, int max_iters)
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

// q: n x (x1, y1, x2, y2) normalised points in global memory.
// info = {found, iters_run, n_inliers}
__global__ __launch_bounds__(ER_THREADS) void k_e_ransac(const double *__restrict__ q, int n,
                                                        const uint32_t *__restrict__ rng_raw, int rng_len,
                                                        double prob, float thr2, int max_iters,
                                                        uint8_t *__restrict__ mask, double *__restrict__ Eout,
                                                        int *__restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    ErShared &S = *reinterpret_cast<ErShared *>(s_dyn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double4 *pts = reinterpret_cast<const double4 *>(q);

    if (tid == 0) {
        S.iter = 0; S.niters = max_iters > 1 ? max_iters : 1; S.max_good = 0; S.done = 0; S.pos = 0;
    }
    __syncthreads();
    while (true) {
        const long long pos = S.pos;
        for (int k = tid; k < ER_WIN; k += ER_THREADS) {
            const long long a = pos + k;
            S.r[k] = a < rng_len ? (uint16_t)(rng_raw[a] % (uint32_t)n) : (uint16_t)0;
        }
        __syncthreads();
        for (int o = tid; o < ER_WIN; o += ER_THREADS) {
            int idx[5];
            int cons = n == 5 ? 0 : draw5(S.r, o, idx);
            if (pos + o + cons > rng_len) cons = 0;
            S.cons[o] = (uint8_t)cons;
        }
        __syncthreads();
        if (tid == 0) {
            int w = 0, nc = 0;
            const int want = min(ER_CHUNK, S.niters - S.iter);
            if (n == 5) {
                // exactly five points: the sample is the point set, no random draw
                while (nc < want) S.start[nc++] = -1;
            } else {
                while (nc < want && w < ER_WIN && S.cons[w] != 0) { S.start[nc++] = w; w += S.cons[w]; }
            }
            S.nchunk = nc;
            S.pos = pos + w;
            if (nc == 0) S.done = 1;  // RNG table exhausted
        }
        __syncthreads();
        const int nc = S.nchunk;
        if (tid < nc) {
            int idx[5] = {0, 1, 2, 3, 4};
            if (S.start[tid] >= 0) draw5(S.r, S.start[tid], idx);
            double q1[5][2], q2[5][2];
            for (int i = 0; i < 5; i++) {
                const double4 p = pts[idx[i]];
                q1[i][0] = p.x; q1[i][1] = p.y; q2[i][0] = p.z; q2[i][1] = p.w;
            }
            double out[90];
            const int nm = solve_e5(q1, q2, out);
            S.nmodels[tid] = nm;
            for (int i = 0; i < 9 * nm; i++) S.models[tid][i / 9][i % 9] = out[i];
        }
        __syncthreads();
        for (int hm = wave; hm < nc * 10; hm += ER_THREADS / 64) {
            const int h = hm / 10, m = hm - 10 * h;
            if (m >= S.nmodels[h]) continue;
            double e[9];
            for (int i = 0; i < 9; i++) e[i] = S.models[h][m][i];
            int cnt = 0;
            for (int i = lane; i < n; i += 64) {
                const double4 p = pts[i];
                cnt += e_error(p.x, p.y, p.z, p.w, e) <= thr2;
            }
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (lane == 0) S.counts[h][m] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            for (int h = 0; h < nc; h++) {
                for (int m = 0; m < S.nmodels[h]; m++) {
                    const int good = S.counts[h][m];
                    if (good > max(S.max_good, 4)) {
                        for (int i = 0; i < 9; i++) S.best[i] = S.models[h][m][i];
                        S.max_good = good;
                        S.niters = e_update_num_iters(prob, (double)(n - good) / n, 5, S.niters);
                    }
                }
                S.iter++;
                if (S.iter >= S.niters) break;
            }
            if (S.iter >= S.niters) S.done = 1;
        }
        __syncthreads();
        if (S.done) break;
    }
    const bool found = S.max_good > 0;
    int n_inl = 0;
    double e[9];
    for (int i = 0; i < 9; i++) e[i] = S.best[i];
    for (int i = tid; i < n; i += ER_THREADS) {
        bool inl = false;
        if (found) {
            const double4 p = pts[i];
            inl = e_error(p.x, p.y, p.z, p.w, e) <= thr2;
        }
        mask[i] = (uint8_t)inl;
        n_inl += inl;
    }
    for (int off = 32; off > 0; off >>= 1) n_inl += __shfl_xor(n_inl, off);
    if (lane == 0) S.wave_tot[wave] = n_inl;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < ER_THREADS / 64; w++) tot += S.wave_tot[w];
        info[0] = found ? 1 : 0;
        info[1] = S.iter;
        info[2] = tot;
        for (int i = 0; i < 9; i++) Eout[i] = found ? S.best[i] : 0.0;
    }
}

// ---- pose from an essential matrix --------------------------------------------

// cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (row-major):
// A becomes diagonal, column j of V the eigenvector of A[j][j]
template <int N>
__device__ __forceinline__ void jacobi_sym(double (&A)[N * N], double (&V)[N * N])
{
#pragma unroll
    for (int i = 0; i < N * N; i++) V[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
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
                const double theta = (A[q * N + q] - A[p * N + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), sn = t * c;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - sn * akq;
                    A[k * N + q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - sn * aqk;
                    A[q * N + k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - sn * vkq;
                    V[k * N + q] = sn * vkp + c * vkq;
                }
            }
    }
}

// cv::decomposeEssentialMat: R1 = U W Vt, R2 = U W^T Vt, t = u2, with the SVD
// taken from the eigen-decomposition of E^T E (u2 = u0 x u1, Vt negated when
// its determinant is negative)
__device__ void decompose_essential(const double *E, double *R1, double *R2, double *t)
{
    double A[9], V[9], U[9], Vt[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += E[k * 3 + i] * E[k * 3 + j];
            A[i * 3 + j] = s;
        }
    jacobi_sym<3>(A, V);
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
                const double x = ev[i]; ev[i] = ev[j]; ev[j] = x;
#pragma unroll
                for (int k = 0; k < 3; k++) { const double u = vc[i][k]; vc[i][k] = vc[j][k]; vc[j][k] = u; }
            }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) Vt[i * 3 + k] = vc[i][k];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        double u[3], nrm = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            u[i] = E[i * 3 + 0] * Vt[c * 3 + 0] + E[i * 3 + 1] * Vt[c * 3 + 1] + E[i * 3 + 2] * Vt[c * 3 + 2];
            nrm += u[i] * u[i];
        }
        nrm = sqrt(nrm);
#pragma unroll
        for (int i = 0; i < 3; i++) U[i * 3 + c] = u[i] / nrm;
    }
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
    const double det = Vt[0] * (Vt[4] * Vt[8] - Vt[5] * Vt[7]) - Vt[1] * (Vt[3] * Vt[8] - Vt[5] * Vt[6]) +
                       Vt[2] * (Vt[3] * Vt[7] - Vt[4] * Vt[6]);
    if (det < 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) Vt[i] = -Vt[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double a = U[i * 3 + 1] * Vt[0 * 3 + j], b = U[i * 3 + 0] * Vt[1 * 3 + j],
                         c = U[i * 3 + 2] * Vt[2 * 3 + j];
            R1[i * 3 + j] = -a + b + c;
            R2[i * 3 + j] = a - b + c;
        }
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = U[i * 3 + 2];
}

// cv::triangulatePoints for one correspondence between [I | 0] and [R | t]
// followed by the cheirality test of cv::recoverPose
__device__ __forceinline__ bool triangulate_vote(const double *R, const double *t, double x1, double y1, double x2,
                                                 double y2, double dist)
{
    double A[4][4] = {{-1, 0, x1, 0}, {0, -1, y1, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        A[2][k] = x2 * R[6 + k] - R[k];
        A[3][k] = y2 * R[6 + k] - R[3 + k];
    }
    A[2][3] = x2 * t[2] - t[0];
    A[3][3] = y2 * t[2] - t[1];
    double M[16], V[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) s += A[k][i] * A[k][j];
            M[i * 4 + j] = s;
        }
    jacobi_sym<4>(M, V);
    double best = M[0];
    double Q[4] = {V[0], V[4], V[8], V[12]};
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (M[i * 5] < best) {
            best = M[i * 5];
#pragma unroll
            for (int k = 0; k < 4; k++) Q[k] = V[k * 4 + i];
        }
    bool ok = Q[2] * Q[3] > 0;
    const double X = Q[0] / Q[3], Y = Q[1] / Q[3], Z = Q[2] / Q[3];
    ok = ok && Z < dist;
    const double z2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    ok = ok && z2 > 0 && z2 < dist;
    return ok;
}

constexpr int RP_THREADS = 256;

// cv::recoverPose for one problem: q = n x (x1, y1, x2, y2) normalised points,
// mask is read when use_mask and rewritten with the winner's votes, votes is an
// n-byte work array.  out = R (9, row-major) then t (3); info = {votes of the
// winner, winner index in the order (R1,t) (R2,t) (R1,-t) (R2,-t)}.
__global__ __launch_bounds__(RP_THREADS) void k_recover_pose(const double *__restrict__ E, const double *__restrict__ q,
                                                            int n, double dist, int use_mask,
                                                            uint8_t *__restrict__ mask, uint8_t *__restrict__ votes,
                                                            double *__restrict__ out, int *__restrict__ info)
{
    __shared__ double s_R[2][9], s_t[2][3];
    __shared__ int s_good[4], s_best;
    const int tid = threadIdx.x;
    if (tid == 0) {
        double e[9], R1[9], R2[9], t[3];
        for (int i = 0; i < 9; i++) e[i] = E[i];
        decompose_essential(e, R1, R2, t);
        for (int i = 0; i < 9; i++) { s_R[0][i] = R1[i]; s_R[1][i] = R2[i]; }
        for (int i = 0; i < 3; i++) { s_t[0][i] = t[i]; s_t[1][i] = -t[i]; }
    }
    if (tid < 4) s_good[tid] = 0;
    __syncthreads();
    int cnt[4] = {0, 0, 0, 0};
    const double4 *pts = reinterpret_cast<const double4 *>(q);
    for (int i = tid; i < n; i += RP_THREADS) {
        const double4 p = pts[i];
        const bool live = !use_mask || mask[i] != 0;
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            double R[9], t[3];
            for (int j = 0; j < 9; j++) R[j] = s_R[k & 1][j];
            for (int j = 0; j < 3; j++) t[j] = s_t[k >> 1][j];
            const bool ok = triangulate_vote(R, t, p.x, p.y, p.z, p.w, dist) && live;
            v |= (uint32_t)ok << k;
            cnt[k] += ok;
        }
        votes[i] = (uint8_t)v;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int c = cnt[k];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if ((tid & 63) == 0) atomicAdd(&s_good[k], c);
    }
    __syncthreads();
    if (tid == 0) {
        const int g0 = s_good[0], g1 = s_good[1], g2 = s_good[2], g3 = s_good[3];
        int best = 3;
        if (g0 >= g1 && g0 >= g2 && g0 >= g3) best = 0;
        else if (g1 >= g0 && g1 >= g2 && g1 >= g3) best = 1;
        else if (g2 >= g0 && g2 >= g1 && g2 >= g3) best = 2;
        s_best = best;
        for (int i = 0; i < 9; i++) out[i] = s_R[best & 1][i];
        for (int i = 0; i < 3; i++) out[9 + i] = s_t[best >> 1][i];
        info[0] = s_good[best];
        info[1] = best;
    }
    __syncthreads();
    const int best = s_best;
    for (int i = tid; i < n; i += RP_THREADS) mask[i] = (votes[i] >> best) & 1;
}

}  // namespace

void launch_e_hypotheses(const double *q, int n, const int *idx, int H, double *E, int *n_models, float thr2,
                         int *counts, hipStream_t s)
{
    hipLaunchKernelGGL(k_e_hypotheses, dim3((H + EH_THREADS - 1) / EH_THREADS), dim3(EH_THREADS), 0, s, q, n, idx, H,
                       E, n_models);
    hipLaunchKernelGGL(k_e_score, dim3((H * 10 + 3) / 4), dim3(256), 0, s, q, n, E, n_models, H, thr2, counts);
}

void launch_e_ransac(const double *q, int n, const uint32_t *rng_raw, int rng_len, double prob, float thr2,
                     int max_iters, uint8_t *mask, double *E, int *info, hipStream_t s)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_e_ransac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(ErShared) + 64);
    hipLaunchKernelGGL(k_e_ransac, dim3(1), dim3(ER_THREADS), sizeof(ErShared) + 16, s, q, n, rng_raw, rng_len, prob,
                       thr2, max_iters, mask, E, info);
}

void launch_recover_pose(const double *E, const double *q, int n, double dist, int use_mask, uint8_t *mask,
                         uint8_t *votes, double *out, int *info, hipStream_t s)
{
    hipLaunchKernelGGL(k_recover_pose, dim3(1), dim3(RP_THREADS), 0, s, E, q, n, dist, use_mask, mask, votes, out,
                       info);
}

}  // namespace vo
