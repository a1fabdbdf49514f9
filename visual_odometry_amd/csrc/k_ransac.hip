// Fundamental-matrix RANSAC (7-point) for gfx950, one workgroup per frame
// pair, replaying cv::RNG's stream so that the hypothesis sequence and the
// adaptive iteration count are those of
//   cv::findFundamentalMat(prev, curr, FM_RANSAC, 3.0, 0.99, mask)
// (reference src/feature_matcher.cpp:52-58), followed by the reference's
// 100 px motion filter (reference src/feature_matcher.cpp:72-88).
//
// Hypotheses are generated and scored in rounds of FR_CHUNK iterations:
//   1. all threads reduce a window of raw RNG words modulo n;
//   2. every window offset is simulated as the start of one subset draw
//      (7 distinct indices, redraw on duplicates, collinearity rejection);
//   3. one lane chains the offsets into the iteration sequence;
//   4. one lane per iteration solves the 7-point problem in fp64;
//   5. all waves count inliers of every candidate matrix (fp64 error,
//      float32 compare, as OpenCV does);
//   6. one lane replays the sequential "best so far" logic and the adaptive
//      iteration bound.
// Shortcuts that leave every result as it is: the collinearity test of step 2
// runs only for the offsets the chain of step 3 actually visits; step 5
// decides "error <= threshold" without the two divisions wherever the margin
// allows and drops a candidate as soon as it cannot beat the best count of
// the earlier rounds any more; step 6 finds the accepted candidates as a
// prefix maximum over the round and walks only those.
// fp64 expressions are evaluated in a fixed order with contraction disabled.
#include <float.h>
#include <math.h>

#include "vo_internal.h"

namespace vo {

constexpr int FR_THREADS = 512;
constexpr int FR_CHUNK = 64;
constexpr int FR_FIRST = 8;  // iterations of the first round
constexpr int FR_WIN = FR_CHUNK * 7 + 192;
constexpr int FR_MAX_ATTEMPTS = 10000;
constexpr int FR_VISIT = 2 * kWave;  // subset draws examined per round (usable or not), one lane each

struct FrShared {
    double models[FR_CHUNK][3][9];
    double best[9];
    double min_median;  // LMedS: smallest median error so far
    int nmodels[FR_CHUNK];
    int counts[FR_CHUNK][3];
    int start[FR_CHUNK];
    uint16_t r[FR_WIN];
    uint8_t cons[FR_WIN];
    int wave_tot[FR_THREADS / kWave];
    int nchunk, nvisit, kend;
    uint16_t visit[FR_VISIT];        // window offsets the chain visits; [nvisit] is where it goes on
    uint16_t jump[2][FR_WIN + 2];    // offset reached from each window offset after 2^level subset draws
    unsigned long long vmask[2];     // which of the visited offsets can be drawn from / hold a usable subset
    double acc_ld[FR_CHUNK][3];      // log(1 - (1 - outlier ratio)^7) of the accepted candidates
    unsigned long long acc_mask[3];  // accepted candidates of the round, per solution index
    int iter, niters, max_good, done, failed;
    int attempts;        // attempts already spent on the iteration being drawn
    long long pos;       // absolute index of the next unread RNG word
    int scan_base;
};

__device__ __forceinline__ bool collinear_last(const float (&m)[7][2])
{
    // OpenCV haveCollinearPoints: last point against every earlier pair
    const int i = 6;
    for (int j = 0; j < i; j++) {
        const double dx1 = m[j][0] - m[i][0], dy1 = m[j][1] - m[i][1];
        for (int k = 0; k < j; k++) {
            const double dx2 = m[k][0] - m[i][0], dy2 = m[k][1] - m[i][1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

// draw one 7-subset starting at window offset o; returns draws consumed or 0
// when the window ends first
__device__ __forceinline__ int draw_subset(const uint16_t *r, int o, int (&idx)[7], int win = FR_WIN)
{
    int p = o;
    for (int i = 0; i < 7; i++) {
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

__device__ int solve_cubic(const double (&c)[4], double (&r)[3])
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
                const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = acos(R / sqrt(Qcubed));
            const double sqrtQ = sqrt(Q);
            const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.14159265358979323846 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.14159265358979323846 / 3)) - t2;
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

// 7-point solver (OpenCV run7Point semantics with Hartley normalisation); the
// null-space basis comes from a Householder QR of A^T.  Returns 0..3 models.
__device__ int solve_f7(const float (&m1)[7][2], const float (&m2)[7][2], double (*Fout)[9])
{
    double m1cx = 0, m1cy = 0, m2cx = 0, m2cy = 0, scale1 = 0, scale2 = 0;
    const double t = 1. / 7;
    for (int i = 0; i < 7; i++) {
        m1cx += m1[i][0]; m1cy += m1[i][1];
        m2cx += m2[i][0]; m2cy += m2[i][1];
    }
    m1cx *= t; m1cy *= t; m2cx *= t; m2cy *= t;
    for (int i = 0; i < 7; i++) {
        double dx = m1[i][0] - m1cx, dy = m1[i][1] - m1cy;
        scale1 += sqrt(dx * dx + dy * dy);
        dx = m2[i][0] - m2cx; dy = m2[i][1] - m2cy;
        scale2 += sqrt(dx * dx + dy * dy);
    }
    scale1 *= t; scale2 *= t;
    if (scale1 < FLT_EPSILON || scale2 < FLT_EPSILON) return 0;
    scale1 = sqrt(2.) / scale1;
    scale2 = sqrt(2.) / scale2;

    double M[9][7];  // A^T
    for (int i = 0; i < 7; i++) {
        const double x0 = (m1[i][0] - m1cx) * scale1, y0 = (m1[i][1] - m1cy) * scale1;
        const double x1 = (m2[i][0] - m2cx) * scale2, y1 = (m2[i][1] - m2cy) * scale2;
        M[0][i] = x1 * x0; M[1][i] = x1 * y0; M[2][i] = x1;
        M[3][i] = y1 * x0; M[4][i] = y1 * y0; M[5][i] = y1;
        M[6][i] = x0; M[7][i] = y0; M[8][i] = 1;
    }
    double V[7][9], beta[7];
    for (int k = 0; k < 7; k++) {
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
        for (int j = k; j < 7; j++) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * M[i][j];
            dot *= beta[k];
            for (int i = k; i < 9; i++) M[i][j] -= dot * V[k][i];
        }
    }
    double f1[9], f2[9];
    for (int c = 0; c < 2; c++) {
        double q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        q[7 + c] = 1.0;
        for (int k = 6; k >= 0; k--) {
            double dot = 0;
            for (int i = k; i < 9; i++) dot += V[k][i] * q[i];
            dot *= beta[k];
            for (int i = k; i < 9; i++) q[i] -= dot * V[k][i];
        }
        for (int i = 0; i < 9; i++) (c == 0 ? f1 : f2)[i] = q[i];
    }

    double c[4], r[3];
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

    const int n = solve_cubic(c, r);
    if (n < 1 || n > 3) return n < 0 ? 0 : n;

    const double T1[9] = {scale1, 0, -scale1 * m1cx, 0, scale1, -scale1 * m1cy, 0, 0, 1};
    const double T2[9] = {scale2, 0, -scale2 * m2cx, 0, scale2, -scale2 * m2cy, 0, 0, 1};
    for (int k = 0; k < n; k++) {
        double Fn[9], tmp[9], F[9];
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; Fn[8] = 1.; }
        else Fn[8] = 0.;
        for (int i = 0; i < 8; i++) Fn[i] = f1[i] * lambda + f2[i] * mu;
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
            const double inv = 1. / F[8];
            for (int i = 0; i < 9; i++) F[i] *= inv;
        }
        for (int i = 0; i < 9; i++) Fout[k][i] = F[i];
    }
    return n;
}

// larger of the two squared point-to-epipolar-line distances, as float
__device__ __forceinline__ float f_error(const float4 p, const double *F)
{
    const double x1 = p.x, y1 = p.y, x2 = p.z, y2 = p.w;
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// (float)max(e1, e2) <= thr of f_error() without its divisions: d^2 / q <= thr is decided as d^2 <= thr * q
// with a relative margin of 1e-6 on either side (the float rounding of the error moves the threshold by 6e-8,
// fp64 rounding by 1e-15); what falls inside the margin, and degenerate lines (q = 0, overflow, NaN), is left
// undecided for the exact expression.  Returns 1 inlier, 0 outlier, -1 undecided.
__device__ __forceinline__ int f_inlier_fast(const float4 p, const double *F, double thr_lo, double thr_hi)
{
    // (fused multiply-adds: this estimate need not round like f_error, the margin absorbs the difference)
    const double x1 = p.x, y1 = p.y, x2 = p.z, y2 = p.w;
    double a = fma(F[0], x1, fma(F[1], y1, F[2]));
    double b = fma(F[3], x1, fma(F[4], y1, F[5]));
    double c = fma(F[6], x1, fma(F[7], y1, F[8]));
    const double q2 = fma(a, a, b * b);
    const double d2 = fma(x2, a, fma(y2, b, c));
    a = fma(F[0], x2, fma(F[3], y2, F[6]));
    b = fma(F[1], x2, fma(F[4], y2, F[7]));
    c = fma(F[2], x2, fma(F[5], y2, F[8]));
    const double q1 = fma(a, a, b * b);
    const double d1 = fma(x1, a, fma(y1, b, c));
    const double dd1 = d1 * d1, dd2 = d2 * d2;
    const bool sane = q1 > 0 && q2 > 0 && q1 + q2 < 1e300 && dd1 + dd2 < 1e300;
    const bool in = dd1 <= thr_lo * q1 && dd2 <= thr_lo * q2;
    const bool out = dd1 > thr_hi * q1 || dd2 > thr_hi * q2;
    return !sane ? -1 : (in ? 1 : (out ? 0 : -1));
}

// cv::RANSACUpdateNumIters(p, ep, model_points, max_iters) in two steps, so that the expensive part (a power and
// a logarithm per candidate) can be evaluated by many lanes at once and the cheap part replayed in sequence:
//   ransac_log_outlier_term: log(1 - (1 - ep)^model_points), or +infinity where OpenCV returns 0 iterations;
//   ransac_bound: the new bound from log(1 - p), that term and the current bound.
__device__ __forceinline__ double ransac_log_confidence(double p)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    const double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    return log(num);
}
__device__ __forceinline__ double ransac_log_outlier_term(double ep, int model_points)
{
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    const double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return INFINITY;
    return log(denom);
}
__device__ __forceinline__ int ransac_bound(double log_num, double log_denom, int max_iters)
{
    if (log_denom == INFINITY) return 0;
    return log_denom >= 0 || -log_num >= max_iters * (-log_denom) ? max_iters : (int)rint(log_num / log_denom);
}
__device__ int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    return ransac_bound(ransac_log_confidence(p), ransac_log_outlier_term(ep, model_points), max_iters);
}

#ifdef VO_FE_STAMPS
// Diagnostic build only: cycle stamps of thread 0 of the first 256 blocks
// (the stamps of the RANSAC loop are those of the last round).
__device__ unsigned long long g_fr_stamps[256 * 16];
#define FR_STAMP(i)                                                                      \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 256)                                        \
            g_fr_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)
extern "C" int vo_debug_fr_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fr_stamps), sizeof(unsigned long long) * 256 * 16);
}
#else
#define FR_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(FR_THREADS) void k_f_ransac(const vo_match *__restrict__ good,
                                                         const int *__restrict__ n_good, int kp_stride,
                                                         const uint32_t *__restrict__ rng_raw, int rng_len,
                                                         FRansacParams prm, uint8_t *__restrict__ mask,
                                                         double *__restrict__ Fout, int *__restrict__ info,
                                                         vo_match *__restrict__ final_m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    FrShared &S = *reinterpret_cast<FrShared *>(s_dyn);
    float4 *pts = reinterpret_cast<float4 *>(s_dyn + ((sizeof(FrShared) + 15) & ~size_t(15)));

    FR_STAMP(0);
    const int pair = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(n_good[pair], kp_stride);
    const vo_match *g = good + (size_t)pair * kp_stride;
    uint8_t *msk = mask + (size_t)pair * kp_stride;
    int *inf = info + pair * 4;

    for (int i = tid; i < n; i += FR_THREADS) {
        pts[i] = make_float4(g[i].px, g[i].py, g[i].cx, g[i].cy);
        msk[i] = 0;
    }
    if (n < 8) {  // the reference does not call findFundamentalMat below 8 points
        if (tid == 0) { inf[0] = 0; inf[1] = 0; inf[2] = 0; inf[3] = 0; }
        return;
    }
    // Below 15 points cv::findFundamentalMat(FM_RANSAC) runs the LMedS estimator
    // instead: a fixed iteration count (45 % outliers assumed, at most 1000), the
    // model with the smallest median error wins, the inlier threshold comes from
    // that median.  Subset drawing and the 7-point solver are shared.
    const bool lmeds = n < 15;
    if (tid == 0) {
        S.iter = 0;
        S.niters = lmeds ? max(ransac_update_num_iters(prm.confidence, 0.45, 7, 1000), 3)
                         : (prm.max_iters > 1 ? prm.max_iters : 1);
        S.max_good = 0; S.done = 0; S.failed = 0; S.attempts = 0; S.pos = 0;
        S.min_median = DBL_MAX;
    }
    __syncthreads();

    const float thr2 = (float)(prm.threshold * prm.threshold);
    FR_STAMP(1);

    const double thr_lo = (double)thr2 * (1.0 - 1e-6), thr_hi = (double)thr2 * (1.0 + 1e-6);
    while (true) {
        FR_STAMP(2);
        // The first round is short: on a typical frame pair the first model or two already explain most
        // matches and the adaptive iteration count drops below it; later rounds take what is still needed.
        const int want = min(S.iter == 0 && !lmeds ? FR_FIRST : FR_CHUNK, S.niters - S.iter);
        const int base_good = S.max_good;  // best count of the earlier rounds
        // the window holds the draws of `want` subsets plus room for repeated indices and rejected subsets
        const int win = min(FR_WIN, want * 7 + 192);
        // 1. window of draws
        const long long pos = S.pos;
        for (int k = tid; k < win; k += FR_THREADS) {
            const long long a = pos + k;
            S.r[k] = a < rng_len ? (uint16_t)(rng_raw[a] % (uint32_t)n) : (uint16_t)0;
        }
        __syncthreads();
        FR_STAMP(3);
        // 2. one subset draw per window offset: draws consumed (0: the window or the RNG table ends first)
        for (int o = tid; o < win; o += FR_THREADS) {
            int idx[7];
            int cons = draw_subset(S.r, o, idx, win);
            if (pos + o + cons > rng_len) cons = 0;
            S.cons[o] = (uint8_t)cons;
        }
        __syncthreads();
        FR_STAMP(4);
        // 3. chain offsets into iterations.  Which offsets the chain visits does not depend on the subsets drawn
        // there (a rejected subset consumes its draws like a usable one), only which of them count as iterations.
        // The k-th visited offset is the k-fold image of offset 0 under o -> o + cons[o]: lane k composes it from
        // the powers of two of that map (pointer doubling, seven levels), then tests the subset drawn there (OpenCV
        // rejects one whose last point is collinear with two earlier ones) and takes its rank among the usable ones.
        {
            const int room = min(FR_VISIT - 1, want + 16);  // (more than 16 rejections shorten the round, nothing else)
            for (int o = tid; o <= win; o += FR_THREADS) {
                const int c = o < win ? S.cons[o] : 0;
                S.jump[0][o] = (uint16_t)(c ? o + c : win);  // win: the chain cannot go on from here
            }
            __syncthreads();
            int at = 0;
            // (only the first `room` visited offsets are used: a short round needs fewer levels)
            const int levels = 32 - __builtin_clz(room);
#pragma unroll 1
            for (int lv = 0; lv < levels; lv++) {
                const uint16_t *J = S.jump[lv & 1];
                uint16_t *Jn = S.jump[(lv + 1) & 1];
                if (tid < FR_VISIT && ((tid >> lv) & 1)) at = J[at];
                if (lv < levels - 1)
                    for (int o = tid; o <= win; o += FR_THREADS) Jn[o] = J[J[o]];
                __syncthreads();
            }
            // (a stuck offset maps to `win`; landing on win exactly is the window's end: both stop the chain)
            bool usable = false, drawable = false;
            if (tid < FR_VISIT) {
                drawable = tid < room && at < win && S.cons[at] != 0;
                S.visit[tid] = (uint16_t)at;
                const unsigned long long bm = __ballot(drawable);
                if (lane == 0) S.vmask[wave] = bm;
            }
            __syncthreads();
            // visited offsets: the drawable ones before the first that is not
            const unsigned long long d0 = S.vmask[0], d1 = S.vmask[1];
            const int nv = ~d0 ? __builtin_ctzll(~d0) : (~d1 ? 64 + __builtin_ctzll(~d1) : 128);
            if (tid < FR_VISIT) {
                if (tid < nv) {
                    int idx[7];
                    draw_subset(S.r, at, idx, win);
                    float a[7][2], b[7][2];
                    for (int i = 0; i < 7; i++) {
                        const float4 p = pts[idx[i]];
                        a[i][0] = p.x; a[i][1] = p.y; b[i][0] = p.z; b[i][1] = p.w;
                    }
                    usable = !collinear_last(a) && !collinear_last(b);
                }
            }
            __syncthreads();  // (everybody has read the drawable masks)
            if (tid < FR_VISIT) {
                const unsigned long long bm = __ballot(usable);
                if (lane == 0) S.vmask[wave] = bm;
            }
            if (tid == 0) S.kend = nv;
            __syncthreads();
            const unsigned long long u0 = S.vmask[0], u1 = S.vmask[1];
            const int total = __popcll(u0) + __popcll(u1);
            if (tid < FR_VISIT && usable) {
                const int rank = tid < 64 ? __popcll(u0 & ((1ull << lane) - 1ull))
                                          : __popcll(u0) + __popcll(u1 & ((1ull << lane) - 1ull));
                if (rank < want) S.start[rank] = (uint16_t)at;
                if (rank == want - 1) S.kend = tid + 1;  // the round ends behind the want-th usable subset
            }
            __syncthreads();
            if (tid == 0) {
                const int kend = S.kend;  // visited offsets this round consumes
                // attempts pending = rejected subsets since the last usable one
                const unsigned long long m0 = kend >= 64 ? u0 : (u0 & ((1ull << kend) - 1ull));
                const unsigned long long m1 = kend >= 128 ? u1 : (kend > 64 ? (u1 & ((1ull << (kend - 64)) - 1ull)) : 0ull);
                int att;
                if (m1) att = kend - 1 - (127 - __builtin_clzll(m1));
                else if (m0) att = kend - 1 - (63 - __builtin_clzll(m0));
                else att = S.attempts + kend;
                int nc = min(total, want);
                // cv's getSubset gives up after FR_MAX_ATTEMPTS rejections in a row: only reachable with rejections
                // carried over from earlier rounds, before the first usable subset of this one
                const int first = m0 ? __builtin_ctzll(m0) : (m1 ? 64 + __builtin_ctzll(m1) : kend);
                if (S.attempts + first >= FR_MAX_ATTEMPTS) {
                    S.failed = 1;
                    nc = 0;
                    att = FR_MAX_ATTEMPTS;
                }
                S.nchunk = nc;
                S.attempts = att;
                S.pos = pos + S.visit[kend];
                if (nv == 0) S.failed = 1;  // RNG table exhausted: cannot progress
            }
            __syncthreads();
        }
        FR_STAMP(5);
        const int nc = S.nchunk;
        // 4. solve
        if (tid < nc) {
            int idx[7];
            draw_subset(S.r, S.start[tid], idx, win);
            float a[7][2], b[7][2];
            for (int i = 0; i < 7; i++) {
                const float4 p = pts[idx[i]];
                a[i][0] = p.x; a[i][1] = p.y; b[i][0] = p.z; b[i][1] = p.w;
            }
            S.nmodels[tid] = solve_f7(a, b, S.models[tid]);
        }
        __syncthreads();
        FR_STAMP(6);
        // 5. score every candidate matrix: one wave per (iteration, model)
        for (int hm = wave; hm < nc * 3; hm += FR_THREADS / kWave) {
            const int h = hm / 3, m = hm - 3 * h;
            if (m >= S.nmodels[h]) continue;
            double F[9];
            for (int i = 0; i < 9; i++) F[i] = S.models[h][m][i];
            if (lmeds) {
                // median = element n/2 of the sorted errors: the lane whose error has that rank
                const float e = lane < n ? f_error(pts[lane], F) : 0.f;
                int rank = 0;
                for (int j = 0; j < n; j++) {
                    const float ej = __shfl(e, j);
                    rank += (ej < e) || (ej == e && j < lane);
                }
                if (lane < n && rank == n / 2) S.counts[h][m] = __float_as_int(e);
                continue;
            }
            // a candidate is accepted only with more inliers than max(best so far, 6): once the points left
            // cannot lift it above the best of the earlier rounds its count no longer matters
            const int need = max(base_good, 6);
            int cnt = 0;
            // four points per lane and step, so that the loads and the arithmetic of one hide the latencies of another
            for (int i0 = 0; i0 < n; i0 += 4 * kWave) {
                int in[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * kWave + lane;
                    const int r = f_inlier_fast(pts[min(i, n - 1)], F, thr_lo, thr_hi);
                    in[u] = i < n ? r : 0;
                }
                if (__any((in[0] | in[1] | in[2] | in[3]) < 0)) {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (in[u] < 0) in[u] = f_error(pts[i0 + u * kWave + lane], F) <= thr2;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) cnt += __popcll(__ballot(in[u] > 0));
                if (cnt + max(n - i0 - 4 * kWave, 0) <= need) break;
            }
            if (lane == 0) S.counts[h][m] = cnt;
        }
        __syncthreads();
        FR_STAMP(7);
        // 6. sequential acceptance
        if (lmeds) {
            if (tid == 0) {
                for (int h = 0; h < nc; h++) {
                    const int nm = S.nmodels[h];
                    for (int m = 0; m < nm; m++) {
                        const double median = (double)__int_as_float(S.counts[h][m]);
                        if (median < S.min_median) {
                            S.min_median = median;
                            for (int i = 0; i < 9; i++) S.best[i] = S.models[h][m][i];
                        }
                    }
                    S.iter++;
                    if (S.iter >= S.niters) break;
                }
                if (S.iter >= S.niters) S.done = 1;
                if (S.failed) S.done = 1;
            }
        } else if (wave == 0) {
            // candidate (h, m) is accepted when its count beats max(best so far, 6) and every earlier candidate of
            // the round: lane h holds the counts of iteration h, a prefix maximum over the lanes finds the accepted
            // ones, lane 0 walks them in order, moves the iteration bound and finds where the loop stops
            int c[3];
            const int nm = lane < nc ? S.nmodels[lane] : 0;
#pragma unroll
            for (int m = 0; m < 3; m++) c[m] = m < nm ? S.counts[lane][m] : -1;
            const int mine = max(c[0], max(c[1], c[2]));
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off) incl = max(incl, y);
            }
            int run = __shfl_up(incl, 1);
            if (lane == 0) run = -1;
            run = max(run, max(base_good, 6));
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const bool acc = c[m] > run;
                run = max(run, c[m]);
                const unsigned long long bm = __ballot(acc);
                if (lane == 0) S.acc_mask[m] = bm;
                // the costly half of the new iteration bound, for every accepted candidate at once
                if (acc) S.acc_ld[lane][m] = ransac_log_outlier_term((double)(n - c[m]) / n, 7);
            }
            if (lane == 0) {
                const double log_conf = ransac_log_confidence(prm.confidence);
                unsigned long long any = S.acc_mask[0] | S.acc_mask[1] | S.acc_mask[2];
                const int it0 = S.iter;
                int N = S.niters, prev = 0, final_iter = it0 + nc;
                bool stop = false;
                // between two accepted candidates the bound N is constant: the loop stops at the first iteration
                // j of the stretch with it0 + j + 1 >= N
                while (any) {
                    const int h = __builtin_ctzll(any);
                    any &= any - 1;
                    const int j = max(prev, N - it0 - 1);
                    if (j <= h - 1) { stop = true; final_iter = it0 + j + 1; break; }
                    for (int m = 0; m < 3; m++)
                        if ((S.acc_mask[m] >> h) & 1ull) {
                            const int goodc = S.counts[h][m];
                            for (int i = 0; i < 9; i++) S.best[i] = S.models[h][m][i];
                            S.max_good = goodc;
                            N = ransac_bound(log_conf, S.acc_ld[h][m], N);
                        }
                    prev = h;
                }
                if (!stop) {
                    const int j = max(prev, N - it0 - 1);
                    if (j <= nc - 1) final_iter = it0 + j + 1;
                }
                S.iter = final_iter;
                S.niters = N;
                if (S.iter >= S.niters) S.done = 1;
                if (S.failed) S.done = 1;  // getSubset gave up: stop (iter == 0 -> nothing found)
            }
        }
        __syncthreads();
        FR_STAMP(8);
        if (S.done) break;
    }

    // final mask, then the motion filter with ordered compaction
    // a model exists: RANSAC accepted one / LMedS saw a finite median
    const bool found = lmeds ? S.min_median < DBL_MAX : S.max_good > 0;
    float thr_final = thr2;
    if (lmeds && found) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(S.min_median);
        sigma = sigma > 0.001 ? sigma : 0.001;
        thr_final = (float)(sigma * sigma);
    }
    if (tid == 0) S.scan_base = 0;
    __syncthreads();
    double F[9];
    for (int i = 0; i < 9; i++) F[i] = S.best[i];
    int n_inl = 0;
    vo_match *fm = final_m + (size_t)pair * kp_stride;
    for (int i0 = 0; i0 < n; i0 += FR_THREADS) {
        const int i = i0 + tid;
        bool inl = false, keep = false;
        if (i < n && found) {
            inl = f_error(pts[i], F) <= thr_final;
            msk[i] = (uint8_t)inl;
            const float dx = pts[i].z - pts[i].x, dy = pts[i].w - pts[i].y;
            keep = inl && sqrtf(dx * dx + dy * dy) < 100.0f;
        }
        const uint64_t bm = __ballot(keep);
        n_inl += __popcll(__ballot(inl));
        if (lane == 0) S.wave_tot[wave] = __popcll(bm);
        __syncthreads();
        int off = S.scan_base;
        for (int w = 0; w < wave; w++) off += S.wave_tot[w];
        if (keep) fm[off + __popcll(bm & ((1ull << lane) - 1ull))] = g[i];
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < FR_THREADS / kWave; w++) tot += S.wave_tot[w];
            S.scan_base += tot;
        }
        __syncthreads();
    }
    // per-wave inlier totals -> block total
    if (lane == 0) S.wave_tot[wave] = n_inl;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < FR_THREADS / kWave; w++) tot += S.wave_tot[w];
        inf[0] = (lmeds ? (found && tot >= 7) : found) ? 1 : 0;  // LMedS reports failure below 7 inliers
        inf[1] = S.iter;
        inf[2] = tot;
        inf[3] = S.scan_base;
        double *Fo = Fout + (size_t)pair * 9;
        for (int i = 0; i < 9; i++) Fo[i] = found ? S.best[i] : 0.0;
    }
    FR_STAMP(9);
}

void launch_f_ransac(const vo_match *good, const int *n_good, int kp_stride, const uint32_t *rng_raw,
                     int rng_len, FRansacParams prm, uint8_t *mask, double *F, int *info, vo_match *final_m,
                     int pairs, hipStream_t s)
{
    const size_t lds = ((sizeof(FrShared) + 15) & ~size_t(15)) + sizeof(float4) * (size_t)kp_stride;
    // per launch: the attribute belongs to the current device's copy of the kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_f_ransac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    hipLaunchKernelGGL(k_f_ransac, dim3(pairs), dim3(FR_THREADS), lds, s, good, n_good, kp_stride, rng_raw,
                       rng_len, prm, mask, F, info, final_m);
}

}  // namespace vo
