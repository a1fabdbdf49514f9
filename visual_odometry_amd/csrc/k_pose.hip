// Pose stage for gfx950: track-age bookkeeping + multi-view filter, depth
// back-projection, and cv::solvePnPRansac semantics (EPnP hypotheses from
// 5-point samples drawn with cv::RNG's stream, reprojection scoring, adaptive
// iteration bound, Levenberg-Marquardt refinement on the inliers).
// reference src/frame_processor.cpp:44-92 (ages), :230-308 (back-projection),
// :176-196 (solvePnPRansac + Rodrigues).
//
// fp64 throughout, evaluated with contraction disabled.  One workgroup per
// frame; hypotheses are solved one per lane, points are scored by all waves.
#include <float.h>
#include <math.h>

#include "vo_internal.h"

namespace vo {

// =====================================================================
// track ages (sequential in time within a stream, one block per stream)
// =====================================================================

constexpr int TA_THREADS = 1024;

// For frame f of stream s (batch index b = s*frames + f):
//   age[curr][train] = age[prev][query] + 1 for every match, later matches
//   overwriting earlier ones, all other ages 0; flag[b] = 1 when more than 80
//   matches exist and at least 50 of them have age >= 3.
__global__ __launch_bounds__(TA_THREADS) void k_track_ages(const vo_match *__restrict__ final_m,
                                                           const int *__restrict__ info, int kp_stride,
                                                           const int *__restrict__ q_slot,
                                                           const int *__restrict__ t_slot,
                                                           const int *__restrict__ has_prev, int frames,
                                                           const int *__restrict__ nkp, int *__restrict__ ages,
                                                           int *__restrict__ filter_flag)
{
    __shared__ int s_winner[VO_MAX_FEATURES_LIMIT];
    __shared__ int s_cnt;
    const int s = blockIdx.x, tid = threadIdx.x;
    for (int f = 0; f < frames; f++) {
        const int b = s * frames + f;
        const int ts = t_slot[b];
        int *age_c = ages + (size_t)ts * kp_stride;
        const int n_curr = nkp[ts];
        if (!has_prev[b]) {
            for (int i = tid; i < kp_stride; i += TA_THREADS) age_c[i] = 0;
            if (tid == 0) filter_flag[b] = 0;
            __syncthreads();
            continue;
        }
        const int qs = q_slot[b];
        const int *age_p = ages + (size_t)qs * kp_stride;
        const int n_prev = nkp[qs];
        const int n = info[b * 4 + 3];
        const vo_match *m = final_m + (size_t)b * kp_stride;
        for (int i = tid; i < kp_stride; i += TA_THREADS) s_winner[i] = -1;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int i = tid; i < n; i += TA_THREADS) {
            const int t = m[i].train;
            if (t >= 0 && t < n_curr) atomicMax(&s_winner[t], i);
        }
        __syncthreads();
        for (int i = tid; i < kp_stride; i += TA_THREADS) {
            const int w = s_winner[i];
            int a = 0;
            if (w >= 0) {
                const int q = m[w].query;
                a = ((q >= 0 && q < n_prev) ? age_p[q] : 0) + 1;
            }
            age_c[i] = a;
        }
        __syncthreads();
        int mature = 0;
        for (int i = tid; i < n; i += TA_THREADS) {
            const int t = m[i].train;
            mature += (t >= 0 && t < n_curr && age_c[t] >= 3);
        }
        if (mature) atomicAdd(&s_cnt, mature);
        __syncthreads();
        if (tid == 0) filter_flag[b] = (n > 80 && s_cnt >= 50) ? 1 : 0;
        __syncthreads();
    }
}

void launch_track_ages(const vo_match *final_m, const int *info, int kp_stride, const int *q_slot,
                       const int *t_slot, const int *has_prev, int streams, int frames, const int *nkp,
                       int *ages, int *filter_flag, hipStream_t s)
{
    hipLaunchKernelGGL(k_track_ages, dim3(streams), dim3(TA_THREADS), 0, s, final_m, info, kp_stride, q_slot,
                       t_slot, has_prev, frames, nkp, ages, filter_flag);
}

// =====================================================================
// multi-view filter + back-projection (one block per frame)
// =====================================================================

constexpr int BP_THREADS = 256;

__device__ __forceinline__ int block_compact_slot(bool keep, int *s_wave, int *s_base, int lane, int wave,
                                                  int nwaves)
{
    const uint64_t bm = __ballot(keep);
    if (lane == 0) s_wave[wave] = __popcll(bm);
    __syncthreads();
    int off = *s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    const int slot = off + __popcll(bm & ((1ull << lane) - 1ull));
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < nwaves; w++) tot += s_wave[w];
        *s_base += tot;
    }
    __syncthreads();
    return slot;
}

// out_info[b] = {n_tracked, n_3d}
__global__ __launch_bounds__(BP_THREADS) void k_filter_backproject(
    const vo_match *__restrict__ final_m, const int *__restrict__ info, int kp_stride,
    const int *__restrict__ t_slot, const int *__restrict__ has_prev, const int *__restrict__ nkp,
    const int *__restrict__ ages, const int *__restrict__ filter_flag, const float *__restrict__ depth,
    int width, int height, int depth_stride, size_t depth_frame_stride, double fx, double fy, double cx,
    double cy, vo_match *__restrict__ tracked, vo_match *__restrict__ valid, float *__restrict__ pts3,
    int *__restrict__ out_info)
{
    __shared__ int s_wave[BP_THREADS / kWave];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = has_prev[b] ? info[b * 4 + 3] : 0;
    const vo_match *m = final_m + (size_t)b * kp_stride;
    vo_match *tr = tracked + (size_t)b * kp_stride;
    vo_match *va = valid + (size_t)b * kp_stride;
    float *p3 = pts3 + (size_t)b * kp_stride * 3;
    const int ts = t_slot[b];
    const int *age_c = ages + (size_t)ts * kp_stride;
    const int n_curr = nkp[ts];
    const bool filt = filter_flag[b] != 0;

    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += BP_THREADS) {
        const int i = i0 + tid;
        bool keep = false;
        vo_match mm;
        if (i < n) {
            mm = m[i];
            keep = !filt || (mm.train >= 0 && mm.train < n_curr && age_c[mm.train] >= 3);
        }
        const int slot = block_compact_slot(keep, s_wave, &s_base, lane, wave, BP_THREADS / kWave);
        if (keep) tr[slot] = mm;
    }
    const int n_tr = s_base;
    __syncthreads();
    if (tid == 0) s_base = 0;
    __syncthreads();
    const float *dimg = depth ? depth + b * depth_frame_stride : nullptr;
    for (int i0 = 0; i0 < n_tr; i0 += BP_THREADS) {
        const int i = i0 + tid;
        bool keep = false;
        vo_match mm;
        float X = 0, Y = 0, z = 0;
        if (i < n_tr && dimg) {
            mm = tr[i];
            const int u = (int)roundf(mm.cx), v = (int)roundf(mm.cy);
            if (u >= 0 && u < width && v >= 0 && v < height) {
                z = dimg[(size_t)v * depth_stride + u];
                if (!(isnan(z) || isinf(z) || z <= 0.0f || z < 50.0f || z > 20000.0f)) {
                    X = (float)(((double)mm.cx - cx) * (double)z / fx);
                    Y = (float)(((double)mm.cy - cy) * (double)z / fy);
                    keep = true;
                }
            }
        }
        const int slot = block_compact_slot(keep, s_wave, &s_base, lane, wave, BP_THREADS / kWave);
        if (keep) {
            va[slot] = mm;
            p3[3 * slot] = X; p3[3 * slot + 1] = Y; p3[3 * slot + 2] = z;
        }
    }
    if (tid == 0) { out_info[b * 2] = n_tr; out_info[b * 2 + 1] = s_base; }
}

void launch_filter_backproject(const vo_match *final_m, const int *info, int kp_stride, const int *t_slot,
                               const int *has_prev, const int *nkp, const int *ages, const int *filter_flag,
                               const float *depth, int width, int height, int depth_stride,
                               size_t depth_frame_stride, double fx, double fy, double cx, double cy,
                               vo_match *tracked, vo_match *valid, float *pts3, int *out_info, int batch,
                               hipStream_t s)
{
    hipLaunchKernelGGL(k_filter_backproject, dim3(batch), dim3(BP_THREADS), 0, s, final_m, info, kp_stride,
                       t_slot, has_prev, nkp, ages, filter_flag, depth, width, height, depth_stride,
                       depth_frame_stride, fx, fy, cx, cy, tracked, valid, pts3, out_info);
}

// =====================================================================
// small dense algebra (device)
// =====================================================================

// Cyclic Jacobi eigen-decomposition of a symmetric N x N matrix; eigenvalues
// descending in w, row i of Vt the matching eigenvector.
template <int N>
__device__ void jacobi_eig_desc(const double *Ain, double *w, double *Vt)
{
    double A[N * N], V[N * N];
    for (int i = 0; i < N * N; i++) A[i] = Ain[i];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) V[i * N + j] = i == j;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                if (i != j) off += A[i * N + j] * A[i * N + j];
                else diag += A[i * N + j] * A[i * N + j];
            }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < N - 1; p++)
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < N; k++) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; k++) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
    int order[N];
    for (int i = 0; i < N; i++) order[i] = i;
    for (int i = 0; i < N; i++)
        for (int j = i + 1; j < N; j++)
            if (A[order[j] * N + order[j]] > A[order[i] * N + order[i]]) {
                const int t = order[i]; order[i] = order[j]; order[j] = t;
            }
    for (int i = 0; i < N; i++) {
        w[i] = A[order[i] * N + order[i]];
        for (int k = 0; k < N; k++) Vt[i * N + k] = V[k * N + order[i]];
    }
}

// nearest rotation (U V^T of the SVD) of a 3x3 matrix
__device__ void polar_rotation3(const double *A, double *R)
{
    double AtA[9], w[3], Vt[9], U[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = s;
        }
    jacobi_eig_desc<3>(AtA, w, Vt);
    for (int c = 0; c < 3; c++) {
        double u[3], nrm = 0;
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
        for (int i = 0; i < 3; i++) U[i * 3 + c] = u[i] / nrm;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = U[i * 3 + 0] * Vt[0 * 3 + j] + U[i * 3 + 1] * Vt[1 * 3 + j] + U[i * 3 + 2] * Vt[2 * 3 + j];
}

// least squares of a 6 x N system (N <= 5) by Householder QR
template <int N>
__device__ void lstsq6(const double *Ain, const double *bin, double *x)
{
    constexpr int M = 6;
    double A[M * N], b[M];
    for (int i = 0; i < M * N; i++) A[i] = Ain[i];
    for (int i = 0; i < M; i++) b[i] = bin[i];
    for (int k = 0; k < N; k++) {
        double s = 0;
        for (int i = k; i < M; i++) s += A[i * N + k] * A[i * N + k];
        const double nrm = sqrt(s);
        if (nrm == 0) continue;
        const double alpha = A[k * N + k] > 0 ? -nrm : nrm;
        double v[M];
        for (int i = k; i < M; i++) v[i] = A[i * N + k];
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < M; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        const double beta = 2.0 / vn;
        for (int j = k; j < N; j++) {
            double dot = 0;
            for (int i = k; i < M; i++) dot += v[i] * A[i * N + j];
            dot *= beta;
            for (int i = k; i < M; i++) A[i * N + j] -= dot * v[i];
        }
        double dot = 0;
        for (int i = k; i < M; i++) dot += v[i] * b[i];
        dot *= beta;
        for (int i = k; i < M; i++) b[i] -= dot * v[i];
    }
    for (int k = N - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < N; j++) s -= A[k * N + j] * x[j];
        x[k] = s / A[k * N + k];
    }
}

__device__ bool solve_gauss6(const double *Ain, const double *bin, double *x)
{
    constexpr int n = 6;
    double A[36], b[6];
    for (int i = 0; i < 36; i++) A[i] = Ain[i];
    for (int i = 0; i < 6; i++) b[i] = bin[i];
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[piv * n + k])) piv = i;
        if (A[piv * n + k] == 0) return false;
        if (piv != k) {
            for (int j = 0; j < n; j++) { const double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            const double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < n; i++) {
            const double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = s / A[k * n + k];
    }
    return true;
}

__device__ bool invert3(const double *m, double *inv)
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

// =====================================================================
// EPnP on exactly five correspondences (one lane per hypothesis)
// =====================================================================

__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double dist2(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

struct Epnp5 {
    double fu, fv, uc, vc;
    double pws[15], us[10], alphas[20], pcs[15];
    double cws[4][3], ccs[4][3];
};

__device__ double epnp_R_and_t(Epnp5 &e, const double *ut, const double *betas, double *R, double *t)
{
    constexpr int n = 5;
    for (int i = 0; i < 4; i++) e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0;
    for (int i = 0; i < 4; i++) {
        const double *v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) e.ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double *a = e.alphas + 4 * i;
        double *pc = e.pcs + 3 * i;
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
    }
    if (e.pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++) e.ccs[i][j] = -e.ccs[i][j];
        for (int i = 0; i < 3 * n; i++) e.pcs[i] = -e.pcs[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double *pc = e.pcs + 3 * i, *pw = e.pws + 3 * i;
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
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// EPnP (epnp::compute_pose) on the five points stored in e.pws / e.us.
__device__ bool epnp5_solve(Epnp5 &e, double *R, double *t)
{
    constexpr int n = 5;
    // control points: centroid + principal directions
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
    for (int j = 0; j < 3; j++) e.cws[0][j] /= n;
    {
        double ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dc[3], uct[9];
        for (int i = 0; i < n; i++) {
            double d[3];
            for (int j = 0; j < 3; j++) d[j] = e.pws[3 * i + j] - e.cws[0][j];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) ptp[a * 3 + b] += d[a] * d[b];
        }
        jacobi_eig_desc<3>(ptp, dc, uct);
        for (int i = 1; i < 4; i++) {
            const double k = sqrt((dc[i - 1] > 0 ? dc[i - 1] : 0) / n);
            for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * uct[3 * (i - 1) + j];
        }
    }
    // barycentric coordinates
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
        if (!invert3(cc, ci)) return false;
        for (int i = 0; i < n; i++) {
            const double *pi = e.pws + 3 * i;
            double *a = e.alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                           ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
    }
    double ut[144];
    {
        double mtm[144], d[12];
        for (int i = 0; i < 144; i++) mtm[i] = 0;
        for (int i = 0; i < n; i++) {
            const double *as = e.alphas + 4 * i;
            double m1[12], m2[12];
            for (int k = 0; k < 4; k++) {
                m1[3 * k] = as[k] * e.fu; m1[3 * k + 1] = 0.0; m1[3 * k + 2] = as[k] * (e.uc - e.us[2 * i]);
                m2[3 * k] = 0.0; m2[3 * k + 1] = as[k] * e.fv; m2[3 * k + 2] = as[k] * (e.vc - e.us[2 * i + 1]);
            }
            for (int a = 0; a < 12; a++)
                for (int b = 0; b < 12; b++) mtm[a * 12 + b] += m1[a] * m1[b] + m2[a] * m2[b];
        }
        jacobi_eig_desc<12>(mtm, d, ut);
    }
    double L[60], rho[6];
    {
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
    }
    double best_rep = 0, Rc[9], tc[3], be[4];
    // approximation 1: [B11 B12 B13 B14]
    {
        double l[24], b4[4];
        for (int i = 0; i < 6; i++) {
            l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1];
            l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6];
        }
        lstsq6<4>(l, rho, b4);
        if (b4[0] < 0) { be[0] = sqrt(-b4[0]); be[1] = -b4[1] / be[0]; be[2] = -b4[2] / be[0]; be[3] = -b4[3] / be[0]; }
        else { be[0] = sqrt(b4[0]); be[1] = b4[1] / be[0]; be[2] = b4[2] / be[0]; be[3] = b4[3] / be[0]; }
        epnp_gauss_newton(L, rho, be);
        best_rep = epnp_R_and_t(e, ut, be, R, t);
    }
    // approximation 2: [B11 B12 B22]
    {
        double l[18], b3[3];
        for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
        lstsq6<3>(l, rho, b3);
        if (b3[0] < 0) { be[0] = sqrt(-b3[0]); be[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0; }
        else { be[0] = sqrt(b3[0]); be[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) be[0] = -be[0];
        be[2] = 0.0; be[3] = 0.0;
        epnp_gauss_newton(L, rho, be);
        const double rep = epnp_R_and_t(e, ut, be, Rc, tc);
        if (rep < best_rep) {
            best_rep = rep;
            for (int i = 0; i < 9; i++) R[i] = Rc[i];
            for (int i = 0; i < 3; i++) t[i] = tc[i];
        }
    }
    // approximation 3: [B11 B12 B22 B13 B23]
    {
        double l[30], b5[5];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 5; j++) l[i * 5 + j] = L[i * 10 + j];
        lstsq6<5>(l, rho, b5);
        if (b5[0] < 0) { be[0] = sqrt(-b5[0]); be[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
        else { be[0] = sqrt(b5[0]); be[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) be[0] = -be[0];
        be[2] = b5[3] / be[0];
        be[3] = 0.0;
        epnp_gauss_newton(L, rho, be);
        const double rep = epnp_R_and_t(e, ut, be, Rc, tc);
        if (rep < best_rep) {
            best_rep = rep;
            for (int i = 0; i < 9; i++) R[i] = Rc[i];
            for (int i = 0; i < 3; i++) t[i] = tc[i];
        }
    }
    bool ok = true;
    for (int i = 0; i < 9; i++) ok = ok && isfinite(R[i]);
    for (int i = 0; i < 3; i++) ok = ok && isfinite(t[i]);
    return ok;
}

// =====================================================================
// PnP RANSAC + refinement (one block per frame)
// =====================================================================

constexpr int PR_THREADS = 512;
constexpr int PR_WAVES = PR_THREADS / kWave;
constexpr int PR_CHUNK = 128;
constexpr int PR_WIN = PR_CHUNK * 5 + 128;
constexpr int PR_NRED = 44;  // 36 JtJ + 6 JtErr + |err|^2 (+1 pad)

struct PrShared {
    double model[PR_CHUNK][6];  // rvec, tvec per hypothesis
    double best[6];
    double red[PR_WAVES][PR_NRED];
    double sums[PR_NRED];
    double param[6], prev[6];
    double LL[144];
    int ok[PR_CHUNK];
    int counts[PR_CHUNK];
    int start[PR_CHUNK];
    uint16_t r[PR_WIN];
    uint8_t cons[PR_WIN];
    int wave_tot[PR_WAVES];
    int nchunk, iter, niters, max_good, done;
    long long pos;
    int scan_base;
    int lm_state, lm_ok;
};

__device__ __forceinline__ int draw_subset5(const uint16_t *r, int o, int (&idx)[5])
{
    int p = o;
    for (int i = 0; i < 5; i++) {
        for (;;) {
            if (p >= PR_WIN) return 0;
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

// block-wide deterministic sum of NV per-thread doubles -> S.sums[0..NV)
template <int NV>
__device__ void block_reduce(PrShared &S, double (&v)[NV])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if (lane == 0) S.red[wave][k] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        double x = 0;
        for (int w = 0; w < PR_WAVES; w++) x += S.red[w][threadIdx.x];
        S.sums[threadIdx.x] = x;
    }
    __syncthreads();
}

// out: pose[b] = {success, n_inliers, iters_run, n_points}, rt[b] = rvec, tvec
__global__ __launch_bounds__(PR_THREADS) void k_pnp_ransac(const vo_match *__restrict__ valid,
                                                           const float *__restrict__ pts3,
                                                           const int *__restrict__ bp_info, int kp_stride,
                                                           const uint32_t *__restrict__ rng_raw, int rng_len,
                                                           double fx, double fy, double cx, double cy,
                                                           int max_iters, double threshold, double confidence,
                                                           int *__restrict__ inliers, int *__restrict__ pose,
                                                           double *__restrict__ rt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    PrShared &S = *reinterpret_cast<PrShared *>(s_dyn);
    unsigned char *dyn = s_dyn + ((sizeof(PrShared) + 15) & ~size_t(15));
    float *obj = reinterpret_cast<float *>(dyn);                 // n x 3
    float *img = obj + 3 * (size_t)kp_stride;                    // n x 2
    uint8_t *msk = reinterpret_cast<uint8_t *>(img + 2 * (size_t)kp_stride);  // n

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(bp_info[b * 2 + 1], kp_stride);
    const vo_match *vm = valid + (size_t)b * kp_stride;
    const float *p3 = pts3 + (size_t)b * kp_stride * 3;
    int *inl = inliers + (size_t)b * kp_stride;
    int *po = pose + b * 4;
    double *rto = rt + (size_t)b * 6;

    if (tid < 6) rto[tid] = 0;
    // reference src/frame_processor.cpp:176-177 needs >= 4 points; with 4 OpenCV
    // uses P3P (not provided): report failure for n < 5
    if (n < 5) {
        if (tid == 0) { po[0] = 0; po[1] = 0; po[2] = 0; po[3] = n; }
        return;
    }
    for (int i = tid; i < n; i += PR_THREADS) {
        obj[3 * i] = p3[3 * i]; obj[3 * i + 1] = p3[3 * i + 1]; obj[3 * i + 2] = p3[3 * i + 2];
        img[2 * i] = vm[i].px; img[2 * i + 1] = vm[i].py;
        msk[i] = 0;
    }
    if (tid == 0) {
        S.iter = 0; S.niters = max_iters > 1 ? max_iters : 1; S.max_good = 0; S.done = 0; S.pos = 0;
    }
    __syncthreads();
    const float thr2 = (float)(threshold * threshold);

    if (n == 5) {
        // OpenCV runs the minimal solver once and reports every point as inlier
        if (tid == 0) {
            Epnp5 e;
            e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
            for (int i = 0; i < 5; i++) {
                for (int j = 0; j < 3; j++) e.pws[3 * i + j] = obj[3 * i + j];
                e.us[2 * i] = img[2 * i]; e.us[2 * i + 1] = img[2 * i + 1];
            }
            double R[9], t[3], rv[3];
            const bool ok = epnp5_solve(e, R, t);
            if (ok) {
                rodrigues_m2v(R, rv);
                for (int i = 0; i < 3; i++) { rto[i] = rv[i]; rto[3 + i] = t[i]; }
                for (int i = 0; i < 5; i++) inl[i] = i;
            }
            po[0] = ok; po[1] = ok ? 5 : 0; po[2] = 0; po[3] = n;
        }
        return;
    }

    // ---------------- RANSAC rounds ----------------
    while (true) {
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
            const int want = min(PR_CHUNK, S.niters - S.iter);
            while (nc < want && w < PR_WIN && S.cons[w] != 0) { S.start[nc++] = w; w += S.cons[w]; }
            S.nchunk = nc;
            S.pos = pos + w;
            if (nc == 0) S.done = 1;  // RNG table exhausted
        }
        __syncthreads();
        const int nc = S.nchunk;
        if (tid < nc) {
            int idx[5];
            draw_subset5(S.r, S.start[tid], idx);
            Epnp5 e;
            e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
            for (int i = 0; i < 5; i++) {
                for (int j = 0; j < 3; j++) e.pws[3 * i + j] = obj[3 * idx[i] + j];
                e.us[2 * i] = img[2 * idx[i]]; e.us[2 * i + 1] = img[2 * idx[i] + 1];
            }
            double R[9], t[3], rv[3];
            const bool ok = epnp5_solve(e, R, t);
            S.ok[tid] = ok;
            if (ok) {
                rodrigues_m2v(R, rv);
                for (int i = 0; i < 3; i++) { S.model[tid][i] = rv[i]; S.model[tid][3 + i] = t[i]; }
            }
        }
        __syncthreads();
        for (int h = wave; h < nc; h += PR_WAVES) {
            if (!S.ok[h]) continue;
            double R[9];
            rodrigues_v2m(S.model[h], R, nullptr);
            const double t[3] = {S.model[h][3], S.model[h][4], S.model[h][5]};
            int cnt = 0;
            for (int i = lane; i < n; i += kWave)
                cnt += (pnp_error(obj + 3 * i, img + 2 * i, R, t, fx, fy, cx, cy) <= thr2);
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (lane == 0) S.counts[h] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            for (int h = 0; h < nc; h++) {
                if (S.ok[h]) {
                    const int goodc = S.counts[h];
                    if (goodc > max(S.max_good, 4)) {
                        for (int i = 0; i < 6; i++) S.best[i] = S.model[h][i];
                        S.max_good = goodc;
                        S.niters = pnp_update_num_iters(confidence, (double)(n - goodc) / n, 5, S.niters);
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

    // ---------------- refinement: solvePnP(SOLVEPNP_ITERATIVE) on inliers ----
    // planarity test + DLT initialisation
    bool init_ok = k >= 6;
    if (init_ok) {
        double v[12];
        for (int j = 0; j < 12; j++) v[j] = 0;
        for (int i = tid; i < n; i += PR_THREADS)
            if (msk[i]) {
                v[0] += obj[3 * i]; v[1] += obj[3 * i + 1]; v[2] += obj[3 * i + 2];
            }
        double c3[3] = {v[0], v[1], v[2]};
        block_reduce<3>(S, c3);
        const double Mc[3] = {S.sums[0] / k, S.sums[1] / k, S.sums[2] / k};
        __syncthreads();
        double mm[9];
        for (int j = 0; j < 9; j++) mm[j] = 0;
        for (int i = tid; i < n; i += PR_THREADS)
            if (msk[i]) {
                const double d[3] = {obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2]};
                for (int a = 0; a < 3; a++)
                    for (int bb = 0; bb < 3; bb++) mm[a * 3 + bb] += d[a] * d[bb];
            }
        block_reduce<9>(S, mm);
        if (tid == 0) {
            double MM[9], W[3], Vt[9];
            for (int j = 0; j < 9; j++) MM[j] = S.sums[j];
            jacobi_eig_desc<3>(MM, W, Vt);
            S.lm_ok = !(W[2] / W[1] < 1e-3);
        }
        __syncthreads();
        init_ok = S.lm_ok != 0;
    }
    if (init_ok) {
        // L^T L: entry (a, b) summed by thread a*12+b over the inliers in index
        // order (same order as a sequential accumulation)
        if (tid < 144) {
            const int a = tid / 12, bb = tid % 12;
            double acc = 0;
            for (int i = 0; i < n; i++) {
                if (!msk[i]) continue;
                const double M0 = obj[3 * i], M1 = obj[3 * i + 1], M2 = obj[3 * i + 2];
                const double x = -((double)img[2 * i] - cx) / fx, y = -((double)img[2 * i + 1] - cy) / fy;
                const double r0[12] = {M0, M1, M2, 1, 0, 0, 0, 0, x * M0, x * M1, x * M2, x};
                const double r1[12] = {0, 0, 0, 0, M0, M1, M2, 1, y * M0, y * M1, y * M2, y};
                acc += r0[a] * r0[bb] + r1[a] * r1[bb];
            }
            S.LL[tid] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            double LW[12], LV[144], LLc[144];
            for (int j = 0; j < 144; j++) LLc[j] = S.LL[j];
            jacobi_eig_desc<12>(LLc, LW, LV);
            const double *RRt = LV + 11 * 12;
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

    if (init_ok) {
        // CvLevMarq: <= 20 iterations, epsilon FLT_EPSILON, lambda = 10^lambdaLg10
        const int max_iter = 20;
        const double eps = FLT_EPSILON;
        const double LOG10 = log(10.);
        double prevErrNorm = DBL_MAX, errNorm = 0;
        int lambdaLg10 = -3, iters = 0;
        for (;;) {
            // J^T J, J^T e, |e|^2 at S.param
            double acc[PR_NRED];
            for (int j = 0; j < PR_NRED; j++) acc[j] = 0;
            {
                double R[9], dRdr[27], p[6];
                for (int j = 0; j < 6; j++) p[j] = S.param[j];
                rodrigues_v2m(p, R, dRdr);
                for (int i = tid; i < n; i += PR_THREADS) {
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
                    for (int q = 0; q < 3; q++) {
                        const double *d = dRdr + 9 * q;
                        const double dX = d[0] * M[0] + d[1] * M[1] + d[2] * M[2];
                        const double dY = d[3] * M[0] + d[4] * M[1] + d[5] * M[2];
                        const double dZ = d[6] * M[0] + d[7] * M[1] + d[8] * M[2];
                        j0[q] = fx * (dX - x * dZ) * z;
                        j1[q] = fy * (dY - y * dZ) * z;
                    }
                    for (int a = 0; a < 6; a++) {
                        for (int bb = 0; bb < 6; bb++) acc[a * 6 + bb] += j0[a] * j0[bb] + j1[a] * j1[bb];
                        acc[36 + a] += j0[a] * e0 + j1[a] * e1;
                    }
                    acc[42] += e0 * e0 + e1 * e1;
                }
            }
            block_reduce<PR_NRED>(S, acc);
            if (tid < 6) S.prev[tid] = S.param[tid];
            if (iters == 0) prevErrNorm = sqrt(S.sums[42]);
            double JtJ[36], JtErr[6];
            for (int j = 0; j < 36; j++) JtJ[j] = S.sums[j];
            for (int j = 0; j < 6; j++) JtErr[j] = S.sums[36 + j];
            __syncthreads();
            for (;;) {
                if (tid == 0) {
                    const double lambda = exp(lambdaLg10 * LOG10);
                    double A[36], dx[6];
                    for (int j = 0; j < 36; j++) A[j] = JtJ[j];
                    for (int a = 0; a < 6; a++) A[a * 6 + a] *= 1. + lambda;
                    if (!solve_gauss6(A, JtErr, dx))
                        for (int a = 0; a < 6; a++) dx[a] = 0;
                    for (int a = 0; a < 6; a++) S.param[a] = S.prev[a] - dx[a];
                }
                __syncthreads();
                double e2[1] = {0};
                {
                    double R[9], p[6];
                    for (int j = 0; j < 6; j++) p[j] = S.param[j];
                    rodrigues_v2m(p, R, nullptr);
                    for (int i = tid; i < n; i += PR_THREADS) {
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
            if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < eps) break;
            prevErrNorm = errNorm;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) {
        // refinement unavailable (planar / < 6 inliers): OpenCV reports failure
        // and hands back the best RANSAC hypothesis
        for (int j = 0; j < 6; j++) rto[j] = init_ok ? S.param[j] : S.best[j];
        po[0] = init_ok ? 1 : 0;
        po[1] = init_ok ? k : 0;
        po[2] = S.iter;
        po[3] = n;
    }
}

void launch_pnp_ransac(const vo_match *valid, const float *pts3, const int *bp_info, int kp_stride,
                       const uint32_t *rng_raw, int rng_len, double fx, double fy, double cx, double cy,
                       int max_iters, double threshold, double confidence, int *inliers, int *pose,
                       double *rt, int batch, hipStream_t s)
{
    const size_t lds = ((sizeof(PrShared) + 15) & ~size_t(15)) + (size_t)kp_stride * (5 * sizeof(float) + 1) + 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pnp_ransac),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_pnp_ransac, dim3(batch), dim3(PR_THREADS), lds, s, valid, pts3, bp_info, kp_stride,
                       rng_raw, rng_len, fx, fy, cx, cy, max_iters, threshold, confidence, inliers, pose, rt);
}


// =====================================================================
// end-of-batch bookkeeping
// =====================================================================

// feature slot of frame f of stream s is streams + s*frames + f; slot s
// carries the previous batch's last frame of stream s
__global__ __launch_bounds__(256) void k_carry(vo_kp *__restrict__ kps, uint8_t *__restrict__ desc,
                                               int *__restrict__ nkp, int *__restrict__ ages, int kp_stride,
                                               int frames)
{
    const int s = blockIdx.y;
    const int dst = s, src = (int)gridDim.y + s * frames + frames - 1;
    const int n = nkp[src];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) nkp[dst] = n;
    if (i < kp_stride) ages[(size_t)dst * kp_stride + i] = ages[(size_t)src * kp_stride + i];
    if (i < n) {
        kps[(size_t)dst * kp_stride + i] = kps[(size_t)src * kp_stride + i];
        const uint4 *sd = reinterpret_cast<const uint4 *>(desc + ((size_t)src * kp_stride + i) * 32);
        uint4 *dd = reinterpret_cast<uint4 *>(desc + ((size_t)dst * kp_stride + i) * 32);
        dd[0] = sd[0];
        dd[1] = sd[1];
    }
}

void launch_carry(vo_kp *kps, uint8_t *desc, int *nkp, int *ages, int kp_stride, int streams, int frames,
                  hipStream_t s)
{
    dim3 grid((kp_stride + 255) / 256, streams);
    hipLaunchKernelGGL(k_carry, grid, dim3(256), 0, s, kps, desc, nkp, ages, kp_stride, frames);
}

__global__ void k_collect(const int *__restrict__ nkp, const int *__restrict__ t_slot,
                          const int *__restrict__ n_good, const int *__restrict__ info,
                          const int *__restrict__ bp_info, const int *__restrict__ pose,
                          const double *__restrict__ rt, vo_frame_result *__restrict__ out, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    vo_frame_result r;
    r.n_keypoints = nkp[t_slot[b]];
    r.n_ratio = n_good[b];
    r.f_found = info[b * 4]; r.f_iters = info[b * 4 + 1]; r.f_inliers = info[b * 4 + 2];
    r.n_matches = info[b * 4 + 3];
    r.n_tracked = bp_info[b * 2]; r.n_3d = bp_info[b * 2 + 1];
    r.pnp_success = pose[b * 4]; r.pnp_inliers = pose[b * 4 + 1]; r.pnp_iters = pose[b * 4 + 2];
    r.pnp_points = pose[b * 4 + 3];
    for (int i = 0; i < 3; i++) { r.rvec[i] = rt[b * 6 + i]; r.tvec[i] = rt[b * 6 + 3 + i]; }
    out[b] = r;
}

void launch_collect(const int *nkp, const int *t_slot, const int *n_good, const int *info,
                    const int *bp_info, const int *pose, const double *rt, vo_frame_result *out, int batch,
                    hipStream_t s)
{
    hipLaunchKernelGGL(k_collect, dim3((batch + 63) / 64), dim3(64), 0, s, nkp, t_slot, n_good, info, bp_info,
                       pose, rt, out, batch);
}

}  // namespace vo
