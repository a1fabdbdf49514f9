// Stereo front end for gfx950: left/right key-point matching inside an
// epipolar band with cross-check, disparity -> depth, and the sparse depth
// plane that the back-projection stage samples at the rounded key point
// (reference src/frame_processor.cpp:230-308).  The reference has no stereo
// matcher (its depth comes from the ZED SDK, src/zed_interface.cpp:84-85); the
// matching rule is this package's own:
//   candidates of left key point i: right key points j with
//       |yL_i - yR_j| <= band  and  min_disp <= xL_i - xR_j <= max_disp,
//   best = smallest Hamming-256 distance (ties: lowest j), Lowe test against the
//   second candidate when there is one, kept only if i is also the best left
//   candidate of j.
#include <math.h>

#include "vo_internal.h"

namespace vo {

namespace {

constexpr int ST_THREADS = 256;
constexpr uint32_t ST_NONE = 0xffffffffu;
constexpr int SI_THREADS = 1024;
constexpr int SI_ROWS = kMaxDim;  // one bucket per image row

// image row of a key point, clamped into the bucket range (monotone, so a range of rows maps to a range of buckets)
__device__ __forceinline__ int row_bucket(float y)
{
    return y > 0.f ? (int)fminf(floorf(y), (float)(SI_ROWS - 1)) : 0;  // (a NaN lands in bucket 0)
}

// The key points of feature slot slot0 + z (z = blockIdx.x) bucketed by image row: `order` lists their indices row by
// row (any order inside a row), row_start[r] is the position of row r's first entry (row_start[SI_ROWS] the total).
// A band of +-2 rows holds about 20 of 4000 key points: the matcher below looks at those instead of all of them
// (testing every pair through LDS tiles took 0.73 ms per 64 HD1080 stereo frames, more than the full 256-bit
// matcher of the temporal stage).
__global__ __launch_bounds__(SI_THREADS) void k_row_index(const vo_kp *__restrict__ kps, const int *__restrict__ nkp,
                                                         int kp_stride, int slot0, uint16_t *__restrict__ order,
                                                         int *__restrict__ row_start)
{
    __shared__ int s_cnt[SI_ROWS];
    __shared__ uint32_t s_scan[SI_THREADS];
    static_assert(SI_ROWS == 4 * SI_THREADS, "four rows per thread");
    const int z = blockIdx.x, tid = threadIdx.x;
    const int n = min(nkp[slot0 + z], kp_stride);
    const vo_kp *k = kps + (size_t)(slot0 + z) * kp_stride;
    order += (size_t)z * kp_stride;
    row_start += (size_t)z * (SI_ROWS + 1);
    for (int r = tid; r < SI_ROWS; r += SI_THREADS) s_cnt[r] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SI_THREADS) atomicAdd(&s_cnt[row_bucket(k[i].y)], 1);
    __syncthreads();
    int c[4];
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { c[j] = s_cnt[4 * tid + j]; sum += (uint32_t)c[j]; }
    const uint32_t inc = block_scan_inclusive<SI_THREADS>(sum, s_scan);
    int at = (int)(inc - sum);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        row_start[4 * tid + j] = at;
        s_cnt[4 * tid + j] = at;  // from here on: the next free position of the row
        at += c[j];
    }
    if (tid == SI_THREADS - 1) row_start[SI_ROWS] = at;
    __syncthreads();
    for (int i = tid; i < n; i += SI_THREADS) order[atomicAdd(&s_cnt[row_bucket(k[i].y)], 1)] = (uint16_t)i;
}

// Two smallest keys (distance << 16 | train index) of every query among the train key points inside its band, for
// frame blockIdx.y of a batch: queries are the key points of feature slot q_slot0 + z, trains those of t_slot0 + z,
// listed row by row in order / row_start.  One thread per query; the keys make the result independent of the order
// in which the candidates come.  sign = +1: queries are left key points (disparity = xq - xt); -1: right ones.
// out: frames x kp_stride.
__global__ __launch_bounds__(ST_THREADS) void k_band_match(const vo_kp *__restrict__ kps, const uint8_t *__restrict__ desc,
                                                          const int *__restrict__ nkp, int kp_stride, int q_slot0,
                                                          int t_slot0, float band, float min_disp, float max_disp,
                                                          int sign, const uint16_t *__restrict__ order,
                                                          const int *__restrict__ row_start, uint2 *__restrict__ out)
{
    const int z = blockIdx.y;
    const int nq = min(nkp[q_slot0 + z], kp_stride);
    const int q = blockIdx.x * ST_THREADS + threadIdx.x;
    if (q >= nq) return;
    const vo_kp *qk = kps + (size_t)(q_slot0 + z) * kp_stride, *tk = kps + (size_t)(t_slot0 + z) * kp_stride;
    const uint4 *qd = reinterpret_cast<const uint4 *>(desc + (size_t)(q_slot0 + z) * kp_stride * 32);
    const uint4 *td = reinterpret_cast<const uint4 *>(desc + (size_t)(t_slot0 + z) * kp_stride * 32);
    order += (size_t)z * kp_stride;
    row_start += (size_t)z * (SI_ROWS + 1);
    const uint4 a0 = qd[2 * q], a1 = qd[2 * q + 1];
    const float qx = qk[q].x, qy = qk[q].y;
    // the rows that can hold a candidate, one more on either side for the rounding of the two sums (the exact test
    // below decides)
    const int r_lo = max(row_bucket(qy - band) - 1, 0), r_hi = min(row_bucket(qy + band) + 1, SI_ROWS - 1);
    uint32_t k0 = ST_NONE, k1 = ST_NONE;
    for (int at = row_start[r_lo], end = row_start[r_hi + 1]; at < end; at++) {
        const int j = order[at];
        const float tx = tk[j].x, ty = tk[j].y;
        const float dy = qy - ty;
        const float dx = sign > 0 ? qx - tx : tx - qx;
        if (!(fabsf(dy) <= band) || !(dx >= min_disp) || !(dx <= max_disp)) continue;
        const uint4 b0 = td[2 * j], b1 = td[2 * j + 1];
        uint32_t d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w);
        d += __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
        const uint32_t key = (d << 16) | (uint32_t)j;
        const uint32_t hi = max(k0, key);
        k0 = min(k0, key);
        k1 = min(k1, hi);
    }
    out[(size_t)z * kp_stride + q] = make_uint2(k0, k1);
}

__device__ __forceinline__ void merge_top2(const uint2 *__restrict__ p, int splits, uint32_t &k0, uint32_t &k1)
{
    k0 = ST_NONE; k1 = ST_NONE;
    for (int s = 0; s < splits; s++) {
        const uint2 v = p[s];
        uint32_t hi = max(k0, v.x);
        k0 = min(k0, v.x);
        k1 = min(k1, hi);
        k1 = min(k1, v.y);
    }
}

// One thread per left key point of frame blockIdx.y: ratio test, cross-check,
// disparity, depth, and the sparse depth plane (bits of a positive float order
// like the value, so an unsigned atomic minimum keeps the nearest depth where
// key points share a pixel; the plane holds 0xffffffff, a NaN, elsewhere).
// Per-key-point outputs may be null.
__global__ __launch_bounds__(ST_THREADS) void k_stereo_finalize(const vo_kp *__restrict__ kps, const int *__restrict__ nkp,
                                                               int kp_stride, int l_slot0, int r_slot0,
                                                               const uint2 *__restrict__ fwd,
                                                               const uint2 *__restrict__ bwd, int splits, float ratio,
                                                               double fx, double baseline, int32_t *__restrict__ right_idx,
                                                               float *__restrict__ disparity, float *__restrict__ depth,
                                                               uint32_t *__restrict__ plane, int w, int h,
                                                               int *__restrict__ n_matched)
{
    const int z = blockIdx.y;
    const int nl = min(nkp[l_slot0 + z], kp_stride);
    const vo_kp *lk = kps + (size_t)(l_slot0 + z) * kp_stride, *rk = kps + (size_t)(r_slot0 + z) * kp_stride;
    fwd += (size_t)z * kp_stride * splits;
    bwd += (size_t)z * kp_stride * splits;
    const int i = blockIdx.x * ST_THREADS + threadIdx.x;
    bool ok = false;
    if (i < nl) {
        uint32_t k0, k1;
        merge_top2(fwd + (size_t)i * splits, splits, k0, k1);
        int j = -1;
        float d = -1.f, zz = __uint_as_float(0x7fc00000u);
        if (k0 != ST_NONE) {
            j = (int)(k0 & 0xffffu);
            ok = k1 == ST_NONE || (float)(k0 >> 16) < ratio * (float)(k1 >> 16);
            if (ok) {
                uint32_t b0, b1;
                merge_top2(bwd + (size_t)j * splits, splits, b0, b1);
                ok = b0 != ST_NONE && (int)(b0 & 0xffffu) == i;
            }
        }
        if (ok) {
            d = lk[i].x - rk[j].x;
            zz = (float)(fx * baseline / (double)d);
            const int u = (int)roundf(lk[i].x), v = (int)roundf(lk[i].y);
            if (plane && u >= 0 && u < w && v >= 0 && v < h)
                atomicMin(&plane[(size_t)z * w * h + (size_t)v * w + u], __float_as_uint(zz));
        } else {
            j = -1;
        }
        if (right_idx) right_idx[(size_t)z * kp_stride + i] = j;
        if (disparity) disparity[(size_t)z * kp_stride + i] = d;
        if (depth) depth[(size_t)z * kp_stride + i] = zz;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m && n_matched) atomicAdd(&n_matched[z], __popcll(m));
}

// Restore the all-NaN state of the depth planes: every pixel a left key point of the batch rounds to.
__global__ __launch_bounds__(ST_THREADS) void k_stereo_clear(const vo_kp *__restrict__ kps, const int *__restrict__ nkp,
                                                            int kp_stride, int l_slot0, uint32_t *__restrict__ plane,
                                                            int w, int h)
{
    const int z = blockIdx.y;
    const int nl = min(nkp[l_slot0 + z], kp_stride);
    const int i = blockIdx.x * ST_THREADS + threadIdx.x;
    if (i >= nl) return;
    const vo_kp kp = kps[(size_t)(l_slot0 + z) * kp_stride + i];
    const int u = (int)roundf(kp.x), v = (int)roundf(kp.y);
    if (u >= 0 && u < w && v >= 0 && v < h) plane[(size_t)z * w * h + (size_t)v * w + u] = 0xffffffffu;
}

}  // namespace

void launch_stereo_match(const vo_kp *kps, const uint8_t *desc, const int *nkp, int kp_stride, int l_slot0,
                         int r_slot0, int frames, float band, float min_disp, float max_disp, float ratio, double fx,
                         double baseline, uint16_t *order, int *row_start, uint2 *fwd, uint2 *bwd, int32_t *right_idx,
                         float *disparity, float *depth, uint32_t *plane, int w, int h, int *n_matched, hipStream_t s)
{
    const int qb = (kp_stride + ST_THREADS - 1) / ST_THREADS;
    // row lists of the right key points (first half of order / row_start), then of the left ones
    uint16_t *order_l = order + (size_t)frames * kp_stride;
    int *rows_l = row_start + (size_t)frames * (SI_ROWS + 1);
    hipLaunchKernelGGL(k_row_index, dim3(frames), dim3(SI_THREADS), 0, s, kps, nkp, kp_stride, r_slot0, order,
                       row_start);
    hipLaunchKernelGGL(k_row_index, dim3(frames), dim3(SI_THREADS), 0, s, kps, nkp, kp_stride, l_slot0, order_l,
                       rows_l);
    hipLaunchKernelGGL(k_band_match, dim3(qb, frames), dim3(ST_THREADS), 0, s, kps, desc, nkp, kp_stride, l_slot0,
                       r_slot0, band, min_disp, max_disp, +1, order, row_start, fwd);
    hipLaunchKernelGGL(k_band_match, dim3(qb, frames), dim3(ST_THREADS), 0, s, kps, desc, nkp, kp_stride, r_slot0,
                       l_slot0, band, min_disp, max_disp, -1, order_l, rows_l, bwd);
    hipLaunchKernelGGL(k_stereo_finalize, dim3(qb, frames), dim3(ST_THREADS), 0, s, kps, nkp, kp_stride, l_slot0,
                       r_slot0, fwd, bwd, /*splits=*/1, ratio, fx, baseline, right_idx, disparity, depth, plane, w, h,
                       n_matched);
}

size_t stereo_index_bytes(int frames, int kp_stride, size_t *rows_off)
{
    const size_t order = (sizeof(uint16_t) * 2 * (size_t)frames * kp_stride + 255) & ~size_t(255);
    *rows_off = order;
    return order + sizeof(int) * 2 * (size_t)frames * (SI_ROWS + 1);
}

void launch_stereo_clear(const vo_kp *kps, const int *nkp, int kp_stride, int l_slot0, int frames, uint32_t *plane,
                         int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_stereo_clear, dim3((kp_stride + ST_THREADS - 1) / ST_THREADS, frames), dim3(ST_THREADS), 0, s,
                       kps, nkp, kp_stride, l_slot0, plane, w, h);
}

}  // namespace vo
