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
constexpr int ST_TILE = 128;    // train key points staged per step
constexpr uint32_t ST_NONE = 0xffffffffu;

// Two smallest keys (distance << 16 | train index) of every query over the
// train slice of this block, for frame blockIdx.z of a batch: queries are the
// key points of feature slot q_slot0 + z, trains those of t_slot0 + z.
// sign = +1: queries are left key points (disparity = xq - xt); -1: queries are
// right ones.  partial: frames x kp_stride x splits.
__global__ __launch_bounds__(ST_THREADS) void k_band_knn2(const vo_kp *__restrict__ kps, const uint8_t *__restrict__ desc,
                                                         const int *__restrict__ nkp, int kp_stride, int q_slot0,
                                                         int t_slot0, float band, float min_disp, float max_disp,
                                                         int sign, int splits, uint2 *__restrict__ partial)
{
    __shared__ uint4 s_desc[ST_TILE][2];
    __shared__ float2 s_xy[ST_TILE];
    const int z = blockIdx.z;
    const int nq = min(nkp[q_slot0 + z], kp_stride), nt = min(nkp[t_slot0 + z], kp_stride);
    if ((int)(blockIdx.x * ST_THREADS) >= nq) return;  // block-uniform
    const vo_kp *qk = kps + (size_t)(q_slot0 + z) * kp_stride, *tk = kps + (size_t)(t_slot0 + z) * kp_stride;
    const uint8_t *qd = desc + (size_t)(q_slot0 + z) * kp_stride * 32, *td = desc + (size_t)(t_slot0 + z) * kp_stride * 32;
    partial += (size_t)z * kp_stride * splits;
    const int tid = threadIdx.x, q = blockIdx.x * ST_THREADS + tid, split = blockIdx.y;
    const int per = (nt + splits - 1) / splits;
    const int t_lo = split * per, t_hi = min(nt, t_lo + per);
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    float qx = 0.f, qy = 0.f;
    if (q < nq) {
        a0 = reinterpret_cast<const uint4 *>(qd)[2 * q];
        a1 = reinterpret_cast<const uint4 *>(qd)[2 * q + 1];
        qx = qk[q].x; qy = qk[q].y;
    }
    uint32_t k0 = ST_NONE, k1 = ST_NONE;
    for (int base = t_lo; base < t_hi; base += ST_TILE) {
        const int cnt = min(ST_TILE, t_hi - base);
        __syncthreads();
        if (tid < 2 * cnt) (&s_desc[0][0])[tid] = reinterpret_cast<const uint4 *>(td)[2 * (size_t)base + tid];
        if (tid < cnt) s_xy[tid] = make_float2(tk[base + tid].x, tk[base + tid].y);
        __syncthreads();
        if (q >= nq) continue;
        for (int j = 0; j < cnt; j++) {
            const float2 t = s_xy[j];
            const float dy = qy - t.y;
            const float dx = sign > 0 ? qx - t.x : t.x - qx;
            if (!(fabsf(dy) <= band) || !(dx >= min_disp) || !(dx <= max_disp)) continue;
            const uint4 b0 = s_desc[j][0], b1 = s_desc[j][1];
            uint32_t d = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w);
            d += __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            const uint32_t key = (d << 16) | (uint32_t)(base + j);
            const uint32_t hi = max(k0, key);
            k0 = min(k0, key);
            k1 = min(k1, hi);
        }
    }
    if (q < nq) partial[(size_t)q * splits + split] = make_uint2(k0, k1);
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
                         double baseline, int splits, uint2 *fwd, uint2 *bwd, int32_t *right_idx, float *disparity,
                         float *depth, uint32_t *plane, int w, int h, int *n_matched, hipStream_t s)
{
    const int qb = (kp_stride + ST_THREADS - 1) / ST_THREADS;
    hipLaunchKernelGGL(k_band_knn2, dim3(qb, splits, frames), dim3(ST_THREADS), 0, s, kps, desc, nkp, kp_stride,
                       l_slot0, r_slot0, band, min_disp, max_disp, +1, splits, fwd);
    hipLaunchKernelGGL(k_band_knn2, dim3(qb, splits, frames), dim3(ST_THREADS), 0, s, kps, desc, nkp, kp_stride,
                       r_slot0, l_slot0, band, min_disp, max_disp, -1, splits, bwd);
    hipLaunchKernelGGL(k_stereo_finalize, dim3(qb, frames), dim3(ST_THREADS), 0, s, kps, nkp, kp_stride, l_slot0,
                       r_slot0, fwd, bwd, splits, ratio, fx, baseline, right_idx, disparity, depth, plane, w, h,
                       n_matched);
}

void launch_stereo_clear(const vo_kp *kps, const int *nkp, int kp_stride, int l_slot0, int frames, uint32_t *plane,
                         int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_stereo_clear, dim3((kp_stride + ST_THREADS - 1) / ST_THREADS, frames), dim3(ST_THREADS), 0, s,
                       kps, nkp, kp_stride, l_slot0, plane, w, h);
}

}  // namespace vo
