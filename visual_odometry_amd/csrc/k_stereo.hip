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
// train slice of this block.  sign = +1: queries are left key points
// (disparity = xq - xt); -1: queries are right ones.  partial: nq x splits.
__global__ __launch_bounds__(ST_THREADS) void k_band_knn2(const vo_kp *__restrict__ qk, const uint8_t *__restrict__ qd,
                                                         int nq, const vo_kp *__restrict__ tk,
                                                         const uint8_t *__restrict__ td, int nt, float band,
                                                         float min_disp, float max_disp, int sign, int splits,
                                                         uint2 *__restrict__ partial)
{
    __shared__ uint4 s_desc[ST_TILE][2];
    __shared__ float2 s_xy[ST_TILE];
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

// One thread per left key point: ratio test, cross-check, disparity, depth, and
// the sparse depth plane (bits of a positive float order like the value, so an
// unsigned atomic minimum keeps the nearest depth where key points share a pixel;
// the plane is pre-filled with 0xffffffff, a NaN).
__global__ __launch_bounds__(ST_THREADS) void k_stereo_finalize(const vo_kp *__restrict__ lk, const vo_kp *__restrict__ rk,
                                                               int nl, const uint2 *__restrict__ fwd,
                                                               const uint2 *__restrict__ bwd, int splits, float ratio,
                                                               double fx, double baseline, int32_t *__restrict__ right_idx,
                                                               float *__restrict__ disparity, float *__restrict__ depth,
                                                               uint32_t *__restrict__ plane, int w, int h,
                                                               int *__restrict__ n_matched)
{
    const int i = blockIdx.x * ST_THREADS + threadIdx.x;
    bool ok = false;
    if (i < nl) {
        uint32_t k0, k1;
        merge_top2(fwd + (size_t)i * splits, splits, k0, k1);
        int j = -1;
        float d = -1.f, z = __uint_as_float(0x7fc00000u);
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
            z = (float)(fx * baseline / (double)d);
            const int u = (int)rintf(lk[i].x), v = (int)rintf(lk[i].y);
            if (plane && u >= 0 && u < w && v >= 0 && v < h) atomicMin(&plane[(size_t)v * w + u], __float_as_uint(z));
        } else {
            j = -1;
        }
        right_idx[i] = j;
        disparity[i] = d;
        depth[i] = z;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_matched, __popcll(m));
}

}  // namespace

void launch_stereo_match(const vo_kp *lk, const uint8_t *ld, int nl, const vo_kp *rk, const uint8_t *rd, int nr,
                         float band, float min_disp, float max_disp, float ratio, double fx, double baseline,
                         int splits, uint2 *fwd, uint2 *bwd, int32_t *right_idx, float *disparity, float *depth,
                         uint32_t *plane, int w, int h, int *n_matched, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_knn2, dim3((nl + ST_THREADS - 1) / ST_THREADS, splits), dim3(ST_THREADS), 0, s, lk, ld, nl,
                       rk, rd, nr, band, min_disp, max_disp, +1, splits, fwd);
    hipLaunchKernelGGL(k_band_knn2, dim3((nr + ST_THREADS - 1) / ST_THREADS, splits), dim3(ST_THREADS), 0, s, rk, rd, nr,
                       lk, ld, nl, band, min_disp, max_disp, -1, splits, bwd);
    hipLaunchKernelGGL(k_stereo_finalize, dim3((nl + ST_THREADS - 1) / ST_THREADS), dim3(ST_THREADS), 0, s, lk, rk, nl,
                       fwd, bwd, splits, ratio, fx, baseline, right_idx, disparity, depth, plane, w, h, n_matched);
}

}  // namespace vo
