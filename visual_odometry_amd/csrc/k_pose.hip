// Glue stages of the frame pipeline for gfx950: track-age bookkeeping and the
// multi-view filter (reference src/frame_processor.cpp:44-92), depth
// back-projection (reference src/frame_processor.cpp:230-308), and the
// end-of-batch carry / result collection.
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
