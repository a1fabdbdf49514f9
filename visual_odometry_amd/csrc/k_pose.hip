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
constexpr int TA_PER_THREAD = (VO_MAX_FEATURES_LIMIT + TA_THREADS - 1) / TA_THREADS;  // key points per thread
constexpr int TA_MAX_FRAMES = 64;  // frames per stream whose scalars are staged at once
static_assert(VO_MAX_FEATURES_LIMIT <= 8192, "match and key point indices are packed as 13 + 13 bits");

// For frame f of stream s (batch index b = s*frames + f):
//   age[curr][train] = age[prev][query] + 1 for every match, later matches
//   overwriting earlier ones, all other ages 0; flag[b] = 1 when more than 80
//   matches exist and at least 50 of them have age >= 3.
// The frames of a stream depend on each other through the ages only, so the
// chain a frame waits on is kept short: the per-frame scalars of a group of
// frames are staged in LDS before the loop, the ages of the newest frame stay
// in LDS for the next one, a frame's matches are loaded (into registers) while
// the frame before it is still being finished, and the winning match of a key
// point carries its query index (key = match index << 13 | query), so nothing
// is read back from global memory inside the loop.
__global__ __launch_bounds__(TA_THREADS) void k_track_ages(const vo_match *__restrict__ final_m,
                                                           const int *__restrict__ info, int kp_stride,
                                                           const int *__restrict__ q_slot,
                                                           const int *__restrict__ t_slot,
                                                           const int *__restrict__ has_prev, int frames,
                                                           const int *__restrict__ nkp, int *__restrict__ ages,
                                                           int *__restrict__ filter_flag, unsigned long long *stamp)
{
    // (optional: the device wall clock at the start of the launch = the end of the matching stages)
    if (stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp = wall_clock64();
    __shared__ int s_winner[VO_MAX_FEATURES_LIMIT];
    __shared__ int s_age[VO_MAX_FEATURES_LIMIT];  // ages of the frame finished last (slot s_age_slot)
    __shared__ int s_ts[TA_MAX_FRAMES], s_qs[TA_MAX_FRAMES], s_hp[TA_MAX_FRAMES], s_n[TA_MAX_FRAMES],
        s_ncurr[TA_MAX_FRAMES], s_nprev[TA_MAX_FRAMES], s_cnt[TA_MAX_FRAMES];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;

    for (int i = tid; i < kp_stride; i += TA_THREADS) s_winner[i] = -1;
    int age_slot = -1;  // the slot whose ages s_age holds (block-uniform)

    for (int g0 = 0; g0 < frames; g0 += TA_MAX_FRAMES) {
        const int gn = min(frames - g0, TA_MAX_FRAMES);
        __syncthreads();  // the previous group's scalars are no longer read
        if (tid < gn) {
            const int b = s * frames + g0 + tid;
            const int ts = t_slot[b], hp = has_prev[b], qs = hp ? q_slot[b] : ts;
            s_ts[tid] = ts; s_qs[tid] = qs; s_hp[tid] = hp;
            s_n[tid] = hp ? min(info[b * 4 + 3], kp_stride) : 0;
            s_ncurr[tid] = nkp[ts]; s_nprev[tid] = nkp[qs];
            s_cnt[tid] = 0;
        }
        __syncthreads();

        // matches of a frame as (train, query) per thread; train = -1: none
        int tr[TA_PER_THREAD], qu[TA_PER_THREAD];
        auto load_matches = [&](int fl) {
            const vo_match *m = final_m + (size_t)(s * frames + g0 + fl) * kp_stride;
            const int n = s_n[fl];
#pragma unroll
            for (int k = 0; k < TA_PER_THREAD; k++) {
                const int i = tid + k * TA_THREADS;
                tr[k] = -1; qu[k] = 0;
                if (i < n) { tr[k] = m[i].train; qu[k] = m[i].query; }
            }
        };
        load_matches(0);

        for (int fl = 0; fl < gn; fl++) {
            const int ts = s_ts[fl], qs = s_qs[fl], n_curr = s_ncurr[fl], n_prev = s_nprev[fl];
            const bool hp = s_hp[fl] != 0;
            int *age_c = ages + (size_t)ts * kp_stride;
            // the ages of the query frame: in LDS already when it is the frame finished last
            if (hp && qs != age_slot) {  // block-uniform
                __syncthreads();  // the frame before may still be counting over s_age
                const int *age_p = ages + (size_t)qs * kp_stride;
                for (int i = tid; i < kp_stride; i += TA_THREADS) s_age[i] = age_p[i];
            }
            if (hp) {
#pragma unroll
                for (int k = 0; k < TA_PER_THREAD; k++) {
                    const int t = tr[k], q = qu[k];
                    if (t >= 0 && t < n_curr) {
                        const int qf = (q >= 0 && q < n_prev) ? q : 8191;  // 8191: no such key point
                        atomicMax(&s_winner[t], ((tid + k * TA_THREADS) << 13) | qf);
                    }
                }
            }
            __syncthreads();
            int a[TA_PER_THREAD];
#pragma unroll
            for (int k = 0; k < TA_PER_THREAD; k++) {
                const int i = tid + k * TA_THREADS;
                a[k] = 0;
                if (hp && i < kp_stride) {
                    const int w = s_winner[i];
                    if (w >= 0) {
                        const int q = w & 8191;
                        a[k] = (q < n_prev ? s_age[q] : 0) + 1;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TA_PER_THREAD; k++) {
                const int i = tid + k * TA_THREADS;
                if (i < kp_stride) { s_age[i] = a[k]; age_c[i] = a[k]; s_winner[i] = -1; }
            }
            age_slot = ts;
            __syncthreads();
            int mature = 0;
            if (hp) {
#pragma unroll
                for (int k = 0; k < TA_PER_THREAD; k++) {
                    const int t = tr[k];
                    mature += (t >= 0 && t < n_curr && s_age[t] >= 3);
                }
            }
            // the next frame's matches are on their way while this frame's count is summed
            if (fl + 1 < gn) load_matches(fl + 1);
            for (int o = 32; o > 0; o >>= 1) mature += __shfl_xor(mature, o);
            if (lane == 0 && mature) atomicAdd(&s_cnt[fl], mature);
        }
        __syncthreads();
        if (tid < gn) filter_flag[s * frames + g0 + tid] = (s_hp[tid] && s_n[tid] > 80 && s_cnt[tid] >= 50) ? 1 : 0;
    }
}

void launch_track_ages(const vo_match *final_m, const int *info, int kp_stride, const int *q_slot,
                       const int *t_slot, const int *has_prev, int streams, int frames, const int *nkp,
                       int *ages, int *filter_flag, hipStream_t s, unsigned long long *stamp)
{
    hipLaunchKernelGGL(k_track_ages, dim3(streams), dim3(TA_THREADS), 0, s, final_m, info, kp_stride, q_slot,
                       t_slot, has_prev, frames, nkp, ages, filter_flag, stamp);
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
    int *__restrict__ out_info, int *__restrict__ clear3, int depth_u16)
{
    __shared__ int s_wave[BP_THREADS / kWave];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // three words per frame that the next stage wants zeroed ([3][batch], see launch_pnp_ransac)
    if (clear3 && tid < 3) clear3[tid * gridDim.x + b] = 0;
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
    // (depth_u16: the planes hold uint16 millimetres, 0 = no measurement; strides are in elements either way)
    const float *dimg = depth ? (depth_u16 ? reinterpret_cast<const float *>(reinterpret_cast<const uint16_t *>(depth) +
                                                                                b * depth_frame_stride)
                                           : depth + b * depth_frame_stride)
                              : nullptr;
    for (int i0 = 0; i0 < n_tr; i0 += BP_THREADS) {
        const int i = i0 + tid;
        bool keep = false;
        vo_match mm;
        float X = 0, Y = 0, z = 0;
        if (i < n_tr && dimg) {
            mm = tr[i];
            const int u = (int)roundf(mm.cx), v = (int)roundf(mm.cy);
            if (u >= 0 && u < width && v >= 0 && v < height) {
                z = depth_u16 ? (float)reinterpret_cast<const uint16_t *>(dimg)[(size_t)v * depth_stride + u]
                              : dimg[(size_t)v * depth_stride + u];
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
                               hipStream_t s, int *clear3, bool depth_u16)
{
    hipLaunchKernelGGL(k_filter_backproject, dim3(batch), dim3(BP_THREADS), 0, s, final_m, info, kp_stride,
                       t_slot, has_prev, nkp, ages, filter_flag, depth, width, height, depth_stride,
                       depth_frame_stride, fx, fy, cx, cy, tracked, valid, pts3, out_info, clear3, depth_u16 ? 1 : 0);
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

// One block per frame: everything a caller reads back after a batch, gathered into one block of `out` so that a
// single device-to-host copy delivers it.  All records are whole 32-bit words.
__global__ __launch_bounds__(256) void k_pack_frames(PackSrc p, PackLayout L, uint8_t *__restrict__ out)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    uint8_t *dst = out + (size_t)f * L.frame_bytes;
    const vo_frame_result r = p.res[f];
    uint32_t *hdr = reinterpret_cast<uint32_t *>(dst);
    const uint32_t *rs = reinterpret_cast<const uint32_t *>(p.res + f);
    if (tid < (int)(sizeof(vo_frame_result) / 4)) hdr[tid] = rs[tid];
    if (tid < 6) hdr[sizeof(vo_frame_result) / 4 + tid] = reinterpret_cast<const uint32_t *>(p.stamps)[tid];
    const int slot = p.t_slot[f];
    auto copy_words = [&](size_t off, const void *src, int words) {
        uint32_t *d = reinterpret_cast<uint32_t *>(dst + off);
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
        for (int i = tid; i < words; i += 256) d[i] = s[i];
    };
    const int nk = min(r.n_keypoints, L.cap);
    copy_words(L.off_kps, p.kps + (size_t)slot * p.kp_stride, nk * (int)(sizeof(vo_kp) / 4));
    copy_words(L.off_desc, p.desc + (size_t)slot * p.kp_stride * 32, nk * 8);
    const int nm[3] = {min(r.n_matches, L.cap), min(r.n_tracked, L.cap), min(r.n_3d, L.cap)};
#pragma unroll
    for (int w = 0; w < 3; w++)
        copy_words(L.off_m[w], p.m[w] + (size_t)f * p.kp_stride, nm[w] * (int)(sizeof(vo_match) / 4));
    copy_words(L.off_pts3, p.pts3 + (size_t)f * p.kp_stride * 3, nm[2] * 3);
    copy_words(L.off_inl, p.inliers + (size_t)f * p.kp_stride, min(r.pnp_inliers, L.cap));
}

void launch_pack_frames(const PackSrc &src, PackLayout layout, uint8_t *out, int batch, hipStream_t s)
{
    hipLaunchKernelGGL(k_pack_frames, dim3(batch), dim3(256), 0, s, src, layout, out);
}

// The end of a batch with packed output in ONE launch: blocks [0, batch) gather their frame's result record (what
// k_collect does) and pack the frame (k_pack_frames); the blocks behind them copy each stream's last frame into its
// carry slot (k_carry).  The three touch disjoint data.
__global__ __launch_bounds__(256) void k_finish_packed(CollectSrc cs, PackSrc p, PackLayout L, uint8_t *__restrict__ out,
                                                       int batch, vo_kp *__restrict__ kps, uint8_t *__restrict__ desc,
                                                       int *__restrict__ nkp, int *__restrict__ ages, int streams,
                                                       int frames, int carry_blocks_x)
{
    __shared__ vo_frame_result s_r;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= batch) {
        const int cb = (int)blockIdx.x - batch;
        const int s = cb / carry_blocks_x, bx = cb - s * carry_blocks_x;
        const int dst = s, src = streams + s * frames + frames - 1;
        const int n = nkp[src];
        const int i = bx * 256 + tid;
        const int kp_stride = p.kp_stride;
        if (i == 0) nkp[dst] = n;
        if (i < kp_stride) ages[(size_t)dst * kp_stride + i] = ages[(size_t)src * kp_stride + i];
        if (i < n) {
            kps[(size_t)dst * kp_stride + i] = kps[(size_t)src * kp_stride + i];
            const uint4 *sd = reinterpret_cast<const uint4 *>(desc + ((size_t)src * kp_stride + i) * 32);
            uint4 *dd = reinterpret_cast<uint4 *>(desc + ((size_t)dst * kp_stride + i) * 32);
            dd[0] = sd[0];
            dd[1] = sd[1];
        }
        return;
    }
    const int f = blockIdx.x;
    if (tid == 0) {
        vo_frame_result r;
        r.n_keypoints = cs.nkp[cs.t_slot[f]];
        r.n_ratio = cs.n_good[f];
        r.f_found = cs.info[f * 4]; r.f_iters = cs.info[f * 4 + 1]; r.f_inliers = cs.info[f * 4 + 2];
        r.n_matches = cs.info[f * 4 + 3];
        r.n_tracked = cs.bp_info[f * 2]; r.n_3d = cs.bp_info[f * 2 + 1];
        r.pnp_success = cs.pose[f * 4]; r.pnp_inliers = cs.pose[f * 4 + 1]; r.pnp_iters = cs.pose[f * 4 + 2];
        r.pnp_points = cs.pose[f * 4 + 3];
        for (int i = 0; i < 3; i++) { r.rvec[i] = cs.rt[f * 6 + i]; r.tvec[i] = cs.rt[f * 6 + 3 + i]; }
        s_r = r;
        cs.out[f] = r;
    }
    __syncthreads();
    uint8_t *dst = out + (size_t)f * L.frame_bytes;
    const vo_frame_result r = s_r;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(dst);
    const uint32_t *rs = reinterpret_cast<const uint32_t *>(&s_r);
    if (tid < (int)(sizeof(vo_frame_result) / 4)) hdr[tid] = rs[tid];
    if (tid < 6) hdr[sizeof(vo_frame_result) / 4 + tid] = reinterpret_cast<const uint32_t *>(p.stamps)[tid];
    const int slot = p.t_slot[f];
    auto copy_words = [&](size_t off, const void *src, int words) {
        uint32_t *d = reinterpret_cast<uint32_t *>(dst + off);
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
        for (int i = tid; i < words; i += 256) d[i] = s[i];
    };
    const int nk = min(r.n_keypoints, L.cap);
    copy_words(L.off_kps, p.kps + (size_t)slot * p.kp_stride, nk * (int)(sizeof(vo_kp) / 4));
    copy_words(L.off_desc, p.desc + (size_t)slot * p.kp_stride * 32, nk * 8);
    const int nm[3] = {min(r.n_matches, L.cap), min(r.n_tracked, L.cap), min(r.n_3d, L.cap)};
#pragma unroll
    for (int w = 0; w < 3; w++)
        copy_words(L.off_m[w], p.m[w] + (size_t)f * p.kp_stride, nm[w] * (int)(sizeof(vo_match) / 4));
    copy_words(L.off_pts3, p.pts3 + (size_t)f * p.kp_stride * 3, nm[2] * 3);
    copy_words(L.off_inl, p.inliers + (size_t)f * p.kp_stride, min(r.pnp_inliers, L.cap));
}

void launch_finish_packed(const CollectSrc &cs, const PackSrc &src, PackLayout layout, uint8_t *out, int batch,
                          vo_kp *kps, uint8_t *desc, int *nkp, int *ages, int streams, int frames, hipStream_t s)
{
    const int bx = (src.kp_stride + 255) / 256;
    hipLaunchKernelGGL(k_finish_packed, dim3(batch + bx * streams), dim3(256), 0, s, cs, src, layout, out, batch, kps,
                       desc, nkp, ages, streams, frames, bx);
}

void launch_collect(const int *nkp, const int *t_slot, const int *n_good, const int *info,
                    const int *bp_info, const int *pose, const double *rt, vo_frame_result *out, int batch,
                    hipStream_t s)
{
    hipLaunchKernelGGL(k_collect, dim3((batch + 63) / 64), dim3(64), 0, s, nkp, t_slot, n_good, info, bp_info,
                       pose, rt, out, batch);
}

}  // namespace vo
