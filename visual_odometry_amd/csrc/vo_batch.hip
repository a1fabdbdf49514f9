// C ABI, part 2: cv::solvePnPRansac operator and the batched pipeline
// (FrameProcessor::processFrame for many frames / streams per call).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "vo_ctx.h"

using namespace vo;

static int bfail(vo_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ctx->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return bfail(ctx, VO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                                     \
    } while (0)

static int ensure_pack(vo_ctx *c, int B);

static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// host memory the GPU can read by DMA without a staging pass (hipHostMalloc / hipHostRegister): copies from it
// return at once, copies from ordinary memory occupy the calling thread until the data has left it
static bool is_pinned_host_ptr(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int vo_ctx::ensure_essential(int n_points, int n_hyp)
{
    vo_ctx *c = this;
    if (n_points > e_pts_cap) {
        if (d_eq) (void)hipFree(d_eq);
        if (d_emask) (void)hipFree(d_emask);
        d_eq = nullptr; d_emask = nullptr; e_pts_cap = 0;
        HIP_TRY(c, hipMalloc(&d_eq, sizeof(double) * 4 * (size_t)n_points));
        HIP_TRY(c, hipMalloc(&d_emask, 2 * (size_t)n_points));  // inlier mask, then per-candidate votes
        e_pts_cap = n_points;
    }
    if (!d_einfo) HIP_TRY(c, hipMalloc(&d_einfo, sizeof(int) * 4));
    if (n_hyp > e_hyp_cap) {
        for (void *p : {(void *)d_eE, (void *)d_eidx, (void *)d_enm, (void *)d_ecnt})
            if (p) (void)hipFree(p);
        d_eE = nullptr; d_eidx = nullptr; d_enm = nullptr; d_ecnt = nullptr; e_hyp_cap = 0;
        HIP_TRY(c, hipMalloc(&d_eE, sizeof(double) * 90 * (size_t)n_hyp));
        HIP_TRY(c, hipMalloc(&d_eidx, sizeof(int) * 5 * (size_t)n_hyp));
        HIP_TRY(c, hipMalloc(&d_enm, sizeof(int) * (size_t)n_hyp));
        HIP_TRY(c, hipMalloc(&d_ecnt, sizeof(int) * 10 * (size_t)n_hyp));
        e_hyp_cap = n_hyp;
    }
    return VO_OK;
}

// pixel points -> normalised (x1, y1, x2, y2) quadruples on the device
static int upload_normalised(vo_ctx *c, const float *p1, const float *p2, int n, double fx, double fy, double cx,
                             double cy)
{
    std::vector<float> a(2 * (size_t)n), b(2 * (size_t)n);
    HIP_TRY(c, hipMemcpy(a.data(), p1, sizeof(float) * 2 * n, hipMemcpyDefault));
    HIP_TRY(c, hipMemcpy(b.data(), p2, sizeof(float) * 2 * n, hipMemcpyDefault));
    std::vector<double> q(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        q[4 * i] = ((double)a[2 * i] - cx) / fx; q[4 * i + 1] = ((double)a[2 * i + 1] - cy) / fy;
        q[4 * i + 2] = ((double)b[2 * i] - cx) / fx; q[4 * i + 3] = ((double)b[2 * i + 1] - cy) / fy;
    }
    HIP_TRY(c, hipMemcpy(c->d_eq, q.data(), sizeof(double) * 4 * n, hipMemcpyHostToDevice));
    return VO_OK;
}

extern "C" {

int vo_find_essential_ransac(vo_ctx *c, const float *p1, const float *p2, int n, double fx, double fy, double cx,
                             double cy, double prob, double threshold, int max_iters, uint8_t *mask, double *E,
                             int *found, int *iters_run)
{
    if (!c) return VO_ERR_ARG;
    if (!p1 || !p2 || !mask || n < 0 || n > 65535 || !(fx > 0 && fy > 0) || !(threshold > 0) || !(prob > 0 && prob < 1))
        return bfail(c, VO_ERR_ARG, "vo_find_essential_ransac: bad arguments (n %d)", n);
    if (found) *found = 0;
    if (iters_run) *iters_run = 0;
    if (n < 5) {  // OpenCV asserts on fewer than five points
        if (n > 0) memset(mask, 0, (size_t)n);
        return VO_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = c->ensure_essential(n, 1);
    if (rc) return rc;
    if ((rc = upload_normalised(c, p1, p2, n, fx, fy, cx, cy))) return rc;
    const double thr = threshold / ((fx + fy) / 2);
    launch_e_ransac(c->d_eq, n, c->d_rng, c->rng_len, prob, (float)(thr * thr), max_iters, c->d_emask, c->d_eE,
                    c->d_einfo, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_find_essential_ransac: %s", hipGetErrorString(e));
    int info[4];
    HIP_TRY(c, hipMemcpy(info, c->d_einfo, sizeof(int) * 3, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(mask, c->d_emask, (size_t)n, hipMemcpyDefault));
    if (E) HIP_TRY(c, hipMemcpy(E, c->d_eE, sizeof(double) * 9, hipMemcpyDefault));
    if (found) *found = info[0];
    if (iters_run) *iters_run = info[1];
    return VO_OK;
}

int vo_essential_hypotheses(vo_ctx *c, const float *p1, const float *p2, int n, double fx, double fy, double cx,
                            double cy, const int32_t *samples, int n_hyp, double threshold, double *E,
                            int32_t *n_models, int32_t *counts)
{
    if (!c) return VO_ERR_ARG;
    if (!p1 || !p2 || !samples || !E || !n_models || !counts || n < 5 || n_hyp < 1 || !(fx > 0 && fy > 0) ||
        !(threshold > 0))
        return bfail(c, VO_ERR_ARG, "vo_essential_hypotheses: bad arguments (n %d, hypotheses %d)", n, n_hyp);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = c->ensure_essential(n, n_hyp);
    if (rc) return rc;
    if ((rc = upload_normalised(c, p1, p2, n, fx, fy, cx, cy))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_eidx, samples, sizeof(int) * 5 * (size_t)n_hyp, hipMemcpyDefault));
    HIP_TRY(c, hipMemsetAsync(c->d_eE, 0, sizeof(double) * 90 * (size_t)n_hyp, c->stream));
    const double thr = threshold / ((fx + fy) / 2);
    launch_e_hypotheses(c->d_eq, n, c->d_eidx, n_hyp, c->d_eE, c->d_enm, (float)(thr * thr), c->d_ecnt, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_essential_hypotheses: %s", hipGetErrorString(e));
    HIP_TRY(c, hipMemcpy(E, c->d_eE, sizeof(double) * 90 * (size_t)n_hyp, hipMemcpyDefault));
    HIP_TRY(c, hipMemcpy(n_models, c->d_enm, sizeof(int) * (size_t)n_hyp, hipMemcpyDefault));
    HIP_TRY(c, hipMemcpy(counts, c->d_ecnt, sizeof(int) * 10 * (size_t)n_hyp, hipMemcpyDefault));
    return VO_OK;
}

int vo_recover_pose(vo_ctx *c, const double *E, const float *p1, const float *p2, int n, double fx, double fy,
                    double cx, double cy, double distance_thresh, int use_mask, uint8_t *mask, double *R, double *t,
                    int *n_good)
{
    if (!c) return VO_ERR_ARG;
    if (!E || !p1 || !p2 || !mask || !R || !t || n < 1 || !(fx > 0 && fy > 0) || !(distance_thresh > 0))
        return bfail(c, VO_ERR_ARG, "vo_recover_pose: bad arguments (n %d)", n);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = c->ensure_essential(n, 1);
    if (rc) return rc;
    if ((rc = upload_normalised(c, p1, p2, n, fx, fy, cx, cy))) return rc;
    HIP_TRY(c, hipMemcpy(c->d_eE, E, sizeof(double) * 9, hipMemcpyDefault));
    if (use_mask) HIP_TRY(c, hipMemcpy(c->d_emask, mask, (size_t)n, hipMemcpyDefault));
    launch_recover_pose(c->d_eE, c->d_eq, n, distance_thresh, use_mask ? 1 : 0, c->d_emask, c->d_emask + c->e_pts_cap,
                        c->d_eE + 16, c->d_einfo, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_recover_pose: %s", hipGetErrorString(e));
    double out[12];
    int info[2];
    HIP_TRY(c, hipMemcpy(out, c->d_eE + 16, sizeof(out), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(info, c->d_einfo, sizeof(info), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(mask, c->d_emask, (size_t)n, hipMemcpyDefault));
    memcpy(R, out, sizeof(double) * 9);
    memcpy(t, out + 9, sizeof(double) * 3);
    if (n_good) *n_good = info[0];
    return VO_OK;
}

int vo_solve_pnp_ransac(vo_ctx *c, const float *obj, const float *img, int n, double fx, double fy, double cx,
                        double cy, int iters, double thr, double conf, double *rvec, double *tvec,
                        int32_t *inliers, int *n_inliers, int *success)
{
    if (!c) return VO_ERR_ARG;
    if (!obj || !img || !rvec || !tvec || n < 0 || n > c->cfg.max_features)
        return bfail(c, VO_ERR_ARG, "vo_solve_pnp_ransac: bad arguments (n %d, limit %d)", n,
                     c->cfg.max_features);
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> o(3 * (size_t)n + 1), im(2 * (size_t)n + 1);
    if (n) {
        HIP_TRY(c, hipMemcpy(o.data(), obj, sizeof(float) * 3 * n, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(im.data(), img, sizeof(float) * 2 * n, hipMemcpyDefault));
    }
    std::vector<vo_match> vm((size_t)n + 1);
    for (int i = 0; i < n; i++) {
        vm[i].query = vm[i].train = i; vm[i].distance = 0;
        vm[i].px = im[2 * i]; vm[i].py = im[2 * i + 1];
        vm[i].cx = vm[i].cy = 0;
    }
    const int bp[2] = {n, n};
    if (n) {
        HIP_TRY(c, hipMemcpy(c->d_valid, vm.data(), sizeof(vo_match) * n, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_pts3, o.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    }
    HIP_TRY(c, hipMemcpy(c->d_bpinfo, bp, sizeof(bp), hipMemcpyHostToDevice));
    launch_pnp_ransac(c->d_valid, c->d_pts3, c->d_bpinfo, c->kp_stride, c->d_rng, c->rng_len, fx, fy, cx, cy,
                      iters, thr, conf, c->d_inliers, c->d_pose, c->d_rt, 1, c->d_pnp2, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_solve_pnp_ransac: %s", hipGetErrorString(e));
    int pose[4];
    double rt[6];
    HIP_TRY(c, hipMemcpy(pose, c->d_pose, sizeof(pose), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(rt, c->d_rt, sizeof(rt), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) { rvec[i] = rt[i]; tvec[i] = rt[3 + i]; }
    if (success) *success = pose[0];
    if (n_inliers) *n_inliers = pose[1];
    if (inliers && pose[1] > 0)
        HIP_TRY(c, hipMemcpy(inliers, c->d_inliers, sizeof(int) * pose[1], hipMemcpyDefault));
    return VO_OK;
}

int vo_batch_configure(vo_ctx *c, const vo_batch_cfg *b)
{
    if (!c) return VO_ERR_ARG;
    if (!b || b->width < 7 || b->height < 7 || b->width > c->cfg.max_width || b->height > c->cfg.max_height ||
        (b->channels != 1 && b->channels != 3 && b->channels != 4) || b->n_streams < 1 ||
        b->frames_per_stream < 1 || (long long)b->n_streams * b->frames_per_stream > c->cfg.max_batch ||
        b->max_features < 1 || b->max_features > c->cfg.max_features)
        return bfail(c, VO_ERR_ARG, "vo_batch_configure: configuration outside the context limits");
    if (b->enable_pose && !(b->fx > 0 && b->fy > 0))
        return bfail(c, VO_ERR_ARG, "vo_batch_configure: pose needs fx, fy > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (b->orb_levels > 0) {
        const int rc = c->ensure_orb(b->width, b->height, b->orb_scale_factor > 0 ? b->orb_scale_factor : 1.2f,
                                     b->orb_levels, b->max_features, b->n_streams * b->frames_per_stream);
        if (rc) return rc;
    }
    if (c->packed) {
        // before anything is committed: a batch too large for the packed output leaves the context as it was
        const int rc = ensure_pack(c, b->n_streams * b->frames_per_stream);
        if (rc) return rc;
    }
    c->bcfg = *b;
    c->configured = true;
    c->submitted = false;
    c->drop_graph();
    c->packed_valid = false;
    // the NaN fill of the stereo depth planes covers the geometry and batch of the configuration it was made under
    c->depth_planes_nan = false;
    return vo_batch_reset(c);
}

int vo_batch_reset(vo_ctx *c)
{
    if (!c) return VO_ERR_ARG;
    if (!c->configured) return bfail(c, VO_ERR_STATE, "vo_batch_reset: configure first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->streams_started = false;
    c->first_pairs_linked = false;
    const int S = c->bcfg.n_streams, F = c->bcfg.frames_per_stream, B = S * F;
    std::vector<int> q(B), t(B), hp(B);
    for (int s = 0; s < S; s++)
        for (int f = 0; f < F; f++) {
            const int b = s * F + f;
            t[b] = S + b;  // slots 0..S-1 carry each stream's previous frame
            hp[b] = f > 0;
            q[b] = f > 0 ? t[b] - 1 : c->empty_slot;
        }
    HIP_TRY(c, hipMemcpy(c->d_qslot, q.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tslot, t.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_has_prev, hp.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_nkp, 0, sizeof(int) * c->slots));
    HIP_TRY(c, hipMemset(c->d_ages, 0, sizeof(int) * (size_t)c->kp_stride * c->slots));
    HIP_TRY(c, hipMemset(c->d_pose, 0, sizeof(int) * 4 * B));
    HIP_TRY(c, hipMemset(c->d_rt, 0, sizeof(double) * 6 * B));
    HIP_TRY(c, hipMemset(c->d_bpinfo, 0, sizeof(int) * 2 * B));
    return VO_OK;
}

// buffers of the packed output for a batch of B frames
static int ensure_pack(vo_ctx *c, int B)
{
    c->pack = pack_layout(c->kp_stride);
    const size_t need = c->pack.frame_bytes * (size_t)B;
    if (need > vo_ctx::kPackMaxBytes)
        return bfail(c, VO_ERR_ARG, "packed output: %d frames x %zu bytes exceed %zu bytes", B, c->pack.frame_bytes,
                     vo_ctx::kPackMaxBytes);
    if (need > c->pack_cap_bytes) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->drop_graph();
        if (c->d_pack) (void)hipFree(c->d_pack);
        if (c->h_pack) (void)hipHostFree(c->h_pack);
        c->d_pack = nullptr; c->h_pack = nullptr; c->pack_cap_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->d_pack, need));
        HIP_TRY(c, hipHostMalloc(&c->h_pack, need));
        c->pack_cap_bytes = need;
    }
    if (!c->d_stamps) {
        HIP_TRY(c, hipMalloc(&c->d_stamps, 3 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_stamps, 0, 3 * sizeof(unsigned long long)));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0)
            khz = 100000;  // the constant 100 MHz counter of gfx9
        c->wall_clock_khz = khz;
    }
    return VO_OK;
}

// Host planes (count x rows x row_bytes, rows row_stride and planes plane_stride bytes apart) into a packed device
// area.  Packed input goes up as ONE linear copy, planes with packed rows as one copy each; only padded rows
// take the 2-D path (which the runtime executes row by row: 13 ms for one HD720 BGRA + depth pair against
// 1 ms for the two linear copies).
// With `stage` (pinned memory of count * rows * row_bytes bytes) the planes are first packed into it piece by piece,
// each piece's DMA starting while the next piece is being packed: for ordinary (pageable) memory the runtime's own
// staging alternates between the two (1 MB packed, 1 MB sent, ...: 145 us for an HD720 BGRA frame, 115 us this way).
static int upload_planes(vo_ctx *c, uint8_t *dst, const uint8_t *src, size_t row_bytes, int rows, size_t row_stride,
                         size_t plane_stride, int count, hipStream_t stream, uint8_t *stage = nullptr)
{
    const size_t plane_bytes = row_bytes * (size_t)rows;
    if (stage) {
        const size_t total = plane_bytes * (size_t)count;
        size_t done = 0, piece = 256u << 10;  // small pieces first, so that the link starts early
        while (done < total) {
            const size_t len = std::min(piece, total - done);
            if (row_stride == row_bytes && (count == 1 || plane_stride == plane_bytes)) {
                memcpy(stage + done, src + done, len);
            } else {
                size_t o = done;
                while (o < done + len) {  // row by row out of the padded source
                    const size_t pl = o / plane_bytes, in_plane = o - pl * plane_bytes;
                    const size_t row = in_plane / row_bytes, col = in_plane - row * row_bytes;
                    const size_t n = std::min(row_bytes - col, done + len - o);
                    memcpy(stage + o, src + pl * plane_stride + row * row_stride + col, n);
                    o += n;
                }
            }
            HIP_TRY(c, hipMemcpyAsync(dst + done, stage + done, len, hipMemcpyHostToDevice, stream));
            done += len;
            if (piece < (1u << 20)) piece *= 2;
        }
        return VO_OK;
    }
    if (row_stride == row_bytes && (count == 1 || plane_stride == plane_bytes)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, plane_bytes * count, hipMemcpyHostToDevice, stream));
        return VO_OK;
    }
    for (int i = 0; i < count; i++) {
        if (row_stride == row_bytes)
            HIP_TRY(c, hipMemcpyAsync(dst + i * plane_bytes, src + i * plane_stride, plane_bytes, hipMemcpyHostToDevice,
                                      stream));
        else
            HIP_TRY(c, hipMemcpy2DAsync(dst + i * plane_bytes, row_bytes, src + i * plane_stride, row_stride, row_bytes,
                                        rows, hipMemcpyHostToDevice, stream));
    }
    return VO_OK;
}

// Calling thread of vo_batch_submit only: queue the copy of every chunk of the job whose pieces are all packed -- by
// whichever thread --, in order; with `block`, wait for the chunks that are not and return when all copies are
// queued.  Returns the first HIP error (the remaining copies are then dropped).
int vo_ctx::stager_send(StageJob &j, bool block)
{
    while (j.sent < j.n_chunks) {
        const int ch = j.sent;
        const int need = std::min((int)StageJob::kChunkPieces, j.n_pieces - ch * StageJob::kChunkPieces);
        while (j.packed[ch].load(std::memory_order_acquire) < need) {
            if (!block) return j.err;
            __builtin_ia32_pause();
        }
        const size_t off = (size_t)ch * StageJob::kChunk, len = std::min(StageJob::kChunk, j.total - off);
        if (j.err == 0 && j.dst) {  // (no destination: the plane is read where it has been packed)
            const hipError_t e = hipMemcpyAsync(j.dst + off, j.stage + off, len, hipMemcpyHostToDevice, j.stream);
            if (e != hipSuccess) j.err = (int)e;
        }
        j.sent++;
    }
    return j.err;
}

// Pack pieces of a staging job into pinned memory until none is left to take.  The calling thread (`caller`) sends
// chunks as they become complete and returns when the whole job is on its way; helpers only pack.
int vo_ctx::stager_work(StageJob &j, bool caller)
{
    const size_t plane_bytes = j.row_bytes * (size_t)j.rows;
    const bool linear = j.row_stride == j.row_bytes && (j.count == 1 || j.plane_stride == plane_bytes);
    for (;;) {
        const int i = j.next.fetch_add(1, std::memory_order_relaxed);
        if (i >= j.n_pieces) break;
        const size_t done = (size_t)i * StageJob::kPiece, len = std::min(StageJob::kPiece, j.total - done);
        if (linear) {
            memcpy(j.stage + done, j.src + done, len);
        } else {
            size_t o = done;
            while (o < done + len) {  // row by row out of the padded source
                const size_t pl = o / plane_bytes, in_plane = o - pl * plane_bytes;
                const size_t row = in_plane / j.row_bytes, col = in_plane - row * j.row_bytes;
                const size_t n = std::min(j.row_bytes - col, done + len - o);
                memcpy(j.stage + o, j.src + pl * j.plane_stride + row * j.row_stride + col, n);
                o += n;
            }
        }
        j.packed[i / StageJob::kChunkPieces].fetch_add(1, std::memory_order_release);
        if (caller) (void)stager_send(j, false);
    }
    return caller ? stager_send(j, true) : 0;
}

void vo_ctx::stager_loop()
{
    (void)hipSetDevice(device);
    StagePool &w = stager;
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(w.m);
    for (;;) {
        w.cv.wait(lk, [&] { return w.quit || w.epoch != seen; });
        if (w.quit) return;
        seen = w.epoch;
        const int n = w.n_jobs;
        w.active++;
        lk.unlock();
        for (int k = 0; k < n; k++) (void)stager_work(w.jobs[k], false);
        lk.lock();
        if (--w.active == 0) w.cv.notify_all();
    }
}

// The caller has filled in jobs[0 .. n_jobs): wake the helpers (started on first use; VO_STAGE_THREADS = 0..3
// helpers, default 1: with the source in cache one helper already keeps the packing ahead of the link, and every
// further one adds its wake-up to the call; a deployment whose frames come from cold memory may want more).
void vo_ctx::stager_post(int n_jobs)
{
    StagePool &w = stager;
    static const int want = [] {
        const char *e = getenv("VO_STAGE_THREADS");
        int n = e ? atoi(e) : 1;
        const int cores = (int)std::thread::hardware_concurrency();
        if (cores > 0) n = std::min(n, cores - 1);
        return std::max(0, std::min(n, (int)StagePool::kMaxHelpers));
    }();
    while (w.n_threads < want) {
        w.th[w.n_threads] = std::thread([this] { stager_loop(); });
        w.n_threads++;
    }
    {
        std::lock_guard<std::mutex> g(w.m);
        w.n_jobs = n_jobs;
        w.epoch++;
    }
    w.cv.notify_all();
}

void vo_ctx::stager_stop()
{
    if (stager.n_threads == 0) return;
    {
        std::lock_guard<std::mutex> g(stager.m);
        stager.quit = true;
    }
    stager.cv.notify_all();
    for (int i = 0; i < stager.n_threads; i++) stager.th[i].join();
    stager.n_threads = 0;
}

// Pinned staging for batches that come from ordinary host memory: room for the frames (twice with a right image)
// and the depth planes of one batch, only while that is small (the single-stream path; a large host-fed batch
// should come from pinned memory in the first place).
static uint8_t *stage_area(vo_ctx *c, size_t bytes)
{
    if (bytes > vo_ctx::kStageMaxBytes) return nullptr;
    if (bytes > c->h_stage_bytes) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        if (hipHostMalloc(&c->h_stage, bytes) != hipSuccess) {
            (void)hipGetLastError();
            c->h_stage = nullptr;
            return nullptr;
        }
        c->h_stage_bytes = bytes;
    }
    return static_cast<uint8_t *>(c->h_stage);
}

// vo_batch_submit and vo_batch_submit_stereo: with `right` (and `st`) the depth planes come from the stereo
// matcher instead of the caller
static int batch_submit(vo_ctx *c, const uint8_t *frames, int row_stride, size_t frame_stride, const float *depth,
                        int depth_row_stride, size_t depth_frame_stride, const uint8_t *right,
                        const vo_stereo_cfg *st, bool depth_u16 = false)
{
    // (depth_u16: `depth` really points at uint16 millimetres; strides stay in elements)
    const size_t dsz = depth_u16 ? sizeof(uint16_t) : sizeof(float);
    if (!c) return VO_ERR_ARG;
    if (!c->configured) return bfail(c, VO_ERR_STATE, "vo_batch_submit: configure first");
    if (c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_submit: previous batch not waited for");
    const vo_batch_cfg &b = c->bcfg;
    const int S = b.n_streams, F = b.frames_per_stream, B = S * F;
    const int row_bytes = b.width * b.channels;
    if (!frames || row_stride < row_bytes || frame_stride < (size_t)row_stride * b.height)
        return bfail(c, VO_ERR_ARG, "vo_batch_submit: bad frame strides");
    if (depth && (depth_row_stride < b.width || depth_frame_stride < (size_t)depth_row_stride * b.height))
        return bfail(c, VO_ERR_ARG, "vo_batch_submit: bad depth strides");
    if (right) {
        if (!st || !(st->baseline > 0) || !(st->min_disparity > 0) || !(st->max_disparity >= st->min_disparity) ||
            !(st->band >= 0) || !(st->ratio > 0))
            return bfail(c, VO_ERR_ARG, "vo_batch_submit_stereo: bad stereo parameters");
        if (2 * B > c->cfg.max_batch)
            return bfail(c, VO_ERR_ARG, "vo_batch_submit_stereo: the context needs max_batch >= %d (has %d)", 2 * B,
                         c->cfg.max_batch);
        if (!b.enable_pose) return bfail(c, VO_ERR_ARG, "vo_batch_submit_stereo: the pose stage is not configured");
    }
    HIP_TRY(c, hipSetDevice(c->device));

    if (c->streams_started && !c->first_pairs_linked) {
        // from the second batch on, frame 0 of every stream pairs with the carry slot
        std::vector<int> q(B), hp(B, 1);
        for (int s = 0; s < S; s++)
            for (int f = 0; f < F; f++) q[s * F + f] = f > 0 ? S + s * F + f - 1 : s;
        HIP_TRY(c, hipMemcpyAsync(c->d_qslot, q.data(), sizeof(int) * B, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_has_prev, hp.data(), sizeof(int) * B, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->first_pairs_linked = true;
    }

    const int row_stride_in = row_stride;
    const size_t frame_stride_in = frame_stride;
    const bool right_on_host = right && !is_device_ptr(right);
    const uint8_t *src = frames;
    const size_t frames_bytes = (size_t)row_bytes * b.height * B;
    const size_t depth_bytes = dsz * (size_t)b.width * b.height * B;
    const bool frames_host = !is_device_ptr(frames);
    const bool depth_host_mem = depth && b.enable_pose && !right && !is_device_ptr(depth);
    // ordinary (not pinned) host memory goes through the context's own pinned staging area
    const bool stage_frames = frames_host && !is_pinned_host_ptr(frames);
    const bool stage_right = right_on_host && !is_pinned_host_ptr(right);
    const bool stage_depth = depth_host_mem && !is_pinned_host_ptr(depth);
    uint8_t *stage = nullptr;
    // VO_STAGE=0: leave ordinary host memory to the runtime's staging (for comparisons)
    static const bool own_stage = !(getenv("VO_STAGE") && getenv("VO_STAGE")[0] == '0');
    if (own_stage && (stage_frames || stage_right || stage_depth))
        stage = stage_area(c, 2 * frames_bytes + depth_bytes);  // (null: too large, the runtime stages)
    // Planes in ordinary memory are packed and sent in pieces by this thread and the context's helpers
    const bool pool_frames = stage_frames && stage != nullptr;
    const bool pool_depth = stage_depth && stage != nullptr;
    struct PoolGuard {  // whatever happens below, a posted job is brought to its end before this call returns
        vo_ctx *c;
        int n = 0;
        ~PoolGuard()
        {
            for (int k = 0; k < n; k++) (void)c->stager_work(c->stager.jobs[k], true);
        }
    } pool{c};
    // A depth plane in host memory is not uploaded at all when the GPU can read it where it lies: the back-projection
    // wants one value per tracked key point, a few hundred per frame, which it fetches over the link -- from the
    // caller's buffer if that is pinned, else from the pinned staging area the plane has been packed into.  (That is
    // half of the bytes of a BGRA + float32-depth frame off the link, and no second stream to wait for.)
    // VO_DEPTH_IN_PLACE=0: upload it as before (for comparisons).
    static const bool in_place_on = !(getenv("VO_DEPTH_IN_PLACE") && getenv("VO_DEPTH_IN_PLACE")[0] == '0');
    const float *depth_view = nullptr;  // device address of the host plane(s)
    if (depth_host_mem && in_place_on && (!stage_depth || pool_depth)) {
        void *dp = nullptr;
        void *hp = stage_depth ? static_cast<void *>(stage + 2 * frames_bytes)
                               : const_cast<void *>(static_cast<const void *>(depth));
        if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) depth_view = static_cast<const float *>(dp);
        else (void)hipGetLastError();
    }
    const bool depth_in_place = depth_view != nullptr;
    int job_frames = -1, job_depth = -1;
    if (pool_frames || pool_depth) {
        vo_ctx::StagePool &w = c->stager;
        {   // helpers that are still on their way out of the previous call's jobs
            std::unique_lock<std::mutex> lk(w.m);
            w.cv.wait(lk, [&] { return w.active == 0; });
        }
        auto fill = [&](vo_ctx::StageJob &j, const uint8_t *from, uint8_t *via, uint8_t *to, size_t rb, size_t rstride,
                        size_t pstride, hipStream_t stream) {
            j.src = from; j.stage = via; j.dst = to;
            j.row_bytes = rb; j.rows = b.height; j.row_stride = rstride; j.plane_stride = pstride; j.count = B;
            j.total = rb * (size_t)b.height * B;
            j.n_pieces = (int)((j.total + vo_ctx::StageJob::kPiece - 1) / vo_ctx::StageJob::kPiece);
            j.n_chunks = (j.n_pieces + vo_ctx::StageJob::kChunkPieces - 1) / vo_ctx::StageJob::kChunkPieces;
            j.stream = stream;
            j.next.store(0);
            for (int k = 0; k < j.n_chunks; k++) j.packed[k].store(0);
            j.sent = 0; j.err = 0;
        };
        int n = 0;
        if (pool_frames) {
            fill(w.jobs[n], frames, stage, c->d_in, (size_t)row_bytes, (size_t)row_stride, frame_stride, c->stream);
            job_frames = n++;
        }
        if (pool_depth) {
            fill(w.jobs[n], reinterpret_cast<const uint8_t *>(depth), stage + 2 * frames_bytes,
                 depth_in_place ? nullptr : reinterpret_cast<uint8_t *>(c->d_depth), dsz * b.width,
                 dsz * (size_t)depth_row_stride,
                 dsz * depth_frame_stride, c->copy_stream);
            job_depth = n++;
        }
        pool.n = n;
        c->stager_post(n);
    }
    if (frames_host) {
        if (pool_frames) {
            const hipError_t he = (hipError_t)c->stager_work(c->stager.jobs[job_frames], true);
            if (he != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_batch_submit: frame upload: %s", hipGetErrorString(he));
        } else {
            int rc0 = upload_planes(c, c->d_in, frames, row_bytes, b.height, row_stride, frame_stride, B, c->stream);
            if (rc0) return rc0;
        }
        src = c->d_in;
        row_stride = row_bytes;
        frame_stride = (size_t)row_bytes * b.height;
    }
    const uint8_t *rsrc = right;
    if (right_on_host) {
        uint8_t *dst = c->d_in + frames_bytes;  // second half of the staging area on the device
        int rc0 = upload_planes(c, dst, right, row_bytes, b.height, row_stride_in, frame_stride_in, B, c->stream,
                                stage_right && stage ? stage + frames_bytes : nullptr);
        if (rc0) return rc0;
        rsrc = dst;
    }
    const float *dsrc = depth;
    // A depth plane in host memory is not needed before the back-projection stage: it goes up on a second stream
    // while detection and matching of the frame already run (half of the bytes of a BGRA + float32-depth frame)
    const bool depth_from_host = depth_host_mem;
    const float *depth_host = depth;
    const int depth_row_stride_host = depth_row_stride;
    const size_t depth_frame_stride_host = depth_frame_stride;
    if (depth_in_place) {
        dsrc = depth_view;
        if (stage_depth) {  // packed planes
            depth_row_stride = b.width;
            depth_frame_stride = (size_t)b.width * b.height;
        }
    } else if (depth_from_host) {
        c->depth_planes_nan = false;
        dsrc = c->d_depth;
        depth_row_stride = b.width;
        depth_frame_stride = (size_t)b.width * b.height;
    }
    const bool frames_from_host = src != frames;
    // Pinned depth: the copy is queued before any kernel is launched, behind the frames' copy (two copies at once
    // would share the link and delay the frames, which everything waits for).  Ordinary memory: the copy call
    // keeps this thread busy, so it is made after the first part of the chain has been launched.
    const bool depth_early = depth_from_host && !depth_in_place && !stage_depth;
    const bool depth_late = depth_from_host && !depth_in_place && !depth_early && !pool_depth;
    auto upload_depth = [&]() -> int {
        if (frames_from_host) {
            HIP_TRY(c, hipEventRecord(c->ev_frames, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_frames, 0));
        }
        int rc0 = upload_planes(c, reinterpret_cast<uint8_t *>(c->d_depth), reinterpret_cast<const uint8_t *>(depth_host),
                                dsz * b.width, b.height, dsz * depth_row_stride_host,
                                dsz * depth_frame_stride_host, B, c->copy_stream);
        if (rc0) return rc0;
        HIP_TRY(c, hipEventRecord(c->ev_depth, c->copy_stream));
        return VO_OK;
    };
    if (depth_early) {
        int rc0 = upload_depth();
        if (rc0) return rc0;
    }
    if (!b.enable_pose) dsrc = nullptr;
    const int r_slot0 = S + B;  // feature slots of the right images
    int rc;
    // everything that may allocate or synchronise happens before the kernel chain (which may be recorded as a graph)
    if (right) {
        if ((rc = c->ensure_stereo())) return rc;
        if (!c->depth_planes_nan) {
            HIP_TRY(c, hipMemsetAsync(c->d_depth, 0xff, sizeof(float) * (size_t)b.width * b.height * B, c->stream));
            c->depth_planes_nan = true;
        }
        dsrc = c->d_depth;
        depth_row_stride = b.width;
        depth_frame_stride = (size_t)b.width * b.height;
    }
    if (b.orb_levels > 0) {
        // a single-frame ORB operator call may have re-planned the pyramid since configure
        rc = c->ensure_orb(b.width, b.height, b.orb_scale_factor > 0 ? b.orb_scale_factor : 1.2f, b.orb_levels,
                           b.max_features, B);
        if (rc) return rc;
    }
    if (c->packed && (rc = ensure_pack(c, B))) return rc;
    const int rs = right_on_host ? row_bytes : row_stride_in;
    const size_t fs = right_on_host ? (size_t)row_bytes * b.height : frame_stride_in;

    // the kernel chain in two parts: everything that needs the frames only, then everything from the depth plane on
    auto chain_front = [&]() {
        // with packed output the first block of the detector / matcher / track-age launches stores the device wall
        // clock: detect = [0, 1), match = [1, 2)
        c->pending_stamp = c->packed ? c->d_stamps : nullptr;
        if (right) {
            // the right images first, without timing marks: the marks of the detector stages are the left images'
            c->marks_on = false;
            if (b.orb_levels > 0)
                c->enqueue_detect_orb(rsrc, b.width, b.height, rs, fs, b.channels, b.fast_threshold, B, r_slot0);
            else
                c->enqueue_detect(rsrc, b.width, b.height, rs, fs, b.channels, b.max_features, b.fast_threshold, B,
                                  r_slot0);
            c->marks_on = true;
        }
        if (b.orb_levels > 0)
            c->enqueue_detect_orb(src, b.width, b.height, row_stride, frame_stride, b.channels, b.fast_threshold, B,
                                  /*first_slot=*/S);
        else
            c->enqueue_detect(src, b.width, b.height, row_stride, frame_stride, b.channels, b.max_features,
                              b.fast_threshold, B, /*first_slot=*/S);
        if (right) {
            // depth planes of the batch from the left / right key points (the kNN partial buffers are free until
            // the temporal matcher runs)
            (void)hipMemsetAsync(c->d_st_n, 0, sizeof(int) * B, c->stream);
            launch_stereo_match(c->d_kps, c->d_desc, c->d_nkp, c->kp_stride, S, r_slot0, B, st->band,
                                st->min_disparity, st->max_disparity, st->ratio, b.fx, st->baseline,
                                reinterpret_cast<uint16_t *>(c->d_st_index),
                                reinterpret_cast<int *>(c->d_st_index + c->st_rows_off),
                                c->d_partial, c->d_partial_rev, c->d_st_idx, c->d_st_disp, c->d_st_depth,
                                reinterpret_cast<uint32_t *>(c->d_depth), b.width, b.height, c->d_st_n, c->stream);
        }
        c->pending_stamp = c->packed ? c->d_stamps + 1 : nullptr;
        c->enqueue_match(B, b.ratio, b.cross_check != 0, FRansacParams{b.f_threshold, b.f_confidence, b.f_max_iters});
        c->mark(8);
        launch_track_ages(c->d_final, c->d_info, c->kp_stride, c->d_qslot, c->d_tslot, c->d_has_prev, S, F,
                          c->d_nkp, c->d_ages, c->d_filter_flag, c->stream, c->packed ? c->d_stamps + 2 : nullptr);
        c->mark(9);
    };
    auto chain_back = [&]() {
        launch_filter_backproject(c->d_final, c->d_info, c->kp_stride, c->d_tslot, c->d_has_prev, c->d_nkp,
                                  c->d_ages, c->d_filter_flag, dsrc, b.width, b.height, depth_row_stride,
                                  depth_frame_stride, b.fx, b.fy, b.cx, b.cy, c->d_tracked, c->d_valid, c->d_pts3,
                                  c->d_bpinfo, B, c->stream, pnp_share_words(c->d_pnp2), depth_u16 && !right);
        if (right)  // the planes go back to all-NaN for the next stereo batch
            launch_stereo_clear(c->d_kps, c->d_nkp, c->kp_stride, S, B, reinterpret_cast<uint32_t *>(c->d_depth),
                                b.width, b.height, c->stream);
        c->mark(10);
        if (!(b.enable_pose && dsrc)) {
            (void)hipMemsetAsync(c->d_pose, 0, sizeof(int) * 4 * B, c->stream);
            (void)hipMemsetAsync(c->d_rt, 0, sizeof(double) * 6 * B, c->stream);
        } else
            launch_pnp_ransac(c->d_valid, c->d_pts3, c->d_bpinfo, c->kp_stride, c->d_rng, c->rng_len, b.fx, b.fy,
                              b.cx, b.cy, b.pnp_iters, b.pnp_threshold, b.pnp_confidence, c->d_inliers, c->d_pose,
                              c->d_rt, B, c->d_pnp2, c->stream, /*zeroed=*/true);
        c->mark(11);
        if (c->packed) {
            // result records, packed frames and carry slots in one launch; the records reach the host inside the
            // packed block
            CollectSrc cs{c->d_nkp, c->d_tslot, c->d_ngood, c->d_info, c->d_bpinfo, c->d_pose, c->d_rt, c->d_results};
            PackSrc ps{c->d_results, c->d_kps, c->d_desc, c->d_tslot, {c->d_final, c->d_tracked, c->d_valid},
                       c->d_pts3, c->d_inliers, c->d_stamps, c->kp_stride};
            launch_finish_packed(cs, ps, c->pack, c->d_pack, B, c->d_kps, c->d_desc, c->d_nkp, c->d_ages, S, F,
                                 c->stream);
            (void)hipMemcpyAsync(c->h_pack, c->d_pack, c->pack.frame_bytes * B, hipMemcpyDeviceToHost, c->stream);
        } else {
            launch_collect(c->d_nkp, c->d_tslot, c->d_ngood, c->d_info, c->d_bpinfo, c->d_pose, c->d_rt, c->d_results,
                           B, c->stream);
            launch_carry(c->d_kps, c->d_desc, c->d_nkp, c->d_ages, c->kp_stride, S, F, c->stream);
            (void)hipMemcpyAsync(c->h_pinned, c->d_results, sizeof(vo_frame_result) * B, hipMemcpyDeviceToHost,
                                 c->stream);
        }
        c->mark(12);
    };

    if (c->use_graph) {
        // (whether the streams' first frames are linked to their carry slots is device data -- d_qslot,
        // d_has_prev -- and needs no new recording)
        if (c->graph_exec && c->graph_key.generation != c->generation) c->drop_graph();
        vo_ctx::GraphKey key;
        key.frames = src; key.depth = dsrc; key.right = rsrc;
        key.row_stride = row_stride; key.depth_row_stride = depth_row_stride;
        key.frame_stride = frame_stride; key.depth_frame_stride = depth_frame_stride;
        if (st) {
            key.st_baseline = st->baseline; key.st_band = st->band; key.st_min_disparity = st->min_disparity;
            key.st_max_disparity = st->max_disparity; key.st_ratio = st->ratio;
        }
        key.packed = c->packed; key.popcount = c->matcher_popcount; key.depth_u16 = depth_u16;
        if (c->graph_exec && !(key == c->graph_key)) c->drop_graph();
        key.generation = c->generation;
        if (!c->graph_exec) {
            // two recordings, so that the wait for the depth plane sits between them
            for (int part = 0; part < 2; part++) {
                hipGraph_t graph = nullptr;
                HIP_TRY(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
                c->capturing = true;
                if (part == 0) chain_front(); else chain_back();
                c->capturing = false;
                hipError_t ce = hipStreamEndCapture(c->stream, &graph);
                if (ce != hipSuccess || !graph) {
                    (void)hipGetLastError();
                    c->drop_graph();
                    return bfail(c, VO_ERR_HIP, "vo_batch_submit: graph capture failed: %s", hipGetErrorString(ce));
                }
                hipGraphExec_t *slot = part == 0 ? &c->graph_exec : &c->graph_exec_back;
                ce = hipGraphInstantiate(slot, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ce != hipSuccess) {
                    *slot = nullptr;
                    c->drop_graph();
                    return bfail(c, VO_ERR_HIP, "vo_batch_submit: hipGraphInstantiate: %s", hipGetErrorString(ce));
                }
            }
            c->graph_key = key;
        }
        if (pool_depth) (void)c->stager_send(c->stager.jobs[job_depth], false);
        HIP_TRY(c, hipGraphLaunch(c->graph_exec, c->stream));
    } else {
        // (chunks of the depth plane that the helpers have packed by now go on their way before the launches)
        if (pool_depth) (void)c->stager_send(c->stager.jobs[job_depth], false);
        chain_front();
    }
    if (depth_late && (rc = upload_depth())) return rc;
    if (pool_depth) {
        const hipError_t he = (hipError_t)c->stager_work(c->stager.jobs[job_depth], true);
        if (he != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_batch_submit: depth upload: %s", hipGetErrorString(he));
        // (read in place: the plane is complete in pinned memory before the kernels that read it are launched)
        if (!depth_in_place) HIP_TRY(c, hipEventRecord(c->ev_depth, c->copy_stream));
    }
    if (depth_from_host && !depth_in_place) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_depth, 0));
    if (c->use_graph) {
        HIP_TRY(c, hipGraphLaunch(c->graph_exec_back, c->stream));
    } else {
        chain_back();
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bfail(c, VO_ERR_HIP, "vo_batch_submit: %s", hipGetErrorString(e));
    c->submitted = true;
    c->streams_started = true;
    c->last_batch_stereo = right != nullptr;
    c->last_submit_graph = c->use_graph;
    c->packed_valid = c->packed;
    return VO_OK;
}

int vo_batch_submit(vo_ctx *c, const uint8_t *frames, int row_stride, size_t frame_stride, const float *depth,
                    int depth_row_stride, size_t depth_frame_stride)
{
    return batch_submit(c, frames, row_stride, frame_stride, depth, depth_row_stride, depth_frame_stride, nullptr,
                        nullptr);
}

int vo_batch_submit_depth16(vo_ctx *c, const uint8_t *frames, int row_stride, size_t frame_stride,
                            const uint16_t *depth_mm, int depth_row_stride, size_t depth_frame_stride)
{
    return batch_submit(c, frames, row_stride, frame_stride, reinterpret_cast<const float *>(depth_mm), depth_row_stride,
                        depth_frame_stride, nullptr, nullptr, /*depth_u16=*/true);
}

int vo_batch_submit_stereo(vo_ctx *c, const uint8_t *left, const uint8_t *right, int row_stride, size_t frame_stride,
                           const vo_stereo_cfg *stereo)
{
    if (!c) return VO_ERR_ARG;
    if (!right || !stereo) return bfail(c, VO_ERR_ARG, "vo_batch_submit_stereo: null pointer");
    return batch_submit(c, left, row_stride, frame_stride, nullptr, 0, 0, right, stereo);
}

int vo_batch_get_stereo(vo_ctx *c, int frame, int32_t *right_idx, float *disparity, float *depth, int *n_matched)
{
    if (!c) return VO_ERR_ARG;
    if (!c->configured || c->submitted || !c->last_batch_stereo)
        return bfail(c, VO_ERR_STATE, "vo_batch_get_stereo: no completed stereo batch");
    const int B = c->bcfg.n_streams * c->bcfg.frames_per_stream;
    if (frame < 0 || frame >= B) return bfail(c, VO_ERR_ARG, "vo_batch_get_stereo: frame %d of %d", frame, B);
    HIP_TRY(c, hipSetDevice(c->device));
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->d_nkp + c->bcfg.n_streams + frame, sizeof(int), hipMemcpyDeviceToHost));
    n = n < c->kp_stride ? n : c->kp_stride;
    const size_t o = (size_t)frame * c->kp_stride;
    if (right_idx && n) HIP_TRY(c, hipMemcpy(right_idx, c->d_st_idx + o, sizeof(int32_t) * n, hipMemcpyDefault));
    if (disparity && n) HIP_TRY(c, hipMemcpy(disparity, c->d_st_disp + o, sizeof(float) * n, hipMemcpyDefault));
    if (depth && n) HIP_TRY(c, hipMemcpy(depth, c->d_st_depth + o, sizeof(float) * n, hipMemcpyDefault));
    if (n_matched) HIP_TRY(c, hipMemcpy(n_matched, c->d_st_n + frame, sizeof(int), hipMemcpyDeviceToHost));
    return VO_OK;
}

int vo_batch_set_camera(vo_ctx *c, double fx, double fy, double cx, double cy)
{
    if (!c) return VO_ERR_ARG;
    if (!c->configured) return bfail(c, VO_ERR_STATE, "vo_batch_set_camera: configure first");
    if (c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_set_camera: a batch is in flight");
    if (!(fx > 0 && fy > 0)) return bfail(c, VO_ERR_ARG, "vo_batch_set_camera: fx, fy must be positive");
    c->bcfg.fx = fx; c->bcfg.fy = fy; c->bcfg.cx = cx; c->bcfg.cy = cy;
    c->bcfg.enable_pose = 1;
    c->generation++;  // the intrinsics are kernel arguments of a recorded chain
    return VO_OK;
}

int vo_batch_set_previous(vo_ctx *c, int stream, const vo_kp *kps, const uint8_t *desc, int n)
{
    if (!c) return VO_ERR_ARG;
    if (!c->configured || c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_set_previous: bad state");
    if (stream < 0 || stream >= c->bcfg.n_streams || n < 0 || n > c->cfg.max_features || (n && (!kps || !desc)))
        return bfail(c, VO_ERR_ARG, "vo_batch_set_previous: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int S = c->bcfg.n_streams, F = c->bcfg.frames_per_stream, B = S * F;
    if (n) {
        HIP_TRY(c, hipMemcpy(c->d_kps + (size_t)stream * c->kp_stride, kps, sizeof(vo_kp) * n, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(c->d_desc + (size_t)stream * c->kp_stride * 32, desc, (size_t)32 * n, hipMemcpyDefault));
    }
    HIP_TRY(c, hipMemcpy(c->d_nkp + stream, &n, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_ages + (size_t)stream * c->kp_stride, 0, sizeof(int) * c->kp_stride));
    // link every stream's first frame to its carry slot (empty carry slots hold 0 keypoints)
    std::vector<int> q(B), hp(B, 1);
    for (int s = 0; s < S; s++)
        for (int f = 0; f < F; f++) q[s * F + f] = f > 0 ? S + s * F + f - 1 : s;
    HIP_TRY(c, hipMemcpy(c->d_qslot, q.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_has_prev, hp.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    c->streams_started = true;
    c->first_pairs_linked = true;
    return VO_OK;
}

int vo_batch_set_packed_output(vo_ctx *c, int enable)
{
    if (!c) return VO_ERR_ARG;
    if (c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_set_packed_output: a batch is in flight");
    if (enable && c->configured) {
        HIP_TRY(c, hipSetDevice(c->device));
        const int rc = ensure_pack(c, c->bcfg.n_streams * c->bcfg.frames_per_stream);
        if (rc) return rc;
    }
    c->packed = enable != 0;
    c->packed_valid = false;
    return VO_OK;
}

int vo_batch_get_packed(vo_ctx *c, int frame, vo_frame_view *view)
{
    if (!c || !view) return VO_ERR_ARG;
    if (!c->configured || c->submitted || !c->packed_valid)
        return bfail(c, VO_ERR_STATE, "vo_batch_get_packed: no completed batch with packed output");
    if (frame < 0 || frame >= c->bcfg.n_streams * c->bcfg.frames_per_stream)
        return bfail(c, VO_ERR_ARG, "vo_batch_get_packed: frame %d out of range", frame);
    const PackLayout &L = c->pack;
    const uint8_t *base = static_cast<const uint8_t *>(c->h_pack) + (size_t)frame * L.frame_bytes;
    memcpy(&view->result, base, sizeof(vo_frame_result));
    unsigned long long st[3];
    memcpy(st, base + sizeof(vo_frame_result), sizeof(st));
    view->detect_ms = (float)((double)(st[1] - st[0]) / c->wall_clock_khz);
    view->match_ms = (float)((double)(st[2] - st[1]) / c->wall_clock_khz);
    const vo_frame_result &r = view->result;
    auto cl = [&](int n) { return n < L.cap ? (n < 0 ? 0 : n) : L.cap; };
    view->n_keypoints = cl(r.n_keypoints);
    view->kps = reinterpret_cast<const vo_kp *>(base + L.off_kps);
    view->desc = base + L.off_desc;
    view->n_matches = cl(r.n_matches); view->n_tracked = cl(r.n_tracked); view->n_valid = cl(r.n_3d);
    view->matches = reinterpret_cast<const vo_match *>(base + L.off_m[0]);
    view->tracked = reinterpret_cast<const vo_match *>(base + L.off_m[1]);
    view->valid = reinterpret_cast<const vo_match *>(base + L.off_m[2]);
    view->pts3 = reinterpret_cast<const float *>(base + L.off_pts3);
    view->n_inliers = cl(r.pnp_inliers);
    view->inliers = reinterpret_cast<const int32_t *>(base + L.off_inl);
    return VO_OK;
}

int vo_batch_set_graph(vo_ctx *c, int enable)
{
    if (!c) return VO_ERR_ARG;
    if (c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_set_graph: a batch is in flight");
    c->use_graph = enable != 0;
    if (!enable) {
        HIP_TRY(c, hipSetDevice(c->device));
        c->drop_graph();
    }
    return VO_OK;
}

void *vo_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void vo_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int vo_batch_wait(vo_ctx *c, vo_frame_result *results)
{
    if (!c) return VO_ERR_ARG;
    if (!c->submitted) return bfail(c, VO_ERR_STATE, "vo_batch_wait: nothing submitted");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->submitted = false;
    if (c->packed_valid) {
        // the result records came up at the head of every packed frame
        const int B = c->bcfg.n_streams * c->bcfg.frames_per_stream;
        for (int f = 0; f < B; f++)
            memcpy(static_cast<vo_frame_result *>(c->h_pinned) + f,
                   static_cast<const uint8_t *>(c->h_pack) + (size_t)f * c->pack.frame_bytes, sizeof(vo_frame_result));
    }
    if (results)
        memcpy(results, c->h_pinned, sizeof(vo_frame_result) * c->bcfg.n_streams * c->bcfg.frames_per_stream);
    return VO_OK;
}

static const char *const kStageNames[vo_ctx::kStages] = {
    "frontend_fast_nms_blur", "pyramid_resize", "frontend_upper_levels", "select_topn", "brief", "knn2",
    "match_finalize", "f_ransac", "track_ages", "filter_backproject", "pnp_ransac", "collect_carry"};

int vo_batch_set_timing(vo_ctx *c, int enable)
{
    if (!c) return VO_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (enable)
        for (hipEvent_t &e : c->ev)
            if (!e) HIP_TRY(c, hipEventCreate(&e));
    c->timing = enable != 0;
    return VO_OK;
}

int vo_batch_stage_count(void) { return vo_ctx::kStages; }

const char *vo_batch_stage_name(int stage)
{
    return stage >= 0 && stage < vo_ctx::kStages ? kStageNames[stage] : "";
}

int vo_batch_get_timing(vo_ctx *c, float *ms)
{
    if (!c) return VO_ERR_ARG;
    if (!c->timing || c->submitted || !ms)
        return bfail(c, VO_ERR_STATE, "vo_batch_get_timing: enable timing, submit and wait first");
    if (c->last_submit_graph)
        return bfail(c, VO_ERR_STATE, "vo_batch_get_timing: the last batch was a graph replay, which records no "
                                      "per-stage events (vo_batch_set_graph(ctx, 0) for stage timing)");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int i = 0; i < vo_ctx::kStages; i++) HIP_TRY(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    return VO_OK;
}

static int frame_ok(vo_ctx *c, int frame, const char *who)
{
    if (!c) return VO_ERR_ARG;
    if (!c->configured || c->submitted) return bfail(c, VO_ERR_STATE, "%s: no completed batch", who);
    if (frame < 0 || frame >= c->bcfg.n_streams * c->bcfg.frames_per_stream)
        return bfail(c, VO_ERR_ARG, "%s: frame %d out of range", who, frame);
    return VO_OK;
}

int vo_batch_get_features(vo_ctx *c, int frame, vo_kp *kps, uint8_t *desc, int *n_out)
{
    int rc = frame_ok(c, frame, "vo_batch_get_features");
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const int slot = c->bcfg.n_streams + frame;
    int n = 0;
    HIP_TRY(c, hipMemcpy(&n, c->d_nkp + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    if (n > 0 && kps)
        HIP_TRY(c, hipMemcpy(kps, c->d_kps + (size_t)slot * c->kp_stride, sizeof(vo_kp) * n, hipMemcpyDefault));
    if (n > 0 && desc)
        HIP_TRY(c, hipMemcpy(desc, c->d_desc + (size_t)slot * c->kp_stride * 32, (size_t)32 * n, hipMemcpyDefault));
    return VO_OK;
}

int vo_batch_get_matches(vo_ctx *c, int frame, int which, vo_match *out, float *pts3, int *n_out)
{
    int rc = frame_ok(c, frame, "vo_batch_get_matches");
    if (rc) return rc;
    if (which < 0 || which > 2) return bfail(c, VO_ERR_ARG, "vo_batch_get_matches: which must be 0, 1 or 2");
    HIP_TRY(c, hipSetDevice(c->device));
    const vo_frame_result *r = reinterpret_cast<const vo_frame_result *>(c->h_pinned) + frame;
    const int n = which == 0 ? r->n_matches : (which == 1 ? r->n_tracked : r->n_3d);
    const vo_match *src = (which == 0 ? c->d_final : (which == 1 ? c->d_tracked : c->d_valid)) +
                          (size_t)frame * c->kp_stride;
    if (n_out) *n_out = n;
    if (n > 0 && out) HIP_TRY(c, hipMemcpy(out, src, sizeof(vo_match) * n, hipMemcpyDefault));
    if (n > 0 && pts3 && which == 2)
        HIP_TRY(c, hipMemcpy(pts3, c->d_pts3 + (size_t)frame * c->kp_stride * 3, sizeof(float) * 3 * n,
                             hipMemcpyDefault));
    return VO_OK;
}

int vo_batch_get_inliers(vo_ctx *c, int frame, int32_t *inliers, int *n_out)
{
    int rc = frame_ok(c, frame, "vo_batch_get_inliers");
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const vo_frame_result *r = reinterpret_cast<const vo_frame_result *>(c->h_pinned) + frame;
    if (n_out) *n_out = r->pnp_inliers;
    if (r->pnp_inliers > 0 && inliers)
        HIP_TRY(c, hipMemcpy(inliers, c->d_inliers + (size_t)frame * c->kp_stride, sizeof(int) * r->pnp_inliers,
                             hipMemcpyDefault));
    return VO_OK;
}

}  // extern "C"
