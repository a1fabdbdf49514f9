// C ABI of libvo_hip.so (declared in include/vo_hip.h): context management
// and the single-frame operators.  Device work is enqueued on the context's
// stream through the launchers of vo_internal.h; there is no host fallback.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/vo_orb_pattern.h"
#include "vo_ctx.h"

using namespace vo;

static thread_local std::string g_create_error;

static int fail(vo_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, VO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// cv::RNG with seed 2^64-1: the raw 32-bit words are the same for every call,
// so they are tabulated once (reference src/feature_matcher.cpp:54 reaches
// RANSACPointSetRegistrator::run, which constructs RNG((uint64)-1)).
static void fill_rng_table(std::vector<uint32_t> &t)
{
    uint64_t state = ~0ull;
    for (auto &w : t) {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        w = (uint32_t)state;
    }
}

// ORB's sampling pattern rotated by the angle FAST keypoints carry (-1 degree,
// i.e. cv::KeyPoint's "not applicable"), exactly as computeOrbDescriptors does:
// float a = (float)cos(angle), b = (float)sin(angle), cvRound of x*a - y*b etc.
static void build_pattern(float angle_deg, int8_t *out)
{
    float angle = angle_deg;
    angle *= (float)(3.14159265358979323846 / 180.f);
    const float a = (float)cos(angle), b = (float)sin(angle);
    for (int i = 0; i < 512; i++) {
        const float px = vo_orb_bit_pattern_31[2 * i], py = vo_orb_bit_pattern_31[2 * i + 1];
        const float x = px * a - py * b, y = px * b + py * a;
        out[2 * i] = (int8_t)lrint((double)x);
        out[2 * i + 1] = (int8_t)lrint((double)y);
    }
}

extern "C" {

int vo_abi_version(void) { return 1; }

const char *vo_last_error(const vo_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

vo_ctx *vo_create(int device, const vo_cfg *cfg)
{
    if (!cfg || cfg->max_width < 16 || cfg->max_height < 16 || cfg->max_width > kMaxDim ||
        cfg->max_height > kMaxDim || (long long)cfg->max_width * cfg->max_height > (1ll << 24) ||
        cfg->max_features < 1 || cfg->max_features > VO_MAX_FEATURES_LIMIT || cfg->max_batch < 1) {
        fail(nullptr, VO_ERR_ARG, "vo_create: bad configuration");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        fail(nullptr, VO_ERR_HIP, "vo_create: no usable HIP device %d (found %d)", device, ndev);
        return nullptr;
    }
    vo_ctx *c = new vo_ctx();
    c->device = device;
    c->cfg = *cfg;
    c->gc = {3735, 19235, 9798, 15};
    if (cfg->gray_shift > 0) c->gc = {cfg->gray_b, cfg->gray_g, cfg->gray_r, cfg->gray_shift};
    int rc = c->allocate();
    if (rc != VO_OK) {
        g_create_error = c->err;
        vo_destroy(c);
        return nullptr;
    }
    return c;
}

void vo_destroy(vo_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->release();
    delete c;
}

}  // extern "C"

int vo_ctx::allocate()
{
    vo_ctx *ctx = this;
    HIP_TRY(ctx, hipSetDevice(device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev_depth, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev_frames, hipEventDisableTiming));
    const int B = cfg.max_batch;
    pitch = align_up(cfg.max_width, 128);
    hpad = align_up(cfg.max_height, 16);
    plane_bytes = (size_t)pitch * hpad;
    kp_stride = align_up(cfg.max_features, 8);
    slots = 2 * B + 3;  // S*(F+1) <= 2B batch slots, two for the single-frame operators, one empty
    max_tiles = tiles_of(cfg.max_width, cfg.max_height);
    in_frame_bytes = (size_t)cfg.max_width * cfg.max_height * 4;

    HIP_TRY(ctx, hipMalloc(&d_in, in_frame_bytes * B));
    HIP_TRY(ctx, hipMalloc(&d_gray, plane_bytes * B));
    HIP_TRY(ctx, hipMalloc(&d_score, plane_bytes * B));
    HIP_TRY(ctx, hipMalloc(&d_blur, plane_bytes * B));
    HIP_TRY(ctx, hipMalloc(&d_cand, sizeof(uint32_t) * (size_t)max_tiles * kTileCap * B));
    HIP_TRY(ctx, hipMalloc(&d_cand_flat, sizeof(uint32_t) * (size_t)max_tiles * kTileCap * B));
    HIP_TRY(ctx, hipMalloc(&d_cand_count, sizeof(uint32_t) * (size_t)max_tiles * B));
    HIP_TRY(ctx, hipMalloc(&d_kps, sizeof(vo_kp) * (size_t)kp_stride * slots));
    HIP_TRY(ctx, hipMalloc(&d_desc, (size_t)32 * kp_stride * slots));
    HIP_TRY(ctx, hipMalloc(&d_nkp, sizeof(int) * slots));
    HIP_TRY(ctx, hipMalloc(&d_pattern, 1024));
    // (knn_splits() picks kKnnSplitsFew only for launches of fewer than 64 pairs)
    const size_t partial_slots = std::max((size_t)kKnnSplits * B, (size_t)kKnnSplitsFew * std::min(B, 64));
    HIP_TRY(ctx, hipMalloc(&d_partial, sizeof(uint2) * partial_slots * kp_stride));
    HIP_TRY(ctx, hipMalloc(&d_partial_rev, sizeof(uint2) * partial_slots * kp_stride));
    HIP_TRY(ctx, hipMalloc(&d_knn, sizeof(vo_knn2) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_good, sizeof(vo_match) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_ngood, sizeof(int) * B));
    HIP_TRY(ctx, hipMalloc(&d_mask, (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_F, sizeof(double) * 9 * B));
    HIP_TRY(ctx, hipMalloc(&d_info, sizeof(int) * 4 * B));
    HIP_TRY(ctx, hipMalloc(&d_final, sizeof(vo_match) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_qslot, sizeof(int) * B));
    HIP_TRY(ctx, hipMalloc(&d_tslot, sizeof(int) * B));
    HIP_TRY(ctx, hipMalloc(&d_depth, sizeof(float) * (size_t)cfg.max_width * cfg.max_height * B));
    HIP_TRY(ctx, hipMalloc(&d_ages, sizeof(int) * (size_t)kp_stride * slots));
    HIP_TRY(ctx, hipMalloc(&d_has_prev, sizeof(int) * B));
    HIP_TRY(ctx, hipMalloc(&d_filter_flag, sizeof(int) * B));
    HIP_TRY(ctx, hipMalloc(&d_tracked, sizeof(vo_match) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_valid, sizeof(vo_match) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_pts3, sizeof(float) * 3 * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_bpinfo, sizeof(int) * 2 * B));
    HIP_TRY(ctx, hipMalloc(&d_inliers, sizeof(int) * (size_t)kp_stride * B));
    HIP_TRY(ctx, hipMalloc(&d_pose, sizeof(int) * 4 * B));
    HIP_TRY(ctx, hipMalloc(&d_rt, sizeof(double) * 6 * B));
    HIP_TRY(ctx, hipMalloc(&d_results, sizeof(vo_frame_result) * B));
    {
        size_t off[4];
        HIP_TRY(ctx, hipMalloc(&d_pnp2, pnp_phase2_bytes(B, off)));
    }
    HIP_TRY(ctx, hipMemset(d_ages, 0, sizeof(int) * (size_t)kp_stride * slots));
    HIP_TRY(ctx, hipMemset(d_pose, 0, sizeof(int) * 4 * B));
    HIP_TRY(ctx, hipMemset(d_rt, 0, sizeof(double) * 6 * B));
    HIP_TRY(ctx, hipMemset(d_bpinfo, 0, sizeof(int) * 2 * B));
    empty_slot = slots - 1;
    op_slot0 = slots - 3;
    op_slot1 = slots - 2;
    HIP_TRY(ctx, hipMemset(d_nkp, 0, sizeof(int) * slots));
    HIP_TRY(ctx, hipMemset(d_score, 0, plane_bytes * B));

    std::vector<uint32_t> rng(1 << 16);
    fill_rng_table(rng);
    rng_len = (int)rng.size();
    HIP_TRY(ctx, hipMalloc(&d_aux, plane_bytes));
    HIP_TRY(ctx, hipMalloc(&d_hist, 256 * sizeof(uint32_t) + 256));
    HIP_TRY(ctx, hipMalloc(&d_rng, sizeof(uint32_t) * rng.size()));
    HIP_TRY(ctx, hipMemcpy(d_rng, rng.data(), sizeof(uint32_t) * rng.size(), hipMemcpyHostToDevice));

    int8_t pat[1024];
    build_pattern(-1.0f, pat);
    HIP_TRY(ctx, hipMemcpy(d_pattern, pat, 1024, hipMemcpyHostToDevice));
    {
        // descriptor byte b, bit t is test 8 b + t: its four coordinates as floats at [(t * 32 + b) * 4], so that the
        // 32 lanes of a key point (one per byte) read consecutive 16-byte entries
        std::vector<float> patf(1024);
        for (int b = 0; b < 32; b++)
            for (int t = 0; t < 8; t++)
                for (int c = 0; c < 4; c++)
                    patf[(size_t)(t * 32 + b) * 4 + c] = (float)vo_orb_bit_pattern_31[(b * 8 + t) * 4 + c];
        HIP_TRY(ctx, hipMalloc(&d_pattern0, 4096));
        HIP_TRY(ctx, hipMemcpy(d_pattern0, patf.data(), 4096, hipMemcpyHostToDevice));
    }
    pinned_bytes = std::max(kPinnedBytes, sizeof(vo_frame_result) * (size_t)B);
    HIP_TRY(ctx, hipHostMalloc(&h_pinned, pinned_bytes));
    return VO_OK;
}

void vo_ctx::release()
{
    stager_stop();
    void *ptrs[] = {d_in, d_gray, d_score, d_blur, d_cand, d_cand_flat, d_cand_count, d_orb_cand_flat, d_kps, d_desc, d_nkp, d_pattern,
                    d_partial, d_partial_rev, d_knn, d_good, d_ngood, d_mask, d_F, d_info, d_final, d_qslot, d_tslot, d_rng, d_aux, d_hist, d_pattern0, d_orb_tab, d_eq, d_eE, d_eidx, d_enm, d_ecnt, d_einfo, d_emask,
                    d_st_idx, d_st_disp, d_st_depth, d_st_plane, d_st_n, d_st_index,
                    d_pyr, d_pyr_blur, d_orb_cand, d_orb_cand_count, d_lvl_ras, d_lvl_resp, d_lvl_n,
                    d_depth, d_ages, d_has_prev, d_filter_flag, d_tracked, d_valid, d_pts3, d_bpinfo,
                    d_inliers, d_pose, d_rt, d_results, d_pnp2};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    drop_graph();
    if (d_pack) (void)hipFree(d_pack);
    if (d_stamps) (void)hipFree(d_stamps);
    if (h_pack) (void)hipHostFree(h_pack);
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (h_stage) (void)hipHostFree(h_stage);
    if (ev_depth) (void)hipEventDestroy(ev_depth);
    if (ev_frames) (void)hipEventDestroy(ev_frames);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

static int check_frame(vo_ctx *c, int w, int h, const char *who)
{
    if (!c) return VO_ERR_ARG;
    if (w < 7 || h < 7 || w > c->cfg.max_width || h > c->cfg.max_height)
        return fail(c, VO_ERR_ARG, "%s: frame %dx%d outside the context limits %dx%d", who, w, h,
                    c->cfg.max_width, c->cfg.max_height);
    return VO_OK;
}

static int finish(vo_ctx *c, const char *who)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, VO_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    return VO_OK;
}

// copy a user image (host or device, any stride) into the packed staging area
static int stage_image(vo_ctx *c, const uint8_t *src, int row_bytes, int h, int stride)
{
    HIP_TRY(c, hipMemcpy2DAsync(c->d_in, row_bytes, src, stride, row_bytes, h, hipMemcpyDefault, c->stream));
    return VO_OK;
}

extern "C" {

int vo_gray(vo_ctx *c, const uint8_t *src, int w, int h, int src_stride, int channels, uint8_t *dst,
            int dst_stride)
{
    int rc = check_frame(c, w, h, "vo_gray");
    if (rc) return rc;
    if (!src || !dst || (channels != 3 && channels != 4) || src_stride < w * channels || dst_stride < w)
        return fail(c, VO_ERR_ARG, "vo_gray: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = stage_image(c, src, w * channels, h, src_stride))) return rc;
    launch_gray(c->d_in, w, h, w * channels, 0, channels, c->plane(c->d_gray), 1, c->gc, c->stream);
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->d_gray, c->pitch, w, h, hipMemcpyDefault, c->stream));
    return finish(c, "vo_gray");
}

int vo_fast_score(vo_ctx *c, const uint8_t *gray, int w, int h, int stride, int threshold, uint8_t *score,
                  int score_stride)
{
    int rc = check_frame(c, w, h, "vo_fast_score");
    if (rc) return rc;
    if (!gray || !score || stride < w || score_stride < w) return fail(c, VO_ERR_ARG, "vo_fast_score: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = stage_image(c, gray, w, h, stride))) return rc;
    launch_frontend(c->d_in, w, h, w, 0, 1, PlaneBatch{nullptr, 0, 0}, c->plane(c->d_score),
                    PlaneBatch{nullptr, 0, 0}, threshold, nullptr, nullptr, 1, c->gc, c->stream);
    HIP_TRY(c, hipMemcpy2DAsync(score, score_stride, c->d_score, c->pitch, w, h, hipMemcpyDefault, c->stream));
    return finish(c, "vo_fast_score");
}

int vo_blur7(vo_ctx *c, const uint8_t *gray, int w, int h, int stride, uint8_t *dst, int dst_stride)
{
    int rc = check_frame(c, w, h, "vo_blur7");
    if (rc) return rc;
    if (!gray || !dst || stride < w || dst_stride < w) return fail(c, VO_ERR_ARG, "vo_blur7: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = stage_image(c, gray, w, h, stride))) return rc;
    launch_frontend(c->d_in, w, h, w, 0, 1, PlaneBatch{nullptr, 0, 0}, PlaneBatch{nullptr, 0, 0},
                    c->plane(c->d_blur), 0, nullptr, nullptr, 1, c->gc, c->stream);
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->d_blur, c->pitch, w, h, hipMemcpyDefault, c->stream));
    return finish(c, "vo_blur7");
}

// cv::GaussianBlur's 8.8 fixed-point taps for 8-bit images: normalised
// Gaussian (small fixed tables for sigma <= 0 and n <= 7), side taps rounded
// to 1/256 with the rounding error carried forward, centre = 256 - the rest.
static bool gaussian_taps_q8(int n, double sigma, uint16_t *taps)
{
    if (n < 1 || n > 31 || (n & 1) == 0) return false;
    const int half = n / 2;
    std::vector<double> g(n);
    if (sigma <= 0 && n <= 7) {
        static const double tab[4][4] = {{1.0}, {0.25, 0.5}, {0.0625, 0.25, 0.375}, {0.03125, 0.109375, 0.21875, 0.28125}};
        for (int i = 0; i <= half; i++) g[i] = g[n - 1 - i] = tab[half][i];
    } else {
        const double sd = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
        const double a = -0.5 / (sd * sd);
        double side = 0;
        for (int i = 0; i < half; i++) {
            const double x = i - half;
            g[i] = std::exp(a * x * x);
            side += g[i];
        }
        const double inv = 1.0 / (side * 2 + 1.0);
        for (int i = 0; i < half; i++) g[i] = g[n - 1 - i] = g[i] * inv;
        g[half] = inv;
    }
    double carry = 0;
    int used = 0;
    for (int i = 0; i < half; i++) {
        const double want = g[i] * 256.0 + carry;
        const int q = (int)std::lrint(want);
        carry = want - (double)q;
        taps[i] = taps[n - 1 - i] = (uint16_t)q;
        used += 2 * q;
    }
    taps[half] = (uint16_t)(256 - used);
    return true;
}

int vo_image_process(vo_ctx *c, const uint8_t *src, int w, int h, int src_stride, int channels, int enhance,
                     int hist_eq, int blur_size, double sigma, uint8_t *gray, uint8_t *enhanced,
                     uint8_t *denoised, uint8_t *masked, int out_stride)
{
    int rc = check_frame(c, w, h, "vo_image_process");
    if (rc) return rc;
    if (!src || !gray || src_stride < w * channels || out_stride < w)
        return fail(c, VO_ERR_ARG, "vo_image_process: bad arguments");
    if (channels != 1 && channels != 3)
        return fail(c, VO_ERR_UNSUPPORTED, "vo_image_process: %d-channel input (1 or 3 expected)", channels);
    if (enhance && (!enhanced || !denoised || !masked))
        return fail(c, VO_ERR_ARG, "vo_image_process: enhance needs all four output planes");
    uint16_t taps[31] = {256};
    int n_taps = 1;
    if (enhance && blur_size > 1) {
        if (!gaussian_taps_q8(blur_size, sigma, taps))
            return fail(c, VO_ERR_ARG, "vo_image_process: blur size %d is not odd and <= 31", blur_size);
        n_taps = blur_size;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (channels == 3) {
        if ((rc = stage_image(c, src, w * 3, h, src_stride))) return rc;
        launch_gray(c->d_in, w, h, w * 3, 0, 3, c->plane(c->d_gray), 1, c->gc, c->stream);
    } else {
        HIP_TRY(c, hipMemcpy2DAsync(c->d_gray, c->pitch, src, src_stride, w, h, hipMemcpyDefault, c->stream));
    }
    HIP_TRY(c, hipMemcpy2DAsync(gray, out_stride, c->d_gray, c->pitch, w, h, hipMemcpyDefault, c->stream));
    if (enhance) {
        uint8_t *lut = nullptr;
        if (hist_eq) {
            lut = reinterpret_cast<uint8_t *>(c->d_hist + 256);
            launch_hist256(c->d_gray, w, h, c->pitch, c->d_hist, c->stream);
            launch_equalize_lut(c->d_hist, w * h, lut, c->stream);
        }
        launch_enhance_blur_mask(c->d_gray, w, h, c->pitch, lut, n_taps, taps, c->d_score, c->d_blur, c->d_aux,
                                 c->stream);
        HIP_TRY(c, hipMemcpy2DAsync(enhanced, out_stride, c->d_score, c->pitch, w, h, hipMemcpyDefault, c->stream));
        HIP_TRY(c, hipMemcpy2DAsync(denoised, out_stride, c->d_blur, c->pitch, w, h, hipMemcpyDefault, c->stream));
        HIP_TRY(c, hipMemcpy2DAsync(masked, out_stride, c->d_aux, c->pitch, w, h, hipMemcpyDefault, c->stream));
    }
    return finish(c, "vo_image_process");
}

int vo_undistort(vo_ctx *c, const uint8_t *src, int w, int h, int src_stride, double fx, double fy, double cx,
                 double cy, const double *dist, int n_dist, uint8_t *dst, int dst_stride)
{
    int rc = check_frame(c, w, h, "vo_undistort");
    if (rc) return rc;
    if (!src || !dst || !dist || src_stride < w || dst_stride < w || !(fx > 0 && fy > 0))
        return fail(c, VO_ERR_ARG, "vo_undistort: bad arguments");
    if (n_dist != 4 && n_dist != 5 && n_dist != 8)
        return fail(c, VO_ERR_UNSUPPORTED, "vo_undistort: %d distortion coefficients (4, 5 or 8 expected)", n_dist);
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist; i++) k[i] = dist[i];
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy2DAsync(c->d_gray, c->pitch, src, src_stride, w, h, hipMemcpyDefault, c->stream));
    launch_undistort(c->d_gray, w, h, (int)c->pitch, fx, fy, cx, cy, k, c->d_aux, (int)c->pitch, c->stream);
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->d_aux, c->pitch, w, h, hipMemcpyDefault, c->stream));
    return finish(c, "vo_undistort");
}

int vo_detect(vo_ctx *c, const uint8_t *gray, int w, int h, int stride, int max_features, int fast_threshold,
              vo_kp *kps, uint8_t *desc, int *n_out)
{
    int rc = check_frame(c, w, h, "vo_detect");
    if (rc) return rc;
    if (!gray || !kps || !desc || !n_out || stride < w || max_features < 1 ||
        max_features > c->cfg.max_features)
        return fail(c, VO_ERR_ARG, "vo_detect: bad arguments (max_features %d, context limit %d)",
                    max_features, c->cfg.max_features);
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = stage_image(c, gray, w, h, stride))) return rc;
    c->enqueue_detect(c->d_in, w, h, w, 0, 1, max_features, fast_threshold, 1, c->op_slot0);
    HIP_TRY(c, hipMemcpyAsync(c->h_pinned, c->d_nkp + c->op_slot0, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish(c, "vo_detect"))) return rc;
    const int n = *reinterpret_cast<int *>(c->h_pinned);
    *n_out = n;
    if (n > 0) {
        HIP_TRY(c, hipMemcpy(kps, c->d_kps + (size_t)c->op_slot0 * c->kp_stride, sizeof(vo_kp) * n, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(desc, c->d_desc + (size_t)c->op_slot0 * c->kp_stride * 32, (size_t)32 * n, hipMemcpyDefault));
    }
    return VO_OK;
}

static int upload_features(vo_ctx *c, int slot, const vo_kp *kps, const uint8_t *desc, int n)
{
    if (kps) HIP_TRY(c, hipMemcpyAsync(c->d_kps + (size_t)slot * c->kp_stride, kps, sizeof(vo_kp) * n,
                                      hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_desc + (size_t)slot * c->kp_stride * 32, desc, (size_t)32 * n,
                              hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_nkp + slot, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // &n is a stack variable
    return VO_OK;
}

static int set_single_pair(vo_ctx *c)
{
    const int q = c->op_slot0, t = c->op_slot1;
    HIP_TRY(c, hipMemcpy(c->d_qslot, &q, sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tslot, &t, sizeof(int), hipMemcpyHostToDevice));
    return VO_OK;
}

int vo_detect_orb(vo_ctx *c, const uint8_t *gray, int w, int h, int stride, int nfeatures, float scale_factor,
                  int nlevels, int fast_threshold, vo_kp *kps, uint8_t *desc, int cap, int *n_out)
{
    int rc = check_frame(c, w, h, "vo_detect_orb");
    if (rc) return rc;
    if (!gray || !kps || !desc || !n_out || stride < w || nfeatures < 1 || nfeatures > c->cfg.max_features || cap < 1)
        return fail(c, VO_ERR_ARG, "vo_detect_orb: bad arguments (nfeatures %d, context limit %d)", nfeatures,
                    c->cfg.max_features);
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = c->ensure_orb(w, h, scale_factor, nlevels, nfeatures, 1))) return rc;
    if ((rc = stage_image(c, gray, w, h, stride))) return rc;
    c->enqueue_detect_orb(c->d_in, w, h, w, 0, 1, fast_threshold, 1, c->op_slot0);
    HIP_TRY(c, hipMemcpyAsync(c->h_pinned, c->d_nkp + c->op_slot0, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if ((rc = finish(c, "vo_detect_orb"))) return rc;
    const int n = std::min(*reinterpret_cast<int *>(c->h_pinned), cap);
    *n_out = n;
    if (n > 0) {
        HIP_TRY(c, hipMemcpy(kps, c->d_kps + (size_t)c->op_slot0 * c->kp_stride, sizeof(vo_kp) * n, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(desc, c->d_desc + (size_t)c->op_slot0 * c->kp_stride * 32, (size_t)32 * n, hipMemcpyDefault));
    }
    return VO_OK;
}

int vo_orb_get_level(vo_ctx *c, int frame, int level, int blurred, uint8_t *dst, int dst_stride, int *w_out,
                     int *h_out)
{
    if (!c) return VO_ERR_ARG;
    if (c->orb.nlevels < 1 || level < 0 || level >= c->orb.nlevels || frame < 0 || frame >= c->orb_batch)
        return fail(c, VO_ERR_ARG, "vo_orb_get_level: no such level (run the ORB detector first)");
    const OrbLevel &L = c->orb.lv[level];
    if (w_out) *w_out = L.w;
    if (h_out) *h_out = L.h;
    if (!dst) return VO_OK;
    if (dst_stride < L.w) return fail(c, VO_ERR_ARG, "vo_orb_get_level: stride too small");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint8_t *base = (blurred ? c->d_pyr_blur : c->d_pyr) + (size_t)frame * c->orb.frame_bytes + L.off;
    HIP_TRY(c, hipMemcpy2D(dst, dst_stride, base, L.pitch, L.w, L.h, hipMemcpyDefault));
    return VO_OK;
}

int vo_set_matcher(vo_ctx *c, int kind)
{
    if (!c) return VO_ERR_ARG;
    if (kind != VO_MATCHER_MFMA && kind != VO_MATCHER_POPCOUNT) return fail(c, VO_ERR_ARG, "vo_set_matcher: kind %d", kind);
    c->matcher_popcount = kind == VO_MATCHER_POPCOUNT;
    return VO_OK;
}

int vo_knn2_hamming(vo_ctx *c, const uint8_t *query, int nq, const uint8_t *train, int nt, vo_knn2 *out)
{
    if (!c) return VO_ERR_ARG;
    if (!query || !train || !out || nq < 0 || nt < 0 || nq > c->cfg.max_features || nt > c->cfg.max_features)
        return fail(c, VO_ERR_ARG, "vo_knn2_hamming: bad arguments (nq %d, nt %d, limit %d)", nq, nt,
                    c->cfg.max_features);
    if (nq == 0) return VO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = upload_features(c, c->op_slot0, nullptr, query, nq))) return rc;
    if ((rc = upload_features(c, c->op_slot1, nullptr, train, nt))) return rc;
    if ((rc = set_single_pair(c))) return rc;
    const int splits = knn_splits(1, std::max(nq, 1));
    launch_knn2(c->d_desc, (size_t)c->kp_stride * 32, c->d_nkp, c->d_qslot, c->d_tslot, c->kp_stride,
                c->d_partial, std::max(nq, 1), 1, splits, c->matcher_popcount, c->stream);
    launch_match_finalize(c->d_partial, c->d_kps, c->kp_stride, c->d_nkp, c->d_qslot, c->d_tslot, 0.7f, nullptr,
                          c->d_knn, c->d_good, c->d_ngood, 1, splits, c->stream);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_knn, sizeof(vo_knn2) * nq, hipMemcpyDefault, c->stream));
    return finish(c, "vo_knn2_hamming");
}

int vo_find_fundamental_ransac(vo_ctx *c, const float *p1, const float *p2, int n, double threshold,
                               double confidence, int max_iters, uint8_t *mask, double *F, int *found,
                               int *iters_run)
{
    if (!c) return VO_ERR_ARG;
    if (!p1 || !p2 || !mask || n < 0 || n > c->cfg.max_features)
        return fail(c, VO_ERR_ARG, "vo_find_fundamental_ransac: bad arguments (n %d, limit %d)", n,
                    c->cfg.max_features);
    if (n < 8)
        return fail(c, VO_ERR_UNSUPPORTED,
                    "vo_find_fundamental_ransac: %d points; the reference calls it with 8 or more", n);
    HIP_TRY(c, hipSetDevice(c->device));
    if (threshold <= 0) threshold = 3;
    if (confidence < 2.220446049250313e-16 || confidence > 1 - 2.220446049250313e-16) confidence = 0.99;
    std::vector<float> a(2 * (size_t)n), b(2 * (size_t)n);
    HIP_TRY(c, hipMemcpy(a.data(), p1, sizeof(float) * 2 * n, hipMemcpyDefault));
    HIP_TRY(c, hipMemcpy(b.data(), p2, sizeof(float) * 2 * n, hipMemcpyDefault));
    std::vector<vo_match> g((size_t)n);
    for (int i = 0; i < n; i++) {
        g[i].query = i; g[i].train = i; g[i].distance = 0;
        g[i].px = a[2 * i]; g[i].py = a[2 * i + 1];
        g[i].cx = b[2 * i]; g[i].cy = b[2 * i + 1];
    }
    HIP_TRY(c, hipMemcpy(c->d_good, g.data(), sizeof(vo_match) * n, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_ngood, &n, sizeof(int), hipMemcpyHostToDevice));
    launch_f_ransac(c->d_good, c->d_ngood, c->kp_stride, c->d_rng, c->rng_len,
                    FRansacParams{threshold, confidence, max_iters}, c->d_mask, c->d_F, c->d_info, c->d_final,
                    1, c->stream);
    int rc = finish(c, "vo_find_fundamental_ransac");
    if (rc) return rc;
    int info[4];
    HIP_TRY(c, hipMemcpy(info, c->d_info, sizeof(info), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(mask, c->d_mask, n, hipMemcpyDefault));
    if (F) HIP_TRY(c, hipMemcpy(F, c->d_F, sizeof(double) * 9, hipMemcpyDefault));
    if (found) *found = info[0];
    if (iters_run) *iters_run = info[1];
    return VO_OK;
}

int vo_match_features(vo_ctx *c, const vo_kp *prev_kps, const uint8_t *prev_desc, int n_prev, const vo_kp *curr_kps,
                      const uint8_t *curr_desc, int n_curr, float ratio, vo_match *out, int *n_out)
{
    return vo_match_features_ex(c, prev_kps, prev_desc, n_prev, curr_kps, curr_desc, n_curr, ratio, 0, out, n_out);
}

int vo_match_features_ex(vo_ctx *c, const vo_kp *prev_kps, const uint8_t *prev_desc, int n_prev,
                         const vo_kp *curr_kps, const uint8_t *curr_desc, int n_curr, float ratio, int cross_check,
                         vo_match *out, int *n_out)
{
    if (!c) return VO_ERR_ARG;
    if (!n_out || n_prev < 0 || n_curr < 0 || n_prev > c->cfg.max_features || n_curr > c->cfg.max_features)
        return fail(c, VO_ERR_ARG, "vo_match_features: bad arguments");
    *n_out = 0;
    // reference src/feature_matcher.cpp:21-23: empty in, empty out
    if (n_prev == 0 || n_curr == 0) return VO_OK;
    if (!prev_kps || !prev_desc || !curr_kps || !curr_desc || !out)
        return fail(c, VO_ERR_ARG, "vo_match_features: null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = upload_features(c, c->op_slot0, prev_kps, prev_desc, n_prev))) return rc;
    if ((rc = upload_features(c, c->op_slot1, curr_kps, curr_desc, n_curr))) return rc;
    if ((rc = set_single_pair(c))) return rc;
    c->enqueue_match(1, ratio, cross_check != 0, FRansacParams{3.0, 0.99, 1000});
    if ((rc = finish(c, "vo_match_features"))) return rc;
    int info[4];
    HIP_TRY(c, hipMemcpy(info, c->d_info, sizeof(info), hipMemcpyDeviceToHost));
    *n_out = info[3];
    if (info[3] > 0) HIP_TRY(c, hipMemcpy(out, c->d_final, sizeof(vo_match) * info[3], hipMemcpyDefault));
    return VO_OK;
}

int vo_ctx::ensure_stereo()
{
    vo_ctx *ctx = this;
    if (d_st_idx) return VO_OK;
    const size_t n = (size_t)kp_stride * cfg.max_batch;
    HIP_TRY(ctx, hipMalloc(&d_st_idx, sizeof(int32_t) * n));
    HIP_TRY(ctx, hipMalloc(&d_st_disp, sizeof(float) * n));
    HIP_TRY(ctx, hipMalloc(&d_st_depth, sizeof(float) * n));
    HIP_TRY(ctx, hipMalloc(&d_st_n, sizeof(int) * cfg.max_batch));
    HIP_TRY(ctx, hipMalloc(&d_st_index, stereo_index_bytes(cfg.max_batch, kp_stride, &st_rows_off)));
    return VO_OK;
}

static bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

int vo_stereo_depth(vo_ctx *c, const vo_kp *left_kps, const uint8_t *left_desc, int n_left, const vo_kp *right_kps,
                    const uint8_t *right_desc, int n_right, double fx, double baseline, float band, float min_disp,
                    float max_disp, float ratio, int32_t *right_idx, float *disparity, float *depth, float *plane,
                    int w, int h, int *n_matched)
{
    if (!c) return VO_ERR_ARG;
    if (n_left < 0 || n_right < 0 || n_left > c->cfg.max_features || n_right > c->cfg.max_features ||
        n_left > 65535 || n_right > 65535 || !(fx > 0) || !(baseline > 0) || !(min_disp > 0) || !(max_disp >= min_disp) ||
        !(band >= 0) || (plane && (w < 1 || h < 1)))
        return fail(c, VO_ERR_ARG, "vo_stereo_depth: bad arguments (n_left %d, n_right %d, limit %d)", n_left, n_right,
                    c->cfg.max_features);
    if ((n_left && (!left_kps || !left_desc || !right_idx || !disparity || !depth)) ||
        (n_right && (!right_kps || !right_desc)))
        return fail(c, VO_ERR_ARG, "vo_stereo_depth: null pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    if ((rc = c->ensure_stereo())) return rc;
    uint32_t *d_plane = nullptr;
    const size_t plane_px = plane ? (size_t)w * h : 0;
    if (plane) {
        if (is_device_pointer(plane)) {
            d_plane = reinterpret_cast<uint32_t *>(plane);
        } else {
            if (plane_px > c->st_plane_cap) {
                if (c->d_st_plane) (void)hipFree(c->d_st_plane);
                c->d_st_plane = nullptr; c->st_plane_cap = 0;
                HIP_TRY(c, hipMalloc(&c->d_st_plane, sizeof(float) * plane_px));
                c->st_plane_cap = plane_px;
            }
            d_plane = reinterpret_cast<uint32_t *>(c->d_st_plane);
        }
        HIP_TRY(c, hipMemsetAsync(d_plane, 0xff, sizeof(float) * plane_px, c->stream));  // all NaN
    }
    if (n_matched) *n_matched = 0;
    if (n_left > 0 && n_right > 0) {
        if ((rc = upload_features(c, c->op_slot0, left_kps, left_desc, n_left))) return rc;
        if ((rc = upload_features(c, c->op_slot1, right_kps, right_desc, n_right))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_st_n, 0, sizeof(int), c->stream));
        launch_stereo_match(c->d_kps, c->d_desc, c->d_nkp, c->kp_stride, c->op_slot0, c->op_slot1, 1, band, min_disp,
                            max_disp, ratio, fx, baseline, reinterpret_cast<uint16_t *>(c->d_st_index),
                            reinterpret_cast<int *>(c->d_st_index + c->st_rows_off), c->d_partial, c->d_partial_rev,
                            c->d_st_idx, c->d_st_disp, c->d_st_depth, d_plane, w, h, c->d_st_n, c->stream);
        if ((rc = finish(c, "vo_stereo_depth"))) return rc;
        HIP_TRY(c, hipMemcpy(right_idx, c->d_st_idx, sizeof(int32_t) * n_left, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(disparity, c->d_st_disp, sizeof(float) * n_left, hipMemcpyDefault));
        HIP_TRY(c, hipMemcpy(depth, c->d_st_depth, sizeof(float) * n_left, hipMemcpyDefault));
        if (n_matched) HIP_TRY(c, hipMemcpy(n_matched, c->d_st_n, sizeof(int), hipMemcpyDeviceToHost));
    } else {
        for (int i = 0; i < n_left; i++) { right_idx[i] = -1; disparity[i] = -1.f; depth[i] = nanf(""); }
        if ((rc = finish(c, "vo_stereo_depth"))) return rc;
    }
    if (plane && d_plane == reinterpret_cast<uint32_t *>(c->d_st_plane))
        HIP_TRY(c, hipMemcpy(plane, c->d_st_plane, sizeof(float) * plane_px, hipMemcpyDeviceToHost));
    return VO_OK;
}

}  // extern "C"

// ---- shared enqueue helpers ------------------------------------------------

void vo_ctx::enqueue_detect(const uint8_t *src, int w, int h, int src_stride, size_t src_frame_stride,
                            int channels, int max_features, int fast_threshold, int batch, int first_slot)
{
    mark(0);
    launch_frontend(src, w, h, src_stride, src_frame_stride, channels, PlaneBatch{nullptr, 0, 0},
                    PlaneBatch{nullptr, 0, 0}, plane(d_blur), fast_threshold, d_cand, d_cand_count, batch, gc,
                    stream, 3, pending_stamp);
    pending_stamp = nullptr;
    mark(1);
    mark(2);  // no pyramid in this mode
    mark(3);
    launch_select(d_cand, d_cand_count, d_cand_flat, w, h, max_features, kFBorder,
                  d_kps + (size_t)first_slot * kp_stride, kp_stride, d_nkp + first_slot, batch, stream);
    mark(4);
    launch_brief(plane(d_blur), d_kps + (size_t)first_slot * kp_stride, kp_stride, d_nkp + first_slot,
                 d_pattern, d_desc + (size_t)first_slot * kp_stride * 32, batch, stream);
}

int vo_ctx::ensure_orb(int w, int h, float scale_factor, int nlevels, int nfeatures, int batch)
{
    vo_ctx *ctx = this;
    if (orb.nlevels == nlevels && orb_w == w && orb_h == h && orb_scale == scale_factor &&
        orb_features == nfeatures && orb_batch == batch)
        return VO_OK;
    OrbPlan p;
    if (!orb_make_plan(w, h, scale_factor, nlevels, nfeatures, batch, &p))
        return fail(ctx, VO_ERR_UNSUPPORTED,
                    "ORB pyramid %dx%d, scale %.3f, %d levels, %d features is outside the supported range "
                    "(1 < scale <= 2, 1..8 levels, smallest level >= 7 px, <= 3840 features per level)",
                    w, h, (double)scale_factor, nlevels, nfeatures);
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    // a recorded batch chain holds the old plan's level tables and buffer addresses
    drop_graph();
    const size_t pyr_need = p.frame_bytes * batch;
    if (pyr_need > orb_pyr_cap) {
        if (d_pyr) (void)hipFree(d_pyr);
        if (d_pyr_blur) (void)hipFree(d_pyr_blur);
        d_pyr = d_pyr_blur = nullptr;
        orb_pyr_cap = 0;
        HIP_TRY(ctx, hipMalloc(&d_pyr, pyr_need));
        HIP_TRY(ctx, hipMalloc(&d_pyr_blur, pyr_need));
        orb_pyr_cap = pyr_need;
    }
    if (p.cand_total > orb_cand_cap) {
        if (d_orb_cand) (void)hipFree(d_orb_cand);
        if (d_orb_cand_flat) (void)hipFree(d_orb_cand_flat);
        d_orb_cand = d_orb_cand_flat = nullptr;
        orb_cand_cap = 0;
        HIP_TRY(ctx, hipMalloc(&d_orb_cand, sizeof(uint32_t) * p.cand_total));
        HIP_TRY(ctx, hipMalloc(&d_orb_cand_flat, sizeof(uint32_t) * p.cand_total));
        orb_cand_cap = p.cand_total;
    }
    if (p.count_total > orb_count_cap) {
        if (d_orb_cand_count) (void)hipFree(d_orb_cand_count);
        d_orb_cand_count = nullptr;
        orb_count_cap = 0;
        HIP_TRY(ctx, hipMalloc(&d_orb_cand_count, sizeof(uint32_t) * p.count_total));
        orb_count_cap = p.count_total;
    }
    const int lvl_need = batch * kOrbMaxLevels;
    if (lvl_need > orb_lvl_cap) {
        for (void *q : {(void *)d_lvl_ras, (void *)d_lvl_resp, (void *)d_lvl_n})
            if (q) (void)hipFree(q);
        d_lvl_ras = nullptr; d_lvl_resp = nullptr; d_lvl_n = nullptr;
        orb_lvl_cap = 0;
        HIP_TRY(ctx, hipMalloc(&d_lvl_ras, sizeof(uint32_t) * (size_t)lvl_need * kOrbLevelCap));
        HIP_TRY(ctx, hipMalloc(&d_lvl_resp, sizeof(float) * (size_t)lvl_need * kOrbLevelCap));
        HIP_TRY(ctx, hipMalloc(&d_lvl_n, sizeof(int) * lvl_need));
        orb_lvl_cap = lvl_need;
    }
    if (p.tab_total > orb_tab_cap) {
        if (d_orb_tab) (void)hipFree(d_orb_tab);
        d_orb_tab = nullptr;
        orb_tab_cap = 0;
        HIP_TRY(ctx, hipMalloc(&d_orb_tab, sizeof(uint32_t) * p.tab_total));
        orb_tab_cap = p.tab_total;
    }
    {
        std::vector<uint32_t> tab(p.tab_total, 0u);
        for (int l = 1; l < p.nlevels; l++) {
            orb_axis_table(p.lv[l - 1].w, p.lv[l].w, p.lv[l].pitch, tab.data() + p.lv[l].xtab_off);
            orb_axis_table(p.lv[l - 1].h, p.lv[l].h, (p.lv[l].h + kResizeRowPad - 1) / kResizeRowPad * kResizeRowPad,
                           tab.data() + p.lv[l].ytab_off);
        }
        HIP_TRY(ctx, hipMemcpy(d_orb_tab, tab.data(), sizeof(uint32_t) * p.tab_total, hipMemcpyHostToDevice));
    }
    orb = p;
    orb_w = w; orb_h = h; orb_scale = scale_factor; orb_features = nfeatures; orb_batch = batch;
    return VO_OK;
}

void vo_ctx::enqueue_detect_orb(const uint8_t *src, int w, int h, int src_stride, size_t src_frame_stride,
                                int channels, int fast_threshold, int batch, int first_slot)
{
    mark(0);
    auto level = [&](uint8_t *base, int l) { return PlaneBatch{base + orb.lv[l].off, orb.lv[l].pitch, orb.frame_bytes}; };
    auto fast_level = [&](int l) {
        const OrbLevel &L = orb.lv[l];
        // level 0 converts to gray on the way in and keeps the gray plane for the pyramid; corners closer than
        // kFBorder to the edge are dropped by the selection, so only their inner neighbours' scores are needed
        const uint8_t *in = l == 0 ? src : d_pyr + L.off;
        launch_frontend(in, L.w, L.h, l == 0 ? src_stride : L.pitch, l == 0 ? src_frame_stride : orb.frame_bytes,
                        l == 0 ? channels : 1, l == 0 ? level(d_pyr, 0) : PlaneBatch{nullptr, 0, 0},
                        PlaneBatch{nullptr, 0, 0}, level(d_pyr_blur, l), fast_threshold, d_orb_cand + L.cand_off,
                        d_orb_cand_count + L.count_off, batch, gc, stream, kFBorder - 1, l == 0 ? pending_stamp : nullptr);
        if (l == 0) pending_stamp = nullptr;
    };
    fast_level(0);
    mark(1);
    for (int l = 1; l < orb.nlevels; l++)
        launch_resize_exact(level(d_pyr, l - 1), orb.lv[l - 1].w, orb.lv[l - 1].h, level(d_pyr, l), orb.lv[l].w,
                            orb.lv[l].h, d_orb_tab + orb.lv[l].xtab_off, d_orb_tab + orb.lv[l].ytab_off, batch, stream);
    mark(2);
    // levels 1 .. n-1 in one launch (VO_ORB_LEVEL_LAUNCHES=1: one launch per level, for A/B measurements)
    static const bool per_level = [] { const char *e = getenv("VO_ORB_LEVEL_LAUNCHES"); return e && e[0] == '1'; }();
    if (per_level)
        for (int l = 1; l < orb.nlevels; l++) fast_level(l);
    else
        launch_frontend_levels(orb, 1, d_pyr, d_pyr_blur, d_orb_cand, d_orb_cand_count, fast_threshold, batch, gc,
                               stream, kFBorder - 1);
    mark(3);
    launch_orb_select(orb, batch, d_pyr, d_orb_cand, d_orb_cand_count, d_orb_cand_flat, d_lvl_ras, d_lvl_resp,
                      d_lvl_n, stream);
    launch_orb_finish(orb, batch, d_pyr, d_lvl_ras, d_lvl_resp, d_lvl_n, d_kps + (size_t)first_slot * kp_stride,
                      kp_stride, cfg.max_features, d_nkp + first_slot, stream);
    mark(4);
    launch_brief_steered(orb, batch, d_pyr_blur, d_kps + (size_t)first_slot * kp_stride, kp_stride,
                         d_nkp + first_slot, d_pattern0, d_desc + (size_t)first_slot * kp_stride * 32, stream);
}

void vo_ctx::enqueue_match(int pairs, float ratio, bool cross_check, FRansacParams prm)
{
    mark(5);
    const int splits = knn_splits(pairs, kp_stride);
    launch_knn2(d_desc, (size_t)kp_stride * 32, d_nkp, d_qslot, d_tslot, kp_stride, d_partial, kp_stride,
                pairs, splits, matcher_popcount, stream, pending_stamp);
    pending_stamp = nullptr;
    if (cross_check)  // the same pairs with the roles of query and train swapped
        launch_knn2(d_desc, (size_t)kp_stride * 32, d_nkp, d_tslot, d_qslot, kp_stride, d_partial_rev, kp_stride,
                    pairs, splits, matcher_popcount, stream);
    mark(6);
    launch_match_finalize(d_partial, d_kps, kp_stride, d_nkp, d_qslot, d_tslot, ratio,
                          cross_check ? d_partial_rev : nullptr, d_knn, d_good, d_ngood, pairs, splits, stream);
    mark(7);
    launch_f_ransac(d_good, d_ngood, kp_stride, d_rng, rng_len, prm, d_mask, d_F, d_info, d_final, pairs,
                    stream);
}
