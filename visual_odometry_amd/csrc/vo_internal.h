// Internal declarations shared by the kernel translation units and the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vo_hip.h"

namespace vo {

constexpr int kWave = 64;
constexpr int kSelectCap = 8192;   // keys the selection kernel can sort in LDS
constexpr int kMaxDim = 4096;      // largest width/height (row histogram bins)
constexpr int kResizeRowPad = 32;  // rows of a resize tile: the row tables of the pyramid are padded to it
constexpr int kKnnSplits = 2;      // train-set splits per query block
constexpr int kKnnSplitsFew = 8;   // ... when the launch has few pairs (the partial buffers hold max(2 x batch, 8) splits)
constexpr int kFBorder = 31;       // ORB edgeThreshold
constexpr int kBriefReach = 18;    // the rotated rBRIEF pattern samples at most this far from a key point
#ifndef VO_TILE_H
#define VO_TILE_H 32   // (64 builds the tile kernel with 128 x 64 tiles: four quads per thread)
#endif
constexpr int kTileW = 128, kTileH = VO_TILE_H;  // tile of the fused image kernel
constexpr int kTileCap = kTileW * kTileH / 4;  // corners one tile can emit (3x3 strict maxima: one per 2x2 block)
inline int tiles_of(int w, int h) { return ((w + kTileW - 1) / kTileW) * ((h + kTileH - 1) / kTileH); }
constexpr int kOrbMaxLevels = 8;   // pyramid levels of the full ORB detector
constexpr int kOrbLevelCap = 4096; // keypoints one level can contribute

// floor(x / d) without a run-time division: umulhi(x, magic_inv(d)), exact while x * d < 2^32.  d = 1 has no
// 32-bit magic number and is encoded as 0.
inline uint32_t magic_inv(int d) { return d > 1 ? (uint32_t)(0x100000000ull / (uint32_t)d) + 1u : 0u; }
__device__ __forceinline__ uint32_t magic_div(uint32_t x, uint32_t inv) { return inv ? __umulhi(x, inv) : x; }

// gray conversion weights (B, G, R, shift)
struct GrayCoeffs {
    int b, g, r, shift;
};

// A batch of equally sized 8-bit planes: plane i starts at base + i*plane_stride.
struct PlaneBatch {
    uint8_t *base;
    int pitch;            // bytes per row
    size_t plane_stride;  // bytes between consecutive planes
};

// One pyramid level of the full ORB detector.
struct OrbLevel {
    int w, h, pitch;   // level image; rows are `pitch` bytes apart
    int quota;         // keypoints wanted from this level
    float scale;       // level-0 pixels per level pixel
    int tiles;         // tiles of the fused image kernel on this level
    size_t off;        // byte offset of the level inside a frame's pyramid
    size_t cand_off;   // element offset of the level's corner lists ([frame][tiles][kTileCap])
    size_t count_off;  // element offset of the level's per-tile corner counts ([frame][tiles])
    size_t xtab_off, ytab_off;  // element offsets of the level's resize tables (pitch / padded-height entries)
};

// Geometry of the pyramid for one (frame size, scale factor, levels, nfeatures, batch).
struct OrbPlan {
    int nlevels;
    size_t frame_bytes;  // bytes of one frame's pyramid (all levels, padded to whole tiles)
    size_t cand_total;   // corner-list elements for the whole batch
    size_t count_total;  // per-tile counters for the whole batch
    size_t tab_total;    // resize-table entries of all levels
    OrbLevel lv[kOrbMaxLevels];
};

// false when the configuration is outside what the kernels support
// (1 < scale_factor <= 2, 1..8 levels, smallest level >= 7 px, quota per level <= 3840)
bool orb_make_plan(int w, int h, float scale_factor, int nlevels, int nfeatures, int batch, OrbPlan *out);

// per-axis table of cv::resize(INTER_LINEAR_EXACT): offset | weight << 16, `padded` entries
void orb_axis_table(int src_n, int dst_n, int padded, uint32_t *out);
// cv::resize(INTER_LINEAR_EXACT) of every plane of `src` into `dst` (tables in device memory)
void launch_resize_exact(PlaneBatch src, int sw, int sh, PlaneBatch dst, int dw, int dh, const uint32_t *xtab,
                         const uint32_t *ytab, int batch, hipStream_t s);
// per (level, frame): border filter, retainBest(2*quota) by FAST score, Harris
// response, retainBest(quota); survivors in raster order
void launch_orb_select(const OrbPlan &plan, int batch, const uint8_t *pyr, const uint32_t *cand,
                       const uint32_t *tile_count, uint32_t *flat, uint32_t *lvl_ras, float *lvl_resp, int *lvl_n,
                       hipStream_t s);
// concatenate the levels (at most kp_cap key points per frame), intensity-centroid angle, cv::KeyPoint records
void launch_orb_finish(const OrbPlan &plan, int batch, const uint8_t *pyr, const uint32_t *lvl_ras,
                       const float *lvl_resp, const int *lvl_n, vo_kp *kps, int kp_stride, int kp_cap, int *nkp,
                       hipStream_t s);
// rBRIEF-256 steered by each keypoint's angle, sampled on its level's blurred image
void launch_brief_steered(const OrbPlan &plan, int batch, const uint8_t *pyr_blur, const vo_kp *kps, int kp_stride,
                          const int *nkp, const float *pattern, uint8_t *desc, hipStream_t s);

// ---- launchers (each enqueues on `s`, no sync) ---------------------------

// BGR/BGRA -> gray for `batch` frames; src rows are src_stride bytes apart,
// frames src_frame_stride bytes apart.
void launch_gray(const uint8_t *src, int width, int height, int src_stride,
                 size_t src_frame_stride, int channels, PlaneBatch dst,
                 int batch, GrayCoeffs c, hipStream_t s);

// Fused tile kernel: (optional BGR[A]->gray) + FAST-9/16 with 3x3 strict
// non-max suppression + 7x7 fixed-point Gaussian.  With T = tiles_of(width,
// height), tile t of frame f writes its surviving corners to
// cand[(f*T + t)*kTileCap ...] as keys ((255-score) << 24) | y << 12 | x and
// their number to tile_count[f*T + t] (no initialisation needed).
// Any of gray_out/score/blur may have base == 0, cand may be null; `score`
// receives the per-pixel score map (diagnostic).  Corners are looked for at
// least `margin` px inside the image (cv::FAST: 3); a caller that drops a
// border of b px afterwards may pass b - 1 (the neighbours of a kept corner
// still get their scores).
void launch_frontend(const uint8_t *src, int width, int height, int src_stride,
                     size_t src_frame_stride, int channels, PlaneBatch gray_out,
                     PlaneBatch score, PlaneBatch blur, int threshold,
                     uint32_t *cand, uint32_t *tile_count, int batch,
                     GrayCoeffs c, hipStream_t s, int margin = 3, unsigned long long *stamp = nullptr);

// launch_frontend for the gray planes of pyramid levels first_level .. nlevels-1 in one launch (input: the gray
// pyramid, output: the blurred pyramid and the levels' corner lists, laid out as the plan says)
void launch_frontend_levels(const OrbPlan &plan, int first_level, const uint8_t *pyr, uint8_t *pyr_blur,
                            uint32_t *cand, uint32_t *tile_count, int threshold, int batch, GrayCoeffs c,
                            hipStream_t s, int margin);

// Keep the max_features best keys per frame (all of them, in raster order,
// when there are not more than max_features), drop keypoints closer than
// `border` px to the image edge, write vo_kp records.  cand/tile_count as
// written by launch_frontend; flat is scratch of the same size as cand.
void launch_select(const uint32_t *cand, const uint32_t *tile_count, uint32_t *flat,
                   int width, int height, int max_features,
                   int border, vo_kp *kps, int kp_stride, int *nkp,
                   int batch, hipStream_t s);

// rBRIEF-256 on the blurred plane with a fixed integer pattern
// (512 x (dx, dy) int8 in device memory).
void launch_brief(PlaneBatch blur, const vo_kp *kps, int kp_stride,
                  const int *nkp, const int8_t *pattern, uint8_t *desc,
                  int batch, hipStream_t s);

// Hamming kNN-2. For pair p the query set is (q_desc + q_slot[p]*slot_stride,
// n[q_slot[p]]) and the train set likewise. partial: [pair][split][kp_stride] uint2, `splits` of them per pair
// (knn_splits(pairs, max_n): kKnnSplits, or kKnnSplitsFew for launches of a few pairs; the same number goes to
// launch_match_finalize).
// popcount: the XOR / popcount kernel on the vector ALU instead of the matrix-core kernel (same results).
int knn_splits(int pairs, int max_n);
void launch_knn2(const uint8_t *desc, size_t slot_stride_bytes, const int *nkp,
                 const int *q_slot, const int *t_slot, int kp_stride,
                 uint2 *partial, int max_n, int pairs, int splits, bool popcount, hipStream_t s,
                 unsigned long long *stamp = nullptr);

// ImageProcessor stages (reference src/image_processor.cpp:41-79), one plane.
void launch_hist256(const uint8_t *img, int w, int h, int pitch, uint32_t *hist, hipStream_t s);
void launch_equalize_lut(const uint32_t *hist, int total, uint8_t *lut, hipStream_t s);
// src -> [lut] -> enh; separable 8.8 fixed-point Gaussian (n_taps odd, <= 31,
// taps sum to 256) -> den; den with rows outside [h/4, 3h/4] zeroed -> msk.
// Null outputs are skipped.  Planes must be padded to whole 128x16 tiles.
void launch_enhance_blur_mask(const uint8_t *src, int w, int h, int pitch, const uint8_t *lut, int n_taps,
                              const uint16_t *taps, uint8_t *enh, uint8_t *den, uint8_t *msk, hipStream_t s);

// Stereo front end, batched over `frames` pairs: left key points in feature slots
// l_slot0 + f, right ones in r_slot0 + f.  Both sets are listed by image row
// (order / row_start: scratch of stereo_index_bytes(frames, kp_stride) bytes, the
// row table *rows_off bytes into it), then band-limited Hamming matching
// left->right and right->left over the rows of each query's band (two smallest
// keys per query in fwd / bwd, frames x kp_stride), then ratio test, cross-check,
// disparity, depth = fx * baseline / disparity (per-key-point arrays of
// frames x kp_stride, may be null) and an atomic-minimum scatter into the
// frames x h x w depth planes (all 0xff bytes = NaN beforehand; may be null).
// n_matched (frames entries, may be null) must be zero on entry.
void launch_stereo_match(const vo_kp *kps, const uint8_t *desc, const int *nkp, int kp_stride, int l_slot0,
                         int r_slot0, int frames, float band, float min_disp, float max_disp, float ratio, double fx,
                         double baseline, uint16_t *order, int *row_start, uint2 *fwd, uint2 *bwd, int32_t *right_idx,
                         float *disparity, float *depth, uint32_t *plane, int w, int h, int *n_matched, hipStream_t s);
size_t stereo_index_bytes(int frames, int kp_stride, size_t *rows_off);
// Puts NaN back at every pixel a left key point of the batch rounds to.
void launch_stereo_clear(const vo_kp *kps, const int *nkp, int kp_stride, int l_slot0, int frames, uint32_t *plane,
                         int w, int h, hipStream_t s);

// cv::undistort of a gray plane (k = k1 k2 p1 p2 k3 k4 k5 k6); dst rows must be
// padded to a multiple of 128 bytes
void launch_undistort(const uint8_t *src, int w, int h, int spitch, double fx, double fy, double cx, double cy,
                      const double *k, uint8_t *dst, int dpitch, hipStream_t s);

// Merge the split partials into vo_knn2 records, apply the ratio test (and,
// when partial_rev holds the kNN of the swapped pair, the mutual-nearest
// cross-check) and write the surviving matches in query order.
void launch_match_finalize(const uint2 *partial, const vo_kp *kps, int kp_stride,
                           const int *nkp, const int *q_slot, const int *t_slot,
                           float ratio, const uint2 *partial_rev, vo_knn2 *knn,
                           vo_match *good, int *n_good, int pairs, int splits, hipStream_t s);

struct FRansacParams {
    double threshold, confidence;
    int max_iters;
};

// 7-point fundamental-matrix RANSAC per pair over good[pair*kp_stride ..],
// followed by the 100 px motion filter.  Outputs: mask (per ratio survivor),
// F (9 doubles), info = {found, iters_run, n_inliers, n_final}, final matches.
void launch_f_ransac(const vo_match *good, const int *n_good, int kp_stride,
                     const uint32_t *rng_raw, int rng_len, FRansacParams prm,
                     uint8_t *mask, double *F, int *info, vo_match *final_m,
                     int pairs, hipStream_t s);


// Essential matrix.  q: n x (x1, y1, x2, y2) normalised points (doubles).
// Hypotheses from an index table idx (H x 5): E (H x 10 x 9, unit Frobenius
// norm), n_models (H), counts (H x 10) = points with Sampson error <= thr2.
void launch_e_hypotheses(const double *q, int n, const int *idx, int H, double *E, int *n_models, float thr2,
                         int *counts, hipStream_t s);
// cv::findEssentialMat's RANSAC on one point set; info = {found, iters_run, n_inliers}
void launch_e_ransac(const double *q, int n, const uint32_t *rng_raw, int rng_len, double prob, float thr2,
                     int max_iters, uint8_t *mask, double *E, int *info, hipStream_t s);
void launch_recover_pose(const double *E, const double *q, int n, double dist, int use_mask, uint8_t *mask,
                         uint8_t *votes, double *out, int *info, hipStream_t s);

// ---- pose stage -------------------------------------------------------------

// Per stream, sequentially over its frames: track ages of the current
// keypoints from the matches, and the multi-view filter decision.
void launch_track_ages(const vo_match *final_m, const int *info, int kp_stride, const int *q_slot,
                       const int *t_slot, const int *has_prev, int streams, int frames, const int *nkp,
                       int *ages, int *filter_flag, hipStream_t s, unsigned long long *stamp = nullptr);

// Per frame: apply the multi-view filter (-> tracked), then back-project the
// current points through the depth map (-> valid + pts3). out_info = {n_tracked, n_3d}.
void launch_filter_backproject(const vo_match *final_m, const int *info, int kp_stride, const int *t_slot,
                               const int *has_prev, const int *nkp, const int *ages, const int *filter_flag,
                               const float *depth, int width, int height, int depth_stride,
                               size_t depth_frame_stride, double fx, double fy, double cx, double cy,
                               vo_match *tracked, vo_match *valid, float *pts3, int *out_info, int batch,
                               hipStream_t s, int *clear3 = nullptr, bool depth_u16 = false);

// cv::solvePnPRansac per frame on (pts3 -> prev 2-D point). pose = {success,
// n_inliers, iters_run, n_points}; rt = rvec[3], tvec[3].
// launch_frontend, launch_knn2 and launch_track_ages take an optional `stamp`: the first block of the launch
// stores the device wall clock there (the packed output's detect / match times), launch_filter_backproject an
// optional `clear3`: 3 x batch words it zeroes (the hand-over state of launch_pnp_ransac, pnp_share_words()).
// phase2: device scratch of pnp_phase2_bytes(batch) bytes for the frames that need many hypotheses (their
// hypotheses are spread over the blocks of the launch and of a follow-up launch); null keeps every frame in its
// own block.
void launch_pnp_ransac(const vo_match *valid, const float *pts3, const int *bp_info, int kp_stride,
                       const uint32_t *rng_raw, int rng_len, double fx, double fy, double cx, double cy,
                       int max_iters, double threshold, double confidence, int *inliers, int *pose,
                       double *rt, int batch, void *phase2, hipStream_t s, bool zeroed = false);
size_t pnp_phase2_bytes(int batch, size_t off[4]);
// the 3 x batch words at the start of that scratch which must be zero when launch_pnp_ransac runs; with
// zeroed == true the caller has seen to it (launch_filter_backproject's clear3), otherwise the launch does
inline int *pnp_share_words(void *phase2) { return static_cast<int *>(phase2); }

// Copy the features and ages of each stream's last frame into its carry slot.
void launch_carry(vo_kp *kps, uint8_t *desc, int *nkp, int *ages, int kp_stride, int streams, int frames,
                  hipStream_t s);

// Gather the per-frame counters and poses into vo_frame_result records.
void launch_collect(const int *nkp, const int *t_slot, const int *n_good, const int *info,
                    const int *bp_info, const int *pose, const double *rt, vo_frame_result *out, int batch,
                    hipStream_t s);

// Packed per-frame output block: header (vo_frame_result + three device wall-clock stamps), then the arrays at
// fixed offsets for `cap` (a multiple of 8) records each.
struct PackLayout {
    int cap;
    size_t off_kps, off_desc, off_m[3], off_pts3, off_inl, frame_bytes;
};
inline PackLayout pack_layout(int cap)
{
    PackLayout L;
    L.cap = cap;
    L.off_kps = 128;
    L.off_desc = L.off_kps + sizeof(vo_kp) * (size_t)cap;
    L.off_m[0] = L.off_desc + 32 * (size_t)cap;
    L.off_m[1] = L.off_m[0] + sizeof(vo_match) * (size_t)cap;
    L.off_m[2] = L.off_m[1] + sizeof(vo_match) * (size_t)cap;
    L.off_pts3 = L.off_m[2] + sizeof(vo_match) * (size_t)cap;
    L.off_inl = L.off_pts3 + 12 * (size_t)cap;
    L.frame_bytes = L.off_inl + 4 * (size_t)cap;
    return L;
}
struct PackSrc {
    const vo_frame_result *res;
    const vo_kp *kps;       // slot 0 of the feature slots
    const uint8_t *desc;
    const int *t_slot;      // feature slot of each frame
    const vo_match *m[3];   // final, tracked, valid
    const float *pts3;
    const int *inliers;
    const unsigned long long *stamps;  // 3 device wall-clock values of the batch
    int kp_stride;
};
void launch_pack_frames(const PackSrc &src, PackLayout layout, uint8_t *out, int batch, hipStream_t s);
struct CollectSrc {
    const int *nkp, *t_slot, *n_good, *info, *bp_info, *pose;
    const double *rt;
    vo_frame_result *out;
};
// launch_collect + launch_pack_frames + launch_carry in one launch
void launch_finish_packed(const CollectSrc &cs, const PackSrc &src, PackLayout layout, uint8_t *out, int batch,
                          vo_kp *kps, uint8_t *desc, int *nkp, int *ages, int streams, int frames, hipStream_t s);

#ifdef __HIPCC__
// Corner keys: (255 - score) << 24 | y << 12 | x.  Ascending key order is score descending, then raster order
// (frames are at most kMaxDim = 4096 wide and high, so x and y take 12 bits each).
static_assert(kMaxDim <= 4096, "corner keys hold 12-bit coordinates");
__device__ __forceinline__ uint32_t corner_key(uint32_t score, int x, int y)
{
    return ((255u - score) << 24) | ((uint32_t)y << 12) | (uint32_t)x;
}
__device__ __forceinline__ int key_x(uint32_t k) { return (int)(k & 0xfffu); }
__device__ __forceinline__ int key_y(uint32_t k) { return (int)((k >> 12) & 0xfffu); }

// Inclusive prefix sum of one value per thread over the whole block (THREADS a
// multiple of 64, at most 1024): wave scans by shuffles, then the waves'
// totals.  s_part: LDS scratch for THREADS / 64 words.  Every thread of the
// block must call it (two barriers inside).
template <int THREADS>
__device__ inline uint32_t block_scan_inclusive(uint32_t v, uint32_t *s_part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) s_part[wave] = x;
    __syncthreads();
    for (int w = 0; w < wave; w++) x += s_part[w];
    __syncthreads();
    return x;
}

// One block gathers the per-tile corner lists of a frame (tile t holds
// count[t] keys at cand[t*kTileCap ...]) into the contiguous list `flat` and
// returns its length.  s_off: LDS scratch for ntiles words, s_scan: THREADS / 64
// words.  Ends with a barrier: every thread may read flat[] afterwards.
template <int THREADS>
__device__ inline int flatten_tile_lists(const uint32_t *__restrict__ cand, const uint32_t *__restrict__ count,
                                         int ntiles, uint32_t *__restrict__ flat, uint32_t *s_off,
                                         uint32_t *s_scan)
{
    __shared__ uint32_t s_total;
    const int tid = threadIdx.x;
    for (int t = tid; t < ntiles; t += THREADS) s_off[t] = min(count[t], (uint32_t)kTileCap);
    __syncthreads();
    const int per = (ntiles + THREADS - 1) / THREADS;
    const int t0 = min(tid * per, ntiles), t1 = min(t0 + per, ntiles);
    uint32_t sum = 0;
    for (int t = t0; t < t1; t++) sum += s_off[t];
    const uint32_t inc = block_scan_inclusive<THREADS>(sum, s_scan);
    uint32_t run = inc - sum;
    for (int t = t0; t < t1; t++) {
        const uint32_t n = s_off[t];
        s_off[t] = run;
        run += n;
    }
    if (tid == THREADS - 1) s_total = inc;
    __syncthreads();
    const uint32_t total = s_total;
    const int lane = tid & 63;
    // a wave copies four tiles per step: the first 64 keys of each are loaded (at a clamped index, so the loads
    // need no guard and are all in flight together) before any is stored; longer lists finish in a tail loop
    constexpr int WAVES = THREADS / 64, U = 4;
    for (int t = tid >> 6; t < ntiles; t += WAVES * U) {
        uint32_t v[U], o[U], n[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int tt = min(t + u * WAVES, ntiles - 1);
            o[u] = s_off[tt];
            n[u] = t + u * WAVES < ntiles ? (tt + 1 < ntiles ? s_off[tt + 1] : total) - o[u] : 0u;
            v[u] = cand[(size_t)tt * kTileCap + min((uint32_t)lane, max(n[u], 1u) - 1u)];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if ((uint32_t)lane < n[u]) flat[o[u] + lane] = v[u];
#pragma unroll
        for (int u = 0; u < U; u++)
            for (uint32_t j = lane + 64; j < n[u]; j += 64)
                flat[o[u] + j] = cand[(size_t)min(t + u * WAVES, ntiles - 1) * kTileCap + j];
    }
    __threadfence_block();
    __syncthreads();
    return (int)total;
}
#endif

}  // namespace vo
