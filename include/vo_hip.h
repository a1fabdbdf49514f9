/* vo_hip.h -- C ABI of libvo_hip.so, the MI355X (gfx950) implementation of the
 * visual-odometry front-end hot path: gray -> FAST-9/16 + NMS -> top-N ->
 * 7x7 blur -> rBRIEF-256 -> Hamming kNN-2 -> ratio test -> fundamental-matrix
 * RANSAC filter -> motion filter -> depth back-projection -> PnP RANSAC pose.
 *
 * Every entry point replaces one call the reference makes into OpenCV (the
 * reference's own code is glue around those calls); the reference file:line of
 * the call site is cited at each declaration.  All arguments are plain
 * pointers and sizes.  Unless stated otherwise a data pointer may address host
 * memory or device memory of the context's GPU (copies use hipMemcpyDefault).
 *
 * Conventions
 *  - every function returns VO_OK (0) or a negative VO_ERR_* code and never
 *    throws; vo_last_error() returns a message for the last failure;
 *  - a vo_ctx is bound to one GPU and one HIP stream and is not re-entrant:
 *    one context per camera stream / caller thread, like one FrameProcessor in
 *    the reference (reference src/vo_node.cpp:940-962);
 *  - there is no CPU fallback: if the GPU or the code object is unusable,
 *    vo_create() fails.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_OK 0
#define VO_ERR_ARG (-1)         /* bad argument / size beyond the context limits */
#define VO_ERR_HIP (-2)         /* a HIP runtime call failed */
#define VO_ERR_STATE (-3)       /* call order violated */
#define VO_ERR_UNSUPPORTED (-4) /* documented gap */

#define VO_MAX_FEATURES_LIMIT 6144 /* hard upper bound of vo_cfg.max_features */

typedef struct vo_ctx vo_ctx;

/* cv::KeyPoint as a POD: pt.x, pt.y, size, angle, response, octave. */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
} vo_kp;

/* the two nearest train descriptors of one query (idx -1 when absent) */
typedef struct {
    int32_t idx0, dist0;
    int32_t idx1, dist1;
} vo_knn2;

/* one cv::DMatch(queryIdx, trainIdx, distance) plus both image points */
typedef struct {
    int32_t query, train;
    float distance;
    float px, py; /* point in the previous (query) frame */
    float cx, cy; /* point in the current (train) frame  */
} vo_match;

typedef struct {
    int max_width, max_height; /* largest frame, width*height <= 2^24 */
    int max_features;          /* largest keypoint budget per frame */
    int max_batch;             /* largest number of frames per vo_batch_run */
    /* BGR->gray fixed-point weights; all 0 selects OpenCV 4.x's
     * (3735, 19235, 9798) >> 15 */
    int gray_b, gray_g, gray_r, gray_shift;
} vo_cfg;

/* ---- context -------------------------------------------------------- */

/* One context = one camera stream's device state (buffers, HIP stream). Takes
 * the place of constructing FeatureDetector / FeatureMatcher / FrameProcessor
 * (reference src/vo_node.cpp:79-80,170-173).  Returns NULL on failure (see
 * vo_last_error(NULL)). */
vo_ctx *vo_create(int device, const vo_cfg *cfg);
void vo_destroy(vo_ctx *ctx);
const char *vo_last_error(const vo_ctx *ctx);
/* library/ABI version, bumped on incompatible change */
int vo_abi_version(void);

/* ---- single-frame operators (synchronous) --------------------------- */

/* cv::cvtColor(src, dst, COLOR_BGR2GRAY) for 8-bit BGR (channels 3) or BGRA
 * (channels 4).  reference src/frame_processor.cpp:204-208, :223,
 * src/feature_detector.cpp:46, src/image_processor.cpp:35 */
int vo_gray(vo_ctx *ctx, const uint8_t *src, int width, int height,
            int src_stride, int channels, uint8_t *dst, int dst_stride);

/* The per-pixel half of cv::FAST(gray, kps, threshold, true): FAST-9/16 corner
 * score of every pixel, 0 where the segment test fails.  Diagnostic view of
 * the detector's first kernel.  reference src/feature_detector.cpp:59 */
int vo_fast_score(vo_ctx *ctx, const uint8_t *gray, int width, int height,
                  int stride, int threshold, uint8_t *score, int score_stride);

/* The smoothing cv::ORB::compute applies before sampling descriptors:
 * GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101 in 8-bit fixed point.
 * reference src/feature_detector.cpp:73 */
int vo_blur7(vo_ctx *ctx, const uint8_t *gray, int width, int height,
             int stride, uint8_t *dst, int dst_stride);

/* FeatureDetector::detectFeatures(gray, max_features, fast_threshold) on a
 * 1-channel frame: cv::FAST with non-max suppression, keep the max_features
 * strongest, cv::ORB::compute (border filter 31 px, blur, rBRIEF-256).
 * reference src/feature_detector.cpp:37-88
 * When more than max_features corners exist they are ordered by (response
 * descending, raster index ascending); otherwise raster order is kept.
 * kps must hold max_features entries, desc 32*max_features bytes. */
int vo_detect(vo_ctx *ctx, const uint8_t *gray, int width, int height,
              int stride, int max_features, int fast_threshold, vo_kp *kps,
              uint8_t *desc, int *n_out);

/* Which kernel computes the all-pairs Hamming distances of this context
 * (vo_knn2_hamming, vo_match_features[_ex], the batched pipeline).  No
 * counterpart in the reference (cv::BFMatcher picks its own loops): for
 * measurements.  Both give identical results.
 *   VO_MATCHER_MFMA      descriptor bits as int8 vectors on the matrix cores
 *                        (default, the faster one);
 *   VO_MATCHER_POPCOUNT  XOR + popcount on the vector ALU. */
#define VO_MATCHER_MFMA 0
#define VO_MATCHER_POPCOUNT 1
int vo_set_matcher(vo_ctx *ctx, int kind);

/* cv::BFMatcher(NORM_HAMMING, false).knnMatch(query, train, k = 2) on
 * 256-bit descriptors; equal distances keep the lower train index first.
 * reference src/feature_matcher.cpp:27-31 */
int vo_knn2_hamming(vo_ctx *ctx, const uint8_t *query, int n_query,
                    const uint8_t *train, int n_train, vo_knn2 *out);

/* vo_knn2_hamming for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It
 * has no counterpart in the reference (whose cv::BFMatcher allocates freely,
 * src/feature_matcher.cpp:27-31).  Same kernels and results as vo_knn2_hamming,
 * but no vo_ctx: every buffer beyond the caller's inputs and output (descriptor
 * slots, per-split partial lists, merged records, control words; 385 KiB) is a
 * static device array inside the library, present on every GPU the process
 * uses.  The call makes no hipMalloc / hipFree and creates no stream or event.
 *   device      GPU to run on (made current);
 *   hip_stream  a hipStream_t to enqueue on, NULL = the device's default stream;
 *   query/train/out  host or device memory of that GPU (copies use
 *               hipMemcpyDefault); with wait == 0 they must stay valid until the
 *               stream has run the work;
 *   wait        non-zero: return after the stream has finished the call.
 * n_query and n_train may not exceed VO_KNN2_STATIC_MAX (VO_ERR_ARG otherwise;
 * vo_static_last_error() holds the message).  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (two threads, two streams) race;
 * serialise them.
 * To record a call into a HIP graph (hipStreamBeginCapture on hip_stream) three
 * conditions hold, because stream capture accepts neither module loading, nor
 * synchronisation, nor copies from pageable host memory: (1) one call on that
 * device has been made before the capture starts -- n_query = 0 is enough, it
 * only looks up the array's address; (2) query, train and out are device
 * memory or pinned host memory; (3) wait == 0. */
#define VO_KNN2_STATIC_MAX 4096
int vo_knn2_hamming_static(int device, void *hip_stream, const uint8_t *query, int n_query,
                           const uint8_t *train, int n_train, vo_knn2 *out, int wait);
/* message of the calling thread's last failed vo_knn2_hamming_static call */
const char *vo_static_last_error(void);

/* cv::findFundamentalMat(p1, p2, FM_RANSAC, threshold, confidence, mask)
 * with max_iters (OpenCV default 1000) and cv::RNG seed 2^64-1.
 * p1, p2: n x 2 float32; mask: n bytes; F: 9 doubles row-major.
 * *found is 0 when no model with more than 6 inliers exists (mask all 0).
 * With 8..14 points OpenCV runs its LMedS estimator instead (least median
 * error, threshold derived from that median) and so does this call; the mask
 * is then written even when *found is 0 (fewer than 7 inliers), as OpenCV
 * leaves it.  Fewer than 8 points -> VO_ERR_UNSUPPORTED.
 * reference src/feature_matcher.cpp:52-58 */
int vo_find_fundamental_ransac(vo_ctx *ctx, const float *p1, const float *p2,
                               int n, double threshold, double confidence,
                               int max_iters, uint8_t *mask, double *F,
                               int *found, int *iters_run);

/* FeatureMatcher::match(prev, curr, ., .): kNN-2, Lowe ratio (float32),
 * F-RANSAC mask (3.0 px, 0.99), motion < 100 px.  Query = prev, train = curr.
 * out must hold n_prev entries.  An empty result (fewer than 8 ratio
 * survivors, or no RANSAC model) returns VO_OK with *n_out = 0.
 * reference src/feature_matcher.cpp:15-104 */
int vo_match_features(vo_ctx *ctx, const vo_kp *prev_kps, const uint8_t *prev_desc,
                      int n_prev, const vo_kp *curr_kps, const uint8_t *curr_desc,
                      int n_curr, float ratio, vo_match *out, int *n_out);

/* The full ORB detector: what the cv::ORB object the reference creates
 * (cv::ORB::create(max_features, scale_factor, n_levels), reference
 * src/feature_detector.cpp:18-22, setters :97-123) computes in
 * detectAndCompute(): an n-level INTER_LINEAR_EXACT pyramid, per level FAST
 * (threshold, NMS) -> 31 px border -> best 2*quota by FAST score -> Harris
 * response -> best quota (boundary ties kept), intensity-centroid angle,
 * pt scaled to level-0 coordinates, rBRIEF steered by the angle on the 7x7
 * Gaussian of the level.  The reference itself never calls this path (it
 * feeds cv::FAST corners to ORB::compute, :59-73 = vo_detect); it is what the
 * setMaxFeatures / setScaleFactor / setNLevels interface describes.
 * Keypoints come out level by level, raster order inside a level (OpenCV's
 * order inside a level is unspecified).  At most min(cap, the context's
 * feature capacity) keypoints are returned.  1 < scale_factor <= 2,
 * 1 <= nlevels <= 8. */
int vo_detect_orb(vo_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                  int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                  vo_kp *kps, uint8_t *desc, int cap, int *n_out);

/* Diagnostic: copy pyramid level `level` of frame `frame` of the last ORB
 * detection (operator or batch) to host memory; blurred != 0 selects the
 * 7x7-Gaussian version the descriptors were sampled from.  Of the blurred
 * levels above level 0 only the pixels a descriptor can sample are defined
 * (key points lie 31 px inside a level, the rotated pattern reaches 18 px): the
 * outermost 13 columns and rows on the right and at the bottom may hold stale
 * data.  dst may be null to query the level size only. */
int vo_orb_get_level(vo_ctx *ctx, int frame, int level, int blurred, uint8_t *dst,
                     int dst_stride, int *w_out, int *h_out);

/* cv::findEssentialMat(p1, p2, K, cv::RANSAC, prob, threshold, maxIters,
 * mask) on pixel points (n x 2 float32): points are normalised with K, the
 * threshold (pixels) is divided by (fx + fy) / 2, then the five-point RANSAC
 * with Sampson error, cv::RNG seed 2^64-1.  E: 9 doubles row-major, Frobenius
 * norm 1, p2^T K^-T E K^-1 p1 = 0.  No counterpart in the reference, which
 * takes its pose from cv::solvePnPRansac (src/frame_processor.cpp:176-196):
 * this is the monocular-pose estimator on top of it.  The five-point solutions
 * agree with OpenCV's as a set (Nister's construction with a deterministic
 * real-root finder), not bitwise. */
int vo_find_essential_ransac(vo_ctx *ctx, const float *p1, const float *p2, int n,
                             double fx, double fy, double cx, double cy, double prob,
                             double threshold, int max_iters, uint8_t *mask, double *E,
                             int *found, int *iters_run);

/* Batched hypothesis generation and scoring from a fixed sample table:
 * samples holds n_hyp x 5 point indices; for each sample the five-point
 * solutions E (n_hyp x 10 x 9 doubles, unused entries 0), their number
 * n_models (n_hyp) and per solution the number of the n points whose Sampson
 * error is <= (threshold / ((fx + fy) / 2))^2 (counts, n_hyp x 10). */
int vo_essential_hypotheses(vo_ctx *ctx, const float *p1, const float *p2, int n,
                            double fx, double fy, double cx, double cy,
                            const int32_t *samples, int n_hyp, double threshold,
                            double *E, int32_t *n_models, int32_t *counts);

/* cv::recoverPose(E, p1, p2, K, R, t, mask) with distanceThresh as given (50
 * in OpenCV's K overload): the monocular counterpart of the pose the reference
 * takes from solvePnPRansac (src/frame_processor.cpp:176-196; the reference
 * itself never decomposes an essential matrix).  The four candidates of
 * cv::decomposeEssentialMat, every correspondence triangulated under each, the
 * first candidate in the order (R1,t) (R2,t) (R1,-t) (R2,-t) with the most
 * points in front of both cameras and nearer than distance_thresh wins.
 * p1/p2: n x 2 float32 pixels; mask (n bytes) is ANDed into the votes when
 * use_mask is non-zero and always receives the winner's 0/1 votes.  R: 3 x 3
 * row-major, t: unit 3-vector, x2 ~ R x1 + t; *n_good = votes of the winner. */
int vo_recover_pose(vo_ctx *ctx, const double *E, const float *p1, const float *p2, int n,
                    double fx, double fy, double cx, double cy, double distance_thresh,
                    int use_mask, uint8_t *mask, double *R, double *t, int *n_good);

/* ImageProcessor::processImage (reference src/image_processor.cpp:6-79,
 * include/visual_odometry/image_processor.hpp:8-38).  gray = BGR2GRAY of a
 * 3-channel frame or a copy of a 1-channel one.  With enhance != 0 also:
 * enhanced = cv::equalizeHist(gray) when hist_eq, else gray; denoised =
 * cv::GaussianBlur(enhanced, Size(blur_size, blur_size), sigma) when
 * blur_size > 1 (odd, <= 31; the 8-bit fixed-point path, BORDER_REFLECT_101),
 * else enhanced; masked = denoised with rows outside [h/4, 3h/4] zeroed.
 * All outputs are host planes with the row stride out_stride. */
int vo_image_process(vo_ctx *ctx, const uint8_t *src, int w, int h, int src_stride,
                     int channels, int enhance, int hist_eq, int blur_size,
                     double sigma, uint8_t *gray, uint8_t *enhanced,
                     uint8_t *denoised, uint8_t *masked, int out_stride);

/* cv::undistort(gray, K, dist) of an 8-bit single-channel image: the inverse
 * map of initUndistortRectifyMap (R = I, new camera matrix = K) quantised to
 * 1/32 px, bilinear cv::remap with 15-bit fixed-point weights and a constant 0
 * border.  dist = k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist 4, 5 or 8).  Not part of
 * the reference's ImageProcessor (src/image_processor.cpp:6-79 has no
 * undistortion; the ZED frames it receives are rectified,
 * src/zed_interface.cpp:112): provided for cameras that deliver raw frames. */
int vo_undistort(vo_ctx *ctx, const uint8_t *src, int w, int h, int src_stride,
                 double fx, double fy, double cx, double cy, const double *dist, int n_dist,
                 uint8_t *dst, int dst_stride);

/* Stereo front end: depth for the key points of the left image of a rectified
 * rig.  Not part of the reference, whose depth plane comes from the ZED SDK
 * (src/zed_interface.cpp:84-85); it produces the depth that
 * FrameProcessor::backprojectAndFilter samples at the rounded key point
 * (src/frame_processor.cpp:230-308), for cameras without such an SDK.
 * Candidates of left key point i are the right key points j with
 * |yL - yR| <= band and min_disp <= xL - xR <= max_disp; the best is the
 * smallest Hamming-256 distance (ties: lowest j), it must pass
 * d0 < ratio * d1 against the second candidate when there is one, and i must
 * be the best left candidate of j under the same rule (cross-check).
 * Per left key point: right_idx (-1 if unmatched), disparity = xL - xR (-1),
 * depth = fx * baseline / disparity in the unit of baseline (NaN).  plane, if
 * not null, is a dense w x h float32 image (host or device memory, row stride
 * w): NaN except at (round(xL), round(yL)) of the matched key points, where it
 * holds the depth (the nearest one when several share a pixel); it can be
 * handed to vo_batch_submit as the depth of that frame. */
int vo_stereo_depth(vo_ctx *ctx, const vo_kp *left_kps, const uint8_t *left_desc, int n_left,
                    const vo_kp *right_kps, const uint8_t *right_desc, int n_right, double fx,
                    double baseline, float band, float min_disp, float max_disp, float ratio,
                    int32_t *right_idx, float *disparity, float *depth, float *plane, int w, int h,
                    int *n_matched);

/* The same with an optional cross-check: a ratio-test survivor (i, j) is kept
 * only when i is also the nearest previous descriptor of j (mutual nearest
 * neighbours, ties to the lower index).  An extension: the reference creates
 * its matcher with crossCheck = false (src/feature_matcher.cpp:8) although
 * its yaml declares feature_detector.matching.cross_check
 * (config/vo_params.yaml). */
int vo_match_features_ex(vo_ctx *ctx, const vo_kp *prev_kps, const uint8_t *prev_desc,
                         int n_prev, const vo_kp *curr_kps, const uint8_t *curr_desc,
                         int n_curr, float ratio, int cross_check, vo_match *out,
                         int *n_out);

/* cv::solvePnPRansac(obj, img, K, dist = 0, rvec, tvec, false, iters,
 * reproj_threshold, confidence, inliers) followed by nothing else (the caller
 * applies cv::Rodrigues): EPnP on 5-point samples drawn with cv::RNG seed
 * 2^64-1, squared reprojection error <= threshold^2, adaptive iteration count,
 * then solvePnP(SOLVEPNP_ITERATIVE) on the inliers.
 * obj: n x 3 float32, img: n x 2 float32; rvec, tvec: 3 doubles each;
 * inliers: up to n ascending indices.  The refinement starts from a homography
 * when the inliers are planar (a wall, the floor) and from the DLT otherwise;
 * five non-planar inliers keep the pose of their minimal sample, as in OpenCV.
 * Exactly four points go to a P3P solve without RANSAC (the fourth point picks
 * among the solutions of the first three), exactly five to one EPnP solve.
 * *success is 0 when no hypothesis has more than 4 inliers or when n < 4.
 * reference src/frame_processor.cpp:176-196 */
int vo_solve_pnp_ransac(vo_ctx *ctx, const float *obj, const float *img, int n,
                        double fx, double fy, double cx, double cy, int iters,
                        double reproj_threshold, double confidence,
                        double *rvec, double *tvec, int32_t *inliers,
                        int *n_inliers, int *success);

/* ---- batched pipeline (throughput path) ------------------------------
 *
 * FrameProcessor::processFrame for n_streams independent camera streams,
 * frames_per_stream consecutive frames of each per call
 * (reference src/frame_processor.cpp:18-202).  Frame f of stream s is batch
 * entry s*frames_per_stream + f.  Within a stream every frame is matched
 * against its predecessor; the first frame of a call is matched against the
 * last frame of the previous call (device-resident state), unless the stream
 * was just configured or reset, in which case it is a "first frame" with no
 * matching, exactly like processFrame(first_frame = true). */

typedef struct {
    int width, height, channels;      /* channels: 1 gray, 3 BGR, 4 BGRA */
    int n_streams, frames_per_stream; /* product <= vo_cfg.max_batch */
    int max_features, fast_threshold; /* the reference runs 1000 / 20 */
    float ratio;                      /* Lowe ratio, reference 0.7f */
    double f_threshold, f_confidence; /* reference 3.0, 0.99 */
    int f_max_iters;                  /* OpenCV default 1000 */
    int enable_pose;                  /* depth back-projection + PnP */
    double fx, fy, cx, cy;            /* CameraParams */
    int pnp_iters;                    /* reference 100 */
    double pnp_threshold, pnp_confidence; /* reference 8.0, 0.99 */
    int cross_check;                  /* mutual-nearest filter after the ratio test (reference: 0) */
    int orb_levels;                   /* 0: FAST + ORB::compute as the reference runs it; 1..8: the full
                                         ORB detector (vo_detect_orb) with that many pyramid levels */
    float orb_scale_factor;           /* pyramid scale factor of the full detector (<= 0: 1.2) */
} vo_batch_cfg;

typedef struct {
    int32_t n_keypoints;  /* Features.keypoints.size() */
    int32_t n_ratio;      /* ratio-test survivors */
    int32_t f_found, f_iters, f_inliers;
    int32_t n_matches;    /* after F-RANSAC + motion filter (FeatureMatcher::match) */
    int32_t n_tracked;    /* after the multi-view (track age) filter */
    int32_t n_3d;         /* with valid depth */
    int32_t pnp_success, pnp_inliers, pnp_iters, pnp_points;
    double rvec[3], tvec[3]; /* T_prev_from_curr, translation in depth units (mm) */
} vo_frame_result;

int vo_batch_configure(vo_ctx *ctx, const vo_batch_cfg *cfg);
/* the next frame of every stream becomes a first frame */
int vo_batch_reset(vo_ctx *ctx);
/* Replace the camera intrinsics (and enable the pose stage) without touching
 * the streams' state; CameraParams is an argument of every processFrame call
 * (reference src/frame_processor.cpp:18-24). */
int vo_batch_set_camera(vo_ctx *ctx, double fx, double fy, double cx, double cy);
/* FrameProcessor::setPreviousFrame(frame, depth, features): make `features`
 * the previous frame of `stream`; its track ages restart at 0.
 * reference src/frame_processor.cpp:220-228 */
int vo_batch_set_previous(vo_ctx *ctx, int stream, const vo_kp *kps,
                          const uint8_t *desc, int n);
/* Enqueue one batch. frames: batch x height x row_stride bytes, consecutive
 * frames frame_stride bytes apart; depth (may be NULL): float32 mm, strides
 * in elements.  Device pointers are read in place; host frames are copied
 * to the device first.  A depth plane in host memory is not copied: the
 * back-projection fetches the few hundred values it needs over the link, from
 * the caller's buffer when that is pinned (vo_host_alloc, hipHostMalloc,
 * hipHostRegister), else from a pinned copy the call makes.  Buffers handed
 * over must stay as they are until vo_batch_wait has returned.  Returns without
 * waiting for the GPU. */
int vo_batch_submit(vo_ctx *ctx, const uint8_t *frames, int row_stride,
                    size_t frame_stride, const float *depth,
                    int depth_row_stride, size_t depth_frame_stride);
/* vo_batch_submit with the depth planes as 16-bit unsigned millimetres (0 = no measurement) instead of float32:
 * every depth value z is used as (float)z, everything else is the same call.  An extension for host-fed
 * deployments, where the upload is what limits the frame rate: gray 8-bit frames (channels = 1) with 16-bit depth
 * are 3 bytes per pixel against the 8 of BGRA + float32 depth.  The reference's camera interface delivers BGRA and
 * float32 millimetres (src/zed_interface.cpp:84-91); depth sensors themselves report 16-bit millimetres.
 * depth_row_stride / depth_frame_stride are in elements. */
int vo_batch_submit_depth16(vo_ctx *ctx, const uint8_t *frames, int row_stride, size_t frame_stride,
                            const uint16_t *depth_mm, int depth_row_stride, size_t depth_frame_stride);
/* Parameters of the stereo matcher (see vo_stereo_depth). */
typedef struct {
    double baseline;      /* distance between the cameras, in the unit the depth is wanted in (mm for the pose stage) */
    float band;           /* largest |yL - yR| of a candidate pair, pixels */
    float min_disparity;  /* > 0 */
    float max_disparity;
    float ratio;          /* Lowe ratio against the second candidate (>= 1 disables the test) */
} vo_stereo_cfg;
/* Stereo variant of vo_batch_submit for rigs without a depth sensor: `right`
 * holds the right image of every frame of the batch (same layout as `left`).
 * Both images go through the configured detector, vo_stereo_depth's matcher
 * runs for every frame on the device, and the sparse depth planes it produces
 * take the place of the `depth` argument of vo_batch_submit; everything after
 * that is the same pipeline (the reference reads its depth plane from the ZED
 * SDK, src/zed_interface.cpp:84-85, src/frame_processor.cpp:230-308).  The
 * context must have been created with max_batch >= 2 * batch. */
int vo_batch_submit_stereo(vo_ctx *ctx, const uint8_t *left, const uint8_t *right, int row_stride,
                           size_t frame_stride, const vo_stereo_cfg *stereo);
/* Stereo matches of one frame of the last completed stereo batch: per left
 * key point the right index (-1), disparity (-1) and depth (NaN); any pointer
 * may be NULL. */
int vo_batch_get_stereo(vo_ctx *ctx, int frame, int32_t *right_idx, float *disparity, float *depth,
                        int *n_matched);
/* Wait for the submitted batch; results: batch entries (host memory). */
int vo_batch_wait(vo_ctx *ctx, vo_frame_result *results);
/* Per-frame outputs of the last completed batch (host or device pointers;
 * any may be NULL).  kps/desc hold up to max_features entries. */
int vo_batch_get_features(vo_ctx *ctx, int frame, vo_kp *kps, uint8_t *desc,
                          int *n_out);
/* which: 0 = FeatureMatcher::match output, 1 = after the track-age filter,
 * 2 = with valid depth (the PnP input).  pts3 (n x 3 float32) only for 2. */
int vo_batch_get_matches(vo_ctx *ctx, int frame, int which, vo_match *out,
                         float *pts3, int *n_out);
int vo_batch_get_inliers(vo_ctx *ctx, int frame, int32_t *inliers, int *n_out);

/* Packed output (latency path).  The reference returns everything by value
 * from one processFrame call (src/frame_processor.cpp:18-202); fetching the
 * same through the getters above costs one blocking copy per array.  With
 * packed output enabled, vo_batch_submit appends one kernel that gathers, for
 * every frame of the batch, the result record, key points, descriptors, the
 * three match lists, the 3-D points and the PnP inliers into one block, and
 * ONE device-to-host copy of all blocks into pinned memory; after
 * vo_batch_wait(), vo_batch_get_packed() hands out pointers into that memory
 * (valid until the next submit or configure on this context; no copy, no GPU
 * call).  detect_ms / match_ms are DEVICE times (differences of the GPU's
 * wall clock, stamped by the kernels themselves) of the detector stages and
 * of kNN + ratio + F-RANSAC for the whole batch: they take the place of the
 * host timers the reference puts around detectFeatures / matchFeatures
 * (:30-40) but do not contain the upload of the frame, the launch overhead
 * or the download of the results; a caller that wants the host view times
 * vo_batch_submit .. vo_batch_wait itself.
 * Only for small batches: batch x (128 + 156 x feature capacity) bytes may not
 * exceed 32 MiB (VO_ERR_ARG otherwise). */
typedef struct {
    vo_frame_result result;
    float detect_ms, match_ms;
    int32_t n_keypoints;
    const vo_kp *kps;
    const uint8_t *desc;      /* n_keypoints x 32 */
    int32_t n_matches, n_tracked, n_valid;
    const vo_match *matches;  /* FeatureMatcher::match output */
    const vo_match *tracked;  /* after the track-age filter */
    const vo_match *valid;    /* with valid depth (the PnP input) */
    const float *pts3;        /* n_valid x 3 */
    int32_t n_inliers;
    const int32_t *inliers;
} vo_frame_view;
int vo_batch_set_packed_output(vo_ctx *ctx, int enable);
int vo_batch_get_packed(vo_ctx *ctx, int frame, vo_frame_view *view);

/* Replay the kernel chain of a batch as captured HIP graphs instead of
 * launching its 10 to 30 kernels one by one (no counterpart in the reference).
 * The chain is captured in two parts -- everything that needs the frames only,
 * then everything from the depth plane on, so that a depth plane in host
 * memory can travel beside the first part -- on the first submit after
 * enabling, and again whenever the configuration, the camera or the device
 * input pointers change; host inputs go through the context's staging area
 * and never force a re-capture.  It pays where the host cannot keep up with
 * launching (many short kernels per batch on a busy CPU); for a single HD720
 * frame on an idle core plain launches are faster, because a graph launch takes
 * about 0.1 ms to reach the GPU.  A graph replay records no per-stage events:
 * vo_batch_get_timing() returns VO_ERR_STATE after a batch that ran as a
 * graph. */
int vo_batch_set_graph(vo_ctx *ctx, int enable);

/* Pinned host memory for frames and depth planes: vo_batch_submit copies from
 * such memory by DMA without the staging pass that ordinary (pageable) memory
 * needs.  A camera wrapper that owns its buffers (the reference clones every
 * frame into a queue, src/vo_node.cpp:581-588) can clone into these instead. */
void *vo_host_alloc(size_t bytes);
void vo_host_free(void *p);

/* Per-kernel timing of vo_batch_submit (measurement aid; the reference only
 * clocks detect and match as a whole, src/frame_processor.cpp:30-40).  When
 * enabled, HIP events are recorded on the context's stream around every
 * kernel of the batch; vo_batch_get_timing() returns, after vo_batch_wait(),
 * the milliseconds of each of the vo_batch_stage_count() stages. */
int vo_batch_set_timing(vo_ctx *ctx, int enable);
int vo_batch_stage_count(void);
const char *vo_batch_stage_name(int stage);
int vo_batch_get_timing(vo_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
