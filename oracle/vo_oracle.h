/* CPU oracle for the visual-odometry front-end hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the HIP path is
 * compared against; nothing in the shipped library links or calls it.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 *
 * PARITY UNPINNED: the reference (hoa0202/visual_odometry) delegates all of
 * this arithmetic to OpenCV >= 4.8 (reference CMakeLists.txt:21-22), which is
 * neither vendored in the reference nor installed here, and the reference
 * ships no tests or golden vectors.  Each function below restates the
 * published OpenCV 4.8 algorithm behind one reference call site and cites
 * that call site; details recalled rather than checked against OpenCV sources
 * are marked [OCV-recall].
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint as a POD (pt.x, pt.y, size, angle, response, octave). */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
} ora_kp;

/* two nearest neighbours of one query descriptor */
typedef struct {
    int32_t idx0, dist0;
    int32_t idx1, dist1;
} ora_knn2;

/* one surviving match: cv::DMatch(queryIdx, trainIdx, distance) + both points */
typedef struct {
    int32_t query, train;
    float distance;
    float px, py; /* prev (query) point */
    float cx, cy; /* curr (train) point */
} ora_match;

/* ---- image stages --------------------------------------------------- */

/* cv::cvtColor(COLOR_BGR2GRAY / BGRA2GRAY) on 8-bit data.
 * reference src/frame_processor.cpp:204-208, :223, src/feature_detector.cpp:46 */
void ora_gray(const uint8_t *src, int w, int h, int stride, int cn,
              uint8_t *dst, int dstride);

/* FAST-9/16 corner score map (0 where not a corner) -- the per-pixel half of
 * cv::FAST(img, kps, thr, true).  reference src/feature_detector.cpp:59 */
void ora_fast_score_map(const uint8_t *img, int w, int h, int stride, int thr,
                        uint8_t *score, int sstride);

/* cv::FAST(img, kps, thr, nonmaxSuppression=true): raster-ordered keypoints.
 * Returns the number found (at most cap are written). */
int ora_fast(const uint8_t *img, int w, int h, int stride, int thr,
             ora_kp *out, int cap);

/* reference src/feature_detector.cpp:62-68: keep the max_features strongest.
 * Canonical total order (response desc, then raster index asc); when n <=
 * max_features the raster order is left untouched, as in the reference. */
int ora_select_top(ora_kp *kps, int n, int w, int max_features);

/* the same truncation done literally (vo_oracle_stdsort.cpp, built into
 * libvo_oracle_stdsort.so with g++): std::sort with the reference's comparator
 * on the raster-ordered corners, then resize.  Used to measure how far the
 * canonical order above is from what libstdc++ leaves. */
int ora_select_top_stdsort(ora_kp *kps, int n, int max_features);

/* KeyPointsFilter::runByImageBorder(kps, size, 31) as run inside
 * ORB::compute (reference src/feature_detector.cpp:73). Order preserved. */
int ora_border_filter(ora_kp *kps, int n, int w, int h, int border);

/* 7x7 sigma=2 Gaussian as ORB applies it to an 8-bit pyramid level:
 * separable integer kernel [18,34,49,55,49,34,18]/256 per axis,
 * (sum + 2^15) >> 16, saturated, BORDER_REFLECT_101. */
void ora_blur7(const uint8_t *src, int w, int h, int stride, uint8_t *dst,
               int dstride);

/* rBRIEF-256 descriptors on the smoothed image.  angle is in degrees and is
 * used exactly as ORB does ((float)cos/sin, cvRound of the rotated pattern);
 * FAST keypoints carry angle = -1.  desc: n x 32 bytes. */
void ora_brief(const uint8_t *blur, int w, int h, int stride,
               const ora_kp *kps, int n, uint8_t *desc);

/* FeatureDetector::detectFeatures on a 1-channel frame
 * (reference src/feature_detector.cpp:37-88). Returns keypoint count. */
int ora_detect(const uint8_t *gray, int w, int h, int stride, int max_features,
               int fast_threshold, ora_kp *kps, uint8_t *desc);

/* ---- full ORB detector ("mode B": cv::ORB::detectAndCompute as configured
 * at reference src/feature_detector.cpp:18-22; never executed there) -------- */
int ora_orb_level_sizes(int w, int h, float scale_factor, int nlevels, int *lw, int *lh, float *lscale);
void ora_orb_quotas(int nfeatures, float scale_factor, int nlevels, int *quota);
void ora_resize_linear_exact(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                             int dstride);
float ora_fast_atan2(float y, float x);
float ora_harris_response(const uint8_t *img, int stride, int x0, int y0);
int ora_orb_umax(int *umax);
float ora_ic_angle(const uint8_t *img, int stride, int x0, int y0);
int ora_orb_detect(const uint8_t *gray, int w, int h, int stride, int nfeatures, float scale_factor,
                   int nlevels, int fast_threshold, ora_kp *kps, uint8_t *desc, int cap, uint8_t *pyr_out,
                   uint8_t *blur_out);

/* ---- ImageProcessor (reference src/image_processor.cpp:41-79) ---------- */

/* cv::equalizeHist */
void ora_equalize_hist(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride);
/* 8.8 fixed-point taps of cv::GaussianBlur on 8-bit data; n odd, 1..31; returns 0 on a bad n */
int ora_gaussian_taps_q8(int n, double sigma, int *taps);
/* cv::GaussianBlur(Size(n,n), sigma), whole image, BORDER_REFLECT_101 */
int ora_gaussian_blur(const uint8_t *src, int w, int h, int stride, int n, double sigma, uint8_t *dst,
                      int dstride);
/* zero outside rows [h/4, 3h/4] */
void ora_roi_mask(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride);
/* cv::undistort of an 8-bit gray image; dist = k1 k2 p1 p2 [k3 [k4 k5 k6]] */
int ora_undistort(const uint8_t *src, int w, int h, int stride, double fx, double fy, double cx, double cy,
                  const double *dist, int n_dist, uint8_t *dst, int dstride);

/* ---- matching ------------------------------------------------------- */

/* BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2)
 * (reference src/feature_matcher.cpp:27-31); ties keep the lower train index. */
void ora_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt,
                      ora_knn2 *out);

/* Lowe ratio test in float32 (reference src/feature_matcher.cpp:34-49). */
int ora_ratio_test(const ora_knn2 *knn, int nq, int nt, float ratio,
                   const ora_kp *qk, const ora_kp *tk, ora_match *out);

/* mutual nearest-neighbour check (cross-check extension, not in reference) */
int ora_cross_check(const ora_knn2 *fwd, int nq, const ora_knn2 *bwd, int nt,
                    uint8_t *keep);

/* ---- RANSAC --------------------------------------------------------- */

/* cv::RNG (multiply-with-carry) */
typedef struct { uint64_t state; } ora_rng;
uint32_t ora_rng_next(ora_rng *r);
int ora_rng_uniform(ora_rng *r, int a, int b);

/* cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, conf, maxIters, mask)
 * (reference src/feature_matcher.cpp:52-58).  p1/p2: n x 2 float32.
 * Returns 1 and fills F (row-major, F[8]==1 when defined) and mask, or 0.
 * iters_run (optional) receives the number of RANSAC iterations executed. */
int ora_find_fundamental_ransac(const float *p1, const float *p2, int n,
                                double thr, double conf, int max_iters,
                                uint8_t *mask, double *F, int *iters_run);

/* 7-point solver on exactly 7 correspondences; returns 0..3 models (9 doubles each) */
int ora_f7(const float *m1, const float *m2, double *Fout);

/* symmetric epipolar error used by the F estimator */
void ora_f_error(const float *p1, const float *p2, int n, const double *F,
                 float *err);

/* ---- essential matrix (monocular pose; not estimated by the reference) ---- */

/* five-point solver on normalised points (5 x 2 doubles each): up to 10 E (9 doubles each, |E|_F = 1) */
int ora_e5(const double *q1, const double *q2, double *Eout);
/* Sampson error of n normalised correspondences */
void ora_e_error(const double *q1, const double *q2, int n, const double *E, float *err);
/* batched hypotheses from an index table (H x 5): solutions and inlier counts */
void ora_e_hypotheses(const double *q1, const double *q2, int n, const int32_t *idx, int H, double thr, double *E,
                      int32_t *n_models, int32_t *counts);
/* cv::findEssentialMat(p1, p2, K, RANSAC, prob, threshold, maxIters, mask) on pixel points */
int ora_find_essential_ransac(const float *p1, const float *p2, int n, double fx, double fy, double cx, double cy,
                              double prob, double threshold, int max_iters, uint8_t *mask, double *E,
                              int *iters_run);
/* cv::decomposeEssentialMat and cv::recoverPose(E, p1, p2, K, R, t, mask) */
void ora_decompose_essential(const double *E, double *R1, double *R2, double *t);
int ora_recover_pose(const double *E, const float *p1, const float *p2, int n, double fx, double fy, double cx,
                     double cy, double dist, int use_mask, uint8_t *mask, double *R, double *t);

/* stereo front end (no counterpart in the reference, whose depth comes from
 * the ZED SDK): band-limited left/right matching with cross-check,
 * disparity -> depth, sparse depth plane */
int ora_stereo_match(const ora_kp *lk, const uint8_t *ld, int nl, const ora_kp *rk, const uint8_t *rd, int nr,
                     float band, float min_disp, float max_disp, float ratio, double fx, double baseline,
                     int32_t *right_idx, float *disparity, float *depth);
void ora_stereo_depth_plane(const ora_kp *lk, const float *depth, int nl, int w, int h, float *plane, int stride);

/* FeatureMatcher::match after kNN: ratio -> F-RANSAC mask -> 100 px motion
 * filter (reference src/feature_matcher.cpp:34-103).  Returns final count. */
int ora_match_filter(const ora_knn2 *knn, int nq, int nt, const ora_kp *qk,
                     const ora_kp *tk, float ratio, ora_match *out,
                     uint8_t *ransac_mask, int *n_ratio);

int ora_match_filter_x(const ora_knn2 *knn, const ora_knn2 *knn_rev, int nq, int nt,
                       const ora_kp *qk, const ora_kp *tk, float ratio, ora_match *out,
                       uint8_t *ransac_mask, int *n_ratio);

/* backprojectAndFilter(use_curr=true) (reference src/frame_processor.cpp:230-308).
 * depth is float32 mm, row stride in elements.  Compacts matches in place,
 * writes pts3 (n x 3 float).  Returns number kept; when it returns 0 the
 * match list is left untouched, as in the reference. */
int ora_backproject(ora_match *m, int n, const float *depth, int w, int h,
                    int dstride, double fx, double fy, double cx, double cy,
                    float *pts3);

/* track-age bookkeeping and multi-view filter
 * (reference src/frame_processor.cpp:44-92).  prev_age has n_prev entries,
 * curr_age (n_curr entries) is overwritten.  Returns the new match count. */
int ora_track_filter(ora_match *m, int n, const int32_t *prev_age, int n_prev,
                     int32_t *curr_age, int n_curr);

/* cv::solvePnPRansac(obj, img, K, dist=0, rvec, tvec, false, iters, thr,
 * conf, inliers) (reference src/frame_processor.cpp:176-196).
 * obj: n x 3 float32, img: n x 2 float32, K: fx, fy, cx, cy.
 * Returns 1 on success; inliers receives ascending indices. */
int ora_solve_pnp_ransac(const float *obj, const float *img, int n, double fx,
                         double fy, double cx, double cy, int iters,
                         double thr, double conf, double *rvec, double *tvec,
                         int32_t *inliers, int *n_inliers);

/* EPnP minimal/general solver: n >= 4 points -> R (row-major 3x3), t */
int ora_epnp(const double *obj, const double *img_norm, int n, double *R,
             double *t);

/* solvePnP(SOLVEPNP_P3P) on four correspondences (obj 4 x 3, img 4 x 2 pixels): R row-major, t */
int ora_p3p(const double *obj, const double *img, double fx, double fy, double cx, double cy, double *R, double *t);

/* Levenberg-Marquardt pose refinement as solvePnP(SOLVEPNP_ITERATIVE) */
int ora_pnp_iterative(const double *obj, const double *img, int n, double fx,
                      double fy, double cx, double cy, double *rvec,
                      double *tvec);

void ora_rodrigues_vec2mat(const double *rvec, double *R);
void ora_rodrigues_mat2vec(const double *R, double *rvec);

#ifdef __cplusplus
}
#endif
#endif
