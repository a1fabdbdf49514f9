// vo::FrameProcessor with the reference's interface
// (reference include/visual_odometry/frame_processor.hpp:12-68,
// src/frame_processor.cpp:9-308).  One FrameProcessor = one camera stream =
// one vo_ctx running the batched pipeline with a batch of one frame; the
// previous frame's features and track ages stay on the device.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <memory>
#include <utility>
#include <vector>

#include "visual_odometry/feature_detector.hpp"
#include "visual_odometry/feature_matcher.hpp"
#include "visual_odometry/vo_device.hpp"

namespace vo {

/** IMU-predicted relative pose (optical frame, T_prev_from_curr). */
struct ImuPredictedPose {
    cv::Mat R;                 // 3x3 rotation (CV_64F)
    cv::Mat t;                 // 3x1 translation (mm, CV_64F)
    double angular_rate{0.0};  // rad/s
    bool valid{false};
};

class FrameProcessor {
public:
    FrameProcessor(std::shared_ptr<FeatureDetector> detector, std::shared_ptr<FeatureMatcher> matcher)
        : feature_detector_(std::move(detector)), feature_matcher_(std::move(matcher))
    {
        if (!feature_detector_ || !feature_matcher_) throw std::runtime_error("Null detector or matcher");
    }

    struct ProcessingResult {
        Features features;
        FeatureMatches matches;
        double feature_detection_time{0.0};
        double feature_matching_time{0.0};
        double visualization_time{0.0};
        bool is_keyframe{false};
        bool pnp_success{false};
        int pnp_inliers{0};
        int pnp_total_matches{0};
        double inlier_ratio{0.0};
        cv::Mat R;  // 3x3
        cv::Mat t;  // 3x1
    };

    // Like the reference this does not catch: the caller (vo_node.cpp:866-868)
    // handles std::exception.
    ProcessingResult processFrame(const cv::Mat &rgb, const cv::Mat &depth, const CameraParams &camera_params,
                                  bool first_frame = false, bool enable_pose_estimation = true,
                                  const ImuPredictedPose &imu_pred = {})
    {
        ProcessingResult result;
        // cv::cvtColor(BGR2GRAY) accepts 3 or 4 channels only (reference src/frame_processor.cpp:204-208)
        if (rgb.empty() || (rgb.channels() != 3 && rgb.channels() != 4) || rgb.depth() != CV_8U)
            throw std::runtime_error("processFrame: expected an 8-bit BGR or BGRA frame");
        configure(rgb.cols, rgb.rows, rgb.channels());
        vo_ctx *ctx = dev_.get();
        if (first_frame) dev_.check(vo_batch_reset(ctx), "vo_batch_reset");
        const bool cam_ok = camera_params.fx > 0 && camera_params.fy > 0;
        if (cam_ok)
            dev_.check(vo_batch_set_camera(ctx, camera_params.fx, camera_params.fy, camera_params.cx,
                                           camera_params.cy),
                       "vo_batch_set_camera");
        const bool want_pose = enable_pose_estimation && !depth.empty() && cam_ok;
        if (want_pose && (depth.depth() != CV_32F || depth.cols != rgb.cols || depth.rows != rgb.rows))
            throw std::runtime_error("processFrame: depth must be float32 of the frame's size");

        dev_.check(vo_batch_submit(ctx, rgb.ptr<uint8_t>(0), (int)rgb.step, rgb.step * (size_t)rgb.rows,
                                   want_pose ? depth.ptr<float>(0) : nullptr,
                                   want_pose ? (int)(depth.step / sizeof(float)) : 0,
                                   want_pose ? depth.step / sizeof(float) * (size_t)depth.rows : 0),
                   "vo_batch_submit");
        dev_.check(vo_batch_wait(ctx, nullptr), "vo_batch_wait");
        // everything the frame produced arrived with that one wait (packed output in pinned memory)
        vo_frame_view view{};
        dev_.check(vo_batch_get_packed(ctx, 0, &view), "vo_batch_get_packed");
        const vo_frame_result &r = view.result;
        // detect = gray + FAST/ORB + descriptors, match = kNN + ratio + F-RANSAC + motion filter
        // (reference src/frame_processor.cpp:30-40), GPU time
        result.feature_detection_time = view.detect_ms;
        result.feature_matching_time = view.match_ms;
        detail::to_features(view.kps, view.desc, view.n_keypoints, result.features);

        // reference src/frame_processor.cpp:97,272-307: the match list is replaced
        // by its depth-valid subset only when that subset is not empty
        const int which = r.n_3d > 0 ? 2 : 1;
        const int nm = which == 2 ? view.n_valid : view.n_tracked;
        detail::to_matches(which == 2 ? view.valid : view.tracked, nm, result.matches);
        if (which == 2) {
            result.matches.prev_points_3d.resize((size_t)nm);
            for (int i = 0; i < nm; i++)
                result.matches.prev_points_3d[(size_t)i] =
                    cv::Point3f(view.pts3[3 * (size_t)i], view.pts3[3 * (size_t)i + 1], view.pts3[3 * (size_t)i + 2]);
        }

        if (which == 2) {
            double rvec[3] = {r.rvec[0], r.rvec[1], r.rvec[2]}, tvec[3] = {r.tvec[0], r.tvec[1], r.tvec[2]};
            int success = r.pnp_success, n_inl = r.pnp_inliers;
            bool have_pnp = nm >= 4;
            if (imu_pred.valid && !imu_pred.R.empty() && !imu_pred.t.empty() && nm >= 100) {
                if (imuGuidedFilter(result.matches, camera_params, imu_pred)) {
                    // the point set changed: solve again on what is left
                    const int k = (int)result.matches.prev_points_3d.size();
                    have_pnp = k >= 4;
                    if (have_pnp) {
                        obj_.resize(3 * (size_t)k); img_.resize(2 * (size_t)k); inl_.resize((size_t)k);
                        for (int i = 0; i < k; i++) {
                            const cv::Point3f &p = result.matches.prev_points_3d[(size_t)i];
                            obj_[3 * (size_t)i] = p.x; obj_[3 * (size_t)i + 1] = p.y; obj_[3 * (size_t)i + 2] = p.z;
                            img_[2 * (size_t)i] = result.matches.prev_points[(size_t)i].x;
                            img_[2 * (size_t)i + 1] = result.matches.prev_points[(size_t)i].y;
                        }
                        dev_.check(vo_solve_pnp_ransac(ctx, obj_.data(), img_.data(), k, camera_params.fx,
                                                       camera_params.fy, camera_params.cx, camera_params.cy, 100,
                                                       8.0, 0.99, rvec, tvec, inl_.data(), &n_inl, &success),
                                   "vo_solve_pnp_ransac");
                    }
                }
            }
            if (have_pnp) {
                result.pnp_success = success != 0;
                result.pnp_inliers = n_inl;
                result.pnp_total_matches = (int)result.matches.prev_points_3d.size();
                result.inlier_ratio = result.pnp_total_matches > 0
                                          ? static_cast<double>(result.pnp_inliers) / result.pnp_total_matches
                                          : 0.0;
                if (result.pnp_success) {
                    result.R = detail::rodrigues_to_matrix(rvec);
                    result.t = cv::Mat(3, 1, CV_64FC1);
                    for (int i = 0; i < 3; i++) result.t.at<double>(i, 0) = tvec[i];
                }
            }
        }
        return result;
    }

    // reference src/frame_processor.cpp:220-228: the features become the
    // stream's previous frame (the gray image and depth are not needed again).
    void setPreviousFrame(const cv::Mat &frame, const cv::Mat & /*depth*/, const Features &features)
    {
        if (frame.empty()) throw std::runtime_error("setPreviousFrame: empty frame");
        configure(frame.cols, frame.rows, frame.channels() == 4 ? 4 : 3);
        const int n = (int)features.keypoints.size();
        std::vector<vo_kp> k((size_t)n);
        std::vector<uint8_t> d((size_t)n * 32);
        for (int i = 0; i < n; i++) {
            k[(size_t)i] = detail::from_cv(features.keypoints[(size_t)i]);
            std::memcpy(d.data() + (size_t)i * 32, features.descriptors.ptr<uint8_t>(i), 32);
        }
        dev_.check(vo_batch_set_previous(dev_.get(), 0, k.data(), d.data(), n), "vo_batch_set_previous");
    }

private:
    static constexpr int kMaxFeatures = 1000;  // detectFeatures(gray) runs with the header defaults
    static constexpr int kFastThreshold = 20;  // (reference src/frame_processor.cpp:210-212)

    void configure(int w, int h, int channels)
    {
        // Default: what the reference's processFrame really runs, FAST + ORB::compute with 1000 / 20 whatever the
        // setters said.  With the detector switched to DetectorMode::Orb its max_features / scale_factor /
        // n_levels (the values vo_node forwards, reference src/vo_node.cpp:333-336) configure the full detector.
        const bool orb = feature_detector_->detectorMode() == DetectorMode::Orb;
        const int nfeat = orb ? feature_detector_->maxFeatures() : kMaxFeatures;
        const int levels = orb ? feature_detector_->nLevels() : 0;
        const float scale = orb ? feature_detector_->scaleFactor() : 0.f;
        const int cap = orb ? nfeat + 64 : nfeat;  // boundary ties of the per-level ranking
        const bool fresh = dev_.ensure(w, h, cap);
        if (!fresh && w == w_ && h == h_ && channels == ch_ && nfeat == nfeat_ && levels == levels_ && scale == scale_)
            return;
        vo_batch_cfg b{};
        b.width = w; b.height = h; b.channels = channels;
        b.n_streams = 1; b.frames_per_stream = 1;
        b.max_features = nfeat; b.fast_threshold = kFastThreshold;
        b.orb_levels = levels; b.orb_scale_factor = scale;
        b.ratio = 0.7f;
        b.f_threshold = 3.0; b.f_confidence = 0.99; b.f_max_iters = 1000;
        b.enable_pose = 0;
        b.pnp_iters = 100; b.pnp_threshold = 8.0; b.pnp_confidence = 0.99;
        dev_.check(vo_batch_configure(dev_.get(), &b), "vo_batch_configure");
        // latency path: one packed device-to-host copy per frame.  The kernels are launched one by one: the first
        // one then starts as soon as the frame has arrived, whereas a graph launch takes about 0.1 ms to reach the
        // GPU, which the frame's upload no longer hides since the depth plane travels beside the first kernels
        // (VO_GRAPH=1 replays the chain as a HIP graph)
        dev_.check(vo_batch_set_packed_output(dev_.get(), 1), "vo_batch_set_packed_output");
        const char *g = std::getenv("VO_GRAPH");
        dev_.check(vo_batch_set_graph(dev_.get(), g && g[0] == '1'), "vo_batch_set_graph");
        w_ = w; h_ = h; ch_ = channels; nfeat_ = nfeat; levels_ = levels; scale_ = scale;
    }

    // reference src/frame_processor.cpp:99-173: drop features whose reprojection
    // under the IMU-predicted motion is off by more than 10 + 5*angular_rate px,
    // keeping at least max(100, 60 %).  Host code, like the reference: it
    // depends on the IMU fusion state that lives on the CPU.  Returns true when
    // something was removed.
    static bool imuGuidedFilter(FeatureMatches &mt, const CameraParams &cam, const ImuPredictedPose &imu)
    {
        const size_t n = mt.prev_points_3d.size();
        double R[9], t[3];
        for (int i = 0; i < 9; i++) R[i] = imu.R.at<double>(i / 3, i % 3);
        // the reference converts R to a rotation vector and back inside projectPoints
        double rv[3];
        detail::matrix_to_rodrigues(imu.R, rv);
        const cv::Mat Rm = detail::rodrigues_to_matrix(rv);
        for (int i = 0; i < 9; i++) R[i] = Rm.at<double>(i / 3, i % 3);
        for (int i = 0; i < 3; i++) t[i] = imu.t.rows == 3 ? imu.t.at<double>(i, 0) : imu.t.at<double>(0, i);
        const double thr = 10.0 + imu.angular_rate * 5.0;
        std::vector<std::pair<double, size_t>> errors;
        errors.reserve(n);
        for (size_t k = 0; k < n; k++) {
            const cv::Point3f &p = mt.prev_points_3d[k];
            const double X = R[0] * p.x + R[1] * p.y + R[2] * p.z + t[0];
            const double Y = R[3] * p.x + R[4] * p.y + R[5] * p.z + t[1];
            double Z = R[6] * p.x + R[7] * p.y + R[8] * p.z + t[2];
            Z = Z ? 1.0 / Z : 1.0;
            const float u = (float)(X * Z * cam.fx + cam.cx), v = (float)(Y * Z * cam.fy + cam.cy);
            const float dx = u - mt.prev_points[k].x, dy = v - mt.prev_points[k].y;
            errors.emplace_back(std::sqrt((double)dx * dx + (double)dy * dy), k);
        }
        size_t min_retain = std::max(static_cast<size_t>(100), static_cast<size_t>(n * 0.6));
        min_retain = std::min(min_retain, n);
        std::vector<size_t> keep;
        for (const auto &e : errors)
            if (e.first <= thr) keep.push_back(e.second);
        if (keep.size() < min_retain) {
            std::sort(errors.begin(), errors.end());
            keep.clear();
            for (size_t i = 0; i < min_retain && i < errors.size(); ++i) keep.push_back(errors[i].second);
            std::sort(keep.begin(), keep.end());
        }
        if (keep.size() == n) return false;
        FeatureMatches f;
        f.reserve(keep.size());
        for (size_t idx : keep) {
            f.prev_points_3d.push_back(mt.prev_points_3d[idx]);
            f.prev_points.push_back(mt.prev_points[idx]);
            f.curr_points.push_back(mt.curr_points[idx]);
            f.matches.push_back(mt.matches[idx]);
        }
        mt = std::move(f);
        return true;
    }

    std::shared_ptr<FeatureDetector> feature_detector_;
    std::shared_ptr<FeatureMatcher> feature_matcher_;
    detail::DeviceContext dev_;
    int w_{0}, h_{0}, ch_{0}, nfeat_{0}, levels_{-1};
    float scale_{0.f};
    std::vector<float> obj_, img_;
    std::vector<int32_t> inl_;
};

}  // namespace vo
