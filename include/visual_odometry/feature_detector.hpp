// vo::FeatureDetector with the reference's interface
// (reference include/visual_odometry/feature_detector.hpp:8-46,
// src/feature_detector.cpp:9-123), computing on the GPU through vo_hip.h.
#pragma once

#include <cstdlib>
#include <string>
#include <vector>

#include "visual_odometry/vo_device.hpp"

namespace vo {

// Which detector runs behind detectFeatures / FrameProcessor::processFrame.
//   FastOrbCompute  what the reference executes: cv::FAST, the strongest N, cv::ORB::compute on one level
//                   (reference src/feature_detector.cpp:59-73); setMaxFeatures / setScaleFactor / setNLevels are
//                   stored and change nothing, and processFrame detects with the header defaults 1000 / 20
//                   (src/frame_processor.cpp:210-212).  The default.
//   Orb             the detector the reference configures but never calls: cv::ORB::create(max_features,
//                   scale_factor, n_levels)->detectAndCompute (src/feature_detector.cpp:18-22, setters :97-123,
//                   driven by src/vo_node.cpp:333-336).  Opt-in: setDetectorMode(DetectorMode::Orb) or the
//                   environment variable VO_DETECTOR=orb; the three setters then take effect.
enum class DetectorMode { FastOrbCompute, Orb };

class FeatureDetector {
public:
    // The reference constructor creates the OpenCV detector objects and throws
    // std::runtime_error on failure; here the device context is created on the
    // first frame (its size is not known earlier) and vo_create failures
    // surface from detectFeatures' error path.
    FeatureDetector()
    {
        if (const char *e = std::getenv("VO_DETECTOR"))
            if (std::string(e) == "orb") mode_ = DetectorMode::Orb;
    }

    void setDetectorMode(DetectorMode m) { mode_ = m; }
    DetectorMode detectorMode() const { return mode_; }
    int maxFeatures() const { return max_features_; }
    float scaleFactor() const { return scale_factor_; }
    int nLevels() const { return n_levels_; }

    // FAST (threshold, NMS) -> keep the max_features strongest -> ORB::compute.
    // Never throws: errors are logged and whatever was produced is returned
    // (reference src/feature_detector.cpp:82-85).
    Features detectFeatures(const cv::Mat &frame, int max_features = 1000, int fast_threshold = 20)
    {
        Features result;
        try {
            if (frame.empty()) throw std::runtime_error("empty frame");
            const int w = frame.cols, h = frame.rows;
            const bool orb = mode_ == DetectorMode::Orb;
            // the full detector keeps boundary ties of its per-level ranking: room for a few more than asked
            const int cap = orb ? max_features_ + 64 : max_features;
            dev_.ensure(w, h, cap);
            const uint8_t *gray = frame.ptr<uint8_t>(0);
            int stride = (int)frame.step;
            if (frame.channels() == 3) {
                // reference src/feature_detector.cpp:44-46
                gray_.resize((size_t)w * h);
                dev_.check(vo_gray(dev_.get(), frame.ptr<uint8_t>(0), w, h, (int)frame.step, 3, gray_.data(), w),
                           "vo_gray");
                gray = gray_.data();
                stride = w;
            } else if (frame.channels() != 1) {
                // the reference clones any non-3-channel frame and hands it to cv::FAST, which rejects it
                throw std::runtime_error("FAST needs a 1-channel image");
            }
            kps_.resize((size_t)cap);
            desc_.resize((size_t)cap * 32);
            int n = 0;
            if (orb)
                dev_.check(vo_detect_orb(dev_.get(), gray, w, h, stride, max_features_, scale_factor_, n_levels_,
                                         fast_threshold, kps_.data(), desc_.data(), cap, &n),
                           "vo_detect_orb");
            else
                dev_.check(vo_detect(dev_.get(), gray, w, h, stride, max_features, fast_threshold, kps_.data(),
                                     desc_.data(), &n),
                           "vo_detect");
            detail::to_features(kps_, desc_, n, result);
        } catch (const std::exception &e) {
            VO_LOG_ERROR("feature_detector", "Error in detectFeatures: %s", e.what());
        }
        return result;
    }

    void setVisualizationType(const std::string &type) { visualization_flags_ = type == "rich" ? 4 : 0; }
    void updateDetector() {}

    // The reference forwards these to cv::ORB::create, whose compute() ignores
    // nfeatures / scaleFactor / nlevels for externally detected keypoints, so
    // in the default mode they never change the result (reference
    // src/vo_node.cpp:333-336); in DetectorMode::Orb they configure the detector.
    void setMaxFeatures(int max_features) { max_features_ = max_features; updateDetector(); }
    void setScaleFactor(float scale_factor) { scale_factor_ = scale_factor; updateDetector(); }
    void setNLevels(int n_levels) { n_levels_ = n_levels; updateDetector(); }

private:
    detail::DeviceContext dev_;
    std::vector<uint8_t> gray_, desc_;
    std::vector<vo_kp> kps_;
    int max_features_{1000};
    float scale_factor_{1.2f};
    int n_levels_{8};
    int visualization_flags_{0};
    DetectorMode mode_{DetectorMode::FastOrbCompute};
};

}  // namespace vo
