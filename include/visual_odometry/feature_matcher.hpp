// vo::FeatureMatcher with the reference's interface
// (reference include/visual_odometry/feature_matcher.hpp:8-21,
// src/feature_matcher.cpp:7-104), computing on the GPU through vo_hip.h.
#pragma once

#include <vector>

#include "visual_odometry/vo_device.hpp"

namespace vo {

class FeatureMatcher {
public:
    FeatureMatcher() : ratio_threshold_(0.7f) {}

    // kNN-2 -> Lowe ratio -> F-RANSAC mask -> motion < 100 px.  The gray
    // frames are unused, as in the reference.  Returns an empty result on any
    // error (reference src/feature_matcher.cpp:98-103).
    FeatureMatches match(const Features &prev_features, const Features &curr_features,
                         const cv::Mat & /*prev_frame_gray*/, const cv::Mat & /*curr_frame_gray*/)
    {
        FeatureMatches result;
        if (prev_features.keypoints.empty() || curr_features.keypoints.empty()) return result;
        try {
            const int np = (int)prev_features.keypoints.size(), nc = (int)curr_features.keypoints.size();
            dev_.ensure(64, 64, np > nc ? np : nc);
            pk_.resize((size_t)np); ck_.resize((size_t)nc);
            pd_.resize((size_t)np * 32); cd_.resize((size_t)nc * 32);
            for (int i = 0; i < np; i++) {
                pk_[(size_t)i] = detail::from_cv(prev_features.keypoints[(size_t)i]);
                std::memcpy(pd_.data() + (size_t)i * 32, prev_features.descriptors.ptr<uint8_t>(i), 32);
            }
            for (int i = 0; i < nc; i++) {
                ck_[(size_t)i] = detail::from_cv(curr_features.keypoints[(size_t)i]);
                std::memcpy(cd_.data() + (size_t)i * 32, curr_features.descriptors.ptr<uint8_t>(i), 32);
            }
            out_.resize((size_t)np);
            int n = 0;
            dev_.check(vo_match_features(dev_.get(), pk_.data(), pd_.data(), np, ck_.data(), cd_.data(), nc,
                                         ratio_threshold_, out_.data(), &n),
                       "vo_match_features");
            detail::to_matches(out_, n, result);
        } catch (const std::exception &e) {
            VO_LOG_ERROR("feature_matcher", "error in matching: %s", e.what());
            result.clear();
        }
        return result;
    }

private:
    detail::DeviceContext dev_;
    std::vector<vo_kp> pk_, ck_;
    std::vector<uint8_t> pd_, cd_;
    std::vector<vo_match> out_;
    float ratio_threshold_;
};

}  // namespace vo
