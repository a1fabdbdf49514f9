// Shared plumbing of the facades: logging macros (rclcpp when available,
// stderr otherwise), a growing vo_ctx holder and POD <-> cv conversions.
#pragma once

#include <cstdio>
#include <stdexcept>
#include <string>

#include "visual_odometry/types.hpp"
#include "vo_hip.h"

#ifdef VO_WITH_RCLCPP
#include <rclcpp/rclcpp.hpp>
#define VO_LOG_INFO(name, ...) RCLCPP_INFO(rclcpp::get_logger(name), __VA_ARGS__)
#define VO_LOG_WARN(name, ...) RCLCPP_WARN(rclcpp::get_logger(name), __VA_ARGS__)
#define VO_LOG_ERROR(name, ...) RCLCPP_ERROR(rclcpp::get_logger(name), __VA_ARGS__)
#else
#define VO_LOG_IMPL(level, name, ...)                  \
    do {                                               \
        std::fprintf(stderr, "[%s] [%s] ", level, name); \
        std::fprintf(stderr, __VA_ARGS__);             \
        std::fprintf(stderr, "\n");                    \
    } while (0)
#define VO_LOG_INFO(name, ...) ((void)0)
#define VO_LOG_WARN(name, ...) VO_LOG_IMPL("WARN", name, __VA_ARGS__)
#define VO_LOG_ERROR(name, ...) VO_LOG_IMPL("ERROR", name, __VA_ARGS__)
#endif

namespace vo {
namespace detail {

// One vo_ctx, re-created when a larger frame or feature budget shows up.
class DeviceContext {
public:
    DeviceContext() = default;
    DeviceContext(const DeviceContext &) = delete;
    DeviceContext &operator=(const DeviceContext &) = delete;
    ~DeviceContext() { vo_destroy(ctx_); }

    // returns true when the context was (re)created
    bool ensure(int width, int height, int max_features)
    {
        if (ctx_ && width <= cfg_.max_width && height <= cfg_.max_height && max_features <= cfg_.max_features)
            return false;
        vo_destroy(ctx_);
        ctx_ = nullptr;
        cfg_ = vo_cfg{};
        cfg_.max_width = width < 64 ? 64 : width;
        cfg_.max_height = height < 64 ? 64 : height;
        cfg_.max_features = max_features < 16 ? 16 : max_features;
        cfg_.max_batch = 1;
        int device = 0;
        if (const char *e = std::getenv("VO_DEVICE")) device = std::atoi(e);
        ctx_ = vo_create(device, &cfg_);
        if (!ctx_) throw std::runtime_error(std::string("vo_create failed: ") + vo_last_error(nullptr));
        return true;
    }
    vo_ctx *get() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != VO_OK) throw std::runtime_error(std::string(what) + ": " + vo_last_error(ctx_));
    }

private:
    vo_ctx *ctx_{nullptr};
    vo_cfg cfg_{};
};

inline cv::KeyPoint to_cv(const vo_kp &k)
{
    cv::KeyPoint o;
    o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response;
    o.octave = k.octave; o.class_id = -1;
    return o;
}

inline vo_kp from_cv(const cv::KeyPoint &k)
{
    return vo_kp{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave};
}

inline void to_features(const vo_kp *kps, const uint8_t *desc, int n, Features &out)
{
    out.keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++) out.keypoints[(size_t)i] = to_cv(kps[(size_t)i]);
    if (n > 0) {
        out.descriptors.create(n, 32, CV_8UC1);
        for (int i = 0; i < n; i++) std::memcpy(out.descriptors.ptr<uint8_t>(i), desc + (size_t)i * 32, 32);
    } else {
        out.descriptors = cv::Mat();
    }
}

inline void to_features(const std::vector<vo_kp> &kps, const std::vector<uint8_t> &desc, int n, Features &out)
{
    to_features(kps.data(), desc.data(), n, out);
}

inline void to_matches(const vo_match *m, int n, FeatureMatches &out)
{
    out.clear();
    out.reserve((size_t)n);
    for (int i = 0; i < n; i++) {
        out.matches.emplace_back(m[(size_t)i].query, m[(size_t)i].train, m[(size_t)i].distance);
        out.prev_points.emplace_back(m[(size_t)i].px, m[(size_t)i].py);
        out.curr_points.emplace_back(m[(size_t)i].cx, m[(size_t)i].cy);
    }
}

inline void to_matches(const std::vector<vo_match> &m, int n, FeatureMatches &out) { to_matches(m.data(), n, out); }

// cv::Rodrigues(rvec) -> 3x3 CV_64F (reference src/frame_processor.cpp:193)
inline cv::Mat rodrigues_to_matrix(const double r[3])
{
    cv::Mat R = cv::Mat::eye(3, 3, CV_64F);
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < 2.220446049250313e-16) return R;
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c;
    const double x = r[0] / theta, y = r[1] / theta, z = r[2] / theta;
    const double v[9] = {c + c1 * x * x, c1 * x * y - s * z, c1 * x * z + s * y,
                         c1 * x * y + s * z, c + c1 * y * y, c1 * y * z - s * x,
                         c1 * x * z - s * y, c1 * y * z + s * x, c + c1 * z * z};
    for (int i = 0; i < 9; i++) R.at<double>(i / 3, i % 3) = v[i];
    return R;
}

// inverse of the above for a rotation matrix (3x3 CV_64F)
inline void matrix_to_rodrigues(const cv::Mat &R, double r[3])
{
    double rx = R.at<double>(2, 1) - R.at<double>(1, 2), ry = R.at<double>(0, 2) - R.at<double>(2, 0),
           rz = R.at<double>(1, 0) - R.at<double>(0, 1);
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R.at<double>(0, 0) + R.at<double>(1, 1) + R.at<double>(2, 2) - 1) * 0.5;
    c = c > 1. ? 1. : (c < -1. ? -1. : c);
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R.at<double>(0, 0) + 1) * 0.5; rx = std::sqrt(t > 0 ? t : 0);
        t = (R.at<double>(1, 1) + 1) * 0.5; ry = std::sqrt(t > 0 ? t : 0) * (R.at<double>(0, 1) < 0 ? -1. : 1.);
        t = (R.at<double>(2, 2) + 1) * 0.5; rz = std::sqrt(t > 0 ? t : 0) * (R.at<double>(0, 2) < 0 ? -1. : 1.);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R.at<double>(1, 2) > 0) != (ry * rz > 0))
            rz = -rz;
        const double k = theta / std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * k; r[1] = ry * k; r[2] = rz * k;
        return;
    }
    const double vth = theta / (2 * s);
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

}  // namespace detail
}  // namespace vo
