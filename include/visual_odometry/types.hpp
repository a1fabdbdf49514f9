// Boundary types of the visual-odometry front end, same names and members as
// the reference's include/visual_odometry/types.hpp:7-69.
#pragma once

#include <cmath>
#include <vector>

#ifdef VO_WITH_OPENCV
#include <opencv2/core.hpp>
#else
#include "visual_odometry/cv_min.hpp"
#endif

namespace vo {

struct ProcessedImages {
    cv::Mat gray;
    cv::Mat enhanced;
    cv::Mat denoised;
    cv::Mat masked;
};

struct CameraParams {
    double fx{0.0};
    double fy{0.0};
    double cx{0.0};
    double cy{0.0};
    int width{0};
    int height{0};

    cv::Mat getCameraMatrix() const
    {
        cv::Mat K = cv::Mat::eye(3, 3, CV_64F);
        K.at<double>(0, 0) = fx;
        K.at<double>(1, 1) = fy;
        K.at<double>(0, 2) = cx;
        K.at<double>(1, 2) = cy;
        return K;
    }
};

struct Features {
    std::vector<cv::KeyPoint> keypoints;
    cv::Mat descriptors;    // N x 32, CV_8U
    cv::Mat visualization;  // left empty, as in the reference
};

struct FeatureMatches {
    std::vector<cv::DMatch> matches;
    std::vector<cv::Point2f> prev_points;
    std::vector<cv::Point2f> curr_points;
    std::vector<cv::Point3f> prev_points_3d;  // current-frame 3-D points (mm), PnP input

    void reserve(size_t n)
    {
        matches.reserve(n);
        prev_points.reserve(n);
        curr_points.reserve(n);
        prev_points_3d.reserve(n);
    }
    void clear()
    {
        matches.clear();
        prev_points.clear();
        curr_points.clear();
        prev_points_3d.clear();
    }
    size_t size() const { return matches.size(); }
    bool empty() const { return matches.empty(); }
};

}  // namespace vo
