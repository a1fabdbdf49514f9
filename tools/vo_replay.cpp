// ROS-free replay of a recorded sequence through the C++ facades, with the
// node's pose accumulation (reference src/vo_node.cpp:681-774): prints one
// trajectory line per frame so that accuracy can be evaluated end to end.
//
// usage: vo_replay <dump> <width> <height> <channels> <n_frames> <fx> <fy> <cx> <cy>
// dump layout: n_frames x (H*W*C bytes), then n_frames x (H*W float32 depth in mm)
// output:  frame i kp K matches M pnp S inl I  T <12 numbers of T_0_from_curr, optical, mm>
//          body x y z roll pitch yaw  dt_detect_ms dt_match_ms
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "visual_odometry/frame_processor.hpp"
#include "visual_odometry/pose_accumulator.hpp"

int main(int argc, char **argv)
{
    if (argc < 10) {
        std::fprintf(stderr, "usage: %s <dump> <width> <height> <channels> <n_frames> <fx> <fy> <cx> <cy>\n", argv[0]);
        return 2;
    }
    const char *path = argv[1];
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), c = std::atoi(argv[4]), n = std::atoi(argv[5]);
    vo::CameraParams cam;
    cam.fx = std::atof(argv[6]); cam.fy = std::atof(argv[7]); cam.cx = std::atof(argv[8]); cam.cy = std::atof(argv[9]);
    cam.width = w; cam.height = h;

    const size_t frame_bytes = (size_t)w * h * c, depth_elems = (size_t)w * h;
    std::vector<uint8_t> frame(frame_bytes);
    std::vector<float> depth(depth_elems);
    FILE *ff = std::fopen(path, "rb"), *fd = std::fopen(path, "rb");
    if (!ff || !fd) return 3;
    if (std::fseek(fd, (long)(frame_bytes * n), SEEK_SET) != 0) return 3;

    auto detector = std::make_shared<vo::FeatureDetector>();
    auto matcher = std::make_shared<vo::FeatureMatcher>();
    vo::FrameProcessor fp(detector, matcher);
    vo::PoseAccumulator acc;

    for (int i = 0; i < n; i++) {
        if (std::fread(frame.data(), 1, frame_bytes, ff) != frame_bytes) return 4;
        if (std::fread(depth.data(), sizeof(float), depth_elems, fd) != depth_elems) return 4;
        cv::Mat rgb(h, w, CV_MAKETYPE(CV_8U, c), frame.data());
        cv::Mat dep(h, w, CV_32FC1, depth.data());
        auto r = fp.processFrame(rgb, dep, cam, i == 0, true);
        if (r.pnp_success) {
            double R[9], t[3];
            for (int k = 0; k < 9; k++) R[k] = r.R.at<double>(k / 3, k % 3);
            for (int k = 0; k < 3; k++) t[k] = r.t.at<double>(k, 0);
            acc.update(true, R, t);
        }
        const double *T = acc.T_0_from_curr();
        const vo::BodyPose p = acc.pose();
        std::printf("frame %d kp %zu matches %zu pnp %d inl %d T", i, r.features.keypoints.size(), r.matches.size(),
                    (int)r.pnp_success, r.pnp_inliers);
        for (int k = 0; k < 12; k++) std::printf(" %.9g", T[k]);
        std::printf(" body %.6f %.6f %.6f %.6f %.6f %.6f %.3f %.3f\n", p.x, p.y, p.z, p.roll, p.pitch, p.yaw,
                    r.feature_detection_time, r.feature_matching_time);
    }
    std::fclose(ff);
    std::fclose(fd);
    return 0;
}
