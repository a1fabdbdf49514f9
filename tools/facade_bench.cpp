// Latency of the path the reference node takes per frame (reference
// src/vo_node.cpp:681: frame_processor_->processFrame(rgb, depth, camera, ...)):
// ONE camera stream, frames and depth handed over in ordinary host memory, every
// result (key points, descriptors, matches, 3-D points, pose) returned by value.
//
// usage: facade_bench <dump> <width> <height> <channels> <n_frames> <fx> <fy> <cx> <cy> <iterations> [fast|orb] [pinned]
// pinned: the frames live in vo_host_alloc() memory, as a camera wrapper that clones into pinned buffers would
// hand them over (default: ordinary pageable memory, like the cv::Mat clones of the reference node)
// dump layout: n_frames x (H*W*C bytes), then n_frames x (H*W float32 depth in mm); the frames are
// replayed cyclically (write a sequence whose last frame is one step from its first).
// prints one JSON object: per-frame wall-clock milliseconds of processFrame.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "visual_odometry/frame_processor.hpp"

int main(int argc, char **argv)
{
    if (argc < 11) {
        std::fprintf(stderr, "usage: %s <dump> <w> <h> <c> <n> <fx> <fy> <cx> <cy> <iterations> [fast|orb]\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), c = std::atoi(argv[4]), n = std::atoi(argv[5]);
    vo::CameraParams cam;
    cam.fx = std::atof(argv[6]); cam.fy = std::atof(argv[7]); cam.cx = std::atof(argv[8]); cam.cy = std::atof(argv[9]);
    cam.width = w; cam.height = h;
    const int iters = std::atoi(argv[10]);
    const bool orb = argc > 11 && std::strcmp(argv[11], "orb") == 0;

    const bool pinned = argc > 12 && std::strcmp(argv[12], "pinned") == 0;
    const size_t fbytes = (size_t)n * w * h * c, delems = (size_t)n * w * h;
    std::vector<uint8_t> frames_v(pinned ? 0 : fbytes);
    std::vector<float> depth_v(pinned ? 0 : delems);
    uint8_t *frames = pinned ? static_cast<uint8_t *>(vo_host_alloc(fbytes)) : frames_v.data();
    float *depth = pinned ? static_cast<float *>(vo_host_alloc(delems * sizeof(float))) : depth_v.data();
    if (!frames || !depth) return 5;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 3;
    if (std::fread(frames, 1, fbytes, f) != fbytes) return 4;
    if (std::fread(depth, sizeof(float), delems, f) != delems) return 4;
    std::fclose(f);

    auto detector = std::make_shared<vo::FeatureDetector>();
    auto matcher = std::make_shared<vo::FeatureMatcher>();
    detector->setMaxFeatures(2000);
    detector->setScaleFactor(1.2f);
    detector->setNLevels(8);
#if 1
    if (orb) detector->setDetectorMode(vo::DetectorMode::Orb);
#else
    (void)orb;
#endif
    vo::FrameProcessor fp(detector, matcher);

    const int warm = std::min(20, iters / 4);
    std::vector<double> ms;
    ms.reserve((size_t)iters);
    size_t kp = 0, matches = 0, poses = 0;
    for (int i = 0; i < iters + warm; i++) {
        const int k = i % n;
        cv::Mat rgb(h, w, CV_MAKETYPE(CV_8U, c), frames + (size_t)k * w * h * c);
        cv::Mat dep(h, w, CV_32FC1, depth + (size_t)k * w * h);
        const auto t0 = std::chrono::steady_clock::now();
        auto r = fp.processFrame(rgb, dep, cam, i == 0, true);
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= warm) {
            ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            kp += r.features.keypoints.size();
            matches += r.matches.size();
            poses += r.pnp_success ? 1 : 0;
        }
    }
    std::sort(ms.begin(), ms.end());
    double sum = 0;
    for (double v : ms) sum += v;
    std::printf("{\"frames\": %d, \"median_ms\": %.4f, \"mean_ms\": %.4f, \"p90_ms\": %.4f, \"min_ms\": %.4f, "
                "\"fps\": %.1f, \"mean_keypoints\": %.1f, \"mean_matches\": %.1f, \"pose_rate\": %.3f, "
                "\"detector\": \"%s\", \"host_memory\": \"%s\"}\n",
                iters, ms[ms.size() / 2], sum / ms.size(), ms[(size_t)(ms.size() * 0.9)], ms.front(),
                1e3 * ms.size() / sum, (double)kp / iters, (double)matches / iters, (double)poses / iters,
                orb ? "orb" : "fast", pinned ? "pinned" : "pageable");
    return 0;
}
