// Drives the C++ facades (FeatureDetector / FeatureMatcher / FrameProcessor)
// exactly as vo_node does (reference src/vo_node.cpp:79-80,170-173,681) over a
// raw frame dump written by tests/test_gpu_facade.py, and prints the results
// for the Python side to compare with the oracle.
//
// usage: facade_test <dump> <width> <height> <channels> <n_frames> <fx> <fy> <cx> <cy> [imu_tx_mm|-] [fast|orb] [out]
// With `out`, everything that crosses the class boundary is written there in binary, per processFrame call and
// for the standalone detectFeatures / match calls: key points (all seven cv::KeyPoint fields), descriptor rows,
// DMatch records with both point lists, the 3-D points.
// dump layout: n_frames x (H*W*C bytes), then n_frames x (H*W float32 depth)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "visual_odometry/frame_processor.hpp"
#include "visual_odometry/image_processor.hpp"

static FILE *g_out = nullptr;

static void put_i32(int v) { if (g_out) std::fwrite(&v, 4, 1, g_out); }
static void put_f32(float v) { if (g_out) std::fwrite(&v, 4, 1, g_out); }

static void dump_features(const vo::Features &f)
{
    if (!g_out) return;
    put_i32((int)f.keypoints.size());
    for (const cv::KeyPoint &k : f.keypoints) {
        put_f32(k.pt.x); put_f32(k.pt.y); put_f32(k.size); put_f32(k.angle); put_f32(k.response);
        put_i32(k.octave); put_i32(k.class_id);
    }
    put_i32(f.descriptors.rows);
    put_i32(f.descriptors.cols);
    for (int i = 0; i < f.descriptors.rows; i++) std::fwrite(f.descriptors.ptr<uint8_t>(i), 1, (size_t)f.descriptors.cols, g_out);
}

static void dump_matches(const vo::FeatureMatches &m)
{
    if (!g_out) return;
    put_i32((int)m.matches.size());
    for (size_t i = 0; i < m.matches.size(); i++) {
        put_i32(m.matches[i].queryIdx); put_i32(m.matches[i].trainIdx); put_i32(m.matches[i].imgIdx);
        put_f32(m.matches[i].distance);
        put_f32(m.prev_points[i].x); put_f32(m.prev_points[i].y);
        put_f32(m.curr_points[i].x); put_f32(m.curr_points[i].y);
    }
    put_i32((int)m.prev_points_3d.size());
    for (const cv::Point3f &p : m.prev_points_3d) { put_f32(p.x); put_f32(p.y); put_f32(p.z); }
}

static unsigned long long fnv1a(const cv::Mat &m)
{
    unsigned long long hsh = 1469598103934665603ull;
    for (int y = 0; y < m.rows; y++)
        for (int x = 0; x < m.cols; x++) hsh = (hsh ^ m.ptr<uint8_t>(y)[x]) * 1099511628211ull;
    return hsh;
}

int main(int argc, char **argv)
{
    if (argc < 10) return 2;
    const char *path = argv[1];
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), c = std::atoi(argv[4]), n = std::atoi(argv[5]);
    vo::CameraParams cam;
    cam.fx = std::atof(argv[6]); cam.fy = std::atof(argv[7]); cam.cx = std::atof(argv[8]); cam.cy = std::atof(argv[9]);
    cam.width = w; cam.height = h;
    const bool use_imu = argc > 10 && std::string(argv[10]) != "-";
    const double imu_tx = use_imu ? std::atof(argv[10]) : 0.0;
    const bool orb = argc > 11 && std::string(argv[11]) == "orb";
    if (argc > 12) g_out = std::fopen(argv[12], "wb");

    std::vector<uint8_t> frames((size_t)n * w * h * c);
    std::vector<float> depth((size_t)n * w * h);
    FILE *f = std::fopen(path, "rb");
    if (!f) return 3;
    if (std::fread(frames.data(), 1, frames.size(), f) != frames.size()) return 4;
    if (std::fread(depth.data(), sizeof(float), depth.size(), f) != depth.size()) return 4;
    std::fclose(f);

    auto detector = std::make_shared<vo::FeatureDetector>();
    auto matcher = std::make_shared<vo::FeatureMatcher>();
    detector->setMaxFeatures(2000);
    detector->setScaleFactor(1.2f);
    detector->setNLevels(8);
    detector->setVisualizationType("rich");
    // opt-in: the three setters above take effect (full ORB detector); default: the reference's effective
    // FAST + ORB::compute with 1000 / 20, whatever the setters said
    if (orb) detector->setDetectorMode(vo::DetectorMode::Orb);
    vo::FrameProcessor fp(detector, matcher);

    vo::Features prev_feat;
    for (int i = 0; i < n; i++) {
        cv::Mat rgb(h, w, CV_MAKETYPE(CV_8U, c), frames.data() + (size_t)i * w * h * c);
        cv::Mat dep(h, w, CV_32FC1, depth.data() + (size_t)i * w * h);
        vo::ImuPredictedPose imu;
        if (use_imu && i > 0) {
            imu.valid = true;
            imu.R = cv::Mat::eye(3, 3, CV_64F);
            imu.t = cv::Mat::zeros(3, 1, CV_64F);
            imu.t.at<double>(0, 0) = imu_tx;
            imu.angular_rate = 0.0;
        }
        auto r = fp.processFrame(rgb, dep, cam, i == 0, true, imu);
        dump_features(r.features);
        dump_matches(r.matches);
        std::printf("frame %d kp %zu matches %zu p3d %zu pnp %d inl %d tot %d", i, r.features.keypoints.size(),
                    r.matches.size(), r.matches.prev_points_3d.size(), (int)r.pnp_success, r.pnp_inliers,
                    r.pnp_total_matches);
        if (r.pnp_success) {
            std::printf(" R");
            for (int k = 0; k < 9; k++) std::printf(" %.17g", r.R.at<double>(k / 3, k % 3));
            std::printf(" t %.17g %.17g %.17g", r.t.at<double>(0, 0), r.t.at<double>(1, 0), r.t.at<double>(2, 0));
        }
        std::printf("\n");
        // the standalone operators, as vo_node's components would use them
        if (i == 1) {
            cv::Mat bgr(h, w, CV_8UC3);
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++)
                    for (int k = 0; k < 3; k++) bgr.ptr<uint8_t>(y)[3 * x + k] = rgb.ptr<uint8_t>(y)[c * x + k];
            vo::Features cur = detector->detectFeatures(bgr);  // defaults 1000 / 20
            vo::FeatureMatches m = matcher->match(prev_feat, cur, cv::Mat(), cv::Mat());
            std::printf("ops kp %zu desc_rows %d matches %zu\n", cur.keypoints.size(), cur.descriptors.rows, m.size());
            dump_features(cur);
            dump_matches(m);
        }
        if (i == 0) {
            cv::Mat bgr(h, w, CV_8UC3);
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++)
                    for (int k = 0; k < 3; k++) bgr.ptr<uint8_t>(y)[3 * x + k] = rgb.ptr<uint8_t>(y)[c * x + k];
            prev_feat = detector->detectFeatures(bgr);
            dump_features(prev_feat);

            // ImageProcessor as declared in vo_node.hpp:56 (defaults, then histogram equalisation + 5x5)
            vo::ImageProcessor ip;
            vo::ProcessedImages p0 = ip.processImage(bgr);
            std::printf("improc plain %llx enhanced_empty %d\n", fnv1a(p0.gray), (int)p0.enhanced.empty());
            vo::ProcessedImages p1 = ip.processImage(bgr, true);
            std::printf("improc default %llx %llx %llx %llx\n", fnv1a(p1.gray), fnv1a(p1.enhanced), fnv1a(p1.denoised),
                        fnv1a(p1.masked));
            ip.setEnableHistogramEq(true);
            ip.setGaussianBlurSize(5);
            ip.setGaussianSigma(1.4);
            vo::ProcessedImages p2 = ip.processImage(bgr, true);
            std::printf("improc histeq5 %llx %llx %llx %llx\n", fnv1a(p2.gray), fnv1a(p2.enhanced), fnv1a(p2.denoised),
                        fnv1a(p2.masked));
            cv::Mat und = ip.undistort(p0.gray, cam, {-0.28, 0.09, 0.0012, -0.0007, -0.015});
            std::printf("improc undistort %llx\n", fnv1a(und));
            ip.setGaussianBlurSize(4);
            try {
                ip.processImage(bgr, true);
                std::printf("improc_even no_throw\n");
            } catch (const std::exception &) {
                std::printf("improc_even threw\n");
            }
        }
    }
    // error convention: detectFeatures swallows, processFrame throws
    vo::Features bad = detector->detectFeatures(cv::Mat());
    std::printf("empty_detect kp %zu\n", bad.keypoints.size());
    try {
        cv::Mat gray(h, w, CV_8UC1, frames.data());
        fp.processFrame(gray, cv::Mat(), cam);
        std::printf("gray_process no_throw\n");
    } catch (const std::exception &e) {
        std::printf("gray_process threw\n");
    }
    if (g_out) std::fclose(g_out);
    return 0;
}
