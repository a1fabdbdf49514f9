// vo::ImageProcessor with the reference's interface
// (reference include/visual_odometry/image_processor.hpp:8-38,
// src/image_processor.cpp:6-79), computing on the GPU through vo_hip.h.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "visual_odometry/vo_device.hpp"

namespace vo {

class ImageProcessor {
public:
    ImageProcessor() = default;
    ~ImageProcessor() = default;

    // gray always; with enhance also the equalised, blurred and masked planes.
    // The returned Mats share this object's buffers, as in the reference
    // (src/image_processor.cpp:11,16,20,24): the next call overwrites them.
    // Throws like the reference's OpenCV calls would (no try/catch there):
    // std::runtime_error for an even blur size, for inputs that are not 8-bit
    // 1- or 3-channel when enhance is set, and for device failures.
    ProcessedImages processImage(const cv::Mat &input, bool enhance = false)
    {
        ProcessedImages result;
        if (input.empty()) throw std::runtime_error("ImageProcessor: empty input");
        const int w = input.cols, h = input.rows, cn = input.channels();
        if (cn != 1 && cn != 3) {
            // the reference copies any non-3-channel input to the gray buffer and
            // cv::equalizeHist rejects what is not CV_8UC1
            if (enhance) throw std::runtime_error("ImageProcessor: enhance needs a 1- or 3-channel image");
            gray_buffer_ = input.clone();
            result.gray = gray_buffer_;
            return result;
        }
        dev_.ensure(w, h, 16);
        gray_buffer_.create(h, w, CV_8UC1);
        if (enhance) {
            enhanced_buffer_.create(h, w, CV_8UC1);
            denoised_buffer_.create(h, w, CV_8UC1);
            masked_buffer_.create(h, w, CV_8UC1);
        }
        dev_.check(vo_image_process(dev_.get(), input.ptr<uint8_t>(0), w, h, (int)input.step, cn, enhance ? 1 : 0,
                                    enable_histogram_eq_ ? 1 : 0, gaussian_blur_size_, gaussian_sigma_,
                                    gray_buffer_.ptr<uint8_t>(0), enhance ? enhanced_buffer_.ptr<uint8_t>(0) : nullptr,
                                    enhance ? denoised_buffer_.ptr<uint8_t>(0) : nullptr,
                                    enhance ? masked_buffer_.ptr<uint8_t>(0) : nullptr, w),
                   "vo_image_process");
        result.gray = gray_buffer_;
        if (enhance) {
            result.enhanced = enhanced_buffer_;
            result.denoised = denoised_buffer_;
            result.masked = masked_buffer_;
        }
        return result;
    }

    // Addition to the reference's interface: cv::undistort(gray, K, dist) for
    // cameras that deliver raw frames (the reference's ZED frames arrive
    // rectified, src/zed_interface.cpp:112).  dist = k1 k2 p1 p2 [k3 [k4 k5 k6]].
    // The returned Mat shares this object's buffer.
    cv::Mat undistort(const cv::Mat &gray, const CameraParams &cam, const std::vector<double> &dist)
    {
        if (gray.empty() || gray.channels() != 1)
            throw std::runtime_error("ImageProcessor::undistort: 8-bit single-channel image expected");
        const int w = gray.cols, h = gray.rows;
        dev_.ensure(w, h, 16);
        undistorted_buffer_.create(h, w, CV_8UC1);
        dev_.check(vo_undistort(dev_.get(), gray.ptr<uint8_t>(0), w, h, (int)gray.step, cam.fx, cam.fy, cam.cx, cam.cy,
                                dist.data(), (int)dist.size(), undistorted_buffer_.ptr<uint8_t>(0), w),
                   "vo_undistort");
        return undistorted_buffer_;
    }

    void setGaussianBlurSize(int size) { gaussian_blur_size_ = size; }
    void setGaussianSigma(double sigma) { gaussian_sigma_ = sigma; }
    void setEnableHistogramEq(bool enable) { enable_histogram_eq_ = enable; }

private:
    detail::DeviceContext dev_;
    cv::Mat gray_buffer_, enhanced_buffer_, denoised_buffer_, masked_buffer_, undistorted_buffer_;
    int gaussian_blur_size_{3};
    double gaussian_sigma_{1.0};
    bool enable_histogram_eq_{false};
};

}  // namespace vo
