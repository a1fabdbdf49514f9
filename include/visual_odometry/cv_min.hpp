// Minimal stand-in for the few OpenCV core types the front-end interfaces use
// (cv::Mat, cv::KeyPoint, cv::DMatch, cv::Point2f/3f).  Only compiled when the
// real OpenCV headers are not available (VO_WITH_OPENCV undefined); with
// OpenCV the facades use <opencv2/core.hpp> and nothing from this file.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace cv {

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

template <typename T>
struct Point_ {
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
using Point2f = Point_<float>;
using Point2d = Point_<double>;

template <typename T>
struct Point3_ {
    T x{}, y{}, z{};
    Point3_() = default;
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
using Point3f = Point3_<float>;

struct Size {
    int width{0}, height{0};
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    bool operator!=(const Size &o) const { return width != o.width || height != o.height; }
};

struct KeyPoint {
    Point2f pt;
    float size{0};
    float angle{-1};
    float response{0};
    int octave{0};
    int class_id{-1};
};

struct DMatch {
    int queryIdx{-1}, trainIdx{-1}, imgIdx{-1};
    float distance{0};
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(0), distance(d) {}
};

// Dense, reference-counted 2-D array (continuous rows unless wrapping user data).
class Mat {
public:
    int rows{0}, cols{0};
    size_t step{0};
    uint8_t *data{nullptr};

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *user, size_t step_ = 0)
        : rows(r), cols(c), step(step_ ? step_ : (size_t)c * elem(type)), data((uint8_t *)user), type_(type) {}

    static int depth_bytes(int type)
    {
        switch (type & 7) {
            case CV_8U: return 1;
            case CV_32S: case CV_32F: return 4;
            case CV_64F: return 8;
            default: return 1;
        }
    }
    static size_t elem(int type) { return (size_t)depth_bytes(type) * (size_t)((type >> 3) + 1); }

    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && owner_) return;
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elem(type);
        owner_ = std::shared_ptr<uint8_t>(new uint8_t[step * (size_t)r + 1], std::default_delete<uint8_t[]>());
        data = owner_.get();
    }
    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        std::memset(m.data, 0, m.step * (size_t)r);
        return m;
    }
    static Mat eye(int r, int c, int type)
    {
        Mat m = zeros(r, c, type);
        for (int i = 0; i < (r < c ? r : c); i++) {
            if ((type & 7) == CV_64F) m.at<double>(i, i) = 1.0;
            else if ((type & 7) == CV_32F) m.at<float>(i, i) = 1.f;
            else m.data[i * m.step + i] = 1;
        }
        return m;
    }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    int depth() const { return type_ & 7; }
    size_t elemSize() const { return elem(type_); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    Mat clone() const
    {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int r = 0; r < rows; r++) std::memcpy(m.data + r * m.step, data + r * step, (size_t)cols * elemSize());
        return m;
    }
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    template <typename T> T &at(int r, int c) { return ptr<T>(r)[c]; }
    template <typename T> const T &at(int r, int c) const { return ptr<T>(r)[c]; }
    template <typename T> T &at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }

private:
    int type_{0};
    std::shared_ptr<uint8_t> owner_;
};

inline double norm(const Mat &m)
{
    double s = 0;
    for (int r = 0; r < m.rows; r++)
        for (int c = 0; c < m.cols; c++) {
            const double v = m.depth() == CV_64F ? m.at<double>(r, c) : (double)m.at<float>(r, c);
            s += v * v;
        }
    return std::sqrt(s);
}

}  // namespace cv
