// amos_cv.h -- the handful of OpenCV core types the front-end API is written against.
//
// With AMOS_WITH_OPENCV defined this is just <opencv2/core.hpp> and the classes in this directory
// are source-compatible drop-ins for the reference's (include/ORBextractor.h, ORBmatcher.h).
// Where OpenCV is not installed (the build and GPU boxes of this project) a minimal stand-in with the
// same names, members and memory layout is used so the classes and their tests still compile and
// run; it covers exactly what the front-end touches: Mat headers over 8-bit / 64-bit planes, ROI,
// clone, KeyPoint, Point, Size, Rect, InputArray / OutputArray.
#pragma once
#ifdef AMOS_WITH_OPENCV
#include <opencv2/core.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_64FC1 6
#ifndef CV_PI
#define CV_PI 3.1415926535897932384626433832795
#endif

namespace cv {

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_ &operator*=(T s) { x *= s; y *= s; return *this; }
    Point_ operator*(T s) const { return Point_(x * s, y * s); }
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

// Same field order and widths as cv::KeyPoint.
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

class Mat {
public:
    int rows, cols;
    unsigned char *data;
    size_t step;  // bytes per row

    Mat() : rows(0), cols(0), data(nullptr), step(0), type_(0) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size s, int type) : Mat() { create(s.height, s.width, type); }
    Mat(int r, int c, int type, void *ext, size_t step_ = 0)
        : rows(r), cols(c), data((unsigned char *)ext), step(step_ ? step_ : (size_t)c * elemSize(type)), type_(type) {}
    Mat(const Mat &m, const Rect &roi)
        : rows(roi.height), cols(roi.width), data(m.data + (size_t)roi.y * m.step + (size_t)roi.x * elemSize(m.type_)), step(m.step),
          type_(m.type_), owner_(m.owner_) {}

    static size_t elemSize(int type) { return type == CV_8U ? 1 : type == CV_32F ? 4 : 8; }
    size_t elemSize() const { return elemSize(type_); }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    void create(int r, int c, int type)
    {
        if (data && rows == r && cols == c && type_ == type && isContinuous()) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize(type);
        owner_.reset(new unsigned char[(size_t)r * step + 64], std::default_delete<unsigned char[]>());
        data = owner_.get();
    }
    void release() { owner_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
        return m;
    }
    Mat operator()(const Rect &roi) const { return Mat(*this, roi); }
    Mat row(int y) const { return Mat(*this, Rect(0, y, cols, 1)); }
    Mat rowRange(int y0, int y1) const { return Mat(*this, Rect(0, y0, cols, y1 - y0)); }
    template <typename T> T *ptr(int y = 0) { return reinterpret_cast<T *>(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return reinterpret_cast<const T *>(data + (size_t)y * step); }
    unsigned char *ptr(int y = 0) { return data + (size_t)y * step; }
    const unsigned char *ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T &at(int y, int x) { return ptr<T>(y)[x]; }
    template <typename T> const T &at(int y, int x) const { return ptr<T>(y)[x]; }
    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        std::memset(m.data, 0, (size_t)r * m.step);
        return m;
    }

private:
    int type_;
    std::shared_ptr<unsigned char> owner_;
};

class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat &m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
private:
    const Mat *m_;
};
class _OutputArray {
public:
    _OutputArray(Mat &m) : m_(&m) {}
    void create(int r, int c, int type) const { m_->create(r, c, type); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat *m_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

}  // namespace cv
#endif  // AMOS_WITH_OPENCV
