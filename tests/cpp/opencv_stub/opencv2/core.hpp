// COMPILE-ONLY declaration stub of the few cv:: names include/adf_ximgproc.hpp touches in its OpenCV branch.
// This image has no OpenCV, so that branch (the one a cv::Mat pipeline uses) would otherwise never meet a
// compiler.  Declarations only -- nothing here is implemented, linked or run, and it is NOT a stand-in for
// building the reference: it exists so that `g++ -fsyntax-only` type-checks the adaptor (tests/test_cpp_adaptor.py).
// Signatures follow the public OpenCV 3.x/4.x API as documented (opencv2/core/mat.hpp, types.hpp, cvstd.hpp).
#pragma once
#include <cstddef>
#include <memory>

#define CV_8U 0
#define CV_16S 3
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))

namespace cv {

typedef unsigned char uchar;

template <class T> struct Ptr : std::shared_ptr<T> {
    using std::shared_ptr<T>::shared_ptr;
    Ptr() {}
    template <class Y> Ptr<Y> dynamicCast() const;
};

struct MatStep {
    operator size_t() const;
};

class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    void create(int rows, int cols, int type);
    int depth() const;
    int channels() const;
    int type() const;
    bool empty() const;
    int rows, cols;
    uchar* data;
    MatStep step;
};

template <class T> class Rect_ {
public:
    Rect_();
    Rect_(T x, T y, T width, T height);
    T area() const;
    T x, y, width, height;
};
typedef Rect_<int> Rect;

} // namespace cv
