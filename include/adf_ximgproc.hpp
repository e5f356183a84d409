// adf_ximgproc.hpp -- header-only C++ adaptor: the reference's operator interface over the C-ABI.
//
// Mirrors cv::ximgproc for the disparity-filter path -- same class and function names, argument
// order and meaning, defaults and error behaviour (exceptions) as
//   modules/ximgproc/include/opencv2/ximgproc/disparity_filter.hpp  (DF.hpp:52-149)
//   modules/ximgproc/include/opencv2/ximgproc/edge_filter.hpp       (EF.hpp:361-413)
// so that a stereo pipeline switches by changing a namespace.  All compute happens in
// libadf_wls.so (HIP kernels); this header only marshals cv::Mat-like images into adf_wls.h calls.
//
// With OpenCV available (opencv2/core.hpp on the include path) the types are cv::Mat / cv::Rect /
// cv::Ptr and the matcher factories take cv::StereoMatcher.  Without it (this repo's CI image has no
// OpenCV) a minimal Mat / Rect stands in and the matcher factories take the three numbers they read.
#pragma once

#include "adf_wls.h"

#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(ADF_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define ADF_HAVE_OPENCV 1
#if __has_include(<opencv2/calib3d.hpp>)
#include <opencv2/calib3d.hpp>
#define ADF_HAVE_CALIB3D 1
#endif
#endif
#endif

namespace adf {

// cv::Exception counterpart: CV_Assert / CV_Error sites of DF.cpp:221-222,262-264, FGS.cpp:143-144,184-189.
struct Exception : std::runtime_error {
    int code;
    Exception(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc != ADF_OK) throw Exception(rc, adf_last_error()); }

#ifdef ADF_HAVE_OPENCV
using Mat = cv::Mat;
using Rect = cv::Rect;
template <class T> using Ptr = cv::Ptr<T>;
enum { D8U = CV_8U, D16S = CV_16S, D32F = CV_32F };
inline int mat_depth(const Mat& m) { return m.depth(); }
inline int mat_channels(const Mat& m) { return m.channels(); }
inline ptrdiff_t mat_step(const Mat& m) { return (ptrdiff_t)m.step; }
inline void mat_create(Mat& m, int rows, int cols, int depth, int cn) { m.create(rows, cols, CV_MAKETYPE(depth, cn)); }
#else
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
    int area() const { return width * height; }
};
enum { D8U = 0, D16S = 3, D32F = 5 }; // cv depth codes
// Minimal dense image: shared buffer, header copies share data (like cv::Mat).
struct Mat {
    int rows = 0, cols = 0, depth_ = D8U, cn = 1;
    size_t step = 0;
    unsigned char* data = nullptr;
    std::shared_ptr<std::vector<unsigned char>> buf;
    Mat() {}
    Mat(int r, int c, int depth, int channels = 1) { create(r, c, depth, channels); }
    static size_t esz(int depth) { return depth == D8U ? 1 : depth == D16S ? 2 : 4; }
    void create(int r, int c, int depth, int channels = 1)
    {
        if (r == rows && c == cols && depth == depth_ && channels == cn && data) return;
        rows = r; cols = c; depth_ = depth; cn = channels;
        step = (size_t)c * channels * esz(depth);
        buf = std::make_shared<std::vector<unsigned char>>(step * (size_t)r);
        data = buf->data();
    }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * (size_t)r); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * (size_t)r); }
};
template <class T> using Ptr = std::shared_ptr<T>;
inline int mat_depth(const Mat& m) { return m.depth_; }
inline int mat_channels(const Mat& m) { return m.cn; }
inline ptrdiff_t mat_step(const Mat& m) { return (ptrdiff_t)m.step; }
inline void mat_create(Mat& m, int rows, int cols, int depth, int cn) { m.create(rows, cols, depth, cn); }
#endif

namespace ximgproc {

// DF.hpp:52-76
class DisparityFilter {
public:
    virtual ~DisparityFilter() {}
    virtual void filter(const Mat& disparity_map_left, const Mat& left_view, Mat& filtered_disparity_map,
                        const Mat& disparity_map_right = Mat(), Rect ROI = Rect(), const Mat& right_view = Mat()) = 0;
};

// DF.hpp:82-122
class DisparityWLSFilter : public DisparityFilter {
public:
    virtual double getLambda() = 0;
    virtual void setLambda(double _lambda) = 0;
    virtual double getSigmaColor() = 0;
    virtual void setSigmaColor(double _sigma_color) = 0;
    virtual int getLRCthresh() = 0;
    virtual void setLRCthresh(int _LRC_thresh) = 0;
    virtual int getDepthDiscontinuityRadius() = 0;
    virtual void setDepthDiscontinuityRadius(int _disc_radius) = 0;
    virtual Mat getConfidenceMap() = 0;
    virtual Rect getROI() = 0;
    // extension: ADF_SOLVER_EXACT (bit-exact scalar order) or ADF_SOLVER_WAVE (on-chip, <= 1 LSB)
    virtual void setSolver(int solver) = 0;
};

class DisparityWLSFilterImpl : public DisparityWLSFilter {
    adf_wls_t* h_ = nullptr;
    bool use_confidence_;
    int last_rows_ = 0, last_cols_ = 0;
public:
    DisparityWLSFilterImpl(bool use_confidence, int l, int r, int t, int b, int min_disp) : use_confidence_(use_confidence)
    {
        check(adf_wls_create(&h_, use_confidence ? 1 : 0, l, r, t, b, min_disp));
    }
    ~DisparityWLSFilterImpl() override { adf_wls_destroy(h_); }
    DisparityWLSFilterImpl(const DisparityWLSFilterImpl&) = delete;
    DisparityWLSFilterImpl& operator=(const DisparityWLSFilterImpl&) = delete;

    void filter(const Mat& dl, const Mat& view, Mat& out, const Mat& dr, Rect ROI, const Mat&) override
    {
        if (dl.empty() || mat_depth(dl) != D16S || mat_channels(dl) != 1)       // DF.cpp:221
            throw Exception(ADF_EBADARG, "disparity_map_left must be a non-empty CV_16SC1 image");
        if (view.empty() || mat_depth(view) != D8U || (mat_channels(view) != 1 && mat_channels(view) != 3)) // :222
            throw Exception(ADF_EBADARG, "left_view must be CV_8UC1 or CV_8UC3");
        const bool have_r = !dr.empty();
        if (use_confidence_) {                                                 // DF.cpp:262-264
            if (!have_r || mat_depth(dr) != D16S || mat_channels(dr) != 1)
                throw Exception(ADF_EBADARG, "disparity_map_right must be a non-empty CV_16SC1 image");
            if (dr.rows != dl.rows || dr.cols != dl.cols)
                throw Exception(ADF_ESIZE, "left and right disparity maps differ in size");
        }
        mat_create(out, view.rows, view.cols, D16S, 1);                        // DF.cpp:252,282 (view-sized)
        adf_rect roi{ROI.x, ROI.y, ROI.width, ROI.height};
        // a lower-resolution disparity map is resized to the view inside the call (DF.cpp:239-247,268-277)
        check(adf_wls_filter_scaled_host(h_, 1, reinterpret_cast<const int16_t*>(dl.data), mat_step(dl), 0, dl.cols, dl.rows,
                                  view.data, mat_step(view), 0, mat_channels(view), view.cols, view.rows,
                                  reinterpret_cast<int16_t*>(out.data), mat_step(out), 0,
                                  have_r ? reinterpret_cast<const int16_t*>(dr.data) : nullptr, have_r ? mat_step(dr) : 0, 0,
                                  ROI.area() != 0 ? &roi : nullptr));
        last_rows_ = view.rows; last_cols_ = view.cols;
    }
    double getLambda() override { double v; check(adf_wls_get_lambda(h_, &v)); return v; }
    void setLambda(double v) override { check(adf_wls_set_lambda(h_, v)); }
    double getSigmaColor() override { double v; check(adf_wls_get_sigma_color(h_, &v)); return v; }
    void setSigmaColor(double v) override { check(adf_wls_set_sigma_color(h_, v)); }
    int getLRCthresh() override { int v; check(adf_wls_get_lrc_thresh(h_, &v)); return v; }
    void setLRCthresh(int v) override { check(adf_wls_set_lrc_thresh(h_, v)); }
    int getDepthDiscontinuityRadius() override { int v; check(adf_wls_get_depth_discontinuity_radius(h_, &v)); return v; }
    void setDepthDiscontinuityRadius(int v) override { check(adf_wls_set_depth_discontinuity_radius(h_, v)); }
    Mat getConfidenceMap() override
    {
        Mat m;                                                                 // empty Mat before the first call (DF.cpp:153)
        if (!use_confidence_ || last_rows_ == 0) return m;
        mat_create(m, last_rows_, last_cols_, D32F, 1);
        check(adf_wls_get_confidence_host(h_, 0, reinterpret_cast<float*>(m.data), mat_step(m)));
        return m;
    }
    Rect getROI() override { adf_rect r; check(adf_wls_get_roi(h_, &r)); return Rect(r.x, r.y, r.width, r.height); }
    void setSolver(int solver) override { check(adf_wls_set_solver(h_, solver)); }
};

// DF.hpp:149, DF.cpp:452-455
inline Ptr<DisparityWLSFilter> createDisparityWLSFilterGeneric(bool use_confidence)
{
    return Ptr<DisparityWLSFilter>(new DisparityWLSFilterImpl(use_confidence, 0, 0, 0, 0, 0));
}

// The arithmetic of createDisparityWLSFilter (DF.cpp:392-409) on the three matcher parameters it reads.
inline Ptr<DisparityWLSFilter> createDisparityWLSFilter(bool is_sgbm, int min_disp, int num_disp, int wsize)
{
    const int wsize2 = wsize / 2;
    Ptr<DisparityWLSFilter> wls;
    if (!is_sgbm) {                                                            // StereoBM, DF.cpp:397-403
        wls = Ptr<DisparityWLSFilter>(new DisparityWLSFilterImpl(true, std::max(0, min_disp + num_disp) + wsize2,
                                                                 std::max(0, -min_disp) + wsize2, wsize2, wsize2, min_disp));
        wls->setDepthDiscontinuityRadius((int)std::ceil(0.33 * wsize));
    } else {                                                                   // StereoSGBM, DF.cpp:404-409
        wls = Ptr<DisparityWLSFilter>(new DisparityWLSFilterImpl(true, std::max(0, min_disp + num_disp),
                                                                 std::max(0, -min_disp), 0, 0, min_disp));
        wls->setDepthDiscontinuityRadius((int)std::ceil(0.5 * wsize));
    }
    return wls;
}

#ifdef ADF_HAVE_CALIB3D
// DF.hpp:131, DF.cpp:386-414 (mutates the matcher exactly like the reference)
inline Ptr<DisparityWLSFilter> createDisparityWLSFilter(cv::Ptr<cv::StereoMatcher> matcher_left)
{
    matcher_left->setDisp12MaxDiff(1000000);
    matcher_left->setSpeckleWindowSize(0);
    const int min_disp = matcher_left->getMinDisparity(), num_disp = matcher_left->getNumDisparities();
    const int wsize = matcher_left->getBlockSize();
    if (cv::Ptr<cv::StereoBM> bm = matcher_left.dynamicCast<cv::StereoBM>()) {
        bm->setTextureThreshold(0);
        bm->setUniquenessRatio(0);
        return createDisparityWLSFilter(false, min_disp, num_disp, wsize);
    }
    if (cv::Ptr<cv::StereoSGBM> sgbm = matcher_left.dynamicCast<cv::StereoSGBM>()) {
        sgbm->setUniquenessRatio(0);
        return createDisparityWLSFilter(true, min_disp, num_disp, wsize);
    }
    throw Exception(ADF_EBADARG, "DisparityWLSFilter natively supports only StereoBM and StereoSGBM");
}

// DF.hpp:139, DF.cpp:417-449
inline cv::Ptr<cv::StereoMatcher> createRightMatcher(cv::Ptr<cv::StereoMatcher> matcher_left)
{
    const int min_disp = matcher_left->getMinDisparity(), num_disp = matcher_left->getNumDisparities();
    const int wsize = matcher_left->getBlockSize();
    if (cv::Ptr<cv::StereoBM> bm = matcher_left.dynamicCast<cv::StereoBM>()) {
        cv::Ptr<cv::StereoBM> right_bm = cv::StereoBM::create(num_disp, wsize);
        right_bm->setMinDisparity(-(min_disp + num_disp) + 1);
        right_bm->setTextureThreshold(0);
        right_bm->setUniquenessRatio(0);
        right_bm->setDisp12MaxDiff(1000000);
        right_bm->setSpeckleWindowSize(0);
        return right_bm;
    }
    if (cv::Ptr<cv::StereoSGBM> sgbm = matcher_left.dynamicCast<cv::StereoSGBM>()) {
        cv::Ptr<cv::StereoSGBM> right_sgbm = cv::StereoSGBM::create(-(min_disp + num_disp) + 1, num_disp, wsize);
        right_sgbm->setUniquenessRatio(0);
        right_sgbm->setP1(sgbm->getP1());
        right_sgbm->setP2(sgbm->getP2());
        right_sgbm->setMode(sgbm->getMode());
        right_sgbm->setPreFilterCap(sgbm->getPreFilterCap());
        right_sgbm->setDisp12MaxDiff(1000000);
        right_sgbm->setSpeckleWindowSize(0);
        return right_sgbm;
    }
    throw Exception(ADF_EBADARG, "createRightMatcher supports only StereoBM and StereoSGBM");
}
#endif

// ---------------------------------------------------------------------------------------------------------
// Block matcher feeding the filter.  cv::StereoBM is a class of OpenCV's calib3d, outside the
// reference tree (parity unpinned there); this class carries its accessor names over adf_bm_* so that the
// sample's pipeline (samples/disparity_filtering.cpp:151-189) reads the same.  Host images in, host image out;
// device-resident pipelines call adf_bm_compute_device directly.
// ---------------------------------------------------------------------------------------------------------
class StereoBM {
    adf_bm_t* h_ = nullptr;
    int min_disp_ = 0, num_disp_, block_, cap_ = 31, texture_ = 10, uniq_ = 15;   // cv::StereoBM's defaults
    int disp12_ = -1, speckle_window_ = 0;
public:
    StereoBM(int numDisparities, int blockSize) : num_disp_(numDisparities > 0 ? numDisparities : 64), block_(blockSize)
    {
        check(adf_bm_create(&h_, num_disp_, block_));
    }
    ~StereoBM() { adf_bm_destroy(h_); }
    StereoBM(const StereoBM&) = delete;
    StereoBM& operator=(const StereoBM&) = delete;
    static Ptr<StereoBM> create(int numDisparities = 0, int blockSize = 21) { return Ptr<StereoBM>(new StereoBM(numDisparities, blockSize)); }
    int getMinDisparity() const { return min_disp_; }        void setMinDisparity(int v) { min_disp_ = v; }
    int getNumDisparities() const { return num_disp_; }      void setNumDisparities(int v) { num_disp_ = v; }
    int getBlockSize() const { return block_; }              void setBlockSize(int v) { block_ = v; }
    int getPreFilterCap() const { return cap_; }             void setPreFilterCap(int v) { cap_ = v; }
    int getTextureThreshold() const { return texture_; }     void setTextureThreshold(int v) { texture_ = v; }
    int getUniquenessRatio() const { return uniq_; }         void setUniquenessRatio(int v) { uniq_ = v; }
    int getDisp12MaxDiff() const { return disp12_; }         void setDisp12MaxDiff(int v) { disp12_ = v; }
    int getSpeckleWindowSize() const { return speckle_window_; } void setSpeckleWindowSize(int v) { speckle_window_ = v; }
    // StereoMatcher::compute: CV_8UC1 views -> CV_16SC1 disparity * 16, rejected pixels (minDisparity - 1) * 16
    void compute(const Mat& left, const Mat& right, Mat& disparity)
    {
        if (left.empty() || right.empty() || mat_depth(left) != D8U || mat_depth(right) != D8U ||
            mat_channels(left) != 1 || mat_channels(right) != 1)
            throw Exception(ADF_EBADARG, "Both input images must have CV_8UC1");
        if (left.rows != right.rows || left.cols != right.cols)
            throw Exception(ADF_ESIZE, "All the images must have the same size");
        if ((disp12_ >= 0 && disp12_ < 1000000) || speckle_window_ > 0)          // the filter factory switches both off (DF.cpp:389-390)
            throw Exception(ADF_EBADARG, "the matcher's own left-right check and speckle filter are not implemented");
        check(adf_bm_set_params(h_, min_disp_, num_disp_, block_, cap_, texture_, uniq_));
        Mat out;
        mat_create(out, left.rows, left.cols, D16S, 1);
        check(adf_bm_compute_host(h_, 1, left.data, mat_step(left), 0, right.data, mat_step(right), 0, left.cols, left.rows,
                                  reinterpret_cast<int16_t*>(out.data), mat_step(out), 0));
        disparity = out;
    }
};

// DF.cpp:386-403 (StereoBM branch): mutates the matcher exactly like the reference, derives ROI offsets and radius
inline Ptr<DisparityWLSFilter> createDisparityWLSFilter(const Ptr<StereoBM>& matcher_left)
{
    matcher_left->setDisp12MaxDiff(1000000);
    matcher_left->setSpeckleWindowSize(0);
    matcher_left->setTextureThreshold(0);
    matcher_left->setUniquenessRatio(0);
    return createDisparityWLSFilter(false, matcher_left->getMinDisparity(), matcher_left->getNumDisparities(),
                                    matcher_left->getBlockSize());
}

// DF.cpp:417-431
inline Ptr<StereoBM> createRightMatcher(const Ptr<StereoBM>& matcher_left)
{
    const int min_disp = matcher_left->getMinDisparity(), num_disp = matcher_left->getNumDisparities();
    Ptr<StereoBM> right_bm = StereoBM::create(num_disp, matcher_left->getBlockSize());
    right_bm->setMinDisparity(-(min_disp + num_disp) + 1);
    right_bm->setTextureThreshold(0);
    right_bm->setUniquenessRatio(0);
    right_bm->setDisp12MaxDiff(1000000);
    right_bm->setSpeckleWindowSize(0);
    // (the reference's BM branch does not copy preFilterCap: the right matcher keeps cv::StereoBM's default 31)
    return right_bm;
}

// ---------------------------------------------------------------------------------------------------------
// Semi-global matcher feeding the filter: cv::StereoSGBM's accessor names over adf_sgbm_*, the
// sample's other producer (samples/disparity_filtering.cpp:166-176).  MODE_SGBM_3WAY (the sample's), MODE_SGBM, MODE_HH.
// ---------------------------------------------------------------------------------------------------------
class StereoSGBM {
    adf_sgbm_t* h_ = nullptr;
    int min_disp_, num_disp_, block_, P1_ = 0, P2_ = 0, cap_ = 0, uniq_ = 0, mode_ = MODE_SGBM;    // cv::StereoSGBM::create's defaults
    int disp12_ = 0, speckle_window_ = 0;
public:
    enum { MODE_SGBM = ADF_SGBM_MODE_SGBM, MODE_HH = ADF_SGBM_MODE_HH, MODE_SGBM_3WAY = ADF_SGBM_MODE_3WAY };
    StereoSGBM(int minDisparity, int numDisparities, int blockSize) : min_disp_(minDisparity), num_disp_(numDisparities), block_(blockSize)
    {
        check(adf_sgbm_create(&h_, min_disp_, num_disp_, block_));
    }
    ~StereoSGBM() { adf_sgbm_destroy(h_); }
    StereoSGBM(const StereoSGBM&) = delete;
    StereoSGBM& operator=(const StereoSGBM&) = delete;
    static Ptr<StereoSGBM> create(int minDisparity = 0, int numDisparities = 16, int blockSize = 3)
    {
        return Ptr<StereoSGBM>(new StereoSGBM(minDisparity, numDisparities, blockSize));
    }
    int getMinDisparity() const { return min_disp_; }        void setMinDisparity(int v) { min_disp_ = v; }
    int getNumDisparities() const { return num_disp_; }      void setNumDisparities(int v) { num_disp_ = v; }
    int getBlockSize() const { return block_; }              void setBlockSize(int v) { block_ = v; }
    int getP1() const { return P1_; }                        void setP1(int v) { P1_ = v; }
    int getP2() const { return P2_; }                        void setP2(int v) { P2_ = v; }
    int getPreFilterCap() const { return cap_; }             void setPreFilterCap(int v) { cap_ = v; }
    int getUniquenessRatio() const { return uniq_; }         void setUniquenessRatio(int v) { uniq_ = v; }
    int getMode() const { return mode_; }                    void setMode(int v) { mode_ = v; }
    int getDisp12MaxDiff() const { return disp12_; }         void setDisp12MaxDiff(int v) { disp12_ = v; }
    int getSpeckleWindowSize() const { return speckle_window_; } void setSpeckleWindowSize(int v) { speckle_window_ = v; }
    // StereoMatcher::compute: CV_8UC1 / CV_8UC3 views -> CV_16SC1 disparity * 16, invalid pixels (minDisparity - 1) * 16
    void compute(const Mat& left, const Mat& right, Mat& disparity)
    {
        if (left.empty() || right.empty() || mat_depth(left) != D8U || mat_depth(right) != D8U ||
            (mat_channels(left) != 1 && mat_channels(left) != 3) || mat_channels(left) != mat_channels(right))
            throw Exception(ADF_EBADARG, "Both input images must have CV_8UC1 or CV_8UC3");
        if (left.rows != right.rows || left.cols != right.cols)
            throw Exception(ADF_ESIZE, "All the images must have the same size");
        if (speckle_window_ > 0)                                                 // the filter factory switches it off (DF.cpp:390)
            throw Exception(ADF_EBADARG, "the matcher's speckle filter is not implemented");
        check(adf_sgbm_set_params(h_, min_disp_, num_disp_, block_, P1_, P2_, cap_, uniq_, mode_));
        check(adf_sgbm_set_disp12_max_diff(h_, disp12_));
        Mat out;
        mat_create(out, left.rows, left.cols, D16S, 1);
        check(adf_sgbm_compute_host(h_, 1, left.data, mat_step(left), 0, right.data, mat_step(right), 0, mat_channels(left),
                                    left.cols, left.rows, reinterpret_cast<int16_t*>(out.data), mat_step(out), 0));
        disparity = out;
    }
};

// DF.cpp:386-391, 404-409 (StereoSGBM branch)
inline Ptr<DisparityWLSFilter> createDisparityWLSFilter(const Ptr<StereoSGBM>& matcher_left)
{
    matcher_left->setDisp12MaxDiff(1000000);
    matcher_left->setSpeckleWindowSize(0);
    matcher_left->setUniquenessRatio(0);
    return createDisparityWLSFilter(true, matcher_left->getMinDisparity(), matcher_left->getNumDisparities(),
                                    matcher_left->getBlockSize());
}

// DF.cpp:432-445
inline Ptr<StereoSGBM> createRightMatcher(const Ptr<StereoSGBM>& matcher_left)
{
    const int min_disp = matcher_left->getMinDisparity(), num_disp = matcher_left->getNumDisparities();
    Ptr<StereoSGBM> right_sgbm = StereoSGBM::create(-(min_disp + num_disp) + 1, num_disp, matcher_left->getBlockSize());
    right_sgbm->setUniquenessRatio(0);
    right_sgbm->setP1(matcher_left->getP1());
    right_sgbm->setP2(matcher_left->getP2());
    right_sgbm->setMode(matcher_left->getMode());
    right_sgbm->setPreFilterCap(matcher_left->getPreFilterCap());
    right_sgbm->setDisp12MaxDiff(1000000);
    right_sgbm->setSpeckleWindowSize(0);
    return right_sgbm;
}

// EF.hpp:361-371
class FastGlobalSmootherFilter {
    adf_fgs_t* h_ = nullptr;
    int rows_, cols_;
public:
    FastGlobalSmootherFilter(const Mat& guide, double lambda, double sigma_color, double lambda_attenuation, int num_iter,
                             int solver = ADF_SOLVER_WAVE)
        : rows_(guide.rows), cols_(guide.cols)
    {
        if (guide.empty() || mat_depth(guide) != D8U)                          // FGS.cpp:143-144
            throw Exception(ADF_EBADARG, "guide must be a non-empty CV_8UC1 / CV_8UC3 image");
        check(adf_fgs_create(&h_, guide.data, mat_step(guide), mat_channels(guide), guide.cols, guide.rows, lambda,
                             sigma_color, lambda_attenuation, num_iter, solver));
    }
    ~FastGlobalSmootherFilter() { adf_fgs_destroy(h_); }
    FastGlobalSmootherFilter(const FastGlobalSmootherFilter&) = delete;
    FastGlobalSmootherFilter& operator=(const FastGlobalSmootherFilter&) = delete;
    void filter(const Mat& src, Mat& dst)                                      // EF.hpp:370, FGS.cpp:182-233
    {
        if (src.empty()) throw Exception(ADF_EBADARG, "src is empty");
        if (src.rows != rows_ || src.cols != cols_)                            // FGS.cpp:185-189
            throw Exception(ADF_ESIZE, "Size of the filtered image must be equal to the size of the guide image");
        Mat out;
        mat_create(out, src.rows, src.cols, mat_depth(src), mat_channels(src));
        check(adf_fgs_filter_host(h_, src.data, mat_step(src), out.data, mat_step(out), mat_depth(src), mat_channels(src)));
        dst = out;
    }
};

// EF.hpp:393
inline Ptr<FastGlobalSmootherFilter> createFastGlobalSmootherFilter(const Mat& guide, double lambda, double sigma_color,
                                                                    double lambda_attenuation = 0.25, int num_iter = 3)
{
    return Ptr<FastGlobalSmootherFilter>(new FastGlobalSmootherFilter(guide, lambda, sigma_color, lambda_attenuation, num_iter));
}

// EF.hpp:413
inline void fastGlobalSmootherFilter(const Mat& guide, const Mat& src, Mat& dst, double lambda, double sigma_color,
                                     double lambda_attenuation = 0.25, int num_iter = 3)
{
    createFastGlobalSmootherFilter(guide, lambda, sigma_color, lambda_attenuation, num_iter)->filter(src, dst);
}

} // namespace ximgproc
} // namespace adf
