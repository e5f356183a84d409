// Type-check of the cv::Mat branch of include/adf_ximgproc.hpp against the declaration stubs in opencv_stub/
// (g++ -fsyntax-only; never linked).  Written like a user of the reference's API would write it
// (samples/disparity_filtering.cpp:151-189, 214-253).
#include "adf_ximgproc.hpp"

#if !defined(ADF_HAVE_OPENCV) || !defined(ADF_HAVE_CALIB3D)
#error "the OpenCV branch was not selected: check the include path of the stub"
#endif

using namespace adf::ximgproc;

void pipeline(const cv::Mat& left, const cv::Mat& right, cv::Mat& filtered, cv::Mat& conf, cv::Rect& roi)
{
    cv::Ptr<cv::StereoBM> left_matcher = cv::StereoBM::create(160, 15);
    cv::Ptr<DisparityWLSFilter> wls = createDisparityWLSFilter(left_matcher);
    cv::Ptr<cv::StereoMatcher> right_matcher = createRightMatcher(left_matcher);
    cv::Mat dl, dr;
    wls->setLambda(8000.0);
    wls->setSigmaColor(1.5);
    wls->filter(dl, left, filtered, dr);
    wls->filter(dl, left, filtered, dr, cv::Rect(160, 0, 1760, 1080), right);
    conf = wls->getConfidenceMap();
    roi = wls->getROI();

    cv::Ptr<cv::StereoSGBM> sgbm = cv::StereoSGBM::create(0, 160, 3);
    cv::Ptr<DisparityWLSFilter> wls2 = createDisparityWLSFilter(sgbm);
    cv::Ptr<cv::StereoMatcher> rm2 = createRightMatcher(sgbm);
    cv::Ptr<DisparityWLSFilter> wls3 = createDisparityWLSFilterGeneric(false);
    wls3->filter(dl, left, filtered);

    // this library's own device matcher through the same Mat type
    adf::Ptr<StereoBM> bm = StereoBM::create(64, 9);
    adf::Ptr<DisparityWLSFilter> wls4 = createDisparityWLSFilter(bm);
    adf::Ptr<StereoBM> rbm = createRightMatcher(bm);
    bm->compute(left, right, dl);
    rbm->compute(right, left, dr);

    adf::Ptr<StereoSGBM> sg = StereoSGBM::create(0, 160, 3);
    sg->setP1(24 * 9); sg->setP2(96 * 9); sg->setPreFilterCap(63); sg->setMode(StereoSGBM::MODE_SGBM_3WAY);
    adf::Ptr<DisparityWLSFilter> wls5 = createDisparityWLSFilter(sg);
    adf::Ptr<StereoSGBM> rsg = createRightMatcher(sg);
    sg->compute(left, right, dl);
    rsg->compute(right, left, dr);

    cv::Mat smooth;
    fastGlobalSmootherFilter(left, dl, smooth, 500.0, 1.5);
    createFastGlobalSmootherFilter(left, 500.0, 1.5, 0.25, 3)->filter(dl, smooth);
}
