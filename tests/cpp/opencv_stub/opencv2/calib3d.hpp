// COMPILE-ONLY declaration stub (see core.hpp in this directory): the cv::StereoMatcher / StereoBM / StereoSGBM
// accessors the matcher factories of include/adf_ximgproc.hpp call (reference: disparity_filters.cpp:386-449).
// Signatures follow the public calib3d API as documented (opencv2/calib3d.hpp).
#pragma once
#include "core.hpp"

namespace cv {

class StereoMatcher {
public:
    virtual ~StereoMatcher();
    virtual int getMinDisparity() const = 0;
    virtual void setMinDisparity(int minDisparity) = 0;
    virtual int getNumDisparities() const = 0;
    virtual void setNumDisparities(int numDisparities) = 0;
    virtual int getBlockSize() const = 0;
    virtual void setBlockSize(int blockSize) = 0;
    virtual int getSpeckleWindowSize() const = 0;
    virtual void setSpeckleWindowSize(int speckleWindowSize) = 0;
    virtual int getDisp12MaxDiff() const = 0;
    virtual void setDisp12MaxDiff(int disp12MaxDiff) = 0;
};

class StereoBM : public StereoMatcher {
public:
    virtual int getPreFilterCap() const = 0;
    virtual void setPreFilterCap(int preFilterCap) = 0;
    virtual int getTextureThreshold() const = 0;
    virtual void setTextureThreshold(int textureThreshold) = 0;
    virtual int getUniquenessRatio() const = 0;
    virtual void setUniquenessRatio(int uniquenessRatio) = 0;
    static Ptr<StereoBM> create(int numDisparities = 0, int blockSize = 21);
};

class StereoSGBM : public StereoMatcher {
public:
    virtual int getPreFilterCap() const = 0;
    virtual void setPreFilterCap(int preFilterCap) = 0;
    virtual int getUniquenessRatio() const = 0;
    virtual void setUniquenessRatio(int uniquenessRatio) = 0;
    virtual int getP1() const = 0;
    virtual void setP1(int P1) = 0;
    virtual int getP2() const = 0;
    virtual void setP2(int P2) = 0;
    virtual int getMode() const = 0;
    virtual void setMode(int mode) = 0;
    static Ptr<StereoSGBM> create(int minDisparity = 0, int numDisparities = 16, int blockSize = 3, int P1 = 0, int P2 = 0,
                                  int disp12MaxDiff = 0, int preFilterCap = 0, int uniquenessRatio = 0,
                                  int speckleWindowSize = 0, int speckleRange = 0, int mode = 0);
};

} // namespace cv
