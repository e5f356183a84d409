"""MI355X-native DisparityWLSFilter / FastGlobalSmootherFilter.

The package mirrors cv::ximgproc's interface for this one path and calls hand-written HIP kernels
through the C-ABI of include/adf_wls.h.  Importing the API does not load the library; the first
filter construction does, and fails loudly if libadf_wls.so has not been built.
"""
from .ximgproc import (  # noqa: F401
    AdfError,
    DisparityFilter,
    DisparityWLSFilter,
    FastGlobalSmootherFilter,
    PATH_CONF_BAND,
    PATH_FUSED_FIRST_PASS,
    PATH_MERGED_PREP,
    PATH_SCALED_FUSED,
    PATH_SCALED_HALF,
    SOLVER_EXACT,
    SOLVER_WAVE,
    StereoBM,
    StereoMatcher,
    StereoSGBM,
    UNKNOWN_DISPARITY,
    computeBadPixelPercent,
    computeMSE,
    createDisparityWLSFilter,
    createDisparityWLSFilterGeneric,
    createFastGlobalSmootherFilter,
    createRightMatcher,
    fastGlobalSmootherFilter,
    releaseCachedMemory,
    getDisparityVis,
    readGT,
)

__version__ = "0.1.0"
