/*
 * adf_oracle.h -- CPU restatement of the DisparityWLSFilter hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference lines it restates.  Shorthand:
 *   DF.cpp  = modules/ximgproc/src/disparity_filters.cpp
 *   FGS.cpp = modules/ximgproc/src/fgs_filter.cpp
 *   EF.hpp  = modules/ximgproc/include/opencv2/ximgproc/edge_filter.hpp
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - the reference cannot be built here (needs OpenCV core/imgproc headers
 *     and libraries that the image lacks), and its golden images live in
 *     opencv_extra which is not in the tree;
 *   - the oracle is pinned by the known-answer / invariant tests the
 *     reference itself holds for this path (T_FGS:59-87 constant-surface,
 *     T_DF:99-153 / T_FGS:109-151 order-reproducibility at <=1 LSB) and by
 *     an independent float64 banded solve (oracle/banded_f64.py);
 *   - the OpenCV-imgproc boundary (boxFilter / sqrBoxFilter normalisation,
 *     resize) has no in-tree known answer: "parity unpinned" there.
 */
#ifndef ADF_ORACLE_H
#define ADF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Floating-point evaluation order of the Thomas denominator.
 * SCALAR   : ((1-cp)-cc) - D*cp          FGS.cpp:455 (process_row), :369, :553
 * REF_SIMD : H pass (1-(cc+cp)) - D*cp   FGS.cpp:305-310 inside 4-row blocks,
 *            V pass 1-((cp+cc)+D*cp)     FGS.cpp:526-534 inside 4-column groups,
 *            scalar order for the leftovers, exactly as the reference's
 *            CV_SIMD128 build arranges them per stripe. */
enum { ADF_ORDER_SCALAR = 0, ADF_ORDER_REF_SIMD = 1 };

/* Source depths accepted by FastGlobalSmootherFilter::filter (FGS.cpp:184). */
enum { ADF_DEPTH_8U = 0, ADF_DEPTH_16S = 3, ADF_DEPTH_32F = 5 };

typedef struct adf_oracle_params {
    double lambda;             /* DF.cpp:150, default 8000 (DF.cpp:215)      */
    double sigma_color;        /* DF.cpp:151, default 1.0  (DF.cpp:215)      */
    int    use_confidence;     /* DF.cpp:152                                 */
    int    lrc_thresh;         /* DF.cpp:154, default 24                     */
    int    disc_radius;        /* DF.cpp:155, default 5                      */
    int    num_iter;           /* EF.hpp:393 default 3 (DF.cpp:292 call)     */
    double lambda_attenuation; /* EF.hpp:393 default 0.25                    */
    int    order;              /* ADF_ORDER_*                                */
    int    threads;            /* num_stripes = getNumThreads() DF.cpp:158   */
} adf_oracle_params;

void adf_oracle_default_params(adf_oracle_params* p);

/* FGS.cpp:663-675  LUT[i] = -expf(-sqrtf(i)/sigma), i in [0, 3*256*256). */
void adf_oracle_lut(float sigma, float* lut);

/* FGS.cpp:586-661  horizontal / vertical edge weights of a guide view.
 * guide: h rows of w pixels, `ch` interleaved uint8 channels, row stride in
 * bytes.  chor / cvert: dense h*w float. */
void adf_oracle_weights(const uint8_t* guide, ptrdiff_t stride, int ch,
                        int w, int h, const float* lut,
                        float* chor, float* cvert, int threads);

/* FGS.cpp:235-584  one horizontal / vertical pass, in place on `cur`
 * (dense h*w), scratch interD (dense h*w). */
void adf_oracle_hpass(float* cur, const float* chor, float* interD,
                      int w, int h, float lambda, int order, int threads);
void adf_oracle_vpass(float* cur, const float* cvert, float* interD,
                      int w, int h, float lambda, int order, int threads);

/* FGS.cpp:141-233  createFastGlobalSmootherFilter + filter on float planes.
 * planes: nplanes dense h*w float images filtered in place with one shared
 * set of weights (what DF.cpp:292-294 does with its two sources). */
int adf_oracle_fgs_planes(const uint8_t* guide, ptrdiff_t stride, int ch,
                          int w, int h, float* planes, int nplanes,
                          double lambda, double sigma_color,
                          double lambda_attenuation, int num_iter,
                          int order, int threads);

/* FGS.cpp:182-233, :687-691  generic fastGlobalSmootherFilter: src/dst of
 * depth 8U / 16S / 32F with 1..4 interleaved channels, dense rows. */
int adf_oracle_fgs_filter(const uint8_t* guide, ptrdiff_t gstride, int gch,
                          int w, int h, const void* src, void* dst,
                          int depth, int channels,
                          double lambda, double sigma_color,
                          double lambda_attenuation, int num_iter,
                          int order, int threads);

/* DF.cpp:161-194 + :343-373  depth-discontinuity map of one view.
 * disp: H rows x W int16 (stride bytes); ROI (rx,ry,rw,rh); dst dense H*W
 * float, zero outside the ROI.  roll_off = 0.001f/resize_factor^2. */
void adf_oracle_discontinuity(const int16_t* disp, ptrdiff_t stride,
                              int W, int H, int rx, int ry, int rw, int rh,
                              int radius, float roll_off, float* dst,
                              int threads);

/* DF.cpp:197-210 + :306-341  confidence map (already multiplied by 255).
 * conf: dense H*W float. */
void adf_oracle_confidence(const int16_t* dispL, ptrdiff_t strideL,
                           const int16_t* dispR, ptrdiff_t strideR,
                           int W, int H, int rx, int ry, int rw, int rh,
                           int radius, int lrc_thresh, float resize_factor,
                           float* conf, int threads);

/* DF.cpp:219-298  DisparityWLSFilterImpl::filter, same-size disparity/view.
 * ROI with rw*rh == 0 is rejected (the caller resolves offsets, DF.cpp:228-233).
 * out: H rows x W int16 (stride bytes).  conf_out: nullable dense H*W float
 * (getConfidenceMap, DF.cpp:138). */
int adf_oracle_wls_filter(const adf_oracle_params* p,
                          const int16_t* dispL, ptrdiff_t strideL,
                          const uint8_t* guide, ptrdiff_t strideG, int gch,
                          int W, int H,
                          const int16_t* dispR, ptrdiff_t strideR,
                          int rx, int ry, int rw, int rh,
                          int16_t* out, ptrdiff_t strideO, float* conf_out);

/* DF.cpp:497-517 computeMSE, :519-539 computeBadPixelPercent, :541-556 getDisparityVis (dense rows). */
double adf_oracle_compute_mse(const int16_t* gt, const int16_t* src, int W, int H, int rx, int ry, int rw, int rh);
double adf_oracle_bad_pixel_percent(const int16_t* gt, const int16_t* src, int W, int H, int rx, int ry, int rw, int rh, int thresh);
void adf_oracle_disparity_vis(const int16_t* src, uint8_t* dst, int W, int H, double scale);

/* cv::resize(INTER_LINEAR) restated for CV_16SC1 (optionally followed by the saturating *x_ratio of
 * DF.cpp:244,273) and CV_32FC1; dense rows.  OpenCV-imgproc boundary: parity unpinned. */
void adf_oracle_resize_linear_16s(const int16_t* src, int sw, int sh, int16_t* dst, int dw, int dh, float post_scale);
void adf_oracle_resize_linear_32f(const float* src, int sw, int sh, float* dst, int dw, int dh);
/* DF.cpp:219-298 with disparity maps (dW x dH, dense) smaller than the view (W x H, dense); ROI in
 * disparity-map coordinates; conf_out: nullable view-sized confidence map. */
int adf_oracle_wls_filter_scaled(const adf_oracle_params* p, const int16_t* dispL, const int16_t* dispR,
                                 int dW, int dH, const uint8_t* guide, int gch, int W, int H,
                                 int rx, int ry, int rw, int rh, int16_t* out, float* conf_out);

/* saturate_cast<short>(float) = cvRound + clamp (DF.cpp:296, FGS.cpp:216);
 * exported so tests can probe the rounding convention directly. */
int16_t adf_oracle_sat16(float v);
/* test hook: ADF_ORDER_REF_SIMD one row / one column at a time (scalar emulation) instead of on 128-bit vectors */
void adf_oracle_set_refsimd_rowwise(int on);

/* ---- block matcher feeding the filter (adf_oracle_bm.c) ----
 * cv::StereoBM is external to the reference (calib3d, unpinned): parity unpinned; see adf_oracle_bm.c. */
typedef struct adf_oracle_bm_params {
    int min_disparity;      /* StereoMatcher::setMinDisparity (right matcher: disparity_filters.cpp:424) */
    int num_disparities;    /* multiple of 16 */
    int block_size;         /* odd, 5..21 */
    int prefilter_cap;      /* 1..63, StereoBM default 31 */
    int texture_threshold;  /* forced to 0 by the filter factory (disparity_filters.cpp:399) */
    int uniqueness_ratio;   /* forced to 0 by the filter factory (disparity_filters.cpp:400) */
} adf_oracle_bm_params;
void adf_oracle_bm_prefilter_xsobel(const uint8_t* src, ptrdiff_t stride, int W, int H, int cap, uint8_t* dst);
/* left/right: CV_8UC1 W x H (strides in bytes); disp: CV_16SC1, stride in ELEMENTS. */
int adf_oracle_bm_compute(const adf_oracle_bm_params* p, const uint8_t* left, ptrdiff_t lstride,
                          const uint8_t* right, ptrdiff_t rstride, int W, int H,
                          int16_t* disp, ptrdiff_t dstride);

/* ---- semi-global matcher feeding the filter (adf_oracle_sgbm.c) ----
 * cv::StereoSGBM is external to the reference (calib3d, unpinned): parity unpinned; see adf_oracle_sgbm.c. */
#define ADF_SGBM_MODE_SGBM 0
#define ADF_SGBM_MODE_HH 1
#define ADF_SGBM_MODE_3WAY 2   /* StereoSGBM::MODE_SGBM_3WAY, the sample's mode (samples/disparity_filtering.cpp:170) */
#define ADF_SGBM_MODE_3WAY_GENERIC 3   /* test hook: the three paths of MODE_3WAY through the general multi-path code */
typedef struct adf_oracle_sgbm_params {
    int min_disparity;      /* right matcher: -(min+num)+1, disparity_filters.cpp:435 */
    int num_disparities;    /* multiple of 16 */
    int block_size;         /* odd; 0 -> 5 */
    int P1, P2;             /* sample: 24*w*w, 96*w*w; 0 -> 2 / 5; P2 >= P1+1 */
    int prefilter_cap;      /* sample: 63 */
    int uniqueness_ratio;   /* forced to 0 by the filter factory (disparity_filters.cpp:406,436); < 0 -> 10 */
    int mode;               /* ADF_SGBM_MODE_3WAY (3 paths), ADF_SGBM_MODE_SGBM (5), ADF_SGBM_MODE_HH (8) */
    int disp12_max_diff;    /* the matcher's own left-right check; <= 0 -> 1; the filter factory sets 1000000 (off) */
} adf_oracle_sgbm_params;
/* (value, min, max over the half-sample neighbours) of every signal of every pixel: rec[H][W][2cn][3] */
void adf_oracle_sgbm_signals(const uint8_t* img, ptrdiff_t stride, int cn, int W, int H, int prefilter_cap, uint8_t* rec);
/* block costs C[H][width1][D] of the whole image (width1 = matchable columns); small images, tests only */
int adf_oracle_sgbm_block_costs(const adf_oracle_sgbm_params* p, const uint8_t* img1, ptrdiff_t s1, const uint8_t* img2,
                                ptrdiff_t s2, int cn, int W, int H, int16_t* C);
/* img1/img2: CV_8UC1 / CV_8UC3 W x H (strides in bytes); disp: CV_16SC1, stride in ELEMENTS; raw (nullable):
 * the map before the 3x3 median, dense W x H. */
int adf_oracle_sgbm_compute(const adf_oracle_sgbm_params* p, const uint8_t* img1, ptrdiff_t s1, const uint8_t* img2,
                            ptrdiff_t s2, int cn, int W, int H, int16_t* disp, ptrdiff_t dstride, int16_t* raw);
void adf_oracle_median3_16s(const int16_t* src, ptrdiff_t sstride, int16_t* dst, ptrdiff_t dstride, int W, int H);

#ifdef __cplusplus
}
#endif
#endif
