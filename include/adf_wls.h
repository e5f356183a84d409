/*
 * adf_wls.h -- C-ABI of the MI355X-native DisparityWLSFilter / FastGlobalSmoother path.
 *
 * This is the drop-in boundary: plain pointers, sizes and strides, no C++ and no
 * torch types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference tree):
 *   DF.hpp  = modules/ximgproc/include/opencv2/ximgproc/disparity_filter.hpp
 *   DF.cpp  = modules/ximgproc/src/disparity_filters.cpp
 *   EF.hpp  = modules/ximgproc/include/opencv2/ximgproc/edge_filter.hpp
 *   FGS.cpp = modules/ximgproc/src/fgs_filter.cpp
 *
 * Conventions
 *   - all strides are in BYTES; images are row-major, channels interleaved;
 *   - "_device" entry points take HIP device pointers and run asynchronously on
 *     `stream` (a hipStream_t passed as void*, NULL = default stream);
 *     "_host" entry points take host pointers, copy, run and synchronise;
 *   - a handle owns a device workspace that is sized on first use and reused; a
 *     handle serves one caller at a time (the reference objects are not
 *     re-entrant either: DF.cpp:224-233, FGS.cpp:202-223);
 *   - a handle is bound to the HIP device that was current when it was created
 *     (adf_*_get_device) and to ONE stream at a time: every call on a handle
 *     stages through the same workspace, so two calls on different streams must
 *     be ordered by the caller (an event, or adf_wls_sync) -- nothing inside the
 *     library serialises them.  Device pointers must belong to the handle's
 *     device.  Concurrency = several handles, each on its own stream;
 *   - every function returns ADF_OK or an error code; adf_last_error() gives the
 *     message of the calling thread's last failure (the reference throws
 *     cv::Exception from CV_Assert / CV_Error: DF.cpp:221-222,262-264,
 *     FGS.cpp:143-144,184-189 -- a C++ adaptor turns codes back into exceptions).
 */
#ifndef ADF_WLS_H
#define ADF_WLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADF_VERSION 100

enum adf_status {
    ADF_OK = 0,
    ADF_EBADARG = 1, /* CV_Assert on types / emptiness            DF.cpp:221-222,262 */
    ADF_ESIZE = 2,   /* size mismatch (Error::StsBadSize)         DF.cpp:263-264, FGS.cpp:185-189 */
    ADF_EHIP = 3,    /* a HIP runtime call failed                                       */
    ADF_ENOMEM = 4,  /* workspace allocation failed                                     */
    ADF_ENODEV = 5   /* no usable gfx950 device                                         */
};

/* Thomas-solve strategy (no counterpart in the reference, which picks its own
 * floating-point order per stripe: FGS.cpp:466-476).
 *   ADF_SOLVER_EXACT : one lane per scanline, canonical scalar order of
 *                      process_row (FGS.cpp:439-464); bit-identical to the CPU
 *                      restatement in oracle/.
 *   ADF_SOLVER_WAVE  : one wavefront per scanline, partitioned solve held in
 *                      registers/LDS; re-associated arithmetic, within the
 *                      reference's own reproducibility tolerance (<=1 LSB of the
 *                      CV_16S output, test_disparity_wls_filter.cpp:104-105).
 * A new handle uses ADF_SOLVER_WAVE (13x lower latency on a single 1920x1080 pair, 1.8x the
 * throughput on 4K batches); the confidence map is bit-exact with either.  Select
 * ADF_SOLVER_EXACT when the filtered map has to reproduce the scalar evaluation order bit for bit. */
enum adf_solver { ADF_SOLVER_EXACT = 0, ADF_SOLVER_WAVE = 1 };

/* cv::Mat depth codes for adf_fgs_filter_* (FGS.cpp:184). */
enum adf_depth { ADF_8U = 0, ADF_16S = 3, ADF_32F = 5 };

typedef struct adf_wls adf_wls_t; /* cv::Ptr<DisparityWLSFilter>        */
typedef struct adf_fgs adf_fgs_t; /* cv::Ptr<FastGlobalSmootherFilter>  */

typedef struct adf_rect { int x, y, width, height; } adf_rect; /* cv::Rect */

int adf_version(void);
const char* adf_last_error(void);
/* Number of visible HIP devices (0 when none); never fails. */
int adf_device_count(void);
/* PCI bus id ("0000:c1:00.0") of HIP device `device` into buf (at least 16 bytes): lets a multi-process harness show
 * that its ranks sit on distinct devices.  No counterpart in the reference (it has no device layer). */
int adf_device_pci_bus_id(int device, char* buf, int len);

/* ---------------- DisparityWLSFilter ---------------- */

/* DisparityWLSFilterImpl::create + init (DF.cpp:142-159, 212-217), reached through
 * createDisparityWLSFilterGeneric(use_confidence) (DF.hpp:149, DF.cpp:452-455; all
 * offsets 0) or createDisparityWLSFilter(matcher) (DF.hpp:131, DF.cpp:386-414; the
 * caller derives the offsets from the matcher).
 * Defaults as the reference: lambda 8000, sigma_color 1.0, LRC_thresh 24,
 * depth_discontinuity_radius 5; min_disp is accepted and ignored (DF.cpp:146,149). */
int adf_wls_create(adf_wls_t** out, int use_confidence, int left_offset, int right_offset,
                   int top_offset, int bottom_offset, int min_disp);
void adf_wls_destroy(adf_wls_t* h);

/* DisparityWLSFilter get/set (DF.hpp:90-122, DF.cpp:126-136). */
int adf_wls_set_lambda(adf_wls_t* h, double lambda);
int adf_wls_get_lambda(const adf_wls_t* h, double* lambda);
/* (every sigma_color a handle has seen keeps its own immutable weight table on the device, up to eight: coming back to
 * one is free and capturable; a NEW value builds its table on the host inside the next filter call and uploads it with
 * one synchronous copy -- use each sigma once BEFORE capturing filter calls that switch between them into a hipGraph) */
int adf_wls_set_sigma_color(adf_wls_t* h, double sigma_color);
int adf_wls_get_sigma_color(const adf_wls_t* h, double* sigma_color);
int adf_wls_set_lrc_thresh(adf_wls_t* h, int lrc_thresh);
int adf_wls_get_lrc_thresh(const adf_wls_t* h, int* lrc_thresh);
int adf_wls_set_depth_discontinuity_radius(adf_wls_t* h, int radius);
int adf_wls_get_depth_discontinuity_radius(const adf_wls_t* h, int* radius);

/* The inner smoother's parameters, fixed to (0.25, 3) by the reference call site
 * createFastGlobalSmootherFilter(src, lambda, sigma_color) (DF.cpp:292, EF.hpp:393);
 * exposed because BASELINE config 5 names num_iter explicitly. */
int adf_wls_set_fgs_params(adf_wls_t* h, double lambda_attenuation, int num_iter);
int adf_wls_set_solver(adf_wls_t* h, int solver);
int adf_wls_get_solver(const adf_wls_t* h, int* solver);
/* Solver the last filter call actually ran: ADF_SOLVER_WAVE covers ROIs up to 8192 x 4352 (an 8K frame: beyond 4096
 * columns two wavefronts share a row, beyond 2176 rows a column is cut into 128 chunks); larger ones fall back to
 * ADF_SOLVER_EXACT. */
int adf_wls_get_last_solver(const adf_wls_t* h, int* solver);
/* Which kernels the confidence stage of the last filter call took (introspection for tests and benchmarks; the
 * results do not depend on it): ADF_PATH_CONF_BAND = computeConfidenceMap (DF.cpp:197-210) ran as the one-sweep band
 * kernel (depth-discontinuity radius 1..8), ADF_PATH_FUSED_FIRST_PASS = the first row pass formed conf*disp itself
 * (DF.cpp:288-290) instead of reading planes a prologue kernel wrote, ADF_PATH_MERGED_PREP = the edge weights, the
 * confidence map and the fill outside the ROI were one launch (calls of at most 2.5 Mpixels of ROI). */
#define ADF_PATH_CONF_BAND 1
#define ADF_PATH_FUSED_FIRST_PASS 2
#define ADF_PATH_MERGED_PREP 4       /* weights + confidence + fill ran as ONE launch (small calls) */
#define ADF_PATH_SCALED_FUSED 8      /* down-scaled call: the first row pass interpolated the low-resolution maps itself
                                        (no resize launch); the view-sized confidence map is produced on demand by
                                        adf_wls_get_confidence_* */
#define ADF_PATH_SCALED_HALF 16      /* ... in its form for maps of exactly half the view's width on a ROI starting on an even
                                        column >= 2 (the sample's default): four output columns share four source elements */
int adf_wls_get_last_path(const adf_wls_t* h, int* path_flags);

/* DisparityFilter::filter (DF.hpp:75, DF.cpp:219-298) on a batch of n_pairs
 * independent, equally sized stereo pairs laid out `*_pair_stride` bytes apart
 * (n_pairs = 1 is the reference call).  disparity maps: CV_16SC1 (disparity*16),
 * W x H; left_view: CV_8UC1 / CV_8UC3 (view_channels), same size; out: CV_16SC1,
 * W x H, filled with -16 outside the ROI (DF.cpp:254,284).  disp_right may be
 * NULL only for a filter created with use_confidence = 0.  roi: NULL or zero
 * area = derive from the offsets given at creation (DF.cpp:228-233). */
int adf_wls_filter_device(adf_wls_t* h, int n_pairs,
                          const int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                          const uint8_t* left_view, ptrdiff_t view_stride, ptrdiff_t view_pair_stride,
                          int view_channels, int W, int H,
                          int16_t* out, ptrdiff_t out_stride, ptrdiff_t out_pair_stride,
                          const int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                          const adf_rect* roi, void* stream);

int adf_wls_filter_host(adf_wls_t* h, int n_pairs,
                        const int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                        const uint8_t* left_view, ptrdiff_t view_stride, ptrdiff_t view_pair_stride,
                        int view_channels, int W, int H,
                        int16_t* out, ptrdiff_t out_stride, ptrdiff_t out_pair_stride,
                        const int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                        const adf_rect* roi);

/* The same call with disparity maps of a LOWER resolution than the view (DF.hpp:59-61: "Disparity map can
 * have any resolution, it will be automatically resized to fit left_view resolution"; DF.cpp:224-227,
 * 239-247, 268-277): maps are disp_W x disp_H, view and output W x H; the confidence map is computed at
 * the maps' resolution with LRC_thresh and the roll-off scaled by resize_factor = disp_W/W, then both are
 * resized (bilinear) and the disparity multiplied by W/disp_W.  roi is in disparity-map coordinates.
 * With disp_W == W and disp_H == H this is adf_wls_filter_*. */
int adf_wls_filter_scaled_device(adf_wls_t* h, int n_pairs,
                                 const int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                                 int disp_W, int disp_H,
                                 const uint8_t* left_view, ptrdiff_t view_stride, ptrdiff_t view_pair_stride,
                                 int view_channels, int W, int H,
                                 int16_t* out, ptrdiff_t out_stride, ptrdiff_t out_pair_stride,
                                 const int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                                 const adf_rect* roi, void* stream);
int adf_wls_filter_scaled_host(adf_wls_t* h, int n_pairs,
                               const int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                               int disp_W, int disp_H,
                               const uint8_t* left_view, ptrdiff_t view_stride, ptrdiff_t view_pair_stride,
                               int view_channels, int W, int H,
                               int16_t* out, ptrdiff_t out_stride, ptrdiff_t out_pair_stride,
                               const int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                               const adf_rect* roi);

/* getConfidenceMap() (DF.hpp:117, DF.cpp:138): CV_32FC1, W x H, values in [0,255],
 * zero outside the ROI, of pair `pair` of the last filter call.  Valid until the
 * next filter call on the handle. */
int adf_wls_get_confidence_device(adf_wls_t* h, int pair, float* dst, ptrdiff_t dst_stride, void* stream);
int adf_wls_get_confidence_host(adf_wls_t* h, int pair, float* dst, ptrdiff_t dst_stride);
/* The HIP device the handle's workspace lives on (the device current at adf_wls_create); device
 * pointers and streams passed to the handle must belong to it. */
int adf_wls_get_device(const adf_wls_t* h, int* device);
/* getROI() (DF.hpp:120, DF.cpp:139): ROI used by the last filter call. */
int adf_wls_get_roi(const adf_wls_t* h, adf_rect* roi);
/* Block until everything the handle queued on `stream` has finished. */
int adf_wls_sync(adf_wls_t* h, void* stream);
/* Bytes of device workspace currently owned by the handle. */
size_t adf_wls_workspace_bytes(const adf_wls_t* h);

/* Measurement hook (no counterpart in the reference, which times filter() from outside with
 * getTickCount: samples/disparity_filtering.cpp:186-191).  While enabled, every kernel launch of
 * adf_wls_filter_device is bracketed by HIP events recorded on the caller's stream;
 * adf_wls_profile_read waits for them and returns per-kernel-class launch counts, summed durations
 * and the algorithmic bytes those launches moved (SURVEY.md 8d figures). */
typedef struct adf_kernel_time {
    char name[48];      /* kernel class, e.g. "pass_h", "pass_v", "weights"            */
    int launches;
    double total_ms;    /* sum of event-to-event durations                               */
    double alg_bytes;   /* algorithmic bytes summed over the launches                    */
    double moved_bytes; /* bytes this implementation reads+writes, by construction       */
} adf_kernel_time;
int adf_wls_profile_enable(adf_wls_t* h, int on); /* also clears what was collected */
int adf_wls_profile_read(adf_wls_t* h, adf_kernel_time* out, int capacity, int* count);

/* ---------------- FastGlobalSmootherFilter ---------------- */

/* createFastGlobalSmootherFilter(guide, lambda, sigma_color, lambda_attenuation,
 * num_iter) (EF.hpp:393, FGS.cpp:141-180, 681-684).  guide: CV_8UC1 / CV_8UC3,
 * w x h, HOST pointer (copied).  The weights are built once and reused by every
 * filter call, as in the reference. */
int adf_fgs_create(adf_fgs_t** out, const uint8_t* guide, ptrdiff_t guide_stride, int guide_channels,
                   int w, int h, double lambda, double sigma_color, double lambda_attenuation,
                   int num_iter, int solver);
/* The same with the guide already resident in HBM (a DEVICE pointer): the guide never crosses PCIe -- its copy into
 * the handle and the weight kernel are queued on `stream`, which is also the stream the handle's filter calls are
 * expected on.  (Creation still uploads the 768 KB weight table built on the host with libm, FGS.cpp:663-675, and
 * synchronises `stream` once for it.)  For device pipelines such as the second in-tree caller, which
 * smooths flow fields against an image it already holds (sparse_match_interpolators.cpp:202-203:
 * fastGlobalSmootherFilter(prevImage, flow, ...)). */
int adf_fgs_create_device(adf_fgs_t** out, const uint8_t* guide, ptrdiff_t guide_stride, int guide_channels,
                          int w, int h, double lambda, double sigma_color, double lambda_attenuation,
                          int num_iter, int solver, void* stream);
void adf_fgs_destroy(adf_fgs_t* h);
/* Filters made and destroyed per call -- the one-shot fastGlobalSmootherFilter (EF.hpp:413, FGS.cpp:687-691), the
 * reference's own perf test (perf/perf_fgs_filter.cpp:70-76) -- would pay hipMalloc + hipFree (which waits for the whole
 * device) and 3*256*256 libm calls for the weight table on every call.  The library therefore keeps a destroyed
 * filter's device block (at most 8 blocks / 3 GB per process, handed to the next filter behind the event of its last
 * user: no host synchronisation) and the weight tables by (device, sigma) (16 of them).  This returns all of it to the
 * driver; it waits for the blocks' last users.  (The library does the same on its own whenever one of its device
 * allocations -- a filter's workspace, a matcher's -- is refused for want of memory, and tries again; allocations the
 * CALLER makes in the same process do not see the cache: call this before a large one.) */
void adf_release_cached_memory(void);
/* The weight table of a sigma_color exactly as the filters build it (FGS.cpp:150-154, 663-675: 3*256*256 entries
 * -exp(-sqrt(i)/sigma) through the host's libm; built by several threads, the underflowed tail stored as -0.0f without
 * calling libm).  Host only -- no device is touched: the CPU test suite compares it bit for bit with the oracle's table. */
#define ADF_WEIGHT_TABLE_LEVELS (3 * 256 * 256)
int adf_weight_table_host(float sigma_color, float* table, int levels);
int adf_fgs_get_device(const adf_fgs_t* h, int* device);

/* FastGlobalSmootherFilter::filter(src, dst) (EF.hpp:370, FGS.cpp:182-233).
 * src/dst: HOST pointers, same size as the guide, depth ADF_8U / ADF_16S / ADF_32F,
 * 1..4 interleaved channels; dst may alias src. */
int adf_fgs_filter_host(adf_fgs_t* h, const void* src, ptrdiff_t src_stride, void* dst,
                        ptrdiff_t dst_stride, int depth, int channels);
/* The same with DEVICE pointers, asynchronous on `stream` (a hipStream_t; NULL = the null
 * stream): for callers whose images already live in HBM (EdgeAwareInterpolator-style flow
 * post-filtering, sparse_match_interpolators.cpp:202-203).  dst may alias src.
 * Stream capture: the call may be captured into a hipGraph.  Calls on different streams are ordered by an event the
 * handle keeps; a call that is being captured neither waits for nor records it, so a handle whose calls are captured
 * must be used on ONE stream (captured and ordinary calls alike), and destroyed outside any capture once its graphs'
 * replays have finished (adf_fgs_destroy then synchronises the device instead of caching the handle's block). */
int adf_fgs_filter_device(adf_fgs_t* h, const void* src, ptrdiff_t src_stride, void* dst,
                          ptrdiff_t dst_stride, int depth, int channels, void* stream);

/* ---------------- evaluation utilities (DF.hpp:163-204) ---------------- */

#define ADF_UNKNOWN_DISPARITY 16320 /* DF.cpp:460 */

/* computeMSE(GT, src, ROI) (DF.hpp:176, DF.cpp:497-517): mean of (GT-src)^2 over ROI pixels whose
 * ground truth is known (!= 16320), divided by 256 (disparities are scaled by 16).  CV_16SC1 maps of
 * equal size W x H; `*_device` takes device pointers and synchronises `stream` to return the value. */
int adf_compute_mse_host(const int16_t* gt, ptrdiff_t gt_stride, const int16_t* src, ptrdiff_t src_stride,
                         int W, int H, const adf_rect* roi, double* mse);
int adf_compute_mse_device(const int16_t* gt, ptrdiff_t gt_stride, const int16_t* src, ptrdiff_t src_stride,
                           int W, int H, const adf_rect* roi, double* mse, void* stream);
/* computeBadPixelPercent(GT, src, ROI, thresh = 24) (DF.hpp:190, DF.cpp:519-539): percentage of known
 * ROI pixels with |GT-src| >= thresh. */
int adf_compute_bad_pixel_percent_host(const int16_t* gt, ptrdiff_t gt_stride, const int16_t* src, ptrdiff_t src_stride,
                                       int W, int H, const adf_rect* roi, int thresh, double* percent);
int adf_compute_bad_pixel_percent_device(const int16_t* gt, ptrdiff_t gt_stride, const int16_t* src, ptrdiff_t src_stride,
                                         int W, int H, const adf_rect* roi, int thresh, double* percent, void* stream);
/* getDisparityVis(src, dst, scale = 1.0) (DF.hpp:202, DF.cpp:541-556): CV_8U visualisation,
 * saturate_cast<uchar>(scale*src/16.0), unknown disparities -> 0. */
int adf_get_disparity_vis_host(const int16_t* src, ptrdiff_t src_stride, uint8_t* dst, ptrdiff_t dst_stride,
                               int W, int H, double scale);
int adf_get_disparity_vis_device(const int16_t* src, ptrdiff_t src_stride, uint8_t* dst, ptrdiff_t dst_stride,
                                 int W, int H, double scale, void* stream);

/* ---------------- block matcher feeding the filter ----------------
 * The reference's filter takes its maps from cv::StereoBM / cv::StereoSGBM (disparity_filters.cpp:386-449,
 * samples/disparity_filtering.cpp:151,214), classes of OpenCV's calib3d module that are not part of the
 * reference tree (parity unpinned there).  adf_bm_* is the published StereoBM algorithm on the device --
 * x-Sobel prefilter, SAD block matching, sub-pixel fit, CV_16SC1 output with 4 fractional bits, rejected
 * pixels = (minDisparity-1)*16 -- so a pair can go from views to filtered disparity without leaving HBM.
 * Bit-exact against oracle/adf_oracle_bm.c; pinned by the reference's own block-matching test data and
 * accuracy bar (modules/stereo/test/test_block_matching.cpp:61-82,148).  Limits: blockSize 5..21. */
typedef struct adf_bm adf_bm_t; /* cv::Ptr<StereoBM> */
/* StereoBM::create(numDisparities, blockSize); other parameters start at cv::StereoBM's defaults
 * (minDisparity 0, preFilterCap 31, textureThreshold 10, uniquenessRatio 15). */
int adf_bm_create(adf_bm_t** out, int num_disparities, int block_size);
void adf_bm_destroy(adf_bm_t* h);
/* The StereoMatcher / StereoBM setters in one call (the filter factory forces textureThreshold = 0 and
 * uniquenessRatio = 0, disparity_filters.cpp:399-400; the right-view matcher uses
 * minDisparity = -(min_disp+num_disp)+1, :424).  Values are checked at compute time like cv::StereoBM. */
int adf_bm_set_params(adf_bm_t* h, int min_disparity, int num_disparities, int block_size,
                      int prefilter_cap, int texture_threshold, int uniqueness_ratio);
int adf_bm_get_device(const adf_bm_t* h, int* device);
int adf_bm_get_params(const adf_bm_t* h, int* min_disparity, int* num_disparities, int* block_size,
                      int* prefilter_cap, int* texture_threshold, int* uniqueness_ratio);
/* StereoMatcher::compute(left, right, disparity) on a batch of n_pairs equally sized CV_8UC1 pairs laid out
 * `*_pair_stride` bytes apart; disparity: CV_16SC1, W x H (strides in bytes).  Asynchronous on `stream`. */
int adf_bm_compute_device(adf_bm_t* h, int n_pairs,
                          const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                          const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                          int W, int H,
                          int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride,
                          void* stream);
/* Extension (not a cv:: call): the left-view map of adf_bm_compute_device AND the map of the right-view matcher of
 * createRightMatcher (disparity_filters.cpp:417-431: views swapped, minDisparity = -(min_disp+num_disp)+1, texture
 * and uniqueness tests off) from one launch -- the two images are prefiltered once and both searches share a grid,
 * which matters for a single pair per call (1920x1080: 0.36 ms for two calls).  Results are identical to the two
 * separate computes. */
int adf_bm_compute_both_device(adf_bm_t* h, int n_pairs,
                               const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                               const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                               int W, int H,
                               int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                               int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                               void* stream);
int adf_bm_compute_host(adf_bm_t* h, int n_pairs,
                        const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                        const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                        int W, int H,
                        int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride);

/* ---------------- semi-global matcher feeding the filter ----------------
 * cv::StereoSGBM as the reference's sample configures it (samples/disparity_filtering.cpp:166-176, 229-235:
 * P1 = 24*w*w, P2 = 96*w*w, preFilterCap 63, MODE_SGBM_3WAY) and as its factories expect it
 * (disparity_filters.cpp:404-409, 432-445): the published semi-global algorithm (Hirschmueller 2008) with the
 * Birchfield-Tomasi block cost and three paths (MODE_SGBM_3WAY: left, top, right; MODE_SGBM: five, MODE_HH: eight --
 * modules/stereo/src/stereo_binary_sgbm.cpp:173-186, 286-301 is the in-tree statement of the path sets), sub-pixel fit,
 * 3x3 median of the result;
 * CV_16SC1 with 4 fractional bits, invalid pixels (minDisparity-1)*16, matchable columns
 * [max(minDisparity+numDisparities,0), W+min(minDisparity,0)).  The class itself lives in OpenCV's calib3d (outside
 * the reference tree: parity unpinned); results are bit-exact against oracle/adf_oracle_sgbm.c and anchored on the
 * reference's own semi-global test (modules/stereo/test/test_block_matching.cpp:157-238).  Not implemented, and not
 * reachable from the filter (its factories switch it off): the speckle filter.  Limits: numDisparities <= 512,
 * blockSize odd <= 11. */
#define ADF_SGBM_MODE_SGBM 0
#define ADF_SGBM_MODE_HH 1
#define ADF_SGBM_MODE_3WAY 2 /* StereoSGBM::MODE_SGBM_3WAY */
typedef struct adf_sgbm adf_sgbm_t; /* cv::Ptr<StereoSGBM> */
/* StereoSGBM::create(minDisparity, numDisparities, blockSize); the other parameters start at its defaults
 * (P1 = P2 = 0, preFilterCap 0, uniquenessRatio 0, disp12MaxDiff 0, mode MODE_SGBM). */
int adf_sgbm_create(adf_sgbm_t** out, int min_disparity, int num_disparities, int block_size);
void adf_sgbm_destroy(adf_sgbm_t* h);
int adf_sgbm_get_device(const adf_sgbm_t* h, int* device);
int adf_sgbm_set_params(adf_sgbm_t* h, int min_disparity, int num_disparities, int block_size, int P1, int P2,
                        int prefilter_cap, int uniqueness_ratio, int mode);
/* StereoSGBM::setDisp12MaxDiff: the matcher's own left-right check (stereo_binary_sgbm.cpp:548-556, 598-613 is the in-tree
 * statement).  cv::StereoSGBM::create's default 0 is read as 1 by the algorithm (check ON, :141); the filter factory sets
 * 1000000 (disparity_filters.cpp:389, 444), which switches it off. */
int adf_sgbm_set_disp12_max_diff(adf_sgbm_t* h, int disp12_max_diff);
int adf_sgbm_get_disp12_max_diff(const adf_sgbm_t* h, int* disp12_max_diff);
int adf_sgbm_get_params(const adf_sgbm_t* h, int* min_disparity, int* num_disparities, int* block_size, int* P1, int* P2,
                        int* prefilter_cap, int* uniqueness_ratio, int* mode);
/* StereoMatcher::compute(left, right, disparity) on n_pairs equally sized CV_8UC1 / CV_8UC3 pairs (`channels`);
 * disparity: CV_16SC1, W x H (strides in bytes).  Asynchronous on `stream`. */
int adf_sgbm_compute_device(adf_sgbm_t* h, int n_pairs,
                            const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                            const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                            int channels, int W, int H,
                            int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride,
                            void* stream);
int adf_sgbm_compute_host(adf_sgbm_t* h, int n_pairs,
                          const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                          const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                          int channels, int W, int H,
                          int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride);

#ifdef __cplusplus
}
#endif
#endif
