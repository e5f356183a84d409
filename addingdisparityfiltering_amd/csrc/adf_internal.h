// adf_internal.h -- shared declarations between the HIP kernels and the C-ABI host code.
// Not installed; the public boundary is include/adf_wls.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define ADF_LUT_LEVELS (3 * 256 * 256) /* FGS.cpp:150 */
#define ADF_EPS 1e-43f                 /* DF.cpp:47  */

namespace adf {

// Geometry of one filter call.  ROI planes are kept in two orientations:
//   N ("natural")    [rh][pw] : column index fastest -> lane-per-column (vertical) sweeps coalesce
//   T ("transposed") [rw][ph] : row index fastest    -> lane-per-row (horizontal) sweeps coalesce
// pw / ph are rw / rh rounded up to 64 floats so every sweep step is one aligned 256-byte row.
struct Geom {
    int W, H;           // full frame
    int rx, ry, rw, rh; // ROI (DF.cpp:228-233)
    int pw, ph;         // padded pitches (floats)
    size_t plane;       // floats per pair per ROI plane = max(rh*pw, rw*ph)
    size_t frame;       // floats per pair per full-frame plane = W*H
    // The confidence plane (library-owned; getConfidenceMap copies it out) has a layout of its own: frame pixel (i, j)
    // at cframe*pair + i*cpitch + cx0 + j, with cx0 chosen so that ROI column 0 is 16-byte aligned for ANY ROI x,
    // cpitch a multiple of 4 floats that leaves at least 3 zero floats behind frame column W-1 (the fused first row
    // pass reads whole float4s: what it over-reads is zero), zeros everywhere outside the ROI.
    int cpitch, cx0;
    size_t cframe;      // floats per pair of the confidence plane = cpitch*H
};

// Destination layout of a plane-writing kernel.
//   ORIENT_PAIR   (wave solver, two right-hand sides) ONE plane of 2*plane floats per image holding
//                 both, tiled [row pair][16-column strip][row parity][U0 x16 | U1 x16]: a strip row of the
//                 column pass is one 128-byte line and two consecutive rows of a strip are 256 contiguous
//                 bytes, while an image row is still 128-byte pieces at a fixed 256-byte stride for the row
//                 pass; the U1 pointer is unused
//   ORIENT_STRIP  (wave solver, Cvert) strip-major [pw/16][rh][16]
enum Orient { ORIENT_N = 0, ORIENT_T = 1, ORIENT_PAIR = 2, ORIENT_STRIP = 3 };
#define ADF_STRIP 16
// Rows per tile of the pair plane: [rh/TR][pw/16][TR rows][U0 x16 | U1 x16] -- a strip visit of the column pass is
// TR*128 contiguous bytes, a row of the row pass is 128-byte pieces at a TR*128-byte stride (power of two).
#ifndef ADF_TILE_ROWS
#define ADF_TILE_ROWS 2
#endif

// What the last sweep of a solve writes (fused epilogues).
enum Epilogue {
    EPI_PLANES = 0,   // float planes in the opposite orientation (feeds the next pass)
    EPI_WLS_CONF = 1, // int16 = sat(u0 * (1/(u1+EPS)))   DF.cpp:295-296
    EPI_I16 = 2,      // int16 = sat(u0)                  FGS.cpp:216 (no-confidence path, DF.cpp:257-258)
    EPI_F32 = 3,      // float natural layout             FGS.cpp:218
    EPI_U8 = 4        // uint8 = sat(u0)                  FGS.cpp:216
};

// Both views of a pair in one launch: index 0 = left, 1 = right (mirrored ROI x, DF.cpp:202-203).
struct DiscArgs {
    const int16_t* disp[2]; ptrdiff_t stride[2], pair_stride[2]; // bytes
    int rx[2]; int ry, rw, rh;
    int radius; float roll_off;
    float* dst[2]; int W; size_t frame; // full-frame float planes, W pitch
    int only_view;                      // -1 = both views, else the one view to compute
};

// Left view: discontinuity map + LRC + x255 in one sweep (no cL round trip through HBM); needs the
// right view's discontinuity map cR complete.  Optionally also writes the two right-hand sides.
struct ConfLeftArgs {
    const int16_t* dL; ptrdiff_t sL, psL; // bytes
    const int16_t* dR; ptrdiff_t sR, psR;
    const float* cR;                      // full-frame, W pitch
    float* conf;                          // confidence plane (x255; Geom::cpitch layout); only ROI pixels are written
    float* U0; float* U1;                 // ORIENT_PAIR plane at U0 (may be null: first pass reads conf/dL itself); U1 unused
    Geom g; int rrx; int thresh;
    int radius; float roll_off;
};

// Both views' discontinuity maps + LRC + x255 in ONE sweep over row bands (conf_band_kernel): the right view's
// map of a row lives in LDS only, so nothing but dL, dR (read) and the confidence map (written) touches HBM.
struct ConfBandArgs {
    const int16_t* dL; ptrdiff_t sL, psL; // bytes
    const int16_t* dR; ptrdiff_t sR, psR;
    float* conf;                          // confidence plane (x255; Geom::cpitch layout); only ROI pixels are written
    Geom g; int rrx; int thresh;
    int radius; float roll_off;
    int rows_per_band;
    // dynamic LDS to ask for at least: above 80 KiB only ONE band workgroup fits a CU, which leaves wave slots for the
    // weight kernel that runs beside this one on the side stream (0 = just what the kernel needs)
    size_t lds_floor;
};

// Non-ROI pixels: filtered map = fill (DF.cpp:284), confidence = 0 (DF.cpp:187-190); either may be null.
struct OutsideArgs {
    int16_t* out; ptrdiff_t stride, pair_stride; int16_t fill;
    float* conf;
    Geom g;
};

struct LrcArgs {
    const int16_t* dL; ptrdiff_t sL, psL; // bytes
    const int16_t* dR; ptrdiff_t sR, psR;
    const float* cL; const float* cR; // full-frame discontinuity maps
    float* conf;                      // confidence plane (x255; Geom::cpitch layout), zero outside ROI
    int16_t* out; ptrdiff_t sO, psO; int16_t fill; // filtered map: `fill` outside the ROI (DF.cpp:284); may be null
    float* U0; float* U1;             // ROI planes: conf*disp, conf
    Geom g; int rrx;                  // right ROI x (DF.cpp:202)
    int thresh; int orient;           // orientation of U0/U1
};

// Source plane -> float right-hand side on the ROI: the no-confidence path's float(disp)
// (DF.cpp:250,257) and FastGlobalSmootherFilter::filter's split + convertTo (FGS.cpp:191-205).
struct PlainPrologueArgs {
    const void* src; ptrdiff_t stride, pair_stride; // bytes
    int depth, cn, c;                               // adf_depth code, channel count, channel index
    float* U0; Geom g; int orient;
    // optional confidence weighting (down-scaled path, DF.cpp:286-290 on the resized maps):
    // U0 = conf*float(src), U1 = conf, conf read from a confidence plane (Geom::cpitch layout)
    const float* conf; float* U1;
    // or a second source channel as the second right-hand side (generic FGS on the wave solver: channel
    // pairs share one factorisation): U1 = float(src channel c2) when pair2 is set (conf must be null)
    int pair2, c2;
};

// cv::resize(INTER_LINEAR) of CV_16SC1 (is16, optional saturating post-scale) or CV_32FC1 images.
struct ResizeArgs {
    const void* src; ptrdiff_t sstride, spair; int sw, sh; // bytes
    void* dst; ptrdiff_t dstride, dpair; int dw, dh;
    double scale_x, scale_y;                               // sw/dw, sh/dh
    float post_scale; int is16;
    int zero_outside = 0, vx0 = 0, vy0 = 0, vx1 = 0, vy1 = 0; // source elements outside [vx0,vx1) x [vy0,vy1) read as zero
};

struct WeightArgs {
    const uint8_t* guide; ptrdiff_t stride, pair_stride; int ch; // bytes
    const float* lut;
    float* chor; float* cvert; int chor_orient, cvert_orient;
    Geom g;
    // exact solver (chor_orient == ORIENT_T): a free plane the streaming kernel can write Chor row-major
    // into before it is transposed; null = the generic tile kernel writes the transposed plane itself
    float* scratch;
};

// One solve pass (forward elimination + back substitution along every scanline).
// Input planes have the scanline index fastest: element (step t, scanline s) at t*pitch_in + s.
struct PassArgs {
    const float* C; const float* U0; const float* U1;
    float* D; float* F0; float* F1; // forward intermediates, same layout as the inputs
    float* O0; float* O1;           // EPI_PLANES: [nscan][pitch_out], step index fastest
    void* out; ptrdiff_t out_stride, out_pair_stride; // other epilogues: natural image, bytes
    int out_x0, out_y0, out_cn, out_c;                // ROI origin, channel count / index of `out`
    int nscan, len, pitch_in, pitch_out;
    size_t plane;
    float lambda;
};

// One pass of the on-chip partitioned solver (fgs_wave.hip).  Planes are row-major [rh][pitch] and
// are solved IN PLACE; horizontal: nscan = rows, len = row length; vertical: nscan = columns,
// len = column length.  The last vertical pass may fuse an epilogue and write `out` instead.
struct WavePassArgs {
    const float* C; float* U0; float* U1;
    // fused prologue of the first horizontal pass (DF.cpp:288-290): when conf_in is set the right-hand
    // sides are read as U1 = conf, U0 = conf * float(dL) from the confidence plane and the disparity map
    const float* conf_in; size_t conf_frame; int conf_pitch, conf_x0, conf_y0;
    const int16_t* dl_in; ptrdiff_t dl_stride, dl_pair_stride; int dl_x0, dl_y0;
    // ... or, down-scaled path (DF.cpp:268-277 then 288-290), from the LOW-resolution maps: when lo_conf is set the
    // pass interpolates the confidence map and the left disparity map itself (cv::resize INTER_LINEAR, tap for tap as
    // resize_kernels.hip / the oracle's resize_linear_*; disparity saturated, multiplied by lo_post_scale and
    // saturated again, DF.cpp:273), so neither view-sized plane is ever written.  ROI column j / row i of the pass is
    // view pixel (hi_x0 + j, hi_y0 + i).  Confidence elements outside [lo_vx0, lo_vx1) x [lo_vy0, lo_vy1) read as
    // zero when lo_zero_outside is set (the band kernel writes the maps' ROI only, DF.cpp:187-190).
    const float* lo_conf; ptrdiff_t lo_conf_stride, lo_conf_pair;      // floats
    const int16_t* lo_dl; ptrdiff_t lo_dl_stride, lo_dl_pair;          // bytes
    int lo_w, lo_h, hi_x0, hi_y0;
    double lo_scale_x, lo_scale_y; float lo_post_scale;
    int lo_zero_outside, lo_vx0, lo_vy0, lo_vx1, lo_vy1;
    int lo_half;      // 0: never the half-width form of the low-resolution prologue (ADF_LO_HALF=0: A/B and test knob)
    float* lo_taps;   // scratch, 4*ceil(len/4) floats, 16-byte aligned: the launcher fills it with the columns' taps (s0 + fx)
    void* out; ptrdiff_t out_stride, out_pair_stride;
    int out_x0, out_y0, out_cn, out_c;
    int nscan, len, pitch;
    size_t plane;
    float lambda;
};

// Records the calling thread's last error message (adf_last_error) and returns `code`.
int set_error(int code, const char* msg);
// hipMalloc that, when the driver refuses, first hands the library's own cache of destroyed filters' device blocks
// (up to 3 GB, adf_api.hip: BlockCache) back to the driver and tries once more: no call may fail for want of memory
// the library itself is sitting on.
hipError_t device_malloc(void** p, size_t bytes);

// Launchers (defined in the .hip files).  All are asynchronous on `st`.
hipError_t launch_discontinuity(const DiscArgs& a, int n_pairs, hipStream_t st);
hipError_t launch_lrc_prologue(const LrcArgs& a, int n_pairs, hipStream_t st);
hipError_t launch_conf_left(const ConfLeftArgs& a, int n_pairs, hipStream_t st); // radius <= 8 only
hipError_t launch_outside(const OutsideArgs& a, int n_pairs, hipStream_t st);
bool conf_band_fits(const Geom& g, int radius);                                 // geometry / radius the band kernel covers
hipError_t launch_conf_band(const ConfBandArgs& a, int n_pairs, hipStream_t st);
// weights + confidence map + outside fill in one launch, for calls small enough to be latency-bound (conf_kernels.hip)
bool prep_small_guide_fits(const Geom& g, ptrdiff_t guide_stride, int channels);
bool prep_small_fits(const Geom& g, int radius, int channels, int n_pairs);
hipError_t launch_prep_small(const ConfBandArgs& c, const WeightArgs& w, const OutsideArgs& o, int n_pairs, hipStream_t st);
int conf_left_max_radius();
hipError_t launch_plain_prologue(const PlainPrologueArgs& a, int n_pairs, hipStream_t st);
hipError_t launch_weights(const WeightArgs& a, int n_pairs, hipStream_t st);
hipError_t launch_resize_linear(const ResizeArgs& a, int n_pairs, hipStream_t st);
hipError_t launch_exact_pass(const PassArgs& a, int n_rhs, int epilogue, int n_pairs, hipStream_t st);
hipError_t launch_wave_hpass(const WavePassArgs& a, int n_rhs, int n_pairs, hipStream_t st);
hipError_t launch_wave_vpass(const WavePassArgs& a, int n_rhs, int epilogue, int n_pairs, hipStream_t st);
int wave_max_row_len();
bool wave_hpass_can_fuse(const WavePassArgs& a);
bool wave_hpass_can_fuse_lo(const WavePassArgs& a);   // the low-resolution form: scale / span limits of the LDS staging
bool wave_hpass_lo_half(const WavePassArgs& a);      // ... in its half-width form (fgs_wave_h.hip, FUSE_LO_HALF)?
int wave_max_col_len();
// largest depth-discontinuity radius the tile kernel supports (LDS bound)
int max_disc_radius();

// Device-side helpers shared by kernels.
#if defined(__HIPCC__)
// float index of U0(i, j) inside an ORIENT_PAIR plane of pitch pw (U1 is ADF_STRIP floats further)
__device__ __forceinline__ size_t pair_index(int i, int j, int pw)
{
    constexpr int TR = ADF_TILE_ROWS;
    return (size_t)(i / TR) * (size_t)(2 * TR * pw) + (size_t)((j >> 4) * (32 * TR) + (i % TR) * 32 + (j & 15));
}
// float index of C(i, j) inside an ORIENT_STRIP plane of rh rows
__device__ __forceinline__ size_t strip_index(int i, int j, int rh)
{
    return ((size_t)(j >> 4) * (size_t)rh + (size_t)i) * ADF_STRIP + (size_t)(j & 15);
}
// saturate_cast<short>(float): cvRound (round-half-even; NaN / out-of-int-range -> INT_MIN) + clamp.
__device__ __forceinline__ int16_t sat16(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int16_t)-32768;
    float r = rintf(v);
    r = fminf(fmaxf(r, -32768.0f), 32767.0f);
    return (int16_t)(int)r;
}
__device__ __forceinline__ uint8_t sat8(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (uint8_t)0;
    float r = rintf(v);
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (uint8_t)(int)r;
}
// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
#endif

} // namespace adf
