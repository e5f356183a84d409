// eval_kernels.hip -- disparity evaluation utilities of DF.cpp:460-556 on the device, so that a
// pipeline whose maps live in HBM can score them without a round trip:
//   computeMSE (DF.cpp:497-517), computeBadPixelPercent (:519-539), getDisparityVis (:541-556).
// The reductions are exact 64-bit integer sums (order-independent, bit-reproducible).
#include "adf_internal.h"
#include "../../include/adf_wls.h"

#include <cstdio>

namespace {

struct EvalArgs {
    const int16_t* gt; ptrdiff_t sg; const int16_t* src; ptrdiff_t ss;
    int x, y, w, h; int thresh;
    unsigned long long* acc; // [0] = sum of squared differences, [1] = known pixels, [2] = bad pixels
};

__global__ void __launch_bounds__(256) eval_reduce_kernel(EvalArgs a)
{
    unsigned long long sq = 0, cnt = 0, bad = 0;
    for (int i = blockIdx.y; i < a.h; i += gridDim.y) {
        const int16_t* g = reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(a.gt) + (ptrdiff_t)(a.y + i) * a.sg) + a.x;
        const int16_t* s = reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(a.src) + (ptrdiff_t)(a.y + i) * a.ss) + a.x;
        for (int j = blockIdx.x * 256 + threadIdx.x; j < a.w; j += gridDim.x * 256) {
            const int gv = g[j];
            if (gv != ADF_UNKNOWN_DISPARITY) {                       // DF.cpp:507,529
                const long long d = (long long)gv - (long long)s[j];
                sq += (unsigned long long)(d * d);
                cnt++;
                if ((d < 0 ? -d : d) >= a.thresh) bad++;            // DF.cpp:531
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sq += __shfl_down(sq, off); cnt += __shfl_down(cnt, off); bad += __shfl_down(bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.acc[0], sq); atomicAdd(&a.acc[1], cnt); atomicAdd(&a.acc[2], bad);
    }
}

__global__ void __launch_bounds__(256) vis_kernel(const int16_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t sd, int W, int H, double scale)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= W || i >= H) return;
    const int v = reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(src) + (ptrdiff_t)i * ss)[j];
    uint8_t o = 0;
    if (v != ADF_UNKNOWN_DISPARITY) {                                // DF.cpp:551-554
        const double t = scale * v / 16.0;                            // saturate_cast<uchar>(double): cvRound + clamp
        if (t >= -2147483648.0 && t < 2147483648.0) {
            const double r = rint(t);
            o = (uint8_t)(r < 0.0 ? 0.0 : r > 255.0 ? 255.0 : r);
        }
    }
    (dst + (ptrdiff_t)i * sd)[j] = o;
}

int run_eval(const int16_t* gt, ptrdiff_t sg, const int16_t* src, ptrdiff_t ss, int W, int H, const adf_rect* roi,
             int thresh, bool device, hipStream_t st, unsigned long long out[3])
{
    if (!gt || !src || W <= 0 || H <= 0) return adf::set_error(ADF_EBADARG, "GT / src must be non-empty CV_16SC1 maps"); // DF.cpp:499-501
    adf_rect r = roi && roi->width * roi->height != 0 ? *roi : adf_rect{0, 0, W, H};
    if (r.x < 0 || r.y < 0 || r.width <= 0 || r.height <= 0 || r.x + r.width > W || r.y + r.height > H)
        return adf::set_error(ADF_ESIZE, "ROI does not fit the maps");
    const int16_t *dg = gt, *ds = src;
    ptrdiff_t dsg = sg, dss = ss;
    void* tmp = nullptr;
    if (!device) {
        const size_t rowb = (size_t)W * 2;
        if (hipMalloc(&tmp, 2 * rowb * H + 64) != hipSuccess) return adf::set_error(ADF_ENOMEM, "hipMalloc failed");
        if (hipMemcpy2DAsync(tmp, rowb, gt, sg, rowb, H, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpy2DAsync((char*)tmp + rowb * H, rowb, src, ss, rowb, H, hipMemcpyHostToDevice, st) != hipSuccess) {
            hipFree(tmp); return adf::set_error(ADF_EHIP, "host to device copy failed");
        }
        dg = (const int16_t*)tmp; ds = (const int16_t*)((char*)tmp + rowb * H); dsg = dss = (ptrdiff_t)rowb;
    }
    unsigned long long* acc = nullptr;
    if (hipMalloc((void**)&acc, 3 * sizeof(unsigned long long)) != hipSuccess) { if (tmp) hipFree(tmp); return ADF_ENOMEM; }
    hipMemsetAsync(acc, 0, 3 * sizeof(unsigned long long), st);
    EvalArgs a{dg, dsg, ds, dss, r.x, r.y, r.width, r.height, thresh, acc};
    dim3 grid((r.width + 255) / 256 > 64 ? 64 : (r.width + 255) / 256, r.height > 256 ? 256 : r.height);
    hipLaunchKernelGGL(eval_reduce_kernel, grid, dim3(256), 0, st, a);
    hipError_t e = hipMemcpyAsync(out, acc, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(acc);
    if (tmp) hipFree(tmp);
    return e == hipSuccess ? ADF_OK : adf::set_error(ADF_EHIP, hipGetErrorString(e));
}

} // namespace

extern "C" int adf_compute_mse_device(const int16_t* gt, ptrdiff_t sg, const int16_t* src, ptrdiff_t ss, int W, int H,
                                      const adf_rect* roi, double* mse, void* stream)
{
    unsigned long long o[3];
    int rc = run_eval(gt, sg, src, ss, W, H, roi, 24, true, (hipStream_t)stream, o);
    if (rc == ADF_OK && mse) *mse = (double)o[0] / ((double)o[1] * 256.0);  // DF.cpp:515 (res /= cnt*256)
    return rc;
}
extern "C" int adf_compute_mse_host(const int16_t* gt, ptrdiff_t sg, const int16_t* src, ptrdiff_t ss, int W, int H,
                                    const adf_rect* roi, double* mse)
{
    unsigned long long o[3];
    int rc = run_eval(gt, sg, src, ss, W, H, roi, 24, false, nullptr, o);
    if (rc == ADF_OK && mse) *mse = (double)o[0] / ((double)o[1] * 256.0);
    return rc;
}
extern "C" int adf_compute_bad_pixel_percent_device(const int16_t* gt, ptrdiff_t sg, const int16_t* src, ptrdiff_t ss, int W,
                                                    int H, const adf_rect* roi, int thresh, double* percent, void* stream)
{
    unsigned long long o[3];
    int rc = run_eval(gt, sg, src, ss, W, H, roi, thresh, true, (hipStream_t)stream, o);
    if (rc == ADF_OK && percent) *percent = (100.0 * (double)o[2]) / (double)o[1];  // DF.cpp:538
    return rc;
}
extern "C" int adf_compute_bad_pixel_percent_host(const int16_t* gt, ptrdiff_t sg, const int16_t* src, ptrdiff_t ss, int W,
                                                  int H, const adf_rect* roi, int thresh, double* percent)
{
    unsigned long long o[3];
    int rc = run_eval(gt, sg, src, ss, W, H, roi, thresh, false, nullptr, o);
    if (rc == ADF_OK && percent) *percent = (100.0 * (double)o[2]) / (double)o[1];
    return rc;
}
extern "C" int adf_get_disparity_vis_device(const int16_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t sd, int W, int H,
                                            double scale, void* stream)
{
    if (!src || !dst || W <= 0 || H <= 0) return ADF_EBADARG;        // DF.cpp:543
    hipLaunchKernelGGL(vis_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, src, ss, dst, sd, W, H, scale);
    return hipGetLastError() == hipSuccess ? ADF_OK : ADF_EHIP;
}
extern "C" int adf_get_disparity_vis_host(const int16_t* src, ptrdiff_t ss, uint8_t* dst, ptrdiff_t sd, int W, int H, double scale)
{
    if (!src || !dst || W <= 0 || H <= 0) return ADF_EBADARG;
    void* tmp = nullptr;
    const size_t sb = (size_t)W * 2, db = (size_t)W;
    if (hipMalloc(&tmp, (sb + db) * H + 64) != hipSuccess) return ADF_ENOMEM;
    hipError_t e = hipMemcpy2D(tmp, sb, src, ss, sb, H, hipMemcpyHostToDevice);
    uint8_t* dd = (uint8_t*)tmp + sb * H;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vis_kernel, dim3((W + 255) / 256, H), dim3(256), 0, nullptr, (const int16_t*)tmp, (ptrdiff_t)sb, dd, (ptrdiff_t)db, W, H, scale);
        e = hipMemcpy2D(dst, sd, dd, db, db, H, hipMemcpyDeviceToHost);
    }
    hipFree(tmp);
    return e == hipSuccess ? ADF_OK : ADF_EHIP;
}
