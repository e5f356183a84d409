// resize_kernels.hip -- cv::resize(..., INTER_LINEAR) for the down-scaled disparity path of
// DisparityWLSFilterImpl::filter (DF.cpp:239-247, 268-277): the low-resolution disparity map
// (CV_16SC1, then multiplied by x_ratio with saturation, DF.cpp:244,273) and the low-resolution
// confidence map (CV_32FC1, DF.cpp:274) are brought to the view's size.
//
// OpenCV's imgproc is not vendored by the reference (version unpinned): "parity unpinned" at this
// boundary.  The arithmetic below restates OpenCV 3.x's published algorithm (imgwarp.cpp,
// resizeGeneric_ + HResizeLinear / VResizeLinear with float weights) exactly as oracle/adf_oracle.c
// does, operation for operation (no fused multiply-add), so the two agree bit for bit:
//   fx = (float)((dx+0.5)*scale_x - 0.5); sx = floor(fx); fx -= sx; sx<0 -> (0,0); sx>=sw-1 -> (sw-1,0)
//   row value  = S[sx]*(1-fx) + S[sx+1]*fx          (S[sx] alone where sx+1 leaves the row)
//   result     = row(sy)*(1-fy) + row(sy+1)*fy      with source rows CLAMPED, then saturate_cast
#include "adf_internal.h"

#pragma clang fp contract(off)

namespace adf {

namespace {

template <bool IS16>
__global__ void __launch_bounds__(256) resize_linear_kernel(ResizeArgs a)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= a.dw) return;
    float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= a.sw - 1) { fx = 0.0f; sx = a.sw - 1; }
    const bool interp = sx + 1 < a.sw;
    const float a0 = 1.0f - fx, a1 = fx;
    float fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const float b0 = 1.0f - fy, b1 = fy;
    const int y0 = min(max(sy, 0), a.sh - 1), y1 = min(max(sy + 1, 0), a.sh - 1);
    const char* base = reinterpret_cast<const char*>(a.src) + (ptrdiff_t)blockIdx.z * a.spair;
    float r[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const char* row = base + (ptrdiff_t)(k ? y1 : y0) * a.sstride;
        const float v0 = IS16 ? (float)reinterpret_cast<const int16_t*>(row)[sx] : reinterpret_cast<const float*>(row)[sx];
        float v = v0;
        if (interp) {
            const float v1 = IS16 ? (float)reinterpret_cast<const int16_t*>(row)[sx + 1] : reinterpret_cast<const float*>(row)[sx + 1];
            v = v0 * a0 + v1 * a1;
        }
        r[k] = v;
    }
    const float v = r[0] * b0 + r[1] * b1;
    char* drow = reinterpret_cast<char*>(a.dst) + (ptrdiff_t)blockIdx.z * a.dpair + (ptrdiff_t)dy * a.dstride;
    if (IS16) {
        int16_t q = sat16(v);
        if (a.post_scale != 1.0f) q = sat16((float)q * a.post_scale + 0.0f); // DF.cpp:244,273
        reinterpret_cast<int16_t*>(drow)[dx] = q;
    } else
        reinterpret_cast<float*>(drow)[dx] = v;
}

} // namespace

hipError_t launch_resize_linear(const ResizeArgs& a, int n_pairs, hipStream_t st)
{
    if (a.sw <= 0 || a.sh <= 0 || a.dw <= 0 || a.dh <= 0 || n_pairs <= 0) return hipErrorInvalidValue;
    dim3 grid((a.dw + 255) / 256, a.dh, n_pairs);
    if (a.is16) hipLaunchKernelGGL(resize_linear_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(resize_linear_kernel<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace adf
