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

// One thread = RC adjacent destination columns x RR destination rows: the horizontal taps and weights are
// computed once per column and reused down the rows, and the RC results of a row leave in one store
// (8 bytes of int16 / 16 bytes of float instead of 2 / 4 per lane).  Per pixel the arithmetic is unchanged.
constexpr int RC = 4, RR = 4;
static_assert(RC == 4, "the vector stores below write four columns");

template <bool IS16>
__global__ void __launch_bounds__(256) resize_linear_kernel(ResizeArgs a)
{
    const int dx0 = (blockIdx.x * 256 + threadIdx.x) * RC, dy0 = blockIdx.y * RR;
    if (dx0 >= a.dw) return;
    int sx[RC]; float a0[RC], a1[RC]; bool interp[RC];
#pragma unroll
    for (int c = 0; c < RC; c++) {
        const int dx = min(dx0 + c, a.dw - 1);
        float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5);
        int s0 = (int)floorf(fx);
        fx -= (float)s0;
        if (s0 < 0) { fx = 0.0f; s0 = 0; }
        if (s0 >= a.sw - 1) { fx = 0.0f; s0 = a.sw - 1; }
        sx[c] = s0; interp[c] = s0 + 1 < a.sw; a0[c] = 1.0f - fx; a1[c] = fx;
    }
    const char* base = reinterpret_cast<const char*>(a.src) + (ptrdiff_t)blockIdx.z * a.spair;
    char* dbase = reinterpret_cast<char*>(a.dst) + (ptrdiff_t)blockIdx.z * a.dpair;
    auto hrow = [&](int y, int c) -> float {                     // horizontally interpolated source row y at column c
        const char* row = base + (ptrdiff_t)y * a.sstride;
        const float v0 = IS16 ? (float)reinterpret_cast<const int16_t*>(row)[sx[c]] : reinterpret_cast<const float*>(row)[sx[c]];
        if (!interp[c]) return v0;
        const float v1 = IS16 ? (float)reinterpret_cast<const int16_t*>(row)[sx[c] + 1] : reinterpret_cast<const float*>(row)[sx[c] + 1];
        return v0 * a0[c] + v1 * a1[c];
    };
    int cy[2] = {-1, -1};
    float ch0[RC], ch1[RC], res[RR][RC];
#pragma unroll
    for (int c = 0; c < RC; c++) { ch0[c] = 0.0f; ch1[c] = 0.0f; }
#pragma unroll
    for (int k = 0; k < RR; k++) {
        const int dy = dy0 + k;
        if (dy >= a.dh) break;
        float fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        const float b0 = 1.0f - fy, b1 = fy;
        const int y0 = min(max(sy, 0), a.sh - 1), y1 = min(max(sy + 1, 0), a.sh - 1);
        // consecutive destination rows share source rows when enlarging: keep the two most recent
        // horizontally interpolated rows (same values, so the same bits) instead of fetching them again
        float v[RC];
        if (y0 != cy[0] && y0 != cy[1]) {           // replace the older entry
            const int e = (cy[0] <= cy[1]) ? 0 : 1;
#pragma unroll
            for (int c = 0; c < RC; c++) { const float t = hrow(y0, c); if (e == 0) ch0[c] = t; else ch1[c] = t; }
            cy[e] = y0;
        }
        if (y1 != cy[0] && y1 != cy[1]) {
            const int e = (cy[0] == y0) ? 1 : 0;    // never evict the row this output still needs
#pragma unroll
            for (int c = 0; c < RC; c++) { const float t = hrow(y1, c); if (e == 0) ch0[c] = t; else ch1[c] = t; }
            cy[e] = y1;
        }
#pragma unroll
        for (int c = 0; c < RC; c++) {
            const float r0 = (y0 == cy[0]) ? ch0[c] : ch1[c];
            const float r1 = (y1 == cy[0]) ? ch0[c] : ch1[c];
            v[c] = r0 * b0 + r1 * b1;
        }
#pragma unroll
        for (int c = 0; c < RC; c++) res[k][c] = v[c];
    }
    // stores after all loads (stores count in vmcnt on this target: a row's loads would otherwise wait for the
    // previous row's stores)
#pragma unroll
    for (int k = 0; k < RR; k++) {
        const int dy = dy0 + k;
        if (dy >= a.dh) break;
        float v[RC];
#pragma unroll
        for (int c = 0; c < RC; c++) v[c] = res[k][c];
        char* drow = dbase + (ptrdiff_t)dy * a.dstride;
        const bool full = dx0 + RC <= a.dw;
        if (IS16) {
            int16_t q[RC];
#pragma unroll
            for (int c = 0; c < RC; c++) {
                q[c] = sat16(v[c]);
                if (a.post_scale != 1.0f) q[c] = sat16((float)q[c] * a.post_scale + 0.0f); // DF.cpp:244,273
            }
            int16_t* d = reinterpret_cast<int16_t*>(drow) + dx0;
            if (full && (reinterpret_cast<uintptr_t>(d) & 7u) == 0)
                *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)(unsigned short)q[0] | ((unsigned)(unsigned short)q[1] << 16),
                                                          (unsigned)(unsigned short)q[2] | ((unsigned)(unsigned short)q[3] << 16));
            else {
#pragma unroll
                for (int c = 0; c < RC; c++) if (dx0 + c < a.dw) d[c] = q[c];
            }
        } else {
            float* d = reinterpret_cast<float*>(drow) + dx0;
            if (full && (reinterpret_cast<uintptr_t>(d) & 15u) == 0) *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int c = 0; c < RC; c++) if (dx0 + c < a.dw) d[c] = v[c];
            }
        }
    }
}

} // namespace

hipError_t launch_resize_linear(const ResizeArgs& a, int n_pairs, hipStream_t st)
{
    if (a.sw <= 0 || a.sh <= 0 || a.dw <= 0 || a.dh <= 0 || n_pairs <= 0) return hipErrorInvalidValue;
    dim3 grid((a.dw + 256 * RC - 1) / (256 * RC), (a.dh + RR - 1) / RR, n_pairs);
    if (a.is16) hipLaunchKernelGGL(resize_linear_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(resize_linear_kernel<false>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace adf
