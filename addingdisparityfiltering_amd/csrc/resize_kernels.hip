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

// One thread = RC adjacent destination columns x RR destination rows.  Every source row the tile needs is requested
// first -- unconditional loads at clamped indices, nothing under a branch, so all of them are in flight at once (the
// first version fetched rows on demand behind per-column branches: eight dependent memory round trips per thread,
// 1.3-2.2 TB/s) -- then the rows are interpolated horizontally once and combined per destination row.  The arithmetic
// per pixel is the oracle's, operation for operation: a tap whose neighbour would leave the row has weight (1, 0) and
// reads the edge element twice, v*1 + v*0 = v exactly as the reference's "S[sx] alone" (no value here is -0 or
// non-finite).
//   UP (scale_y <= 0.75, the path's own case: maps smaller than the view): consecutive destination rows share source
//   rows -- RR rows touch at most RR + 1 consecutive source rows, chosen per destination row by a wave-uniform index;
//   otherwise every destination row brings its own two rows (2 * RR loads rows).
// Source elements outside [vx0, vx1) x [vy0, vy1) read as zero when `zero_outside` is set: the low-resolution
// confidence map is only written inside its ROI by the band kernel (DF.cpp:187-190: zero elsewhere), which saves the
// separate fill launch.
#ifndef ADF_RESIZE_RR
#define ADF_RESIZE_RR 4
#endif
#ifndef ADF_RESIZE_TY
#define ADF_RESIZE_TY 4
#endif
constexpr int RC = 4, RR = ADF_RESIZE_RR, TY = ADF_RESIZE_TY;
static_assert(RC == 4, "the vector stores below write four columns");

//   VEC (scale_x <= 0.6 as well, i.e. the maps at most 0.6 of the view's width): the RC destination columns of a thread
//   tap at most four adjacent source elements, fetched as ONE 16-byte (8-byte for CV_16S) load per source row at
//   element alignment; a tap picks its element by a per-lane index fixed for the whole tile.  A dword load per tap cost
//   ~32 cycles of the CU's address path each (40 per thread): the kernels ran at 1.0-1.8 TB/s whatever the loads'
//   order.
template <bool IS16, bool UP, bool VEC>
__global__ void __launch_bounds__(256) resize_linear_kernel(ResizeArgs a)
{
    static_assert(!VEC || UP, "the vector path is an upscaling path");
    constexpr int NS = UP ? RR + 1 : 2 * RR;
    const int dx0 = (blockIdx.x * 256 + threadIdx.x) * RC;
    if (dx0 >= a.dw) return;
    unsigned o0[RC], o1[RC]; float a0[RC], a1[RC]; bool in0[RC], in1[RC];
    int j0[RC], j1[RC], e0 = 0;
#pragma unroll
    for (int c = 0; c < RC; c++) {
        const int dx = min(dx0 + c, a.dw - 1);
        float fx = (float)(((double)dx + 0.5) * a.scale_x - 0.5);
        int s0 = (int)floorf(fx);
        fx -= (float)s0;
        if (s0 < 0) { fx = 0.0f; s0 = 0; }
        if (s0 >= a.sw - 1) { fx = 0.0f; s0 = a.sw - 1; }
        const int s1 = min(s0 + 1, a.sw - 1);
        o0[c] = (unsigned)s0 * (IS16 ? 2u : 4u); o1[c] = (unsigned)s1 * (IS16 ? 2u : 4u);
        a0[c] = 1.0f - fx; a1[c] = fx;
        in0[c] = !a.zero_outside || (s0 >= a.vx0 && s0 < a.vx1);
        in1[c] = !a.zero_outside || (s1 >= a.vx0 && s1 < a.vx1);
        if (VEC) {
            if (c == 0) e0 = min(s0, a.sw - 4);                         // (the launcher guarantees sw >= 4)
            j0[c] = min(s0 - e0, 3); j1[c] = min(s1 - e0, 3);           // 0..3 by the scale bound
        }
    }
    // a thread walks TY tiles down its columns (the column taps, weights and masks are made once; waves live long
    // enough to hide their start-up)
#pragma unroll 1
    for (int tile = 0; tile < TY; tile++) {
        const int dy0 = ((int)blockIdx.y * TY + tile) * RR;
        if (dy0 >= a.dh) break;
        // rows (wave-uniform): weights per destination row, source rows of the tile
        float b0[RR], b1[RR]; int i0[RR], i1[RR], ys[NS];
        int ymin = 0;
    #pragma unroll
        for (int k = 0; k < RR; k++) {
            const int dy = min(dy0 + k, a.dh - 1);
            float fy = (float)(((double)dy + 0.5) * a.scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy -= (float)sy;
            b0[k] = 1.0f - fy; b1[k] = fy;
            const int y0 = min(max(sy, 0), a.sh - 1), y1 = min(max(sy + 1, 0), a.sh - 1);
            if (UP) {
                if (k == 0) ymin = y0;
                i0[k] = min(y0 - ymin, NS - 1); i1[k] = min(y1 - ymin, NS - 1);
            } else {
                ys[2 * k] = y0; ys[2 * k + 1] = y1; i0[k] = 2 * k; i1[k] = 2 * k + 1;
            }
        }
        if (UP) {
    #pragma unroll
            for (int r = 0; r < NS; r++) ys[r] = min(ymin + r, a.sh - 1);
        }
        const char* base = reinterpret_cast<const char*>(a.src) + (ptrdiff_t)blockIdx.z * a.spair;
        char* dbase = reinterpret_cast<char*>(a.dst) + (ptrdiff_t)blockIdx.z * a.dpair;
        float v0[NS][RC], v1[NS][RC];
        if constexpr (VEC) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef short s4u __attribute__((ext_vector_type(4), aligned(2)));
            float w[NS][4];
            const unsigned eo = (unsigned)e0 * (IS16 ? 2u : 4u);
    #pragma unroll
            for (int r = 0; r < NS; r++) {
                const char* row = base + (ptrdiff_t)__builtin_amdgcn_readfirstlane(ys[r]) * a.sstride + eo;
                if (IS16) {
                    const s4u q = *reinterpret_cast<const s4u*>(row);
    #pragma unroll
                    for (int e = 0; e < 4; e++) w[r][e] = (float)q[e];
                } else {
                    const f4u q = *reinterpret_cast<const f4u*>(row);
    #pragma unroll
                    for (int e = 0; e < 4; e++) w[r][e] = q[e];
                }
            }
            auto pick = [](const float (&x)[4], int j) -> float {
                float r = x[0];
                r = j == 1 ? x[1] : r; r = j == 2 ? x[2] : r; r = j == 3 ? x[3] : r;
                return r;
            };
    #pragma unroll
            for (int r = 0; r < NS; r++)
    #pragma unroll
                for (int c = 0; c < RC; c++) { v0[r][c] = pick(w[r], j0[c]); v1[r][c] = pick(w[r], j1[c]); }
        } else
    #pragma unroll
        for (int r = 0; r < NS; r++) {
            const char* row = base + (ptrdiff_t)__builtin_amdgcn_readfirstlane(ys[r]) * a.sstride;
    #pragma unroll
            for (int c = 0; c < RC; c++) {
                if (IS16) {
                    v0[r][c] = (float)*reinterpret_cast<const int16_t*>(row + o0[c]);
                    v1[r][c] = (float)*reinterpret_cast<const int16_t*>(row + o1[c]);
                } else {
                    v0[r][c] = *reinterpret_cast<const float*>(row + o0[c]);
                    v1[r][c] = *reinterpret_cast<const float*>(row + o1[c]);
                }
            }
        }
        float hr[NS][RC];
    #pragma unroll
        for (int r = 0; r < NS; r++) {
            const bool rin = !a.zero_outside || (ys[r] >= a.vy0 && ys[r] < a.vy1);
    #pragma unroll
            for (int c = 0; c < RC; c++) {
                const float p = (rin && in0[c]) ? v0[r][c] : 0.0f, q = (rin && in1[c]) ? v1[r][c] : 0.0f;
                hr[r][c] = p * a0[c] + q * a1[c];
            }
        }
    #pragma unroll
        for (int k = 0; k < RR; k++) {
            const int dy = dy0 + k;
            const bool row_on = dy < a.dh;                                // (no break: the loop must unroll, its indices are static)
            float v[RC];
    #pragma unroll
            for (int c = 0; c < RC; c++) {
                float r0 = hr[0][c], r1 = hr[0][c];
                if (UP) {
                    // (source rows advance by at most one per destination row: row k uses tile rows <= k and <= k + 1)
    #pragma unroll
                    for (int r = 1; r <= k; r++) r0 = i0[k] == r ? hr[r][c] : r0;
    #pragma unroll
                    for (int r = 1; r <= k + 1; r++) r1 = i1[k] == r ? hr[r][c] : r1;
                } else {
                    r0 = hr[2 * k][c]; r1 = hr[2 * k + 1][c];
                }
                v[c] = r0 * b0[k] + r1 * b1[k];
            }
            if (!row_on) continue;
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
                {
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    const v2u o = {(unsigned)(unsigned short)q[0] | ((unsigned)(unsigned short)q[1] << 16),
                                   (unsigned)(unsigned short)q[2] | ((unsigned)(unsigned short)q[3] << 16)};
                    __builtin_nontemporal_store(o, reinterpret_cast<v2u*>(d));   // (read next from HBM anyway: -7 % on the kernels)
                }
                else {
    #pragma unroll
                    for (int c = 0; c < RC; c++) if (dx0 + c < a.dw) d[c] = q[c];
                }
            } else {
                float* d = reinterpret_cast<float*>(drow) + dx0;
                if (full && (reinterpret_cast<uintptr_t>(d) & 15u) == 0) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f o = {v[0], v[1], v[2], v[3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(d));
                }
                else {
    #pragma unroll
                    for (int c = 0; c < RC; c++) if (dx0 + c < a.dw) d[c] = v[c];
                }
            }
        }
    }
}

} // namespace

hipError_t launch_resize_linear(const ResizeArgs& a, int n_pairs, hipStream_t st)
{
    if (a.sw <= 0 || a.sh <= 0 || a.dw <= 0 || a.dh <= 0 || n_pairs <= 0) return hipErrorInvalidValue;
    dim3 grid((a.dw + 256 * RC - 1) / (256 * RC), (a.dh + RR * TY - 1) / (RR * TY), n_pairs);
    const bool up = a.scale_y <= 0.75;                               // RR destination rows within RR + 1 source rows, with a margin for float rounding
    const bool vec = up && a.scale_x <= 0.6 && a.sw >= 4;            // RC destination columns within 4 source columns
    if (a.is16) {
        if (vec) hipLaunchKernelGGL((resize_linear_kernel<true, true, true>), grid, dim3(256), 0, st, a);
        else if (up) hipLaunchKernelGGL((resize_linear_kernel<true, true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((resize_linear_kernel<true, false, false>), grid, dim3(256), 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((resize_linear_kernel<false, true, true>), grid, dim3(256), 0, st, a);
        else if (up) hipLaunchKernelGGL((resize_linear_kernel<false, true, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((resize_linear_kernel<false, false, false>), grid, dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

} // namespace adf
