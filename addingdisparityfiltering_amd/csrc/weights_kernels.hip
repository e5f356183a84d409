// weights_kernels.hip -- Fast Global Smoother edge weights for gfx950.
//
//   Chor[i][j]  = LUT[ sum_c (g[i][j][c]-g[i][j+1][c])^2 ], 0 in the last column  (FGS.cpp:586-616)
//   Cvert[i][j] = LUT[ sum_c (g[i][j][c]-g[i+1][j][c])^2 ], 0 in the last row     (FGS.cpp:618-661)
//   LUT[k]      = -expf(-sqrtf(k)/sigma) is built on the HOST with libm and uploaded
//                 (FGS.cpp:663-675): a device expf differs in the last bits and the error would be
//                 amplified by lambda.  768 KB, L2-resident.
//
// One block = one 64 x 32 tile of the guide ROI staged in LDS as bytes; each plane is written
// either in natural [rh][pw] or transposed [rw][ph] orientation (through an LDS tile) so that the
// pass that consumes it reads 256-byte rows.
#include "adf_internal.h"

namespace adf {

namespace {

constexpr int TX = 64, TY = 32, NT = 256;

template <int CH>
__global__ void __launch_bounds__(NT) weights_kernel(WeightArgs a)
{
    __shared__ unsigned char gt[(TY + 1) * (TX + 1) * CH + 16];
    __shared__ float th[TX * (TY + 1)];
    __shared__ float tv[TX * (TY + 1)];
    const Geom& g = a.g;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY; // ROI coordinates
    const size_t pz = blockIdx.z;
    const unsigned char* gp = a.guide + (ptrdiff_t)pz * a.pair_stride;
    constexpr int RB = (TX + 1) * CH; // bytes per staged row

    for (int idx = tid; idx < (TY + 1) * RB; idx += NT) {
        const int rr = idx / RB, b = idx - rr * RB;
        const int c = b / CH, k = b - c * CH;
        const int gi = min(y0 + rr, g.rh - 1), gj = min(x0 + c, g.rw - 1);
        gt[idx] = gp[(ptrdiff_t)(g.ry + gi) * a.stride + (ptrdiff_t)(g.rx + gj) * CH + k];
    }
    __syncthreads();

    float* chor = a.chor + pz * g.plane;
    float* cvert = a.cvert + pz * g.plane;
#pragma unroll
    for (int kk = 0; kk < TY / 4; kk++) {
        const int r = ty + 4 * kk;
        const int i = y0 + r, j = x0 + tx;
        const unsigned char* p = gt + r * RB + tx * CH;
        int hidx = 0, vidx = 0;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int dh = (int)p[c] - (int)p[CH + c];
            const int dv = (int)p[c] - (int)p[RB + c];
            hidx += dh * dh; vidx += dv * dv;
        }
        const bool ok = i < g.rh && j < g.rw;
        float wh = 0.0f, wv = 0.0f;
        if (ok) {
            wh = (j == g.rw - 1) ? 0.0f : a.lut[hidx]; // FGS.cpp:614
            wv = (i == g.rh - 1) ? 0.0f : a.lut[vidx]; // FGS.cpp:658-660
        }
        if (a.chor_orient == ORIENT_N) { if (ok) chor[(size_t)i * g.pw + j] = wh; }
        else th[tx * (TY + 1) + r] = wh;
        if (a.cvert_orient == ORIENT_N) { if (ok) cvert[(size_t)i * g.pw + j] = wv; }
        else tv[tx * (TY + 1) + r] = wv;
    }
    if (a.chor_orient == ORIENT_T || a.cvert_orient == ORIENT_T) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TX / 8; m++) {
            const int cidx = tid / TY + 8 * m, ridx = tid % TY;
            const int jj = x0 + cidx, ii = y0 + ridx;
            if (jj < g.rw && ii < g.rh) {
                const size_t o = (size_t)jj * g.ph + ii;
                if (a.chor_orient == ORIENT_T) chor[o] = th[cidx * (TY + 1) + ridx];
                if (a.cvert_orient == ORIENT_T) cvert[o] = tv[cidx * (TY + 1) + ridx];
            }
        }
    }
}

} // namespace

hipError_t launch_weights(const WeightArgs& a, int n_pairs, hipStream_t st)
{
    dim3 grid((a.g.rw + TX - 1) / TX, (a.g.rh + TY - 1) / TY, n_pairs);
    if (a.ch == 1) hipLaunchKernelGGL(weights_kernel<1>, grid, dim3(NT), 0, st, a);
    else if (a.ch == 3) hipLaunchKernelGGL(weights_kernel<3>, grid, dim3(NT), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace adf
