// weights_kernels.hip -- Fast Global Smoother edge weights for gfx950.
//
//   Chor[i][j]  = LUT[ sum_c (g[i][j][c]-g[i][j+1][c])^2 ], 0 in the last column  (FGS.cpp:586-616)
//   Cvert[i][j] = LUT[ sum_c (g[i][j][c]-g[i+1][j][c])^2 ], 0 in the last row     (FGS.cpp:618-661)
//   LUT[k]      = -expf(-sqrtf(k)/sigma) is built on the HOST with libm and uploaded
//                 (FGS.cpp:663-675): a device expf differs in the last bits and the error would be
//                 amplified by lambda.  768 KB, L2-resident.
//
// One block = one 64 x 32 tile of the guide ROI staged in LDS (aligned dword loads); Cvert is written
// row-major [rh][pw]; Chor row-major for the wave solver or transposed [rw][ph] (through an LDS tile)
// for the exact solver, whose horizontal sweep wants the row index fastest.
#include "adf_internal.h"
#include "prep_bodies.h"
#include <cstdlib>

namespace adf {

namespace {

// streaming outputs are written once and read by a later kernel after gigabytes of other traffic
#define ADF_ST(p, v) __builtin_nontemporal_store((v), (p))
constexpr int TX = 64, TY = 32, NT = 256;
constexpr int LUT_HEAD = 2048;

template <int CH>
__global__ void __launch_bounds__(NT) weights_kernel(WeightArgs a)
{
    // guide tile staged as 32-bit words: each staged row starts at the 4-byte boundary at or below its
    // first needed byte (`mis[r]` = bytes skipped) so the global loads are aligned dword loads
    constexpr int RB = (TX + 1) * CH;              // bytes needed per staged row
    constexpr int RW = (RB + 3) / 4 + 1;           // words per staged row
    __shared__ unsigned gw[(TY + 1) * RW];
    __shared__ int mis[TY + 1];
    __shared__ float th[TX * (TY + 1)];
    // the head of the LUT (small colour differences: the overwhelmingly common case) is cached in LDS:
    // a 64-lane gather from the global table costs one L1 tag lookup per distinct line, the same
    // gather from LDS a few cycles.  Larger indices fall back to the global table.
    __shared__ float lut_head[LUT_HEAD];
    const Geom& g = a.g;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY; // ROI coordinates
    const size_t pz = blockIdx.z;
    const unsigned char* gp = a.guide + (ptrdiff_t)pz * a.pair_stride;
    // bytes actually needed from each row: pixels x0 .. min(x0+TX, rw-1) (the right neighbour of the
    // ROI's last column is never used: its weight is forced to 0, FGS.cpp:614)
    const int last_px = min(x0 + TX, g.rw - 1);
    const int need = (last_px - x0 + 1) * CH;

    for (int rr = ty; rr < TY + 1; rr += NT / TX) {
        const int gi = min(y0 + rr, g.rh - 1);
        const unsigned char* rowp = gp + (ptrdiff_t)(g.ry + gi) * a.stride + (ptrdiff_t)(g.rx + x0) * CH;
        const int m = (int)(reinterpret_cast<uintptr_t>(rowp) & 3u);
        const unsigned* wp = reinterpret_cast<const unsigned*>(rowp - m);
        const int nw = (m + need + 3) >> 2;
        for (int w = tx; w < nw; w += TX) gw[rr * RW + w] = wp[w];
        if (tx == 0) mis[rr] = m;
    }
    for (int q = tid; q < LUT_HEAD; q += NT) lut_head[q] = a.lut[q];
    __syncthreads();

    const unsigned char* gb = reinterpret_cast<const unsigned char*>(gw);
    float* chor = a.chor + pz * g.plane;
    float* cvert = a.cvert + pz * g.plane;
    // Everything up to the stores is unconditional: staged bytes outside the ROI are stale LDS, but any
    // byte triple still indexes inside the 3*255^2+1-entry table, and their results are never stored.
    // Batched that way the LDS reads and table gathers of a thread's 8 pixels overlap instead of each
    // waiting for the previous pixel's.
    int hidx[TY / 4], vidx[TY / 4];
#pragma unroll
    for (int kk = 0; kk < TY / 4; kk++) {
        const int r = ty + 4 * kk;
        const unsigned char* p = gb + r * (RW * 4) + mis[r] + tx * CH;
        const unsigned char* pd = gb + (r + 1) * (RW * 4) + mis[r + 1] + tx * CH;
        int hi = 0, vi = 0;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int v = p[c];
            const int dh = v - (int)p[CH + c];
            const int dv = v - (int)pd[c];
            hi += dh * dh; vi += dv * dv;
        }
        hidx[kk] = hi; vidx[kk] = vi;
    }
    float wh[TY / 4], wv[TY / 4];
    bool big = false;
#pragma unroll
    for (int kk = 0; kk < TY / 4; kk++) {
        wh[kk] = lut_head[min(hidx[kk], LUT_HEAD - 1)];
        wv[kk] = lut_head[min(vidx[kk], LUT_HEAD - 1)];
        big = big || hidx[kk] >= LUT_HEAD || vidx[kk] >= LUT_HEAD;
    }
    if (big) { // rare: strong colour edges index past the cached head of the table
#pragma unroll
        for (int kk = 0; kk < TY / 4; kk++) {
            if (hidx[kk] >= LUT_HEAD) wh[kk] = a.lut[hidx[kk]];
            if (vidx[kk] >= LUT_HEAD) wv[kk] = a.lut[vidx[kk]];
        }
    }
    const int j = x0 + tx;
#pragma unroll
    for (int kk = 0; kk < TY / 4; kk++) {
        const int r = ty + 4 * kk;
        const int i = y0 + r;
        const bool ok = i < g.rh && j < g.rw;
        const float h = (j == g.rw - 1) ? 0.0f : wh[kk];   // FGS.cpp:614
        const float v = (i == g.rh - 1) ? 0.0f : wv[kk];   // FGS.cpp:658-660
        if (a.chor_orient == ORIENT_N) { if (ok) chor[(size_t)i * g.pw + j] = h; }
        else th[tx * (TY + 1) + r] = ok ? h : 0.0f;
        if (ok) cvert[a.cvert_orient == ORIENT_STRIP ? strip_index(i, j, g.rh) : (size_t)i * g.pw + j] = v;   // row-major, or the wave solver's strips
    }
    if (a.chor_orient == ORIENT_T) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TX / 8; m++) {
            const int cidx = tid / TY + 8 * m, ridx = tid % TY;
            const int jj = x0 + cidx, ii = y0 + ridx;
            if (jj < g.rw && ii < g.rh) chor[(size_t)jj * g.ph + ii] = th[cidx * (TY + 1) + ridx];
        }
    }
}

// Row-major outputs (wave solver): the column-walking streaming kernel; its body is shared with the merged preparation
// kernel of conf_kernels.hip (prep_bodies.h).
template <int CH>
__global__ void __launch_bounds__(prep::WS_NT) weights_stream_kernel(WeightArgs a)
{
    __shared__ prep::WsShared<CH> sh;
    prep::weights_stream_body<CH>(a, blockIdx.x, blockIdx.y, gridDim.y, blockIdx.z, sh, true);
}

// Row-major [rh][pw] -> transposed [rw][ph] through a 64 x 64 LDS tile, 16-byte accesses on both sides
// (pw and ph are multiples of 64).  Rows past rh are not part of the source plane: they become zeros.
__global__ void __launch_bounds__(256) transpose_n_to_t_kernel(const float* __restrict__ src, float* __restrict__ dst, Geom g)
{
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const size_t pb = (size_t)blockIdx.z * g.plane;
    const int c4 = tid & 15, r = tid >> 4;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int row = r + 16 * m;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + row < g.rh) v = *reinterpret_cast<const float4*>(src + pb + (size_t)(i0 + row) * g.pw + j0 + 4 * c4);
        tile[row][4 * c4 + 0] = v.x; tile[row][4 * c4 + 1] = v.y; tile[row][4 * c4 + 2] = v.z; tile[row][4 * c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int col = r + 16 * m;
        if (j0 + col < g.rw) {
            const float4 v = make_float4(tile[4 * c4 + 0][col], tile[4 * c4 + 1][col], tile[4 * c4 + 2][col], tile[4 * c4 + 3][col]);
            *reinterpret_cast<float4*>(dst + pb + (size_t)(j0 + col) * g.ph + i0 + 4 * c4) = v;
        }
    }
}

} // namespace

hipError_t launch_weights(const WeightArgs& a, int n_pairs, hipStream_t st)
{
    if (a.chor_orient == ORIENT_T && a.cvert_orient == ORIENT_N && a.scratch) {
        // exact solver: the streaming kernel (one pass over the guide at streaming speed) + one tiled
        // transpose of Chor beat the generic tile kernel's scattered look-ups by 3x
        WeightArgs b = a;
        b.chor = a.scratch; b.chor_orient = ORIENT_N; b.scratch = nullptr;
        hipError_t e = launch_weights(b, n_pairs, st);
        if (e != hipSuccess) return e;
        dim3 grid(a.g.pw / 64, a.g.ph / 64, n_pairs);
        hipLaunchKernelGGL(transpose_n_to_t_kernel, grid, dim3(256), 0, st, a.scratch, a.chor, a.g);
        return hipGetLastError();
    }
    if (a.cvert_orient != ORIENT_N && !(a.cvert_orient == ORIENT_STRIP && a.chor_orient == ORIENT_N)) return hipErrorInvalidValue;
    // row-major Chor: the streaming kernel -- unless a block's slice of the guide cannot be put behind one 32-bit buffer
    // descriptor (row strides of a gigabyte: the generic tile kernel below takes those)
    static const bool force_generic = [] { const char* e = getenv("ADF_WEIGHTS_GENERIC"); return e && atoi(e) != 0; }();   // test hook
    if (!force_generic && a.chor_orient == ORIENT_N && a.stride > 0 && a.stride < ((ptrdiff_t)1 << 29) && (size_t)a.g.W * a.ch < ((size_t)1 << 29)) {
        dim3 sgrid((a.g.rw + prep::WS_BCOLS - 1) / prep::WS_BCOLS, 1, n_pairs);
        {
            // rows per block: tall blocks amortise the table load and the extra row; enough blocks for the chip; and few
            // enough rows that a block's slice of the guide stays inside one descriptor (prep_bodies.h)
            int rpb = 128;
            while (rpb > 16 && ((a.g.rh + rpb - 1) / rpb) * (int)(sgrid.x * sgrid.z) < 2048) rpb >>= 1;
            while (rpb > 4 && ((a.g.rh + rpb - 1) / rpb) * (int)(sgrid.x * sgrid.z) < 512) rpb >>= 1;
            while (rpb > 1 && (size_t)(rpb + 1) * (size_t)a.stride >= ((size_t)1 << 30)) rpb >>= 1;
            sgrid.y = (a.g.rh + rpb - 1) / rpb;
        }
        if (a.ch == 1) hipLaunchKernelGGL(weights_stream_kernel<1>, sgrid, dim3(prep::WS_NT), 0, st, a);
        else if (a.ch == 3) hipLaunchKernelGGL(weights_stream_kernel<3>, sgrid, dim3(prep::WS_NT), 0, st, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    dim3 grid((a.g.rw + TX - 1) / TX, (a.g.rh + TY - 1) / TY, n_pairs);
    if (a.ch == 1) hipLaunchKernelGGL(weights_kernel<1>, grid, dim3(NT), 0, st, a);
    else if (a.ch == 3) hipLaunchKernelGGL(weights_kernel<3>, grid, dim3(NT), 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace adf
