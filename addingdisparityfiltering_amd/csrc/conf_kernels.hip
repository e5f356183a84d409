// conf_kernels.hip -- confidence-map half of DisparityWLSFilter::filter for gfx950.
//
//   discontinuity_kernel : DF.cpp:161-194 (box mean / mean of squares on the ROI copy,
//                          BORDER_REFLECT_101) + DF.cpp:343-373 (variance -> roll-off map)
//   lrc_prologue_kernel  : DF.cpp:306-341 (discontinuity-aware left-right check), DF.cpp:209
//                          (x255) and DF.cpp:288-290 (conf*float(disp)), fused; writes the
//                          two right-hand sides of the solve in the orientation the first
//                          pass wants
//   plain_prologue_kernel: source channel -> float right-hand side: the no-confidence path's
//                          float(disp) (DF.cpp:250,257) and FGS.cpp:191-205 (split + convertTo)
//   outside_kernel       : out = 16*(min_disp-1) = -16 and confidence 0 outside the ROI
//                          (DF.cpp:149,254,284; :187-190)
//
// All of this is integer / elementwise float work: HBM-bound, no MFMA.  Arithmetic that must
// match the CPU restatement bit for bit is written as separate roundings (contraction off).
#include "adf_internal.h"
#include <mutex>
#include <cstdlib>
#include "prep_bodies.h"

#pragma clang fp contract(off)

namespace adf {

namespace {

// streaming outputs are written once and read by a later kernel after gigabytes of other traffic
#define ADF_ST(p, v) __builtin_nontemporal_store((v), (p))
constexpr int TX = 64; // tile width  (one wavefront wide: 128-byte int16 rows, 256-byte float rows)
constexpr int TY = 32; // tile height
constexpr int NT = 256;
constexpr int MAX_RADIUS = 40;

// ---------------------------------------------------------------------------------------
// Depth-discontinuity maps.  One block = one TX x TY tile of ROI outputs of one view of one pair
// (blockIdx.z = 2*pair + view).  LDS: int16 input tile with halo, then horizontal window sums
// (int32 sum, int64 sum of squares; both exact), then vertical window sums, both by sliding
// windows of 8 outputs.  No integer division anywhere: all index maps are 2-D loops.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) discontinuity_kernel(DiscArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int r = a.radius, k = 2 * r + 1;
    const int IH = TY + 2 * r, IW = TX + 2 * r;
    const int IWP = IW | 1;                                              // odd pitch (in int16)
    long long* h2 = reinterpret_cast<long long*>(smem);                  // [IH][TX]
    int* h1 = reinterpret_cast<int*>(h2 + (size_t)IH * TX);              // [IH][TX]
    int16_t* in = reinterpret_cast<int16_t*>(h1 + (size_t)IH * TX);      // [IH][IWP]

    const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid / TX;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int view = a.only_view >= 0 ? a.only_view : (int)(blockIdx.z & 1);
    const size_t pz = a.only_view >= 0 ? blockIdx.z : (blockIdx.z >> 1);
    const char* base = reinterpret_cast<const char*>(a.disp[view]) + (ptrdiff_t)pz * a.pair_stride[view];
    const int rx = a.rx[view];

    for (int yy = ty; yy < IH; yy += NT / TX) {
        const int gy = reflect101(y0 + yy - r, a.rh);
        const int16_t* row = reinterpret_cast<const int16_t*>(base + (ptrdiff_t)(a.ry + gy) * a.stride[view]) + rx;
        for (int xx = tx; xx < IW; xx += TX)
            in[yy * IWP + xx] = row[reflect101(x0 + xx - r, a.rw)];
    }
    __syncthreads();

    // horizontal sums: work item = (row, segment of 8 outputs); 8 segments per row
    for (int item = tid; item < IH * 8; item += NT) {
        const int yy = item >> 3, xs = (item & 7) * 8;
        const int16_t* p = in + yy * IWP + xs;
        int s1 = 0; long long s2 = 0;
        for (int q = 0; q < k; q++) { const int v = p[q]; s1 += v; s2 += (long long)(v * v); }
        int* o1 = h1 + yy * TX + xs; long long* o2 = h2 + yy * TX + xs;
        o1[0] = s1; o2[0] = s2;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            const int va = p[i + k - 1], vb = p[i - 1];
            s1 += va - vb; s2 += (long long)(va * va - vb * vb);
            o1[i] = s1; o2[i] = s2;
        }
    }
    __syncthreads();

    // vertical sums: thread = (column, segment of 8 rows)
    {
        const int ys = ty * 8;
        const double scale = 1.0 / ((double)k * (double)k);
        int s1 = 0; long long s2 = 0;
        for (int q = 0; q < k; q++) { s1 += h1[(ys + q) * TX + tx]; s2 += h2[(ys + q) * TX + tx]; }
        float* dst = a.dst[view] + pz * a.frame;
        const int gx = x0 + tx;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i > 0) {
                s1 += h1[(ys + i + k - 1) * TX + tx] - h1[(ys + i - 1) * TX + tx];
                s2 += h2[(ys + i + k - 1) * TX + tx] - h2[(ys + i - 1) * TX + tx];
            }
            const int gy = y0 + ys + i;
            if (gy < a.rh && gx < a.rw) {
                const float mean = (float)((double)s1 * scale);
                const float sq = (float)((double)s2 * scale);
                const float variance = sq - mean * mean;       // DF.cpp:369
                const float v = 1.0f - a.roll_off * variance;  // DF.cpp:370
                dst[(size_t)(a.ry + gy) * a.W + rx + gx] = v < 0.0f ? 0.0f : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Depth-discontinuity maps, fast path for radius <= 8 (every radius the matcher factories derive:
// ceil(0.33*wsize) / ceil(0.5*wsize), DF.cpp:402,408; default 5).  One block walks down a strip of
// 256-2R output columns: every row is loaded once (coalesced), exchanged through a double-buffered
// LDS row, reduced horizontally by each thread (2R+1 LDS reads), and folded into vertical running
// sums whose 2R+1-deep history lives in registers (the row loop is unrolled by the window height so
// ring slots are compile-time).  All sums are exact 32-bit integers: the sum of squares is kept as
// sum(d^2 >> 16) and sum(d^2 & 0xffff) and recombined in double.
// ---------------------------------------------------------------------------------------
// Output rows per block of the column-walking kernels: gridDim.y row blocks share the ROI's rows.
// The launchers use 128 rows per block for big batches (halo rows and the LUT / prefetch ramp are
// amortised) and fewer when the whole launch would otherwise be too few blocks to fill 256 CUs
// (single-image latency).
#ifndef ADF_CONF_GROUP
#define ADF_CONF_GROUP 8 // rows per prefetch / gather group of the column-walking kernels (measured: 8 beats 16 and 32:
                         // fewer registers -> more resident waves matters more than prefetch depth)
#endif
#ifndef ADF_LRC_GROUP
#define ADF_LRC_GROUP 4  // same for the fused left-view kernel (it also carries the gathered dR / cR per row)
#endif
#define DC_ROWS ((rows_total + (int)gridDim.y - 1) / (int)gridDim.y)

inline int row_blocks(int rows, int blocks_xz)
{
    int rpb = 128;
    while (rpb > 16 && ((rows + rpb - 1) / rpb) * blocks_xz < 2048) rpb >>= 1;
    return (rows + rpb - 1) / rpb;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load (s_waitcnt vmcnt(0)), which would serialise the row prefetch and the LRC gathers of the
// column-walking kernels behind each row's barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exact window sums of an int16 map from 32-bit full-rate integer operations only.  d*d (<= 2^30) is
// formed once per pixel (24-bit multiply: int16 operands) and staged next to d; a tap of the horizontal
// sum is one 8-byte LDS read and three adds -- d, and the two 16-bit halves of d*d, which the sub-dword
// operand selectors take apart for free.  (Sums of the halves stay below 2^31 for any window the column
// kernels support; the square sum is reassembled in double, where it is exact.)  64-bit integer adds
// and shifts would need fewer instructions but issue at a quarter of the rate.
struct WinSum {
    int s1, lo, hi;
    __device__ __forceinline__ void clear() { s1 = 0; lo = 0; hi = 0; }
    __device__ __forceinline__ void tap(const int2 v) { s1 += v.x; lo += v.y & 0xffff; hi += (int)((unsigned)v.y >> 16); }
    __device__ __forceinline__ void add(const WinSum& o) { s1 += o.s1; lo += o.lo; hi += o.hi; }
    __device__ __forceinline__ void sub(const WinSum& o) { s1 -= o.s1; lo -= o.lo; hi -= o.hi; }
    __device__ __forceinline__ double sum2() const { return (double)hi * 65536.0 + (double)lo; }
    __device__ static __forceinline__ int2 stage(int d) { return make_int2(d, __mul24(d, d)); }
};

template <int RT>
__global__ void __launch_bounds__(NT) discontinuity_col_kernel(DiscArgs a)
{
    constexpr int K = 2 * RT + 1;
    constexpr int OUTW = NT - 2 * RT;
    __shared__ int2 rowbuf[2][NT];
    const int tid = threadIdx.x;
    const int rows_total = a.rh;
    const int view = a.only_view >= 0 ? a.only_view : (int)(blockIdx.z & 1);
    const size_t pz = a.only_view >= 0 ? blockIdx.z : (blockIdx.z >> 1);
    const int x_out0 = blockIdx.x * OUTW, y_out0 = blockIdx.y * DC_ROWS;
    const char* base = reinterpret_cast<const char*>(a.disp[view]) + (ptrdiff_t)pz * a.pair_stride[view];
    const int rx = a.rx[view];
    const int gx_in = reflect101(x_out0 - RT + tid, a.rw);       // input column of this thread
    const int gx_out = x_out0 + tid - RT;                        // output column (threads RT .. NT-RT-1)
    const bool writer = tid >= RT && tid < NT - RT && gx_out < a.rw;
    float* dst = a.dst[view] + pz * a.frame + (size_t)a.ry * a.W + rx + gx_out;
    const int nrows = min(DC_ROWS, a.rh - y_out0) + 2 * RT;      // input rows this block consumes
    const double scale = 1.0 / ((double)K * (double)K);

    auto load = [&](int n) -> int {                              // input row n of the block (reflected)
        const int gy = reflect101(y_out0 - RT + n, a.rh);
        return reinterpret_cast<const int16_t*>(base + (ptrdiff_t)(a.ry + gy) * a.stride[view])[rx + gx_in];
    };

    // rows are consumed in groups of U (a multiple of the window height K, so that ring slots stay
    // compile-time) and the next group's U loads are in flight while the current group is reduced
    constexpr int U = K * ((ADF_CONF_GROUP + K - 1) / K);
    WinSum ring[K], S;
    S.clear();
    int nxt[U];
#pragma unroll
    for (int s = 0; s < U; s++) nxt[s] = (s < nrows) ? load(s) : 0;
    for (int n0 = 0; n0 < nrows; n0 += U) {
        int cur[U];
#pragma unroll
        for (int s = 0; s < U; s++) { cur[s] = nxt[s]; asm volatile("" : "+v"(cur[s])); }   // the group's one wait happens here
#pragma unroll
        for (int s = 0; s < U; s++) nxt[s] = (n0 + U + s < nrows) ? load(n0 + U + s) : 0;          // in flight across the rows below
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int n = n0 + s;
            const int slot = s % K;                               // compile-time after unrolling
            if (n < nrows) {                                     // block-uniform
                rowbuf[n & 1][tid] = WinSum::stage(cur[s]);
                lds_barrier();
                WinSum hsum;
                hsum.clear();
                if (tid >= RT && tid < NT - RT) {
#pragma unroll
                    for (int d = -RT; d <= RT; d++) hsum.tap(rowbuf[n & 1][tid + d]);
                }
                if (n >= K) S.sub(ring[slot]);                    // row n-K leaves the window
                ring[slot] = hsum;
                S.add(hsum);
                if (n >= 2 * RT && writer) {                     // window rows n-2RT..n are complete
                    const int oy = y_out0 + n - 2 * RT;
                    const float mean = (float)((double)S.s1 * scale);
                    const float sq = (float)(S.sum2() * scale);
                    const float variance = sq - mean * mean;      // DF.cpp:369
                    const float v = 1.0f - a.roll_off * variance; // DF.cpp:370
                    ADF_ST(&dst[(size_t)oy * a.W], v < 0.0f ? 0.0f : v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Left view, fused: the column-walking discontinuity sweep above, and for every completed window the
// left-right check against the (already finished) right map, x255, and the confidence map written
// straight away -- the left discontinuity map never exists in memory (DF.cpp:204-209 in one pass).
// With WRITE_U the right-hand sides conf*float(dL), conf (DF.cpp:288-290) are written as well.
// ---------------------------------------------------------------------------------------
template <int RT, bool WRITE_U>
__global__ void __launch_bounds__(NT) conf_left_kernel(ConfLeftArgs a)
{
    constexpr int K = 2 * RT + 1;
    constexpr int OUTW = NT - 2 * RT;
    constexpr int U = K * ((ADF_LRC_GROUP + K - 1) / K);
    constexpr int CR = RT > 0 ? RT : 1;
    __shared__ int2 rowbuf[2][NT];
    const Geom& g = a.g;
    const int tid = threadIdx.x;
    const int rows_total = g.rh;
    const size_t pz = blockIdx.z;
    const int x_out0 = blockIdx.x * OUTW, y_out0 = blockIdx.y * DC_ROWS;
    const char* baseL = reinterpret_cast<const char*>(a.dL) + (ptrdiff_t)pz * a.psL;
    const char* baseR = reinterpret_cast<const char*>(a.dR) + (ptrdiff_t)pz * a.psR;
    const float* cR = a.cR + pz * g.frame;
    float* conf = a.conf + pz * g.cframe + g.cx0;
    const int gx_in = reflect101(x_out0 - RT + tid, g.rw);
    const int gx_out = x_out0 + tid - RT;                          // == gx_in for writer threads
    const bool writer = tid >= RT && tid < NT - RT && gx_out < g.rw;
    const int j_abs = g.rx + gx_out;                              // frame column of this thread's output
    const int rows_out = min(DC_ROWS, g.rh - y_out0);
    const int nrows = rows_out + 2 * RT;                          // input rows this block consumes
    const double scale = 1.0 / ((double)K * (double)K);
    const int right_end = a.rrx + g.rw;

    auto load = [&](int n) -> int {
        const int gy = reflect101(y_out0 - RT + n, g.rh);
        return reinterpret_cast<const int16_t*>(baseL + (ptrdiff_t)(g.ry + gy) * a.sL)[g.rx + gx_in];
    };

    // The LRC gathers (dR and cR at column j - d/16 of the same row) depend on the disparity of the
    // window's CENTRE row, which this thread loaded itself RT rows before the window completes.  They
    // are issued for a whole group of rows as soon as the group's values have arrived and consumed as
    // the windows complete, so a group pays one memory latency instead of two per row.
    WinSum ring[K], S;
    S.clear();
    int nxt[U], cur[U];
    int gd[U]; float gc[U];                                       // gathered dR / cR of this group's centre rows
    int cd[CR], cdr[CR]; float ccr[CR];                           // carried over: last RT centre rows of the previous group
#pragma unroll
    for (int k = 0; k < CR; k++) { cd[k] = 0; cdr[k] = 0; ccr[k] = 0.0f; }
#pragma unroll
    for (int s = 0; s < U; s++) nxt[s] = (s < nrows) ? load(s) : 0;
    for (int n0 = 0; n0 < nrows; n0 += U) {
#pragma unroll
        for (int s = 0; s < U; s++) { cur[s] = nxt[s]; asm volatile("" : "+v"(cur[s])); }
#pragma unroll
        for (int s = 0; s < U; s++) {                             // gathers for centre rows n0 .. n0+U-1
            const int n = n0 + s;
            gd[s] = 0; gc[s] = 0.0f;
            if (writer && n >= RT && n < nrows - RT) {
                const int ridx = j_abs - (cur[s] >> 4);           // DF.cpp:331
                if (ridx >= a.rrx && ridx < right_end) {
                    const int i_abs = g.ry + y_out0 + n - RT;
                    gd[s] = reinterpret_cast<const int16_t*>(baseR + (ptrdiff_t)i_abs * a.sR)[ridx];
                    gc[s] = cR[(size_t)i_abs * g.W + ridx];
                }
            }
        }
        // One wait per group, placed by hand: the gathers have to be back before the first window of the
        // group completes, and behind the block-uniform branches of the row loop the compiler can only wait
        // for vmcnt(0) at every use -- which, stores counting in vmcnt here, also waits for the previous
        // row's store to be acknowledged.  After this point the group's inputs are plain registers; the next
        // group's rows are requested only now, so they and the stores stay in flight across the row loop.
#pragma unroll
        for (int s = 0; s < U; s++) asm volatile("" : "+v"(gd[s]), "+v"(gc[s]));
#pragma unroll
        for (int s = 0; s < U; s++) nxt[s] = (n0 + U + s < nrows) ? load(n0 + U + s) : 0;
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int n = n0 + s;
            const int slot = s % K;
            if (n < nrows) {                                     // block-uniform
                rowbuf[n & 1][tid] = WinSum::stage(cur[s]);
                lds_barrier();
                WinSum hsum;
                hsum.clear();
                if (tid >= RT && tid < NT - RT) {
#pragma unroll
                    for (int d = -RT; d <= RT; d++) hsum.tap(rowbuf[n & 1][tid + d]);
                }
                if (n >= K) S.sub(ring[slot]);
                ring[slot] = hsum;
                S.add(hsum);
                if (n >= 2 * RT && writer) {                     // window centred on input row n-RT is complete
                    const int oy = y_out0 + n - 2 * RT;
                    const int i_abs = g.ry + oy;
                    const float mean = (float)((double)S.s1 * scale);
                    const float sq = (float)(S.sum2() * scale);
                    const float variance = sq - mean * mean;      // DF.cpp:369
                    float c = 1.0f - a.roll_off * variance;       // DF.cpp:370
                    c = c < 0.0f ? 0.0f : c;
                    // centre row's own disparity and its gathers: this group's slot s-RT, or carried over
                    const int d = (s >= RT) ? cur[(s >= RT) ? s - RT : 0] : cd[(s < RT) ? s : 0];
                    const int dr = (s >= RT) ? gd[(s >= RT) ? s - RT : 0] : cdr[(s < RT) ? s : 0];
                    const float b = (s >= RT) ? gc[(s >= RT) ? s - RT : 0] : ccr[(s < RT) ? s : 0];
                    const int ridx = j_abs - (d >> 4);            // DF.cpp:331
                    if (ridx >= a.rrx && ridx < right_end) {
                        if (abs(d + dr) < a.thresh) c = b < c ? b : c;   // DF.cpp:334-335
                        else c = 0.0f;                                   // DF.cpp:337
                    }
                    c = 255.0f * c;                               // DF.cpp:209
                    ADF_ST(&conf[(size_t)i_abs * g.cpitch + j_abs], c);
                    if (WRITE_U) {
                        const size_t o = pz * 2 * g.plane + pair_index(oy, gx_out, g.pw);   // ORIENT_PAIR
                        a.U0[o] = c * (float)d;                   // DF.cpp:289-290
                        a.U0[o + ADF_STRIP] = c;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RT; k++) { cd[k] = cur[U - RT + k]; cdr[k] = gd[U - RT + k]; ccr[k] = gc[U - RT + k]; }
    }
}

// ---------------------------------------------------------------------------------------
// Both views in ONE sweep (DF.cpp:197-210 whole: 161-194 + 343-373 for both views, 306-341, :209): the right
// view's discontinuity map never exists in HBM.  A workgroup owns a band of rows over the FULL ROI width; every
// row step it forms the window statistics of both views, parks the right view's map of that row in LDS (16 KB
// at 4K) together with the raw right disparities of the last few rows, and the left-view lanes gather from
// there -- any column of the row, so the LRC is exact for arbitrary disparities without a fall-back path.
// HBM sees dL and dR once (plus 2*RT halo rows per band) and the confidence map: 8 B per ROI pixel.
//
// Layout of the work: lane = 4 adjacent columns (one 8-byte load per view and row), wave = 64 lanes = 248
// output columns for radius <= 4, 240 for radius 5..8 (the first and last ceil(radius/4) lanes only supply the
// horizontal halo, so a wave never needs another wave's sums).  Sums run VERTICAL first: each lane keeps the raw values of the last K = 2*RT+1 rows of its
// columns packed in registers and updates the column sums (sum d, and sum d^2 split in 16-bit halves: all
// 32-bit integer, exact) by new row minus oldest row; the horizontal window is then a sliding sum over the
// lane's own four column sums and the neighbours' edge columns, fetched with whole-wave DPP shifts (no LDS, no
// barrier).  Virtual columns: the row is extended by RT reflected columns on both sides (BORDER_REFLECT_101
// inside the ROI copy, DF.cpp:167-185) and the lanes at the two image edges load the reflected real columns
// (one 8-byte window + a byte permute), so the horizontal window itself never needs a border case.
// One workgroup barrier per row (right map written -> gathered).
// ---------------------------------------------------------------------------------------
constexpr int CB_COLS = 4;
#define CB_HALO(RT) (((RT) + CB_COLS - 1) / CB_COLS)                 // halo lanes on each side of a wave
#define CB_WOUT(RT) ((64 - 2 * CB_HALO(RT)) * CB_COLS)               // output columns per wave: 248 (radius <= 4), 240 (5..8)
#define CB_RLDS(RT) ((RT) >= 4)                                     // right view's row ring in LDS by default (conf_band_body)
constexpr int CB_MAX_WAVES = 16;
constexpr int CB_MAX_BAND_ROWS = 2048;                             // output rows of a band (ColSum::lo must not wrap)
constexpr int CB_MAX_RADIUS = 2 * CB_COLS;

__device__ __forceinline__ int dpp_from_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }  // wave_shr:1
__device__ __forceinline__ int dpp_from_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }  // wave_shl:1

struct ColSum {
    int s1; unsigned lo; int hi;
    // d enters / leaves the column window.  sum d^2 = 65536 * hi + lo exactly, carried as two 32-bit halves so that the
    // horizontal sums need no carries: the change q_new - q_old is split into its low 16 bits (added to lo: never
    // negative, so lo only grows -- by less than 65536 per row, which is why a band is at most CB_MAX_BAND_ROWS rows:
    // 17 columns x 2064 rows x 65535 < 2^32) and its arithmetic shift by 16 (added to hi).  Round 3: 7 instructions
    // instead of the 10 of summing (q & 0xffff) and (q >> 16) of both squares separately.
    __device__ __forceinline__ void slide(int dn, int dold)
    {
        const int qn = __mul24(dn, dn), qo = __mul24(dold, dold);
        s1 += dn - dold;
        const int diff = qn - qo;
        lo += (unsigned)diff & 0xffffu;
        hi += diff >> 16;
    }
};

__device__ __forceinline__ float disc_value(int s1, unsigned lo, int hi, double scale, float roll_off)
{
    const float mean = (float)((double)s1 * scale);
    // (65536 * hi + lo is an integer below 2^53: the fused form is exact, like the product and the sum it replaces)
    const float sq = (float)(__builtin_fma((double)hi, 65536.0, (double)lo) * scale);
    const float variance = sq - mean * mean;          // DF.cpp:369
    const float v = 1.0f - roll_off * variance;       // DF.cpp:370
    return v < 0.0f ? 0.0f : v;
}

// The four map values of a lane's columns from its column sums.  Horizontal window of output column q (0..3): virtual
// columns q-RT .. q+RT relative to the lane's first column, i.e. the lane's own columns max(0,q-RT) .. min(3,q+RT) --
// differences of the lane's prefix sums -- plus, from the j-th lane before it (columns -4j .. -4j+3), its LAST
// min(4, 4-4j+RT-q) columns and, from the j-th lane after it, its FIRST min(4, q+RT-4j+1) columns; j runs to
// CB_HALO(RT) = ceil(RT/4) (one lane for radius <= 4, two for radius 5..8: round 3).  Every neighbour contribution is
// ONE partial sum fetched by whole-wave DPP shifts (wave_shr:1 / wave_shl:1, applied j times); where a fetched value
// has a single use the compiler folds the shift into the add that consumes it.
template <int J> __device__ __forceinline__ int dpp_prev_n(int v) { if constexpr (J <= 0) return v; else return dpp_prev_n<J - 1>(dpp_from_prev(v)); }
template <int J> __device__ __forceinline__ int dpp_next_n(int v) { if constexpr (J <= 0) return v; else return dpp_next_n<J - 1>(dpp_from_next(v)); }

template <int RT, int J>
__device__ __forceinline__ void band_neighbours(const int (&p1)[CB_COLS + 1], const unsigned (&pl)[CB_COLS + 1], const int (&ph)[CB_COLS + 1],
                                                int q, int& h1, unsigned& hl, int& hh)
{
    if constexpr (J >= 1) {
        const int mp = 4 - 4 * J + RT - q, mn = q + RT - 4 * J + 1;      // columns of lane -J / lane +J inside the window
        const int m = mp < 0 ? 0 : (mp > CB_COLS ? CB_COLS : mp), m2 = mn < 0 ? 0 : (mn > CB_COLS ? CB_COLS : mn);
        if (m > 0) {
            h1 += dpp_prev_n<J>(p1[CB_COLS] - p1[CB_COLS - m]); hl += (unsigned)dpp_prev_n<J>((int)(pl[CB_COLS] - pl[CB_COLS - m]));
            hh += dpp_prev_n<J>(ph[CB_COLS] - ph[CB_COLS - m]);
        }
        if (m2 > 0) { h1 += dpp_next_n<J>(p1[m2]); hl += (unsigned)dpp_next_n<J>((int)pl[m2]); hh += dpp_next_n<J>(ph[m2]); }
        band_neighbours<RT, J - 1>(p1, pl, ph, q, h1, hl, hh);
    }
}

template <int RT>
__device__ __forceinline__ void band_row_values(const ColSum (&V)[CB_COLS], double scale, float roll_off, float (&out)[CB_COLS])
{
    static_assert(RT >= 1 && RT <= 2 * CB_COLS, "the halo must fit two lanes");
    int p1[CB_COLS + 1], ph[CB_COLS + 1];                            // prefix sums P[i] = V[0] + .. + V[i-1]
    unsigned pl[CB_COLS + 1];
    p1[0] = ph[0] = 0; pl[0] = 0u;
#pragma unroll
    for (int i = 0; i < CB_COLS; i++) { p1[i + 1] = p1[i] + V[i].s1; pl[i + 1] = pl[i] + V[i].lo; ph[i + 1] = ph[i] + V[i].hi; }
#pragma unroll
    for (int q = 0; q < CB_COLS; q++) {
        const int a = q - RT > 0 ? q - RT : 0, b = (q + RT < CB_COLS - 1 ? q + RT : CB_COLS - 1) + 1;
        int h1 = p1[b] - p1[a], hh = ph[b] - ph[a];
        unsigned hl = pl[b] - pl[a];
        band_neighbours<RT, CB_HALO(RT)>(p1, pl, ph, q, h1, hl, hh);
        out[q] = disc_value(h1, hl, hh, scale, roll_off);
    }
}

// (band `band` of image `pz`; smem = the workgroup's dynamic LDS.  Waves past the ROI's width own no columns and only
// keep the barriers company: the merged preparation kernel launches at least four waves per block.)
template <int RT, bool RLDS = CB_RLDS(RT)>
__device__ __forceinline__ void conf_band_body(const ConfBandArgs& a, const int band, const size_t pz, unsigned char* smem)
{
    constexpr int K = 2 * RT + 1;
    // RLDS (radius 4..8 wherever K + 1 rows of the ROI's width fit the LDS: always up to radius 6, up to ~3700 columns at
    // radius 7 and ~3300 at radius 8): the right view's K raw rows live in LDS only (they are parked there for the gather
    // anyway) and the row that leaves the column window is read back from it, instead of a second register ring: 18..34
    // registers fewer per lane -- at radius 4..5 that is what lets a wave of the weight kernel sit beside the band's
    // four waves on a SIMD (the radius 1..3 kernels always could), at radius 6..8 it ends the register spills.
    // raw right rows kept in LDS: the centre row + one row of slack for the slowest wave; with RLDS all K rows + the one
    // being written (row n - K is read before this iteration's barrier, row n written before it: never the same slot)
    constexpr int RING = RLDS ? K + 1 : RT + 2;
    typedef int v2i_u __attribute__((ext_vector_type(2), aligned(2)));
    typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
    const Geom& g = a.g;
    const int rw = g.rw, rwp = (rw + 3) & ~3;
    float* crow = reinterpret_cast<float*>(smem);                     // [2][rwp]   right map of the current row
    int16_t* draw = reinterpret_cast<int16_t*>(crow + 2 * rwp);       // [RING][rwp] raw right disparities
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int y0 = band * a.rows_per_band;
    const int rows_out = min(a.rows_per_band, g.rh - y0);
    const int nrows = rows_out + 2 * RT;               // input rows this band consumes
    constexpr int HL = CB_HALO(RT);
    const int vbase = wv * CB_WOUT(RT) + CB_COLS * (lane - HL);       // first (virtual) column of this lane
    const bool out_lane = lane >= HL && lane <= 63 - HL && vbase < rw;  // owns output columns vbase .. vbase+3 (those < rw)

    // which real columns feed this lane's four virtual columns: one 8-byte window [lcol, lcol+4) of the row and a
    // byte permute (identity for every lane away from the two image edges)
    int lcol = vbase;
    unsigned perm0 = 0x03020100u, perm1 = 0x07060504u;
    if (!(vbase >= 0 && vbase + CB_COLS - 1 < rw)) {
        int rv[CB_COLS], lo_need = rw;
#pragma unroll
        for (int q = 0; q < CB_COLS; q++) {
            const int v = vbase + q;
            const int vc = v < -RT ? -RT : (v > rw - 1 + RT ? rw - 1 + RT : v);
            rv[q] = reflect101(vc, rw);
            if (v >= -RT && v <= rw - 1 + RT) lo_need = min(lo_need, rv[q]);
        }
        lcol = lo_need >= rw ? 0 : lo_need;
        lcol = max(0, min(lcol, rw - CB_COLS));
        unsigned sb[CB_COLS];
#pragma unroll
        for (int q = 0; q < CB_COLS; q++) {
            const unsigned e = (unsigned)max(0, min(rv[q] - lcol, CB_COLS - 1));
            sb[q] = (2u * e) | ((2u * e + 1u) << 8);
        }
        perm0 = sb[0] | (sb[1] << 16);
        perm1 = sb[2] | (sb[3] << 16);
    }
    // row base = wave-uniform (scalar registers), lane part = a 32-bit byte offset
    const char* baseL = reinterpret_cast<const char*>(a.dL) + (ptrdiff_t)pz * a.psL + (ptrdiff_t)g.ry * a.sL + (ptrdiff_t)g.rx * 2;
    const char* baseR = reinterpret_cast<const char*>(a.dR) + (ptrdiff_t)pz * a.psR + (ptrdiff_t)g.ry * a.sR + (ptrdiff_t)a.rrx * 2;
    const unsigned lane_off = (unsigned)lcol * 2u;
    float* conf = a.conf + pz * g.cframe + (size_t)g.ry * g.cpitch + g.cx0 + g.rx;   // 16-byte aligned (Geom::cx0)
    const double scale = 1.0 / ((double)K * (double)K);
    const int right_end = a.rrx + rw;

    // packed raw rows: .x = columns (0,1), .y = columns (2,3) of the lane, 16 bits each
    auto load = [&](const char* base, ptrdiff_t stride, int n) -> v2i_u {
        const int nn = n < nrows ? n : nrows - 1;
        int gy = abs(y0 - RT + nn);                                  // BORDER_REFLECT_101 with one reflection:
        gy = gy >= g.rh ? 2 * (g.rh - 1) - gy : gy;                  // the launcher guarantees rh > RT
        return *reinterpret_cast<const v2i_u*>(base + (ptrdiff_t)gy * stride + lane_off);
    };
    auto permute = [&](v2i_u p) -> int2 {
        return make_int2((int)__builtin_amdgcn_perm((unsigned)p.y, (unsigned)p.x, perm0),
                         (int)__builtin_amdgcn_perm((unsigned)p.y, (unsigned)p.x, perm1));
    };
#define CB_ELEM(P, q) ((q) == 0 ? (int)(short)((P).x & 0xffff) : (q) == 1 ? ((P).x >> 16) : (q) == 2 ? (int)(short)((P).y & 0xffff) : ((P).y >> 16))

    int2 ringL[K], ringR[RLDS ? 1 : K];
    ColSum VL[CB_COLS], VR[CB_COLS];
#pragma unroll
    for (int k = 0; k < K; k++) { ringL[k] = make_int2(0, 0); if (!RLDS) ringR[k] = make_int2(0, 0); }
    // (RLDS) an 8-byte LDS read needs an 8-byte aligned address: true for every lane away from the image edges
    // (lcol = vbase, a multiple of 4 columns); waves with an edge lane read 2-byte aligned
    const bool unaligned_wave = RLDS && __builtin_amdgcn_ballot_w64((lcol & 3) != 0) != 0;
#pragma unroll
    for (int q = 0; q < CB_COLS; q++) { VL[q].s1 = VL[q].hi = 0; VL[q].lo = 0u; VR[q].s1 = VR[q].hi = 0; VR[q].lo = 0u; }

    // rows n+1 and n+2 are in flight while row n is reduced
    v2i_u pfL0 = load(baseL, a.sL, 0), pfR0 = load(baseR, a.sR, 0);
    v2i_u pfL1 = load(baseL, a.sL, 1), pfR1 = load(baseR, a.sR, 1);
    for (int n0 = 0; n0 < nrows; n0 += K) {
#pragma unroll
        for (int s = 0; s < K; s++) {
            const int n = n0 + s;
            if (n < nrows) {                                             // block-uniform
                const int2 curL = permute(pfL0), curR = permute(pfR0);
                pfL0 = pfL1; pfR0 = pfR1;
                pfL1 = load(baseL, a.sL, n + 2); pfR1 = load(baseR, a.sR, n + 2);
                // vertical: row n enters the column windows, row n-K leaves (zeros during the first K rows)
                int2 oldR;
                if constexpr (RLDS) {
                    // row n-K of the right view, this lane's four (virtual) columns: the same window + permute as the
                    // global load, from the LDS ring (zeros while the window fills)
                    const int16_t* rp = draw + (n >= K ? (n - K) % RING : 0) * rwp + lcol;
                    v2i_u q2;
                    if (unaligned_wave) q2 = *reinterpret_cast<const v2i_u*>(rp);
                    else { const int2 t = *reinterpret_cast<const int2*>(rp); q2.x = t.x; q2.y = t.y; }
                    oldR = permute(q2);
                    if (n < K) oldR = make_int2(0, 0);
                } else
                    oldR = ringR[s];
#pragma unroll
                for (int q = 0; q < CB_COLS; q++) {
                    VL[q].slide(CB_ELEM(curL, q), CB_ELEM(ringL[s], q));
                    VR[q].slide(CB_ELEM(curR, q), CB_ELEM(oldR, q));
                }
                ringL[s] = curL;
                if constexpr (!RLDS) ringR[s] = curR;
                float cl[CB_COLS], cr[CB_COLS];
                const bool complete = n >= 2 * RT;                      // windows centred on input row n-RT are complete
                if (complete) {                                          // (one basic block: the DPP fetches fold into their adds)
                    band_row_values<RT>(VR, scale, a.roll_off, cr);
                    band_row_values<RT>(VL, scale, a.roll_off, cl);
                }
                if (out_lane) {
                    // raw right row n -> LDS (it is the centre row of the window that completes RT rows from now)
                    *reinterpret_cast<int2*>(draw + (n % RING) * rwp + vbase) = curR;
                    if (complete) *reinterpret_cast<float4*>(crow + (n & 1) * rwp + vbase) = make_float4(cr[0], cr[1], cr[2], cr[3]);
                }
                lds_barrier();
                if (complete && out_lane) {
                    const int oy = y0 + n - 2 * RT;                     // ROI row of the output
                    const int2 cen = ringL[(s + K - RT) % K];            // the centre row's own disparities
                    const float* cb = crow + (n & 1) * rwp;
                    const int16_t* db = draw + ((n - RT) % RING) * rwp;
                    float res[CB_COLS];
#pragma unroll
                    for (int q = 0; q < CB_COLS; q++) {
                        const int d = CB_ELEM(cen, q);
                        const int ridx = g.rx + vbase + q - (d >> 4);    // DF.cpp:331 (frame column of the right view)
                        const bool hit = ridx >= a.rrx && ridx < right_end;  // DF.cpp:332
                        const int rr = hit ? ridx - a.rrx : 0;           // (a miss reads column 0 and discards it)
                        const float b = cb[rr];
                        const int dr = db[rr];
                        const float cmin = b < cl[q] ? b : cl[q];        // DF.cpp:334-335 (std::min)
                        const float chit = abs(d + dr) < a.thresh ? cmin : 0.0f;   // DF.cpp:337
                        res[q] = 255.0f * (hit ? chit : cl[q]);          // DF.cpp:209
                    }
                    float* dst = conf + (size_t)oy * g.cpitch + vbase;
                    if (vbase + CB_COLS <= rw) {
                        const v4f_u o = {res[0], res[1], res[2], res[3]};
                        __builtin_nontemporal_store(o, reinterpret_cast<v4f_u*>(dst));
                    } else {
#pragma unroll
                        for (int q = 0; q < CB_COLS; q++)
                            if (vbase + q < rw) ADF_ST(&dst[q], res[q]);
                    }
                }
            }
        }
    }
#undef CB_ELEM
}

template <int RT, bool RLDS = CB_RLDS(RT)>
__global__ void __launch_bounds__(64 * CB_MAX_WAVES) conf_band_kernel(ConfBandArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    conf_band_body<RT, RLDS>(a, blockIdx.x, blockIdx.y, smem);
}

constexpr int OUT_ROWS = 16; // rows per block of outside_kernel

// Non-ROI pixels of the frame: the side columns of the ROI's rows (part A: W - rw columns x rh rows, blocks of NT
// columns x OUT_ROWS rows) and the whole rows above / below the ROI (part B: (H - rh) x W pixels, a flat index in
// blocks of NT * OUT_ROWS pixels).  blockIdx.x < nA: part A, else part B -- no block without work (round 3: the BM
// factory's ROI has rows outside it, and a grid over the whole frame spent 0.49 ms per 64 x 4K step finding that out).
__device__ __forceinline__ void outside_body(const OutsideArgs& a, const int nAx, const int nA, const unsigned blk, const size_t pz)
{
    const Geom& g = a.g;
    char* out = a.out ? reinterpret_cast<char*>(a.out) + (ptrdiff_t)pz * a.pair_stride : nullptr;
    float* conf = a.conf ? a.conf + pz * g.cframe + g.cx0 : nullptr;
    if ((int)blk < nA) {
        const int bx = blk % nAx, by = blk / nAx;
        const int t = bx * NT + threadIdx.x;
        if (t >= g.W - g.rw) return;
        const int j = t < g.rx ? t : t + g.rw;
        const int i0 = g.ry + by * OUT_ROWS, i1 = min(i0 + OUT_ROWS, g.ry + g.rh);
        for (int i = i0; i < i1; i++) {
            if (out) reinterpret_cast<int16_t*>(out + (ptrdiff_t)i * a.stride)[j] = a.fill;
            if (conf) conf[(size_t)i * g.cpitch + j] = 0.0f;
        }
    } else {
        const size_t npix = (size_t)(g.H - g.rh) * g.W;
        size_t p = (size_t)(blk - nA) * (NT * OUT_ROWS) + threadIdx.x;
#pragma unroll 4
        for (int k = 0; k < OUT_ROWS; k++, p += NT) {
            if (p >= npix) break;
            int i = (int)(p / (size_t)g.W);
            const int j = (int)(p - (size_t)i * g.W);
            if (i >= g.ry) i += g.rh;                        // rows below the ROI
            if (out) reinterpret_cast<int16_t*>(out + (ptrdiff_t)i * a.stride)[j] = a.fill;
            if (conf) conf[(size_t)i * g.cpitch + j] = 0.0f;
        }
    }
}

__global__ void __launch_bounds__(NT) outside_kernel(OutsideArgs a, int nAx, int nA)
{
    outside_body(a, nAx, nA, blockIdx.x, blockIdx.z);
}

// ---------------------------------------------------------------------------------------
// Everything a confidence-mode call prepares before its first solve pass, in ONE launch (round 3): blocks
// [0, nC) are bands of the one-sweep confidence kernel (the longest role first), [nC, nC + nW) blocks of the streaming
// weight kernel, the rest fill what lies outside the ROI.  The three touch disjoint data, so for batches they are
// separate launches on two streams (adf_api.hip); for ONE small frame per call the cross-stream event and launch
// latencies are a third of the call (profiles/r03_latency_timeline.txt), which this kernel removes.  Blocks are as
// wide as the confidence role needs (at least the weight role's 256 threads); waves a role has no use for only keep
// its barriers company.
// ---------------------------------------------------------------------------------------
struct PrepArgs {
    ConfBandArgs c; WeightArgs w; OutsideArgs o;
    int nC, nW, nWx, nWy, nOAx, nOA;
};

template <int CH, int RT>
__global__ void __launch_bounds__(64 * CB_MAX_WAVES) prep_small_kernel(PrepArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ prep::WsShared<CH> ws;
    const unsigned b = blockIdx.x;
    const size_t pz = blockIdx.y;
    if ((int)b < a.nC) {
        conf_band_body<RT>(a.c, (int)b, pz, smem);
    } else if ((int)b < a.nC + a.nW) {
        const int k = (int)b - a.nC;
        prep::weights_stream_body<CH>(a.w, k % a.nWx, k / a.nWx, a.nWy, pz, ws, threadIdx.x < prep::WS_NT);
    } else {
        if (threadIdx.x < NT) outside_body(a.o, a.nOAx, a.nOA, b - (unsigned)(a.nC + a.nW), pz);
    }
}

// ---------------------------------------------------------------------------------------
// LRC + x255 + prologue.  Grid covers the full frame so the confidence plane is written
// exactly once everywhere (zero outside the ROI, DF.cpp:187-190,209).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) lrc_prologue_kernel(LrcArgs a)
{
    __shared__ float t0[TX * (TY + 1)];
    __shared__ float t1[TX * (TY + 1)];
    const Geom& g = a.g;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const size_t pz = blockIdx.z;
    const char* pL = reinterpret_cast<const char*>(a.dL) + (ptrdiff_t)pz * a.psL;
    const char* pR = reinterpret_cast<const char*>(a.dR) + (ptrdiff_t)pz * a.psR;
    const float* cL = a.cL + pz * g.frame;
    const float* cR = a.cR + pz * g.frame;
    float* conf = a.conf + pz * g.cframe + g.cx0;
    const bool pair = a.orient == ORIENT_PAIR;
    float* U0 = a.U0 ? a.U0 + pz * (pair ? 2 : 1) * g.plane : nullptr;   // null: confidence only (down-scaled path)
    float* U1 = a.U0 ? (pair ? U0 + ADF_STRIP : a.U1 + pz * g.plane) : nullptr;
    const int j = x0 + tx;
    const int right_end = a.rrx + g.rw;

    // Three phases over the thread's TY/4 pixels -- own values, the gathers they address, then arithmetic and
    // stores -- so that no loaded value is first used inside the storing loop (stores count in vmcnt on this
    // target: a wait for a load there would also wait for every store issued before it).
    constexpr int NK = TY / 4;
    int dv[NK], drv[NK]; float cv[NK], bv[NK]; bool roi_k[NK], hit[NK];
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        const int i = y0 + ty + 4 * kk;
        roi_k[kk] = i < g.H && j < g.W && j >= g.rx && j < g.rx + g.rw && i >= g.ry && i < g.ry + g.rh;
        dv[kk] = 0; cv[kk] = 0.0f;
        if (roi_k[kk]) {
            dv[kk] = reinterpret_cast<const int16_t*>(pL + (ptrdiff_t)i * a.sL)[j];
            cv[kk] = cL[(size_t)i * g.W + j];
        }
    }
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        const int i = y0 + ty + 4 * kk;
        const int ridx = j - (dv[kk] >> 4);                             // DF.cpp:331
        hit[kk] = roi_k[kk] && ridx >= a.rrx && ridx < right_end;
        drv[kk] = 0; bv[kk] = 0.0f;
        if (hit[kk]) {
            drv[kk] = reinterpret_cast<const int16_t*>(pR + (ptrdiff_t)i * a.sR)[ridx];
            bv[kk] = cR[(size_t)i * g.W + ridx];
        }
    }
#pragma unroll
    for (int kk = 0; kk < NK; kk++) asm volatile("" : "+v"(drv[kk]), "+v"(bv[kk]));   // the one wait for the gathers
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        const int i = y0 + ty + 4 * kk;
        const bool in_frame = i < g.H && j < g.W;
        const bool in_roi = roi_k[kk];
        const int d = dv[kk];
        float c = cv[kk], u0 = 0.0f;
        if (in_roi) {
            if (hit[kk]) {
                if (abs(d + drv[kk]) < a.thresh) c = bv[kk] < c ? bv[kk] : c;   // DF.cpp:334-335 (std::min)
                else c = 0.0f;                                                  // DF.cpp:337
            }
            c = 255.0f * c;                                             // DF.cpp:209
            u0 = c * (float)d;                                          // DF.cpp:289-290
        }
        if (in_frame) {
            conf[(size_t)i * g.cpitch + j] = c;
            if (a.out && !in_roi)                                          // DF.cpp:284
                reinterpret_cast<int16_t*>(reinterpret_cast<char*>(a.out) + (ptrdiff_t)pz * a.psO +
                                           (ptrdiff_t)i * a.sO)[j] = a.fill;
        }
        if (!U0) continue;
        if (a.orient != ORIENT_T) {
            if (in_roi) {
                const size_t o = pair ? pair_index(i - g.ry, j - g.rx, g.pw) : (size_t)(i - g.ry) * g.pw + (j - g.rx);
                U0[o] = u0; U1[o] = c;
            }
        } else {
            t0[tx * (TY + 1) + ty + 4 * kk] = u0;
            t1[tx * (TY + 1) + ty + 4 * kk] = c;
        }
    }
    if (U0 && a.orient == ORIENT_T) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TX / 8; m++) {
            const int cidx = tid / TY + 8 * m, ridx = tid % TY;
            const int jj = x0 + cidx, ii = y0 + ridx;
            if (jj >= g.rx && jj < g.rx + g.rw && ii >= g.ry && ii < g.ry + g.rh && ii < g.H && jj < g.W) {
                size_t o = (size_t)(jj - g.rx) * g.ph + (ii - g.ry);
                U0[o] = t0[cidx * (TY + 1) + ridx];
                U1[o] = t1[cidx * (TY + 1) + ridx];
            }
        }
    }
}

__global__ void __launch_bounds__(NT) plain_prologue_kernel(PlainPrologueArgs a)
{
    __shared__ float t0[TX * (TY + 1)];
    __shared__ float t1[TX * (TY + 1)];
    const Geom& g = a.g;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY; // ROI coordinates
    const size_t pz = blockIdx.z;
    const char* pL = reinterpret_cast<const char*>(a.src) + (ptrdiff_t)pz * a.pair_stride;
    const float* cf = a.conf ? a.conf + pz * g.cframe + g.cx0 : nullptr;
    const bool pair = a.orient == ORIENT_PAIR;          // only with confidence weighting (two right-hand sides)
    float* U0 = a.U0 + pz * (pair ? 2 : 1) * g.plane;
    const bool two = a.conf != nullptr || a.pair2;      // two right-hand sides
    float* U1 = two ? (pair ? U0 + ADF_STRIP : a.U1 + pz * g.plane) : nullptr;
    const int j = x0 + tx;
    // loads of all TY/4 pixels first, stores afterwards (see lrc_prologue_kernel)
    constexpr int NK = TY / 4;
    float v0[NK], v1[NK];
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        const int i = y0 + ty + 4 * kk;
        const bool ok = i < g.rh && j < g.rw;
        float u0 = 0.0f, u1 = 0.0f;
        if (ok) {
            const char* row = pL + (ptrdiff_t)(g.ry + i) * a.stride;
            const size_t e = (size_t)(g.rx + j) * a.cn + a.c;
            if (a.depth == 3) u0 = (float)reinterpret_cast<const int16_t*>(row)[e];      // CV_16S
            else if (a.depth == 0) u0 = (float)reinterpret_cast<const uint8_t*>(row)[e]; // CV_8U
            else u0 = reinterpret_cast<const float*>(row)[e];                            // CV_32F
            if (a.pair2) {                                                               // second channel, FGS.cpp:200-205
                const size_t e2 = (size_t)(g.rx + j) * a.cn + a.c2;
                if (a.depth == 3) u1 = (float)reinterpret_cast<const int16_t*>(row)[e2];
                else if (a.depth == 0) u1 = (float)reinterpret_cast<const uint8_t*>(row)[e2];
                else u1 = reinterpret_cast<const float*>(row)[e2];
            }
            if (cf) {                                                                    // DF.cpp:286-290
                u1 = cf[(size_t)(g.ry + i) * g.cpitch + g.rx + j];
                u0 = u1 * u0;
            }
        }
        v0[kk] = u0; v1[kk] = u1;
    }
#pragma unroll
    for (int kk = 0; kk < NK; kk++) asm volatile("" : "+v"(v0[kk]), "+v"(v1[kk]));   // the one wait
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
        const int i = y0 + ty + 4 * kk;
        const bool ok = i < g.rh && j < g.rw;
        if (a.orient != ORIENT_T) {
            if (ok) {
                const size_t o = pair ? pair_index(i, j, g.pw) : (size_t)i * g.pw + j;
                U0[o] = v0[kk]; if (two) U1[o] = v1[kk];
            }
        } else {
            t0[tx * (TY + 1) + ty + 4 * kk] = v0[kk];
            t1[tx * (TY + 1) + ty + 4 * kk] = v1[kk];
        }
    }
    if (a.orient == ORIENT_T) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TX / 8; m++) {
            const int cidx = tid / TY + 8 * m, ridx = tid % TY;
            const int jj = x0 + cidx, ii = y0 + ridx;
            if (jj < g.rw && ii < g.rh) {
                U0[(size_t)jj * g.ph + ii] = t0[cidx * (TY + 1) + ridx];
                if (two) U1[(size_t)jj * g.ph + ii] = t1[cidx * (TY + 1) + ridx];
            }
        }
    }
}

inline size_t disc_lds_bytes(int r)
{
    const size_t IH = TY + 2 * r, IWP = (TX + 2 * r) | 1;
    return IH * TX * 8 + IH * TX * 4 + ((IH * IWP * 2 + 15) & ~(size_t)15);
}

} // namespace

int max_disc_radius() { return MAX_RADIUS; }

hipError_t launch_discontinuity(const DiscArgs& a, int n_pairs, hipStream_t st)
{
    if (a.rw <= 0 || a.rh <= 0 || n_pairs <= 0) return hipSuccess;
    if (a.radius < 0 || a.radius > MAX_RADIUS) return hipErrorInvalidValue;
    if (a.radius <= 8) {
        dim3 grid(1, 1, (a.only_view >= 0 ? 1 : 2) * n_pairs);
#define ADF_DC(RR)                                                                             \
    case RR:                                                                                   \
        grid.x = (a.rw + (NT - 2 * RR) - 1) / (NT - 2 * RR);                                   \
        grid.y = row_blocks(a.rh, grid.x * grid.z);                                            \
        hipLaunchKernelGGL(discontinuity_col_kernel<RR>, grid, dim3(NT), 0, st, a);            \
        break;
        switch (a.radius) {
            ADF_DC(0) ADF_DC(1) ADF_DC(2) ADF_DC(3) ADF_DC(4) ADF_DC(5) ADF_DC(6) ADF_DC(7) ADF_DC(8)
        }
#undef ADF_DC
        return hipGetLastError();
    }
    static_assert(TX == 64 && TY == 32 && NT == 256, "discontinuity_kernel assumes a 64x32 tile and 256 threads");
    const size_t lds = disc_lds_bytes(a.radius);
    if (lds > 48 * 1024) {      // per function AND device: set whenever needed, never cached process-wide
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(discontinuity_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((a.rw + TX - 1) / TX, (a.rh + TY - 1) / TY, (a.only_view >= 0 ? 1 : 2) * n_pairs);
    hipLaunchKernelGGL(discontinuity_kernel, grid, dim3(NT), lds, st, a);
    return hipGetLastError();
}

int conf_left_max_radius() { return 8; }

constexpr size_t CB_LDS_LIMIT = 150 * 1024;
static inline size_t conf_band_lds_rows(int rw, size_t ring_rows) { const size_t rwp = (size_t)((rw + 3) & ~3); return 2 * rwp * 4 + ring_rows * rwp * 2; }
// does the band kernel keep the right view's whole row ring in LDS for this width?
static inline bool conf_band_rlds(int rw, int radius) { return CB_RLDS(radius) && conf_band_lds_rows(rw, 2 * (size_t)radius + 2) <= CB_LDS_LIMIT; }
static inline size_t conf_band_lds(int rw, int radius)
{
    return conf_band_lds_rows(rw, conf_band_rlds(rw, radius) ? 2 * (size_t)radius + 2 : (size_t)radius + 2);   // rows of raw right disparities
}

bool conf_band_fits(const Geom& g, int radius)
{
    return radius >= 1 && radius <= CB_MAX_RADIUS && g.rw >= 8 && g.rw > radius && g.rw <= CB_MAX_WAVES * CB_WOUT(radius) &&
           g.rh > radius && conf_band_lds(g.rw, radius) <= CB_LDS_LIMIT;
}

// Band workgroups of this shape a CU holds at once (occupancy query, remembered per radius for the last shape asked).
static int conf_band_resident(int radius, bool rlds, int threads, size_t lds, int dev)
{
    struct Memo { int dev, threads; size_t lds; bool rlds; int value; };
    static Memo memo[CB_MAX_RADIUS + 1] = {};
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lk(mu);
        const Memo& m = memo[radius];
        if (m.value > 0 && m.dev == dev && m.threads == threads && m.lds == lds && m.rlds == rlds) return m.value;
    }
    int v = 0;
    hipError_t e = hipErrorInvalidValue;
    // (radius 1..3 have no LDS-ring variant, 4..6 always use it, 7..8 by the ROI's width)
#define ADF_CBQ(RR) case RR: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conf_band_kernel<RR>, threads, lds); break;
#define ADF_CBQ2(RR) case RR: e = rlds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conf_band_kernel<RR, true>, threads, lds) \
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, conf_band_kernel<RR, false>, threads, lds); break;
    switch (radius) { ADF_CBQ(1) ADF_CBQ(2) ADF_CBQ(3) ADF_CBQ(4) ADF_CBQ(5) ADF_CBQ(6) ADF_CBQ2(7) ADF_CBQ2(8) }
#undef ADF_CBQ
#undef ADF_CBQ2
    if (e != hipSuccess || v < 1) { (void)hipGetLastError(); v = 1; }
    std::lock_guard<std::mutex> lk(mu);
    memo[radius] = Memo{dev, threads, lds, rlds, v};
    return v;
}

template <int RR, bool RL>
static hipError_t launch_conf_band_variant(const ConfBandArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st)
{
    // (the attribute belongs to the function ON THE CURRENT DEVICE and a process may hold handles on several: no cache)
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conf_band_kernel<RR, RL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conf_band_kernel<RR, RL>), grid, block, lds, st, a);
    return hipSuccess;
}

hipError_t launch_conf_band(const ConfBandArgs& a0, int n_pairs, hipStream_t st)
{
    if (!conf_band_fits(a0.g, a0.radius)) return hipErrorInvalidValue;
    ConfBandArgs a = a0;
    const int waves = (a.g.rw + CB_WOUT(a.radius) - 1) / CB_WOUT(a.radius);
    // bands: at most ONE workgroup per CU, on three quarters of the CUs (round 3; it was two rounds of the chip).  A band workgroup is 15 waves with 57 KB of
    // LDS: it only starts on a CU that has all of that free at once, and the weight kernel that runs beside this one on
    // the side stream refills every slot a finished band leaves with its own small workgroups -- the second round's
    // bands then queue until the weight kernel is done (kernel timeline: this kernel 2.38 ms beside it, 1.32 alone).
    // With no more bands than CUs every band starts at once and runs through: 1.39-1.43 ms beside the weight kernel, the
    // pair 2.19-2.22 ms instead of 2.46.  A band is walked row by row (one workgroup barrier per row, ~1.3 us each): with
    // few pairs per call the band height IS the kernel's latency (one 1242x375 frame in 32-row bands: 12 workgroups,
    // 47 us), so small calls get bands down to max(4, 2 * radius) rows.
    static const int bands_env = [] { const char* e = getenv("ADF_CONF_BANDS_TOTAL"); return e ? atoi(e) : 0; }();   // A/B knob
    int bands_total = bands_env;
    if (bands_total <= 0) {
        static int cu_count[64] = {0};                        // per device, read once (racing writers store the same value)
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        if (dev >= 0 && dev < 64) cus = __atomic_load_n(&cu_count[dev], __ATOMIC_RELAXED);
        if (cus <= 0) {
            if (dev < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            if (dev >= 0 && dev < 64) __atomic_store_n(&cu_count[dev], cus, __ATOMIC_RELAXED);
        }
        // a quarter of the CUs left to the weight kernel (and the fill) alone: radius 2 -- both finish together, 1.82 / 1.92 ms
        // instead of 1.35 / 2.0 with a band on every CU, the step 13.12-13.18 -> 13.04-13.09 ms; radius 3 -- 13.55-13.78 ->
        // 13.09-13.13 ms; radius 5 on the StereoBM factory's ROI -- 13.87-14.16 -> 13.69-13.96 (the band kernels of radius 4..8
        // fill a SIMD's registers with four waves: nothing runs BESIDE them, but the fill no longer crawls behind them)
        // (narrow ROIs make small band workgroups, several of which share a CU: ask the runtime how many -- 256 frames of
        // 1242x375 per call lost 6 % with one tall band per frame)
        int per_cu = conf_band_resident(a.radius, conf_band_rlds(a.g.rw, a.radius), 64 * waves, conf_band_lds(a.g.rw, a.radius), dev);
        if (per_cu < 1) per_cu = 1;
        // (after the slide's instruction diet the kernels whose registers leave room for a wave of the weight kernel beside
        // the band's four on a SIMD -- at most 104 per lane: radius 1..3, and 4..5 since their right-view ring moved to
        // LDS -- do better with a band on EVERY CU: radius 2 12.81-13.24 against 13.05-13.20 ms per 64 x 4K step, four
        // alternating runs each; radius 3 12.78-13.22 against 13.18-13.33; StereoBM factory's geometry (radius 5)
        // 13.04-13.17 against 13.46-13.63, and 13.59-13.64 with the ring in registers; radius 4 13.02-13.17 against 13.54.
        // Radius 6..8 (108..128 registers) keep the quarter free: radius 6 13.65-13.78 against 13.92)
        static const int quarters_env = [] { const char* e = getenv("ADF_CONF_BAND_QUARTERS"); return e ? atoi(e) : 0; }();   // A/B knob
        // (narrow ROIs, whose small band workgroups share a CU, keep the quarter free at every radius: 256 frames of
        // 1242x375 per call 2.95-2.97 against 3.06 ms)
        const int quarters = quarters_env >= 1 && quarters_env <= 4 ? quarters_env : (a.radius <= 5 && per_cu == 1 ? 4 : 3);
        bands_total = cus * quarters / 4 * per_cu;
    }
    int bands = (bands_total + n_pairs - 1) / n_pairs;
    int rpb = (a.g.rh + bands - 1) / bands;
    const int min_rpb = 2 * a.radius > 4 ? 2 * a.radius : 4;     // (at least as many output rows as halo rows)
    if (rpb < min_rpb) rpb = min_rpb;
    if (rpb > CB_MAX_BAND_ROWS) rpb = CB_MAX_BAND_ROWS;
    if (rpb > a.g.rh) rpb = a.g.rh;
    a.rows_per_band = rpb;
    const dim3 grid((a.g.rh + rpb - 1) / rpb, n_pairs), block(64 * waves);
    size_t lds = conf_band_lds(a.g.rw, a.radius);
    if (a.lds_floor > lds) lds = a.lds_floor;         // (occupancy control: see ConfBandArgs::lds_floor)
    const bool rl = conf_band_rlds(a.g.rw, a.radius);
    hipError_t le = hipErrorInvalidValue;
#define ADF_CB(RR, RL) le = launch_conf_band_variant<RR, RL>(a, grid, block, lds, st)
    switch (a.radius) {
    case 1: ADF_CB(1, false); break;
    case 2: ADF_CB(2, false); break;
    case 3: ADF_CB(3, false); break;
    case 4: ADF_CB(4, true); break;
    case 5: ADF_CB(5, true); break;
    case 6: ADF_CB(6, true); break;
    case 7: if (rl) ADF_CB(7, true); else ADF_CB(7, false); break;
    case 8: if (rl) ADF_CB(8, true); else ADF_CB(8, false); break;
    }
    if (le != hipSuccess) return le;
#undef ADF_CB
    return hipGetLastError();
}

// The merged preparation launch: for calls small enough that three kernels' latencies, not their work, are the time.
bool prep_small_fits(const Geom& g, int radius, int channels, int n_pairs)
{
    return conf_band_fits(g, radius) && radius <= 5 && (channels == 1 || channels == 3) &&
           (double)g.rw * g.rh * n_pairs <= 2.5e6;
}

// (the weight role reads the guide through a 32-bit buffer descriptor per block: prep_bodies.h)
bool prep_small_guide_fits(const Geom& g, ptrdiff_t guide_stride, int channels)
{
    return guide_stride > 0 && (size_t)(g.rh + 1) * (size_t)guide_stride < ((size_t)1 << 30) && (size_t)g.W * channels < ((size_t)1 << 29);
}

hipError_t launch_prep_small(const ConfBandArgs& c0, const WeightArgs& w, const OutsideArgs& o, int n_pairs, hipStream_t st)
{
    if (!prep_small_fits(c0.g, c0.radius, w.ch, n_pairs)) return hipErrorInvalidValue;
    if (!(w.chor_orient == ORIENT_N && w.cvert_orient == ORIENT_STRIP)) return hipErrorInvalidValue;
    if (!prep_small_guide_fits(c0.g, w.stride, w.ch)) return hipErrorInvalidValue;
    PrepArgs a{};
    a.c = c0; a.w = w; a.o = o;
    const Geom& g = c0.g;
    int waves = (g.rw + CB_WOUT(c0.radius) - 1) / CB_WOUT(c0.radius);
    if (waves < prep::WS_NT / 64) waves = prep::WS_NT / 64;
    if (waves < NT / 64) waves = NT / 64;                  // (the fill role works in blocks of NT threads)
    // short bands and few rows per weight block: every role is walked row by row, a barrier per row
    // (as short as the halo allows while that still leaves no more than about one band per CU, and about two weight
    // blocks per CU: beyond that the roles only queue behind each other)
    int rpb = 2 * c0.radius > 4 ? 2 * c0.radius : 4;
    const int rows_all = g.rh * n_pairs;
    if (rpb < (rows_all + 255) / 256) rpb = (rows_all + 255) / 256;
    if (rpb > CB_MAX_BAND_ROWS) rpb = CB_MAX_BAND_ROWS;
    if (rpb > g.rh) rpb = g.rh;
    a.c.rows_per_band = rpb;
    a.nC = (g.rh + rpb - 1) / rpb;
    a.nWx = (g.rw + prep::WS_BCOLS - 1) / prep::WS_BCOLS;
    int wrows = (rows_all * a.nWx + 511) / 512;
    if (wrows < 4) wrows = 4;
    a.nWy = (g.rh + wrows - 1) / wrows;
    a.nW = a.nWx * a.nWy;
    const int side = g.W - g.rw;
    a.nOAx = (side + NT - 1) / NT;
    a.nOA = side > 0 ? a.nOAx * ((g.rh + OUT_ROWS - 1) / OUT_ROWS) : 0;
    if (a.nOAx < 1) a.nOAx = 1;
    const size_t npix = (size_t)(g.H - g.rh) * g.W;
    const int nOB = (int)((npix + (size_t)NT * OUT_ROWS - 1) / ((size_t)NT * OUT_ROWS));
    const dim3 grid(a.nC + a.nW + a.nOA + nOB, n_pairs), block(64 * waves);
    const size_t lds = conf_band_lds(g.rw, c0.radius);
    if (lds + sizeof(prep::WsShared<3>) > 150 * 1024) return hipErrorInvalidValue;
#define ADF_PS(CC, RR)                                                                                     \
    case RR:                                                                                               \
        if (lds > 32 * 1024) {                                                                             \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prep_small_kernel<CC, RR>),   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
            if (e != hipSuccess) return e;                                                                 \
        }                                                                                                  \
        hipLaunchKernelGGL((prep_small_kernel<CC, RR>), grid, block, lds, st, a);                          \
        break;
    if (w.ch == 1) { switch (c0.radius) { ADF_PS(1, 1) ADF_PS(1, 2) ADF_PS(1, 3) ADF_PS(1, 4) ADF_PS(1, 5) } }
    else { switch (c0.radius) { ADF_PS(3, 1) ADF_PS(3, 2) ADF_PS(3, 3) ADF_PS(3, 4) ADF_PS(3, 5) } }
#undef ADF_PS
    return hipGetLastError();
}

hipError_t launch_conf_left(const ConfLeftArgs& a, int n_pairs, hipStream_t st)
{
    if (a.radius < 0 || a.radius > 8) return hipErrorInvalidValue;
    const bool wu = a.U0 != nullptr;
    dim3 grid(1, 1, n_pairs);
#define ADF_CL(RR)                                                                             \
    case RR:                                                                                   \
        grid.x = (a.g.rw + (NT - 2 * RR) - 1) / (NT - 2 * RR);                                 \
        grid.y = row_blocks(a.g.rh, grid.x * grid.z);                                          \
        if (wu) hipLaunchKernelGGL((conf_left_kernel<RR, true>), grid, dim3(NT), 0, st, a);    \
        else hipLaunchKernelGGL((conf_left_kernel<RR, false>), grid, dim3(NT), 0, st, a);      \
        break;
    switch (a.radius) { ADF_CL(0) ADF_CL(1) ADF_CL(2) ADF_CL(3) ADF_CL(4) ADF_CL(5) ADF_CL(6) ADF_CL(7) ADF_CL(8) }
#undef ADF_CL
    return hipGetLastError();
}

hipError_t launch_outside(const OutsideArgs& a, int n_pairs, hipStream_t st)
{
    const int side = a.g.W - a.g.rw;
    const int nAx = (side + NT - 1) / NT, nA = side > 0 ? nAx * ((a.g.rh + OUT_ROWS - 1) / OUT_ROWS) : 0;
    const size_t npix = (size_t)(a.g.H - a.g.rh) * a.g.W;
    const size_t nB = (npix + (size_t)NT * OUT_ROWS - 1) / ((size_t)NT * OUT_ROWS);
    if (nA + nB == 0) return hipSuccess;
    if (nA + nB > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(outside_kernel, dim3((unsigned)(nA + nB), 1, n_pairs), dim3(NT), 0, st, a, nAx > 0 ? nAx : 1, nA);
    return hipGetLastError();
}

hipError_t launch_lrc_prologue(const LrcArgs& a, int n_pairs, hipStream_t st)
{
    dim3 grid((a.g.W + TX - 1) / TX, (a.g.H + TY - 1) / TY, n_pairs);
    hipLaunchKernelGGL(lrc_prologue_kernel, grid, dim3(NT), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_plain_prologue(const PlainPrologueArgs& a, int n_pairs, hipStream_t st)
{
    dim3 grid((a.g.rw + TX - 1) / TX, (a.g.rh + TY - 1) / TY, n_pairs);
    hipLaunchKernelGGL(plain_prologue_kernel, grid, dim3(NT), 0, st, a);
    return hipGetLastError();
}

} // namespace adf
