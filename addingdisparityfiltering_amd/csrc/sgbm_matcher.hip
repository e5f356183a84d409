// sgbm_matcher.hip -- semi-global matcher on the device (second producer).
//
// The reference's sample feeds the filter from cv::StereoSGBM with MODE_SGBM_3WAY
// (samples/disparity_filtering.cpp:166-176); the filter factories only touch its parameters
// (disparity_filters.cpp:404-409, 432-445).  cv::StereoSGBM lives in OpenCV's calib3d, outside the reference tree:
// PARITY UNPINNED there.  What is built is the published algorithm (Hirschmueller 2008, formula 13, three paths,
// Birchfield-Tomasi block cost) exactly as oracle/adf_oracle_sgbm.c states it -- integer work, bit-identical to that
// oracle -- with the in-tree derivative of OpenCV's aggregation loop (modules/stereo/src/stereo_binary_sgbm.cpp) as
// the line-cited anchor of the recurrence, the winner / tie rule, the uniqueness test and the sub-pixel fit.
//
// Five kernels per call, all integer, none with a dense contraction (no MFMA):
//   sgbm_signals_kernel   per pixel and channel: clipped x-derivative + intensity, each with the extrema over its
//                         half-sample neighbours, packed as 16-bit pairs (derivative | intensity) so that one packed
//                         instruction serves both Birchfield-Tomasi terms
//   sgbm_cost_kernel      block cost volume C[y][x][d] (int16): lanes along x, 16 (8) disparities per wave, walking down
//                         a band of rows; one pixel cost per lane, row and disparity, the horizontal window from the
//                         neighbouring lanes by DPP, the vertical one as a running sum over a register ring
//                         (stereo_binary_sgbm.cpp:205-276 is the same running-sum structure)
//   sgbm_path_kernel      formula 13 along one direction, one wavefront per scanline, the D path costs of a pixel held
//                         4 (2, 1, 8) per lane; d-1 / d+1 across lanes by whole-wave DPP shifts, min_k by a DPP
//                         butterfly + 4 readlanes.  TOP writes the volume S, LEFT adds to it, RIGHT adds, picks the
//                         winner and fits the sub-pixel parabola (stereo_binary_sgbm.cpp:286-301, 419-446, 519-596)
//   sgbm_fill_kernel / sgbm_median_kernel   invalid value everywhere first; 3x3 median of the CV_16S map last
// HBM: C, S and L2 volumes of H x width1 x D int16 each (4K, 256 disparities: 4 GB each, per image in flight).
#include "adf_internal.h"
#include "../../include/adf_wls.h"

#include <algorithm>
#include <cstdlib>
#include <new>

namespace {

constexpr int SG_MAX_COST = 32767;   // SHRT_MAX: guard value of the d = -1 / d = D neighbours (stereo_binary_sgbm.cpp:323-324)
constexpr int SG_DISP_SHIFT = 4, SG_DISP_SCALE = 16;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int sat16i(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// ---------------------------------------------------------------------------------------------------------------
// signals: rec[(y*W + x)*cn + c] = {V, V0, V1}, each (derivative | intensity << 16) of channel c
// ---------------------------------------------------------------------------------------------------------------
struct SignalArgs {
    const uint8_t* img; ptrdiff_t stride, pair_stride; int cn, W, H, ftzero;
    uint32_t* rec; size_t rec_pair;     // dwords per image
};

__device__ __forceinline__ uint32_t sg_signal(const uint8_t* row, const uint8_t* up, const uint8_t* dn, int x, int c, int cn, int W, int ftzero)
{
    if (x <= 0 || x >= W - 1) return (uint32_t)ftzero | ((uint32_t)ftzero << 16);     // border columns hold ftzero
    const int a = x * cn + c;
    int g = (row[a + cn] - row[a - cn]) * 2 + up[a + cn] - up[a - cn] + dn[a + cn] - dn[a - cn];
    g = min(max(g, -ftzero), ftzero) + ftzero;
    return (uint32_t)g | ((uint32_t)row[a] << 16);
}

__global__ void __launch_bounds__(256) sgbm_signals_kernel(SignalArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.W) return;
    const uint8_t* row = a.img + (ptrdiff_t)blockIdx.z * a.pair_stride + (ptrdiff_t)y * a.stride;
    const uint8_t* up = y > 0 ? row - a.stride : row;
    const uint8_t* dn = y < a.H - 1 ? row + a.stride : row;
    uint32_t* o = a.rec + (size_t)blockIdx.z * a.rec_pair + ((size_t)y * a.W + x) * a.cn * 3;
    for (int c = 0; c < a.cn; c++) {
        const uint32_t v = sg_signal(row, up, dn, x, c, a.cn, a.W, a.ftzero);
        const uint32_t l = sg_signal(row, up, dn, x - 1, c, a.cn, a.W, a.ftzero);
        const uint32_t r = sg_signal(row, up, dn, x + 1, c, a.cn, a.W, a.ftzero);
        uint32_t out0 = 0, out1 = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {                          // the two signals of the pair
            const int vv = (v >> (16 * h)) & 0xffff, ll = (l >> (16 * h)) & 0xffff, rr = (r >> (16 * h)) & 0xffff;
            const int vl = x > 0 ? (vv + ll) / 2 : vv;
            const int vr = x < a.W - 1 ? (vv + rr) / 2 : vv;
            out0 |= (uint32_t)min(min(vl, vr), vv) << (16 * h);
            out1 |= (uint32_t)max(max(vl, vr), vv) << (16 * h);
        }
        o[c * 3 + 0] = v; o[c * 3 + 1] = out0; o[c * 3 + 2] = out1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// block costs
// ---------------------------------------------------------------------------------------------------------------
struct CostArgs {
    const uint32_t* rec1; const uint32_t* rec2; size_t rec_pair;
    int16_t* C; size_t vol;              // int16 elements per image volume = H * w1 * D
    int W, H, D, minD, minX1, w1, rows_per_band;
};

struct Rec { uint32_t v, v0, v1; };

// Birchfield-Tomasi cost of the pair of signals: distance of u to [v0, v1] and of v to [u0, u1], the smaller one, as
// saturating 16-bit pairs (v0 <= v1, so at most one of the two differences of a distance is non-zero); the intensity
// half is scaled by 1/4 before the terms are summed.
__device__ __forceinline__ us2 bt_pair(const Rec& u, const Rec& v)
{
    const us2 U = __builtin_bit_cast(us2, u.v), U0 = __builtin_bit_cast(us2, u.v0), U1 = __builtin_bit_cast(us2, u.v1);
    const us2 V = __builtin_bit_cast(us2, v.v), V0 = __builtin_bit_cast(us2, v.v0), V1 = __builtin_bit_cast(us2, v.v1);
    const us2 c0 = __builtin_elementwise_sub_sat(V0, U) | __builtin_elementwise_sub_sat(U, V1);
    const us2 c1 = __builtin_elementwise_sub_sat(U0, V) | __builtin_elementwise_sub_sat(V, U1);
    const us2 m = __builtin_elementwise_min(c0, c1);
    const us2 sh = {0, 2};
    return m >> sh;
}

// Lanes run along x: lane = one matchable column (the first and last BS/2 lanes of a wave are halo and repeat the edge
// column at the borders of the matchable area = the clamped window), wave = DD consecutive disparities of 64 - 2*(BS/2)
// output columns, workgroup = 4 waves = 4*DD disparities of the same columns (their int16 results are one contiguous
// piece per column).  Per row and disparity a lane evaluates ONE pixel cost (its own column of image 1 against column
// x - d of image 2: consecutive lanes read consecutive records) and receives its neighbours' through whole-wave DPP
// shifts; the vertical window is a running sum over a register ring with static slots (the row loop is unrolled by BS).
template <int BS, int CN, int DD>
__global__ void __launch_bounds__(256) sgbm_cost_kernel(CostArgs a)
{
    constexpr int R = BS / 2, OUTW = 64 - 2 * R;
    constexpr int TD = 4 * DD;                                // disparities of the workgroup: one contiguous piece per column
    constexpr int TP = TD + 8;                                // LDS row pitch in int16 (16-byte aligned, staggers the banks)
    __shared__ __align__(16) int16_t tile[2][64 * TP];        // the workgroup's results of one row, [column][disparity]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dbase = blockIdx.y * TD, d0 = dbase + wv * DD;
    const bool wave_active = d0 < a.D;                        // (D is a multiple of 16, DD is 8 or 16: all or nothing per wave)
    const int nbands = (a.H + a.rows_per_band - 1) / a.rows_per_band;
    const int img = blockIdx.z / nbands, band = blockIdx.z % nbands;
    const int y0 = band * a.rows_per_band;
    const int rows_out = min(a.rows_per_band, a.H - y0);
    const int xt0 = blockIdx.x * OUTW;
    const int x1 = xt0 + lane - R;
    const int X = a.minX1 + min(max(x1, 0), a.w1 - 1);         // image column (window clamped inside the matchable area)
    const uint32_t* rec1 = a.rec1 + (size_t)img * a.rec_pair;
    const uint32_t* rec2 = a.rec2 + (size_t)img * a.rec_pair;
    int16_t* C = a.C + (size_t)img * a.vol;
    const int dl = wave_active ? d0 : dbase;                  // idle waves walk along (they share the barriers and the stores)
    // Round 4: TWO disparities share a register from the pixel cost on (low / high half) wherever the block sum fits 16
    // bits -- a pixel cost is at most 189 per channel (derivative distance <= 2 * 63, intensity distance <= 255 / 4) --
    // so the window shifts, the running sums and the clamp run once per pair: about a third fewer vector instructions
    // in a kernel that is bound by issuing them.  (Sums of halves never carry: every intermediate stays below 2^16, and
    // the ring entry is subtracted from the running sum BEFORE the new one is added.)
    constexpr bool PACK = BS * BS * CN * 189 < 65536;
    int ring[PACK ? 1 : BS][PACK ? 1 : DD], csum[PACK ? 1 : DD];
    uint32_t ring2[PACK ? BS : 1][PACK ? DD / 2 : 1], csum2[PACK ? DD / 2 : 1];
    if constexpr (PACK) {
#pragma unroll
        for (int j = 0; j < DD / 2; j++) {
            csum2[j] = 0;
#pragma unroll
            for (int k = 0; k < BS; k++) ring2[k][j] = 0;
        }
    } else {
#pragma unroll
        for (int dd = 0; dd < DD; dd++) {
            csum[dd] = 0;
#pragma unroll
            for (int k = 0; k < BS; k++) ring[k][dd] = 0;
        }
    }
    // write-out: 16 bytes per thread, TD/8 threads per column => every store instruction covers whole contiguous pieces
    constexpr int TPC = TD / 8;                               // threads per column
    const int nd_blk = min(TD, a.D - dbase);                  // disparities this workgroup really has
    const int nsteps = rows_out + BS - 1;
    for (int n0 = 0; n0 < nsteps; n0 += BS) {
#pragma unroll
        for (int s = 0; s < BS; s++) {
            const int n = n0 + s;
            if (n < nsteps) {                                    // block-uniform
                const int yy = min(max(y0 - R + n, 0), a.H - 1);
                const uint32_t* pu = rec1 + ((size_t)yy * a.W + X) * (CN * 3);
                const uint32_t* pv = rec2 + ((size_t)yy * a.W + (X - (dl + a.minD))) * (CN * 3);
                Rec u[CN];
#pragma unroll
                for (int c = 0; c < CN; c++) u[c] = Rec{pu[c * 3], pu[c * 3 + 1], pu[c * 3 + 2]};
                short res[PACK ? 1 : DD];
                uint32_t res2[PACK ? DD / 2 : 1];
                auto pixel_cost = [&](int dd) -> uint32_t {
                    us2 acc = {0, 0};
#pragma unroll
                    for (int c = 0; c < CN; c++) {
                        const uint32_t* q = pv - dd * (CN * 3) + c * 3;           // column X - (d0 + dd + minD)
                        const Rec v = {q[0], q[1], q[2]};
                        acc += bt_pair(u[c], v);
                    }
                    return (uint32_t)acc.x + (uint32_t)acc.y;
                };
                if constexpr (PACK) {
#pragma unroll
                    for (int j = 0; j < DD / 2; j++) {
                        const uint32_t pp = pixel_cost(2 * j) | (pixel_cost(2 * j + 1) << 16);
                        uint32_t hs = pp, tl = pp, tr = pp;
#pragma unroll
                        for (int k = 0; k < R; k++) {               // neighbours' pixel costs, one more column per shift
                            tl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tl, 0x138, 0xf, 0xf, true);   // wave_shr:1, 0 past the wave's end
                            tr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)tr, 0x130, 0xf, 0xf, true);   // wave_shl:1
                            hs += tl + tr;
                        }
                        csum2[j] = (csum2[j] - ring2[s][j]) + hs;
                        ring2[s][j] = hs;
                        const us2 cl = __builtin_elementwise_min(__builtin_bit_cast(us2, csum2[j]), (us2){(unsigned short)SG_MAX_COST, (unsigned short)SG_MAX_COST});
                        res2[j] = __builtin_bit_cast(uint32_t, cl);
                    }
                } else {
#pragma unroll
                for (int dd = 0; dd < DD; dd++) {
                    const int pix = (int)pixel_cost(dd);
                    int hs = pix, tl = pix, tr = pix;
#pragma unroll
                    for (int k = 0; k < R; k++) {                   // neighbours' pixel costs, one more column per shift
                        tl = __builtin_amdgcn_update_dpp(0, tl, 0x138, 0xf, 0xf, false);   // wave_shr:1
                        tr = __builtin_amdgcn_update_dpp(0, tr, 0x130, 0xf, 0xf, false);   // wave_shl:1
                        hs += tl + tr;
                    }
                    csum[dd] += hs - ring[s][dd];
                    ring[s][dd] = hs;
                    res[dd] = (short)min(csum[dd], SG_MAX_COST);
                }
                }
                if (n >= BS - 1) {                               // block-uniform: a row of results goes out through LDS
                    typedef short v8s __attribute__((ext_vector_type(8)));
                    int16_t* tl_ = tile[n & 1];
                    if (wave_active) {
#pragma unroll
                        for (int h = 0; h < DD / 8; h++) {
                            if constexpr (PACK) {
                                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                                *reinterpret_cast<v4u*>(tl_ + lane * TP + wv * DD + h * 8) = v4u{res2[4 * h], res2[4 * h + 1], res2[4 * h + 2], res2[4 * h + 3]};
                            } else {
                                v8s q;
#pragma unroll
                                for (int k = 0; k < 8; k++) q[k] = res[h * 8 + k];
                                *reinterpret_cast<v8s*>(tl_ + lane * TP + wv * DD + h * 8) = q;
                            }
                        }
                    }
                    __syncthreads();
                    int16_t* dst = C + (size_t)(y0 + n - (BS - 1)) * a.w1 * a.D + dbase;
                    const int part = threadIdx.x % TPC;
#pragma unroll
                    for (int col = threadIdx.x / TPC; col < OUTW; col += 256 / TPC) {
                        const int xo = xt0 + col;
                        if (xo < a.w1 && part * 8 < nd_blk)
                            *reinterpret_cast<v8s*>(dst + (size_t)xo * a.D + part * 8) = *reinterpret_cast<const v8s*>(tl_ + (col + R) * TP + part * 8);
                    }
                    // (the other tile buffer is written next: one barrier per row is enough)
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// path costs
// ---------------------------------------------------------------------------------------------------------------
// DIR_TOP: first path, writes S; DIR_ADD: any direction of travel (dx, dy) in {-1,0,1}^2, adds to S; DIR_RIGHT: the last
// path (from the right), adds, picks the winner
// DIR_ADD_OUT (round 4): the LAST accumulating path writes its costs to a volume of its own (L2) instead of adding them into
// S -- it no longer reads S (8 bytes per pixel and disparity moved instead of 12) -- and DIR_RIGHT, which is bound by
// vector issue and has bandwidth to spare, reads both and adds them in the accumulation's own order:
// sat(sat(S + L_last) + L_right).
enum { DIR_TOP = 0, DIR_ADD = 1, DIR_RIGHT = 2, DIR_ADD_OUT = 3 };

struct PathArgs {
    const int16_t* C; int16_t* S; int16_t* L2; size_t vol;
    int W, H, D, minD, minX1, w1, P1, P2, ur;
    int16_t* out; ptrdiff_t out_stride, out_pair;   // raw disparity map (elements)
    int dx, dy;                                     // DIR_ADD: direction of travel
    int disp12;                                     // DIR_RIGHT: the matcher's own left-right check (>= 100000: off)
};

__device__ __forceinline__ int wave_min_i32(int v)
{
    // min is idempotent: rotations by 1, 2, 4, 8 inside each row of 16 lanes leave the row minimum in every lane
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), e = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, e));
}

// num / den truncated toward zero (C semantics) for |num|, den < 2^24, den > 0: float quotient + one exact correction
// step instead of the ~40-instruction integer division sequence.
__device__ __forceinline__ int div_trunc_small(int num, int den)
{
    int q = (int)((float)num * __builtin_amdgcn_rcpf((float)den));       // truncation toward zero, off by at most one
    int r = num - q * den;
    if (num >= 0) { if (r < 0) { q--; } else if (r >= den) { q++; } }
    else          { if (r > 0) { q++; } else if (r <= -den) { q--; } }
    return q;
}

template <int DPL>
__device__ __forceinline__ void load_costs(const int16_t* p, int (&o)[DPL])
{
    // unconditional (lanes without disparities read lane 0's and ignore them): no load under a branch
    typedef short vs __attribute__((ext_vector_type(DPL)));
    if constexpr (DPL == 1) o[0] = p[0];
    else {
        const vs q = *reinterpret_cast<const vs*>(p);
#pragma unroll
        for (int k = 0; k < DPL; k++) o[k] = q[k];
    }
}

template <int DPL>
__device__ __forceinline__ void store_costs(int16_t* p, bool active, const int (&v)[DPL])
{
    typedef short vs __attribute__((ext_vector_type(DPL)));
    if (!active) return;
    if constexpr (DPL == 1) p[0] = (int16_t)v[0];
    else {
        vs q;
#pragma unroll
        for (int k = 0; k < DPL; k++) q[k] = (short)v[k];
        *reinterpret_cast<vs*>(p) = q;
    }
}

template <int DPL, int DIR>
__global__ void __launch_bounds__(256) sgbm_path_kernel(PathArgs a)
{
    __shared__ int16_t sS[4][64 * DPL];                       // DIR_RIGHT: the pixel's S(d) for the sub-pixel fit
    extern __shared__ int16_t sOut[];                         // DIR_RIGHT: [4][w1] the scanline's results (written out coalesced),
                                                              // then [4][W] disp2 and [4][W] its cost (left-right check)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int line = blockIdx.x * 4 + wv;                     // scanline: a column (TOP), a row (RIGHT), any straight line (ADD)
    // scanline -> first pixel (x0, y0), direction (dx, dy), length.  Diagonals: one line per pixel of the row the
    // path enters through, then one per remaining pixel of the column it enters through.
    int dx, dy, x0, y0, nlines, nsteps;
    if (DIR == DIR_TOP) { dx = 0; dy = 1; nlines = a.w1; x0 = line; y0 = 0; nsteps = a.H; }
    else if (DIR == DIR_RIGHT) { dx = -1; dy = 0; nlines = a.H; x0 = a.w1 - 1; y0 = line; nsteps = a.w1; }
    else {
        dx = a.dx; dy = a.dy;
        const int ys = dy > 0 ? 0 : a.H - 1, xs = dx > 0 ? 0 : a.w1 - 1;
        if (dy == 0) { nlines = a.H; x0 = xs; y0 = line; nsteps = a.w1; }
        else if (dx == 0) { nlines = a.w1; x0 = line; y0 = ys; nsteps = a.H; }
        else {
            nlines = a.w1 + a.H - 1;
            if (line < a.w1) { x0 = line; y0 = ys; }
            else { const int j = line - a.w1 + 1; x0 = xs; y0 = dy > 0 ? j : a.H - 1 - j; }
            const int nx = dx > 0 ? a.w1 - x0 : x0 + 1, ny = dy > 0 ? a.H - y0 : y0 + 1;
            nsteps = min(nx, ny);
        }
    }
    if (line >= nlines) return;
    const int d0 = lane * DPL;
    const bool active = d0 < a.D;
    const size_t vol0 = (size_t)blockIdx.y * a.vol;
    // a lane's DPL disparities are all inside [0, D) or all outside (D is a multiple of 16, DPL divides 16)
    const int16_t* C = a.C + vol0 + (active ? d0 : 0);
    int16_t* S = a.S + vol0 + (active ? d0 : 0);
    int16_t* L2 = a.L2 + vol0 + (active ? d0 : 0);
    constexpr bool READS_S = DIR == DIR_ADD || DIR == DIR_RIGHT;
    // element offset of step t
    auto offs = [&](int t) -> size_t {
        return ((size_t)(y0 + t * dy) * a.w1 + (size_t)(x0 + t * dx)) * a.D;
    };
    int L[DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) L[k] = active ? 0 : SG_MAX_COST;      // zero start (stereo_binary_sgbm.cpp:191-194)
    int minprev = 0;
    const int16_t invalid = (int16_t)((a.minD - 1) * SG_DISP_SCALE);
    int16_t* myOut = sOut + wv * a.w1;
    // the matcher's own left-right check (stereo_binary_sgbm.cpp:548-556, 598-613): disp2 = per column of image 2 the
    // disparity of the cheapest winner pointing at it, kept in LDS for the row
    const bool lrc = DIR == DIR_RIGHT && a.disp12 < 100000;
    int16_t* d2p = sOut + 4 * a.w1 + wv * a.W;
    int16_t* d2c = sOut + 4 * a.w1 + 4 * a.W + wv * a.W;
    if (lrc)
        for (int x = lane; x < a.W; x += 64) { d2p[x] = invalid; d2c[x] = (int16_t)SG_MAX_COST; }   // :449-453

    // one step of formula 13 at scanline position t with the operands c (block cost) and s (S so far)
    auto step = [&](int t, const int (&c)[DPL], const int (&s_in)[DPL], const int (&l2)[DPL]) {
        int s[DPL];
#pragma unroll
        for (int k = 0; k < DPL; k++) s[k] = DIR == DIR_RIGHT ? sat16i(s_in[k] + l2[k]) : s_in[k];   // the last accumulating path's costs
        const size_t o = offs(t);
        // stereo_binary_sgbm.cpp:419-446: neighbours d-1 / d+1, guards SHRT_MAX outside [0, D)
        const int lm = __builtin_amdgcn_update_dpp(SG_MAX_COST, L[DPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
        const int lp = __builtin_amdgcn_update_dpp(SG_MAX_COST, L[0], 0x130, 0xf, 0xf, false);         // wave_shl:1
        const int delta = minprev + a.P2;
        int Ln[DPL], lmin = SG_MAX_COST;
#pragma unroll
        for (int k = 0; k < DPL; k++) {
            const int below = k == 0 ? lm : L[k - 1], above = k == DPL - 1 ? lp : L[k + 1];
            const int m = min(min(L[k], below + a.P1), min(above + a.P1, delta));
            Ln[k] = active ? sat16i(c[k] + m - delta) : SG_MAX_COST;      // :423: minus delta = min_k + P2, L in [C-P2, C]
            lmin = min(lmin, Ln[k]);
        }
#pragma unroll
        for (int k = 0; k < DPL; k++) L[k] = Ln[k];
        minprev = wave_min_i32(lmin);
        if (DIR == DIR_TOP) {
            store_costs<DPL>(S + o, active, Ln);
        } else if (DIR == DIR_ADD_OUT) {
            store_costs<DPL>(L2 + o, active, Ln);
        } else {
            int st[DPL];
#pragma unroll
            for (int k = 0; k < DPL; k++) st[k] = sat16i(s[k] + Ln[k]);
            if (DIR == DIR_ADD) {
                store_costs<DPL>(S + o, active, st);
            } else {
                // winner: the FIRST disparity with the smallest S (stereo_binary_sgbm.cpp:519-528) -- one wave minimum over
                // the key S*1024 + d: |S| < 2^15 (S is negative as a rule: every path cost lies in [C-P2, C]), d < 1024
                int key = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < DPL; k++) if (active) key = min(key, st[k] * 1024 + (d0 + k));
                key = wave_min_i32(key);
                const int minS = key >> 10, best = key & 1023;
                bool reject = false;
                if (a.ur > 0) {                                // stereo_binary_sgbm.cpp:543-547
#pragma unroll
                    for (int k = 0; k < DPL; k++)
                        reject |= active && st[k] * (100 - a.ur) < minS * 100 && abs(best - (d0 + k)) > 1;
                }
                const bool any_reject = __builtin_amdgcn_ballot_w64(reject) != 0;
                store_costs<DPL>(&sS[wv][d0], active, st);
                // (results go to LDS only: no vector-memory operation under a branch inside the pipelined loop)
                int res = invalid;
                if (minS < SG_MAX_COST && !any_reject) {
                    int d = best < a.D ? best : 0;
                    if (lrc && lane == 0) {                   // :549-554 (x runs from the right: sequential in this wave)
                        const int x2 = (a.w1 - 1 - t) + a.minX1 - d - a.minD;
                        if (d2c[x2] > minS) { d2c[x2] = (int16_t)minS; d2p[x2] = (int16_t)(d + a.minD); }
                    }
                    const int dm = d > 0 ? d - 1 : 0, dp = d < a.D - 1 ? d + 1 : d;
                    const int sm = sS[wv][dm], s0 = sS[wv][d], sp = sS[wv][dp];
                    if (0 < d && d < a.D - 1) {                // stereo_binary_sgbm.cpp:584-591
                        const int denom2 = max(sm + sp - 2 * s0, 1);
                        d = d * SG_DISP_SCALE + div_trunc_small((sm - sp) * SG_DISP_SCALE + denom2, denom2 * 2);
                    } else
                        d *= SG_DISP_SCALE;
                    res = d + a.minD * SG_DISP_SCALE;           // :596
                }
                if (lane == 0) myOut[a.w1 - 1 - t] = (int16_t)res;
            }
        }
    };

    // Operands of the next PF steps are in flight while a step is reduced (a step is far shorter than a memory
    // latency).  The pipelined loop has NO branch around a vector-memory operation: with one, the compiler can only
    // wait for every outstanding load at each use, which serialises the ring.
    constexpr int PF = DPL >= 8 ? 4 : 8;
    int cq[PF][DPL], sq[PF][DPL], lq[DIR == DIR_RIGHT ? PF : 1][DPL];
#pragma unroll
    for (int k = 0; k < DPL; k++) lq[0][k] = 0;
#pragma unroll
    for (int p = 0; p < PF; p++) {
        const int tt = p < nsteps ? p : nsteps - 1;
        load_costs<DPL>(C + offs(tt), cq[p]);
        if (READS_S) load_costs<DPL>(S + offs(tt), sq[p]);
        else {
#pragma unroll
            for (int k = 0; k < DPL; k++) sq[p][k] = 0;
        }
        if (DIR == DIR_RIGHT) load_costs<DPL>(L2 + offs(tt), lq[p]);
    }
    int t0 = 0;
    for (; t0 + PF <= nsteps; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int t = t0 + p;
            int c[DPL], s[DPL], l2[DPL];
#pragma unroll
            for (int k = 0; k < DPL; k++) { c[k] = cq[p][k]; s[k] = sq[p][k]; l2[k] = lq[DIR == DIR_RIGHT ? p : 0][k]; }
            const int tt = t + PF < nsteps ? t + PF : nsteps - 1;
            load_costs<DPL>(C + offs(tt), cq[p]);
            if (READS_S) load_costs<DPL>(S + offs(tt), sq[p]);
            if (DIR == DIR_RIGHT) load_costs<DPL>(L2 + offs(tt), lq[p]);
            step(t, c, s, l2);
        }
    }
#pragma unroll
    for (int p = 0; p < PF; p++)                              // the last nsteps % PF steps: operands already in the ring
        if (t0 + p < nsteps) step(t0 + p, cq[p], sq[p], lq[DIR == DIR_RIGHT ? p : 0]);

    if (DIR == DIR_RIGHT) {                                   // the scanline's results, coalesced
        int16_t* out = a.out + (ptrdiff_t)blockIdx.y * a.out_pair + (ptrdiff_t)line * a.out_stride + a.minX1;
        const int maxdiff = a.disp12 > 0 ? a.disp12 : 1;      // :141
        for (int x1 = lane; x1 < a.w1; x1 += 64) {
            int d1 = myOut[x1];
            if (lrc && d1 != invalid) {                       // :598-613: both roundings must disagree to invalidate
                const int x = x1 + a.minX1;
                const int dlo = d1 >> SG_DISP_SHIFT, dhi = (d1 + SG_DISP_SCALE - 1) >> SG_DISP_SHIFT;
                const int xlo = x - dlo, xhi = x - dhi;
                if (0 <= xlo && xlo < a.W && d2p[xlo] >= a.minD && abs(d2p[xlo] - dlo) > maxdiff &&
                    0 <= xhi && xhi < a.W && d2p[xhi] >= a.minD && abs(d2p[xhi] - dhi) > maxdiff)
                    d1 = invalid;
            }
            out[x1] = (int16_t)d1;
        }
    }
}

struct FillArgs { int16_t* p; ptrdiff_t stride, pair; int W, H; int16_t v; };
__global__ void __launch_bounds__(256) sgbm_fill_kernel(FillArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x < a.W) a.p[(ptrdiff_t)blockIdx.z * a.pair + (ptrdiff_t)blockIdx.y * a.stride + x] = a.v;
}

struct MedianArgs { const int16_t* src; ptrdiff_t sstride, spair; int16_t* dst; ptrdiff_t dstride, dpair; int W, H; };
__device__ __forceinline__ void cswap(int& a, int& b) { const int lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__global__ void __launch_bounds__(256) sgbm_median_kernel(MedianArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= a.W) return;
    const int16_t* s = a.src + (ptrdiff_t)blockIdx.z * a.spair;
    int v[9];
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++)
            v[(dy + 1) * 3 + dx + 1] = s[(ptrdiff_t)min(max(y + dy, 0), a.H - 1) * a.sstride + min(max(x + dx, 0), a.W - 1)];
    // median of nine by the classic 19-exchange network
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]); cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]); cswap(v[0], v[3]); cswap(v[5], v[8]); cswap(v[4], v[7]);
    cswap(v[3], v[6]); cswap(v[1], v[4]); cswap(v[2], v[5]); cswap(v[4], v[7]); cswap(v[4], v[2]); cswap(v[6], v[4]);
    cswap(v[4], v[2]);
    a.dst[(ptrdiff_t)blockIdx.z * a.dpair + (ptrdiff_t)y * a.dstride + x] = (int16_t)v[4];
}

int sg_fail(int code, const char* msg) { return adf::set_error(code, msg); }

struct DevScope {
    int prev = -1; bool sw = false;
    explicit DevScope(int d) { if (hipGetDevice(&prev) == hipSuccess && prev != d) sw = hipSetDevice(d) == hipSuccess; }
    ~DevScope() { if (sw) hipSetDevice(prev); }
};

int sg_reserve(void** p, size_t* have, size_t need, hipStream_t st)
{
    if (need <= *have) return ADF_OK;
    if (*p) { if (hipStreamSynchronize(st) != hipSuccess) return sg_fail(ADF_EHIP, "hipStreamSynchronize failed"); hipFree(*p); *p = nullptr; *have = 0; }
    need = (need + 255) / 256 * 256;
    hipError_t e = adf::device_malloc(p, need);   // (gives the filter cache's blocks back first if it must)
    if (e != hipSuccess) { *p = nullptr; return sg_fail(e == hipErrorOutOfMemory ? ADF_ENOMEM : ADF_EHIP, "hipMalloc failed for the matcher workspace"); }
    *have = need;
    return ADF_OK;
}

template <int CN>
hipError_t launch_cost(const CostArgs& a, int bs, int n_images, hipStream_t st)
{
    // DD disparities per wave: 16 while the register ring (BS x DD) stays small, 8 for the tall windows
    const int dd = bs <= 5 ? 16 : 8;
    const int outw = 64 - 2 * (bs / 2);
    const int nbands = (a.H + a.rows_per_band - 1) / a.rows_per_band;
    const dim3 grid((a.w1 + outw - 1) / outw, (a.D + 4 * dd - 1) / (4 * dd), nbands * n_images), block(256);
    switch (bs) {
    case 1: hipLaunchKernelGGL((sgbm_cost_kernel<1, CN, 16>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((sgbm_cost_kernel<3, CN, 16>), grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL((sgbm_cost_kernel<5, CN, 16>), grid, block, 0, st, a); break;
    case 7: hipLaunchKernelGGL((sgbm_cost_kernel<7, CN, 8>), grid, block, 0, st, a); break;
    case 9: hipLaunchKernelGGL((sgbm_cost_kernel<9, CN, 8>), grid, block, 0, st, a); break;
    case 11: hipLaunchKernelGGL((sgbm_cost_kernel<11, CN, 8>), grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int DPL>
hipError_t launch_paths(const PathArgs& a0, int mode, int n, hipStream_t st)
{
    PathArgs a = a0;
    hipLaunchKernelGGL((sgbm_path_kernel<DPL, DIR_TOP>), dim3((a.w1 + 3) / 4, n), dim3(256), 0, st, a);      // from above
    // the other accumulating paths of the mode (direction of travel): MODE_SGBM_3WAY: from the left; MODE_SGBM: + the two
    // upper diagonals; MODE_HH: + the three from below (stereo_binary_sgbm.cpp:173-186, 286-301)
    static const int add[6][2] = { {1, 0}, {1, 1}, {-1, 1}, {-1, -1}, {0, -1}, {1, -1} };
    const int nadd = mode == ADF_SGBM_MODE_3WAY ? 1 : mode == ADF_SGBM_MODE_SGBM ? 3 : 6;
    for (int k = 0; k < nadd; k++) {
        a.dx = add[k][0]; a.dy = add[k][1];
        const int nlines = a.dy == 0 ? a.H : (a.dx == 0 ? a.w1 : a.w1 + a.H - 1);
        // the last of them writes its own volume (no read-modify-write of S); the winner kernel adds it in
        if (k == nadd - 1) hipLaunchKernelGGL((sgbm_path_kernel<DPL, DIR_ADD_OUT>), dim3((nlines + 3) / 4, n), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((sgbm_path_kernel<DPL, DIR_ADD>), dim3((nlines + 3) / 4, n), dim3(256), 0, st, a);
    }
    const size_t lds_out = ((size_t)a.w1 + (a.disp12 < 100000 ? 2 * (size_t)a.W : 0)) * 4 * sizeof(int16_t);
    if (lds_out > 150 * 1024) return hipErrorInvalidValue;
    if (lds_out > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sgbm_path_kernel<DPL, DIR_RIGHT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_out);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((sgbm_path_kernel<DPL, DIR_RIGHT>), dim3((a.H + 3) / 4, n), dim3(256), lds_out, st, a);   // from the right + winner
    return hipGetLastError();
}

} // namespace

// ----------------------------------------------------------------------------------------------
// C-ABI (include/adf_wls.h, "semi-global matcher")
// ----------------------------------------------------------------------------------------------
struct adf_sgbm {
    int device = 0;
    int min_disp = 0, num_disp = 16, block = 3;
    int P1 = 0, P2 = 0, cap = 0, uniq = 0, mode = ADF_SGBM_MODE_SGBM;    // cv::StereoSGBM::create's defaults
    int disp12 = 0;                                                        // ... incl. disp12MaxDiff = 0, which the algorithm reads as 1 (check ON)
    void* ws = nullptr; size_t ws_bytes = 0;
    void* stage = nullptr; size_t stage_bytes = 0;
    size_t ws_limit = (size_t)64 << 30;
};

extern "C" int adf_sgbm_create(adf_sgbm_t** out, int min_disparity, int num_disparities, int block_size)
{
    if (!out) return sg_fail(ADF_EBADARG, "out is NULL");
    *out = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return sg_fail(ADF_ENODEV, "no HIP device");
    adf_sgbm* h = new (std::nothrow) adf_sgbm;
    if (!h) return sg_fail(ADF_ENOMEM, "out of host memory");
    h->device = dev; h->min_disp = min_disparity; h->num_disp = num_disparities; h->block = block_size;
    if (const char* e = getenv("ADF_WS_LIMIT_GB")) {
        const double gb = atof(e);
        if (gb > 0) h->ws_limit = (size_t)(gb * (double)((size_t)1 << 30));
    }
    *out = h;
    return ADF_OK;
}

extern "C" void adf_sgbm_destroy(adf_sgbm_t* h)
{
    if (!h) return;
    DevScope ds(h->device);
    if (h->ws) hipFree(h->ws);
    if (h->stage) hipFree(h->stage);
    delete h;
}

extern "C" int adf_sgbm_set_params(adf_sgbm_t* h, int min_disparity, int num_disparities, int block_size, int P1, int P2,
                                   int prefilter_cap, int uniqueness_ratio, int mode)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    h->min_disp = min_disparity; h->num_disp = num_disparities; h->block = block_size;
    h->P1 = P1; h->P2 = P2; h->cap = prefilter_cap; h->uniq = uniqueness_ratio; h->mode = mode;
    return ADF_OK;
}

extern "C" int adf_sgbm_get_params(const adf_sgbm_t* h, int* min_disparity, int* num_disparities, int* block_size, int* P1, int* P2,
                                   int* prefilter_cap, int* uniqueness_ratio, int* mode)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    if (min_disparity) *min_disparity = h->min_disp;
    if (num_disparities) *num_disparities = h->num_disp;
    if (block_size) *block_size = h->block;
    if (P1) *P1 = h->P1;
    if (P2) *P2 = h->P2;
    if (prefilter_cap) *prefilter_cap = h->cap;
    if (uniqueness_ratio) *uniqueness_ratio = h->uniq;
    if (mode) *mode = h->mode;
    return ADF_OK;
}

extern "C" int adf_sgbm_set_disp12_max_diff(adf_sgbm_t* h, int v)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    h->disp12 = v;
    return ADF_OK;
}

extern "C" int adf_sgbm_get_disp12_max_diff(const adf_sgbm_t* h, int* v)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    if (v) *v = h->disp12;
    return ADF_OK;
}

extern "C" int adf_sgbm_get_device(const adf_sgbm_t* h, int* device)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    if (device) *device = h->device;
    return ADF_OK;
}

static int sgbm_check(const adf_sgbm* h, int n, const void* l, const void* r, const void* d, int cn, int W, int H,
                      ptrdiff_t ls, ptrdiff_t rs, ptrdiff_t dstr)
{
    if (!h) return sg_fail(ADF_EBADARG, "handle is NULL");
    if (n <= 0 || !l || !r || !d) return sg_fail(ADF_EBADARG, "views and disparity must be non-NULL, n_pairs positive");
    if (cn != 1 && cn != 3) return sg_fail(ADF_EBADARG, "views must be CV_8UC1 or CV_8UC3");
    if (W <= 0 || H <= 0 || ls < (ptrdiff_t)W * cn || rs < (ptrdiff_t)W * cn || dstr < (ptrdiff_t)W * 2) return sg_fail(ADF_ESIZE, "bad size or stride");
    if ((dstr & 1) || (reinterpret_cast<uintptr_t>(d) & 1)) return sg_fail(ADF_ESIZE, "disparity rows must be 2-byte aligned");
    if (h->mode != ADF_SGBM_MODE_3WAY && h->mode != ADF_SGBM_MODE_SGBM && h->mode != ADF_SGBM_MODE_HH)
        return sg_fail(ADF_EBADARG, "mode must be StereoSGBM::MODE_SGBM, MODE_HH or MODE_SGBM_3WAY");
    if (h->num_disp <= 0 || h->num_disp % 16) return sg_fail(ADF_EBADARG, "numDisparities must be positive and divisible by 16");
    if (h->num_disp > 512) return sg_fail(ADF_EBADARG, "numDisparities above 512 is not supported");
    const int bs = h->block > 0 ? h->block : 5;
    if (bs % 2 == 0 || bs > 11) return sg_fail(ADF_EBADARG, "blockSize must be odd and at most 11");
    if (h->min_disp < -2047 || h->min_disp + h->num_disp > 2047) return sg_fail(ADF_EBADARG, "disparity range does not fit CV_16S with 4 fractional bits");
    if (h->P1 < 0 || h->P2 < 0 || h->P1 > 8000 || h->P2 > 16000) return sg_fail(ADF_EBADARG, "P1 / P2 out of range");
    if (h->disp12 < 100000 && (size_t)(W + 2 * (size_t)W) * 8 > 150 * 1024)
        return sg_fail(ADF_ESIZE, "the matcher's own left-right check (disp12MaxDiff) supports images up to 6400 columns");
    return ADF_OK;
}

extern "C" int adf_sgbm_compute_device(adf_sgbm_t* h, int n_pairs,
                                       const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                                       const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                                       int channels, int W, int H,
                                       int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride,
                                       void* stream)
{
    int rc = sgbm_check(h, n_pairs, left, right, disparity, channels, W, H, left_stride, right_stride, disp_stride);
    if (rc) return rc;
    if (n_pairs > 1 && (disp_pair_stride & 1)) return sg_fail(ADF_ESIZE, "disparity maps must be 2-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    DevScope ds(h->device);
    const int cn = channels, D = h->num_disp, minD = h->min_disp, maxD = minD + D;
    const int bs = h->block > 0 ? h->block : 5;
    const int P1 = h->P1 > 0 ? h->P1 : 2, P2 = std::max(h->P2 > 0 ? h->P2 : 5, P1 + 1);
    const int ur = h->uniq >= 0 ? h->uniq : 10;
    const int ftzero = std::max(h->cap, 15) | 1;
    const int minX1 = std::max(maxD, 0), w1 = (W + std::min(minD, 0)) - minX1;
    const int16_t invalid = (int16_t)((minD - 1) * SG_DISP_SCALE);

    // workspace per image in flight: two signal planes, the C and S volumes, the raw map (every part 256-byte aligned)
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t rec_bytes = up256((size_t)W * H * cn * 3 * 4);
    const size_t rec_pair = rec_bytes / 4;                                            // dwords
    const size_t vol_bytes = up256(w1 > 0 ? (size_t)H * w1 * D * 2 : 0);
    const size_t volp = vol_bytes / 2;                                                // int16 elements
    const size_t raw_bytes = up256((size_t)W * H * 2);
    const size_t raw_el = raw_bytes / 2;
    const size_t per_img = 2 * rec_bytes + 3 * vol_bytes + raw_bytes;     // C, S and the last accumulating path's own volume
    int chunk = (int)(h->ws_limit / per_img);
    if (chunk < 1) chunk = 1;
    if (chunk > n_pairs) chunk = n_pairs;
    rc = sg_reserve(&h->ws, &h->ws_bytes, per_img * (size_t)chunk, st);
    if (rc) return rc;
    char* wsb = (char*)h->ws;
    uint32_t* rec1 = (uint32_t*)wsb;
    uint32_t* rec2 = (uint32_t*)(wsb + rec_bytes * chunk);
    int16_t* Cv = (int16_t*)(wsb + 2 * rec_bytes * chunk);
    int16_t* Sv = (int16_t*)(wsb + 2 * rec_bytes * chunk + vol_bytes * chunk);
    int16_t* Lv = (int16_t*)(wsb + 2 * rec_bytes * chunk + 2 * vol_bytes * chunk);
    int16_t* raw = (int16_t*)(wsb + 2 * rec_bytes * chunk + 3 * vol_bytes * chunk);

    for (int first = 0; first < n_pairs; first += chunk) {
        const int n = std::min(chunk, n_pairs - first);
        const uint8_t* L = left + (ptrdiff_t)first * left_pair_stride;
        const uint8_t* R = right + (ptrdiff_t)first * right_pair_stride;
        int16_t* out = (int16_t*)((char*)disparity + (ptrdiff_t)first * disp_pair_stride);
        FillArgs fa{raw, W, (ptrdiff_t)raw_el, W, H, invalid};
        hipLaunchKernelGGL(sgbm_fill_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, st, fa);
        if (w1 > 0) {
            SignalArgs sa{L, left_stride, left_pair_stride, cn, W, H, ftzero, rec1, rec_pair};
            hipLaunchKernelGGL(sgbm_signals_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, st, sa);
            sa.img = R; sa.stride = right_stride; sa.pair_stride = right_pair_stride; sa.rec = rec2;
            hipLaunchKernelGGL(sgbm_signals_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, st, sa);
            CostArgs ca{rec1, rec2, rec_pair, Cv, volp, W, H, D, minD, minX1, w1, 0};
            // bands: the bs-1 warm-up rows are paid per band; enough workgroups to fill the chip when the image is small
            int rpb = 128;
            while (rpb > 16 && (size_t)((H + rpb - 1) / rpb) * ((w1 + 61) / 62) * ((D + 63) / 64) * n < 2048) rpb >>= 1;
            ca.rows_per_band = rpb;
            if ((size_t)((H + rpb - 1) / rpb) * n > 65535) return sg_fail(ADF_ESIZE, "too many images per call for the cost kernel's grid");
            hipError_t e = cn == 1 ? launch_cost<1>(ca, bs, n, st) : launch_cost<3>(ca, bs, n, st);
            if (e != hipSuccess) return sg_fail(ADF_EHIP, hipGetErrorString(e));
            PathArgs pa{Cv, Sv, Lv, volp, W, H, D, minD, minX1, w1, P1, P2, ur, raw, (ptrdiff_t)W, (ptrdiff_t)raw_el, 0, 0, h->disp12};
            e = D <= 64 ? launch_paths<1>(pa, h->mode, n, st) : D <= 128 ? launch_paths<2>(pa, h->mode, n, st)
              : D <= 256 ? launch_paths<4>(pa, h->mode, n, st) : launch_paths<8>(pa, h->mode, n, st);
            if (e != hipSuccess) return sg_fail(ADF_EHIP, hipGetErrorString(e));
        }
        MedianArgs ma{raw, (ptrdiff_t)W, (ptrdiff_t)raw_el, out, disp_stride / 2, disp_pair_stride / 2, W, H};
        hipLaunchKernelGGL(sgbm_median_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, st, ma);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sg_fail(ADF_EHIP, hipGetErrorString(e));
    return ADF_OK;
}

extern "C" int adf_sgbm_compute_host(adf_sgbm_t* h, int n_pairs,
                                     const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                                     const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                                     int channels, int W, int H,
                                     int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride)
{
    int rc = sgbm_check(h, n_pairs, left, right, disparity, channels, W, H, left_stride, right_stride, disp_stride);
    if (rc) return rc;
    DevScope ds(h->device);
    const size_t vrow = (size_t)W * channels, vbytes = vrow * H, dbytes = (size_t)W * H * 2;
    rc = sg_reserve(&h->stage, &h->stage_bytes, (2 * vbytes + dbytes) * (size_t)n_pairs + 512, nullptr);
    if (rc) return rc;
    uint8_t* dl = (uint8_t*)h->stage;
    uint8_t* dr = dl + vbytes * n_pairs;
    int16_t* dd = (int16_t*)(((uintptr_t)(dr + vbytes * n_pairs) + 255) & ~(uintptr_t)255);
    for (int i = 0; i < n_pairs; i++) {
        if (hipMemcpy2D(dl + vbytes * i, vrow, left + (ptrdiff_t)i * left_pair_stride, left_stride, vrow, H, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy2D(dr + vbytes * i, vrow, right + (ptrdiff_t)i * right_pair_stride, right_stride, vrow, H, hipMemcpyHostToDevice) != hipSuccess)
            return sg_fail(ADF_EHIP, "copying the views to the device failed");
    }
    rc = adf_sgbm_compute_device(h, n_pairs, dl, (ptrdiff_t)vrow, (ptrdiff_t)vbytes, dr, (ptrdiff_t)vrow, (ptrdiff_t)vbytes, channels, W, H,
                                 dd, (ptrdiff_t)W * 2, (ptrdiff_t)dbytes, nullptr);
    if (rc) return rc;
    for (int i = 0; i < n_pairs; i++)
        if (hipMemcpy2D((char*)disparity + (ptrdiff_t)i * disp_pair_stride, disp_stride, (char*)dd + dbytes * i, (size_t)W * 2,
                        (size_t)W * 2, H, hipMemcpyDeviceToHost) != hipSuccess)
            return sg_fail(ADF_EHIP, "copying the disparity map back failed");
    return ADF_OK;
}
