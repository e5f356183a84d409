// bm_matcher.hip -- block matcher on the device, so a stereo pair can go from views to filtered disparity
// without leaving HBM.
//
// The reference's filter is fed by cv::StereoBM / cv::StereoSGBM (disparity_filters.cpp:386-449, sample
// disparity_filtering.cpp:151,214), which live in OpenCV's calib3d module and are NOT in the reference tree:
// PARITY UNPINNED at that boundary.  This file implements the published StereoBM algorithm exactly as
// oracle/adf_oracle_bm.c states it (x-Sobel prefilter, SAD block matching, ties to the largest disparity,
// texture / uniqueness tests, sub-pixel fit, 4 fractional bits), bit for bit -- it is integer work -- and keeps
// the conventions the reference does fix: the parameters of the right-view matcher (:421-431) and the
// settings the filter factory forces on the matcher (:389-390, 399-400).
//
// Layout.  The prefiltered views are stored "row-group interleaved": dword (g, x) holds the four
// vertically adjacent pixels (x, 4g .. 4g+3) of column x, one byte each, as value+1 (so no byte is 0); PG
// groups of replicated rows pad the top and the bottom.  One v_sad_u8 / v_msad_u8 then adds four rows of one
// column of the SAD window, and a lane reads its four adjacent columns of a group as one 16-byte load.
//
// Kernel.  One wave = 256 adjacent columns, four per lane (128, two per lane, in the instantiation with the
// uniqueness test; the outer W2 on each side are window halo) x one group of four output rows; the four waves of a workgroup take four consecutive row groups.  Per disparity
// and column a lane forms the four vertical window sums (groups fully inside all four windows are summed
// once, the partial ones through v_msad_u8 with the rows outside the window zeroed in the left operand), two
// disparities are packed in one register (low / high half; a window sum is below 2^16, and the prefix
// arithmetic is exact modulo 2^32), a local prefix plus one DPP wave scan of the lane totals gives the prefix
// over the tile's columns, the neighbours' prefix vectors come through a per-wave LDS slot, and the winner is
// tracked a pair of disparities at a time: the winner pair and the pairs before / after it are captured as whole
// registers (for the uniqueness test also the smallest pair minimum two or more pairs away), and which half won
// is decided once at the end.  Nothing but the int16 result is written.
// The kernel is bound by integer VALU issue (about 11 instructions per pixel and disparity; EXPERIMENTS.md section 10), not
// by HBM: the views are read through L1/L2 once per disparity, HBM sees them once.
#include "adf_internal.h"
#include "../../include/adf_wls.h"

#include <new>

namespace {

constexpr int PG = 3;        // padding groups above and below (half windows up to 10 rows)
constexpr unsigned INF = 0x7fffffffu;

struct PrefilterArgs {
    const uint8_t* src; ptrdiff_t stride, pair_stride;
    uint32_t* dst; size_t dst_pair;   // dwords per view
    int W, Wp, H, HGP, cap;
};

// x-Sobel, clipped to [-cap, cap] and offset by cap (oracle: adf_oracle_bm_prefilter_xsobel); stored +1.
__global__ void __launch_bounds__(256) bm_prefilter_kernel(PrefilterArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, gp = blockIdx.y;
    if (x >= a.W) return;
    const uint8_t* src = a.src + (ptrdiff_t)blockIdx.z * a.pair_stride;
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int y = min(max(4 * (gp - PG) + b, 0), a.H - 1);
        int v = a.cap;
        if (x > 0 && x < a.W - 1) {
            const uint8_t* r0 = src + (ptrdiff_t)(y > 0 ? y - 1 : (a.H > 1 ? 1 : 0)) * a.stride;
            const uint8_t* r1 = src + (ptrdiff_t)y * a.stride;
            const uint8_t* r2 = src + (ptrdiff_t)(y < a.H - 1 ? y + 1 : (a.H > 1 ? a.H - 2 : 0)) * a.stride;
            const int s = ((int)r0[x + 1] - (int)r0[x - 1]) + 2 * ((int)r1[x + 1] - (int)r1[x - 1]) + ((int)r2[x + 1] - (int)r2[x - 1]);
            v = min(max(s, -a.cap), a.cap) + a.cap;
        }
        out |= (uint32_t)(v + 1) << (8 * b);
    }
    a.dst[(size_t)blockIdx.z * a.dst_pair + (size_t)gp * a.Wp + x] = out;
}

// One view's search: its prefiltered "left" (reference) and "right" (searched) images, output and disparity range.
struct ViewArgs {
    const uint32_t* Lt; const uint32_t* Rt;                  // prefiltered views, t_pair dwords per image
    int16_t* disp; ptrdiff_t dstride, dpair;                 // bytes
    int mindisp;
    int xs, xe;            // matched columns [xs, xe)
    int texthr, uniq;
};
// A launch matches one view per image pair (nviews = 1: StereoMatcher::compute) or both views of every pair
// (nviews = 2: the left matcher and the right matcher of disparity_filters.cpp:417-431 in one grid, blockIdx.z =
// 2 * pair + view; the two views' searches are the same size, so one grid covers both).
struct MatchArgs {
    ViewArgs v[2];
    int nviews;
    size_t t_pair;
    int W, Wp, H, HG;                                        // Wp: dwords per prefiltered row (W + XPAD, multiple of 4)
    int ndisp, cap;
};

// rows of group gi (relative to the output group) that lie in the window of output row r
constexpr uint32_t window_mask(int W2, int GB, int gi, int r)
{
    uint32_t m = 0;
    for (int b = 0; b < 4; b++) {
        const int rel = 4 * (gi - GB) + b - r;
        if (rel >= -W2 && rel <= W2) m |= 0xFFu << (8 * b);
    }
    return m;
}
constexpr bool group_common(int W2, int GB, int gi)
{
    for (int r = 0; r < 4; r++) if (window_mask(W2, GB, gi, r) != 0xFFFFFFFFu) return false;
    return true;
}

template <int CTRL, int ROWMASK, bool BOUND>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, BOUND);
}
// inclusive prefix sum over the 64 lanes of the wave
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
    v = dpp_add<0x111, 0xF, true>(v);   // row_shr:1
    v = dpp_add<0x112, 0xF, true>(v);   // row_shr:2
    v = dpp_add<0x114, 0xF, true>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF, true>(v);   // row_shr:8
    v = dpp_add<0x142, 0xA, false>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC, false>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// adjacent columns per lane: four on the filter's path; two with the uniqueness test, whose four extra state
// registers per pixel would otherwise push the kernel past 256 registers (measured per 4K pair: 3.3 ms with four
// columns, 1.98 ms with two)
constexpr int cpl_of(bool uniq) { return uniq ? 2 : 4; }
constexpr int XPAD = 64 * 4 + 4;       // columns of padding right of a prefiltered row (lanes past the image read it)
constexpr int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

template <int W2, bool UNIQ>
__global__ void __launch_bounds__(256) bm_match_kernel(MatchArgs a)
{
    constexpr int CPL = cpl_of(UNIQ);
    constexpr int TILE = 64 * CPL;          // columns per wave, window halo included
    constexpr int TOUT = TILE - 2 * W2;
    constexpr int GB = (W2 + 3) / 4;        // groups above the output group that the windows reach
    constexpr int NG = 2 * GB + 1;
    static_assert(GB <= PG, "padding too small for this window");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bases and the LDS slot stay scalar
    const int g = blockIdx.y * 4 + wave;
    if (g >= a.HG) return;                                           // whole wave; no barrier below
    const int vi = a.nviews == 2 ? (int)(blockIdx.z & 1) : 0;
    const unsigned pz = a.nviews == 2 ? blockIdx.z >> 1 : blockIdx.z;
    const ViewArgs& va = a.v[vi];
    const int xs = va.xs, xe = va.xe, mindisp = va.mindisp, texthr = va.texthr, uniq = va.uniq;
    if (xs + (int)blockIdx.x * TOUT >= xe) return;                  // tile past this view's matched columns
    const int c = xs + (int)blockIdx.x * TOUT - W2 + lane * CPL;     // this lane's first column (>= 0)
    const int Wp = a.Wp;
    const size_t voff = (size_t)pz * a.t_pair + (size_t)(g - GB + PG) * Wp;
    const uint32_t* __restrict__ Lt = va.Lt + voff;
    const uint32_t* __restrict__ Rt = va.Rt + voff;

    // one dword-aligned vector load of CPL adjacent columns; the byte offset is formed in 32 bits so that the load
    // takes the scalar row base plus a 32-bit vector offset
    auto load4 = [&](const uint32_t* row, unsigned x, uint32_t (&d)[CPL]) {   // x .. x+CPL-1 are inside the padded row
        struct __attribute__((aligned(4))) Vec { uint32_t v[CPL]; };
        const Vec q = *reinterpret_cast<const Vec*>(reinterpret_cast<const char*>(row) + x * 4u);
#pragma unroll
        for (int j = 0; j < CPL; j++) d[j] = q.v[j];
    };
    const int cl = min(c, Wp - CPL);
    uint32_t Ld[NG][CPL];
#pragma unroll
    for (int gi = 0; gi < NG; gi++) load4(Lt + (size_t)gi * Wp, cl, Ld[gi]);

    // four vertical window sums (one per output row of the group) of |L - R| down column j of this lane
    auto vertical = [&](const uint32_t (&Rd)[NG][CPL], int j, uint32_t (&V)[4]) {
        uint32_t F = 0;
#pragma unroll
        for (int gi = 0; gi < NG; gi++)
            if (group_common(W2, GB, gi)) F = __builtin_amdgcn_sad_u8(Rd[gi][j], Ld[gi][j], F);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t acc = F;
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                const uint32_t m = window_mask(W2, GB, gi, r);
                if (group_common(W2, GB, gi) || m == 0) continue;
                if (m == 0xFFFFFFFFu) acc = __builtin_amdgcn_sad_u8(Rd[gi][j], Ld[gi][j], acc);
                else acc = __builtin_amdgcn_msad_u8(Rd[gi][j], Ld[gi][j] & m, acc);   // rows with a zero reference byte are skipped
            }
            V[r] = acc;
        }
    };
    // horizontal window sums of the lane's CPL per-column values: the prefix over the tile's columns is a local
    // prefix plus the wave scan of the lane totals; the window sum of column t is Q(t+W2) - Q(t-W2-1), both held
    // by a neighbouring lane in a register known at compile time.  The neighbours' prefixes travel through a
    // per-wave LDS slot as whole vectors (one ds_write_b128, one ds_read_b128 per lane offset): ds_bpermute_b32
    // costs the LDS pipe about 6 cycles per wave and dword, three times the vector path.
    struct __attribute__((aligned(16))) Vec { uint32_t v[CPL]; };
    __shared__ Vec xch[4][4][65];                            // [wave][output row][lane]; entry 64 stays zero = Q(-1)
    constexpr int LH0 = floordiv(W2, CPL), LH1 = floordiv(CPL - 1 + W2, CPL);          // lanes holding Q(t+W2)
    constexpr int LL0 = floordiv(-W2 - 1, CPL), LL1 = floordiv(CPL - 2 - W2, CPL);     // lanes holding Q(t-W2-1)
    const int iH0 = min(lane + LH0, 63), iH1 = min(lane + LH1, 63);
    const int iL0 = lane + LL0 < 0 ? 64 : lane + LL0, iL1 = lane + LL1 < 0 ? 64 : lane + LL1;   // a lane left of the tile: all its columns are
    if (lane < 4) {
        Vec z;
#pragma unroll
        for (int j = 0; j < CPL; j++) z.v[j] = 0;
        xch[wave][lane][64] = z;
    }
    auto horizontal = [&](uint32_t (&v)[CPL], int r) {
        uint32_t tot = v[0];
#pragma unroll
        for (int j = 1; j < CPL; j++) tot += v[j];
        Vec q;                                                // inclusive prefix: the scan result is the last column's
        q.v[CPL - 1] = wave_scan(tot);
#pragma unroll
        for (int j = CPL - 2; j >= 0; j--) q.v[j] = q.v[j + 1] - v[j + 1];
        Vec* slot = xch[wave][r];
        slot[lane] = q;
        __builtin_amdgcn_wave_barrier();                      // same wave: the LDS queue is in order
        const Vec hA = slot[iH0], hB = (LH1 != LH0) ? slot[iH1] : hA;
        const Vec lA = slot[iL0], lB = (LL1 != LL0) ? slot[iL1] : lA;
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            const int th = j + W2, tl = j - W2 - 1;                       // column offsets relative to the lane's first
            const int lh = floordiv(th, CPL), jh = th - lh * CPL, ll = floordiv(tl, CPL), jl = tl - ll * CPL;
            const uint32_t qh = (lh == LH0) ? hA.v[jh] : hB.v[jh];
            const uint32_t ql = (ll == LL0) ? lA.v[jl] : lB.v[jl];
            v[j] = qh - ql;
        }
    };

    // texture of the window: sum of |L - cap| (only needed with a texture threshold)
    uint32_t tex[4][CPL];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < CPL; j++) tex[r][j] = INF;
    if (texthr > 0) {
        uint32_t Rd[NG][CPL], V[CPL][4];
        const uint32_t ft = (uint32_t)(a.cap + 1) * 0x01010101u;
#pragma unroll
        for (int gi = 0; gi < NG; gi++)
#pragma unroll
            for (int j = 0; j < CPL; j++) Rd[gi][j] = ft;
#pragma unroll
        for (int j = 0; j < CPL; j++) vertical(Rd, j, V[j]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t h[CPL];
#pragma unroll
            for (int j = 0; j < CPL; j++) h[j] = V[j][r];
            horizontal(h, r);
#pragma unroll
            for (int j = 0; j < CPL; j++) tex[r][j] = h[j];
        }
    }

    // Winner tracking, a pair of disparities (k, k+1 = low / high half of one register) at a time.  A pair is
    // compared through its smaller half, and when it beats the winner the pair itself (capW), the pair before it
    // (pAt) and -- one step later -- the pair after it (nAt) are kept as whole registers.  Which half won (the
    // high one on a tie: ties go to the larger disparity) and which halves are the neighbours of the sub-pixel fit
    // is decided once, at the end:
    //   winner in the low half  (even k): below = high half of the previous pair, above = high half of capW
    //   winner in the high half (odd k):  below = low half of capW,              above = low half of the next pair
    // `after`: the previous pair became the winner (the previous step's compare result, kept in scalar registers).
    // Uniqueness test (UNIQ): the smallest cost at least two disparities away from the winner = the smallest pair
    // minimum at least two PAIRS before (lmin: the prefix minimum delayed by two steps, taken when the winner
    // changes) or after it (rmin: restarted when the winner changes, fed from the second pair after it on), plus
    // the halves of the two neighbouring pairs that are not adjacent to the winner -- read from pAt / nAt at the end.
    uint32_t best[4][CPL], bk1[4][CPL], capW[4][CPL], pAt[4][CPL], nAt[4][CPL], prev[4][CPL];
    uint32_t lmin[4][CPL], rmin[4][CPL], pm1[4][CPL], pm2[4][CPL];
    bool after[4][CPL];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            best[r][j] = INF; bk1[r][j] = 1u; capW[r][j] = 0; pAt[r][j] = 0; nAt[r][j] = 0; prev[r][j] = 0;
            lmin[r][j] = INF; rmin[r][j] = INF; pm1[r][j] = INF; pm2[r][j] = INF;
            after[r][j] = false;
        }
    auto track_pair = [&](int r, int j, uint32_t k, uint32_t Pk) {
        const uint32_t m = min(Pk & 0xFFFFu, Pk >> 16);
        const bool upd = m <= best[r][j];
        best[r][j] = min(best[r][j], m);
        if (UNIQ) {
            if (!upd && !after[r][j]) rmin[r][j] = min(rmin[r][j], m);
            if (upd) { lmin[r][j] = pm2[r][j]; rmin[r][j] = INF; }
            pm2[r][j] = pm1[r][j]; pm1[r][j] = min(pm1[r][j], m);
        }
        if (after[r][j]) nAt[r][j] = Pk;
        if (upd) { capW[r][j] = Pk; pAt[r][j] = prev[r][j]; bk1[r][j] = k + 1u; }
        after[r][j] = upd;
        prev[r][j] = Pk;
    };

    const int xbase = c - mindisp;                           // R column of disparity index 0; xbase - k >= 0 for matched columns
    auto rcol = [&](int k) -> unsigned { return (unsigned)min(max(xbase - k, 0), Wp - CPL); };
    uint32_t R0[NG][CPL], R1[NG][CPL];
    {
        const unsigned x0 = rcol(0), x1 = rcol(1);
#pragma unroll
        for (int gi = 0; gi < NG; gi++) { load4(Rt + (size_t)gi * (unsigned)Wp, x0, R0[gi]); load4(Rt + (size_t)gi * (unsigned)Wp, x1, R1[gi]); }
    }
    for (int k = 0; k < a.ndisp; k += 2) {
        uint32_t P[4][CPL];
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            uint32_t V0[4], V1[4];
            vertical(R0, j, V0);
            vertical(R1, j, V1);
#pragma unroll
            for (int r = 0; r < 4; r++) P[r][j] = V0[r] + (V1[r] << 16);   // exact: both halves stay below 2^16
        }
        // the next two disparities' columns are fetched while this pair is reduced (the last fetch is unused)
        {
            const unsigned x0 = rcol(k + 2), x1 = rcol(k + 3);
#pragma unroll
            for (int gi = 0; gi < NG; gi++) { load4(Rt + (size_t)gi * (unsigned)Wp, x0, R0[gi]); load4(Rt + (size_t)gi * (unsigned)Wp, x1, R1[gi]); }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            horizontal(P[r], r);
#pragma unroll
            for (int j = 0; j < CPL; j++) track_pair(r, j, (uint32_t)k, P[r][j]);
        }
    }

    const int16_t filtered = (int16_t)((mindisp - 1) * 16);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = 4 * g + r;
        if (y >= a.H) continue;
        int16_t* drow = reinterpret_cast<int16_t*>(reinterpret_cast<char*>(va.disp) + (ptrdiff_t)pz * va.dpair + (ptrdiff_t)y * va.dstride);
#pragma unroll
        for (int j = 0; j < CPL; j++) {
            const int t = lane * CPL + j;
            if (t < W2 || t >= TILE - W2 || c + j >= xe) continue;
            const int kw = (int)bk1[r][j] - 1;                            // first disparity of the winner pair
            const int sb = (int)best[r][j];
            const uint32_t wl = capW[r][j] & 0xFFFFu, wh = capW[r][j] >> 16;
            const uint32_t bl = pAt[r][j] & 0xFFFFu, bh = pAt[r][j] >> 16;  // the pair before (if kw > 0)
            const uint32_t al = nAt[r][j] & 0xFFFFu, ah = nAt[r][j] >> 16;  // the pair after (if kw + 2 < ndisp)
            const bool odd = wh <= wl;
            const int kb = kw + (odd ? 1 : 0);
            const uint32_t below = odd ? wl : bh, above = odd ? al : wh;
            const int pv = (int)(kb > 0 ? below : above);
            const int nv = (int)(kb < a.ndisp - 1 ? above : below);
            const int dd = pv + nv - 2 * sb + abs(pv - nv);
            int16_t out = (int16_t)(((kb + mindisp) * 256 + (dd != 0 ? (pv - nv) * 256 / dd : 0) + 15) >> 4);
            if ((int)tex[r][j] < texthr) out = filtered;
            if (y < W2 || y >= a.H - W2) out = filtered;                 // rows without a full window (calib3d's valid rectangle)
            if (UNIQ && uniq > 0) {                                       // (a view of the launch may have the test off)
                const int thresh = sb + sb * uniq / 100;
                uint32_t other = min(lmin[r][j], rmin[r][j]);
                if (kw > 0) other = min(other, odd ? min(bl, bh) : bl);                // kb-1 is adjacent, kb-2 and kb-3 are not
                if (kw + 2 < a.ndisp) other = min(other, odd ? ah : min(al, ah));      // kb+1 is adjacent, kb+2 and kb+3 are not
                if ((int)other <= thresh) out = filtered;
            }
            drow[c + j] = out;
        }
    }
}

// columns without a full search range or window: (minDisparity - 1) * 16
__global__ void __launch_bounds__(256) bm_border_kernel(MatchArgs a)
{
    const int y = blockIdx.y;
    const int vi = a.nviews == 2 ? (int)(blockIdx.z & 1) : 0;
    const unsigned pz = a.nviews == 2 ? blockIdx.z >> 1 : blockIdx.z;
    const ViewArgs& va = a.v[vi];
    const int nleft = max(min(va.xs, a.W), 0), xr = (va.xe > va.xs) ? max(va.xe, nleft) : nleft;
    const int n = nleft + (a.W - xr);
    const int16_t filtered = (int16_t)((va.mindisp - 1) * 16);
    int16_t* row = reinterpret_cast<int16_t*>(reinterpret_cast<char*>(va.disp) + (ptrdiff_t)pz * va.dpair + (ptrdiff_t)y * va.dstride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) row[i < nleft ? i : xr + (i - nleft)] = filtered;
}

template <bool UNIQ>
hipError_t launch_match(const MatchArgs& a, int w2, dim3 block, int n, hipStream_t st)
{
    const int tout = 64 * cpl_of(UNIQ) - 2 * w2;
    int cols = a.v[0].xe - a.v[0].xs;
    if (a.nviews == 2 && a.v[1].xe - a.v[1].xs > cols) cols = a.v[1].xe - a.v[1].xs;
    dim3 grid((cols + tout - 1) / tout, (a.HG + 3) / 4, n * a.nviews);
    switch (w2) {
#define ADF_BM_CASE(K) case K: hipLaunchKernelGGL((bm_match_kernel<K, UNIQ>), grid, block, 0, st, a); break;
    ADF_BM_CASE(2) ADF_BM_CASE(3) ADF_BM_CASE(4) ADF_BM_CASE(5) ADF_BM_CASE(6)
    ADF_BM_CASE(7) ADF_BM_CASE(8) ADF_BM_CASE(9) ADF_BM_CASE(10)
#undef ADF_BM_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace

// ----------------------------------------------------------------------------------------------
// C-ABI (include/adf_wls.h, "block matcher")
// ----------------------------------------------------------------------------------------------
struct adf_bm {
    int device = 0;
    int min_disp = 0, num_disp = 0, block = 21;
    int cap = 31, texthr = 10, uniq = 15;       // cv::StereoBM's defaults
    void* views = nullptr; size_t views_bytes = 0;   // prefiltered left + right views of the batch
    void* stage = nullptr; size_t stage_bytes = 0;   // host-pointer entry: device copies of the I/O
};

namespace {
int bm_fail(int code, const char* msg) { return adf::set_error(code, msg); }
int reserve(void** p, size_t* have, size_t need, hipStream_t st)
{
    if (need <= *have) return ADF_OK;
    if (*p) { if (hipStreamSynchronize(st) != hipSuccess) return bm_fail(ADF_EHIP, "hipStreamSynchronize failed"); hipFree(*p); *p = nullptr; *have = 0; }
    need = (need + 255) / 256 * 256;
    hipError_t e = adf::device_malloc(p, need);   // (gives the filter cache's blocks back first if it must)
    if (e != hipSuccess) { *p = nullptr; return bm_fail(e == hipErrorOutOfMemory ? ADF_ENOMEM : ADF_EHIP, "hipMalloc failed for the matcher workspace"); }
    *have = need;
    return ADF_OK;
}
struct DevScope {
    int prev = -1; bool sw = false;
    explicit DevScope(int d) { if (hipGetDevice(&prev) == hipSuccess && prev != d) sw = hipSetDevice(d) == hipSuccess; }
    ~DevScope() { if (sw) hipSetDevice(prev); }
};
} // namespace

extern "C" int adf_bm_create(adf_bm_t** out, int num_disparities, int block_size)
{
    if (!out) return bm_fail(ADF_EBADARG, "out is NULL");
    *out = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return bm_fail(ADF_ENODEV, "no HIP device");
    adf_bm* h = new (std::nothrow) adf_bm;
    if (!h) return bm_fail(ADF_ENOMEM, "out of host memory");
    h->device = dev; h->num_disp = num_disparities; h->block = block_size;
    *out = h;
    return ADF_OK;
}

extern "C" void adf_bm_destroy(adf_bm_t* h)
{
    if (!h) return;
    DevScope ds(h->device);
    if (h->views) hipFree(h->views);
    if (h->stage) hipFree(h->stage);
    delete h;
}

extern "C" int adf_bm_set_params(adf_bm_t* h, int min_disparity, int num_disparities, int block_size,
                                 int prefilter_cap, int texture_threshold, int uniqueness_ratio)
{
    if (!h) return bm_fail(ADF_EBADARG, "handle is NULL");
    h->min_disp = min_disparity; h->num_disp = num_disparities; h->block = block_size;
    h->cap = prefilter_cap; h->texthr = texture_threshold; h->uniq = uniqueness_ratio;
    return ADF_OK;
}

extern "C" int adf_bm_get_device(const adf_bm_t* h, int* device)
{
    if (!h) return bm_fail(ADF_EBADARG, "handle is NULL");
    if (device) *device = h->device;
    return ADF_OK;
}

extern "C" int adf_bm_get_params(const adf_bm_t* h, int* min_disparity, int* num_disparities, int* block_size,
                                 int* prefilter_cap, int* texture_threshold, int* uniqueness_ratio)
{
    if (!h) return bm_fail(ADF_EBADARG, "handle is NULL");
    if (min_disparity) *min_disparity = h->min_disp;
    if (num_disparities) *num_disparities = h->num_disp;
    if (block_size) *block_size = h->block;
    if (prefilter_cap) *prefilter_cap = h->cap;
    if (texture_threshold) *texture_threshold = h->texthr;
    if (uniqueness_ratio) *uniqueness_ratio = h->uniq;
    return ADF_OK;
}

static int bm_check(const adf_bm* h, int n, const void* l, const void* r, const void* d, int W, int H,
                    ptrdiff_t ls, ptrdiff_t rs, ptrdiff_t dstr)
{
    if (!h) return bm_fail(ADF_EBADARG, "handle is NULL");
    if (n <= 0 || !l || !r || !d) return bm_fail(ADF_EBADARG, "views and disparity must be non-NULL, n_pairs positive");
    if (W <= 0 || H <= 0 || ls < W || rs < W || dstr < (ptrdiff_t)W * 2) return bm_fail(ADF_ESIZE, "bad size or stride");
    if ((dstr & 1) || (reinterpret_cast<uintptr_t>(d) & 1)) return bm_fail(ADF_ESIZE, "disparity rows must be 2-byte aligned");
    // the checks cv::StereoBM::compute makes on its parameters; the window is limited to 21 so that a
    // window sum fits 16 bits
    if (h->num_disp <= 0 || h->num_disp % 16) return bm_fail(ADF_EBADARG, "numDisparities must be positive and divisible by 16");
    if (h->block < 5 || h->block > 21 || h->block % 2 == 0) return bm_fail(ADF_EBADARG, "blockSize must be odd and within 5..21");
    if (h->block >= (W < H ? W : H)) return bm_fail(ADF_EBADARG, "blockSize must be smaller than the image");
    if (h->cap < 1 || h->cap > 63) return bm_fail(ADF_EBADARG, "preFilterCap must be within 1..63");
    if (h->texthr < 0 || h->uniq < 0) return bm_fail(ADF_EBADARG, "textureThreshold and uniquenessRatio must be non-negative");
    if (h->min_disp < -32768 || h->min_disp + h->num_disp > 2047) return bm_fail(ADF_EBADARG, "disparity range does not fit CV_16S with 4 fractional bits");
    return ADF_OK;
}

// Shared body: prefilter both images once, then one launch for the left view alone (disp_right == NULL) or for
// both views.
static int bm_compute_impl(adf_bm_t* h, int n_pairs,
                           const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                           const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                           int W, int H,
                           int16_t* disp_left, ptrdiff_t dl_stride, ptrdiff_t dl_pair_stride,
                           int16_t* disp_right, ptrdiff_t dr_stride, ptrdiff_t dr_pair_stride,
                           hipStream_t st)
{
    int rc = bm_check(h, n_pairs, left, right, disp_left, W, H, left_stride, right_stride, dl_stride);
    if (rc) return rc;
    if (n_pairs > 1 && (dl_pair_stride & 1)) return bm_fail(ADF_ESIZE, "disparity maps must be 2-byte aligned");
    const int nviews = disp_right ? 2 : 1;
    if (disp_right) {
        if (dr_stride < (ptrdiff_t)W * 2 || (dr_stride & 1) || (reinterpret_cast<uintptr_t>(disp_right) & 1) ||
            (n_pairs > 1 && (dr_pair_stride & 1)))
            return bm_fail(ADF_ESIZE, "bad stride or alignment of the right disparity map");
        if (h->min_disp + h->num_disp - 1 > 32767 || -(h->min_disp + h->num_disp) + 1 < -32768)
            return bm_fail(ADF_EBADARG, "disparity range of the right-view matcher does not fit");
    }
    DevScope ds(h->device);
    const int HG = (H + 3) / 4, HGP = HG + 2 * PG;
    const int Wp = (W + XPAD + 3) / 4 * 4;                     // padded row: lanes past the image read (and discard) it
    const size_t view = (size_t)HGP * Wp;                      // dwords per prefiltered view
    rc = reserve(&h->views, &h->views_bytes, 2 * view * (size_t)n_pairs * sizeof(uint32_t), st);
    if (rc) return rc;
    uint32_t* Lt = (uint32_t*)h->views;
    uint32_t* Rt = Lt + view * (size_t)n_pairs;

    PrefilterArgs p;
    p.W = W; p.Wp = Wp; p.H = H; p.HGP = HGP; p.cap = h->cap; p.dst_pair = view;
    dim3 pgrid((W + 255) / 256, HGP, n_pairs);
    p.src = left; p.stride = left_stride; p.pair_stride = left_pair_stride; p.dst = Lt;
    hipLaunchKernelGGL(bm_prefilter_kernel, pgrid, dim3(256), 0, st, p);
    p.src = right; p.stride = right_stride; p.pair_stride = right_pair_stride; p.dst = Rt;
    hipLaunchKernelGGL(bm_prefilter_kernel, pgrid, dim3(256), 0, st, p);

    const int w2 = h->block / 2;
    auto view_args = [&](const uint32_t* ref, const uint32_t* srch, int16_t* d, ptrdiff_t ds_, ptrdiff_t dp, int md, int texthr, int uniq) {
        ViewArgs v;
        v.Lt = ref; v.Rt = srch; v.disp = d; v.dstride = ds_; v.dpair = dp;
        v.mindisp = md;
        const int maxd = md + h->num_disp - 1;
        v.xs = (maxd > 0 ? maxd : 0) + w2;
        v.xe = W - (md < 0 ? -md : 0) - w2;
        v.texthr = texthr; v.uniq = uniq;
        return v;
    };
    MatchArgs a;
    a.nviews = nviews; a.t_pair = view;
    a.W = W; a.Wp = Wp; a.H = H; a.HG = HG; a.ndisp = h->num_disp; a.cap = h->cap;
    a.v[0] = view_args(Lt, Rt, disp_left, dl_stride, dl_pair_stride, h->min_disp, h->texthr, h->uniq);
    // the right-view matcher of createRightMatcher (disparity_filters.cpp:421-431): views swapped,
    // minDisparity = -(min_disp + num_disp) + 1, texture and uniqueness tests off
    a.v[1] = disp_right ? view_args(Rt, Lt, disp_right, dr_stride, dr_pair_stride, -(h->min_disp + h->num_disp) + 1, 0, 0) : a.v[0];
    hipLaunchKernelGGL(bm_border_kernel, dim3(4, H, n_pairs * nviews), dim3(256), 0, st, a);
    if (a.v[0].xe > a.v[0].xs || (nviews == 2 && a.v[1].xe > a.v[1].xs)) {
        hipError_t e = h->uniq > 0 ? launch_match<true>(a, w2, dim3(256), n_pairs, st)
                                   : launch_match<false>(a, w2, dim3(256), n_pairs, st);
        if (e != hipSuccess) return bm_fail(ADF_EHIP, hipGetErrorString(e));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bm_fail(ADF_EHIP, hipGetErrorString(e));
    return ADF_OK;
}

extern "C" int adf_bm_compute_device(adf_bm_t* h, int n_pairs,
                                     const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                                     const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                                     int W, int H,
                                     int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride,
                                     void* stream)
{
    return bm_compute_impl(h, n_pairs, left, left_stride, left_pair_stride, right, right_stride, right_pair_stride, W, H,
                           disparity, disp_stride, disp_pair_stride, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int adf_bm_compute_both_device(adf_bm_t* h, int n_pairs,
                                          const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                                          const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                                          int W, int H,
                                          int16_t* disp_left, ptrdiff_t disp_left_stride, ptrdiff_t disp_left_pair_stride,
                                          int16_t* disp_right, ptrdiff_t disp_right_stride, ptrdiff_t disp_right_pair_stride,
                                          void* stream)
{
    if (!disp_right) return bm_fail(ADF_EBADARG, "disp_right is NULL");
    return bm_compute_impl(h, n_pairs, left, left_stride, left_pair_stride, right, right_stride, right_pair_stride, W, H,
                           disp_left, disp_left_stride, disp_left_pair_stride,
                           disp_right, disp_right_stride, disp_right_pair_stride, (hipStream_t)stream);
}

extern "C" int adf_bm_compute_host(adf_bm_t* h, int n_pairs,
                                   const uint8_t* left, ptrdiff_t left_stride, ptrdiff_t left_pair_stride,
                                   const uint8_t* right, ptrdiff_t right_stride, ptrdiff_t right_pair_stride,
                                   int W, int H,
                                   int16_t* disparity, ptrdiff_t disp_stride, ptrdiff_t disp_pair_stride)
{
    int rc = bm_check(h, n_pairs, left, right, disparity, W, H, left_stride, right_stride, disp_stride);
    if (rc) return rc;
    DevScope ds(h->device);
    const size_t vbytes = (size_t)W * H, dbytes = (size_t)W * H * 2;
    rc = reserve(&h->stage, &h->stage_bytes, (2 * vbytes + dbytes) * (size_t)n_pairs, nullptr);
    if (rc) return rc;
    uint8_t* dl = (uint8_t*)h->stage;
    uint8_t* dr = dl + vbytes * n_pairs;
    int16_t* dd = (int16_t*)(dr + vbytes * n_pairs);
    for (int i = 0; i < n_pairs; i++) {
        if (hipMemcpy2D(dl + vbytes * i, W, left + (ptrdiff_t)i * left_pair_stride, left_stride, W, H, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy2D(dr + vbytes * i, W, right + (ptrdiff_t)i * right_pair_stride, right_stride, W, H, hipMemcpyHostToDevice) != hipSuccess)
            return bm_fail(ADF_EHIP, "copying the views to the device failed");
    }
    rc = adf_bm_compute_device(h, n_pairs, dl, W, (ptrdiff_t)vbytes, dr, W, (ptrdiff_t)vbytes, W, H, dd, (ptrdiff_t)W * 2, (ptrdiff_t)dbytes, nullptr);
    if (rc) return rc;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return bm_fail(ADF_EHIP, "the matcher kernels failed");
    for (int i = 0; i < n_pairs; i++)
        if (hipMemcpy2D(reinterpret_cast<char*>(disparity) + (ptrdiff_t)i * disp_pair_stride, disp_stride, dd + (size_t)W * H * i, (size_t)W * 2,
                        (size_t)W * 2, H, hipMemcpyDeviceToHost) != hipSuccess)
            return bm_fail(ADF_EHIP, "copying the disparity map back failed");
    return ADF_OK;
}
