// prep_bodies.h -- device bodies shared by the stand-alone preparation kernels and the merged one (round 3).
//
// A filter call prepares three independent things before its first solve pass: the edge weights (from the guide), the
// confidence map (from the two disparity maps) and the fill outside the ROI.  For batches they are three kernels on two
// streams; for one small frame per call their launch / cross-stream latencies ARE the time, so conf_kernels.hip also
// has one kernel whose blocks take one of the three roles.  The weight kernel's body lives here so that both
// translation units can instantiate it.
#pragma once
#include "adf_internal.h"

namespace adf {
namespace prep {

// streaming outputs are written once and read by a later kernel after gigabytes of other traffic
#define ADF_PREP_ST(p, v) __builtin_nontemporal_store((v), (p))

// ---------------------------------------------------------------------------------------
// Row-major outputs (wave solver; also the exact solver's first stage): a block walks down a strip of
// WS_BCOLS columns, FOUR ADJACENT COLUMNS PER LANE (round 3).  Per row a lane fetches the bytes of its own four
// pixels and of the pixel to their right as ONE aligned window (20 bytes for three channels, 8 for one), keeps its
// previous row's pixels in registers and forms the squared colour distances with packed-byte dot products:
//   sum_c (a_c - b_c)^2 = a.a + b.b - 2 a.b        (v_dot4_u32_u8 on pixels held as [c0 c1 c2 0])
// -- about 18 vector instructions per pixel where the one-pixel-per-thread version (a row staged through LDS, a
// workgroup barrier per row, six LDS byte reads and six multiplies per pixel) spent about 70.  That matters twice:
// the kernel runs beside the confidence kernel, which is bound by vector issue, so every instruction saved here is
// time saved there.  No LDS traffic but the table look-ups, no barrier in the row loop, 16-byte stores.
//
// The guide is read through a buffer descriptor covering exactly the rows of the block: a lane's window may reach up
// to 19 bytes past the last byte its row needs -- past the end of the caller's buffer in the last row of the last
// image -- and the descriptor's range check (per dword for buffer_load_dword / dwordx2 / dwordx4) returns zeros
// there instead of touching memory the caller never promised.
// ---------------------------------------------------------------------------------------
#ifndef ADF_WS_GROUP
#define ADF_WS_GROUP 8
#endif
constexpr int WS_U = ADF_WS_GROUP;             // rows in flight per lane (prefetch group)
constexpr int WS_NT = 128;                     // threads of a block of the streaming weight kernel
constexpr int WS_COLS = 4;                     // columns per lane
constexpr int WS_BCOLS = WS_NT * WS_COLS;      // columns per block
constexpr int WS_LUT_HEAD = 2048;              // entries of the weight table cached in LDS

template <int CH>
struct WsShared {
    float lut_head[WS_LUT_HEAD];
};

typedef unsigned ws_v4u __attribute__((ext_vector_type(4)));
typedef unsigned ws_v2u __attribute__((ext_vector_type(2)));
typedef float ws_v4f __attribute__((ext_vector_type(4)));

template <int CH> struct WsWin;                                   // a lane's raw window of one row
template <> struct WsWin<3> { ws_v4u a; unsigned b; };
template <> struct WsWin<1> { ws_v2u a; };

// Block (bx, by) of nby row blocks, image pz.  `active`: the thread is one of the WS_NT that do the work -- a launch
// with wider blocks (the merged preparation kernel below conf_band_kernel) parks its other waves here: they take part
// in the one barrier (behind the table load) and leave.
template <int CH>
__device__ __forceinline__ void weights_stream_body(const WeightArgs& a, int bx, int by, int nby, size_t pz, WsShared<CH>& sh, bool active)
{
    static_assert(CH == 1 || CH == 3, "guides have one or three channels");
    constexpr int LUT_HEAD = WS_LUT_HEAD;
    float (&lut_head)[LUT_HEAD] = sh.lut_head;
    const Geom& g = a.g;
    const int tid = active ? (int)threadIdx.x : 0;
    if (active)
        for (int q = tid; q < LUT_HEAD; q += WS_NT) lut_head[q] = a.lut[q];
    __syncthreads();
    const int ws_rows = (g.rh + nby - 1) / nby;
    const int x0 = bx * WS_BCOLS, y0 = by * ws_rows;
    if (!active || y0 >= g.rh) return;
#ifdef ADF_WS_PRIO
    __builtin_amdgcn_s_setprio(ADF_WS_PRIO);                    // experiment: few instructions, many bytes -- let them issue first
#endif
    const int nrows = min(ws_rows, g.rh - y0) + 1;             // one extra row feeds the last vertical difference
    const int j0 = x0 + WS_COLS * tid;                         // first ROI column of this lane
    float* chor = a.chor + pz * g.plane;
    float* cvert = a.cvert + pz * g.plane;
    const bool strip = a.cvert_orient == ORIENT_STRIP;

    // buffer descriptor over image rows r_first .. r_last of this pair, base aligned down to a dword; everything it is
    // built from is uniform (kernel arguments and block indices) and said to be so
    const int r_first = g.ry + y0, r_last = g.ry + min(y0 + nrows - 1, g.rh - 1);
    const uintptr_t base0 = reinterpret_cast<uintptr_t>(a.guide) + (uintptr_t)((ptrdiff_t)pz * a.pair_stride + (ptrdiff_t)r_first * a.stride);
    const unsigned mis = (unsigned)__builtin_amdgcn_readfirstlane((int)(base0 & 3u));
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((base0 - mis) & 0xffffffffu));
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((base0 - mis) >> 32));
    const unsigned stride = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a.stride);
    const unsigned records = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((mis + (unsigned)(r_last - r_first) * stride + (unsigned)(g.W * CH) + 3u) & ~3u));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>((uintptr_t)blo | ((uintptr_t)bhi << 32)), (short)0, (int)records, 0x00020000);
    const unsigned col_off = mis + (unsigned)((g.rx + x0) * CH);   // bytes from the aligned base to the block's first pixel, row r_first
    const unsigned lane_off = (unsigned)(tid * WS_COLS * CH);

    // row n of the block = ROI row min(y0+n, rh-1): byte offset of the block's first pixel and its misalignment
    auto row_off = [&](int n) -> unsigned { return col_off + (unsigned)(min(y0 + n, g.rh - 1) - y0) * stride; };
    auto load = [&](int n) -> WsWin<CH> {
        const unsigned wo = (row_off(n) & ~3u) + lane_off;
        WsWin<CH> w;
        if constexpr (CH == 3) {
            w.a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wo, 0, 0);
            w.b = __builtin_amdgcn_raw_buffer_load_b32(rsrc, wo + 16u, 0, 0);
        } else {
            w.a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, wo, 0, 0);
        }
        return w;
    };
    // Head of the table from LDS.  The rare large index (a strong colour edge) is fetched with a SCALAR
    // load, one needy lane at a time: a vector load here -- even one that almost never executes -- makes the
    // compiler wait for vmcnt(0) before every store of the row loop, and on this target stores count in
    // vmcnt too, so every row's stores would wait for the previous row's to be acknowledged.
    auto lookup_big = [&](int idx, float w) -> float {
        bool need = idx >= LUT_HEAD;
        unsigned long long m = __ballot(need);
        while (m) {                                            // wave-uniform
            const int first = __ffsll((long long)m) - 1;
            const int sidx = __builtin_amdgcn_readfirstlane(__shfl(idx, first));
            // constant address space + uniform index = s_load_dword (lgkmcnt, not vmcnt); the table is
            // written once by the host, long before this launch
            const float ws = reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<uintptr_t>(a.lut))[sidx];
            if ((int)(threadIdx.x & 63) == first) { w = ws; need = false; }
            m = __ballot(need);
        }
        return w;
    };

    // per-lane store masks: Chor is 0 in the ROI's last column (FGS.cpp:614) and both planes stay 0 on pitch padding
    unsigned mh[WS_COLS], mv[WS_COLS];
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) { mh[k] = (j0 + k < g.rw - 1) ? 0xffffffffu : 0u; mv[k] = (j0 + k < g.rw) ? 0xffffffffu : 0u; }
    const bool st_ok = j0 < g.pw;
    unsigned ho = (unsigned)y0 * (unsigned)g.pw + (unsigned)j0;                                    // float index of Chor(y0, j0)
    unsigned vo = strip ? ((unsigned)(j0 >> 4) * (unsigned)g.rh + (unsigned)y0) * ADF_STRIP + (unsigned)(j0 & 15)
                        : (unsigned)y0 * (unsigned)g.pw + (unsigned)j0;                            // ... of Cvert(y0, j0)
    const unsigned vstep = strip ? (unsigned)ADF_STRIP : (unsigned)g.pw;

    unsigned q[WS_COLS], qa[WS_COLS];                             // previous row: pixels / their a.a (CH == 1: q[0] holds the four bytes)
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) { q[k] = 0; qa[k] = 0; }
    WsWin<CH> nxt[WS_U], cur[WS_U];
#pragma unroll
    for (int s = 0; s < WS_U; s++) nxt[s] = load(s < nrows ? s : nrows - 1);
    for (int n0 = 0; n0 < nrows; n0 += WS_U) {
#pragma unroll
        for (int s = 0; s < WS_U; s++) {                        // the group's one wait happens here
            cur[s] = nxt[s];
            if constexpr (CH == 3) asm volatile("" : "+v"(cur[s].a), "+v"(cur[s].b)); else asm volatile("" : "+v"(cur[s].a));
        }
#pragma unroll
        for (int s = 0; s < WS_U; s++) nxt[s] = load(min(n0 + WS_U + s, nrows - 1));   // in flight across the rows below
#pragma unroll
        for (int s = 0; s < WS_U; s++) {
            const int n = n0 + s;
            if (n < nrows) {                                   // block-uniform
                const unsigned m = row_off(n) & 3u;            // (uniform) bytes the window starts before the first pixel
                int hidx[WS_COLS], vidx[WS_COLS];
                if constexpr (CH == 3) {
                    const unsigned e0 = __builtin_amdgcn_alignbyte(cur[s].a.y, cur[s].a.x, m), e1 = __builtin_amdgcn_alignbyte(cur[s].a.z, cur[s].a.y, m);
                    const unsigned e2 = __builtin_amdgcn_alignbyte(cur[s].a.w, cur[s].a.z, m), e3 = __builtin_amdgcn_alignbyte(cur[s].b, cur[s].a.w, m);
                    unsigned p[WS_COLS + 1], pa[WS_COLS + 1];  // pixels as [c0 c1 c2 0], and a.a
                    p[0] = e0 & 0x00ffffffu;
                    p[1] = __builtin_amdgcn_perm(e1, e0, 0x0c050403u);
                    p[2] = __builtin_amdgcn_perm(e2, e1, 0x0c040302u);
                    p[3] = e2 >> 8;
                    p[4] = e3 & 0x00ffffffu;
#pragma unroll
                    for (int k = 0; k <= WS_COLS; k++) pa[k] = __builtin_amdgcn_udot4(p[k], p[k], 0u, false);
#pragma unroll
                    for (int k = 0; k < WS_COLS; k++) {
                        hidx[k] = (int)(pa[k] + pa[k + 1]) - 2 * (int)__builtin_amdgcn_udot4(p[k], p[k + 1], 0u, false);
                        vidx[k] = (int)(pa[k] + qa[k]) - 2 * (int)__builtin_amdgcn_udot4(p[k], q[k], 0u, false);
                        q[k] = p[k]; qa[k] = pa[k];
                    }
                } else {
                    const unsigned e = __builtin_amdgcn_alignbyte(cur[s].a.y, cur[s].a.x, m);
                    const unsigned nb = __builtin_amdgcn_alignbyte(0u, cur[s].a.y, m) & 0xffu;
#pragma unroll
                    for (int k = 0; k < WS_COLS; k++) {
                        const int v = (int)((e >> (8 * k)) & 0xffu);
                        const int r = k < WS_COLS - 1 ? (int)((e >> (8 * k + 8)) & 0xffu) : (int)nb;
                        const int u = (int)((q[0] >> (8 * k)) & 0xffu);
                        hidx[k] = (v - r) * (v - r);
                        vidx[k] = (u - v) * (u - v);
                    }
                    q[0] = e;
                }
                float wh[WS_COLS], wv[WS_COLS];
                int big = 0;
#pragma unroll
                for (int k = 0; k < WS_COLS; k++) {
                    wh[k] = lut_head[min(hidx[k], LUT_HEAD - 1)];
                    wv[k] = lut_head[min(vidx[k], LUT_HEAD - 1)];
                    big = max(big, max(hidx[k], vidx[k]));
                }
                if (__ballot(big >= LUT_HEAD)) {               // rare (wave-uniform): a strong edge somewhere in the wave's row
#pragma unroll
                    for (int k = 0; k < WS_COLS; k++) { wh[k] = lookup_big(hidx[k], wh[k]); wv[k] = lookup_big(vidx[k], wv[k]); }
                }
                const int i = y0 + n;                          // ROI row of this input row (when n < nrows-1)
                if (n < nrows - 1 && st_ok) {                  // Chor of this row, FGS.cpp:607-614
                    ws_v4f o;
#pragma unroll
                    for (int k = 0; k < WS_COLS; k++) o[k] = __uint_as_float(__float_as_uint(wh[k]) & mh[k]);
                    ADF_PREP_ST(reinterpret_cast<ws_v4f*>(chor + ho), o);
                }
                if (n >= 1 && st_ok) {                         // Cvert of the previous row, FGS.cpp:635-660 (0 in the last row)
                    const unsigned last = (i - 1 == g.rh - 1) ? 0u : 0xffffffffu;
                    ws_v4f o;
#pragma unroll
                    for (int k = 0; k < WS_COLS; k++) o[k] = __uint_as_float(__float_as_uint(wv[k]) & mv[k] & last);
                    // strip-major (ORIENT_STRIP): four lanes write one 64-byte piece of the strip's stream per row and the
                    // following rows complete the line, so plain stores (L2 merges them); row-major: streaming stores
                    if (strip) *reinterpret_cast<ws_v4f*>(cvert + (vo - vstep)) = o;
                    else ADF_PREP_ST(reinterpret_cast<ws_v4f*>(cvert + (vo - vstep)), o);
                }
                ho += (unsigned)g.pw; vo += vstep;
            }
        }
    }
}


} // namespace prep
} // namespace adf
