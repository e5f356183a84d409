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
#define ADF_WS_GROUP 4
#endif
#ifndef ADF_WS_PAIR_ROWS
#define ADF_WS_PAIR_ROWS 0   // 1: strip-major Cvert is stored two rows at a time, whole 128-byte lines per instruction (round-4 experiment: correct, no faster -- profiles/r04_ab_weights_pair_rows.txt)
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
template <int CH> struct WsPrev { unsigned q[WS_COLS], qa[WS_COLS]; };   // the lane's pixels of the previous row (CH == 1: q[0] = four bytes) and their a.a

// Descriptor over `rows` image rows of `row_bytes` valid bytes each starting at `base0` (any alignment): the base is
// aligned down to a dword (`mis` = bytes skipped) and the record count rounded up to one, so that every byte a window
// may legitimately want lies inside an in-range aligned dword.  All inputs must be uniform; they are said to be so.
struct WsGuide {
    __amdgpu_buffer_rsrc_t rsrc; unsigned mis, stride;
};
__device__ __forceinline__ WsGuide ws_guide(uintptr_t base0, ptrdiff_t stride_bytes, int rows, int row_bytes)
{
    WsGuide gd;
    gd.mis = (unsigned)__builtin_amdgcn_readfirstlane((int)(base0 & 3u));
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((base0 - gd.mis) & 0xffffffffu));
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((base0 - gd.mis) >> 32));
    gd.stride = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)stride_bytes);
    const unsigned records = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((gd.mis + (unsigned)(rows - 1) * gd.stride + (unsigned)row_bytes + 3u) & ~3u));
    gd.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)blo | ((uintptr_t)bhi << 32)), (short)0, (int)records, 0x00020000);
    return gd;
}

// the lane's window of a row: `wo` = byte offset (from the descriptor's base) of the aligned dword holding the lane's first pixel
template <int CH>
__device__ __forceinline__ WsWin<CH> ws_load(const WsGuide& gd, unsigned wo)
{
    WsWin<CH> w;
    if constexpr (CH == 3) {
        w.a = __builtin_amdgcn_raw_buffer_load_b128(gd.rsrc, wo, 0, 0);
        w.b = __builtin_amdgcn_raw_buffer_load_b32(gd.rsrc, wo + 16u, 0, 0);
    } else {
        w.a = __builtin_amdgcn_raw_buffer_load_b64(gd.rsrc, wo, 0, 0);
    }
    return w;
}

// Table indices of a lane's four pixels in one row: hidx[k] = |pixel k - pixel k+1|^2 (FGS.cpp:607-612), vidx[k] =
// |previous row's pixel k - pixel k|^2 (FGS.cpp:640-653); m = bytes the window starts before the first pixel.
template <int CH>
__device__ __forceinline__ void ws_indices(const WsWin<CH>& w, unsigned m, WsPrev<CH>& pv, int (&hidx)[WS_COLS], int (&vidx)[WS_COLS])
{
    if constexpr (CH == 3) {
        const unsigned e0 = __builtin_amdgcn_alignbyte(w.a.y, w.a.x, m), e1 = __builtin_amdgcn_alignbyte(w.a.z, w.a.y, m);
        const unsigned e2 = __builtin_amdgcn_alignbyte(w.a.w, w.a.z, m), e3 = __builtin_amdgcn_alignbyte(w.b, w.a.w, m);
        unsigned p[WS_COLS + 1], pa[WS_COLS + 1];              // pixels as [c0 c1 c2 0], and a.a
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
            vidx[k] = (int)(pa[k] + pv.qa[k]) - 2 * (int)__builtin_amdgcn_udot4(p[k], pv.q[k], 0u, false);
            pv.q[k] = p[k]; pv.qa[k] = pa[k];
        }
    } else {
        const unsigned e = __builtin_amdgcn_alignbyte(w.a.y, w.a.x, m);
        const unsigned nb = __builtin_amdgcn_alignbyte(0u, w.a.y, m) & 0xffu;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) {
            const int v = (int)((e >> (8 * k)) & 0xffu);
            const int r = k < WS_COLS - 1 ? (int)((e >> (8 * k + 8)) & 0xffu) : (int)nb;
            const int u = (int)((pv.q[0] >> (8 * k)) & 0xffu);
            hidx[k] = (v - r) * (v - r);
            vidx[k] = (u - v) * (u - v);
        }
        pv.q[0] = e;
    }
}

// A range-checked window on `bytes` bytes at `p` (uniform inputs): stores through it at an offset beyond `bytes` are
// dropped by the hardware, which is how the row loops below switch a store off WITHOUT a branch around it -- behind a
// branch the compiler cannot count the store among the operations in flight and waits, at every later use of a
// prefetched row, until all older stores have been acknowledged (vmcnt counts loads and stores in issue order).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_window(const void* p, size_t bytes)
{
    const uintptr_t b = reinterpret_cast<uintptr_t>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffu));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)lo | ((uintptr_t)hi << 32)), (short)0, (int)n, 0x00020000);
}
constexpr unsigned WS_DROP = 0x80000000u;      // a byte offset no window reaches (planes stay below 2 GiB: Geom)

// The weights of a lane's row in two steps, a row apart.  ws_lookup_issue: the head of the table from LDS for all eight
// indices, and for every index beyond the head ONE gather from the full table in memory (768 KB, L2-resident) -- issued
// unconditionally through a range-checked window, at an offset no window reaches when the index lies inside the head,
// so lanes that need nothing fetch nothing and no branch surrounds a memory operation.  ws_lookup_finish, a row later:
// a fetched weight is never +0 (the table holds -exp(..): at worst -0), a dropped fetch returns +0 -- that bit pattern
// is the selector.  Round 3: the table indices of NATURAL images exceed the head in 1.5-4 % of the pixels
// (tools/real_guide_time.py); the version before this one fetched those one lane at a time with dependent scalar loads
// inside a branch and took 3.5x as long on the KITTI fixture as on the benchmark's synthetic scene.
struct WsPending { float wl[2 * WS_COLS]; unsigned g[2 * WS_COLS]; };
__device__ __forceinline__ void ws_lookup_issue(const float* lut_head, const __amdgpu_buffer_rsrc_t& lutwin, const int (&hidx)[WS_COLS],
                                                const int (&vidx)[WS_COLS], WsPending& p)
{
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) {
        p.wl[k] = lut_head[min(hidx[k], WS_LUT_HEAD - 1)];
        p.wl[WS_COLS + k] = lut_head[min(vidx[k], WS_LUT_HEAD - 1)];
    }
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) {
        p.g[k] = __builtin_amdgcn_raw_buffer_load_b32(lutwin, hidx[k] >= WS_LUT_HEAD ? (unsigned)hidx[k] * 4u : WS_DROP, 0, 0);
        p.g[WS_COLS + k] = __builtin_amdgcn_raw_buffer_load_b32(lutwin, vidx[k] >= WS_LUT_HEAD ? (unsigned)vidx[k] * 4u : WS_DROP, 0, 0);
    }
}
__device__ __forceinline__ void ws_lookup_finish(const WsPending& p, float (&wh)[WS_COLS], float (&wv)[WS_COLS])
{
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) {
        wh[k] = p.g[k] != 0u ? __uint_as_float(p.g[k]) : p.wl[k];
        wv[k] = p.g[WS_COLS + k] != 0u ? __uint_as_float(p.g[WS_COLS + k]) : p.wl[WS_COLS + k];
    }
}

// Store side of a lane: Chor row-major [rh][pw], Cvert strip-major [pw/16][rh][16] or row-major; Chor is 0 in the ROI's
// last column (FGS.cpp:614), Cvert in its last row (FGS.cpp:658-660), and both planes stay 0 on pitch padding.
struct WsOut {
    __amdgpu_buffer_rsrc_t chor, cvert; unsigned mh[WS_COLS], mv[WS_COLS]; unsigned ho, vo, hstep, vstep; bool st_ok, strip;
    __device__ __forceinline__ void init(const WeightArgs& a, size_t pz, int j0, int y_first)
    {
        const Geom& g = a.g;
        chor = ws_window(a.chor + pz * g.plane, g.plane * sizeof(float));
        cvert = ws_window(a.cvert + pz * g.plane, g.plane * sizeof(float));
        strip = a.cvert_orient == ORIENT_STRIP;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) { mh[k] = (j0 + k < g.rw - 1) ? 0xffffffffu : 0u; mv[k] = (j0 + k < g.rw) ? 0xffffffffu : 0u; }
        st_ok = j0 >= 0 && j0 < g.pw;
        ho = ((unsigned)y_first * (unsigned)g.pw + (unsigned)j0) * 4u;                             // byte offset of Chor(y_first, j0)
        vo = (strip ? ((unsigned)(j0 >> 4) * (unsigned)g.rh + (unsigned)y_first) * ADF_STRIP + (unsigned)(j0 & 15)
                    : (unsigned)y_first * (unsigned)g.pw + (unsigned)j0) * 4u;                     // ... of Cvert(y_first, j0)
        hstep = (unsigned)g.pw * 4u; vstep = (strip ? (unsigned)ADF_STRIP : (unsigned)g.pw) * 4u;
#if ADF_WS_PAIR_ROWS
        pair_back = (unsigned)g.rh * ADF_STRIP * 4u;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) held[k] = 0u;
#endif
    }
    // Chor of ROI row y_first + t (`on`: uniform -- this step has such a row)
    __device__ __forceinline__ void store_h(int t, const float (&wh)[WS_COLS], bool on) const
    {
        ws_v4u o;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) o[k] = __float_as_uint(wh[k]) & mh[k];
#ifdef ADF_WS_TEST_NOSTORE   // timing experiment only (wrong results): the values are formed, nothing is stored
        asm volatile("" :: "v"(o));
        return;
#endif
        __builtin_amdgcn_raw_buffer_store_b128(o, chor, (on && st_ok) ? ho + (unsigned)t * hstep : WS_DROP, 0, 2 /* nt: read much later */);
    }
    // Cvert of ROI row y_first + t (zero when that is the ROI's last row)
    __device__ __forceinline__ void store_v(int t, const float (&wv)[WS_COLS], bool last_row, bool on) const
    {
        const unsigned last = last_row ? 0u : 0xffffffffu;
        ws_v4u o;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) o[k] = __float_as_uint(wv[k]) & mv[k] & last;
#ifdef ADF_WS_TEST_NOSTORE
        asm volatile("" :: "v"(o));
        return;
#endif
        // strip-major: four lanes write one 64-byte piece of the strip's stream per row and the following rows
        // complete the line, so plain stores (L2 merges them); row-major: streaming stores
        const unsigned off = (on && st_ok) ? vo + (unsigned)t * vstep : WS_DROP;
        if (strip) __builtin_amdgcn_raw_buffer_store_b128(o, cvert, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(o, cvert, off, 0, 2);
    }
#if ADF_WS_PAIR_ROWS
    // Strip-major Cvert, two rows at a time (round 4): a strip row is 64 bytes, so a store of ONE row writes half lines
    // -- sixteen 64-byte pieces per instruction.  With the weights of rows t-1 (held) and t (fresh) in hand, the two
    // halves of every group of eight lanes -- two strips -- trade them (lane L and lane L ^ 4), and each of the two
    // stores then writes whole 128-byte lines: strip s rows (t-1, t) from lanes (0..3 | 4..7), then strip s + 1.
    // t - 1 must be even for the pair to start a line (the launcher makes the blocks' first rows even).
    unsigned held[WS_COLS];
    __device__ __forceinline__ void mask_v(const float (&wv)[WS_COLS], bool last_row, unsigned (&o)[WS_COLS]) const
    {
        const unsigned last = last_row ? 0u : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) o[k] = __float_as_uint(wv[k]) & mv[k] & last;
    }
    // (`on`: uniform -- the step has such a row; steps past the block's last row must not disturb what is held)
    __device__ __forceinline__ void hold_v(const float (&wv)[WS_COLS], bool last_row, bool on)
    {
        unsigned o[WS_COLS];
        mask_v(wv, last_row, o);
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) held[k] = on ? o[k] : held[k];
    }
    // rows t - 1 (held) and t (`wv`); `on`: uniform -- both rows exist
    __device__ __forceinline__ void store_v_pair(int t, const float (&wv)[WS_COLS], bool last_row, bool on, int lane) const
    {
        unsigned b[WS_COLS];
        mask_v(wv, last_row, b);
        const bool hi = (lane & 4) != 0;                          // the lane's strip is the second of its group's two
        ws_v4u s1, s2;
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) {
            const unsigned give = hi ? held[k] : b[k];            // what the partner stores for me
            const unsigned got = (unsigned)__shfl_xor((int)give, 4);
            s1[k] = hi ? got : held[k];                           // first strip:  rows t-1 (own, lanes 0..3) | t (from L-4)
            s2[k] = hi ? b[k] : got;                              // second strip: rows t-1 (from L+4) | t (own, lanes 4..7)
        }
        // byte offset of (first strip of the pair, row t-1, this lane's piece) + one row for the upper four lanes
        const unsigned base = vo + (unsigned)(t - 1) * vstep - (hi ? pair_back : 0u) + (hi ? vstep : 0u);
        const bool ok = on && st_ok;
        __builtin_amdgcn_raw_buffer_store_b128(s1, cvert, ok ? base : WS_DROP, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(s2, cvert, ok ? base + pair_back : WS_DROP, 0, 0);
    }
    unsigned pair_back;                                           // bytes from a strip's stream to the next strip's: rh * 64
#endif
};

// Block (bx, by) of nby row blocks, image pz.  `active`: the thread is one of the WS_NT that do the work -- a launch
// with wider blocks (the merged preparation kernel below conf_band_kernel) parks its other waves here: they take part
// in the one barrier (behind the table load) and leave.
template <int CH>
__device__ __forceinline__ void weights_stream_body(const WeightArgs& a, int bx, int by, int nby, size_t pz, WsShared<CH>& sh, bool active)
{
    static_assert(CH == 1 || CH == 3, "guides have one or three channels");
    float (&lut_head)[WS_LUT_HEAD] = sh.lut_head;
    const Geom& g = a.g;
    const int tid = active ? (int)threadIdx.x : 0;
    if (active)
        for (int q = tid; q < WS_LUT_HEAD; q += WS_NT) lut_head[q] = a.lut[q];
    __syncthreads();
    const int ws_rows = (g.rh + nby - 1) / nby;
    const int x0 = bx * WS_BCOLS, y0 = by * ws_rows;
#if ADF_WS_PAIR_ROWS
    const bool pair_rows = (ws_rows & 1) == 0 || nby == 1;       // every block starts on an even ROI row (uniform over the launch)
#endif
    if (!active || y0 >= g.rh) return;
#ifndef ADF_WS_PRIO
#define ADF_WS_PRIO 3
#endif
    // few instructions, many bytes: beside the confidence kernel's waves (bound by vector issue) these go first when they
    // have something to issue -- the two kernels then finish closer together (weights 2.0 -> 1.7 ms, confidence 1.35 ->
    // 1.9 ms beside each other: the pair 2.0 -> 1.9 ms)
    __builtin_amdgcn_s_setprio(ADF_WS_PRIO);
    const int nrows = min(ws_rows, g.rh - y0) + 1;             // one extra row feeds the last vertical difference
    const int j0 = x0 + WS_COLS * tid;                         // first ROI column of this lane

    // image rows r_first .. r_last of this pair behind one descriptor
    const int r_first = g.ry + y0, r_last = g.ry + min(y0 + nrows - 1, g.rh - 1);
    const WsGuide gd = ws_guide(reinterpret_cast<uintptr_t>(a.guide) + (uintptr_t)((ptrdiff_t)pz * a.pair_stride + (ptrdiff_t)r_first * a.stride),
                                a.stride, r_last - r_first + 1, g.W * CH);
    const unsigned col_off = gd.mis + (unsigned)((g.rx + x0) * CH);   // bytes from the aligned base to the block's first pixel, row r_first
    const unsigned lane_off = (unsigned)(tid * WS_COLS * CH);
    // row n of the block = ROI row min(y0+n, rh-1): byte offset of the block's first pixel (its low two bits: the misalignment)
    auto row_off = [&](int n) -> unsigned { return col_off + (unsigned)(min(y0 + n, g.rh - 1) - y0) * gd.stride; };
    auto load = [&](int n) -> WsWin<CH> { return ws_load<CH>(gd, (row_off(n) & ~3u) + lane_off); };

    WsOut out;
    out.init(a, pz, j0, y0);
    const __amdgpu_buffer_rsrc_t lutwin = ws_window(a.lut, sizeof(float) * ADF_LUT_LEVELS);
    WsPrev<CH> pv;
#pragma unroll
    for (int k = 0; k < WS_COLS; k++) { pv.q[k] = 0; pv.qa[k] = 0; }
    WsWin<CH> nxt[WS_U], cur[WS_U];
#pragma unroll
    for (int s = 0; s < WS_U; s++) nxt[s] = load(s < nrows ? s : nrows - 1);
    // Software pipeline, one row deep: row n+1's indices are formed and its look-ups issued BEFORE row n's weights are
    // completed and stored, so a fetch from the full table has a row of work to arrive in.  Every step of a group runs,
    // with no branch around a load or a store: steps past the block's last row work on a repeated row and their stores
    // are switched off through the windows.
    WsPending pend;
    {
        int hidx[WS_COLS], vidx[WS_COLS];
        ws_indices<CH>(nxt[0], row_off(0) & 3u, pv, hidx, vidx);
        ws_lookup_issue(lut_head, lutwin, hidx, vidx, pend);
    }
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
            WsPending ahead;
            {
                int hidx[WS_COLS], vidx[WS_COLS];
                ws_indices<CH>(s + 1 < WS_U ? cur[s + 1 < WS_U ? s + 1 : 0] : nxt[0], row_off(min(n + 1, nrows - 1)) & 3u, pv, hidx, vidx);
                ws_lookup_issue(lut_head, lutwin, hidx, vidx, ahead);
            }
            float wh[WS_COLS], wv[WS_COLS];
            ws_lookup_finish(pend, wh, wv);
            out.store_h(n, wh, n < nrows - 1);                                         // Chor of ROI row y0+n, FGS.cpp:607-614
#if ADF_WS_PAIR_ROWS
            static_assert(WS_U % 2 == 0, "row pairs need an even prefetch group");
            if (out.strip && pair_rows) {
                // Cvert of the previous row (t = n - 1, FGS.cpp:635-660): even t is held, odd t goes out with it as whole lines
                // (t's parity is known at compile time: n0 is a multiple of WS_U, the block's first row is even)
                if (((s + WS_U - 1) & 1) == 0) out.hold_v(wv, y0 + n - 1 == g.rh - 1, n >= 1 && n < nrows);
                else out.store_v_pair(n - 1, wv, y0 + n - 1 == g.rh - 1, n >= 2 && n < nrows, tid);
            } else
#endif
            out.store_v(n - 1, wv, y0 + n - 1 == g.rh - 1, n >= 1 && n < nrows);       // Cvert of the previous row, FGS.cpp:635-660
            pend = ahead;
        }
    }
#if ADF_WS_PAIR_ROWS
    // a block with an odd number of rows (only the ROI's last block, when rh is odd): its last row was held, never paired
    if (out.strip && pair_rows && ((nrows - 1) & 1)) {
        float wv[WS_COLS];
#pragma unroll
        for (int k = 0; k < WS_COLS; k++) wv[k] = __uint_as_float(out.held[k]);
        out.store_v(nrows - 2, wv, false, true);                 // (already masked, the last row already zero)
    }
#endif
}


} // namespace prep
} // namespace adf
