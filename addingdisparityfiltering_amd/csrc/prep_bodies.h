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
// Row-major outputs (wave solver): a block walks down a strip of 256 columns.  Each row's guide bytes
// are fetched once as aligned dwords (prefetched a group of rows ahead), exchanged through a
// double-buffered LDS row; a thread keeps its own pixel of the previous row in registers, so a row
// costs CH + CH LDS byte reads, two table look-ups (head of the LUT cached in LDS once per block) and
// two coalesced 1 KiB stores: Chor of this row and Cvert of the previous one.
// ---------------------------------------------------------------------------------------
// rows per block: gridDim.y row blocks share the ROI's rows (see conf_kernels.hip, row_blocks)
#ifndef ADF_WS_GROUP
#define ADF_WS_GROUP 16
#endif
constexpr int WS_U = ADF_WS_GROUP;
constexpr int WS_NT = 256;         // threads (= columns) of a block of the streaming weight kernel
constexpr int WS_LUT_HEAD = 2048;  // entries of the weight table cached in LDS

template <int CH>
struct WsShared {
    static constexpr int ROWW = ((WS_NT + 1) * CH + 3) / 4 + 1;   // dwords per staged row (incl. misalignment)
    unsigned rowbuf[2][ROWW + 3];
    float lut_head[WS_LUT_HEAD];
};

// Block (bx, by) of nby row blocks, image pz.  `active`: the thread is one of the WS_NT that do the work -- a launch
// with wider blocks (the merged preparation kernel below conf_band_kernel) parks its other waves here: they run the
// same row loop, and therefore the same barriers, with every load, store and table access switched off.
template <int CH>
__device__ __forceinline__ void weights_stream_body(const WeightArgs& a, int bx, int by, int nby, size_t pz, WsShared<CH>& sh, bool active)
{
    constexpr int NT = WS_NT, LUT_HEAD = WS_LUT_HEAD, ROWW = WsShared<CH>::ROWW;
    unsigned (&rowbuf)[2][ROWW + 3] = sh.rowbuf;
    float (&lut_head)[LUT_HEAD] = sh.lut_head;
    const Geom& g = a.g;
    const int tid = active ? (int)threadIdx.x : 0;
    const int ws_rows = (g.rh + nby - 1) / nby;
    const int x0 = bx * NT, y0 = by * ws_rows;
    const unsigned char* gp = a.guide + (ptrdiff_t)pz * a.pair_stride + (ptrdiff_t)(g.rx + x0) * CH;
    const int j = x0 + tid;
    const bool okx = active && j < g.rw;
    const int last_px = min(x0 + NT, g.rw - 1);                // right neighbour of the last ROI column is unused
    const int need = (last_px - x0 + 1) * CH;                  // bytes needed per row
    const int nrows = min(ws_rows, g.rh - y0) + 1;             // one extra row feeds the last vertical difference
    float* chor = a.chor + pz * g.plane;
    float* cvert = a.cvert + pz * g.plane;
    const bool strip = a.cvert_orient == ORIENT_STRIP;

    if (active)
        for (int q = tid; q < LUT_HEAD; q += NT) lut_head[q] = a.lut[q];

    // row n of the block = ROI row min(y0+n, rh-1); returns this thread's aligned dword (or 0)
    auto row_ptr = [&](int n) { return gp + (ptrdiff_t)(g.ry + min(y0 + n, g.rh - 1)) * a.stride; };
    auto load = [&](int n) -> unsigned {
        const unsigned char* rp = row_ptr(n);
        const int m = (int)(reinterpret_cast<uintptr_t>(rp) & 3u);
        const int nw = (m + need + 3) >> 2;
        return (active && tid < nw) ? reinterpret_cast<const unsigned*>(rp - m)[tid] : 0u;
    };
    // Head of the table from LDS.  The rare large index (a strong colour edge) is fetched with a SCALAR
    // load, one needy lane at a time: a vector load here -- even one that almost never executes -- makes the
    // compiler wait for vmcnt(0) before every store of the row loop, and on this target stores count in
    // vmcnt too, so every row's stores would wait for the previous row's to be acknowledged.
    auto lookup = [&](int idx) -> float {
        float w = lut_head[min(idx, LUT_HEAD - 1)];
        bool need = active && idx >= LUT_HEAD;
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

    int prev[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) prev[c] = 0;
    unsigned nxt[WS_U], cur[WS_U];
#pragma unroll
    for (int s = 0; s < WS_U; s++) nxt[s] = (s < nrows) ? load(s) : 0u;
    for (int n0 = 0; n0 < nrows; n0 += WS_U) {
#pragma unroll
        for (int s = 0; s < WS_U; s++) { cur[s] = nxt[s]; asm volatile("" : "+v"(cur[s])); }   // the group's one wait happens here
#pragma unroll
        for (int s = 0; s < WS_U; s++) nxt[s] = (n0 + WS_U + s < nrows) ? load(n0 + WS_U + s) : 0u;  // in flight across the rows below
#pragma unroll
        for (int s = 0; s < WS_U; s++) {
            const int n = n0 + s;
            if (n < nrows) {                                   // block-uniform
                if (active && tid < ROWW) rowbuf[n & 1][tid] = cur[s];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const int m = (int)(reinterpret_cast<uintptr_t>(row_ptr(n)) & 3u);
                const unsigned char* p = reinterpret_cast<const unsigned char*>(rowbuf[n & 1]) + m + tid * CH;
                int px[CH], hidx = 0, vidx = 0;
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    px[c] = p[c];
                    const int dh = px[c] - (int)p[CH + c];
                    const int dv = prev[c] - px[c];
                    hidx += dh * dh; vidx += dv * dv;
                    prev[c] = px[c];
                }
                const int i = y0 + n;                          // ROI row of this input row (when n < nrows-1)
                if (okx) {
                    if (n < nrows - 1)                         // Chor of this row, FGS.cpp:607-614
                        ADF_PREP_ST(&chor[(size_t)i * g.pw + j], (j == g.rw - 1) ? 0.0f : lookup(hidx));
                    if (n >= 1) {                              // Cvert of the previous row, FGS.cpp:635-660
                        // strip-major (ORIENT_STRIP): 16 lanes write one 64-byte piece of the strip's stream
                        // per row and the following rows complete the line, so plain stores (L2 merges them)
                        const float v = (i - 1 == g.rh - 1) ? 0.0f : lookup(vidx);
                        if (strip) cvert[strip_index(i - 1, j, g.rh)] = v;
                        else ADF_PREP_ST(&cvert[(size_t)(i - 1) * g.pw + j], v);
                    }
                }
            }
        }
    }
}


} // namespace prep
} // namespace adf
