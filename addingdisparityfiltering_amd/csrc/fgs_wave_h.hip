// fgs_wave_h.hip -- horizontal pass of the on-chip partitioned solver (see fgs_wave_common.h).
#include "fgs_wave_common.h"

#include <algorithm>

// A/B knobs of the low-resolution fused prologue (tools/ab_lo.sh)
#ifndef ADF_LO_UMASK
#define ADF_LO_UMASK 1      // end-of-row masks only in the float4 group that can reach the row's end (wave-uniform branch)
#endif
#ifndef ADF_LO_NEEDMASK
#define ADF_LO_NEEDMASK 1   // zero-window masks only where a half's staged span leaves the window (wave-uniform branch)
#endif
#ifndef ADF_LO_TAPS_EARLY
#define ADF_LO_TAPS_EARLY 2 // tap table entries requested: 0 = per half, when it has been staged; 1 = per half, before; 2 = all, with the row's first loads
#endif
#ifndef ADF_H_TWO_WAVE_MAX
#define ADF_H_TWO_WAVE_MAX 60   // longest chunk whose two-right-hand-side kernel fits two waves per SIMD
#endif

namespace adf {

namespace {
using namespace wave;

// ---------------------------------------------------------------------------------------------
// Horizontal pass: one wavefront per row, in place.
// ---------------------------------------------------------------------------------------------
// FUSED: first pass of a confidence-mode call -- the right-hand sides are formed on the fly from the
// confidence plane and the left disparity map (U1 = conf, U0 = conf*float(dL), DF.cpp:288-290) instead
// of being read from planes a prologue kernel would have had to write.
// (the longest chunk with two right-hand sides does not fit two waves per SIMD without spilling: the
// pair staging below keeps both right-hand sides and two of the three load batches alive at once)
// NW = 2 (round 3): rows longer than 64 chunks of 64 elements (ROIs wider than 4096 columns: 8K frames) are solved by TWO
// wavefronts of one workgroup, wave w owning columns [w*64*M, (w+1)*64*M) -- its own staging buffer, its own loads and
// stores, the chunk sweeps unchanged -- and meeting the other three times through LDS: the weight in front of chunk 64,
// the left-end coefficients of chunk 64 for chunk 63's separator row, and the 128-row reduced system, which wave 0
// solves (fgs_wave_common.h, reduced128).
// FUSED == 2 (round 4): the down-scaled path's first pass -- the maps are LOW-resolution (the sample's default: matcher on
// half-size views) and cv::resize is part of the prologue: the two source rows an output row taps (confidence and left
// disparity) are staged in the wave's LDS buffer, coalesced, one half of the row's columns at a time, and every lane
// interpolates its own columns from there with the exact tap arithmetic of resize_kernels.hip.  The two view-sized
// planes the resize kernels wrote (6 B/px) and this pass read back (6 B/px) never exist.
// FUSE_LO_HALF: FUSE_LO for maps of exactly half the view's width (the sample's default) on a ROI that starts on an even
// column >= 2: the four columns of a float4 group then share FOUR consecutive source elements whatever the lane -- columns
// 2m, 2m+1, 2m+2, 2m+3 tap (m-1, m), (m, m+1), (m, m+1), (m+1, m+2) -- so the group makes three LDS reads instead of eight
// and decodes one tap position instead of four.  Same operands, same arithmetic: bit-identical to FUSE_LO (tests).
constexpr int FUSE_NONE = 0, FUSE_VIEW = 1, FUSE_LO = 2, FUSE_LO_HALF = 3;

// cv::resize's INTER_LINEAR tap of destination index d: source index s0 (and s0 + 1) with weights (1 - fx, fx);
// borders clamp with weight (1, 0).  Same operations, same order as resize_linear_kernel / the oracle (host and device).
__host__ __device__ __forceinline__ void lin_tap(int d, double scale, int sn, int& s0, float& fx)
{
    fx = (float)(((double)d + 0.5) * scale - 0.5);
    s0 = (int)floorf(fx);
    fx -= (float)s0;
    if (s0 < 0) { fx = 0.0f; s0 = 0; }
    if (s0 >= sn - 1) { fx = 0.0f; s0 = sn - 1; }
}
// The taps of a call's ROI columns are the same for every row and every pair: lo_tap_table_kernel forms them once per
// call (the double arithmetic and the conversions are slow instructions; per row they cost more than the interpolation).
// table[j], j < n (n = the row's float4 count * 4) = the tap of ROI column min(j, len - 1) as ONE float: the source
// coordinate with the border rules already applied -- s0 + fx, which is exact: fx is what the coordinate's own
// fraction bits hold -- so that the row pass recovers s0 = (int)t and fx = fract(t) in two instructions.
__global__ void __launch_bounds__(256) lo_tap_table_kernel(float* table, int n, int len, int hi_x0, double scale, int sn)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int s0; float fx;
    lin_tap(hi_x0 + min(j, len - 1), scale, sn, s0, fx);
    table[j] = (float)s0 + fx;
}
// floats per staged confidence row = shorts per staged disparity row: two of each fit the wave's M*256-byte buffer
// (the shortest bucket's buffer is enlarged instead: its single float4 group per lane spans 256 columns)
__host__ __device__ constexpr int lo_row_cap(int m) { return (((m * 64) / 3) & ~7) < 168 ? 168 : (((m * 64) / 3) & ~7); }
__host__ __device__ constexpr int lo_stage_vec4(int m) { return 12 * lo_row_cap(m) > 256 * m ? (12 * lo_row_cap(m) + 15) / 16 : m * 16; }

template <int M, int R, int FUSED, int NW = 1>
__global__ void __launch_bounds__(64 * NW, (M > (NW == 2 ? 40 : ADF_H_TWO_WAVE_MAX) && R > 1) ? 1 : 2) wave_hpass_kernel(WavePassArgs a)
{
    static_assert(M % 4 == 0 && M >= 4, "chunk length must be a multiple of 4");
    static_assert(NW == 1 || NW == 2, "one or two wavefronts per row");
    constexpr bool LO = FUSED == FUSE_LO || FUSED == FUSE_LO_HALF;
    __shared__ float4 stage_all[NW][LO ? lo_stage_vec4(M) : M * 16];
    __shared__ float xch[NW == 2 ? 5 : 1];              // c in front of chunk 64; GS0, GS1, PS, QS of chunk 64
    __shared__ float red[NW == 2 ? 5 : 1][NW == 2 ? 128 : 1];   // separator rows
    __shared__ float xsol[NW == 2 ? 2 : 1][NW == 2 ? 128 : 1];  // their solutions
    const int lane = threadIdx.x & 63, wv = NW == 2 ? (int)(threadIdx.x >> 6) : 0;
    float4* stage = stage_all[wv];
    const int v0 = wv * 16 * M;                          // first float4 of this wave's columns in a row-major row
    const size_t off = (size_t)blockIdx.y * a.plane + (size_t)blockIdx.x * a.pitch;
    const int nvec = a.pitch >> 2;
    // R == 2: the two right-hand sides live in one pair plane, interleaved per 16 columns
    // ([U0 x16 | U1 x16] per strip, see fgs_wave_common.h): a row is 2*pitch contiguous floats
    constexpr bool PAIR = R > 1;
    // (rows come in tiles of TR: float4 #q of pair row r lives at (r/TR)*(TR*nvecU) + (q/8)*8*TR + (r%TR)*8 + q%8)
    constexpr int TR = ADF_TILE_ROWS;
    const size_t offU = PAIR ? (size_t)blockIdx.y * 2 * a.plane + (size_t)(blockIdx.x / TR) * (size_t)(2 * TR * a.pitch) + (size_t)(blockIdx.x % TR) * 32 : off;
#define ADF_PIDX(q) (PAIR ? ((((q) >> 3) * (8 * TR)) + ((q) & 7)) : (q))
    const int nvecU = PAIR ? 2 * nvec : nvec;
    constexpr int MQ = M / 4;
    float c[1][M], f0[1][M], f1[1][M];

    // All of the row's coalesced loads (16 B per lane, 1 KiB per instruction) are issued before the
    // first use so that a row pays one memory latency, not one per plane; each plane is then turned
    // from "float4 #(64k+lane)" into "chunk of lane" through the wave's LDS staging buffer.
    float4 tC[M / 4], t0[M / 4], t1[M / 4];
    {
        const float4* sC = reinterpret_cast<const float4*>(a.C + off);
        // PAIR: t0 / t1 hold the first / second half of the interleaved row instead of U0 / U1
        const float4* s0 = reinterpret_cast<const float4*>(a.U0 + offU);
        // fused inputs (the launcher guarantees a 16-byte aligned conf row start and len >= 4)
        const float4* sF = nullptr; const char* sD = nullptr;
        if (FUSED == FUSE_VIEW) {
            sF = reinterpret_cast<const float4*>(a.conf_in + (size_t)blockIdx.y * a.conf_frame +
                                                 (size_t)(a.conf_y0 + blockIdx.x) * a.conf_pitch + a.conf_x0);
            sD = reinterpret_cast<const char*>(a.dl_in) + (ptrdiff_t)blockIdx.y * a.dl_pair_stride +
                 (ptrdiff_t)(a.dl_y0 + blockIdx.x) * a.dl_stride + (ptrdiff_t)a.dl_x0 * 2;
        }
        // fused: the row is ceil(len/4) vectors; conf (the library's own plane, Geom::cx0 / cpitch) is always 16-byte
        // aligned, dL is the caller's and only 2-byte aligned for an odd ROI x (8-byte loads at any even address).
        // A ROI width that is not a multiple of 4 ends in a partial vector: its dL load is moved back so that it
        // ends with the row (never past the caller's buffer) and shifted into place afterwards, its conf elements
        // past the row are cleared -- both in the second loop, behind one wave-uniform branch, so that no loaded
        // value is touched while loads are still being issued.
        const int nfull = a.len >> 2, rem = a.len & 3;
        const int nfused = nfull + (rem ? 1 : 0);
        const unsigned dl_last = (unsigned)a.len * 2u - 8u;      // byte offset of the last whole vector (len >= 4)
        typedef short v4s_u __attribute__((ext_vector_type(4), aligned(2)));
        short4 draw[FUSED == FUSE_VIEW ? M / 4 : 1];             // fused: the row of the left disparity map
        // (an explicit branch per load: "cond ? *p : zero" would make the compiler select between
        // addresses and park the zero in scratch memory)
        // (idx: float4 of the row-major row; uidx: float4 of this wave's part of the interleaved pair row, 2 * M * 64 floats)
        const int u0 = PAIR ? 2 * v0 : v0;
        if constexpr (LO) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef short s8u __attribute__((ext_vector_type(8), aligned(2)));
            constexpr int KH = (MQ + 1) / 2;                      // float4 groups of the first half of the wave's columns
            constexpr int CROW = lo_row_cap(M);
            constexpr int TC4 = (CROW / 4 + 63) / 64, TD8 = (CROW / 8 + 63) / 64;
            static_assert(12 * CROW <= 16 * lo_stage_vec4(M) && CROW % 8 == 0, "two confidence rows and two disparity rows fit the staging buffer");
            const int sw = a.lo_w, sh = a.lo_h;
            // the C row first: its loads are in flight while the low-resolution rows are fetched and staged
#pragma unroll
            for (int k = 0; k < MQ; k++) {
                const int idx = v0 + 64 * k + lane;
                tC[k] = make_float4(0.f, 0.f, 0.f, 0.f); t0[k] = tC[k]; t1[k] = tC[k];
                if (idx < nvec) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sC) + idx); tC[k] = make_float4(q.x, q.y, q.z, q.w); }
            }
            // rows (wave-uniform)
            int sy; float fy;
            {
                const int dy = a.hi_y0 + (int)blockIdx.x;
                fy = (float)(((double)dy + 0.5) * a.lo_scale_y - 0.5);
                sy = (int)floorf(fy);
                fy -= (float)sy;                                 // (rows clamp with their weights kept, like the resize kernel)
            }
            const float b0 = 1.0f - fy, b1 = fy;
            const bool post_scaled = a.lo_post_scale != 1.0f;
            int yr[2] = {min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)};
            yr[0] = __builtin_amdgcn_readfirstlane(yr[0]); yr[1] = __builtin_amdgcn_readfirstlane(yr[1]);
            const float* cbase = a.lo_conf + (ptrdiff_t)blockIdx.y * a.lo_conf_pair;
            const char* dbase = reinterpret_cast<const char*>(a.lo_dl) + (ptrdiff_t)blockIdx.y * a.lo_dl_pair;
            // columns of the two halves (wave-uniform): first source element and number of staged slots (the slot behind
            // the last tap included: past the row's end it repeats the edge element, which carries weight 0)
            int ss[2], ns[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int cfirst = 64 * M * wv + 256 * (hh ? KH : 0);
                const int clast = min(64 * M * wv + 256 * (hh ? MQ : KH), a.len) - 1;
                int s_first = 0, s_last = -2; float f_;
                if (clast >= cfirst) { lin_tap(a.hi_x0 + cfirst, a.lo_scale_x, sw, s_first, f_); lin_tap(a.hi_x0 + clast, a.lo_scale_x, sw, s_last, f_); }
                ss[hh] = __builtin_amdgcn_readfirstlane(s_first);
                ns[hh] = __builtin_amdgcn_readfirstlane(s_last + 2 - s_first);      // 0 for an empty half
            }
            // does a staged element of the half lie outside the confidence map's window (wave-uniform)?
            bool need_mask[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
                need_mask[hh] = a.lo_zero_outside && (!ADF_LO_NEEDMASK || yr[0] < a.lo_vy0 || yr[1] >= a.lo_vy1 || ss[hh] < a.lo_vx0 || min(ss[hh] + ns[hh], sw) > a.lo_vx1);
            // Per half: fetch the two source rows (coalesced: lane i takes source elements ss + 4i .. of the confidence
            // rows, ss + 8i .. of the disparity rows; a vector that would cross the row's end is fetched element by
            // element, clamped, which also fills the slots behind the row with the edge element), stage them, tap them.
            // The second half's loads are issued when the first half has been staged, so they fly during its taps and
            // only one half's raw rows occupy registers.
            v4f rc[2][2][TC4]; s8u rd[2][2][TD8];
            auto fetch = [&](int hh, v4f (&qc)[2][TC4], s8u (&qd)[2][TD8]) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const float* crow = cbase + (ptrdiff_t)yr[r] * a.lo_conf_stride;
                    const int16_t* drow = reinterpret_cast<const int16_t*>(dbase + (ptrdiff_t)yr[r] * a.lo_dl_stride);
#pragma unroll
                    for (int t = 0; t < TC4; t++) {
                        const int e = ss[hh] + 4 * (64 * t + lane);
                        qc[r][t] = v4f{0.f, 0.f, 0.f, 0.f};
                        if (4 * (64 * t + lane) < ns[hh]) {
                            if (e + 3 < sw) qc[r][t] = *reinterpret_cast<const f4u*>(crow + e);
                            else {
#pragma unroll
                                for (int c = 0; c < 4; c++) qc[r][t][c] = crow[min(e + c, sw - 1)];
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < TD8; t++) {
                        const int e = ss[hh] + 8 * (64 * t + lane);
                        qd[r][t] = s8u{0, 0, 0, 0, 0, 0, 0, 0};
                        if (8 * (64 * t + lane) < ns[hh]) {
                            if (e + 7 < sw) qd[r][t] = *reinterpret_cast<const s8u*>(drow + e);
                            else {
#pragma unroll
                                for (int c = 0; c < 8; c++) qd[r][t][c] = drow[min(e + c, sw - 1)];
                            }
                        }
                    }
                }
            };
            // LDS layout: slot s of the half holds (row0[s], row1[s]) -- two floats of the confidence rows, then, behind
            // all confidence slots, two shorts of the disparity rows -- so that a tap's four values (slots s and s + 1
            // of both rows) are ONE 16-byte / ONE 8-byte read, and a lane, which holds the same source elements of
            // both rows, stages them with whole 16-byte writes.
            const float* Lf = reinterpret_cast<const float*>(stage);
            const short* Ls = reinterpret_cast<const short*>(stage) + 4 * CROW;      // behind the 2 * CROW confidence floats
            v4f* Lf4 = reinterpret_cast<v4f*>(stage);
            typedef short s8a __attribute__((ext_vector_type(8)));
            s8a* Ls8 = reinterpret_cast<s8a*>(stage) + CROW / 2;
            auto put = [&](int hh, const v4f (&qc)[2][TC4], const s8u (&qd)[2][TD8]) {
                const bool rin0 = !a.lo_zero_outside || (yr[0] >= a.lo_vy0 && yr[0] < a.lo_vy1);
                const bool rin1 = !a.lo_zero_outside || (yr[1] >= a.lo_vy0 && yr[1] < a.lo_vy1);
#pragma unroll
                for (int t = 0; t < TC4; t++) {
                    const int i4 = 64 * t + lane, e = ss[hh] + 4 * i4;
                    v4f q0 = qc[0][t], q1 = qc[1][t];
                    if (need_mask[hh]) {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int ec = min(e + c, sw - 1);
                            const bool cin = ec >= a.lo_vx0 && ec < a.lo_vx1;
                            if (!(rin0 && cin)) q0[c] = 0.0f;
                            if (!(rin1 && cin)) q1[c] = 0.0f;
                        }
                    }
                    if (4 * i4 < CROW) {
                        Lf4[2 * i4] = v4f{q0[0], q1[0], q0[1], q1[1]};
                        Lf4[2 * i4 + 1] = v4f{q0[2], q1[2], q0[3], q1[3]};
                    }
                }
#pragma unroll
                for (int t = 0; t < TD8; t++) {
                    const int i8 = 64 * t + lane;
                    const s8u q0 = qd[0][t], q1 = qd[1][t];
                    if (8 * i8 < CROW) {
                        Ls8[2 * i8] = s8a{q0[0], q1[0], q0[1], q1[1], q0[2], q1[2], q0[3], q1[3]};
                        Ls8[2 * i8 + 1] = s8a{q0[4], q1[4], q0[5], q1[5], q0[6], q1[6], q0[7], q1[7]};
                    }
                }
            };
            typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
            typedef short s4a4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f bb0 = {b0, b0}, bb1 = {b1, b1};
            // the columns' taps (one float per column, the same for every row of the call: L2 hits)
            v4f tp[MQ];
            auto load_taps = [&](int hh) {
#pragma unroll
                for (int k = (hh ? KH : 0); k < (hh ? MQ : KH); k++) {
                    const int idx = v0 + 64 * k + lane;
                    tp[k] = v4f{0.f, 0.f, 0.f, 0.f};
                    if (idx < nfused) tp[k] = reinterpret_cast<const v4f*>(a.lo_taps)[idx];
                }
            };
            fetch(0, rc[0], rd[0]);
#if ADF_LO_TAPS_EARLY == 2
            load_taps(0); load_taps(1);
#endif
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
#if ADF_LO_TAPS_EARLY == 1
                load_taps(hh);
#endif
                put(hh, rc[hh], rd[hh]);
                __syncthreads();
                if (hh == 0) {
                    asm volatile("" ::: "memory");               // (the next half's loads: not before this half is staged)
                    fetch(1, rc[1], rd[1]);
                }
#if ADF_LO_TAPS_EARLY == 0
                load_taps(hh);
#endif
#pragma unroll
                for (int k = (hh ? KH : 0); k < (hh ? MQ : KH); k++) {
                    // (opaque: nothing here depends on a load, and the compiler would otherwise form every group's
                    // taps while the loads are in flight, spilling the row)
                    int lane_t = lane;
                    asm volatile("" : "+v"(lane_t) :: "memory");
                    const int idx = v0 + 64 * k + lane_t;
                    // all four columns' staged values first (eight LDS reads in flight), the arithmetic afterwards
                    f4a8 cq[4]; s4a4 dq[4]; float fxs[4];
                    if constexpr (FUSED == FUSE_LO_HALF) {
                        typedef short s8a4 __attribute__((ext_vector_type(8), aligned(4)));
                        // (the group's first column taps m - 1: in range by construction, clamped like the general form)
                        const int sl0 = min(max((int)tp[k][0] - ss[hh], 0), CROW - 4);
                        const f4a8 ca = *reinterpret_cast<const f4a8*>(Lf + 2 * sl0), cb = *reinterpret_cast<const f4a8*>(Lf + 2 * sl0 + 4);
                        const s8a4 da = *reinterpret_cast<const s8a4*>(Ls + 2 * sl0);
                        cq[0] = f4a8{ca[0], ca[1], ca[2], ca[3]}; dq[0] = s4a4{da[0], da[1], da[2], da[3]};
                        cq[1] = f4a8{ca[2], ca[3], cb[0], cb[1]}; dq[1] = s4a4{da[2], da[3], da[4], da[5]};
                        cq[2] = cq[1]; dq[2] = dq[1];
                        cq[3] = f4a8{cb[0], cb[1], cb[2], cb[3]}; dq[3] = s4a4{da[4], da[5], da[6], da[7]};
                        // the weights from the table all the same: 0.75 / 0.25, and 0 at the frame's clamped last column
#pragma unroll
                        for (int c = 0; c < 4; c++) fxs[c] = __builtin_amdgcn_fractf(tp[k][c]);
                    } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float t = tp[k][c];
                        fxs[c] = __builtin_amdgcn_fractf(t);                          // exact: t = s0 + fx, t >= 0
                        const int sl = min(max((int)t - ss[hh], 0), CROW - 2);          // (in range by construction; the clamp keeps a bug from reading other waves' LDS)
                        cq[c] = *reinterpret_cast<const f4a8*>(Lf + 2 * sl);         // conf: row0[s], row1[s], row0[s+1], row1[s+1]
                        dq[c] = *reinterpret_cast<const s4a4*>(Ls + 2 * sl);         // disparity, the same four
                    }
                    }
                    float cv[4], dv[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        // {confidence, disparity} side by side: packed multiplies / adds, every rounding where the scalar
                        // statement has it (products and sums separately: no fused multiply-add)
                        const float a0 = 1.0f - fxs[c], a1 = fxs[c];
                        const v2f aa0 = {a0, a0}, aa1 = {a1, a1};
                        const v2f p0 = {cq[c][0], (float)dq[c][0]}, p1 = {cq[c][1], (float)dq[c][1]};     // rows 0 / 1 at s
                        const v2f n0 = {cq[c][2], (float)dq[c][2]}, n1 = {cq[c][3], (float)dq[c][3]};     // ... at s + 1
                        const v2f h0 = p0 * aa0 + n0 * aa1, h1 = p1 * aa0 + n1 * aa1;
                        const v2f v = h0 * bb0 + h1 * bb1;                          // DF.cpp:274 | DF.cpp:272
                        // saturate_cast<short> twice (DF.cpp:272, then x_ratio, :273) without branches: both arguments are
                        // finite and far inside the int range here (a convex combination of int16 values; that times the
                        // size ratio), so sat16's guard for NaN / out-of-int-range cannot fire -- and a branch per column
                        // would let the compiler sink every column's arithmetic behind the last one's (registers)
                        const float q1 = fminf(fmaxf(rintf(v[1]), -32768.0f), 32767.0f);
                        const float q2 = fminf(fmaxf(rintf(q1 * a.lo_post_scale), -32768.0f), 32767.0f);
                        cv[c] = v[0];
                        dv[c] = post_scaled ? q2 : q1;
                    }
                    // columns behind the row's end (the last, partial float4 and the lanes past it) are zero
                    if (!ADF_LO_UMASK || 4 * (v0 + 64 * k + 64) > a.len) {       // (wave-uniform: only the group that holds the row's end, and those past it)
                        const int left = a.len - 4 * idx;
#pragma unroll
                        for (int c = 0; c < 4; c++) { const bool on = c < left; cv[c] = on ? cv[c] : 0.0f; dv[c] = on ? dv[c] : 0.0f; }
                    }
                    t1[k] = make_float4(cv[0], cv[1], cv[2], cv[3]);               // U1 = conf, U0 = conf * float(dL)  (DF.cpp:288-290)
                    t0[k] = make_float4(cv[0] * dv[0], cv[1] * dv[1], cv[2] * dv[2], cv[3] * dv[3]);
                    // (pinned here: nothing reads t0 / t1 before the transposes, and the compiler would sink the arithmetic
                    // down to them, holding sixteen staged values per column in registers all the way)
                    asm volatile("" : "+v"(t1[k].x), "+v"(t1[k].y), "+v"(t1[k].z), "+v"(t1[k].w),
                                      "+v"(t0[k].x), "+v"(t0[k].y), "+v"(t0[k].z), "+v"(t0[k].w));
                    __builtin_amdgcn_sched_barrier(0);           // one float4 of columns at a time (register pressure)
                }
                __syncthreads();
            }
        } else {
#pragma unroll
        for (int k = 0; k < M / 4; k++) {
            const int idx = v0 + 64 * k + lane, uidx = u0 + 64 * k + lane;
            tC[k] = make_float4(0.f, 0.f, 0.f, 0.f); t0[k] = tC[k]; t1[k] = tC[k];
            if (FUSED == FUSE_VIEW) draw[k] = make_short4(0, 0, 0, 0);
            if (FUSED == FUSE_VIEW) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                if (idx < nvec) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sC) + idx); tC[k] = make_float4(q.x, q.y, q.z, q.w); }
                // loads only: the products conf*float(dL) wait for the second loop, or every iteration would
                // wait for its own loads before the next one's are issued (14 memory latencies per row)
                if (idx < nfused) {
                    const v4f cq = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sF) + idx);
                    const unsigned doff = min((unsigned)idx * 8u, dl_last);
                    const v4s_u dq = __builtin_nontemporal_load(reinterpret_cast<const v4s_u*>(sD + doff));
                    t1[k] = make_float4(cq.x, cq.y, cq.z, cq.w);
                    draw[k] = make_short4(dq.x, dq.y, dq.z, dq.w);
                }
            } else {
                // non-temporal: every byte of a row pass is used exactly once (measured -5 % on the pass)
                typedef float v4f __attribute__((ext_vector_type(4)));
                if (idx < nvec) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(sC) + idx); tC[k] = make_float4(q.x, q.y, q.z, q.w); }
                if (uidx < nvecU) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s0) + ADF_PIDX(uidx)); t0[k] = make_float4(q.x, q.y, q.z, q.w); }
                if (PAIR && uidx + 64 * MQ < nvecU) { const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s0) + ADF_PIDX(uidx + 64 * MQ)); t1[k] = make_float4(q.x, q.y, q.z, q.w); }
            }
        }
        }
        if (FUSED == FUSE_VIEW) {                                // U1 = conf, U0 = conf * float(dL)  (DF.cpp:288-290)
            const int tl = nfull - v0;                           // the partial vector, counted from this wave's first
            const int ktail = (rem && tl >= 0 && tl < 16 * M) ? (tl >> 6) : -1;   // wave-uniform: the one k that holds it
#pragma unroll
            for (int k = 0; k < M / 4; k++) {
                if (k == ktail && lane == (tl & 63)) {           // elements rem..3 lie past the row
                    const int sh = 16 * (4 - rem);
                    unsigned long long w = (unsigned long long)(unsigned short)draw[k].x | ((unsigned long long)(unsigned short)draw[k].y << 16) |
                                           ((unsigned long long)(unsigned short)draw[k].z << 32) | ((unsigned long long)(unsigned short)draw[k].w << 48);
                    w >>= sh;
                    draw[k] = make_short4((short)(w & 0xffff), (short)((w >> 16) & 0xffff), (short)((w >> 32) & 0xffff), (short)(w >> 48));
                    if (rem < 2) t1[k].y = 0.0f;
                    if (rem < 3) t1[k].z = 0.0f;
                    t1[k].w = 0.0f;
                }
                t0[k] = make_float4(t1[k].x * (float)draw[k].x, t1[k].y * (float)draw[k].y, t1[k].z * (float)draw[k].z, t1[k].w * (float)draw[k].w);
            }
        }
    }
    // columns [len, pitch) of every plane are zero by construction (the host zeroes the workspace
    // whenever the geometry changes and no kernel writes non-zeros there), and float4s past the pitch
    // were loaded as zeros: the tail of the row is identity rows without masks
#define ADF_TRANSPOSE_IN(T, DST, SCALE)                                                   \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < M / 4; k++) stage[64 * k + lane] = T[k];    \
        __syncthreads();                                                                  \
        _Pragma("unroll") for (int k = 0; k < M / 4; k++) {                               \
            const float4 v = stage[lane * (M / 4) + k];                                   \
            DST[4 * k + 0] = v.x * (SCALE); DST[4 * k + 1] = v.y * (SCALE);               \
            DST[4 * k + 2] = v.z * (SCALE); DST[4 * k + 3] = v.w * (SCALE);               \
        }                                                                                 \
        __syncthreads();                                                                  \
    }
    // PAIR, not fused: half HALF of the interleaved row (strips [2M*HALF, 2M*HALF + 2M)) holds both
    // right-hand sides of the chunks of lanes [32*HALF, 32*HALF + 32); float4 #q of a chunk starts at
    // column j = lane'*M + 4q of the half, i.e. at float4 8*(j/16) + (j%16)/4 (+4 for U1) of the stage.
#define ADF_PAIR_IN(T, HALF)                                                              \
    {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < MQ; k++) stage[64 * k + lane] = T[k];       \
        __syncthreads();                                                                  \
        if ((lane >> 5) == (HALF)) {                                                      \
            _Pragma("unroll") for (int k = 0; k < MQ; k++) {                              \
                const int j = (lane & 31) * M + 4 * k;                                    \
                const int sidx = ((j >> 4) << 3) + ((j & 15) >> 2);                       \
                const float4 v = stage[sidx], w = stage[sidx + 4];                        \
                f0[0][4 * k + 0] = v.x; f0[0][4 * k + 1] = v.y; f0[0][4 * k + 2] = v.z; f0[0][4 * k + 3] = v.w; \
                f1[0][4 * k + 0] = w.x; f1[0][4 * k + 1] = w.y; f1[0][4 * k + 2] = w.z; f1[0][4 * k + 3] = w.w; \
            }                                                                             \
        }                                                                                 \
        __syncthreads();                                                                  \
    }
    ADF_TRANSPOSE_IN(tC, c[0], a.lambda)
    if (PAIR && FUSED == FUSE_NONE) {
        ADF_PAIR_IN(t0, 0)
        ADF_PAIR_IN(t1, 1)
    } else {
        ADF_TRANSPOSE_IN(t0, f0[0], 1.0f)
        if (R > 1) ADF_TRANSPOSE_IN(t1, f1[0], 1.0f)
        else {
#pragma unroll
            for (int i = 0; i < M; i++) f1[0][i] = 0.0f;
        }
    }
#undef ADF_PAIR_IN
#undef ADF_TRANSPOSE_IN

    float a_s[1] = {__shfl_up(c[0][M - 1], 1)};
    if (lane == 0) a_s[0] = 0.0f;
    if constexpr (NW == 2) {                                     // chunk 64 follows chunk 63
        if (wv == 0 && lane == 63) xch[0] = c[0][M - 1];
        __syncthreads();
        if (wv == 1 && lane == 0) a_s[0] = xch[0];
    }

    Boundary<R> bd[1];
    chunk_boundary<M, R, 1>(c, f0, f1, a_s, bd);
    float nGS0 = __shfl_down(bd[0].GS0, 1), nGS1 = (R > 1) ? __shfl_down(bd[0].GS1, 1) : 0.0f;
    float nPS = __shfl_down(bd[0].PS, 1), nQS = __shfl_down(bd[0].QS, 1);
    if (lane == 63) { nGS0 = 0.0f; nGS1 = 0.0f; nPS = 0.0f; nQS = 0.0f; }
    if constexpr (NW == 2) {                                     // chunk 63's next chunk is wave 1's first
        if (wv == 1 && lane == 0) { xch[1] = bd[0].GS0; xch[2] = (R > 1) ? bd[0].GS1 : 0.0f; xch[3] = bd[0].PS; xch[4] = bd[0].QS; }
        __syncthreads();
        if (wv == 0 && lane == 63) { nGS0 = xch[1]; nGS1 = xch[2]; nPS = xch[3]; nQS = xch[4]; }
    }
    float al, be, ga, p0, p1, xs0[1], xs1[1];
    separator_row<M, R>(c[0], f0[0], f1[0], bd[0], nGS0, nGS1, nPS, nQS, al, be, ga, p0, p1);
    float xL0[1], xL1[1];
    if constexpr (NW == 2) {
        const int g = 64 * wv + lane;                            // chunk of this lane
        red[0][g] = al; red[1][g] = be; red[2][g] = ga; red[3][g] = p0; red[4][g] = p1;
        __syncthreads();
        if (wv == 0) reduced128<R>(lane, red[0], red[1], red[2], red[3], red[4], 1, xsol[0], xsol[1]);
        __syncthreads();
        xs0[0] = xsol[0][g]; xs1[0] = (R > 1) ? xsol[1][g] : 0.0f;
        xL0[0] = g > 0 ? xsol[0][g - 1] : 0.0f; xL1[0] = (R > 1 && g > 0) ? xsol[1][g - 1] : 0.0f;
    } else {
        pcr64<R>(lane, al, be, ga, p0, p1, xs0[0], xs1[0]);
        xL0[0] = __shfl_up(xs0[0], 1); xL1[0] = (R > 1) ? __shfl_up(xs1[0], 1) : 0.0f;
        if (lane == 0) { xL0[0] = 0.0f; xL1[0] = 0.0f; }
    }
    chunk_solve<M, R, 1>(c, f0, f1, a_s, xL0, xL1, xs0, xs1);

    typedef float v4f __attribute__((ext_vector_type(4)));
    // The pass works in place: the store addresses ARE the load addresses, and the compiler would keep
    // those (a 64-bit pointer per load) alive across the whole solve to reuse them.  An opaque copy of
    // the lane index makes it recompute them here instead.
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    if (!PAIR) {
#pragma unroll
        for (int k = 0; k < MQ; k++)
            stage[lane * MQ + k] = make_float4(f0[0][4 * k], f0[0][4 * k + 1], f0[0][4 * k + 2], f0[0][4 * k + 3]);
        __syncthreads();
        v4f* d4 = reinterpret_cast<v4f*>(a.U0 + off);
#pragma unroll
        for (int k = 0; k < MQ; k++) {
            const int idx = 64 * k + lane_s;
            if (v0 + idx < nvec) { const float4 q = stage[idx]; const v4f o = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(o, d4 + v0 + idx); }
        }
    } else {
        // the mirror image of ADF_PAIR_IN: half a row of the pair plane at a time
        v4f* d4 = reinterpret_cast<v4f*>(a.U0 + offU);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if ((lane >> 5) == half) {
#pragma unroll
                for (int k = 0; k < MQ; k++) {
                    const int j = (lane & 31) * M + 4 * k;
                    const int sidx = ((j >> 4) << 3) + ((j & 15) >> 2);
                    stage[sidx] = make_float4(f0[0][4 * k], f0[0][4 * k + 1], f0[0][4 * k + 2], f0[0][4 * k + 3]);
                    stage[sidx + 4] = make_float4(f1[0][4 * k], f1[0][4 * k + 1], f1[0][4 * k + 2], f1[0][4 * k + 3]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MQ; k++) {
                const int idx = 2 * v0 + 64 * (k + half * MQ) + lane_s;
                if (idx < nvecU) { const float4 q = stage[64 * k + lane_s]; const v4f o = {q.x, q.y, q.z, q.w}; __builtin_nontemporal_store(o, d4 + ADF_PIDX(idx)); }
            }
            __syncthreads();
        }
    }
}

template <int M, int NW = 1>
hipError_t launch_h(const WavePassArgs& a, int n_rhs, int n_pairs, hipStream_t st)
{
    dim3 grid(a.nscan, n_pairs), block(64 * NW);
    if (a.lo_conf) {
        if (n_rhs != 2 || !a.lo_taps) return hipErrorInvalidValue;
        const int n = ((a.len + 3) / 4) * 4;
        hipLaunchKernelGGL(lo_tap_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a.lo_taps, n, a.len, a.hi_x0, a.lo_scale_x, a.lo_w);
        // maps of exactly half the view's width, ROI on an even column >= 2: the form with shared source elements
        if (wave_hpass_lo_half(a))
            hipLaunchKernelGGL((wave_hpass_kernel<M, 2, FUSE_LO_HALF, NW>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((wave_hpass_kernel<M, 2, FUSE_LO, NW>), grid, block, 0, st, a);
    } else if (a.conf_in) {
        if (n_rhs != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((wave_hpass_kernel<M, 2, FUSE_VIEW, NW>), grid, block, 0, st, a);
    } else if (n_rhs == 2) hipLaunchKernelGGL((wave_hpass_kernel<M, 2, FUSE_NONE, NW>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wave_hpass_kernel<M, 1, FUSE_NONE, NW>), grid, block, 0, st, a);
    return hipGetLastError();
}

// chunk length / wavefronts per row the launcher picks for a row of `len` elements
void pick_row_bucket(int len, int& m, int& nw)
{
    if (len > 64 * 64) {
        const int q = (len + 127) / 128;
        nw = 2; m = q <= 40 ? 40 : q <= 48 ? 48 : q <= 56 ? 56 : q <= 60 ? 60 : 64;
        return;
    }
    const int q = (len + 63) / 64;
    nw = 1; m = q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 20 ? 20 : q <= 28 ? 28 : q <= 40 ? 40 : q <= 56 ? 56 : q <= 60 ? 60 : 64;
}

} // namespace

int wave_max_row_len() { return 128 * 64; }

// The fused first pass reads conf as float4s -- the confidence plane is the library's own and laid out so that the ROI
// row starts 16-byte aligned whatever the ROI is (Geom::cx0 / cpitch) -- and dL, the caller's map, in 8-byte pieces at
// any 2-byte aligned address (round 3: any ROI x / width, any even stride).  Rows shorter than one vector go through
// the prologue kernels instead.
bool wave_hpass_can_fuse(const WavePassArgs& a)
{
    if (!a.conf_in || !a.dl_in) return false;
    if (a.len < 4 || a.conf_pitch % 4 != 0 || a.conf_x0 % 4 != 0 || a.conf_frame % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.conf_in) & 15u) != 0) return false;
    if (a.dl_stride % 2 != 0 || a.dl_pair_stride % 2 != 0 || (reinterpret_cast<uintptr_t>(a.dl_in) & 1u) != 0) return false;
    return true;
}

bool wave_hpass_lo_half(const WavePassArgs& a)
{
    if (!(a.lo_conf && a.lo_half && a.lo_scale_x == 0.5 && (a.hi_x0 & 1) == 0 && a.hi_x0 >= 2)) return false;
    int m, nw;
    pick_row_bucket(a.len, m, nw);
    return !(m == 60 && nw == 1);     // (that bucket's half-width form needs five registers more than a lane has: the general form)
}

// The low-resolution form stages, per wavefront and per half of its columns, the source elements its taps touch plus
// one: that span must fit lo_row_cap(M) slots (scale factors up to about 0.66 do, whatever the bucket), the strides
// must keep rows 4-byte / 2-byte aligned.  The same tap function runs here and in the kernel.
bool wave_hpass_can_fuse_lo(const WavePassArgs& a)
{
    if (!a.lo_conf || !a.lo_dl || a.lo_w < 2 || a.lo_w > 65535 || a.lo_h < 1 || a.len < 2 || a.len > wave_max_row_len()) return false;
    if ((reinterpret_cast<uintptr_t>(a.lo_conf) & 3u) != 0 || (reinterpret_cast<uintptr_t>(a.lo_dl) & 1u) != 0) return false;
    if (a.lo_dl_stride % 2 != 0 || a.lo_dl_pair % 2 != 0) return false;
    if (!(a.lo_scale_x > 0.0 && a.lo_scale_y > 0.0) || a.hi_x0 < 0 || a.hi_y0 < 0) return false;
    int m, nw;
    pick_row_bucket(a.len, m, nw);
    const int mq = m / 4, kh = (mq + 1) / 2, cap = lo_row_cap(m);
    for (int wv = 0; wv < nw; wv++)
        for (int hh = 0; hh < 2; hh++) {
            const int cfirst = 64 * m * wv + 256 * (hh ? kh : 0);
            const int clast = std::min(64 * m * wv + 256 * (hh ? mq : kh), a.len) - 1;
            if (clast < cfirst) continue;
            int s_first, s_last; float f_;
            lin_tap(a.hi_x0 + cfirst, a.lo_scale_x, a.lo_w, s_first, f_);
            lin_tap(a.hi_x0 + clast, a.lo_scale_x, a.lo_w, s_last, f_);
            if (s_last + 2 - s_first > cap) return false;
        }
    return true;
}

hipError_t launch_wave_hpass(const WavePassArgs& a, int n_rhs, int n_pairs, hipStream_t st)
{
    if (a.len < 2 || a.len > wave_max_row_len() || a.pitch % 64 != 0 || a.pitch < a.len) return hipErrorInvalidValue;
    if (a.lo_conf && !wave_hpass_can_fuse_lo(a)) return hipErrorInvalidValue;
    if (!a.lo_conf && a.conf_in && !wave_hpass_can_fuse(a)) return hipErrorInvalidValue;
    int m, nw;
    pick_row_bucket(a.len, m, nw);
    if (nw == 2) {           // wider than 4096 columns: two wavefronts per row
        switch (m) {
        case 40: return launch_h<40, 2>(a, n_rhs, n_pairs, st);
        case 48: return launch_h<48, 2>(a, n_rhs, n_pairs, st);
        case 56: return launch_h<56, 2>(a, n_rhs, n_pairs, st);
        case 60: return launch_h<60, 2>(a, n_rhs, n_pairs, st);   // 7680 columns: a full 8K row
        default: return launch_h<64, 2>(a, n_rhs, n_pairs, st);
        }
    }
    switch (m) {
    case 4: return launch_h<4>(a, n_rhs, n_pairs, st);
    case 8: return launch_h<8>(a, n_rhs, n_pairs, st);
    case 16: return launch_h<16>(a, n_rhs, n_pairs, st);
    case 20: return launch_h<20>(a, n_rhs, n_pairs, st);
    case 28: return launch_h<28>(a, n_rhs, n_pairs, st);
    case 40: return launch_h<40>(a, n_rhs, n_pairs, st);
    case 56: return launch_h<56>(a, n_rhs, n_pairs, st);
    case 60: return launch_h<60>(a, n_rhs, n_pairs, st);       // 3840 columns: a full 4K row
    default: return launch_h<64>(a, n_rhs, n_pairs, st);
    }
}

} // namespace adf
