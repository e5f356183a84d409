// fgs_wave_h.hip -- horizontal pass of the on-chip partitioned solver (see fgs_wave_common.h).
#include "fgs_wave_common.h"

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
template <int M, int R, bool FUSED, int NW = 1>
__global__ void __launch_bounds__(64 * NW, (M > (NW == 2 ? 40 : ADF_H_TWO_WAVE_MAX) && R > 1) ? 1 : 2) wave_hpass_kernel(WavePassArgs a)
{
    static_assert(M % 4 == 0 && M >= 4, "chunk length must be a multiple of 4");
    static_assert(NW == 1 || NW == 2, "one or two wavefronts per row");
    __shared__ float4 stage_all[NW][M * 16];
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
        if (FUSED) {
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
        short4 draw[FUSED ? M / 4 : 1];                          // fused: the row of the left disparity map
        // (an explicit branch per load: "cond ? *p : zero" would make the compiler select between
        // addresses and park the zero in scratch memory)
        // (idx: float4 of the row-major row; uidx: float4 of this wave's part of the interleaved pair row, 2 * M * 64 floats)
        const int u0 = PAIR ? 2 * v0 : v0;
#pragma unroll
        for (int k = 0; k < M / 4; k++) {
            const int idx = v0 + 64 * k + lane, uidx = u0 + 64 * k + lane;
            tC[k] = make_float4(0.f, 0.f, 0.f, 0.f); t0[k] = tC[k]; t1[k] = tC[k];
            if (FUSED) draw[k] = make_short4(0, 0, 0, 0);
            if (FUSED) {
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
        if (FUSED) {                                             // U1 = conf, U0 = conf * float(dL)  (DF.cpp:288-290)
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
    if (PAIR && !FUSED) {
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
    if (a.conf_in) {
        if (n_rhs != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((wave_hpass_kernel<M, 2, true, NW>), grid, block, 0, st, a);
    } else if (n_rhs == 2) hipLaunchKernelGGL((wave_hpass_kernel<M, 2, false, NW>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wave_hpass_kernel<M, 1, false, NW>), grid, block, 0, st, a);
    return hipGetLastError();
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

hipError_t launch_wave_hpass(const WavePassArgs& a, int n_rhs, int n_pairs, hipStream_t st)
{
    if (a.len < 2 || a.len > wave_max_row_len() || a.pitch % 64 != 0 || a.pitch < a.len) return hipErrorInvalidValue;
    if (a.conf_in && !wave_hpass_can_fuse(a)) return hipErrorInvalidValue;
    if (a.len > 64 * 64) {   // wider than 4096 columns: two wavefronts per row
        const int m = (a.len + 127) / 128;
        if (m <= 40) return launch_h<40, 2>(a, n_rhs, n_pairs, st);
        if (m <= 48) return launch_h<48, 2>(a, n_rhs, n_pairs, st);
        if (m <= 56) return launch_h<56, 2>(a, n_rhs, n_pairs, st);
        if (m <= 60) return launch_h<60, 2>(a, n_rhs, n_pairs, st);   // 7680 columns: a full 8K row
        return launch_h<64, 2>(a, n_rhs, n_pairs, st);
    }
    const int m = (a.len + 63) / 64;
    if (m <= 4) return launch_h<4>(a, n_rhs, n_pairs, st);
    if (m <= 8) return launch_h<8>(a, n_rhs, n_pairs, st);
    if (m <= 16) return launch_h<16>(a, n_rhs, n_pairs, st);
    if (m <= 20) return launch_h<20>(a, n_rhs, n_pairs, st);
    if (m <= 28) return launch_h<28>(a, n_rhs, n_pairs, st);
    if (m <= 40) return launch_h<40>(a, n_rhs, n_pairs, st);
    if (m <= 56) return launch_h<56>(a, n_rhs, n_pairs, st);
    if (m <= 60) return launch_h<60>(a, n_rhs, n_pairs, st);   // 3840 columns: a full 4K row
    return launch_h<64>(a, n_rhs, n_pairs, st);
}

} // namespace adf
