// fgs_exact.hip -- lane-per-scanline Thomas solve, canonical scalar order (ADF_SOLVER_EXACT).
//
// Restates FastGlobalSmootherFilterImpl::process_row (FGS.cpp:439-464) -- and the scalar leftover of
// VerticalPass_ParBody (FGS.cpp:549-556, 581-582), which is the same recurrence down a column --
// with one GPU lane per scanline, so every scanline sees exactly the reference's operation order:
//
//   cp = l*C[0]; D[0] = cp/(1-cp); u[0] = u[0]/(1-cp)
//   cc = l*C[t]; den = (1-cp-cc) - D[t-1]*cp; D[t] = cc/den; u[t] = (u[t]-u[t-1]*cp)/den; cp = cc
//   u[t] = u[t] - D[t]*u[t+1]                                   (back substitution)
//
// Results are bit-identical to oracle/adf_oracle.c in ADF_ORDER_SCALAR (contraction off, IEEE
// division, denormals kept).  Up to two right-hand sides share one factorisation (DF.cpp:293-294
// filters conf*disp and conf with the same weights).
//
// Memory layout: the input planes have the SCANLINE index fastest (element (t, s) at t*pitch + s),
// so the 64 lanes of a wavefront read one aligned 256-byte row per step -- every load and store of
// the sweep is fully coalesced with no staging.  The solved scanlines are written in the opposite
// orientation (through a 64 x 32 LDS tile, 128-byte segments) so that the next pass, which runs
// along the other image axis, again finds its scanline index fastest.  The last pass of a filter
// call fuses the epilogue (DF.cpp:295-296 / FGS.cpp:216) and writes the image row-major.
//
// HBM traffic per element and pass with R right-hand sides.  D and the eliminated right-hand sides do
// not fit on chip for a lane-per-scanline sweep, and spilling them costs 4+4R bytes written and read
// again (12+16R in total, 44 at R=2).  Instead the forward sweep keeps only a checkpoint of its running
// state every SEG steps, and the backward sweep walks the scanline segment by segment from the end:
// it reloads the segment's inputs, repeats the forward recurrence from the checkpoint -- the same
// operations on the same operands, hence the same bits -- with D and the right-hand sides of those SEG
// steps in registers, and back-substitutes.  Forward read 4+4R, backward read 4+4R, write 4R, plus
// (4+4R)/SEG of checkpoints: 8+12R+ bytes (32.4 at R=2) against the algorithmic 4+8R.  The rest is
// the price of bit-exactness; ADF_SOLVER_WAVE removes it.
#include "adf_internal.h"

#pragma clang fp contract(off)

namespace adf {

namespace {

constexpr int PF = 8;        // steps per software-pipelined chunk of the forward sweep (loads run one chunk ahead)
constexpr int SEG = 16;      // steps per checkpointed segment (its D and right-hand sides live in registers)
constexpr int TT = 32;       // steps per transposed output block (128-byte segments)
static_assert(TT == 2 * SEG, "a transposed output block is two segments");
constexpr int TPITCH = 36;   // LDS tile row pitch in floats: 16-byte aligned, conflict-free b128

template <int R, int EPI>
__global__ void __launch_bounds__(64) exact_pass_kernel(PassArgs a)
{
    const int lane = threadIdx.x;
    const int s = blockIdx.x * 64 + lane; // scanline (always < pitch_in)
    const size_t pz = blockIdx.y;
    const size_t pb = pz * a.plane;
    const size_t pitch = (size_t)a.pitch_in;
    const int len = a.len;
    const float lam = a.lambda;

    const float* pC = a.C + pb + s;
    const float* pU0 = a.U0 + pb + s;
    const float* pU1 = (R > 1) ? a.U1 + pb + s : nullptr;
    // checkpoint k (row k of these planes, k >= 1): D, F0, F1 after step k*SEG - 1
    float* pD = a.D + pb + s;
    float* pF0 = a.F0 + pb + s;
    float* pF1 = (R > 1) ? a.F1 + pb + s : nullptr;

    // ------------------------------ forward elimination: checkpoints only ------------------------------
    {
        float cp = lam * pC[0];
        const float om = 1.0f - cp;
        float d = cp / om;
        float f0 = pU0[0] / om;
        float f1 = 0.0f;
        if (R > 1) f1 = pU1[0] / om;

        float nc[PF], n0[PF], n1[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const size_t o = (size_t)min(1 + k, len - 1) * pitch;
            nc[k] = pC[o]; n0[k] = pU0[o];
            if (R > 1) n1[k] = pU1[o];
        }
        for (int t0 = 1; t0 < len; t0 += PF) {
            float c_[PF], u0_[PF], u1_[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) { c_[k] = nc[k]; u0_[k] = n0[k]; if (R > 1) u1_[k] = n1[k]; }
#pragma unroll
            for (int k = 0; k < PF; k++) { // prefetch the next chunk (clamped at the end of the scanline)
                const size_t o = (size_t)min(t0 + PF + k, len - 1) * pitch;
                nc[k] = pC[o]; n0[k] = pU0[o];
                if (R > 1) n1[k] = pU1[o];
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int t = t0 + k;
                if (t < len) {
                    const float cc = lam * c_[k];
                    const float den = (1.0f - cp - cc) - d * cp;
                    d = cc / den;
                    f0 = (u0_[k] - f0 * cp) / den;
                    if (R > 1) f1 = (u1_[k] - f1 * cp) / den;
                    cp = cc;
                    if (((t + 1) & (SEG - 1)) == 0 && t + 1 < len) {   // state the next segment starts from
                        const size_t o = (size_t)((t + 1) / SEG) * pitch;
                        pD[o] = d;
                        pF0[o] = f0;
                        if (R > 1) pF1[o] = f1;
                    }
                }
            }
        }
    }

    // ------------------------- back substitution, one recomputed segment at a time -------------------------
    // one 64 x 32 tile, used by the right-hand sides one after the other: 9 KB per wavefront keeps four
    // wavefronts per SIMD resident (the whole row pass of a 4K batch is then a single round of waves)
    __shared__ __align__(16) float tile[(EPI == EPI_PLANES) ? 64 * TPITCH : 4];
    float x0 = 0.0f, x1 = 0.0f;
    float hi0[SEG], hi1[SEG];   // solved upper half of the current output block (EPI_PLANES)
#pragma unroll
    for (int j = 0; j < SEG; j++) { hi0[j] = 0.0f; hi1[j] = 0.0f; }
    const int nseg = (len + SEG - 1) / SEG;
    for (int k = nseg - 1; k >= 0; k--) {
        const int tb = k * SEG;
        // D / F overwrite the inputs in place as the recurrence passes them
        float c_[SEG], u0_[SEG], u1_[SEG];
#pragma unroll
        for (int j = 0; j < SEG; j++) {
            const size_t o = (size_t)min(tb + j, len - 1) * pitch;
            c_[j] = pC[o]; u0_[j] = pU0[o];
            u1_[j] = (R > 1) ? pU1[o] : 0.0f;
        }
        float cp, d, f0, f1 = 0.0f;
        if (k == 0) {
            cp = lam * c_[0];
            const float om = 1.0f - cp;
            d = cp / om;
            f0 = u0_[0] / om;
            if (R > 1) f1 = u1_[0] / om;
        } else {
            cp = lam * pC[(size_t)(tb - 1) * pitch];
            const size_t o = (size_t)k * pitch;
            d = pD[o]; f0 = pF0[o];
            if (R > 1) f1 = pF1[o];
        }
#pragma unroll
        for (int j = 0; j < SEG; j++) {
            if (tb + j < len && (k > 0 || j > 0)) {
                const float cc = lam * c_[j];
                const float den = (1.0f - cp - cc) - d * cp;
                d = cc / den;
                f0 = (u0_[j] - f0 * cp) / den;
                if (R > 1) f1 = (u1_[j] - f1 * cp) / den;
                cp = cc;
            }
            c_[j] = d; u0_[j] = f0; u1_[j] = f1;
        }
#pragma unroll
        for (int j = SEG - 1; j >= 0; j--) {
            const int t = tb + j;
            if (t < len) {
                const bool last = (t == len - 1);
                x0 = last ? u0_[j] : u0_[j] - c_[j] * x0;
                if (R > 1) x1 = last ? u1_[j] : u1_[j] - c_[j] * x1;
            }
            u0_[j] = x0;
            u1_[j] = x1;
        }

        if (EPI == EPI_PLANES) {
            // stage the segment of 64 scanlines; flush a 64 x 32 block of 128-byte segments once its
            // lowest segment is done (segments arrive in descending order)
            // A transposed output block is TT = 2 * SEG steps: the upper segment (processed first) waits in
            // registers until the lower one is solved, then each right-hand side goes through the tile.
            if ((tb & (TT - 1)) != 0) {
#pragma unroll
                for (int j = 0; j < SEG; j++) { hi0[j] = u0_[j]; hi1[j] = u1_[j]; }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float4* w = reinterpret_cast<float4*>(&tile[lane * TPITCH]);
#pragma unroll
                    for (int q = 0; q < SEG / 4; q++) {
                        w[q] = r == 0 ? make_float4(u0_[4 * q], u0_[4 * q + 1], u0_[4 * q + 2], u0_[4 * q + 3])
                                      : make_float4(u1_[4 * q], u1_[4 * q + 1], u1_[4 * q + 2], u1_[4 * q + 3]);
                        w[SEG / 4 + q] = r == 0 ? make_float4(hi0[4 * q], hi0[4 * q + 1], hi0[4 * q + 2], hi0[4 * q + 3])
                                                : make_float4(hi1[4 * q], hi1[4 * q + 1], hi1[4 * q + 2], hi1[4 * q + 3]);
                    }
                    __syncthreads();
                    float* O = r == 0 ? a.O0 : a.O1;
                    const int q = lane & 7;
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        const int row = m * 8 + (lane >> 3);
                        const int srow = blockIdx.x * 64 + row;
                        if (srow < a.nscan)
                            *reinterpret_cast<float4*>(O + pb + (size_t)srow * a.pitch_out + tb + 4 * q) =
                                *reinterpret_cast<const float4*>(&tile[row * TPITCH + 4 * q]);
                    }
                    __syncthreads();
                }
            }
        } else {
            // fused epilogue, row-major image: step t = image row, scanline s = image column
            if (s < a.nscan) {
#pragma unroll
                for (int j = SEG - 1; j >= 0; j--) {
                    const int t = tb + j;
                    if (t < len) {
                        char* row = reinterpret_cast<char*>(a.out) + (ptrdiff_t)pz * a.out_pair_stride +
                                    (ptrdiff_t)(a.out_y0 + t) * a.out_stride;
                        const size_t e = (size_t)(a.out_x0 + s) * a.out_cn + a.out_c;
                        if (EPI == EPI_WLS_CONF) {
                            const float rcp = 1.0f / (u1_[j] + ADF_EPS);     // DF.cpp:295
                            reinterpret_cast<int16_t*>(row)[e] = sat16(u0_[j] * rcp); // DF.cpp:296
                        } else if (EPI == EPI_I16)
                            reinterpret_cast<int16_t*>(row)[e] = sat16(u0_[j]);
                        else if (EPI == EPI_U8)
                            reinterpret_cast<uint8_t*>(row)[e] = sat8(u0_[j]);
                        else
                            reinterpret_cast<float*>(row)[e] = u0_[j];
                    }
                }
            }
        }
    }
}

} // namespace

hipError_t launch_exact_pass(const PassArgs& a, int n_rhs, int epilogue, int n_pairs, hipStream_t st)
{
    if (a.len < 1 || a.nscan < 1 || n_pairs < 1) return hipErrorInvalidValue;
    if (a.pitch_in % 64 != 0 || a.pitch_in < a.nscan) return hipErrorInvalidValue;
    if (epilogue == EPI_PLANES && (a.pitch_out % 64 != 0 || a.pitch_out < ((a.len + TT - 1) / TT) * TT))
        return hipErrorInvalidValue;
    dim3 grid(a.pitch_in / 64, n_pairs), block(64);
#define ADF_LAUNCH(RR, EE) hipLaunchKernelGGL((exact_pass_kernel<RR, EE>), grid, block, 0, st, a)
    if (n_rhs == 2 && epilogue == EPI_PLANES) ADF_LAUNCH(2, EPI_PLANES);
    else if (n_rhs == 2 && epilogue == EPI_WLS_CONF) ADF_LAUNCH(2, EPI_WLS_CONF);
    else if (n_rhs == 1 && epilogue == EPI_PLANES) ADF_LAUNCH(1, EPI_PLANES);
    else if (n_rhs == 1 && epilogue == EPI_I16) ADF_LAUNCH(1, EPI_I16);
    else if (n_rhs == 1 && epilogue == EPI_F32) ADF_LAUNCH(1, EPI_F32);
    else if (n_rhs == 1 && epilogue == EPI_U8) ADF_LAUNCH(1, EPI_U8);
    else return hipErrorInvalidValue;
#undef ADF_LAUNCH
    return hipGetLastError();
}

} // namespace adf
