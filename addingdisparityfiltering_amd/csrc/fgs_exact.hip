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
// HBM traffic per element and pass with R right-hand sides: forward read 4+4R, write 4+4R
// (D and the eliminated right-hand sides do not fit on chip for a lane-per-scanline sweep),
// backward read 4+4R, write 4R  =>  12+16R bytes (44 at R=2) against the algorithmic 4+8R.
// This is the price of bit-exactness; ADF_SOLVER_WAVE removes it.
#include "adf_internal.h"

#pragma clang fp contract(off)

namespace adf {

namespace {

constexpr int PF = 8;        // steps per software-pipelined chunk (loads run one chunk ahead)
constexpr int TT = 32;       // steps per transposed output block (128-byte segments)
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
    float* pD = a.D + pb + s;
    float* pF0 = a.F0 + pb + s;
    float* pF1 = (R > 1) ? a.F1 + pb + s : nullptr;

    // ------------------------------ forward elimination ------------------------------
    {
        float cp = lam * pC[0];
        const float om = 1.0f - cp;
        float d = cp / om;
        float f0 = pU0[0] / om;
        float f1 = 0.0f;
        pD[0] = d;
        pF0[0] = f0;
        if (R > 1) { f1 = pU1[0] / om; pF1[0] = f1; }

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
                    const size_t o = (size_t)t * pitch;
                    pD[o] = d;
                    pF0[o] = f0;
                    if (R > 1) { f1 = (u1_[k] - f1 * cp) / den; pF1[o] = f1; }
                    cp = cc;
                }
            }
        }
    }

    // ------------------------------ back substitution ------------------------------
    __shared__ __align__(16) float tile[(EPI == EPI_PLANES) ? R * 64 * TPITCH : 4];
    float x0 = 0.0f, x1 = 0.0f;
    const int tb = ((len - 1) / PF) * PF;
    float nd[PF], n0[PF], n1[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const size_t o = (size_t)min(tb + k, len - 1) * pitch;
        nd[k] = pD[o]; n0[k] = pF0[o];
        if (R > 1) n1[k] = pF1[o];
    }
    for (int t0 = tb; t0 >= 0; t0 -= PF) {
        float d_[PF], f0_[PF], f1_[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) { d_[k] = nd[k]; f0_[k] = n0[k]; if (R > 1) f1_[k] = n1[k]; }
        if (t0 >= PF) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const size_t o = (size_t)(t0 - PF + k) * pitch;
                nd[k] = pD[o]; n0[k] = pF0[o];
                if (R > 1) n1[k] = pF1[o];
            }
        }
        float o0[PF], o1[PF];
#pragma unroll
        for (int k = PF - 1; k >= 0; k--) {
            const int t = t0 + k;
            if (t < len) {
                const bool last = (t == len - 1);
                x0 = last ? f0_[k] : f0_[k] - d_[k] * x0;
                if (R > 1) x1 = last ? f1_[k] : f1_[k] - d_[k] * x1;
            }
            o0[k] = x0;
            o1[k] = x1;
        }

        if (EPI == EPI_PLANES) {
            // stage 8 steps of 64 scanlines; flush a 64 x 32 block as 128-byte segments
            const int col = t0 & (TT - 1);
            float4* w0 = reinterpret_cast<float4*>(&tile[lane * TPITCH + col]);
            w0[0] = make_float4(o0[0], o0[1], o0[2], o0[3]);
            w0[1] = make_float4(o0[4], o0[5], o0[6], o0[7]);
            if (R > 1) {
                float4* w1 = reinterpret_cast<float4*>(&tile[64 * TPITCH + lane * TPITCH + col]);
                w1[0] = make_float4(o1[0], o1[1], o1[2], o1[3]);
                w1[1] = make_float4(o1[4], o1[5], o1[6], o1[7]);
            }
            if (col == 0) {
                __syncthreads();
                const int tblk = t0; // first step of the block
                const int q = lane & 7;
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int row = m * 8 + (lane >> 3);
                    const int srow = blockIdx.x * 64 + row;
                    if (srow < a.nscan) {
                        const size_t o = pb + (size_t)srow * a.pitch_out + tblk + 4 * q;
                        *reinterpret_cast<float4*>(a.O0 + o) =
                            *reinterpret_cast<const float4*>(&tile[row * TPITCH + 4 * q]);
                        if (R > 1)
                            *reinterpret_cast<float4*>(a.O1 + o) =
                                *reinterpret_cast<const float4*>(&tile[64 * TPITCH + row * TPITCH + 4 * q]);
                    }
                }
                __syncthreads();
            }
        } else {
            // fused epilogue, row-major image: step t = image row, scanline s = image column
            if (s < a.nscan) {
#pragma unroll
                for (int k = PF - 1; k >= 0; k--) {
                    const int t = t0 + k;
                    if (t < len) {
                        char* row = reinterpret_cast<char*>(a.out) + (ptrdiff_t)pz * a.out_pair_stride +
                                    (ptrdiff_t)(a.out_y0 + t) * a.out_stride;
                        const size_t e = (size_t)(a.out_x0 + s) * a.out_cn + a.out_c;
                        if (EPI == EPI_WLS_CONF) {
                            const float rcp = 1.0f / (o1[k] + ADF_EPS);     // DF.cpp:295
                            reinterpret_cast<int16_t*>(row)[e] = sat16(o0[k] * rcp); // DF.cpp:296
                        } else if (EPI == EPI_I16)
                            reinterpret_cast<int16_t*>(row)[e] = sat16(o0[k]);
                        else if (EPI == EPI_U8)
                            reinterpret_cast<uint8_t*>(row)[e] = sat8(o0[k]);
                        else
                            reinterpret_cast<float*>(row)[e] = o0[k];
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
