// fgs_wave_common.h -- on-chip partitioned Thomas solve (ADF_SOLVER_WAVE).
//
// Same tridiagonal systems as FastGlobalSmootherFilterImpl::process_row / VerticalPass_ParBody
// (FGS.cpp:439-464, 484-584):  a_j x_{j-1} + b_j x_j + c_j x_{j+1} = f_j  with  c_j = lambda*C[j],
// a_j = c_{j-1}, b_j = 1 - a_j - c_j -- but solved so that every scanline stays on chip and HBM sees
// only the algorithmic traffic (read C and the R right-hand sides, write the R solutions: 4+8R bytes
// per pixel instead of the 12+16R of a lane-per-scanline sweep that must spill D and the eliminated
// right-hand sides).
//
// A scanline is cut into 64 chunks of M elements.  The last element of each chunk is a separator:
//   phase 1  two running sweeps over the chunk interior (left->right LU, right->left UL) give, with
//            O(1) state, how the interior's two end elements depend on the neighbouring separators:
//            x_first = GS - PS*xL - QS*xR,  x_last = GE - PE*xL - QE*xR;
//   reduce   the 64 separator equations form a tridiagonal system (alpha,beta,gamma,phi) solved by
//            parallel cyclic reduction across the 64 lanes of a wavefront (6 shuffle steps);
//   phase 2  with xL, xR known, a plain Thomas solve of the interior held entirely in registers.
// The matrix is strictly diagonally dominant (b = 1 + |a| + |c|), so every step is stable without
// pivoting.  Arithmetic is re-associated (FMA, v_rcp + one Newton step) => results differ from the
// scalar order in the last bits; the tests hold them to the reference's own reproducibility bar
// (<=1 LSB of the CV_16S output, mean <=1/256 LSB: test_disparity_wls_filter.cpp:104-105).
//
//   horizontal pass: one wavefront per image row; lane l owns columns [l*M, l*M+M); the row goes
//                    HBM -> (coalesced 16 B/lane) -> LDS -> (chunk per lane) -> registers and back.
//   vertical pass:   one 512-thread workgroup per strip of 16 columns; thread (chunk, column pair)
//                    owns M rows of 2 columns; the whole strip (16 x H x 3 floats) lives in the CU's
//                    register file; neighbour / separator exchange through LDS.
//
// Data layout (everything is solved in place, nothing is transposed between the passes):
//   Chor            row-major [rh][pw]
//   Cvert           strip-major [pw/16][rh][16]: the weights of a vertical strip are one contiguous stream
//   one right-hand side (R == 1)    row-major [rh][pw]; a strip row is then a 64-byte half line (the
//                   other half is served from L2/MALL to the neighbouring strip)
//   two right-hand sides (R == 2)   ONE pair plane [rh/2][pw/16][row parity][U0 x16 | U1 x16] of 2*plane
//                   floats per image: a strip row of the column pass is one full 128-byte line holding
//                   both right-hand sides and two consecutive rows are 256 contiguous bytes; the row pass
//                   reads an image row as 128-byte pieces at a 256-byte stride (the other row of the pair
//                   fills the gaps) and de-interleaves in its LDS staging buffer.  Measured against plain
//                   row-major pair rows: column pass -6 %, row pass +3.5 %.  U1 pointers are ignored.
#pragma once
#include "adf_internal.h"

namespace adf {
namespace wave {

__device__ __forceinline__ float rcp_nr(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

// ---------------------------------------------------------------------------------------------
// Chunk kernels shared by both passes.  c[] is already multiplied by lambda; element M-1 is the
// separator, elements 0..M-2 the interior; a_s is c of the element before the chunk (0 for chunk 0).
// ---------------------------------------------------------------------------------------------
template <int R>
struct Boundary { float GS0, GS1, PS, QS, GE0, GE1, PE, QE; };

// The sweeps below are serial recurrences; hipcc's scheduler, left alone, hoists every
// chain-independent temporary (b_i = 1 - a_i - c_i, c_i^2, negations) of a fully unrolled sweep to the
// front and keeps them alive -- several extra registers per element, i.e. spills at the chunk lengths
// a 4K scanline needs.  A scheduling barrier per element keeps the source order (temporaries die
// within their own step), and an empty asm on c[] keeps temporaries from being shared between sweeps.
#define ADF_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int M>
__device__ __forceinline__ void launder(float (&v)[M])
{
#pragma unroll
    for (int i = 0; i < M; i++) asm volatile("" : "+v"(v[i]));
}

// Reciprocal used inside the sweeps.  v_rcp_f32 is accurate to 1 ulp; one Newton step makes it
// (nearly) correctly rounded at the price of two more dependent FMAs on the serial chain.
template <bool REFINE>
__device__ __forceinline__ float rcp_sel(float x) { return REFINE ? rcp_nr(x) : __builtin_amdgcn_rcpf(x); }

// Quotient x / den from a reciprocal r of den.  x * r carries the reciprocal's rounding AND the product's;
// one residual step (two FMAs) makes the quotient correctly rounded in all but rare cases, i.e. as good as
// the IEEE division of the scalar order.  Measured against the float64 banded solve on the badly
// conditioned draws of tests/test_gpu_fuzz.py (mean |error| of the float planes, in LSB of the output):
// scalar order 0.060; wave solver 0.073 with plain products, 0.059 with the residual step in the boundary
// sweeps (bit 0; the other two phases do not matter), at +2.3 % of a BASELINE step (the fused first row
// pass +11 %).  It changes nothing in how often the two float32 evaluations round differently, which is
// what the parity tests measure, so the build default is 0; -DADF_WAVE_DIVFIX=1 buys the accuracy.
#ifndef ADF_WAVE_DIVFIX
#define ADF_WAVE_DIVFIX 0   // bit 0: boundary sweeps, bit 1: interior solve, bit 2: reduced system
#endif
template <int WHERE>
__device__ __forceinline__ float qdiv(float x, float den, float r)
{
    const float q = x * r;
    return (ADF_WAVE_DIVFIX & WHERE) ? __builtin_fmaf(__builtin_fmaf(-den, q, x), r, q) : q;
}

#ifndef ADF_WAVE_REFINE_BOUNDARY
#define ADF_WAVE_REFINE_BOUNDARY 1
#endif
#ifndef ADF_WAVE_REFINE_SOLVE
#define ADF_WAVE_REFINE_SOLVE 1
#endif

// Phase 1 for NC independent chunks (columns): boundary coefficients with O(1) state.  The
// left->right (LU) and right->left (UL) sweeps are independent serial chains; they advance together,
// one element each per step, so that every step carries 2*NC independent chains (the passes are bound
// by VALU dependency stalls, not by issue slots).
template <int M, int R, int NC>
__device__ __forceinline__ void chunk_boundary(const float (&c)[NC][M], const float (&f0)[NC][M], const float (&f1)[NC][M],
                                               const float (&a_s)[NC], Boundary<R> (&o)[NC])
{
    constexpr bool NRB = ADF_WAVE_REFINE_BOUNDARY != 0;
#ifdef ADF_WAVE_DEBUG_SKIP_COMPUTE  // timing experiment only: keep the data flow, drop the sweeps
#pragma unroll
    for (int e = 0; e < NC; e++) {
        o[e].GE0 = f0[e][0]; o[e].GE1 = f1[e][0]; o[e].PE = c[e][0]; o[e].QE = c[e][M - 1];
        o[e].GS0 = f0[e][M - 1]; o[e].GS1 = f1[e][M - 1]; o[e].PS = a_s[e]; o[e].QS = c[e][1];
    }
    return;
#endif
    // left -> right: x_i + D_i x_{i+1} = g_i - p_i xL;   right -> left: x_i + E_i x_{i-1} = h_i - q_i xR
    float D[NC], g0[NC], g1[NC], p[NC];
    float r[NC], dr[NC], h0[NC], h1[NC], q[NC];
#pragma unroll
    for (int e = 0; e < NC; e++) {
        const float a = a_s[e];
        const float dl = (1.0f - a) - c[e][0];
        const float rl = rcp_sel<NRB>(dl);
        D[e] = qdiv<1>(c[e][0], dl, rl); g0[e] = qdiv<1>(f0[e][0], dl, rl); g1[e] = (R > 1) ? qdiv<1>(f1[e][0], dl, rl) : 0.0f; p[e] = qdiv<1>(a, dl, rl);
        const float ci = c[e][M - 2];
        const float ar = (M - 2 == 0) ? a_s[e] : c[e][(M - 3 > 0) ? M - 3 : 0];
        dr[e] = (1.0f - ar) - ci;
        r[e] = rcp_sel<NRB>(dr[e]);
        h0[e] = qdiv<1>(f0[e][M - 2], dr[e], r[e]); h1[e] = (R > 1) ? qdiv<1>(f1[e][M - 2], dr[e], r[e]) : 0.0f; q[e] = qdiv<1>(ci, dr[e], r[e]);
    }
#pragma unroll
    for (int t = 1; t <= M - 2; t++) {
        const int i = t, j = M - 2 - t; // LU element, UL element
#pragma unroll
        for (int e = 0; e < NC; e++) {
            {
                const float a = c[e][i - 1];
                const float b = (1.0f - a) - c[e][i];
                const float dl = __builtin_fmaf(-a, D[e], b);
                const float rl = rcp_sel<NRB>(dl);
                D[e] = qdiv<1>(c[e][i], dl, rl);
                g0[e] = qdiv<1>(__builtin_fmaf(-a, g0[e], f0[e][i]), dl, rl);
                if (R > 1) g1[e] = qdiv<1>(__builtin_fmaf(-a, g1[e], f1[e][i]), dl, rl);
                p[e] = qdiv<1>(-a * p[e], dl, rl);
                asm volatile("" : "+v"(p[e])); // p feeds nothing until the end: keep its chain in step
            }
            {
                // opaque copies: without them the compiler shares b_j = 1 - a_j - c_j between the two
                // sweeps and keeps it alive from one sweep's visit of j to the other's
                float ci = c[e][j];
                float a = (j == 0) ? a_s[e] : c[e][(j > 0) ? j - 1 : 0];
                asm volatile("" : "+v"(ci), "+v"(a));
                const float b = (1.0f - a) - ci;
                dr[e] = __builtin_fmaf(-ci * ci, r[e], b);
                r[e] = rcp_sel<NRB>(dr[e]);
                h0[e] = qdiv<1>(__builtin_fmaf(-ci, h0[e], f0[e][j]), dr[e], r[e]);
                if (R > 1) h1[e] = qdiv<1>(__builtin_fmaf(-ci, h1[e], f1[e][j]), dr[e], r[e]);
                q[e] = qdiv<1>(-ci * q[e], dr[e], r[e]);
                asm volatile("" : "+v"(q[e]));
            }
        }
        ADF_STEP_FENCE();
    }
#pragma unroll
    for (int e = 0; e < NC; e++) {
        o[e].GE0 = g0[e]; o[e].GE1 = g1[e]; o[e].PE = p[e]; o[e].QE = D[e];
        o[e].GS0 = h0[e]; o[e].GS1 = h1[e]; o[e].PS = qdiv<1>(a_s[e], dr[e], r[e]); o[e].QS = q[e];
    }
}

// Phase 2: interior Thomas solve with both neighbours known; solutions overwrite f0 / f1.
template <int M, int R, int NC>
__device__ __forceinline__ void chunk_solve(float (&c)[NC][M], float (&f0)[NC][M], float (&f1)[NC][M],
                                            const float (&a_s)[NC], const float (&xL0)[NC], const float (&xL1)[NC],
                                            const float (&xR0)[NC], const float (&xR1)[NC])
{
    constexpr bool NRS = ADF_WAVE_REFINE_SOLVE != 0;
#ifdef ADF_WAVE_DEBUG_SKIP_COMPUTE
#pragma unroll
    for (int e = 0; e < NC; e++)
#pragma unroll
        for (int i = 0; i < M; i++) {
            f0[e][i] = __builtin_fmaf(c[e][i], xR0[e] + xL0[e], f0[e][i]);
            f1[e][i] = __builtin_fmaf(c[e][i], xR1[e] + xL1[e], f1[e][i]);
        }
    return;
#endif
#pragma unroll
    for (int e = 0; e < NC; e++) launder<M>(c[e]);
    float corig[NC], D[NC], g0[NC], g1[NC];
#pragma unroll
    for (int e = 0; e < NC; e++) {
        const float a = a_s[e];
        corig[e] = c[e][0];
        const float dn = (1.0f - a) - corig[e];
        const float r = rcp_sel<NRS>(dn);
        D[e] = qdiv<2>(corig[e], dn, r);
        g0[e] = qdiv<2>(__builtin_fmaf(-a, xL0[e], f0[e][0]), dn, r);
        g1[e] = (R > 1) ? qdiv<2>(__builtin_fmaf(-a, xL1[e], f1[e][0]), dn, r) : 0.0f;
        c[e][0] = D[e]; f0[e][0] = g0[e]; if (R > 1) f1[e][0] = g1[e];
    }
#pragma unroll
    for (int i = 1; i <= M - 2; i++) {
#pragma unroll
        for (int e = 0; e < NC; e++) {
            const float a = corig[e];
            corig[e] = c[e][i];
            const float b = (1.0f - a) - corig[e];
            const float dn = __builtin_fmaf(-a, D[e], b);
            const float r = rcp_sel<NRS>(dn);
            D[e] = qdiv<2>(corig[e], dn, r);
            g0[e] = qdiv<2>(__builtin_fmaf(-a, g0[e], f0[e][i]), dn, r);
            if (R > 1) g1[e] = qdiv<2>(__builtin_fmaf(-a, g1[e], f1[e][i]), dn, r);
            c[e][i] = D[e]; f0[e][i] = g0[e]; if (R > 1) f1[e][i] = g1[e];
        }
        ADF_STEP_FENCE();
    }
    float x0[NC], x1[NC];
#pragma unroll
    for (int e = 0; e < NC; e++) {
        x0[e] = xR0[e]; x1[e] = xR1[e];
        f0[e][M - 1] = x0[e]; if (R > 1) f1[e][M - 1] = x1[e];
    }
#pragma unroll
    for (int i = M - 2; i >= 0; i--) {
#pragma unroll
        for (int e = 0; e < NC; e++) {
            x0[e] = __builtin_fmaf(-c[e][i], x0[e], f0[e][i]);
            f0[e][i] = x0[e];
            if (R > 1) { x1[e] = __builtin_fmaf(-c[e][i], x1[e], f1[e][i]); f1[e][i] = x1[e]; }
        }
        ADF_STEP_FENCE();
    }
}

// ---------------------------------------------------------------------------------------------
// Two-column versions for the vertical pass.  A thread owns the same rows of two adjacent columns;
// written on 2-vectors the sweeps compile to the packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32,
// v_pk_add_f32: two lanes of work per issue slot), and the register pair an 8-byte load fills is the
// pair the arithmetic and the 8-byte store use -- no moves between "two scalars" and "a pair".  The
// operations and their order are exactly those of the scalar templates above (v_rcp_f32 has no packed
// form and is issued per element), so results are bit-identical to them.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f vsplat(float x) { return (v2f){x, x}; }
template <bool REFINE>
__device__ __forceinline__ v2f vrcp_sel(v2f x)
{
    const v2f r = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    return REFINE ? vfma(vfma(-x, r, vsplat(1.0f)), r, r) : r;
}

template <int WHERE>
__device__ __forceinline__ v2f vqdiv(v2f x, v2f den, v2f r)
{
    const v2f q = x * r;
    return (ADF_WAVE_DIVFIX & WHERE) ? vfma(vfma(-den, q, x), r, q) : q;
}

template <int R>
struct Boundary2 { v2f GS0, GS1, PS, QS, GE0, GE1, PE, QE; };

template <int M, int R>
__device__ __forceinline__ void chunk_boundary2(const v2f (&c)[M], const v2f (&f0)[M], const v2f (&f1)[M], v2f a_s, Boundary2<R>& o)
{
    constexpr bool NRB = ADF_WAVE_REFINE_BOUNDARY != 0;
    const v2f one = vsplat(1.0f), zero = vsplat(0.0f);
    // left -> right: x_i + D_i x_{i+1} = g_i - p_i xL;   right -> left: x_i + E_i x_{i-1} = h_i - q_i xR
    v2f D, g0, g1, p, r, dr, h0, h1, q;
    {
        const v2f dl = (one - a_s) - c[0];
        const v2f rl = vrcp_sel<NRB>(dl);
        D = vqdiv<1>(c[0], dl, rl); g0 = vqdiv<1>(f0[0], dl, rl); g1 = (R > 1) ? vqdiv<1>(f1[0], dl, rl) : zero; p = vqdiv<1>(a_s, dl, rl);
        const v2f ci = c[M - 2];
        const v2f ar = (M - 2 == 0) ? a_s : c[(M - 3 > 0) ? M - 3 : 0];
        dr = (one - ar) - ci;
        r = vrcp_sel<NRB>(dr);
        h0 = vqdiv<1>(f0[M - 2], dr, r); h1 = (R > 1) ? vqdiv<1>(f1[M - 2], dr, r) : zero; q = vqdiv<1>(ci, dr, r);
    }
#pragma unroll
    for (int t = 1; t <= M - 2; t++) {
        const int i = t, j = M - 2 - t; // LU element, UL element
        {
            const v2f a = c[i - 1];
            const v2f b = (one - a) - c[i];
            const v2f dl = vfma(-a, D, b);
            const v2f rl = vrcp_sel<NRB>(dl);
            D = vqdiv<1>(c[i], dl, rl);
            g0 = vqdiv<1>(vfma(-a, g0, f0[i]), dl, rl);
            if (R > 1) g1 = vqdiv<1>(vfma(-a, g1, f1[i]), dl, rl);
            p = vqdiv<1>(-a * p, dl, rl);
            // pin every chain to its step: pure arithmetic is not ordered against the fence below by
            // instruction selection, and a chain that drifts out of the loop drags one reciprocal per
            // step along with it (the right-hand-side chains feed nothing until the end)
            if (R > 1) asm volatile("" : "+v"(g0), "+v"(g1), "+v"(p));
            else asm volatile("" : "+v"(g0), "+v"(p));
        }
        {
            // opaque copies: without them the compiler shares b_j = 1 - a_j - c_j between the two
            // sweeps and keeps it alive from one sweep's visit of j to the other's
            v2f ci = c[j];
            v2f a = (j == 0) ? a_s : c[(j > 0) ? j - 1 : 0];
            asm volatile("" : "+v"(ci), "+v"(a));
            const v2f b = (one - a) - ci;
            dr = vfma(-ci * ci, r, b);
            r = vrcp_sel<NRB>(dr);
            h0 = vqdiv<1>(vfma(-ci, h0, f0[j]), dr, r);
            if (R > 1) h1 = vqdiv<1>(vfma(-ci, h1, f1[j]), dr, r);
            q = vqdiv<1>(-ci * q, dr, r);
            if (R > 1) asm volatile("" : "+v"(h0), "+v"(h1), "+v"(q));
            else asm volatile("" : "+v"(h0), "+v"(q));
        }
        ADF_STEP_FENCE();
    }
    o.GE0 = g0; o.GE1 = g1; o.PE = p; o.QE = D;
    o.GS0 = h0; o.GS1 = h1; o.PS = vqdiv<1>(a_s, dr, r); o.QS = q;
}

// `emit(i, x0, x1)` is called the moment row i of the chunk is final -- the separator row first, then up the chunk as the
// back-substitution forms them -- so that a caller can issue the row's store under the remaining arithmetic instead of
// after it (round 4).
template <int M, int R, typename Emit>
__device__ __forceinline__ void chunk_solve2(v2f (&c)[M], v2f (&f0)[M], v2f (&f1)[M], v2f a_s, v2f xL0, v2f xL1, v2f xR0, v2f xR1, Emit&& emit)
{
    constexpr bool NRS = ADF_WAVE_REFINE_SOLVE != 0;
    const v2f one = vsplat(1.0f), zero = vsplat(0.0f);
#pragma unroll
    for (int i = 0; i < M; i++) asm volatile("" : "+v"(c[i]));
    v2f corig = c[0], D, g0, g1;
    {
        const v2f dn = (one - a_s) - corig;
        const v2f r = vrcp_sel<NRS>(dn);
        D = vqdiv<2>(corig, dn, r);
        g0 = vqdiv<2>(vfma(-a_s, xL0, f0[0]), dn, r);
        g1 = (R > 1) ? vqdiv<2>(vfma(-a_s, xL1, f1[0]), dn, r) : zero;
        c[0] = D; f0[0] = g0; if (R > 1) f1[0] = g1;
    }
#pragma unroll
    for (int i = 1; i <= M - 2; i++) {
        const v2f a = corig;
        corig = c[i];
        const v2f b = (one - a) - corig;
        const v2f dn = vfma(-a, D, b);
        const v2f r = vrcp_sel<NRS>(dn);
        D = vqdiv<2>(corig, dn, r);
        g0 = vqdiv<2>(vfma(-a, g0, f0[i]), dn, r);
        if (R > 1) g1 = vqdiv<2>(vfma(-a, g1, f1[i]), dn, r);
        c[i] = D; f0[i] = g0; if (R > 1) f1[i] = g1;
        if (R > 1) asm volatile("" : "+v"(D), "+v"(g0), "+v"(g1));
        else asm volatile("" : "+v"(D), "+v"(g0));
        ADF_STEP_FENCE();
    }
    v2f x0 = xR0, x1 = xR1;
    f0[M - 1] = x0; if (R > 1) f1[M - 1] = x1;
    emit(M - 1, x0, x1);
#pragma unroll
    for (int i = M - 2; i >= 0; i--) {
        x0 = vfma(-c[i], x0, f0[i]);
        f0[i] = x0;
        if (R > 1) { x1 = vfma(-c[i], x1, f1[i]); f1[i] = x1; }
        if (R > 1) asm volatile("" : "+v"(x0), "+v"(x1));
        else asm volatile("" : "+v"(x0));
        emit(i, x0, x1);
        ADF_STEP_FENCE();
    }
}
template <int M, int R>
__device__ __forceinline__ void chunk_solve2(v2f (&c)[M], v2f (&f0)[M], v2f (&f1)[M], v2f a_s, v2f xL0, v2f xL1, v2f xR0, v2f xR1)
{
    chunk_solve2<M, R>(c, f0, f1, a_s, xL0, xL1, xR0, xR1, [](int, v2f, v2f) {});
}

template <int M, int R>
__device__ __forceinline__ void separator_row2(const v2f (&c)[M], const v2f (&f0)[M], const v2f (&f1)[M], const Boundary2<R>& o,
                                               v2f nGS0, v2f nGS1, v2f nPS, v2f nQS, v2f& al, v2f& be, v2f& ga, v2f& p0, v2f& p1)
{
    const v2f ae = c[M - 2], ce = c[M - 1];
    const v2f bb = (vsplat(1.0f) - ae) - ce;
    al = -ae * o.PE;
    be = vfma(-ce, nPS, vfma(-ae, o.QE, bb));
    ga = -ce * nQS;
    p0 = vfma(-ce, nGS0, vfma(-ae, o.GE0, f0[M - 1]));
    p1 = (R > 1) ? vfma(-ce, nGS1, vfma(-ae, o.GE1, f1[M - 1])) : vsplat(0.0f);
}

// Separator equation of a chunk: alpha*x_prev + beta*x + gamma*x_next = phi.
// nGS*/nPS/nQS are the NEXT chunk's left-end coefficients (zero for the last chunk).
template <int M, int R>
__device__ __forceinline__ void separator_row(const float (&c)[M], const float (&f0)[M], const float (&f1)[M],
                                              const Boundary<R>& o, float nGS0, float nGS1, float nPS, float nQS,
                                              float& al, float& be, float& ga, float& p0, float& p1)
{
    const float ae = c[M - 2], ce = c[M - 1];
    const float bb = (1.0f - ae) - ce;
    al = -ae * o.PE;
    be = __builtin_fmaf(-ce, nPS, __builtin_fmaf(-ae, o.QE, bb));
    ga = -ce * nQS;
    p0 = __builtin_fmaf(-ce, nGS0, __builtin_fmaf(-ae, o.GE0, f0[M - 1]));
    p1 = (R > 1) ? __builtin_fmaf(-ce, nGS1, __builtin_fmaf(-ae, o.GE1, f1[M - 1])) : 0.0f;
}

// Parallel cyclic reduction of a 64-row tridiagonal system held one row per lane.
template <int R>
__device__ __forceinline__ void pcr64(int lane, float al, float be, float ga, float p0, float p1, float& x0, float& x1)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float am = __shfl_up(al, d), bm = __shfl_up(be, d), gm = __shfl_up(ga, d), fm0 = __shfl_up(p0, d);
        float fm1 = (R > 1) ? __shfl_up(p1, d) : 0.0f;
        float ap = __shfl_down(al, d), bp = __shfl_down(be, d), gp = __shfl_down(ga, d), fp0 = __shfl_down(p0, d);
        float fp1 = (R > 1) ? __shfl_down(p1, d) : 0.0f;
        if (lane < d) { am = 0.0f; bm = 1.0f; gm = 0.0f; fm0 = 0.0f; fm1 = 0.0f; }
        if (lane + d > 63) { ap = 0.0f; bp = 1.0f; gp = 0.0f; fp0 = 0.0f; fp1 = 0.0f; }
        const float k1 = qdiv<4>(al, bm, rcp_nr(bm)), k2 = qdiv<4>(ga, bp, rcp_nr(bp));
        be = __builtin_fmaf(-ap, k2, __builtin_fmaf(-gm, k1, be));
        p0 = __builtin_fmaf(-fp0, k2, __builtin_fmaf(-fm0, k1, p0));
        if (R > 1) p1 = __builtin_fmaf(-fp1, k2, __builtin_fmaf(-fm1, k1, p1));
        al = -am * k1;
        ga = -gp * k2;
    }
    const float rb = rcp_nr(be);
    x0 = qdiv<4>(p0, be, rb);
    x1 = (R > 1) ? qdiv<4>(p1, be, rb) : 0.0f;
}


// A 128-row reduced system (scanlines longer than 64 chunks: two waves per row / 128 chunks per column, round 3), solved
// by ONE wavefront from LDS: lane l takes rows 2l-1, 2l, 2l+1, eliminates the odd neighbours from the even row (one
// step of cyclic reduction), the 64 even rows go through the wave-wide PCR above, and the odd rows follow by
// substitution.  Rows are read as row[k * stride]; solutions are written the same way.  The caller puts a workgroup
// barrier before (rows complete) and after (solutions visible).
template <int R>
__device__ __forceinline__ void reduced128(int lane, const float* al, const float* be, const float* ga, const float* p0, const float* p1,
                                           int stride, float* x0, float* x1)
{
    const int e = 2 * lane, o = e + 1, m = lane > 0 ? e - 1 : 0;
    const float al_e = al[e * stride], be_e = be[e * stride], ga_e = ga[e * stride], p0_e = p0[e * stride];
    const float al_o = al[o * stride], be_o = be[o * stride], ga_o = ga[o * stride], p0_o = p0[o * stride];
    const float al_m = al[m * stride], be_m = be[m * stride], ga_m = ga[m * stride], p0_m = p0[m * stride];
    const float p1_e = (R > 1) ? p1[e * stride] : 0.0f, p1_o = (R > 1) ? p1[o * stride] : 0.0f, p1_m = (R > 1) ? p1[m * stride] : 0.0f;
    const float k1 = lane > 0 ? qdiv<4>(al_e, be_m, rcp_nr(be_m)) : 0.0f;     // (row 0 has no predecessor: its alpha is 0)
    const float k2 = qdiv<4>(ga_e, be_o, rcp_nr(be_o));
    const float AL = -al_m * k1, GA = -ga_o * k2;
    const float BE = __builtin_fmaf(-al_o, k2, __builtin_fmaf(-ga_m, k1, be_e));
    const float P0 = __builtin_fmaf(-p0_o, k2, __builtin_fmaf(-p0_m, k1, p0_e));
    const float P1 = (R > 1) ? __builtin_fmaf(-p1_o, k2, __builtin_fmaf(-p1_m, k1, p1_e)) : 0.0f;
    float xe0, xe1;
    pcr64<R>(lane, AL, BE, GA, P0, P1, xe0, xe1);
    const float xn0 = __shfl_down(xe0, 1), xn1 = (R > 1) ? __shfl_down(xe1, 1) : 0.0f;   // (lane 63: row 127's gamma is 0)
    const float rb = rcp_nr(be_o);
    const float xo0 = qdiv<4>(__builtin_fmaf(-ga_o, xn0, __builtin_fmaf(-al_o, xe0, p0_o)), be_o, rb);
    x0[e * stride] = xe0; x0[o * stride] = xo0;
    if (R > 1) {
        const float xo1 = qdiv<4>(__builtin_fmaf(-ga_o, xn1, __builtin_fmaf(-al_o, xe1, p1_o)), be_o, rb);
        x1[e * stride] = xe1; x1[o * stride] = xo1;
    }
}


} // namespace wave
} // namespace adf
