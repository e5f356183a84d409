// fgs_wave_v.hip -- vertical pass of the on-chip partitioned solver (see fgs_wave_common.h).
// Built with -fno-slp-vectorize: the SLP vectorizer packs the per-element temporaries of the unrolled
// sweeps into v_pk_* bundles hoisted to the front of the sweep, which costs ~50 registers at the
// chunk length a 2160-row column needs and turns into scratch spills.
#include "fgs_wave_common.h"

#ifdef ADF_V_PHASE_TIMING
// Measurement builds only (build.build_variant("vphase", ["ADF_V_PHASE_TIMING"]), tools/vphase.py): per-workgroup
// phase time stamps of the plain column pass on the 100 MHz clock, plus the CU the workgroup ran on.
__device__ unsigned long long adf_vphase[1 << 20];  // [wg][0..6 stamps, 7 = XCC_ID << 32 | HW_ID]
__device__ unsigned long long adf_vwave[1 << 21];   // [wg][wave][start, end]
extern "C" int adf_debug_read_vphase(unsigned long long* dst, int n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(adf_vphase), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int adf_debug_read_vwave(unsigned long long* dst, int n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(adf_vwave), sizeof(unsigned long long) * (size_t)n);
}
#ifndef ADF_V_PHASE_EPI
#define ADF_V_PHASE_EPI EPI_PLANES   // which pass is stamped (1 = the last pass with the fused epilogue)
#endif
#define ADF_WG_ID ((size_t)(((size_t)blockIdx.y * gridDim.x + blockIdx.x) % (1u << 17)))
#define ADF_STAMP(k) do { if (EPI == ADF_V_PHASE_EPI && threadIdx.x == 0) adf_vphase[ADF_WG_ID * 8 + (k)] = wall_clock64(); } while (0)
#define ADF_WSTAMP(k) do { if (EPI == ADF_V_PHASE_EPI && (threadIdx.x & 63) == 0) adf_vwave[(ADF_WG_ID * 8 + (threadIdx.x >> 6)) * 2 + (k)] = wall_clock64(); } while (0)
#define ADF_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ADF_STAMP(k) do { } while (0)
#define ADF_WSTAMP(k) do { } while (0)
#define ADF_DRAIN() do { } while (0)
#endif

namespace adf {

namespace {
using namespace wave;

// ---------------------------------------------------------------------------------------------
// Vertical pass: one 512-thread workgroup per 16-column strip, in place (or fused epilogue).
// thread = (chunk cidx in [0,64), column pair xp in [0,8)); M rows x 2 columns per thread.
// ---------------------------------------------------------------------------------------------
constexpr int VT = 512;  // threads per strip
constexpr int VC = 16;   // columns per strip (the layouts' strip: fgs_wave_common.h)
// Round 3: columns longer than 64 chunks of 34 rows (ROIs taller than 2176 rows: 8K frames) run HALF strips of 128 chunks
// -- VCW = 8 columns, NCH = 128, still 512 threads and the same rows and registers per thread, i.e. the same 415 KB of a
// CU's register file per workgroup -- with a 128-row reduced system per column (fgs_wave_common.h, reduced128).  Half
// strips read 32-byte pieces (a quarter less bandwidth on a full chip, tools/micro/vpattern.hip), still well ahead of
// the exact solver such ROIs fell back to.  Everything below is written for VCW columns x NCH chunks.

// ---------------------------------------------------------------------------------------------
// Whole-line loads by LDS-DMA (round 3; NOT the default: -DADF_V_GLDS=1 selects it).  Moving the pass's bytes alone
// (tools/micro/vpattern.hip) this shape is 15 % faster than the register loads below, but in the pass every wave meets
// the others at the barrier before the reduced system, so the strip is not done before its slowest wave's loads are:
// A/B on the 64 x 4K step 4.25-4.29 ms (DMA) against 4.19-4.21 ms (register loads) for the two plain passes, and
// the short columns of 1242 x 375 frames lose 8 % to the ring's LDS (profiles/r03_ab_glds.txt, EXPERIMENTS.md section 9).
// A thread owns (chunk, column pair), so loading straight into its registers
// makes every wave instruction fetch 8 rows x 64 bytes at 8 bytes per lane ("fragment-shaped"): measured on the pass's
// own layout (tools/micro/vpattern.hip, profiles/r03_vpattern.txt) that moves the pass's bytes at 4.5 TB/s, while
// global_load_lds_dwordx4 instructions that fetch 8 WHOLE 128-byte lines each (16 bytes per lane, half as many
// instructions, no register destination) into a small per-wave LDS ring, followed by 8-byte LDS reads into the same
// registers, move them at 5.3 TB/s.  Items of a wave's load sequence, for its 8 chunks at once:
//   line item i    row r0+i of the pair plane: [U0 x16 | U1 x16] = one line per chunk, lane (chunk j, piece p)
//   weight item k  rows r0+2k, r0+2k+1 of the strip-major weights: 2 x 64 bytes per chunk, lane (j, row parity, piece)
// in the order L0 L1 W0 L2 L3 W1 ..., 3M/2 items, VRING slots of 1 KiB in flight per wave.  The DMA is issued by
// inline asm (M0 = LDS destination), so its waits are counted here by hand: item K is complete once at most
// (items issued after K) operations are outstanding -- anything else the compiler has in flight is older or younger
// than all of them and only makes the wait stronger.
// ---------------------------------------------------------------------------------------------
#ifndef ADF_V_STORE_IN_SOLVE
#define ADF_V_STORE_IN_SOLVE 0   // bit 0: plane passes, bit 1: the last pass -- a row's store is issued inside the back-substitution (round-4 experiment)
#endif
#ifndef ADF_V_GLDS
#define ADF_V_GLDS 0   // 1: LDS-DMA whole-line loads (round-3 experiment, kept for A/B: the pass gains nothing, see below)
#endif
// (short columns leave room for two workgroups per CU: their rings stay small enough not to take that away)
template <int M> struct VRing { static constexpr int ITEMS = 3 * M / 2, WANT = M >= 18 ? 12 : 4, SLOTS = ITEMS < WANT ? ITEMS : WANT; };

template <int N> __device__ __forceinline__ void v_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void v_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void v_glds16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct VLoadCtx {
    const char* bC; const char* b0;
    unsigned lo, line_safe;        // byte offset of the NEXT line item's line / of row 0 of this strip (piece offset included)
    unsigned wo, w_safe;           // byte offset of the NEXT weight item's row / of row 0 (row parity and piece included)
    unsigned tile_b, r0, h, q;     // q = row parity this lane fetches in a weight item
    unsigned ring;                 // LDS byte address of the wave's ring (SGPR)
    unsigned rd;                   // this thread's read position inside a slot, as an offset into the dynamic LDS array
};

// items are issued in order, so the source offsets simply walk down the rows (rows live in tiles of TR rows:
// consecutive rows are 128 bytes apart inside a tile and a tile apart, minus the rows already walked, at its end)
template <int M, int K>
__device__ __forceinline__ void v_issue(VLoadCtx& x)
{
    constexpr unsigned TR = ADF_TILE_ROWS;
    if constexpr (K < VRing<M>::ITEMS) {
        const unsigned slot = x.ring + (unsigned)(K % VRing<M>::SLOTS) * 1024u;
        if constexpr (K % 3 == 2) {
            const unsigned row = x.r0 + 2u * (K / 3) + x.q;
            v_glds16(x.bC + (row < x.h ? x.wo : x.w_safe), slot);
            x.wo += 128u;
        } else {
            constexpr unsigned i = K - K / 3;
            v_glds16(x.b0 + (x.r0 + i < x.h ? x.lo : x.line_safe), slot);
            x.lo += (((x.r0 + i + 1u) & (TR - 1u)) == 0u) ? x.tile_b - (TR - 1u) * 128u : 128u;
        }
    }
}

template <int M, int K>
__device__ __forceinline__ void v_load_items(VLoadCtx& x, const char* lds, v2f (&c)[M], v2f (&f0)[M], v2f (&f1)[M])
{
    if constexpr (K < VRing<M>::ITEMS) {
        constexpr int S = VRing<M>::SLOTS, T = VRing<M>::ITEMS;
        v_wait_vm<(K + S - 1 < T ? S - 1 : T - 1 - K)>();
        const char* s = lds + x.rd + (K % S) * 1024;      // (an index into the __shared__ array: ds_read with an immediate offset)
        if constexpr (K % 3 == 2) {
            c[2 * (K / 3)] = *reinterpret_cast<const v2f*>(s);
            c[2 * (K / 3) + 1] = *reinterpret_cast<const v2f*>(s + 64);
        } else {
            f0[K - K / 3] = *reinterpret_cast<const v2f*>(s);
            f1[K - K / 3] = *reinterpret_cast<const v2f*>(s + 64);
        }
        if constexpr (K + S < T) {
            v_wait_lds();                                   // the slot has been read: it may be filled again
            v_issue<M, K + S>(x);
        }
        v_load_items<M, K + 1>(x, lds, c, f0, f1);
    }
}

template <int M, int K>
__device__ __forceinline__ void v_prime(VLoadCtx& x)
{
    if constexpr (K < VRing<M>::SLOTS) { v_issue<M, K>(x); v_prime<M, K + 1>(x); }
}

// saturate_cast<short> of both columns of a thread, packed (low half = first column).  cvRound semantics as sat16() in
// adf_internal.h: round half to even; NaN and anything outside the int range become INT_MIN and hence -32768; the clamp
// to [-32768, 32767] is v_cvt_pk_i16_i32's.  EPI_WLS_CONF forms u0 * (1 / (u1 + EPS)) first (DF.cpp:295-296) with
// v_rcp_f32 + one Newton step instead of the IEEE division's ten instructions: this solver is held to the reference's
// 1-LSB bar, not to bit identity.  Two things the division did must be kept:
//  * far from every confident pixel the filtered confidence u1 (and u0 with it) decays into the DENORMAL range while
//    the ratio stays an ordinary disparity (config 2: thousands of such pixels), and v_rcp_f32 flushes denormal
//    operands: the denominator is scaled by 2^64 before the reciprocal and the reciprocal by 2^64 after it --
//    unconditionally (u1 never exceeds a few thousand, so neither product leaves the normal range at the top);
//  * u1 == 0 exactly: 1 / 1e-43 overflows to +inf -- here 2^64 * rcp(2^64 * 1e-43) does -- and 0 * inf, x * inf and NaN
//    all leave the int range -> -32768, as in the reference (tests: zero-confidence edge case).
// (u1 * 2^64 + EPS * 2^64 in one FMA: it differs from (u1 + EPS) * 2^64 only below the last bit of a denormal sum.)
template <int EPI>
__device__ __forceinline__ unsigned epi_pack16(v2f u0, v2f u1)
{
    v2f x = u0;
    if (EPI == EPI_WLS_CONF) {
        const v2f sc = vsplat(0x1p64f);
        x = u0 * (vrcp_sel<true>(vfma(u1, sc, vsplat(ADF_EPS * 0x1p64f))) * sc);
    }
    const bool o0 = !(__builtin_fabsf(x.x) < 2147483648.0f), o1 = !(__builtin_fabsf(x.y) < 2147483648.0f);
    const int i0 = (int)(o0 ? -32768.0f : __builtin_rintf(x.x)), i1 = (int)(o1 ? -32768.0f : __builtin_rintf(x.y));
    typedef short s2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, (s2)__builtin_amdgcn_cvt_pk_i16(i0, i1));
}

// FS (round-4 experiment, ADF_V_STORE_IN_SOLVE): the rows are stored from inside the back-substitution; for the last pass
// the launcher picks it when the packed 4-byte output stores apply (conditions uniform over the launch, checked on the host)
template <int M, int R, int EPI, int VCW = VC, int NCH = 64, bool FS = false>
__global__ void __launch_bounds__(VT) wave_vpass_kernel(WavePassArgs a)
{
    static_assert((VCW == 16 && NCH == 64) || (VCW == 8 && NCH == 128), "whole strips of 64 chunks or half strips of 128");
    static_assert((VCW / 2) * NCH == VT, "one thread per (chunk, column pair)");
    constexpr int XPN = VCW / 2;         // column pairs per workgroup
    __shared__ float nb[4][NCH][VCW];   // next-chunk exchange: GS0, GS1, PS, QS
    __shared__ float red[5][VCW][NCH];  // separator rows by (column, chunk)
    __shared__ float xs[2][VCW][NCH];   // separator solutions
    extern __shared__ __align__(16) char vring[];   // R == 2: 8 waves x VRing<M>::SLOTS KiB (LDS-DMA landing zone)
    const int tid = threadIdx.x;
    const int xp = tid & (XPN - 1), cidx = tid / XPN;
#ifdef ADF_V_STAGGER
    // Experiment (tools/vstagger.sh): every workgroup of a pass runs the same program for the same time, so the CUs
    // stay in step -- all loading, then all computing.  Delay the first round's workgroups by a fraction of the
    // period so that later rounds inherit the offset.
    {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < 256u) {
            const unsigned g = (lin / 8u) % ADF_V_STAGGER;           // (blocks b, b+8, .. share an XCD)
            for (unsigned k = 0; k < g * ADF_V_STAGGER_UNIT; k++) __builtin_amdgcn_s_sleep(127);
        }
    }
#endif
    ADF_STAMP(0); ADF_WSTAMP(0);
#ifdef ADF_V_PHASE_TIMING
    if (EPI == ADF_V_PHASE_EPI && threadIdx.x == 0)   // hwreg(HW_REG_XCC_ID) and hwreg(HW_REG_HW_ID), 32 bits each
        adf_vphase[ADF_WG_ID * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#endif
    // A strip row is a 64-byte half of a 128-byte line of a single right-hand-side plane (R == 1) and a
    // 32-byte quarter of a line of the int16 output; the rest of the line belongs to the neighbouring
    // strips (the pair plane of R == 2 and the weights are laid out so that this does not happen).  Blocks b,
    // b+8, b+16, b+24 are dealt to the same XCD back to back (speed only, never correctness), so four
    // consecutive strips are mapped to them: later requests for a line hit (or merge in) that XCD's L2
    // instead of going to the fabric again, and partial-line writes combine there before eviction.
    int strip = VCW == VC ? (int)blockIdx.x : (int)(blockIdx.x >> 1);   // the layouts' 16-column strip this workgroup works in
    if (VCW == VC) {
        const int nfull = (int)(gridDim.x / 32) * 32;
        // (round 3: only the passes that WRITE partial lines are remapped -- the plain pass of two right-hand sides reads
        // and writes whole lines, and with consecutive blocks on consecutive strips it runs 4 % faster)
        if ((EPI != EPI_PLANES || R == 1) && (int)blockIdx.x < nfull) {
            const int grp = blockIdx.x >> 5, w = blockIdx.x & 31;
            strip = (grp << 5) + ((w & 7) << 2) + (w >> 3);
        }
    }
    const unsigned c16 = 2u * (unsigned)xp + (VCW == VC ? 0u : 8u * (blockIdx.x & 1u));   // the thread's first column inside the strip
    const int col = strip * VC + (int)c16;           // < pitch by construction of the grid
    const size_t pb = (size_t)blockIdx.y * a.plane;
    const int r0 = cidx * M;
    const int h = a.len;                            // scanline length = ROI height

    // Addressing: wave-uniform plane bases (SGPRs) + 32-bit byte offsets per thread that walk down the
    // rows.  Keeping a 64-bit address per row alive from the loads to the stores would cost more
    // registers than the strip itself.
    //   C (Cvert)  strip-major [strip][row][16]: the strip's weights are one contiguous stream
    //   R == 2     pair plane [row pair][strip][row parity][U0 x16 | U1 x16]: a strip row is one full 128-byte
    //              line, two consecutive rows 256 contiguous bytes
    //   R == 1     plain row-major plane: a strip row is a 64-byte half line
    // (see fgs_wave_common.h; measured -17 % on the pass against three row-major planes)
    const char* bC = reinterpret_cast<const char*>(a.C + pb);
    char* b0 = reinterpret_cast<char*>(a.U0 + (R > 1 ? 2 * pb : pb));
    char* b1 = (R > 1) ? b0 + 4 * VC : nullptr;
    const unsigned pitch_b = (unsigned)a.pitch * (R > 1 ? 8u : 4u);
    // pair plane: [row tile][strip][row in tile][32 floats]; consecutive rows are 128 bytes apart inside a tile and a
    // tile apart (minus the rows already walked) at a tile boundary -- which rows those are depends on the chunk's start
    constexpr unsigned TR = ADF_TILE_ROWS;
    const unsigned tile_b = 2u * TR * (unsigned)a.pitch * 4u;            // bytes from a tile to the next one
#define ADF_VSTEP(i) ((R > 1) ? (((((unsigned)r0 + (unsigned)(i) + 1u) & (TR - 1u)) == 0u) ? tile_b - (TR - 1u) * 128u : 128u) : pitch_b)
    const unsigned voff0 = (R > 1) ? (((unsigned)r0 / TR) * (2u * TR * (unsigned)a.pitch) + (unsigned)strip * (32u * TR) + ((unsigned)r0 % TR) * 32u + c16) * 4u
                                   : ((unsigned)r0 * (unsigned)a.pitch + (unsigned)col) * 4u;
    const unsigned pitch_c = 4u * VC;
    const unsigned coff0 = (((unsigned)strip * (unsigned)h + (unsigned)r0) * VC + c16) * 4u;

    // both columns of a row in one register pair, from the 8-byte load to the 8-byte store (see the
    // two-column templates in fgs_wave_common.h)
    v2f c[M], f0[M], f1[M];
    // row 0 of the column: always inside the planes
    const unsigned safe = (R > 1) ? ((unsigned)strip * (32u * TR) + c16) * 4u : (unsigned)col * 4u;
    const unsigned csafe = ((unsigned)strip * (unsigned)h * VC + c16) * 4u;
    v2f a_s = vsplat(0.0f);
    if (cidx > 0 && r0 - 1 < h) a_s = *reinterpret_cast<const v2f*>(bC + (coff0 - pitch_c)) * vsplat(a.lambda);
    if constexpr (R > 1 && ADF_V_GLDS && VCW == VC) {
        // Rows past the end of the column are fetched from row 0 of the same strip (always inside the planes) and NOT
        // masked: Cvert is 0 in the last row (FGS.cpp:658-660), so whatever finite, diagonally dominant system those
        // rows form is decoupled from the real one by exact zeros (0 * finite), and the stores below skip them.
        VLoadCtx x;
        x.bC = bC; x.b0 = b0;
        const unsigned piece = 16u * (unsigned)xp;
        x.line_safe = (unsigned)strip * (32u * TR) * 4u + piece;
        // (an opaque copy of the chunk's first row: the row tests of the load phase must not be shared with the store
        // phase's, or one register per row stays alive across the whole solve)
        unsigned r0l = (unsigned)r0;
        asm volatile("" : "+v"(r0l));
        x.lo = ((r0l / TR) * (2u * TR * (unsigned)a.pitch) + (unsigned)strip * (32u * TR) + (r0l % TR) * 32u) * 4u + piece;
        x.q = (unsigned)xp >> 2;
        x.w_safe = ((unsigned)strip * (unsigned)h) * (VC * 4u) + 16u * ((unsigned)xp & 3u);
        x.wo = ((unsigned)strip * (unsigned)h + r0l + x.q) * (VC * 4u) + 16u * ((unsigned)xp & 3u);
        x.tile_b = tile_b; x.r0 = r0l; x.h = (unsigned)h;
        char* ringp = vring + (tid >> 6) * (VRing<M>::SLOTS * 1024);
        x.ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ringp);
        x.rd = (unsigned)((tid >> 6) * (VRing<M>::SLOTS * 1024) + ((tid >> 3) & 7) * 128 + xp * 8);
        v_prime<M, 0>(x);
        v_load_items<M, 0>(x, vring, c, f0, f1);
        const v2f lam = vsplat(a.lambda);
#pragma unroll
        for (int i = 0; i < M; i++) c[i] *= lam;
    } else {
        // Rows past the end of the column are loaded from row 0 of the same column (always inside the
        // planes, no load under a divergent branch) and NOT masked: Cvert is 0 in the last row
        // (FGS.cpp:658-660), so whatever finite, diagonally dominant system those rows form is decoupled
        // from the real one by exact zeros (0 * finite), and the stores below skip them.  Untouched
        // loaded pairs stay where the load put them -- a select here would copy every pair.
        unsigned voff = voff0, coff = coff0;
        const v2f lam = vsplat(a.lambda);
#pragma unroll
        for (int i = 0; i < M; i++) {
            const bool ok = r0 + i < h;
            const unsigned vo = ok ? voff : safe;
            c[i] = *reinterpret_cast<const v2f*>(bC + (ok ? coff : csafe));
            f0[i] = *reinterpret_cast<const v2f*>(b0 + vo);
            f1[i] = (R > 1) ? *reinterpret_cast<const v2f*>(b1 + vo) : vsplat(0.0f);
            voff += ADF_VSTEP(i); coff += pitch_c;
            ADF_STEP_FENCE();   // one row's addresses at a time: hoisting all of them costs 2 registers per row
        }
#pragma unroll
        for (int i = 0; i < M; i++) c[i] *= lam;
    }

    ADF_DRAIN(); ADF_STAMP(1);
    Boundary2<R> bd;
    chunk_boundary2<M, R>(c, f0, f1, a_s, bd);
    ADF_STAMP(2);
    *reinterpret_cast<v2f*>(&nb[0][cidx][2 * xp]) = bd.GS0;
    *reinterpret_cast<v2f*>(&nb[1][cidx][2 * xp]) = bd.GS1;
    *reinterpret_cast<v2f*>(&nb[2][cidx][2 * xp]) = bd.PS;
    *reinterpret_cast<v2f*>(&nb[3][cidx][2 * xp]) = bd.QS;
    __syncthreads();
    {
        v2f nGS0 = vsplat(0.f), nGS1 = nGS0, nPS = nGS0, nQS = nGS0;
        if (cidx < NCH - 1) {
            nGS0 = *reinterpret_cast<const v2f*>(&nb[0][cidx + 1][2 * xp]); nGS1 = *reinterpret_cast<const v2f*>(&nb[1][cidx + 1][2 * xp]);
            nPS = *reinterpret_cast<const v2f*>(&nb[2][cidx + 1][2 * xp]); nQS = *reinterpret_cast<const v2f*>(&nb[3][cidx + 1][2 * xp]);
        }
        v2f al, be, ga, p0, p1;
        separator_row2<M, R>(c, f0, f1, bd, nGS0, nGS1, nPS, nQS, al, be, ga, p0, p1);
        red[0][2 * xp][cidx] = al.x; red[1][2 * xp][cidx] = be.x; red[2][2 * xp][cidx] = ga.x;
        red[3][2 * xp][cidx] = p0.x; red[4][2 * xp][cidx] = p1.x;
        red[0][2 * xp + 1][cidx] = al.y; red[1][2 * xp + 1][cidx] = be.y; red[2][2 * xp + 1][cidx] = ga.y;
        red[3][2 * xp + 1][cidx] = p0.y; red[4][2 * xp + 1][cidx] = p1.y;
    }
    __syncthreads();
    if constexpr (NCH == 64) {   // 8 wavefronts x 2 columns each: one separator row per lane
        const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int cc = 2 * wv + e;
            float x0, x1;
            pcr64<R>(lane, red[0][cc][lane], red[1][cc][lane], red[2][cc][lane], red[3][cc][lane], red[4][cc][lane], x0, x1);
            xs[0][cc][lane] = x0; xs[1][cc][lane] = x1;
        }
    } else {                     // 8 wavefronts, one column each: a 128-row system, two rows per lane
        const int cc = tid >> 6, lane = tid & 63;
        reduced128<R>(lane, red[0][cc], red[1][cc], red[2][cc], red[3][cc], red[4][cc], 1, xs[0][cc], xs[1][cc]);
    }
    __syncthreads();
    {
        const int cc = 2 * xp;
        const v2f xR0 = {xs[0][cc][cidx], xs[0][cc + 1][cidx]}, xR1 = {xs[1][cc][cidx], xs[1][cc + 1][cidx]};
        v2f xL0 = vsplat(0.0f), xL1 = xL0;
        if (cidx > 0) {
            xL0 = (v2f){xs[0][cc][cidx - 1], xs[0][cc + 1][cidx - 1]};
            xL1 = (v2f){xs[1][cc][cidx - 1], xs[1][cc + 1][cidx - 1]};
        }
        ADF_STAMP(3);
        // Round 4: a row is final the moment the back-substitution forms it; its store is issued
        // there, under the remaining arithmetic, instead of in a loop of its own behind the solve.  The offsets walk
        // UP the rows from the chunk's last one.
        if constexpr (FS && EPI == EPI_PLANES) {
            unsigned r0s = (unsigned)r0;
            asm volatile("" : "+v"(r0s));        // (recomputed here: no load address stays alive across the sweeps)
            const unsigned rl = r0s + (unsigned)(M - 1);
            unsigned vo = (R > 1) ? ((rl / TR) * (2u * TR * (unsigned)a.pitch) + (unsigned)strip * (32u * TR) + (rl % TR) * 32u + c16) * 4u
                                  : (rl * (unsigned)a.pitch + (unsigned)col) * 4u;
            const int hv = h - (int)r0s;         // rows of this chunk inside the column
            chunk_solve2<M, R>(c, f0, f1, a_s, xL0, xL1, xR0, xR1, [&](int i, v2f x0, v2f x1) {
                if (i < hv) {
                    *reinterpret_cast<v2f*>(b0 + vo) = x0;
                    if (R > 1) *reinterpret_cast<v2f*>(b1 + vo) = x1;
                }
                if (i > 0) vo -= ADF_VSTEP(i - 1);
            });
            ADF_STAMP(4); ADF_STAMP(5); ADF_DRAIN(); ADF_STAMP(6); ADF_WSTAMP(1);
            return;
        } else if constexpr (FS) {
            static_assert(!FS || EPI == EPI_PLANES || (R > 1 && EPI == EPI_WLS_CONF), "fused stores: plane passes and the disparity filter's last pass");
            int r0e = r0, cole = col;
            asm volatile("" : "+v"(r0e), "+v"(cole));
            // (a wave-uniform base in scalar registers + one 32-bit offset per thread that walks up the rows)
            char* obase = reinterpret_cast<char*>(a.out) + (ptrdiff_t)blockIdx.y * a.out_pair_stride + (ptrdiff_t)a.out_y0 * a.out_stride +
                          (ptrdiff_t)a.out_x0 * 2;
            const unsigned os = (unsigned)a.out_stride;
            unsigned oo = (unsigned)(r0e + (M - 1)) * os + (unsigned)cole * 2u;
            const int hv = (cole < a.nscan ? h : 0) - r0e;
            chunk_solve2<M, R>(c, f0, f1, a_s, xL0, xL1, xR0, xR1, [&](int i, v2f x0, v2f x1) {
                const unsigned v = epi_pack16<EPI>(x0, x1);
                if (i < hv) *reinterpret_cast<unsigned*>(obase + oo) = v;
                oo -= os;
            });
            ADF_STAMP(4); ADF_STAMP(5); ADF_DRAIN(); ADF_STAMP(6); ADF_WSTAMP(1);
            return;
        }
        chunk_solve2<M, R>(c, f0, f1, a_s, xL0, xL1, xR0, xR1);
        ADF_STAMP(4);
    }

    unsigned voff = voff0;
    asm volatile("" : "+v"(voff)); // recompute the row offsets instead of keeping the load addresses alive
    if (EPI == EPI_PLANES) {
#pragma unroll
        for (int i = 0; i < M; i++) {
            if (r0 + i < h) {
                *reinterpret_cast<v2f*>(b0 + voff) = f0[i];
                if (R > 1) *reinterpret_cast<v2f*>(b1 + voff) = f1[i];
            }
            voff += ADF_VSTEP(i);
        }
    } else {
        // (opaque copies made AFTER the solve: nothing of the epilogue's addressing may be formed while the strip
        // and the sweeps' temporaries fill the register file)
        int r0e = r0, cole = col;
        asm volatile("" : "+v"(r0e), "+v"(cole));
        char* ob = reinterpret_cast<char*>(a.out) + (ptrdiff_t)blockIdx.y * a.out_pair_stride +
                   (ptrdiff_t)(a.out_y0 + r0e) * a.out_stride;
        const int esz = (EPI == EPI_F32) ? 4 : (EPI == EPI_U8) ? 1 : 2;
        unsigned ooff = (unsigned)((a.out_x0 + cole) * a.out_cn + a.out_c) * (unsigned)esz;
        // Single-channel int16 output with an even number of columns (every call of the disparity filter on an even-width
        // ROI): both columns of the thread go out as one packed dword -- or, when the ROI starts on an odd column (the
        // StereoBM factory's ROIs do: DF.cpp:401), as its two halves -- with no per-row alignment test and no branch but
        // the row mask.  Round 3: the general loop below spent 4.5 us per strip (of 38) on IEEE divisions, conversions
        // and exec-mask branches (profiles/r03_vphase.txt).  All three conditions are uniform over the launch.
        // Round 4: an ODD number of columns takes the same path -- the one thread per strip row that holds the ROI's last
        // column stores its low half only (2 bytes), everything else is unchanged.
        const bool fast16 = (EPI == EPI_WLS_CONF || EPI == EPI_I16) && a.out_cn == 1;
        if (fast16 && (a.nscan & 1)) {
            const bool al4 = ((reinterpret_cast<uintptr_t>(a.out) | (uintptr_t)a.out_stride | (uintptr_t)a.out_pair_stride) & 3u) == 0 &&
                             ((a.out_x0 * 2) & 3) == 0;
            const int hv2 = (cole + 1 < a.nscan ? h : 0) - r0e;   // rows of a thread with two columns inside the ROI
            const int hv1 = (cole + 1 == a.nscan ? h : 0) - r0e;  // ... with only its first column inside
            char* dst = ob + ooff;
#pragma unroll
            for (int i = 0; i < M; i++) {
                const unsigned v = epi_pack16<EPI>(f0[i], f1[i]);
                if (i < hv2) {
                    if (al4) *reinterpret_cast<unsigned*>(dst) = v;
                    else { reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v; reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16); }
                }
                if (i < hv1) reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;
                dst += a.out_stride;
                ADF_STEP_FENCE();
            }
        } else if (fast16) {
            const bool al4 = ((reinterpret_cast<uintptr_t>(a.out) | (uintptr_t)a.out_stride | (uintptr_t)a.out_pair_stride) & 3u) == 0 &&
                             ((a.out_x0 * 2) & 3) == 0;
            const int hv = (cole < a.nscan ? h : 0) - r0e;        // rows of this thread to store (threads on pitch padding: none)
            char* dst = ob + ooff;
            if (al4) {
#pragma unroll
                for (int i = 0; i < M; i++) {
                    const unsigned v = epi_pack16<EPI>(f0[i], f1[i]);
                    if (i < hv) *reinterpret_cast<unsigned*>(dst) = v;
                    dst += a.out_stride;
                    ADF_STEP_FENCE();
                }
            } else {
#pragma unroll
                for (int i = 0; i < M; i++) {
                    const unsigned v = epi_pack16<EPI>(f0[i], f1[i]);
                    if (i < hv) {
                        reinterpret_cast<uint16_t*>(dst)[0] = (uint16_t)v;
                        reinterpret_cast<uint16_t*>(dst)[1] = (uint16_t)(v >> 16);
                    }
                    dst += a.out_stride;
                    ADF_STEP_FENCE();
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < M; i++) {
            if (r0 + i < h) {
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if (col + e < a.nscan) {
                        char* dst = ob + ooff + (unsigned)(e * a.out_cn * esz);
                        if (EPI == EPI_WLS_CONF) {
                            const float rcp = 1.0f / (f1[i][e] + ADF_EPS);             // DF.cpp:295
                            *reinterpret_cast<int16_t*>(dst) = sat16(f0[i][e] * rcp);  // DF.cpp:296
                        } else {
                            // generic FGS (FGS.cpp:216-218): with two right-hand sides the second one is
                            // the next interleaved channel of the same image
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                const float x = r ? f1[i][e] : f0[i][e];
                                char* d = dst + r * esz;
                                if (EPI == EPI_I16) *reinterpret_cast<int16_t*>(d) = sat16(x);
                                else if (EPI == EPI_U8) *reinterpret_cast<uint8_t*>(d) = sat8(x);
                                else *reinterpret_cast<float*>(d) = x;
                            }
                        }
                    }
                }
            }
            ooff += (unsigned)a.out_stride;
        }
        }
    }
    ADF_STAMP(5); ADF_DRAIN(); ADF_STAMP(6); ADF_WSTAMP(1);
}

// Dynamic LDS of a two-right-hand-side instantiation: the wave rings of the LDS-DMA loads (A/B build -DADF_V_GLDS=1 only).
// Beyond 48 KiB the function needs its limit raised, per function AND device.  No "already done" memo: one keyed on the
// function's TYPE -- identical for every instantiation -- would skip the attribute for all kernels but the first
// (ADVICE r3); the call is cheap next to a pass and only this experimental build makes it.
template <typename K>
hipError_t v_allow_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int M, int VCW = VC, int NCH = 64>
hipError_t launch_v(const WavePassArgs& a, int n_rhs, int epi, int n_pairs, hipStream_t st)
{
    dim3 grid(a.pitch / VCW, n_pairs), block(VT);
    const size_t ring = (size_t)(VT / 64) * VRing<M>::SLOTS * 1024;
#define ADF_LVF(RR, EE, FF)                                                                                   \
    do {                                                                                                      \
        const size_t lds = ((RR) > 1 && ADF_V_GLDS && VCW == VC) ? ring : 0;                                  \
        if (lds > 16 * 1024) {                                                                                \
            hipError_t e = v_allow_lds(wave_vpass_kernel<M, RR, EE, VCW, NCH, FF>, lds);                      \
            if (e != hipSuccess) return e;                                                                    \
        }                                                                                                     \
        hipLaunchKernelGGL((wave_vpass_kernel<M, RR, EE, VCW, NCH, FF>), grid, block, lds, st, a);            \
    } while (0)
#define ADF_LV(RR, EE) ADF_LVF(RR, EE, false)
    // (the packed 4-byte stores of the last pass: single-channel output, even ROI width, everything 4-byte aligned)
    const bool packed_out = a.out_cn == 1 && (a.nscan & 1) == 0 && ((a.out_x0 * 2) & 3) == 0 &&
                            ((reinterpret_cast<uintptr_t>(a.out) | (uintptr_t)a.out_stride | (uintptr_t)a.out_pair_stride) & 3u) == 0;
    if (n_rhs == 2 && epi == EPI_PLANES) ADF_LVF(2, EPI_PLANES, (ADF_V_STORE_IN_SOLVE & 1) != 0);
    else if (n_rhs == 2 && epi == EPI_WLS_CONF && (ADF_V_STORE_IN_SOLVE & 2) && packed_out) ADF_LVF(2, EPI_WLS_CONF, (ADF_V_STORE_IN_SOLVE & 2) != 0);
    else if (n_rhs == 2 && epi == EPI_WLS_CONF) ADF_LV(2, EPI_WLS_CONF);
    else if (n_rhs == 1 && epi == EPI_PLANES) ADF_LV(1, EPI_PLANES);
    else if (n_rhs == 1 && epi == EPI_I16) ADF_LV(1, EPI_I16);
    else if (n_rhs == 1 && epi == EPI_F32) ADF_LV(1, EPI_F32);
    else if (n_rhs == 1 && epi == EPI_U8) ADF_LV(1, EPI_U8);
    else if (n_rhs == 2 && epi == EPI_I16) ADF_LV(2, EPI_I16);   // channel pairs of a generic FGS source
    else if (n_rhs == 2 && epi == EPI_F32) ADF_LV(2, EPI_F32);
    else if (n_rhs == 2 && epi == EPI_U8) ADF_LV(2, EPI_U8);
    else return hipErrorInvalidValue;
#undef ADF_LV
#undef ADF_LVF
    return hipGetLastError();
}

} // namespace

int wave_max_col_len() { return 128 * 34; }

hipError_t launch_wave_vpass(const WavePassArgs& a, int n_rhs, int epilogue, int n_pairs, hipStream_t st)
{
    if (a.len < 2 || a.len > wave_max_col_len() || a.pitch % 64 != 0 || a.pitch < a.nscan) return hipErrorInvalidValue;
    if (a.len > 64 * 34) {   // taller than 2176 rows: half strips of 128 chunks
        const int m = (a.len + 127) / 128;
        if (m <= 20) return launch_v<20, VC / 2, 128>(a, n_rhs, epilogue, n_pairs, st);
        if (m <= 26) return launch_v<26, VC / 2, 128>(a, n_rhs, epilogue, n_pairs, st);
        return launch_v<34, VC / 2, 128>(a, n_rhs, epilogue, n_pairs, st);
    }
    const int m = (a.len + 63) / 64;
    if (m <= 2) return launch_v<2>(a, n_rhs, epilogue, n_pairs, st);
    if (m <= 4) return launch_v<4>(a, n_rhs, epilogue, n_pairs, st);
    if (m <= 8) return launch_v<8>(a, n_rhs, epilogue, n_pairs, st);
    if (m <= 12) return launch_v<12>(a, n_rhs, epilogue, n_pairs, st);
    if (m <= 18) return launch_v<18>(a, n_rhs, epilogue, n_pairs, st);
    if (m <= 26) return launch_v<26>(a, n_rhs, epilogue, n_pairs, st);
    return launch_v<34>(a, n_rhs, epilogue, n_pairs, st);
}

} // namespace adf
