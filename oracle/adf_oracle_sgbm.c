/*
 * adf_oracle_sgbm.c -- CPU oracle of the semi-global matcher that feeds the filter.
 *
 * TEST INFRASTRUCTURE ONLY (see adf_oracle.h).
 *
 * The reference's sample takes its maps from cv::StereoSGBM with P1 = 24*w^2, P2 = 96*w^2, preFilterCap 63 and
 * MODE_SGBM_3WAY (samples/disparity_filtering.cpp:166-176, 229-235); the filter factories only read / set its
 * parameters (disparity_filters.cpp:404-409, 432-445).  The class lives in OpenCV's calib3d, which is NOT under
 * /root/reference (version unpinned): PARITY UNPINNED at that boundary.  Restated here is the
 * published algorithm -- H. Hirschmueller, "Stereo Processing by Semiglobal Matching and Mutual Information"
 * (PAMI 2008), formula 13, with the Birchfield-Tomasi pixel cost -- with the conventions of OpenCV's
 * implementation as far as they are documented or visible in the in-tree derivative of that code,
 * modules/stereo/src/stereo_binary_sgbm.cpp (the reference's own copy of the aggregation loop with a census
 * cost in place of BT), which is cited line by line below:
 *
 *   1. per pixel and channel a clipped x-derivative (Sobel-like 3x3, clipped to [-ftzero, ftzero] and offset by
 *      ftzero, ftzero = max(preFilterCap, 15) | 1) and the raw intensity; border columns hold ftzero;
 *   2. Birchfield-Tomasi cost between pixel x of image 1 and x-d of image 2 on both signals (half-sample
 *      interpolated extrema), the raw-intensity term scaled by 1/4; summed over the channels;
 *   3. block cost C(x,d): box sum of (2) over blockSize x blockSize, window clamped at the borders of the
 *      matchable area (stereo_binary_sgbm.cpp:205-276 is the same running-sum structure);
 *   4. path costs L_r(p,d) = C(p,d) + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, delta) - delta with
 *      delta = min_k L_r(p-r,k) + P2 (stereo_binary_sgbm.cpp:286-301, 352, 419-446, 517): the code subtracts delta, i.e.
 *      P2 MORE than formula 13 of the paper, so L lies in [C-P2, C] and S is the paper's sum minus npaths*P2 -- winner
 *      and sub-pixel fit do not see the shift, the uniqueness test (:543-547) and the 16-bit saturation do (round 3:
 *      rounds 1-2 subtracted min_k only).  Neighbours outside [0,D) = SHRT_MAX (:323-324),
 *      path buffers start at zero (:191-194), costs kept in 16 bits with saturation (S is saturated after every path
 *      here; the in-tree loop adds its four forward paths before it saturates, :443 -- the two differ only when a
 *      partial sum leaves 16 bits and the full sum does not); MODE_SGBM_3WAY uses three
 *      paths: from the left, from the top, from the right; MODE_SGBM five (left, up-left, up, up-right, right: the
 *      forward sweep of :286-446 plus the backward one of :456-534), MODE_HH eight (two mirrored passes, :173-186);
 *   5. S = sum of the three paths; winner = the FIRST disparity with the smallest S (:519-528, strict <);
 *      uniqueness test (:543-547); sub-pixel parabola fit with 4 fractional bits (:584-591); result
 *      d + minDisparity*16 (:596); pixels outside [minX1, maxX1) and rejected ones hold (minDisparity-1)*16
 *      (:449-453, INVALID_DISP_SCALED :156);
 *   6. a 3x3 median over the CV_16S map (StereoSGBM::compute post-filter), border replicated.
 * The matcher's own left-right check (disp12MaxDiff, :548-556, 598-613) is restated too, so that cv::StereoSGBM::create's
 * defaults run; the filter factories switch it off (disp12MaxDiff = 1000000, disparity_filters.cpp:389, 444).
 * Deliberately NOT restated (not reachable from the filter: speckleWindowSize = 0, disparity_filters.cpp:390, 445): the
 * speckle filter.  OpenCV's 3-way code also cuts the image into horizontal stripes for its threads and restarts the
 * vertical path in each (results depend on the stripe count); this restatement is the one-stripe case.
 * Matchable columns: minX1 = max(maxD, 0), maxX1 = W + min(minD, 0) as in calib3d (the in-tree census variant has
 * max(-maxD, 0) at :148); the SGBM branch of createDisparityWLSFilter cuts exactly these columns off the ROI
 * (disparity_filters.cpp:407).
 *
 * The one known-answer anchor the reference holds for a semi-global matcher is its stereo module's test
 * (modules/stereo/test/test_block_matching.cpp:157-238): the Tsukuba pair against testdata/groundtruth.bmp, 16
 * disparities, at most 10 % of the pixels off by more than two disparity levels after scaling the map to 8 bits;
 * tests/test_oracle_sgbm.py applies that bar.
 */
#include "adf_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define SGBM_MAX_COST SHRT_MAX
#define DISP_SHIFT 4
#define DISP_SCALE (1 << DISP_SHIFT)

static inline int imin2(int a, int b) { return a < b ? a : b; }
static inline int imax2(int a, int b) { return a > b ? a : b; }
static inline int16_t sat_s16(int v) { return (int16_t)(v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v)); }

/* Steps 1-2 precomputation: for every pixel and every signal s in [0, 2cn) (s < cn: derivative of channel s,
 * s >= cn: intensity of channel s-cn) the triple (v, v0, v1) = value and the extrema of the value and its two
 * half-sample neighbours.  rec: [H][W][2cn][3] bytes. */
void adf_oracle_sgbm_signals(const uint8_t* img, ptrdiff_t stride, int cn, int W, int H, int prefilter_cap, uint8_t* rec)
{
    const int ftzero = imax2(prefilter_cap, 15) | 1;
    const int ns = 2 * cn;
    uint8_t* line = (uint8_t*)malloc((size_t)W * ns);
    for (int y = 0; y < H; y++) {
        const uint8_t* row = img + (ptrdiff_t)y * stride;
        const uint8_t* up = y > 0 ? row - stride : row;
        const uint8_t* dn = y < H - 1 ? row + stride : row;
        for (int s = 0; s < ns; s++) { line[(size_t)s * W] = (uint8_t)ftzero; line[(size_t)s * W + W - 1] = (uint8_t)ftzero; }
        for (int x = 1; x < W - 1; x++)
            for (int c = 0; c < cn; c++) {
                const int a = x * cn + c;
                int g = (row[a + cn] - row[a - cn]) * 2 + up[a + cn] - up[a - cn] + dn[a + cn] - dn[a - cn];
                g = (g < -ftzero ? -ftzero : (g > ftzero ? ftzero : g)) + ftzero;
                line[(size_t)c * W + x] = (uint8_t)g;
                line[(size_t)(cn + c) * W + x] = row[a];
            }
        for (int s = 0; s < ns; s++)
            for (int x = 0; x < W; x++) {
                const uint8_t* p = line + (size_t)s * W;
                const int v = p[x];
                const int vl = x > 0 ? (v + p[x - 1]) / 2 : v;
                const int vr = x < W - 1 ? (v + p[x + 1]) / 2 : v;
                uint8_t* o = rec + (((size_t)y * W + x) * ns + s) * 3;
                o[0] = (uint8_t)v; o[1] = (uint8_t)imin2(imin2(vl, vr), v); o[2] = (uint8_t)imax2(imax2(vl, vr), v);
            }
    }
    free(line);
}

/* Step 2: Birchfield-Tomasi cost of image-1 pixel record u against image-2 pixel record v, all signals. */
static inline int bt_cost(const uint8_t* u, const uint8_t* v, int cn)
{
    int cost = 0;
    for (int s = 0; s < 2 * cn; s++, u += 3, v += 3) {
        int c0 = imax2(0, u[0] - v[2]); c0 = imax2(c0, v[1] - u[0]);
        int c1 = imax2(0, v[0] - u[2]); c1 = imax2(c1, u[1] - v[0]);
        cost += imin2(c0, c1) >> (s < cn ? 0 : 2);
    }
    return cost;
}

typedef struct {
    int minD, D, W, H, cn, SW2, SH2, minX1, width1;
    const uint8_t *r1, *r2;
} sgbm_geom;

/* horizontal window sums of the pixel cost for image row y: hs[x1][d], x1 in [0, width1) */
static void hsum_row(const sgbm_geom* g, int y, int* pix, int* hs)
{
    const int ns3 = 6 * g->cn, D = g->D, w1 = g->width1;
    for (int x1 = 0; x1 < w1; x1++) {
        const int x = x1 + g->minX1;
        const uint8_t* u = g->r1 + ((size_t)y * g->W + x) * ns3;
        for (int d = 0; d < D; d++)
            pix[(size_t)x1 * D + d] = bt_cost(u, g->r2 + ((size_t)y * g->W + (x - (d + g->minD))) * ns3, g->cn);
    }
    for (int x1 = 0; x1 < w1; x1++)
        for (int d = 0; d < D; d++) {
            int s = 0;
            for (int j = -g->SW2; j <= g->SW2; j++) {
                int xx = x1 + j; xx = xx < 0 ? 0 : (xx > w1 - 1 ? w1 - 1 : xx);
                s += pix[(size_t)xx * D + d];
            }
            hs[(size_t)x1 * D + d] = s;
        }
}

static int sgbm_setup(const adf_oracle_sgbm_params* p, int W, int H, int cn, sgbm_geom* g, int* P1, int* P2, int* ur)
{
    if (!p || W <= 0 || H <= 0 || (cn != 1 && cn != 3)) return -1;
    if (p->num_disparities <= 0 || p->num_disparities % 16) return -1;
    if (p->mode != ADF_SGBM_MODE_3WAY && p->mode != ADF_SGBM_MODE_SGBM && p->mode != ADF_SGBM_MODE_HH &&
        p->mode != ADF_SGBM_MODE_3WAY_GENERIC) return -2;
    const int bs = p->block_size > 0 ? p->block_size : 5;
    if (!(bs & 1)) return -1;
    g->minD = p->min_disparity; g->D = p->num_disparities; g->W = W; g->H = H; g->cn = cn;
    g->SW2 = g->SH2 = bs / 2;
    const int maxD = g->minD + g->D;
    g->minX1 = imax2(maxD, 0);
    g->width1 = (W + imin2(g->minD, 0)) - g->minX1;
    *P1 = p->P1 > 0 ? p->P1 : 2;
    *P2 = imax2(p->P2 > 0 ? p->P2 : 5, *P1 + 1);
    *ur = p->uniqueness_ratio >= 0 ? p->uniqueness_ratio : 10;
    return 0;
}

/* Step 3 for the whole image: C[y][x1][d] (tests of the device cost kernel; small images only). */
int adf_oracle_sgbm_block_costs(const adf_oracle_sgbm_params* p, const uint8_t* img1, ptrdiff_t s1, const uint8_t* img2,
                                ptrdiff_t s2, int cn, int W, int H, int16_t* C)
{
    sgbm_geom g; int P1, P2, ur;
    int rc = sgbm_setup(p, W, H, cn, &g, &P1, &P2, &ur);
    if (rc) return rc;
    if (g.width1 <= 0) return 0;
    uint8_t* r1 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
    uint8_t* r2 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
    adf_oracle_sgbm_signals(img1, s1, cn, W, H, p->prefilter_cap, r1);
    adf_oracle_sgbm_signals(img2, s2, cn, W, H, p->prefilter_cap, r2);
    g.r1 = r1; g.r2 = r2;
    const size_t rowsz = (size_t)g.width1 * g.D;
    int* pix = (int*)malloc(sizeof(int) * rowsz);
    int* hs = (int*)malloc(sizeof(int) * rowsz * (size_t)H);
    for (int y = 0; y < H; y++) hsum_row(&g, y, pix, hs + rowsz * (size_t)y);
    for (int y = 0; y < H; y++)
        for (size_t i = 0; i < rowsz; i++) {
            int s = 0;
            for (int k = -g.SH2; k <= g.SH2; k++) {
                int yy = y + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                s += hs[rowsz * (size_t)yy + i];
            }
            C[rowsz * (size_t)y + i] = sat_s16(s);
        }
    free(pix); free(hs); free(r1); free(r2);
    return 0;
}

/* One step of the recurrence for all disparities (stereo_binary_sgbm.cpp:423: "L0 = Cpd + min(...) - delta0"): Lprev /
 * Lout have D+2 entries, index 0 and D+1 are the d = -1 and d = D guards (SHRT_MAX).  Returns min_k Lout[k]. */
static inline int path_step(const int16_t* Cp, const int16_t* Lprev, int minprev, int16_t* Lout, int D, int P1, int P2)
{
    int mn = SGBM_MAX_COST;
    const int delta = minprev + P2;
    for (int d = 0; d < D; d++) {
        const int m = imin2(imin2((int)Lprev[d + 1], Lprev[d] + P1), imin2(Lprev[d + 2] + P1, delta));
        const int L = sat_s16(Cp[d] + m - delta);
        Lout[d + 1] = (int16_t)L;
        mn = imin2(mn, L);
    }
    Lout[0] = Lout[D + 1] = SGBM_MAX_COST;
    return mn;
}

/* Winner, uniqueness test, sub-pixel fit and the matcher's own left-right check for ONE image row whose summed
 * costs S[x1][d] are complete (stereo_binary_sgbm.cpp:449-453, 519-613): x runs from the right (:456); the reverse
 * map disp2 keeps, per column of image 2, the disparity of the cheapest winner pointing at it (:548-556); a pixel
 * is invalidated when BOTH roundings of its disparity disagree with disp2 by more than disp12MaxDiff (:598-613).
 * disp12 <= 0 means 1 (:141); the filter factory sets 1000000, which switches the check off. */
static void select_row(const int16_t* S, int w1, int D, int W, int minD, int minX1, int ur, int disp12, int16_t* out,
                       int16_t* disp2ptr, int16_t* disp2cost)
{
    const int invalid = (minD - 1) * DISP_SCALE;
    const int maxdiff = disp12 > 0 ? disp12 : 1;
    for (int x = 0; x < W; x++) { disp2ptr[x] = (int16_t)invalid; disp2cost[x] = SGBM_MAX_COST; }     /* :449-453 */
    for (int x = w1 - 1; x >= 0; x--) {
        const int16_t* Sp = S + (size_t)x * D;
        int minS = SGBM_MAX_COST, best = -1;
        for (int d = 0; d < D; d++) if (Sp[d] < minS) { minS = Sp[d]; best = d; }                     /* :524-528 */
        if (best < 0) continue;                                          /* every S saturated: stays invalid */
        int d;
        for (d = 0; d < D; d++)                                          /* :543-547 */
            if (Sp[d] * (100 - ur) < minS * 100 && abs(best - d) > 1) break;
        if (d < D) continue;
        d = best;
        const int x2 = x + minX1 - d - minD;                             /* :549-554 */
        if (disp2cost[x2] > minS) { disp2cost[x2] = (int16_t)minS; disp2ptr[x2] = (int16_t)(d + minD); }
        if (0 < d && d < D - 1) {                                        /* :584-591 */
            const int denom2 = imax2(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
            d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
        } else
            d *= DISP_SCALE;
        out[x + minX1] = (int16_t)(d + minD * DISP_SCALE);               /* :596 */
    }
    for (int x = minX1; x < minX1 + w1; x++) {                           /* :598-613 */
        const int d1 = out[x];
        if (d1 == invalid) continue;
        const int dlo = d1 >> DISP_SHIFT, dhi = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
        const int xlo = x - dlo, xhi = x - dhi;
        if (0 <= xlo && xlo < W && disp2ptr[xlo] >= minD && abs(disp2ptr[xlo] - dlo) > maxdiff &&
            0 <= xhi && xhi < W && disp2ptr[xhi] >= minD && abs(disp2ptr[xhi] - dhi) > maxdiff)
            out[x] = (int16_t)invalid;
    }
}

/* Formula 13 along an arbitrary direction of travel (dx, dy), added into the volume S[y][x][d] (saturating):
 * the pixel before (x, y) on the path is (x - dx, y - dy); a path entering the matchable area starts from zeros
 * (the border cells of the Lr / minLr buffers are cleared, stereo_binary_sgbm.cpp:191-194, 279-283).  MODE_SGBM sums
 * five such paths -- from the left, up-left, up, up-right (the four of the forward sweep, :286-301) and from the
 * right (the backward sweep of the single-pass mode, :456-534); MODE_HH eight (the second pass mirrors the first, :173-186). */
static void add_path(const int16_t* C, int16_t* S, int H, int w1, int D, int dx, int dy, int P1, int P2)
{
    const size_t LW = (size_t)D + 2;
    int16_t* prev = (int16_t*)malloc(sizeof(int16_t) * LW * (size_t)w1);
    int16_t* cur = (int16_t*)malloc(sizeof(int16_t) * LW * (size_t)w1);
    int* mprev = (int*)malloc(sizeof(int) * (size_t)w1);
    int* mcur = (int*)malloc(sizeof(int) * (size_t)w1);
    int16_t* zero = (int16_t*)calloc(LW, sizeof(int16_t));
    zero[0] = zero[D + 1] = SGBM_MAX_COST;
    const int ystart = dy >= 0 ? 0 : H - 1, ystep = dy >= 0 ? 1 : -1;
    const int xstart = dx >= 0 ? 0 : w1 - 1, xstep = dx >= 0 ? 1 : -1;
    for (int yi = 0, y = ystart; yi < H; yi++, y += ystep) {
        for (int xi = 0, x = xstart; xi < w1; xi++, x += xstep) {
            const int px = x - dx;
            const int16_t* Lp = zero; int mp = 0;
            if (dy == 0) { if (xi > 0) { Lp = cur + LW * (size_t)px; mp = mcur[px]; } }
            else if (yi > 0 && px >= 0 && px < w1) { Lp = prev + LW * (size_t)px; mp = mprev[px]; }
            const size_t o = ((size_t)y * w1 + x) * D;
            mcur[x] = path_step(C + o, Lp, mp, cur + LW * (size_t)x, D, P1, P2);
            for (int d = 0; d < D; d++) S[o + d] = sat_s16((int)S[o + d] + (int)cur[LW * (size_t)x + d + 1]);
        }
        { int16_t* t = prev; prev = cur; cur = t; int* m = mprev; mprev = mcur; mcur = m; }
    }
    free(prev); free(cur); free(mprev); free(mcur); free(zero);
}

/* MODE_SGBM / MODE_HH: whole volumes in memory (tests and small images). */
static void sgbm_multipath(const sgbm_geom* g, int mode, int P1, int P2, int ur, int disp12, int minD, int16_t* tmp)
{
    const int D = g->D, w1 = g->width1, H = g->H, W = g->W;
    const size_t rowsz = (size_t)w1 * D, K = (size_t)(2 * g->SH2 + 1);
    int* pix = (int*)malloc(sizeof(int) * rowsz);
    int* hs = (int*)malloc(sizeof(int) * rowsz * (size_t)H);
    int16_t* C = (int16_t*)malloc(sizeof(int16_t) * rowsz * (size_t)H);
    int16_t* S = (int16_t*)calloc(rowsz * (size_t)H, sizeof(int16_t));
    (void)K;
    for (int y = 0; y < H; y++) hsum_row(g, y, pix, hs + rowsz * (size_t)y);
    for (int y = 0; y < H; y++)
        for (size_t i = 0; i < rowsz; i++) {
            int s = 0;
            for (int k = -g->SH2; k <= g->SH2; k++) {
                int yy = y + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                s += hs[rowsz * (size_t)yy + i];
            }
            C[rowsz * (size_t)y + i] = sat_s16(s);
        }
    static const int dirs[8][2] = { {1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1} };
    const int ndirs = mode == ADF_SGBM_MODE_HH ? 8 : 5;
    for (int k = 0; k < ndirs; k++) {
        if (mode == ADF_SGBM_MODE_3WAY_GENERIC && (k & 1)) continue;   /* test hook: left, up, right only */
        add_path(C, S, H, w1, D, dirs[k][0], dirs[k][1], P1, P2);
    }
    int16_t* d2p = (int16_t*)malloc(sizeof(int16_t) * (size_t)W);
    int16_t* d2c = (int16_t*)malloc(sizeof(int16_t) * (size_t)W);
    for (int y = 0; y < H; y++)
        select_row(S + (size_t)y * rowsz, w1, D, W, minD, g->minX1, ur, disp12, tmp + (size_t)y * W, d2p, d2c);
    free(d2p); free(d2c);
    free(pix); free(hs); free(C); free(S);
}

/* Step 6: cv::medianBlur(disp, disp, 3) on CV_16SC1, border replicated. */
void adf_oracle_median3_16s(const int16_t* src, ptrdiff_t sstride, int16_t* dst, ptrdiff_t dstride, int W, int H)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int16_t v[9]; int n = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
                    v[n++] = src[(ptrdiff_t)yy * sstride + xx];
                }
            for (int i = 1; i < 9; i++) {                 /* insertion sort of nine values */
                int16_t t = v[i]; int j = i - 1;
                while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
                v[j + 1] = t;
            }
            dst[(ptrdiff_t)y * dstride + x] = v[4];
        }
}

/* disp: CV_16SC1, stride in ELEMENTS.  raw != NULL receives the map before the median filter. */
int adf_oracle_sgbm_compute(const adf_oracle_sgbm_params* p, const uint8_t* img1, ptrdiff_t s1, const uint8_t* img2,
                            ptrdiff_t s2, int cn, int W, int H, int16_t* disp, ptrdiff_t dstride, int16_t* raw)
{
    sgbm_geom g; int P1, P2, ur;
    int rc = sgbm_setup(p, W, H, cn, &g, &P1, &P2, &ur);
    if (rc) return rc;
    const int D = g.D, w1 = g.width1, minD = g.minD;
    const int16_t invalid = (int16_t)((minD - 1) * DISP_SCALE);
    int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * H);
    for (size_t i = 0; i < (size_t)W * H; i++) tmp[i] = invalid;
    if (w1 > 0 && p->mode != ADF_SGBM_MODE_3WAY) {
        uint8_t* r1 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
        uint8_t* r2 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
        adf_oracle_sgbm_signals(img1, s1, cn, W, H, p->prefilter_cap, r1);
        adf_oracle_sgbm_signals(img2, s2, cn, W, H, p->prefilter_cap, r2);
        g.r1 = r1; g.r2 = r2;
        sgbm_multipath(&g, p->mode, P1, P2, ur, p->disp12_max_diff, minD, tmp);
        free(r1); free(r2);
    } else if (w1 > 0) {
        uint8_t* r1 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
        uint8_t* r2 = (uint8_t*)malloc((size_t)W * H * 6 * cn);
        adf_oracle_sgbm_signals(img1, s1, cn, W, H, p->prefilter_cap, r1);
        adf_oracle_sgbm_signals(img2, s2, cn, W, H, p->prefilter_cap, r2);
        g.r1 = r1; g.r2 = r2;
        const size_t rowsz = (size_t)w1 * D;
        const int K = 2 * g.SH2 + 1;
        int* pix = (int*)malloc(sizeof(int) * rowsz);
        int* ring = (int*)malloc(sizeof(int) * rowsz * (size_t)K);      /* hs of rows y-SH2 .. y+SH2, slot = row % K */
        int* ring_row = (int*)malloc(sizeof(int) * (size_t)K);
        for (int k = 0; k < K; k++) ring_row[k] = -1;
        int16_t* C = (int16_t*)malloc(sizeof(int16_t) * rowsz);
        int16_t* S = (int16_t*)malloc(sizeof(int16_t) * rowsz);
        int16_t* Ltop = (int16_t*)calloc((size_t)w1 * (D + 2), sizeof(int16_t));   /* zero start, :191-194 */
        int* minTop = (int*)calloc((size_t)w1, sizeof(int));
        int16_t* d2p = (int16_t*)malloc(sizeof(int16_t) * (size_t)W);
        int16_t* d2c = (int16_t*)malloc(sizeof(int16_t) * (size_t)W);
        int16_t* La = (int16_t*)malloc(sizeof(int16_t) * (size_t)(D + 2));
        int16_t* Lb = (int16_t*)malloc(sizeof(int16_t) * (size_t)(D + 2));
        int16_t* Lt = (int16_t*)malloc(sizeof(int16_t) * (size_t)(D + 2));
        for (int x = 0; x < w1; x++) { Ltop[(size_t)x * (D + 2)] = SGBM_MAX_COST; Ltop[(size_t)x * (D + 2) + D + 1] = SGBM_MAX_COST; }
        for (int y = 0; y < H; y++) {
            /* step 3: block cost of row y */
            for (size_t i = 0; i < rowsz; i++) C[i] = 0;
            int acc_started = 0;
            for (int k = -g.SH2; k <= g.SH2; k++) {
                int yy = y + k; yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                int* slot = ring + rowsz * (size_t)(yy % K);
                if (ring_row[yy % K] != yy) { hsum_row(&g, yy, pix, slot); ring_row[yy % K] = yy; }
                (void)acc_started;
                for (size_t i = 0; i < rowsz; i++) { int s = C[i] + slot[i]; C[i] = sat_s16(s); }
            }
            /* step 4: path from the top and path from the left, S = their sum */
            memset(La, 0, sizeof(int16_t) * (size_t)(D + 2)); La[0] = La[D + 1] = SGBM_MAX_COST;
            int minLeft = 0;
            for (int x = 0; x < w1; x++) {
                const int16_t* Cp = C + (size_t)x * D;
                int16_t* Lt_x = Ltop + (size_t)x * (D + 2);
                const int mt = path_step(Cp, Lt_x, minTop[x], Lt, D, P1, P2);
                memcpy(Lt_x, Lt, sizeof(int16_t) * (size_t)(D + 2));
                minTop[x] = mt;
                minLeft = path_step(Cp, La, minLeft, Lb, D, P1, P2);
                { int16_t* t = La; La = Lb; Lb = t; }
                int16_t* Sp = S + (size_t)x * D;
                for (int d = 0; d < D; d++) Sp[d] = sat_s16((int)La[d + 1] + (int)Lt_x[d + 1]);
            }
            /* path from the right, winner, sub-pixel fit */
            memset(La, 0, sizeof(int16_t) * (size_t)(D + 2)); La[0] = La[D + 1] = SGBM_MAX_COST;
            int minRight = 0;
            int16_t* out = tmp + (size_t)y * W;
            for (int x = w1 - 1; x >= 0; x--) {
                const int16_t* Cp = C + (size_t)x * D;
                int16_t* Sp = S + (size_t)x * D;
                minRight = path_step(Cp, La, minRight, Lb, D, P1, P2);
                { int16_t* t = La; La = Lb; Lb = t; }
                for (int d = 0; d < D; d++) Sp[d] = sat_s16((int)Sp[d] + (int)La[d + 1]);
            }
            select_row(S, w1, D, W, minD, g.minX1, ur, p->disp12_max_diff, out, d2p, d2c);
        }
        free(d2p); free(d2c);
        free(pix); free(ring); free(ring_row); free(C); free(S); free(Ltop); free(minTop); free(La); free(Lb); free(Lt);
        free(r1); free(r2);
    }
    if (raw) memcpy(raw, tmp, sizeof(int16_t) * (size_t)W * H);
    adf_oracle_median3_16s(tmp, W, disp, dstride, W, H);
    free(tmp);
    return 0;
}
