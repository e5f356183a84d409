/*
 * adf_oracle_bm.c -- CPU oracle of the block matcher that feeds the filter.
 *
 * TEST INFRASTRUCTURE ONLY (see adf_oracle.h).
 *
 * The reference's filter takes its disparity maps from cv::StereoBM / cv::StereoSGBM
 * (disparity_filters.cpp:386-449: createDisparityWLSFilter / createRightMatcher; sample
 * disparity_filtering.cpp:151,214).  Those classes live in OpenCV's calib3d module, which is NOT under
 * /root/reference (version unpinned): PARITY UNPINNED at that boundary.  What is restated
 * here is the published StereoBM algorithm (Konolige's block matcher as OpenCV 3.x implements it),
 * from its documentation and from memory of its structure:
 *   1. x-Sobel prefilter clipped to [-cap, cap] and offset by cap (PREFILTER_XSOBEL, the default);
 *   2. SAD over a blockSize x blockSize window for every disparity in [minDisparity,
 *      minDisparity + numDisparities);
 *   3. winner-take-all (ties go to the LARGEST disparity: the search buffer runs from the largest
 *      disparity down and keeps the first strict minimum), optional texture and uniqueness tests,
 *      parabola-like sub-pixel fit, result in fixed point with 4 fractional bits (CV_16SC1);
 *   4. pixels without a full search range or a full window (column-wise AND row-wise: the valid rectangle is
 *      x in [max(maxD, 0) + w/2, W + min(minD, 0) - w/2), y in [w/2, H - w/2), calib3d's getValidDisparityROI over the
 *      whole image), and rejected pixels, hold (minDisparity - 1) * 16.  The row rule is evidenced by the one raw
 *      StereoBM map the reference tree publishes: tutorials/images/ambush_5_bm.png is non-zero exactly from row 4 to
 *      row H-5 and from column 131 on for StereoBM(128, 9) (tests/tutorial_replay.py); until round 4 this statement
 *      matched rows outside that rectangle with replicated image rows instead.
 * Suspected deviations from calib3d's StereoBM (round-1 code review, from memory of OpenCV's source): (a) SETTLED in
 * round 4 -- the suspicion was that calib3d emits disparities for the outer blockSize/2 columns of [lofs, W-rofs)
 * with clamped window columns; the published ambush_5_bm.png has column 130 empty and column 131 = maxD + w/2 filled
 * (349 of 428 rows), so calib3d marks that band FILTERED as this statement does; (b) OPEN -- calib3d's
 * prefilterXSobel works on row pairs and may fill the last row of an odd-height image with the cap value, where this
 * statement filters every row with reflected neighbours: the difference would reach the block sums of output row
 * H-1-w/2 of odd-height images only (the tutorial pair has even heights at both resolutions, so it cannot tell).
 * What the reference itself fixes are the conventions around the call: the right-view matcher's
 * parameters (disparity_filters.cpp:421-431), the settings the filter forces on the matcher (:389-390,
 * 399-400: texture threshold 0, uniqueness ratio 0, no speckle filter, no left-right check) and the
 * ROI it derives from them (:401).  The one known-answer anchor the reference holds for a block matcher
 * is its stereo module's test (modules/stereo/test/test_block_matching.cpp:61-82,88-92,148): the
 * Tsukuba pair testdata/imL2l.bmp / imL2.bmp against testdata/groundtruth.bmp with at most 20 % of the
 * pixels off by more than two disparity levels; tests/test_oracle_bm.py applies exactly that bar.
 */
#include "adf_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Step 1.  dst: dense W x H.  Row borders reflect (row -1 -> row 1), columns 0 and W-1 hold cap. */
void adf_oracle_bm_prefilter_xsobel(const uint8_t* src, ptrdiff_t stride, int W, int H, int cap, uint8_t* dst)
{
    for (int y = 0; y < H; y++) {
        const uint8_t* r0 = src + (ptrdiff_t)(y > 0 ? y - 1 : (H > 1 ? 1 : 0)) * stride;
        const uint8_t* r1 = src + (ptrdiff_t)y * stride;
        const uint8_t* r2 = src + (ptrdiff_t)(y < H - 1 ? y + 1 : (H > 1 ? H - 2 : 0)) * stride;
        uint8_t* d = dst + (size_t)y * W;
        d[0] = (uint8_t)cap;
        if (W > 1) d[W - 1] = (uint8_t)cap;
        for (int x = 1; x < W - 1; x++) {
            const int v = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
            d[x] = (uint8_t)(clampi(v, -cap, cap) + cap);
        }
    }
}

int adf_oracle_bm_compute(const adf_oracle_bm_params* p, const uint8_t* left, ptrdiff_t lstride,
                          const uint8_t* right, ptrdiff_t rstride, int W, int H,
                          int16_t* disp, ptrdiff_t dstride)
{
    const int nd = p->num_disparities, md = p->min_disparity, wsz = p->block_size, w2 = wsz / 2;
    const int cap = p->prefilter_cap;
    if (nd <= 0 || nd % 16 || wsz < 5 || wsz > 21 || !(wsz & 1) || cap < 1 || cap > 63 || W <= 0 || H <= 0) return -1;
    const int maxd = md + nd - 1;
    const int lofs = maxd > 0 ? maxd : 0, rofs = md < 0 ? -md : 0;     /* columns without a full search range */
    const int xs = lofs + w2, xe = W - rofs - w2;                       /* outputs [xs, xe) are matched */
    const int16_t filtered = (int16_t)((md - 1) * 16);
    uint8_t* L = (uint8_t*)malloc((size_t)W * H);
    uint8_t* R = (uint8_t*)malloc((size_t)W * H);
    int* vsum = (int*)malloc(sizeof(int) * (size_t)W);
    int* S = (int*)malloc(sizeof(int) * (size_t)nd * (size_t)W);       /* S[k][x] of the current row */
    int* T = (int*)malloc(sizeof(int) * (size_t)W);
    if (!L || !R || !vsum || !S || !T) { free(L); free(R); free(vsum); free(S); free(T); return -2; }
    adf_oracle_bm_prefilter_xsobel(left, lstride, W, H, cap, L);
    adf_oracle_bm_prefilter_xsobel(right, rstride, W, H, cap, R);
    for (int y = 0; y < H; y++) {
        int16_t* drow = disp + (ptrdiff_t)y * dstride;
        for (int x = 0; x < W; x++) drow[x] = filtered;
        if (xe <= xs) continue;
        if (y < w2 || y >= H - w2) continue;                           /* rows without a full window: see (4) in the header */
        /* texture: sum over the window of |L - cap| */
        for (int x = xs - w2; x < xe + w2; x++) {
            int s = 0;
            for (int dy = -w2; dy <= w2; dy++) s += abs((int)L[(size_t)clampi(y + dy, 0, H - 1) * W + x] - cap);
            vsum[x] = s;
        }
        for (int x = xs; x < xe; x++) {
            int s = 0;
            for (int dx = -w2; dx <= w2; dx++) s += vsum[x + dx];
            T[x] = s;
        }
        for (int k = 0; k < nd; k++) {
            const int d = md + k;
            for (int x = xs - w2; x < xe + w2; x++) {
                int s = 0;
                for (int dy = -w2; dy <= w2; dy++) {
                    const size_t row = (size_t)clampi(y + dy, 0, H - 1) * W;
                    s += abs((int)L[row + x] - (int)R[row + x - d]);
                }
                vsum[x] = s;
            }
            int* Sk = S + (size_t)k * W;
            for (int x = xs; x < xe; x++) {
                int s = 0;
                for (int dx = -w2; dx <= w2; dx++) s += vsum[x + dx];
                Sk[x] = s;
            }
        }
        for (int x = xs; x < xe; x++) {
            int best = 0x7fffffff, bk = -1;
            for (int k = nd - 1; k >= 0; k--) {                       /* largest disparity first, strict minimum */
                const int s = S[(size_t)k * W + x];
                if (s < best) { best = s; bk = k; }
            }
            if (T[x] < p->texture_threshold) continue;
            if (p->uniqueness_ratio > 0) {
                const int thresh = best + best * p->uniqueness_ratio / 100;
                int k;
                for (k = 0; k < nd; k++)
                    if ((k < bk - 1 || k > bk + 1) && S[(size_t)k * W + x] <= thresh) break;
                if (k < nd) continue;
            }
            const int pv = S[(size_t)(bk > 0 ? bk - 1 : 1) * W + x];          /* cost one disparity below */
            const int nv = S[(size_t)(bk < nd - 1 ? bk + 1 : nd - 2) * W + x]; /* one above */
            const int dd = pv + nv - 2 * best + abs(pv - nv);
            drow[x] = (int16_t)(((bk + md) * 256 + (dd != 0 ? (pv - nv) * 256 / dd : 0) + 15) >> 4);
        }
    }
    free(L); free(R); free(vsum); free(S); free(T);
    return 0;
}
