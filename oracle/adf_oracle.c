/*
 * adf_oracle.c -- CPU restatement of cv::ximgproc::DisparityWLSFilter::filter
 * (LRC confidence map + Fast Global Smoother).  TEST INFRASTRUCTURE ONLY: see
 * the header for who may use it and for the pinning status.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: the reference is an SSE2-era build with separate
 * multiply and subtract roundings; a fused multiply-add would change results.
 *
 * Conventions chosen where the reference delegates to un-vendored OpenCV
 * (imgproc / core, version unpinned) -- "parity unpinned", documented here:
 *   box mean      : (float)((double)int_sum   * (1.0/k^2))   [boxFilter 16S->32F]
 *   box sq. mean  : (float)((double)exact_sum * (1.0/k^2))   [sqrBoxFilter, 64F sums]
 *   border        : BORDER_REFLECT_101 inside the ROI copy (DF.cpp:167-185)
 *   float->int16  : round-half-even, out-of-int-range/NaN -> INT_MIN -> -32768
 *                   (cvRound on SSE2 = cvtss2si "integer indefinite")
 *   1/(M+EPS)     : 1.0f / (x + 1e-43f) in float (denormals honoured)
 */
#include "adf_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ADF_EPS 1e-43f /* DF.cpp:47 */
#define ADF_LUT_LEVELS (3 * 256 * 256) /* FGS.cpp:150 */

/* ------------------------------------------------------------------ */
/* cv::parallel_for_(Range(0,num_stripes), body) stand-in: one thread  */
/* per stripe index, body converts the index to rows/cols itself       */
/* (FGS.cpp:468-469, :486-487, DF.cpp:316-317).                         */
/* ------------------------------------------------------------------ */
typedef void (*stripe_fn)(int stripe, int nstripes, void* ctx);

/* A persistent pool like the one behind cv::parallel_for_: workers are created once (up to the
 * largest stripe count asked for, minus the calling thread) and pick stripe indices off a shared
 * counter; the caller works too and returns when every stripe is done.  One region at a time. */
static struct {
    pthread_mutex_t mu; pthread_cond_t work, done;
    pthread_t* th; int nthreads;
    stripe_fn fn; void* ctx; int nstripes, next, running; unsigned long gen;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
             NULL, 0, NULL, NULL, 0, 0, 0, 0 };
static pthread_mutex_t g_region = PTHREAD_MUTEX_INITIALIZER;

static void pool_drain_locked(void)
{
    /* called with g_pool.mu held: run stripes until none is left */
    while (g_pool.next < g_pool.nstripes) {
        int s = g_pool.next++;
        stripe_fn fn = g_pool.fn; void* ctx = g_pool.ctx; int n = g_pool.nstripes;
        g_pool.running++;
        pthread_mutex_unlock(&g_pool.mu);
        fn(s, n, ctx);
        pthread_mutex_lock(&g_pool.mu);
        g_pool.running--;
    }
}

static void* pool_worker(void* arg)
{
    (void)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.work, &g_pool.mu);
        seen = g_pool.gen;
        pool_drain_locked();
        if (g_pool.running == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

static void parallel_stripes(int nstripes, stripe_fn fn, void* ctx)
{
    if (nstripes < 1) nstripes = 1;
    if (nstripes == 1) { fn(0, 1, ctx); return; }
    pthread_mutex_lock(&g_region);
    pthread_mutex_lock(&g_pool.mu);
    if (g_pool.nthreads < nstripes - 1) {           /* grow the pool (threads live until exit) */
        pthread_t* th = (pthread_t*)realloc(g_pool.th, sizeof(pthread_t) * (size_t)(nstripes - 1));
        if (th) {
            g_pool.th = th;
            while (g_pool.nthreads < nstripes - 1) {
                pthread_attr_t at;
                pthread_attr_init(&at);
                pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
                int rc = pthread_create(&g_pool.th[g_pool.nthreads], &at, pool_worker, NULL);
                pthread_attr_destroy(&at);
                if (rc != 0) break;                 /* fewer workers: the stripes still all run */
                g_pool.nthreads++;
            }
        }
    }
    g_pool.fn = fn; g_pool.ctx = ctx; g_pool.nstripes = nstripes; g_pool.next = 0; g_pool.running = 0;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.work);
    pool_drain_locked();
    while (g_pool.running > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_region);
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int stripe_size(int n, int nstripes) { return (int)ceil(n / (double)nstripes); }

/* ------------------------------------------------------------------ */
/* scalar conversions                                                   */
/* ------------------------------------------------------------------ */
static inline int cv_round_f(float v)
{
    /* cvRound(float) on SSE2 = cvtss2si: round-half-even, INT_MIN when the
     * value is NaN or does not fit an int. */
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)lrintf(v);
}

int16_t adf_oracle_sat16(float v)
{
    int r = cv_round_f(v);
    return (int16_t)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}

static inline uint8_t sat8(float v)
{
    int r = cv_round_f(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

void adf_oracle_default_params(adf_oracle_params* p)
{
    p->lambda = 8000.0;           /* DF.cpp:215 */
    p->sigma_color = 1.0;         /* DF.cpp:215 */
    p->use_confidence = 1;
    p->lrc_thresh = 24;           /* DF.cpp:154 */
    p->disc_radius = 5;           /* DF.cpp:155 */
    p->num_iter = 3;              /* EF.hpp:393 */
    p->lambda_attenuation = 0.25; /* EF.hpp:393 */
    p->order = ADF_ORDER_SCALAR;
    p->threads = 1;
}

/* ------------------------------------------------------------------ */
/* A5: weight LUT, FGS.cpp:663-675                                      */
/* ------------------------------------------------------------------ */
void adf_oracle_lut(float sigma, float* lut)
{
    for (int i = 0; i < ADF_LUT_LEVELS; i++)
        lut[i] = -expf(-sqrtf((float)i) / sigma);
}

/* the same table filled by stripes, FGS.cpp:154 (parallel_for_ over ComputeLUT_ParBody, :663-675) */
typedef struct { float sigma; float* lut; } lut_ctx;
static void lut_stripe(int s, int n, void* vctx)
{
    lut_ctx* c = (lut_ctx*)vctx;
    int sz = stripe_size(ADF_LUT_LEVELS, n);
    int start = imin(s * sz, ADF_LUT_LEVELS), end = imin((s + 1) * sz, ADF_LUT_LEVELS);
    for (int i = start; i < end; i++) c->lut[i] = -expf(-sqrtf((float)i) / c->sigma);
}

/* ------------------------------------------------------------------ */
/* A6: edge weights, FGS.cpp:50-59, :586-661                            */
/* ------------------------------------------------------------------ */
static inline float weight_of(const float* lut, const uint8_t* p1, const uint8_t* p2, int ch)
{
    int d0 = p1[0] - p2[0];
    int idx = d0 * d0;
    if (ch == 3) {
        int d1 = p1[1] - p2[1], d2 = p1[2] - p2[2];
        idx += d1 * d1 + d2 * d2;
    }
    return lut[idx];
}

typedef struct {
    const uint8_t* guide; ptrdiff_t stride; int ch, w, h;
    const float* lut; float* chor; float* cvert;
} weights_ctx;

static void weights_stripe(int s, int n, void* vctx)
{
    weights_ctx* c = (weights_ctx*)vctx;
    int sz = stripe_size(c->h, n);
    int start = imin(s * sz, c->h), end = imin((s + 1) * sz, c->h);
    for (int i = start; i < end; i++) {
        const uint8_t* row = c->guide + (ptrdiff_t)i * c->stride;
        float* hor = c->chor + (size_t)i * c->w;
        float* ver = c->cvert + (size_t)i * c->w;
        for (int j = 0; j < c->w - 1; j++) /* FGS.cpp:607-613 */
            hor[j] = weight_of(c->lut, row + j * c->ch, row + (j + 1) * c->ch, c->ch);
        hor[c->w - 1] = 0.0f;              /* FGS.cpp:614 */
        if (i < c->h - 1) {                /* FGS.cpp:635-656 */
            const uint8_t* nxt = row + c->stride;
            for (int j = 0; j < c->w; j++)
                ver[j] = weight_of(c->lut, row + j * c->ch, nxt + j * c->ch, c->ch);
        } else {
            for (int j = 0; j < c->w; j++) ver[j] = 0.0f; /* FGS.cpp:658-660 */
        }
    }
}

void adf_oracle_weights(const uint8_t* guide, ptrdiff_t stride, int ch, int w, int h,
                        const float* lut, float* chor, float* cvert, int threads)
{
    weights_ctx c = { guide, stride, ch, w, h, lut, chor, cvert };
    parallel_stripes(threads, weights_stripe, &c);
}

/* ------------------------------------------------------------------ */
/* A7/A8/A10: horizontal pass, FGS.cpp:251-476                          */
/* ------------------------------------------------------------------ */
typedef struct {
    float* cur; const float* C; float* D; int w, h; float lambda; int order;
} pass_ctx;

/* FGS.cpp:439-464 process_row -- the canonical scalar order. */
static void hrow_scalar(float* u, const float* C, float* D, int w, float lambda)
{
    float cp = lambda * C[0];
    D[0] = cp / (1 - cp);
    u[0] = u[0] / (1 - cp);
    for (int j = 1; j < w; j++) {
        float cc = lambda * C[j];
        float den = (1 - cp - cc) - D[j - 1] * cp;
        D[j] = cc / den;
        u[j] = (u[j] - u[j - 1] * cp) / den;
        cp = cc;
    }
    for (int j = w - 2; j >= 0; j--)
        u[j] = u[j] - D[j] * u[j + 1];
}

/* FGS.cpp:251-437 process_4row_block as built with CV_SIMD128: for one row of
 * the block, columns 1 .. (largest j with j < w-3, step 4)+3 use the PROC4
 * order (:305-314), the tail uses the scalar order (:357-383).  The four rows
 * of a block are independent, so one row at a time gives the same bits. */
static void hrow_refsimd(float* u, const float* C, float* D, int w, float lambda)
{
    float cp = lambda * C[0];
    D[0] = cp / (1 - cp);
    u[0] = u[0] / (1 - cp);
    int j = 1;
    for (; j < w - 3; j += 4) {
        for (int k = j; k < j + 4; k++) {
            float cc = lambda * C[k];
            float aux0 = D[k - 1] * cp;
            float aux1 = cc + cp;
            aux1 = 1.0f - aux1;
            aux0 = aux1 - aux0;
            D[k] = cc / aux0;
            aux1 = u[k - 1] * cp;
            aux1 = u[k] - aux1;
            u[k] = aux1 / aux0;
            cp = cc;
        }
    }
    for (; j < w; j++) {
        float cprev = lambda * C[j - 1];
        float cc = lambda * C[j];
        float den = (1 - cprev - cc) - D[j - 1] * cprev;
        D[j] = cc / den;
        u[j] = (u[j] - u[j - 1] * cprev) / den;
    }
    for (j = w - 2; j >= 0; j--) /* FGS.cpp:385-436: same operations either way */
        u[j] = u[j] - D[j] * u[j + 1];
}

/* The same block the way the reference's default build runs it: four rows at a time on 128-bit vectors,
 * 4x4 tiles transposed on the way in and out (FGS.cpp:295-351, 389-424).  Plain SSE2-class code through
 * gcc's vector extensions; lane r holds row r, and every lane performs exactly the operations of
 * hrow_refsimd in the same order, so the two produce the same bits (tests/test_oracle.py checks it). */
typedef float v4 __attribute__((vector_size(16)));
typedef int v4i __attribute__((vector_size(16)));
static inline v4 ld4(const float* p) { v4 v; memcpy(&v, p, sizeof v); return v; }
static inline void st4(float* p, v4 v) { memcpy(p, &v, sizeof v); }
#define TRANSPOSE4(a, b, c, d)                                        \
    do {                                                              \
        v4 t0 = __builtin_shuffle(a, b, (v4i){0, 4, 1, 5});           \
        v4 t1 = __builtin_shuffle(c, d, (v4i){0, 4, 1, 5});           \
        v4 t2 = __builtin_shuffle(a, b, (v4i){2, 6, 3, 7});           \
        v4 t3 = __builtin_shuffle(c, d, (v4i){2, 6, 3, 7});           \
        a = __builtin_shuffle(t0, t1, (v4i){0, 1, 4, 5});             \
        b = __builtin_shuffle(t0, t1, (v4i){2, 3, 6, 7});             \
        c = __builtin_shuffle(t2, t3, (v4i){0, 1, 4, 5});             \
        d = __builtin_shuffle(t2, t3, (v4i){2, 3, 6, 7});             \
    } while (0)

static void hblock4_refsimd(float* u, const float* C, float* D, int w, float lambda)
{
    /* u, C, D point at the first of four consecutive rows of pitch w */
    const size_t P = (size_t)w;
    const v4 one = {1.0f, 1.0f, 1.0f, 1.0f}, lam = {lambda, lambda, lambda, lambda};
    v4 cp, Dp, up;
    for (int r = 0; r < 4; r++) {                    /* FGS.cpp:278-290: first column, scalar */
        float c0 = lambda * C[r * P];
        D[r * P] = c0 / (1 - c0);
        u[r * P] = u[r * P] / (1 - c0);
        cp[r] = c0; Dp[r] = D[r * P]; up[r] = u[r * P];
    }
    int j = 1;
    for (; j < w - 3; j += 4) {                      /* FGS.cpp:295-351 */
        v4 c0 = ld4(C + j), c1 = ld4(C + P + j), c2 = ld4(C + 2 * P + j), c3 = ld4(C + 3 * P + j);
        v4 u0 = ld4(u + j), u1 = ld4(u + P + j), u2 = ld4(u + 2 * P + j), u3 = ld4(u + 3 * P + j);
        TRANSPOSE4(c0, c1, c2, c3);                  /* now c_k = column j+k of the four rows */
        TRANSPOSE4(u0, u1, u2, u3);
        v4 cc[4] = {c0 * lam, c1 * lam, c2 * lam, c3 * lam};
        v4 uu[4] = {u0, u1, u2, u3}, dd[4];
        for (int k = 0; k < 4; k++) {                /* PROC4, FGS.cpp:305-314 */
            v4 aux0 = Dp * cp;
            v4 aux1 = cc[k] + cp;
            aux1 = one - aux1;
            aux0 = aux1 - aux0;
            dd[k] = cc[k] / aux0;
            aux1 = up * cp;
            aux1 = uu[k] - aux1;
            uu[k] = aux1 / aux0;
            cp = cc[k]; Dp = dd[k]; up = uu[k];
        }
        v4 d0 = dd[0], d1 = dd[1], d2 = dd[2], d3 = dd[3];
        u0 = uu[0]; u1 = uu[1]; u2 = uu[2]; u3 = uu[3];
        TRANSPOSE4(d0, d1, d2, d3);
        TRANSPOSE4(u0, u1, u2, u3);
        st4(D + j, d0); st4(D + P + j, d1); st4(D + 2 * P + j, d2); st4(D + 3 * P + j, d3);
        st4(u + j, u0); st4(u + P + j, u1); st4(u + 2 * P + j, u2); st4(u + 3 * P + j, u3);
    }
    for (int r = 0; r < 4; r++) {                    /* FGS.cpp:357-383: scalar tail + backward sweep */
        float* ur = u + r * P; const float* Cr = C + r * P; float* Dr = D + r * P;
        for (int k = j; k < w; k++) {
            float cprev = lambda * Cr[k - 1];
            float ccur = lambda * Cr[k];
            float den = (1 - cprev - ccur) - Dr[k - 1] * cprev;
            Dr[k] = ccur / den;
            ur[k] = (ur[k] - ur[k - 1] * cprev) / den;
        }
        for (int k = w - 2; k >= 0; k--) ur[k] = ur[k] - Dr[k] * ur[k + 1];
    }
}

/* test hook: 1 = run the reference-SIMD order one row at a time (the scalar emulation) */
static int g_refsimd_rowwise = 0;
void adf_oracle_set_refsimd_rowwise(int on) { g_refsimd_rowwise = on; }

static void hpass_stripe(int s, int n, void* vctx)
{
    pass_ctx* c = (pass_ctx*)vctx;
    int sz = stripe_size(c->h, n);
    int start = imin(s * sz, c->h), end = imin((s + 1) * sz, c->h);
    int i = start;
    if (c->order == ADF_ORDER_REF_SIMD) /* FGS.cpp:472-473: 4-row blocks from the stripe start */
        for (; i < end - 3; i += 4) {
            if (c->order == ADF_ORDER_REF_SIMD && c->w >= 2 && !g_refsimd_rowwise)
                hblock4_refsimd(c->cur + (size_t)i * c->w, c->C + (size_t)i * c->w, c->D + (size_t)i * c->w, c->w, c->lambda);
            else
                for (int k = 0; k < 4; k++)
                    hrow_refsimd(c->cur + (size_t)(i + k) * c->w, c->C + (size_t)(i + k) * c->w,
                                 c->D + (size_t)(i + k) * c->w, c->w, c->lambda);
        }
    for (; i < end; i++)                /* FGS.cpp:474-475 */
        hrow_scalar(c->cur + (size_t)i * c->w, c->C + (size_t)i * c->w,
                    c->D + (size_t)i * c->w, c->w, c->lambda);
}

void adf_oracle_hpass(float* cur, const float* chor, float* interD, int w, int h,
                      float lambda, int order, int threads)
{
    pass_ctx c = { cur, chor, interD, w, h, lambda, order };
    parallel_stripes(threads, hpass_stripe, &c);
}

/* ------------------------------------------------------------------ */
/* A9: vertical pass, FGS.cpp:484-584                                   */
/* ------------------------------------------------------------------ */
static void vpass_stripe(int s, int n, void* vctx)
{
    pass_ctx* c = (pass_ctx*)vctx;
    const int w = c->w, h = c->h;
    const float lambda = c->lambda;
    int sz = stripe_size(w, n);
    int start = imin(s * sz, w), end = imin((s + 1) * sz, w);
    /* columns [start,end4) take the SIMD order in a CV_SIMD128 build (:520-547) */
    int end4 = (c->order == ADF_ORDER_REF_SIMD) ? start + 4 * ((end - start) / 4) : start;

    float* u0 = c->cur; const float* C0 = c->C; float* D0 = c->D;
    for (int j = start; j < end; j++) {          /* FGS.cpp:500-505 */
        float cc = lambda * C0[j];
        D0[j] = cc / (1 - cc);
        u0[j] = u0[j] / (1 - cc);
    }
    for (int i = 1; i < h; i++) {                /* FGS.cpp:506-557 */
        const float* Cr = c->C + (size_t)i * w; const float* Cp = Cr - w;
        float* Dr = c->D + (size_t)i * w;       const float* Dp = Dr - w;
        float* ur = c->cur + (size_t)i * w;     const float* up = ur - w;
        int j = start;
        if (!g_refsimd_rowwise)
            for (; j + 3 < end4; j += 4) {       /* :516-547 on 128-bit vectors */
                const v4 one = {1.0f, 1.0f, 1.0f, 1.0f}, lam = {lambda, lambda, lambda, lambda};
                v4 cp = ld4(Cp + j) * lam;
                v4 cc = ld4(Cr + j) * lam;
                v4 a = ld4(Dp + j) * cp;
                v4 b = cp + cc;
                b = b + a;
                a = one - b;
                st4(Dr + j, cc / a);
                v4 cm = ld4(up + j) * cp;
                v4 d = ld4(ur + j) - cm;
                st4(ur + j, d / a);
            }
        for (; j < end4; j++) {                  /* :524-546 */
            float cp = lambda * Cp[j];
            float cc = lambda * Cr[j];
            float a = Dp[j] * cp;
            float b = cp + cc;
            b = b + a;
            a = 1.0f - b;
            Dr[j] = cc / a;
            float cm = up[j] * cp;
            float d = ur[j] - cm;
            ur[j] = d / a;
        }
        for (; j < end; j++) {                   /* :549-556 */
            float cp = lambda * Cp[j];
            float cc = lambda * Cr[j];
            float den = (1 - cp - cc) - Dp[j] * cp;
            Dr[j] = cc / den;
            ur[j] = (ur[j] - up[j] * cp) / den;
        }
    }
    for (int i = h - 2; i >= 0; i--) {           /* FGS.cpp:560-583 */
        const float* Dr = c->D + (size_t)i * w;
        float* ur = c->cur + (size_t)i * w; const float* un = ur + w;
        for (int j = start; j < end; j++)
            ur[j] = ur[j] - Dr[j] * un[j];
    }
}

void adf_oracle_vpass(float* cur, const float* cvert, float* interD, int w, int h,
                      float lambda, int order, int threads)
{
    pass_ctx c = { cur, cvert, interD, w, h, lambda, order };
    parallel_stripes(threads, vpass_stripe, &c);
}

/* ------------------------------------------------------------------ */
/* A5+A6+A11: FGS init + filter on float planes, FGS.cpp:141-233         */
/* ------------------------------------------------------------------ */
int adf_oracle_fgs_planes(const uint8_t* guide, ptrdiff_t stride, int ch, int w, int h,
                          float* planes, int nplanes, double lambda, double sigma_color,
                          double lambda_attenuation, int num_iter, int order, int threads)
{
    /* FGS.cpp:143-144 */
    if (!guide || w <= 0 || h <= 0 || lambda < 0 || sigma_color < 0 || num_iter < 1) return 1;
    if (ch != 1 && ch != 3) return 1;
    size_t n = (size_t)w * h;
    float* lut = (float*)malloc(sizeof(float) * ADF_LUT_LEVELS);
    float* chor = (float*)malloc(sizeof(float) * n);
    float* cvert = (float*)malloc(sizeof(float) * n);
    float* interD = (float*)malloc(sizeof(float) * n);
    if (!lut || !chor || !cvert || !interD) { free(lut); free(chor); free(cvert); free(interD); return 4; }
    { lut_ctx lc = { (float)sigma_color, lut }; parallel_stripes(threads, lut_stripe, &lc); } /* FGS.cpp:145,154 */
    adf_oracle_weights(guide, stride, ch, w, h, lut, chor, cvert, threads);
    for (int p = 0; p < nplanes; p++) {
        float lam = (float)lambda;                      /* FGS.cpp:146,202 */
        float att = (float)lambda_attenuation;          /* FGS.cpp:147 */
        float* cur = planes + (size_t)p * n;
        for (int it = 0; it < num_iter; it++) {         /* FGS.cpp:207-212 */
            adf_oracle_hpass(cur, chor, interD, w, h, lam, order, threads);
            adf_oracle_vpass(cur, cvert, interD, w, h, lam, order, threads);
            lam *= att;
        }
    }
    free(lut); free(chor); free(cvert); free(interD);
    return 0;
}

/* FGS.cpp:182-233 / :687-691 generic entry: split, convert, filter, convert, merge */
int adf_oracle_fgs_filter(const uint8_t* guide, ptrdiff_t gstride, int gch, int w, int h,
                          const void* src, void* dst, int depth, int channels,
                          double lambda, double sigma_color, double lambda_attenuation,
                          int num_iter, int order, int threads)
{
    if (!src || !dst || channels < 1 || channels > 4) return 1;           /* FGS.cpp:184 */
    if (depth != ADF_DEPTH_8U && depth != ADF_DEPTH_16S && depth != ADF_DEPTH_32F) return 1;
    size_t n = (size_t)w * h;
    float* planes = (float*)malloc(sizeof(float) * n * (size_t)channels);
    if (!planes) return 4;
    for (int c = 0; c < channels; c++)                                     /* split + convertTo 32F */
        for (size_t i = 0; i < n; i++) {
            size_t k = i * (size_t)channels + (size_t)c;
            planes[(size_t)c * n + i] = depth == ADF_DEPTH_8U ? (float)((const uint8_t*)src)[k]
                                      : depth == ADF_DEPTH_16S ? (float)((const int16_t*)src)[k]
                                      : ((const float*)src)[k];
        }
    int rc = adf_oracle_fgs_planes(guide, gstride, gch, w, h, planes, channels, lambda, sigma_color,
                                   lambda_attenuation, num_iter, order, threads);
    if (rc == 0)
        for (int c = 0; c < channels; c++)                                 /* convertTo + merge */
            for (size_t i = 0; i < n; i++) {
                size_t k = i * (size_t)channels + (size_t)c;
                float v = planes[(size_t)c * n + i];
                if (depth == ADF_DEPTH_8U) ((uint8_t*)dst)[k] = sat8(v);
                else if (depth == ADF_DEPTH_16S) ((int16_t*)dst)[k] = adf_oracle_sat16(v);
                else ((float*)dst)[k] = v;
            }
    free(planes);
    return rc;
}

/* ------------------------------------------------------------------ */
/* Row-wise elementwise sweeps of DisparityWLSFilterImpl::filter (fill,  */
/* prologue, epilogue).  The reference runs them as whole-Mat operations */
/* (Scalar assignment, convertTo, mul: DF.cpp:284-296), which OpenCV     */
/* parallelises internally; here they are striped over rows like the     */
/* rest so that the port scales end to end.                              */
/* ------------------------------------------------------------------ */
enum { ROWS_ZERO_F32, ROWS_FILL_I16, ROWS_PROLOGUE, ROWS_EPILOGUE };
typedef struct {
    int op, rows, cols;
    float* f0;                       /* ZERO: plane; PROLOGUE/EPILOGUE: u0 plane (cols pitch) */
    float* f1;                       /* PROLOGUE/EPILOGUE: u1 plane */
    int16_t fill;
    int16_t* i16; ptrdiff_t i16_stride;          /* FILL / EPILOGUE destination (already offset to the ROI) */
    const int16_t* disp; const float* conf;      /* PROLOGUE sources (already offset to the ROI) */
    ptrdiff_t disp_stride; int conf_pitch;
    int unused0, unused1;
} rows_ctx;

static void rows_stripe(int s, int n, void* vctx)
{
    rows_ctx* c = (rows_ctx*)vctx;
    int sz = stripe_size(c->rows, n);
    int start = imin(s * sz, c->rows), end = imin((s + 1) * sz, c->rows);
    for (int i = start; i < end; i++) {
        switch (c->op) {
        case ROWS_ZERO_F32:
            memset(c->f0 + (size_t)i * c->cols, 0, sizeof(float) * (size_t)c->cols);
            break;
        case ROWS_FILL_I16: {
            int16_t* o = (int16_t*)((char*)c->i16 + (ptrdiff_t)i * c->i16_stride);
            for (int j = 0; j < c->cols; j++) o[j] = c->fill;
            break;
        }
        case ROWS_PROLOGUE: {                     /* DF.cpp:286-290 */
            const int16_t* d = (const int16_t*)((const char*)c->disp + (ptrdiff_t)i * c->disp_stride);
            const float* cf = c->conf + (size_t)i * c->conf_pitch;
            float* u0 = c->f0 + (size_t)i * c->cols;
            float* u1 = c->f1 + (size_t)i * c->cols;
            for (int j = 0; j < c->cols; j++) { u0[j] = cf[j] * (float)d[j]; u1[j] = cf[j]; }
            break;
        }
        case ROWS_EPILOGUE: {                     /* DF.cpp:295-296 */
            int16_t* o = (int16_t*)((char*)c->i16 + (ptrdiff_t)i * c->i16_stride);
            const float* u0 = c->f0 + (size_t)i * c->cols;
            const float* u1 = c->f1 + (size_t)i * c->cols;
            for (int j = 0; j < c->cols; j++) {
                float rcp = 1.0f / (u1[j] + ADF_EPS);
                o[j] = adf_oracle_sat16(u0[j] * rcp);
            }
            break;
        }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A3: depth-discontinuity maps, DF.cpp:105-115, :161-194, :343-373      */
/* ------------------------------------------------------------------ */
static inline int reflect101(int p, int len)
{
    /* cv::borderInterpolate(BORDER_REFLECT_101) */
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

typedef struct {
    const int16_t* disp; ptrdiff_t stride; int W, rx, ry, rw, rh, radius;
    float roll_off; float* dst;
} disc_ctx;

static void disc_stripe(int s, int n, void* vctx)
{
    disc_ctx* c = (disc_ctx*)vctx;
    const int rw = c->rw, rh = c->rh, r = c->radius, k = 2 * r + 1;
    const double scale = 1.0 / ((double)k * k);      /* boxFilter normalize=true */
    int sz = stripe_size(rh, n);
    int start = imin(s * sz, rh), end = imin((s + 1) * sz, rh);
    if (start >= end) return;
    /* running column sums over the 2r+1 reflected rows of the ROI copy */
    int64_t* cs1 = (int64_t*)calloc((size_t)rw, sizeof(int64_t));
    int64_t* cs2 = (int64_t*)calloc((size_t)rw, sizeof(int64_t));
    int* xi = (int*)malloc(sizeof(int) * (size_t)(rw + 2 * r));
    for (int x = -r; x < rw + r; x++) xi[x + r] = reflect101(x, rw);
#define ROWPTR(y) ((const int16_t*)((const char*)c->disp + (ptrdiff_t)(c->ry + reflect101((y), rh)) * c->stride) + c->rx)
    for (int dy = -r; dy <= r; dy++) {
        const int16_t* row = ROWPTR(start + dy);
        for (int x = 0; x < rw; x++) { int64_t v = row[x]; cs1[x] += v; cs2[x] += v * v; }
    }
    for (int y = start; y < end; y++) {
        float* out = c->dst + (size_t)(c->ry + y) * c->W + c->rx;
        int64_t s1 = 0, s2 = 0;
        for (int x = -r; x <= r; x++) { s1 += cs1[xi[x + r]]; s2 += cs2[xi[x + r]]; }
        for (int x = 0; x < rw; x++) {
            float mean = (float)((double)s1 * scale);          /* boxFilter    -> CV_32F */
            float sq = (float)((double)s2 * scale);            /* sqrBoxFilter -> CV_32F */
            float variance = sq - mean * mean;                 /* DF.cpp:369 */
            float v = 1.0f - c->roll_off * variance;           /* DF.cpp:370 */
            out[x] = v < 0.0f ? 0.0f : v;                      /* std::max(v, 0.0f) */
            if (x + 1 < rw) {
                s1 += cs1[xi[x + 1 + r + r]] - cs1[xi[x]];
                s2 += cs2[xi[x + 1 + r + r]] - cs2[xi[x]];
            }
        }
        if (y + 1 < end) {
            const int16_t* add = ROWPTR(y + 1 + r);
            const int16_t* sub = ROWPTR(y - r);
            for (int x = 0; x < rw; x++) {
                int64_t a = add[x], b = sub[x];
                cs1[x] += a - b; cs2[x] += a * a - b * b;
            }
        }
    }
#undef ROWPTR
    free(xi); free(cs1); free(cs2);
}

void adf_oracle_discontinuity(const int16_t* disp, ptrdiff_t stride, int W, int H,
                              int rx, int ry, int rw, int rh, int radius, float roll_off,
                              float* dst, int threads)
{
    rows_ctx z = { ROWS_ZERO_F32, H, W, dst, NULL, 0, NULL, 0, NULL, NULL, 0, 0, 0, 0 };
    parallel_stripes(threads, rows_stripe, &z);       /* Mat::zeros, DF.cpp:187-188 */
    if (rw <= 0 || rh <= 0) return;
    disc_ctx c = { disp, stride, W, rx, ry, rw, rh, radius, roll_off, dst };
    parallel_stripes(threads, disc_stripe, &c);
}

/* ------------------------------------------------------------------ */
/* A2+A4: confidence map, DF.cpp:197-210, :306-341                      */
/* ------------------------------------------------------------------ */
typedef struct {
    const int16_t* dL; ptrdiff_t sL; const int16_t* dR; ptrdiff_t sR;
    int W, H, lx, lw, rx, rw, thresh; const float* cR; float* conf;
} lrc_ctx;

static void lrc_stripe(int s, int n, void* vctx)
{
    lrc_ctx* c = (lrc_ctx*)vctx;
    int sz = stripe_size(c->H, n);
    int start = imin(s * sz, c->H), end = imin((s + 1) * sz, c->H);
    int j_end = c->lx + c->lw, right_end = c->rx + c->rw;
    for (int i = start; i < end; i++) {              /* all rows, DF.cpp:319 */
        const int16_t* rl = (const int16_t*)((const char*)c->dL + (ptrdiff_t)i * c->sL);
        const int16_t* rr = (const int16_t*)((const char*)c->dR + (ptrdiff_t)i * c->sR);
        const float* crr = c->cR + (size_t)i * c->W;
        float* dst = c->conf + (size_t)i * c->W;     /* in place on the left map, DF.cpp:206 */
        for (int j = c->lx; j < j_end; j++) {
            int right_idx = j - (rl[j] >> 4);        /* DF.cpp:331 */
            if (right_idx >= c->rx && right_idx < right_end) {
                if (abs(rl[j] + rr[right_idx]) < c->thresh) {
                    float a = dst[j], b = crr[right_idx];
                    dst[j] = b < a ? b : a;          /* std::min */
                } else
                    dst[j] = 0.0f;
            }
        }
        for (int j = 0; j < c->W; j++) dst[j] = 255.0f * dst[j];   /* DF.cpp:209 */
    }
}

void adf_oracle_confidence(const int16_t* dispL, ptrdiff_t strideL, const int16_t* dispR,
                           ptrdiff_t strideR, int W, int H, int rx, int ry, int rw, int rh,
                           int radius, int lrc_thresh, float resize_factor, float* conf,
                           int threads)
{
    /* right ROI mirrors the left one, DF.cpp:202-203 */
    int rrx = W - (rx + rw);
    float roll_off = 0.001f / (resize_factor * resize_factor);      /* DF.cpp:156,359 */
    float* cR = (float*)malloc(sizeof(float) * (size_t)W * H);
    adf_oracle_discontinuity(dispL, strideL, W, H, rx, ry, rw, rh, radius, roll_off, conf, threads);
    adf_oracle_discontinuity(dispR, strideR, W, H, rrx, ry, rw, rh, radius, roll_off, cR, threads);
    lrc_ctx c = { dispL, strideL, dispR, strideR, W, H, rx, rw, rrx, rw,
                  (int)(resize_factor * lrc_thresh) /* DF.cpp:318 */, cR, conf };
    parallel_stripes(threads, lrc_stripe, &c);
    free(cR);
}

/* ------------------------------------------------------------------ */
/* A1: DisparityWLSFilterImpl::filter, DF.cpp:219-298 (same-size case)   */
/* ------------------------------------------------------------------ */
int adf_oracle_wls_filter(const adf_oracle_params* p, const int16_t* dispL, ptrdiff_t strideL,
                          const uint8_t* guide, ptrdiff_t strideG, int gch, int W, int H,
                          const int16_t* dispR, ptrdiff_t strideR, int rx, int ry, int rw, int rh,
                          int16_t* out, ptrdiff_t strideO, float* conf_out)
{
    if (!p || !dispL || !guide || !out || W <= 0 || H <= 0) return 1;      /* DF.cpp:221-222 */
    if (gch != 1 && gch != 3) return 1;
    if (rw <= 0 || rh <= 0 || rx < 0 || ry < 0 || rx + rw > W || ry + rh > H) return 2;
    if (p->use_confidence && !dispR) return 1;                             /* DF.cpp:262 */
    const size_t P = (size_t)rw * rh;
    const int16_t fill = (int16_t)(16 * (0 - 1));  /* min_disp forced to 0: DF.cpp:149,254,284 */
    {
        rows_ctx fc = { ROWS_FILL_I16, H, W, NULL, NULL, fill, out, strideO, NULL, NULL, 0, 0, 0, 0 };
        parallel_stripes(p->threads, rows_stripe, &fc);
    }
    const uint8_t* groi = guide + (ptrdiff_t)ry * strideG + (ptrdiff_t)rx * gch;
    int rc;
    if (!p->use_confidence) {                      /* DF.cpp:235-259 */
        int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * P);
        if (!tmp) return 4;
        for (int i = 0; i < rh; i++)
            memcpy(tmp + (size_t)i * rw,
                   (const int16_t*)((const char*)dispL + (ptrdiff_t)(ry + i) * strideL) + rx,
                   sizeof(int16_t) * (size_t)rw);
        rc = adf_oracle_fgs_filter(groi, strideG, gch, rw, rh, tmp, tmp, ADF_DEPTH_16S, 1,
                                   p->lambda, p->sigma_color, p->lambda_attenuation, p->num_iter,
                                   p->order, p->threads);          /* DF.cpp:257 */
        if (rc == 0)
            for (int i = 0; i < rh; i++)                                   /* DF.cpp:258 */
                memcpy((int16_t*)((char*)out + (ptrdiff_t)(ry + i) * strideO) + rx,
                       tmp + (size_t)i * rw, sizeof(int16_t) * (size_t)rw);
        free(tmp);
        if (conf_out) memset(conf_out, 0, sizeof(float) * (size_t)W * H);
        return rc;
    }
    /* DF.cpp:260-297 */
    float* conf = conf_out ? conf_out : (float*)malloc(sizeof(float) * (size_t)W * H);
    float* planes = (float*)malloc(sizeof(float) * 2 * P);
    if (!conf || !planes) { if (!conf_out) free(conf); free(planes); return 4; }
    adf_oracle_confidence(dispL, strideL, dispR, strideR, W, H, rx, ry, rw, rh, p->disc_radius,
                          p->lrc_thresh, 1.0f, conf, p->threads);          /* DF.cpp:265 */
    {                                                                      /* DF.cpp:286-290 */
        rows_ctx pc = { ROWS_PROLOGUE, rh, rw, planes, planes + P, 0, NULL, 0,
                        (const int16_t*)((const char*)dispL + (ptrdiff_t)ry * strideL) + rx,
                        conf + (size_t)ry * W + rx, strideL, W, 0, 0 };
        parallel_stripes(p->threads, rows_stripe, &pc);
    }
    rc = adf_oracle_fgs_planes(groi, strideG, gch, rw, rh, planes, 2, p->lambda, p->sigma_color,
                               p->lambda_attenuation, p->num_iter, p->order, p->threads); /* :292-294 */
    if (rc == 0) {                                                         /* DF.cpp:295-296 */
        rows_ctx ec = { ROWS_EPILOGUE, rh, rw, planes, planes + P, 0,
                        (int16_t*)((char*)out + (ptrdiff_t)ry * strideO) + rx, strideO, NULL, NULL, 0, 0, 0, 0 };
        parallel_stripes(p->threads, rows_stripe, &ec);
    }
    if (!conf_out) free(conf);
    free(planes);
    return rc;
}

/* ------------------------------------------------------------------ */
/* N3: evaluation utilities, DF.cpp:460-556                             */
/* ------------------------------------------------------------------ */
#define ADF_UNKNOWN_DISPARITY 16320 /* DF.cpp:460 */

double adf_oracle_compute_mse(const int16_t* gt, const int16_t* src, int W, int H, int rx, int ry, int rw, int rh)
{
    (void)H;
    double res = 0; long long cnt = 0;
    for (int i = 0; i < rh; i++)
        for (int j = 0; j < rw; j++) {
            int g = gt[(size_t)(ry + i) * W + rx + j], s = src[(size_t)(ry + i) * W + rx + j];
            if (g != ADF_UNKNOWN_DISPARITY) {                    /* DF.cpp:507 */
                long long d = (long long)g - s;                  /* the reference squares in int; 64 bits avoid its overflow */
                res += (double)(d * d);
                cnt++;
            }
        }
    return res / ((double)cnt * 256.0);                          /* DF.cpp:515 */
}

double adf_oracle_bad_pixel_percent(const int16_t* gt, const int16_t* src, int W, int H, int rx, int ry, int rw, int rh, int thresh)
{
    (void)H;
    long long bad = 0, cnt = 0;
    for (int i = 0; i < rh; i++)
        for (int j = 0; j < rw; j++) {
            int g = gt[(size_t)(ry + i) * W + rx + j], s = src[(size_t)(ry + i) * W + rx + j];
            if (g != ADF_UNKNOWN_DISPARITY) {
                if (abs(g - s) >= thresh) bad++;                 /* DF.cpp:531 */
                cnt++;
            }
        }
    return (100.0 * (double)bad) / (double)cnt;                  /* DF.cpp:538 */
}

void adf_oracle_disparity_vis(const int16_t* src, uint8_t* dst, int W, int H, double scale)
{
    for (size_t k = 0; k < (size_t)W * H; k++) {
        if (src[k] == ADF_UNKNOWN_DISPARITY) dst[k] = 0;         /* DF.cpp:551-552 */
        else {
            double t = scale * src[k] / 16.0;                    /* saturate_cast<uchar>(double) = cvRound + clamp */
            long r = (t >= -2147483648.0 && t < 2147483648.0) ? lrint(t) : (long)INT_MIN;
            dst[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
}

/* ------------------------------------------------------------------ */
/* N1: down-scaled disparity path, DF.cpp:224-227, :239-247, :268-277   */
/* ------------------------------------------------------------------ */
/* cv::resize(..., INTER_LINEAR) for CV_16SC1 / CV_32FC1 as OpenCV 3.x's imgwarp.cpp computes it
 * (un-vendored, version unpinned => "parity unpinned"; restated from the published algorithm):
 *   fx = (float)((dx+0.5)*scale_x - 0.5), sx = floor(fx), fx -= sx; sx<0 -> (0, fx=0);
 *   sx >= sw-1 -> (sw-1, fx=0); alpha = (1.f-fx, fx) as float; rows likewise but with CLAMPED
 *   source rows instead of zeroed weights; horizontal pass S[sx]*a0 + S[sx+1]*a1 in float (exact
 *   S[sx] where sx+1 would leave the row), vertical pass r0*b0 + r1*b1 in float, then
 *   saturate_cast for CV_16S.  No fused multiply-add. */
typedef struct { int s0; float a0, a1; int interp; } lin_tab;

static void lin_table(int ssize, int dsize, lin_tab* t)
{
    double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        t[d].s0 = s; t[d].a0 = 1.f - f; t[d].a1 = f;
        t[d].interp = (s + 1 < ssize);                  /* dx < xmax */
    }
}

static void resize_linear_f(const void* src, int is16, int sw, int sh, ptrdiff_t sstride,
                            void* dst, int dw, int dh, ptrdiff_t dstride, float post_scale)
{
    lin_tab* tx = (lin_tab*)malloc(sizeof(lin_tab) * (size_t)dw);
    lin_table(sw, dw, tx);
    double scale_y = (double)sh / dh;
    float* r0 = (float*)malloc(sizeof(float) * (size_t)dw);
    float* r1 = (float*)malloc(sizeof(float) * (size_t)dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const float b0 = 1.f - fy, b1 = fy;
        int y0 = sy < 0 ? 0 : sy > sh - 1 ? sh - 1 : sy;
        int y1 = sy + 1 < 0 ? 0 : sy + 1 > sh - 1 ? sh - 1 : sy + 1;
        for (int k = 0; k < 2; k++) {
            const char* row = (const char*)src + (ptrdiff_t)(k ? y1 : y0) * sstride;
            float* r = k ? r1 : r0;
            for (int dx = 0; dx < dw; dx++) {
                const int s = tx[dx].s0;
                const float v0 = is16 ? (float)((const int16_t*)row)[s] : ((const float*)row)[s];
                if (tx[dx].interp) {
                    const float v1 = is16 ? (float)((const int16_t*)row)[s + 1] : ((const float*)row)[s + 1];
                    r[dx] = v0 * tx[dx].a0 + v1 * tx[dx].a1;
                } else
                    r[dx] = v0;
            }
        }
        char* drow = (char*)dst + (ptrdiff_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            const float v = r0[dx] * b0 + r1[dx] * b1;
            if (is16) {
                int16_t q = adf_oracle_sat16(v);
                /* disp_full_size*x_ratio (DF.cpp:244,273): convertTo with a float scale, saturating */
                ((int16_t*)drow)[dx] = post_scale == 1.0f ? q : adf_oracle_sat16((float)q * post_scale + 0.0f);
            } else
                ((float*)drow)[dx] = v;
        }
    }
    free(r0); free(r1); free(tx);
}

void adf_oracle_resize_linear_16s(const int16_t* src, int sw, int sh, int16_t* dst, int dw, int dh, float post_scale)
{
    resize_linear_f(src, 1, sw, sh, (ptrdiff_t)sw * 2, dst, dw, dh, (ptrdiff_t)dw * 2, post_scale);
}

void adf_oracle_resize_linear_32f(const float* src, int sw, int sh, float* dst, int dw, int dh)
{
    resize_linear_f(src, 0, sw, sh, (ptrdiff_t)sw * 4, dst, dw, dh, (ptrdiff_t)dw * 4, 1.0f);
}

/* DF.cpp:219-298 with disparity maps (dW x dH) smaller than the view (W x H).  ROI is given in
 * DISPARITY-MAP coordinates (DF.cpp:228-233); conf_out is the view-sized confidence map. */
int adf_oracle_wls_filter_scaled(const adf_oracle_params* p, const int16_t* dispL, const int16_t* dispR,
                                 int dW, int dH, const uint8_t* guide, int gch, int W, int H,
                                 int rx, int ry, int rw, int rh, int16_t* out, float* conf_out)
{
    if (!p || !dispL || !guide || !out || dW <= 0 || dH <= 0 || W <= 0 || H <= 0) return 1;
    if (rw <= 0 || rh <= 0 || rx < 0 || ry < 0 || rx + rw > dW || ry + rh > dH) return 2;
    const float resize_factor = dW / (float)W;                             /* DF.cpp:225 */
    const float x_ratio = W / (float)dW, y_ratio = H / (float)dH;          /* DF.cpp:241-242,270-271 */
    const int hx = (int)(rx * x_ratio), hy = (int)(ry * y_ratio);           /* DF.cpp:245-246,275-276 */
    const int hw = (int)(rw * x_ratio), hh = (int)(rh * y_ratio);
    if (hw <= 0 || hh <= 0 || hx + hw > W || hy + hh > H) return 2;
    int16_t* dhi = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * H);
    float* chi = conf_out ? conf_out : (float*)malloc(sizeof(float) * (size_t)W * H);
    if (!dhi || !chi) { free(dhi); if (!conf_out) free(chi); return 4; }
    adf_oracle_resize_linear_16s(dispL, dW, dH, dhi, W, H, x_ratio);      /* DF.cpp:243-244,272-273 */
    const size_t P = (size_t)hw * hh;
    const uint8_t* groi = guide + ((size_t)hy * W + hx) * gch;
    for (size_t k = 0; k < (size_t)W * H; k++) out[k] = (int16_t)-16;      /* DF.cpp:254,284 */
    int rc;
    if (!p->use_confidence) {
        int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * P);
        for (int i = 0; i < hh; i++) memcpy(tmp + (size_t)i * hw, dhi + (size_t)(hy + i) * W + hx, sizeof(int16_t) * (size_t)hw);
        rc = adf_oracle_fgs_filter(groi, (ptrdiff_t)W * gch, gch, hw, hh, tmp, tmp, ADF_DEPTH_16S, 1, p->lambda,
                                   p->sigma_color, p->lambda_attenuation, p->num_iter, p->order, p->threads);
        if (rc == 0)
            for (int i = 0; i < hh; i++) memcpy(out + (size_t)(hy + i) * W + hx, tmp + (size_t)i * hw, sizeof(int16_t) * (size_t)hw);
        free(tmp);
        if (conf_out) memset(conf_out, 0, sizeof(float) * (size_t)W * H);
    } else {
        if (!dispR) { free(dhi); if (!conf_out) free(chi); return 1; }
        float* clo = (float*)malloc(sizeof(float) * (size_t)dW * dH);
        float* planes = (float*)malloc(sizeof(float) * 2 * P);
        adf_oracle_confidence(dispL, (ptrdiff_t)dW * 2, dispR, (ptrdiff_t)dW * 2, dW, dH, rx, ry, rw, rh, p->disc_radius,
                              p->lrc_thresh, resize_factor, clo, p->threads);          /* DF.cpp:265 */
        adf_oracle_resize_linear_32f(clo, dW, dH, chi, W, H);                         /* DF.cpp:274 */
        for (int i = 0; i < hh; i++)
            for (int j = 0; j < hw; j++) {
                const float c = chi[(size_t)(hy + i) * W + hx + j];
                planes[(size_t)i * hw + j] = c * (float)dhi[(size_t)(hy + i) * W + hx + j];
                planes[P + (size_t)i * hw + j] = c;
            }
        rc = adf_oracle_fgs_planes(groi, (ptrdiff_t)W * gch, gch, hw, hh, planes, 2, p->lambda, p->sigma_color,
                                   p->lambda_attenuation, p->num_iter, p->order, p->threads);
        if (rc == 0)
            for (int i = 0; i < hh; i++)
                for (int j = 0; j < hw; j++) {
                    const float rcp = 1.0f / (planes[P + (size_t)i * hw + j] + ADF_EPS);
                    out[(size_t)(hy + i) * W + hx + j] = adf_oracle_sat16(planes[(size_t)i * hw + j] * rcp);
                }
        free(clo); free(planes);
    }
    free(dhi);
    if (!conf_out) free(chi);
    return rc;
}
