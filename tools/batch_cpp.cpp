// Does the weight kernel (forked to the handle's side stream) really run BESIDE the confidence kernel when the caller is
// a plain C++ program with a stream of its own?  HIP maps streams of one priority onto a small pool of hardware queues and
// two streams that share a queue run one after the other.  16 pairs of 3840x2160 per call (constant images: timing only),
// on the null stream, on streams from hipStreamCreateWithFlags, and with ADF_NO_OVERLAP=1 (everything on one stream) for
// comparison.     tools/batch_cpp [pairs]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "adf_wls.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ADF(x) do { int r_ = (x); if (r_ != ADF_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, adf_last_error()); return 3; } } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const int W = 3840, H = 2160;
    const adf_rect roi = {256, 0, 3584, 2160};
    const size_t px = (size_t)W * H;
    uint8_t* view; int16_t *dl, *dr, *out;
    CHECK(hipMalloc(&view, px * 3 * n)); CHECK(hipMalloc(&dl, 2 * px * n)); CHECK(hipMalloc(&dr, 2 * px * n)); CHECK(hipMalloc(&out, 2 * px * n));
    CHECK(hipMemset(view, 0x60, px * 3 * n));
    {   // left map 20 px (x16), right map -20 px: consistent everywhere
        std::vector<int16_t> a(px, 320), b(px, -320);
        for (int k = 0; k < n; k++) { CHECK(hipMemcpy(dl + k * px, a.data(), 2 * px, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dr + k * px, b.data(), 2 * px, hipMemcpyHostToDevice)); }
    }
    std::vector<hipStream_t> streams = {nullptr};
    for (int k = 0; k < 5; k++) { hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); streams.push_back(s); }
    printf("# %d pairs of %dx%d per call, 3-channel guide, ROI (256,0,3584,2160), radius 2; ms per call (best of 3 x 5 calls)\n", n, W, H);
    for (size_t si = 0; si < streams.size(); si++) {
        adf_wls_t* h;
        ADF(adf_wls_create(&h, 1, 0, 0, 0, 0, 0));
        ADF(adf_wls_set_sigma_color(h, 1.5)); ADF(adf_wls_set_depth_discontinuity_radius(h, 2));
        auto call = [&]() {
            return adf_wls_filter_device(h, n, dl, 2 * W, 2 * px, view, 3 * W, 3 * px, 3, W, H, out, 2 * W, 2 * px, dr, 2 * W, 2 * px, &roi, streams[si]);
        };
        for (int r = 0; r < 2; r++) ADF(call());
        CHECK(hipDeviceSynchronize());
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now_s();
            for (int r = 0; r < 5; r++) ADF(call());
            CHECK(hipDeviceSynchronize());
            best = std::min(best, (now_s() - t0) / 5 * 1e3);
        }
        printf("  stream %zu (%s): %.3f ms\n", si, si == 0 ? "null stream" : "hipStreamCreateWithFlags", best);
        adf_wls_destroy(h);
    }
    return 0;
}
