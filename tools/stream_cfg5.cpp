// BASELINE config 5 as a stream, issued from C++ (the reference's host language) straight through the C-ABI: what
// tools/stream_cfg5.py measures from Python, without the interpreter between the calls.  Frames of 1242x375 (8UC1
// guide, ROI (128,0,1114,375), radius 2, 3 iterations), ONE frame per adf_wls_filter_device call, dealt round-robin to K
// handles on K streams.  Modes: "calls" (the plain call), "graphs" (the call captured once per handle, one
// hipGraphLaunch per frame, the handle's input buffers written by the producer).  Prints sustained Mpixels/s (wall
// clock over all frames), the host's issue time per frame, and for K = 1 the device latency per frame.
//
//   hipcc -O2 -std=c++17 tools/stream_cfg5.cpp -Iinclude -Laddingdisparityfiltering_amd -ladf_wls \
//         -Wl,-rpath,'$ORIGIN/../addingdisparityfiltering_amd' -o tools/stream_cfg5_cpp
//   tools/stream_cfg5_cpp [frames [streams_created [with_priority]]]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "adf_wls.h"

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define ADF(x)                                                                           \
    do {                                                                                 \
        int r_ = (x);                                                                    \
        if (r_ != ADF_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, adf_last_error()); return 3; } \
    } while (0)

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 1024;
    const int W = 1242, H = 375;
    const adf_rect roi = {128, 0, 1114, 375};
    const size_t px = (size_t)W * H;

    // synthetic frames: a smooth guide with edges, disparities in 0..127 (x16) with an occluded band, right map = left
    // shifted -- the values only have to be plausible, the checked runs are tools/stream_cfg5.py's
    std::vector<uint8_t> view(px * frames);
    std::vector<int16_t> dl(px * frames), dr(px * frames);
    std::mt19937 rng(5);
    for (int f = 0; f < frames; f++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t i = (size_t)f * px + (size_t)y * W + x;
                const int d = 16 * (20 + ((x / 97 + y / 61 + f) % 5) * 12) + (int)(rng() % 9) - 4;
                view[i] = (uint8_t)((x / 97 + y / 61 + f) % 5 * 40 + 30 + rng() % 7);
                dl[i] = (int16_t)((rng() % 50 == 0) ? -16 : d);
                dr[i] = (int16_t)(-d);
            }
    uint8_t* d_view; int16_t *d_dl, *d_dr, *d_out;
    CHECK(hipMalloc(&d_view, px * frames)); CHECK(hipMalloc(&d_dl, 2 * px * frames)); CHECK(hipMalloc(&d_dr, 2 * px * frames));
    CHECK(hipMalloc(&d_out, 2 * px * frames));
    CHECK(hipMemcpy(d_view, view.data(), px * frames, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dl, dl.data(), 2 * px * frames, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_dr, dr.data(), 2 * px * frames, hipMemcpyHostToDevice));

    printf("# config 5 as a stream from C++: %d frames of %dx%d, ROI (128,0,1114,375), radius 2, 3 iterations, one frame per call\n", frames, W, H);
    printf("# K | mode   | sustained Mpx/s (best of 3) | frames/s | host issue us/frame | device latency us/frame (K=1: wall/frames)\n");
    // the streams live for the whole run (a pipeline creates its streams once)
    const int S = argc > 2 ? std::max(1, std::min(32, atoi(argv[2]))) : 8;
    const int with_priority = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<hipStream_t> st(S);
    for (int k = 0; k < S; k++) {
        if (with_priority == 2) CHECK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, (k % 3) - 1));   // three priority levels, cycling
        else if (with_priority) CHECK(hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, 0));
        else CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    printf("# %d streams created up front (%s)\n", S, with_priority == 2 ? "hipStreamCreateWithPriority, priorities -1 / 0 / 1 cycling" : with_priority ? "hipStreamCreateWithPriority(.., 0)" : "hipStreamCreateWithFlags");
    for (int K : {1, 2, 4, 8}) {
        if (K > S) break;
        std::vector<adf_wls_t*> h(K);
        for (int k = 0; k < K; k++) {
            ADF(adf_wls_create(&h[k], 1, 0, 0, 0, 0, 0));
            ADF(adf_wls_set_lambda(h[k], 8000.0)); ADF(adf_wls_set_sigma_color(h[k], 1.5));
            ADF(adf_wls_set_depth_discontinuity_radius(h[k], 2)); ADF(adf_wls_set_fgs_params(h[k], 0.25, 3));
        }
        auto call = [&](int k, size_t i) {
            return adf_wls_filter_device(h[k], 1, d_dl + i * px, 2 * W, 0, d_view + i * px, W, 0, 1, W, H, d_out + i * px, 2 * W, 0,
                                         d_dr + i * px, 2 * W, 0, &roi, st[k]);
        };
        for (int k = 0; k < K; k++) for (int r = 0; r < 3; r++) ADF(call(k, k));
        CHECK(hipDeviceSynchronize());
        for (int mode = 0; mode < 2; mode++) {
            std::vector<hipGraphExec_t> ge(K, nullptr);
            if (mode == 1)
                for (int k = 0; k < K; k++) {
                    hipGraph_t g;
                    CHECK(hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal));
                    ADF(call(k, k));
                    CHECK(hipStreamEndCapture(st[k], &g));
                    CHECK(hipGraphInstantiate(&ge[k], g, nullptr, nullptr, 0));
                    CHECK(hipGraphDestroy(g));
                }
            double best = 0, best_issue = 0, best_wall = 0;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipDeviceSynchronize());
                const double t0 = now_s();
                for (int i = 0; i < frames; i++) {
                    const int k = i % K;
                    if (mode == 0) ADF(call(k, i));
                    else CHECK(hipGraphLaunch(ge[k], st[k]));
                }
                const double t1 = now_s();
                CHECK(hipDeviceSynchronize());
                const double t2 = now_s();
                const double rate = frames * (double)px / (t2 - t0) / 1e6;
                if (rate > best) { best = rate; best_issue = (t1 - t0) / frames * 1e6; best_wall = (t2 - t0) / frames * 1e6; }
            }
            printf("  %d | %-6s | %10.1f | %9.1f | %6.1f | %6.1f\n", K, mode ? "graphs" : "calls", best, best * 1e6 / px, best_issue, best_wall);
            for (auto e : ge) if (e) CHECK(hipGraphExecDestroy(e));
        }
        for (int k = 0; k < K; k++) adf_wls_destroy(h[k]);
    }
    for (int k = 0; k < S; k++) CHECK(hipStreamDestroy(st[k]));
    return 0;
}
