// Host time per hook call through the C ABI (no Python in the way): N calls of mi355cv_gaussianBlurBinomial / mi355cv_cvtBGRtoGray / mi355cv_threshold on device-resident
// 4K frames, (a) enqueue only (mi355cv_setAsync(1): how long the calling thread is busy per call), (b) synchronous calls (the hook's default contract: results visible
// at return).  Build on the GPU box:  hipcc -O2 -I include tools/ubench/call_latency.cpp -L opencv_amd -lmi355cv -Wl,-rpath,$PWD/opencv_amd -o /tmp/call_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include "mi355cv.h"
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int W = 3840, H = 2160, NF = 16, N = 2000;
    unsigned char *src, *dst, *bgr;
    if (hipMalloc(&src, (size_t)W * H * NF) != hipSuccess || hipMalloc(&dst, (size_t)W * H * NF) != hipSuccess || hipMalloc(&bgr, (size_t)W * H * 3 * 4) != hipSuccess) return 1;
    (void)hipMemset(src, 7, (size_t)W * H * NF); (void)hipMemset(bgr, 9, (size_t)W * H * 3 * 4);
    if (mi355cv_init(-1) != 0) { printf("no device\n"); return 1; }
    auto gauss = [&](int i) { return mi355cv_gaussianBlurBinomial(src + (size_t)(i % NF) * W * H, W, dst + (size_t)(i % NF) * W * H, W, W, H, 0, 1, 0, 0, 0, 0, 5, 4); };
    auto gray = [&](int i) { return mi355cv_cvtBGRtoGray(bgr + (size_t)(i % 4) * W * H * 3, (size_t)W * 3, dst + (size_t)(i % NF) * W * H, W, W, H, 0, 3, false); };
    auto thr = [&](int i) { return mi355cv_threshold(src + (size_t)(i % NF) * W * H, W, dst + (size_t)(i % NF) * W * H, W, W, H, 0, 1, 127.0, 255.0, 0); };
    struct { const char* name; int (*fn)(void*, int); void* ctx; } dummy; (void)dummy;
    hipStream_t user = nullptr;
    (void)hipStreamCreateWithFlags(&user, hipStreamNonBlocking);
    // mode 2 (round 5): the DEFAULT contract (no mi355cv_setAsync) with device-resident images on a stream the caller bound with mi355cv_setStream: the hook returns after
    // the enqueue (nothing host-side can see the image before the caller synchronises its own stream); mode 1: explicit async; mode 0: default contract on the library's own
    // stream = synchronous
    for (int mode = 2; mode >= 0; mode--) {
        mi355cv_setAsync(mode == 1);
        if (mode == 2) mi355cv_setStream(user); else mi355cv_resetStream();
        for (int which = 0; which < 3; which++) {
            int rc = 0;
            for (int i = 0; i < 50; i++) rc |= which == 0 ? gauss(i) : which == 1 ? gray(i) : thr(i);
            mi355cv_synchronize(); (void)hipStreamSynchronize(user);
            const double t0 = now();
            for (int i = 0; i < N; i++) rc |= which == 0 ? gauss(i) : which == 1 ? gray(i) : thr(i);
            const double t1 = now();
            mi355cv_synchronize(); (void)hipStreamSynchronize(user);
            const double t2 = now();
            printf("%-26s %s: %6.2f us host time per call, %6.2f us per call incl. the final drain (rc %d)\n", which == 0 ? "gaussianBlurBinomial 5x5 4K" : which == 1 ? "cvtBGRtoGray 4K" : "threshold 4K",
                   mode == 2 ? "default, caller's stream" : mode ? "async enqueue" : "synchronous  ", (t1 - t0) / N, (t2 - t0) / N, rc);
        }
    }
    return 0;
}
