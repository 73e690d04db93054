// "One cv:: call per frame" rows through the C ABI with no Python in the way: a caller that loops over device-resident frames -- what a cv:: program
// on FrameAllocatorDevice Mats does -- with its own stream bound (mi355cv_setStream): every hook returns after the enqueue (3-4 us of host time), so the GPU runs the
// frames back to back.  Each row walks >= 2 GiB of DISTINCT frames per pass (the 256 MiB Infinity Cache cannot serve it), is timed with HIP events on the caller's stream
// over whole passes, and reports us per call and the fraction of 8 TB/s on the algorithmic bytes, plus the host's own time per call.  One JSON object per line.
// Build on the GPU box:  hipcc -O2 -I include tools/ubench/per_frame_rows.cpp -L opencv_amd -lmi355cv -Wl,-rpath,$PWD/opencv_amd -o /tmp/per_frame_rows
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#include "mi355cv.h"

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int W = 3840, H = 2160;
    const size_t PIX = (size_t)W * H;
    if (mi355cv_init(-1) != 0) { printf("{\"error\": \"no device\"}\n"); return 1; }
    hipStream_t user = nullptr;
    if (hipStreamCreateWithFlags(&user, hipStreamNonBlocking) != hipSuccess) return 1;
    mi355cv_setStream(user);                                                    // the default contract: device images on a caller-bound stream return after the enqueue
    const int NF8 = 144, NF32 = 40;                                             // 144 x 8.3 MB x 2 = 2.4 GB per pass; 40 x 33 MB x 2 = 2.65 GB
    unsigned char *s8 = nullptr, *d8 = nullptr; float *s32 = nullptr, *d32 = nullptr; short* d16 = nullptr;
    if (hipMalloc(&s8, PIX * NF8) != hipSuccess || hipMalloc(&d8, PIX * NF8) != hipSuccess || hipMalloc(&s32, PIX * 4 * NF32) != hipSuccess ||
        hipMalloc(&d32, PIX * 4 * NF32) != hipSuccess || hipMalloc(&d16, PIX * 2 * 96) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    {   // any non-constant contents (the kernels' time does not depend on the values; a constant image would flatter the DVFS)
        std::vector<unsigned char> h(PIX); unsigned x = 809564u;
        for (size_t i = 0; i < PIX; i++) { x = x * 1664525u + 1013904223u; h[i] = (unsigned char)(x >> 24); }
        for (int f = 0; f < NF8; f++) (void)hipMemcpy(s8 + PIX * f, h.data(), PIX, hipMemcpyHostToDevice);
        std::vector<float> hf(PIX);
        for (size_t i = 0; i < PIX; i++) { x = x * 1664525u + 1013904223u; hf[i] = (float)(x >> 8) * (1.0f / 16777216.0f); }
        for (int f = 0; f < NF32; f++) (void)hipMemcpy(s32 + PIX * f, hf.data(), PIX * 4, hipMemcpyHostToDevice);
    }
    const float k3[9] = {0, -1, 0, -1, 5, -1, 0, -1, 0};
    float g5[5];
    { double kd[5]; mi355cv_getGaussianKernel(5, 1.2, kd); for (int i = 0; i < 5; i++) g5[i] = (float)kd[i]; }
    cvhalFilter2D *f8 = nullptr, *f32 = nullptr, *sp32 = nullptr;
    int rc0 = mi355cv_filterInit(&f8, (unsigned char*)k3, 12, MI355CV_32F, 3, 3, W, H, MI355CV_8U, MI355CV_8U, 4, 0.0, -1, -1, false, false);
    rc0 |= mi355cv_filterInit(&f32, (unsigned char*)k3, 12, MI355CV_32F, 3, 3, W, H, MI355CV_32F, MI355CV_32F, 4, 0.0, -1, -1, false, false);
    rc0 |= mi355cv_sepFilterInit(&sp32, MI355CV_32F, MI355CV_32F, MI355CV_32F, (unsigned char*)g5, 5, (unsigned char*)g5, 5, -1, -1, 0.0, 4);
    if (rc0) { printf("{\"error\": \"filterInit %d: %s\"}\n", rc0, mi355cv_lastError()); return 1; }
    struct Row { const char* name; int frames; size_t bytes; std::function<int(int)> fn; };
    const std::vector<Row> rows = {
        {"a1 GaussianBlur 5x5 4K 8UC1", NF8, 2 * PIX, [&](int i) { return mi355cv_gaussianBlurBinomial(s8 + PIX * i, W, d8 + PIX * i, W, W, H, MI355CV_8U, 1, 0, 0, 0, 0, 5, 4); }},
        {"a3 filter2D 3x3 4K 8UC1", NF8, 2 * PIX, [&](int i) { return mi355cv_filter(f8, s8 + PIX * i, W, d8 + PIX * i, W, W, H, W, H, 0, 0); }},
        {"a4 Sobel dx 3x3 4K 8U->16S", 96, 3 * PIX, [&](int i) { return mi355cv_sobel(s8 + PIX * i, W, (unsigned char*)(d16 + PIX * i), (size_t)W * 2, W, H, MI355CV_8U, MI355CV_16S, 1, 0, 0, 0, 0, 1, 0, 3, 1.0, 0.0, 4); }},
        {"a5 boxFilter 5x5 4K 8U", NF8, 2 * PIX, [&](int i) { return mi355cv_boxFilter(s8 + PIX * i, W, d8 + PIX * i, W, W, H, MI355CV_8U, MI355CV_8U, 1, 0, 0, 0, 0, 5, 5, -1, -1, true, 4); }},
        {"f1 threshold BINARY 4K 8U", NF8, 2 * PIX, [&](int i) { return mi355cv_threshold(s8 + PIX * i, W, d8 + PIX * i, W, W, H, MI355CV_8U, 1, 127.0, 255.0, 0); }},
        {"a1 GaussianBlur 5x5 sigma 1.2 4K 32FC1 (sepFilter2D)", NF32, 8 * PIX, [&](int i) { return mi355cv_sepFilter(sp32, (unsigned char*)(s32 + PIX * i), (size_t)W * 4, (unsigned char*)(d32 + PIX * i), (size_t)W * 4, W, H, W, H, 0, 0); }},
        {"a4 Sobel dx 3x3 4K 32F->32F", NF32, 8 * PIX, [&](int i) { return mi355cv_sobel((unsigned char*)(s32 + PIX * i), (size_t)W * 4, (unsigned char*)(d32 + PIX * i), (size_t)W * 4, W, H, MI355CV_32F, MI355CV_32F, 1, 0, 0, 0, 0, 1, 0, 3, 1.0, 0.0, 4); }},
        {"a5 boxFilter 5x5 4K 32FC1", NF32, 8 * PIX, [&](int i) { return mi355cv_boxFilter((unsigned char*)(s32 + PIX * i), (size_t)W * 4, (unsigned char*)(d32 + PIX * i), (size_t)W * 4, W, H, MI355CV_32F, MI355CV_32F, 1, 0, 0, 0, 0, 5, 5, -1, -1, true, 4); }},
        {"a3 filter2D 3x3 4K 32FC1", NF32, 8 * PIX, [&](int i) { return mi355cv_filter(f32, (unsigned char*)(s32 + PIX * i), (size_t)W * 4, (unsigned char*)(d32 + PIX * i), (size_t)W * 4, W, H, W, H, 0, 0); }},
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (const Row& r : rows) {
        int rc = 0;
        for (int p = 0; p < 2; p++) for (int i = 0; i < r.frames; i++) rc |= r.fn(i);        // warm-up: two passes
        (void)hipStreamSynchronize(user);
        const int passes = 6;
        (void)hipEventRecord(e0, user);
        const double t0 = now();
        for (int p = 0; p < passes; p++) for (int i = 0; i < r.frames; i++) rc |= r.fn(i);
        const double t1 = now();
        (void)hipEventRecord(e1, user);
        (void)hipStreamSynchronize(user);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double calls = (double)passes * r.frames, us = ms * 1e3 / calls;
        printf("{\"config\": \"%s (one C-ABI call per frame on the caller's stream)\", \"kind\": \"per-frame calls, C ABI\", \"frames\": %d, \"us_per_call\": %.2f, \"host_us_per_call\": %.2f, "
               "\"working_set_GB\": %.3f, \"frac\": %.4f, \"rc\": %d, \"kernel\": \"%s\"}\n",
               r.name, r.frames, us, (t1 - t0) / calls, (double)r.bytes * r.frames / 1e9, (double)r.bytes / (us * 1e-6) / 8e12, rc, mi355cv_lastKernel());
        fflush(stdout);
    }
    mi355cv_filterFree(f8); mi355cv_filterFree(f32); mi355cv_sepFilterFree(sp32);
    return 0;
}
