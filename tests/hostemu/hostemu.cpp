// tests/hostemu -- TEST INFRASTRUCTURE ONLY.  Host build of per-pixel arithmetic headers that the HIP kernels include (opencv_amd/csrc/*_math.h),
// so that the CPU test-suite can check the very lines the GPU runs against the pinned restatement when no GPU is available.
// Built by tests/test_hostemu.py with g++ -ffp-contract=off (the kernels are compiled with the same setting).
#include <cstddef>
#include <cstdint>
#include "hsv_math.h"

extern "C" void emu_hsv2bgr(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange)
{
    const int bidx = swapBlue ? 2 : 0, body = (w / 32) * 32;
    const float hscale = 6.0f / (fullRange ? 255 : 180);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            int b, g, r;
            mi355_hsv2bgr_px(s[0], s[1], s[2], x < body, hscale, b, g, r);
            d[bidx] = (uint8_t)b; d[1] = (uint8_t)g; d[bidx ^ 2] = (uint8_t)r;
            if (dcn == 4) d[3] = 255;
        }
}

// BGR/RGB(A) <-> HLS, CV_8U: the lines k_bgr2hls_u8 / k_hls2bgr_u8 (color_yuv.hip) run per pixel
extern "C" void emu_bgr2hls(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int fullRange)
{
    const int bidx = swapBlue ? 2 : 0;
    const float hscale = (fullRange ? 256.f : 180.f) / 360.f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * scn;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * 3;
            float H, L, S;
            mi355_rgb2hls_px(s[bidx ^ 2] * (1.f / 255.f), s[1] * (1.f / 255.f), s[bidx] * (1.f / 255.f), hscale, mi355_hls_in_vector_body(x, w), H, L, S);
            d[0] = (uint8_t)mi355_round_sat8(H); d[1] = (uint8_t)mi355_round_sat8(L * 255.f); d[2] = (uint8_t)mi355_round_sat8(S * 255.f);
        }
}
extern "C" void emu_hls2bgr(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int fullRange)
{
    const int bidx = swapBlue ? 2 : 0;
    const float hscale = 6.f / (fullRange ? 255.f : 180.f);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* s = src + (size_t)y * sstep + (size_t)x * 3;
            uint8_t* d = dst + (size_t)y * dstep + (size_t)x * dcn;
            float B, G, R;
            mi355_hls2rgb_px((float)s[0], s[1] * (1.f / 255.f), s[2] * (1.f / 255.f), hscale, mi355_hls_in_vector_body(x, w), B, G, R);
            d[bidx] = (uint8_t)mi355_round_sat8(B * 255.f); d[1] = (uint8_t)mi355_round_sat8(G * 255.f); d[bidx ^ 2] = (uint8_t)mi355_round_sat8(R * 255.f);
            if (dcn == 4) d[3] = 255;
        }
}
