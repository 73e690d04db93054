// orb_host.h -- the host side of cv::ORB's control flow (modules/features2d/src/orb.cpp, features2d/src/keypoint.cpp): buffer layout, the two culls with
// the C++ library's own algorithms, pattern and disc tables.  Plain C++ (no HIP): orb.hip includes it, and the CPU test-suite compiles the same lines
// into tests/hostemu/liborbemu.so and checks each function against the pinned restatement.
#pragma once
#include "orb_math.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orbh {

constexpr int MAX_LEVELS = 32;
struct KP { float x, y, size, angle, response; int octave, class_id; };     // cv::KeyPoint (core/types.hpp:777), the layout of mi355cv_KeyPoint

static const int bitPattern31[1024] = {
#include "orb_pattern.inc"
};

struct Layout { int nLevels, border, bufW, bufH, pitch; orbm::Layer layer[MAX_LEVELS]; float scale[MAX_LEVELS]; };

inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvCeilD(double v) { const int i = (int)v; return i + (i < v); }
inline int cvFloorD(double v) { const int i = (int)v; return i - (i > v); }
inline float getScale(int level, int firstLevel, double scaleFactor) { return (float)std::pow(scaleFactor, (double)(level - firstLevel)); }       // orb.cpp:653

// orb.cpp:1026-1095
inline void buildLayout(Layout& L, int w, int h, int nLevels, int firstLevel, double scaleFactor, int edgeThreshold, int patchSize)
{
    const int halfPatch = patchSize / 2, descPatch = cvCeilD(halfPatch * std::sqrt(2.0));
    L.border = std::max(edgeThreshold, std::max(descPatch, 9 / 2)) + 1;
    L.nLevels = nLevels;
    const float l0inv = 1.0f / getScale(0, firstLevel, scaleFactor);
    const size_t l0w = (size_t)cvRoundF(w * l0inv), l0h = (size_t)cvRoundF(h * l0inv);
    L.bufW = (int)((l0w + L.border * 2 + 15) & ~(size_t)15);
    int level_dy = (int)l0h + L.border * 2, ox = 0, oy = 0;
    for (int level = 0; level < nLevels; level++) {
        const float scale = getScale(level, firstLevel, scaleFactor);
        L.scale[level] = scale;
        const float inv = 1.0f / scale;
        const int sw = cvRoundF(w * inv), sh = cvRoundF(h * inv);
        const int ww = sw + L.border * 2, wh = sh + L.border * 2;
        if (ox + ww > L.bufW) { ox = 0; oy += level_dy; level_dy = wh; }
        L.layer[level] = {ox + L.border, oy + L.border, sw, sh};
        ox += ww;
    }
    L.bufH = oy + level_dy;
    L.pitch = (L.bufW + 63) & ~63;
}

// the threads of the border pass of one level: dwords g0 .. g0 + ng - 1 of buffer rows row0 .. row0 + nrows - 1 cover its extended rectangle
struct BorderGrid { int g0, ng, row0, nrows; };
inline BorderGrid borderGrid(const Layout& L, int level)
{
    const orbm::Layer r = L.layer[level];
    const int x0 = r.x - L.border, x1 = r.x + r.w + L.border;
    const int g0 = x0 >> 2;
    return {g0, ((x1 + 3) >> 2) - g0, r.y - L.border, r.h + 2 * L.border};
}

// KeyPointsFilter::retainBest (keypoint.cpp:70-92) with the same two library algorithms
inline void retainBest(std::vector<KP>& k, int npoints)
{
    if (npoints < 0 || k.size() <= (size_t)npoints) return;
    if (npoints == 0) { k.clear(); return; }
    std::nth_element(k.begin(), k.begin() + npoints - 1, k.end(), [](const KP& a, const KP& b) { return a.response > b.response; });
    const float amb = k[npoints - 1].response;
    auto e = std::partition(k.begin() + npoints, k.end(), [amb](const KP& a) { return a.response >= amb; });
    k.resize(e - k.begin());
}

// The first cull of a level sees every FAST candidate inside the border (10^5 on a 4K level 0; the border test itself runs in the collect kernel): the
// same two algorithms on 8-byte (response, pixel index) records instead of 28-byte keypoints.  std::nth_element / std::partition move elements by comparison outcomes and positions only, so the survivors and their order are
// those of the keypoint vector (tests/test_hostemu.py holds both against the pinned restatement); keypoints are then built for the survivors alone.
struct Cand { float response; uint32_t idx; };
inline void retainBestCand(std::vector<Cand>& k, int npoints)
{
    if (npoints < 0 || k.size() <= (size_t)npoints) return;
    if (npoints == 0) { k.clear(); return; }
    std::nth_element(k.begin(), k.begin() + npoints - 1, k.end(), [](const Cand& a, const Cand& b) { return a.response > b.response; });
    const float amb = k[npoints - 1].response;
    auto e = std::partition(k.begin() + npoints, k.end(), [amb](const Cand& a) { return a.response >= amb; });
    k.resize(e - k.begin());
}
// KeyPointsFilter::runByImageBorder (keypoint.cpp:107-119): Rect((b, b), (w - b, h - b)).contains(Point_<int>(pt)) -- the conversion rounds
inline void runByImageBorder(std::vector<KP>& k, int w, int h, int b)
{
    if (b <= 0) return;
    if (h <= b * 2 || w <= b * 2) { k.clear(); return; }
    k.erase(std::remove_if(k.begin(), k.end(), [=](const KP& p) { const int x = cvRoundF(p.x), y = cvRoundF(p.y); return !(x >= b && x < w - b && y >= b && y < h - b); }), k.end());
}

// cv::RNG (core/operations.hpp:349-373)
struct Rng {
    uint64_t s;
    unsigned next() { s = (uint64_t)(unsigned)s * 4164903690U + (unsigned)(s >> 32); return (unsigned)s; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// the sampling pattern of a parameter set (orb.cpp:1205-1223) as signed bytes x, y; returns the byte count
inline int buildPattern(int patchSize, int wta_k, signed char* out)
{
    int pool[1024];
    if (patchSize == 31) memcpy(pool, bitPattern31, sizeof pool);
    else {                                                                              // makeRandomPattern :641-650
        Rng r{0x34985739};
        for (int i = 0; i < 512; i++) { pool[2 * i] = r.uniform(-patchSize / 2, patchSize / 2 + 1); pool[2 * i + 1] = r.uniform(-patchSize / 2, patchSize / 2 + 1); }
    }
    if (wta_k == 2) { for (int i = 0; i < 1024; i++) out[i] = (signed char)pool[i]; return 1024; }
    const int ntuples = 32 * 4;                                                         // initializeOrbPattern :352-376
    int pat[1024];
    Rng r{0x12345678};
    for (int i = 0; i < ntuples; i++)
        for (int k = 0; k < wta_k; k++)
            for (;;) {
                const int idx = r.uniform(0, 512);
                const int px = pool[2 * idx], py = pool[2 * idx + 1];
                int k1 = 0;
                for (; k1 < k; k1++) if (pat[2 * (wta_k * i + k1)] == px && pat[2 * (wta_k * i + k1) + 1] == py) break;
                if (k1 == k) { pat[2 * (wta_k * i + k)] = px; pat[2 * (wta_k * i + k) + 1] = py; break; }
            }
    const int nb = ntuples * wta_k * 2;
    for (int i = 0; i < nb; i++) out[i] = (signed char)pat[i];
    return nb;
}

// last column of every row of the circular patch (orb.cpp:806-823)
inline void buildUmax(int half, std::vector<int>& umax)
{
    umax.assign(half + 2, 0);
    const int vmax = cvFloorD(half * std::sqrt(2.f) / 2 + 1), vmin = cvCeilD(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = (int)lrint(std::sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

} // namespace orbh
