// orb_emu.cpp -- the lines of opencv_amd/csrc/orb_math.h (what the ORB kernels run per lane) and orb_host.h (the host control flow of orb.hip) compiled
// for the CPU and driven lane by lane / thread by thread, so that tests/test_hostemu.py can hold them against the pinned restatement (oracle/orb.c)
// where no GPU is present.  Test infrastructure.
#include "orb_host.h"

using namespace orbh;

extern "C" {

// layout as oracle/orb.c's orc_orbPyramid reports it: {nLevels, border, bufW, bufH, then x, y, w, h per level}; returns the pitch the kernels use
int emu_orb_layout(int w, int h, int nlevels, int firstLevel, double scaleFactor, int edgeThreshold, int patchSize, int* out, float* scales)
{
    Layout L;
    buildLayout(L, w, h, nlevels, firstLevel, (double)scaleFactor, edgeThreshold, patchSize);
    out[0] = L.nLevels; out[1] = L.border; out[2] = L.bufW; out[3] = L.bufH;
    for (int i = 0; i < nlevels; i++) { out[4 + 4 * i] = L.layer[i].x; out[5 + 4 * i] = L.layer[i].y; out[6 + 4 * i] = L.layer[i].w; out[7 + 4 * i] = L.layer[i].h; scales[i] = L.scale[i]; }
    return L.pitch;
}

// k_orb_border over its launch grid (every thread of every workgroup, including the ones the guards turn away); returns the count of threads that ran
long emu_orb_border(unsigned char* pyr, int w, int h, int nlevels, int firstLevel, double scaleFactor, int edgeThreshold, int patchSize, int level, const unsigned char* src, size_t sstep)
{
    Layout L;
    buildLayout(L, w, h, nlevels, firstLevel, (double)scaleFactor, edgeThreshold, patchSize);
    const BorderGrid b = borderGrid(L, level);
    long ran = 0;
    const int gx = (b.ng + 63) / 64, gy = (b.nrows + 3) / 4;
    for (int by = 0; by < gy; by++) for (int bx = 0; bx < gx; bx++) for (int t = 0; t < 256; t++) {
        const int g = bx * 64 + (t & 63), row = by * 4 + (t >> 6);
        if (g >= b.ng || row >= b.nrows) continue;
        orbm::borderDword(pyr, L.pitch, L.layer[level], L.border, src, sstep, b.row0 + row, b.g0 + g);
        ran++;
    }
    return ran;
}

// k_orb_score_angle for one keypoint: the 64 lanes one after another, the wave sums as plain sums
void emu_orb_score_angle(const unsigned char* pyr, int pitch, int cx, int cy, int half, float harris_k, float* out)
{
    std::vector<int> umax;
    buildUmax(half, umax);
    int A = 0, B = 0, C = 0, M01 = 0, M10 = 0;
    for (int lane = 0; lane < 64; lane++) {
        int a, b, c, m01, m10;
        orbm::harrisLane(pyr, pitch, cx, cy, lane, a, b, c);
        orbm::angleLane(pyr + (size_t)cy * pitch + cx, pitch, umax.data(), half, lane, m01, m10);
        A += a; B += b; C += c; M01 += m01; M10 += m10;
    }
    out[0] = orbm::harrisFinish(A, B, C, harris_k);
    out[1] = orbm::fastAtan2((float)M01, (float)M10);
}

// k_orb_desc for one keypoint (32 threads), with the host's preparation of (cos, sin) and of the pattern bytes
void emu_orb_desc(const unsigned char* pyr, int pitch, int cx, int cy, float angleDeg, int patchSize, int wta_k, unsigned char* desc)
{
    signed char pat[1024];
    buildPattern(patchSize, wta_k, pat);
    float angle = angleDeg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    const float a = cosf(angle), b = sinf(angle);
    for (int byte = 0; byte < 32; byte++) desc[byte] = (unsigned char)orbm::descByte(pyr + (size_t)cy * pitch + cx, pitch, a, b, pat, wta_k, byte);
}

int emu_orb_pattern(int patchSize, int wta_k, signed char* out) { return buildPattern(patchSize, wta_k, out); }
void emu_orb_umax(int half, int* out) { std::vector<int> u; buildUmax(half, u); for (size_t i = 0; i < u.size(); i++) out[i] = u[i]; }
float emu_orb_fastAtan2(float y, float x) { return orbm::fastAtan2(y, x); }

int emu_orb_retainBest(KP* k, int n, int npoints)
{
    std::vector<KP> v(k, k + n);
    retainBest(v, npoints);
    for (size_t i = 0; i < v.size(); i++) k[i] = v[i];
    return (int)v.size();
}
// the culls on (response, pixel index) records: keypoints come in with x = pixel index, the survivors go back out in their order
int emu_orb_cullCand(KP* k, int n, int w, int npoints)
{
    std::vector<Cand> c((size_t)n);
    for (int i = 0; i < n; i++) c[(size_t)i] = {k[i].response, (uint32_t)k[i].class_id};
    retainBestCand(c, npoints);
    for (size_t i = 0; i < c.size(); i++) { k[i].response = c[i].response; k[i].class_id = (int)c[i].idx; k[i].x = (float)(c[i].idx % (unsigned)w); k[i].y = (float)(c[i].idx / (unsigned)w); }
    return (int)c.size();
}
int emu_orb_runByImageBorder(KP* k, int n, int w, int h, int b)
{
    std::vector<KP> v(k, k + n);
    runByImageBorder(v, w, h, b);
    for (size_t i = 0; i < v.size(); i++) k[i] = v[i];
    return (int)v.size();
}

}
