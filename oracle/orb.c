/*
 * orb.c -- CPU restatement of cv::ORB (modules/features2d/src/orb.cpp), SURVEY section 8 f3 "features2d detectors".
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Pinned against the reference itself (oracle/_ref, ref_ORB) by tests/test_oracle_orb.py.
 *
 * ORB_Impl::detectAndCompute (orb.cpp:1012-1255):
 *   1. pyramid: every level is cv::resize(INTER_LINEAR_EXACT) of the previous one, packed side by side into one 8-bit buffer with a reflected
 *      border of max(edgeThreshold, ceil(halfPatch * sqrt 2), 4) + 1 pixels around each level (:1040-1143);
 *   2. computeKeyPoints (:775-1000): per level cv::FAST with suppression, KeyPointsFilter::runByImageBorder (keypoint.cpp:107) and
 *      retainBest (:70, std::nth_element + std::partition of libstdc++: restated below because the ORDER of the survivors is the order of
 *      the output); HarrisResponses (:131-180, integer gradients, one float expression), a second retainBest, ICAngles (:184-219) with
 *      cv::fastAtan2 (core mathfuncs_core.simd.hpp:50-74, baseline build: no fused operations), coordinates scaled to level 0;
 *   3. descriptors: every level smoothed by cv::GaussianBlur(7 x 7, sigma 2) -- on a submatrix with a non-isolated border, i.e. NOT the
 *      bit-exact 8-bit path but cv::sepFilter2D with float taps (smooth.dispatch.cpp:656, :829) -- then computeOrbDescriptors (:223-349) on
 *      the pattern rotated by the keypoint angle with cos / sin of libm in float.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } KP;    /* cv::KeyPoint, 28 bytes (core/types.hpp:777) */
typedef struct { int x, y, w, h; } RectI;

static const int bitPattern31[1024] = {
#include "orb_pattern.inc"
};

static int roundD(double v) { return (int)lrint(v); }          /* cvRound: round half to even (fast_math.hpp:200) */
static int roundF(float v) { return (int)lrintf(v); }
static int floorD(double v) { int i = (int)v; return i - (i > v); }
static int ceilD(double v) { int i = (int)v; return i + (i < v); }

/* ------------------------------------------------------------------------------------------------ libstdc++ (GCC 11, bits/stl_algo.h, stl_heap.h)
 * std::nth_element = __introselect: median-of-three quickselect down to ranges of three, insertion sort at the end, heap select when the
 * depth budget 2 * floor(log2 n) runs out.  comp(a, b) = a.response > b.response (keypoint.cpp:60). */
static int gt(const KP* a, const KP* b) { return a->response > b->response; }
static void swp(KP* a, KP* b) { KP t = *a; *a = *b; *b = t; }

static void adjustHeap(KP* first, long hole, long len, KP value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt(first + child, first + (child - 1))) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;                                            /* __push_heap */
    while (hole > top && gt(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static int heapSelectCalls = 0;
int orc_heapSelectCalls(void) { return heapSelectCalls; }          /* lets the test see that the fallback branch was really taken */
static void heapSelect(KP* first, KP* middle, KP* last)
{
    heapSelectCalls++;
    const long len = middle - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; parent--) {                       /* __make_heap */
            KP v = first[parent];
            adjustHeap(first, parent, len, v);
            if (parent == 0) break;
        }
    for (KP* i = middle; i < last; i++)
        if (gt(i, first)) { KP v = *i; *i = *first; adjustHeap(first, 0, len, v); }      /* __pop_heap */
}
static void insertionSort(KP* first, KP* last)
{
    if (first == last) return;
    for (KP* i = first + 1; i != last; i++) {
        KP v = *i;
        if (gt(i, first)) { memmove(first + 1, first, (size_t)(i - first) * sizeof(KP)); *first = v; }
        else { KP* l = i; KP* nx = i - 1; while (gt(&v, nx)) { *l = *nx; l = nx; nx--; } *l = v; }
    }
}
static void nthElement(KP* first, KP* nth, KP* last)
{
    if (first == last || nth == last) return;
    long n = last - first, depth = 0;
    while (n > 1) { n >>= 1; depth++; }
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) { heapSelect(first, nth + 1, last); swp(first, nth); return; }
        depth--;
        KP* a = first + 1; KP* b = first + (last - first) / 2; KP* c = last - 1;           /* __move_median_to_first(first, a, b, c) */
        if (gt(a, b)) { if (gt(b, c)) swp(first, b); else if (gt(a, c)) swp(first, c); else swp(first, a); }
        else if (gt(a, c)) swp(first, a);
        else if (gt(b, c)) swp(first, c);
        else swp(first, b);
        KP* lo = first + 1; KP* hi = last;                                               /* __unguarded_partition(first + 1, last, pivot = first) */
        for (;;) {
            while (gt(lo, first)) lo++;
            hi--;
            while (gt(first, hi)) hi--;
            if (!(lo < hi)) break;
            swp(lo, hi);
            lo++;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    insertionSort(first, last);
}
/* KeyPointsFilter::retainBest (keypoint.cpp:70-92); std::partition is the bidirectional form (stl_algo.h __partition) */
static int retainBest(KP* k, int n, int npoints)
{
    if (npoints < 0 || n <= npoints) return n;
    if (npoints == 0) return 0;
    nthElement(k, k + npoints - 1, k + n);
    const float amb = k[npoints - 1].response;
    KP* first = k + npoints; KP* last = k + n;
    for (;;) {
        for (;;) { if (first == last) return (int)(first - k); if (first->response >= amb) first++; else break; }
        last--;
        for (;;) { if (first == last) return (int)(first - k); if (!(last->response >= amb)) last--; else break; }
        swp(first, last);
        first++;
    }
}
int orc_retainBest(void* kps, int n, int npoints) { return retainBest((KP*)kps, n, npoints); }

/* KeyPointsFilter::runByImageBorder (keypoint.cpp:107-119): Rect((b, b), (w - b, h - b)).contains(Point_<int>(pt)), the conversion rounds */
static int runByImageBorder(KP* k, int n, int w, int h, int b)
{
    if (b <= 0) return n;
    if (h <= b * 2 || w <= b * 2) return 0;
    int m = 0;
    for (int i = 0; i < n; i++) {
        const int x = roundF(k[i].x), y = roundF(k[i].y);
        if (x >= b && x < w - b && y >= b && y < h - b) k[m++] = k[i];
    }
    return m;
}

/* cv::fastAtan2 (mathfuncs_core.simd.hpp:50-74 atan_f32 in the baseline unit: separate multiplies and adds) */
float orc_fastAtan2(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795), p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795),
                p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795), p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)2.2204460492503131e-16); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)2.2204460492503131e-16); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cv::RNG (core/operations.hpp:349-373): multiply-with-carry, uniform(a, b) = a + next() % (b - a) */
static unsigned rngNext(uint64_t* s) { *s = (uint64_t)(unsigned)*s * 4164903690U + (unsigned)(*s >> 32); return (unsigned)*s; }
static int rngUniform(uint64_t* s, int a, int b) { return a == b ? a : (int)(rngNext(s) % (unsigned)(b - a) + a); }

/* the sampling pattern of a parameter set (orb.cpp:1205-1223): 512 points for WTA_K = 2, descriptorSize * 4 tuples drawn from them otherwise */
static int buildPattern(int patchSize, int wta_k, int* pat /* up to 512 points, x then y */)
{
    int pool[1024];
    if (patchSize == 31) memcpy(pool, bitPattern31, sizeof(pool));
    else {                                                                              /* makeRandomPattern :641-650 */
        uint64_t s = 0x34985739;
        for (int i = 0; i < 512; i++) { pool[2 * i] = rngUniform(&s, -patchSize / 2, patchSize / 2 + 1); pool[2 * i + 1] = rngUniform(&s, -patchSize / 2, patchSize / 2 + 1); }
    }
    if (wta_k == 2) { memcpy(pat, pool, sizeof(pool)); return 512; }
    const int ntuples = 32 * 4;                                                         /* initializeOrbPattern :352-376 */
    uint64_t s = 0x12345678;
    for (int i = 0; i < ntuples; i++)
        for (int k = 0; k < wta_k; k++)
            for (;;) {
                const int idx = rngUniform(&s, 0, 512);
                const int px = pool[2 * idx], py = pool[2 * idx + 1];
                int k1 = 0;
                for (; k1 < k; k1++) if (pat[2 * (wta_k * i + k1)] == px && pat[2 * (wta_k * i + k1) + 1] == py) break;
                if (k1 == k) { pat[2 * (wta_k * i + k)] = px; pat[2 * (wta_k * i + k) + 1] = py; break; }
            }
    return ntuples * wta_k;
}

/* the circular patch of ICAngles: last column of every row (orb.cpp:806-823) */
static void buildUmax(int halfPatch, int* umax /* halfPatch + 2 */)
{
    const int vmax = floorD(halfPatch * sqrtf(2.f) / 2 + 1), vmin = ceilD(halfPatch * sqrtf(2.f) / 2);
    for (int v = 0; v <= halfPatch + 1; v++) umax[v] = 0;
    for (int v = 0; v <= vmax; v++) umax[v] = roundD(sqrt((double)halfPatch * halfPatch - v * v));
    for (int v = halfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}
void orc_orbUmax(int halfPatch, int* umax) { buildUmax(halfPatch, umax); }

typedef struct {
    int nLevels, border, bufW, bufH;
    RectI layer[64];
    float scale[64];
} Layout;

static float getScale(int level, int firstLevel, double scaleFactor) { return (float)pow(scaleFactor, (double)(level - firstLevel)); }

/* orb.cpp:1026-1095 */
static void buildLayout(Layout* L, int w, int h, int nLevels, int firstLevel, double scaleFactor, int edgeThreshold, int patchSize)
{
    const int halfPatch = patchSize / 2, descPatch = ceilD(halfPatch * sqrt(2.0));
    const int HB = 9 / 2;
    int border = descPatch > HB ? descPatch : HB;
    if (edgeThreshold > border) border = edgeThreshold;
    border += 1;
    L->border = border; L->nLevels = nLevels;
    const float l0inv = 1.0f / getScale(0, firstLevel, scaleFactor);
    const size_t l0w = (size_t)roundF(w * l0inv), l0h = (size_t)roundF(h * l0inv);
    L->bufW = (int)((l0w + border * 2 + 15) & ~(size_t)15);
    int level_dy = (int)l0h + border * 2, ox = 0, oy = 0;
    for (int level = 0; level < nLevels; level++) {
        const float scale = getScale(level, firstLevel, scaleFactor);
        L->scale[level] = scale;
        const float inv = 1.0f / scale;
        const int sw = roundF(w * inv), sh = roundF(h * inv);
        const int ww = sw + border * 2, wh = sh + border * 2;
        if (ox + ww > L->bufW) { ox = 0; oy += level_dy; level_dy = wh; }
        L->layer[level].x = ox + border; L->layer[level].y = oy + border; L->layer[level].w = sw; L->layer[level].h = sh;
        ox += ww;
    }
    L->bufH = oy + level_dy;
}

/* copyMakeBorder(BORDER_REFLECT_101) of the level that already sits in the buffer, or of the source image (orb.cpp:1125-1136) */
static void borderFill(uint8_t* pyr, int pitch, RectI r, int border, const uint8_t* src, size_t sstep)
{
    for (int y = -border; y < r.h + border; y++) {
        const int sy = orc_borderInterpolate(y, r.h, ORC_BORDER_REFLECT_101);
        for (int x = -border; x < r.w + border; x++) {
            const int sx = orc_borderInterpolate(x, r.w, ORC_BORDER_REFLECT_101);
            if (!src && sx == x && sy == y) continue;
            pyr[(size_t)(r.y + y) * pitch + r.x + x] = src ? src[(size_t)sy * sstep + sx] : pyr[(size_t)(r.y + sy) * pitch + r.x + sx];
        }
    }
}

/* mpyr / mask: the mask pyramid of orb.cpp:1104-1148 (interiors only: its constant ring is never read -- candidates are interior pixels);
 * a resized mask level above firstLevel goes through cv::threshold(254, 0, THRESH_TOZERO), i.e. only 255 survives */
static int buildPyramid(const Layout* L, uint8_t* pyr, const uint8_t* img, size_t step, int w, int h, int firstLevel, uint8_t* mpyr, const uint8_t* mask, size_t mstep)
{
    const int pitch = L->bufW;
    const uint8_t* prev = img; size_t pstep = step; int pw = w, ph = h;
    const uint8_t* prevM = mask; size_t pmstep = mstep;
    for (int level = 0; level < L->nLevels; level++) {
        const RectI r = L->layer[level];
        uint8_t* cur = pyr + (size_t)r.y * pitch + r.x;
        uint8_t* curM = mpyr ? mpyr + (size_t)r.y * pitch + r.x : NULL;
        if (level != firstLevel) {
            if (orc_resize(prev, pstep, pw, ph, cur, (size_t)pitch, r.w, r.h, 0, 1, 0.0, 0.0, 5)) return 1;
            borderFill(pyr, pitch, r, L->border, NULL, 0);
            if (curM) {
                if (orc_resize(prevM, pmstep, pw, ph, curM, (size_t)pitch, r.w, r.h, 0, 1, 0.0, 0.0, 5)) return 1;
                if (level > firstLevel)
                    for (int y = 0; y < r.h; y++) for (int x = 0; x < r.w; x++) if (curM[(size_t)y * pitch + x] <= 254) curM[(size_t)y * pitch + x] = 0;
            }
        } else {
            borderFill(pyr, pitch, r, L->border, img, step);
            if (curM) for (int y = 0; y < r.h; y++) memcpy(curM + (size_t)y * pitch, mask + (size_t)y * mstep, (size_t)r.w);
        }
        if (level > firstLevel) { prev = cur; pstep = (size_t)pitch; pw = r.w; ph = r.h; prevM = curM; pmstep = (size_t)pitch; }
    }
    return 0;
}

/* HarrisResponses (orb.cpp:131-180), blockSize 7 */
static float harrisAt(const uint8_t* pyr, int step, int cx, int cy, float harris_k)
{
    const int blockSize = 7, r = blockSize / 2;
    const float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* ptr0 = pyr + (size_t)(cy - r) * step + (cx - r);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* p = ptr0 + i * step + j;
            const int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            const int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ICAngles (orb.cpp:184-219) */
static float icAngleAt(const uint8_t* center, int step, const int* umax, int half_k)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int vp = center[u + v * step], vm = center[u - v * step];
            v_sum += (vp - vm);
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return orc_fastAtan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptors (orb.cpp:223-349) for one keypoint */
static void describe(const uint8_t* center, int step, float angleDeg, const int* pat, int wta_k, uint8_t* desc)
{
    float angle = angleDeg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    const float a = cosf(angle), b = sinf(angle);
#define GV(idx) (center[roundF(pat[2 * (idx)] * b + pat[2 * (idx) + 1] * a) * step + roundF(pat[2 * (idx)] * a - pat[2 * (idx) + 1] * b)])
    for (int i = 0; i < 32; i++) {
        int val = 0;
        if (wta_k == 2) {
            for (int k = 0; k < 8; k++) { const int t0 = GV(2 * k), t1 = GV(2 * k + 1); val |= (t0 < t1) << k; }
            pat += 32;
        } else if (wta_k == 3) {
            for (int k = 0; k < 4; k++) {
                const int t0 = GV(3 * k), t1 = GV(3 * k + 1), t2 = GV(3 * k + 2);
                val |= (t2 > t1 ? (t2 > t0 ? 2 : 0) : (t1 > t0)) << (2 * k);
            }
            pat += 24;
        } else {
            for (int k = 0; k < 4; k++) {
                int t0 = GV(4 * k), t1 = GV(4 * k + 1), t2 = GV(4 * k + 2), t3 = GV(4 * k + 3), u = 0, v = 2;
                if (t1 > t0) t0 = t1, u = 1;
                if (t3 > t2) t2 = t3, v = 3;
                val |= (t0 > t2 ? u : v) << (2 * k);
            }
            pat += 32;
        }
        desc[i] = (uint8_t)val;
    }
#undef GV
}

/* the pyramid buffer of a parameter set, for tests that look at intermediate stages: returns bufW * bufH bytes (caller frees), layout in out[] =
 * {nLevels, border, bufW, bufH, then x, y, w, h per level} */
uint8_t* orc_orbPyramid(const uint8_t* img, size_t step, int w, int h, int nlevels, double scaleFactor, int edgeThreshold, int firstLevel, int patchSize, int* out)
{
    Layout L;
    if (nlevels < 1 || nlevels > 64) return NULL;
    buildLayout(&L, w, h, nlevels, firstLevel, (double)scaleFactor, edgeThreshold, patchSize);
    uint8_t* pyr = (uint8_t*)calloc((size_t)L.bufW * L.bufH, 1);
    if (!pyr || buildPyramid(&L, pyr, img, step, w, h, firstLevel, NULL, NULL, 0)) { free(pyr); return NULL; }
    out[0] = L.nLevels; out[1] = L.border; out[2] = L.bufW; out[3] = L.bufH;
    for (int i = 0; i < L.nLevels; i++) { out[4 + 4 * i] = L.layer[i].x; out[5 + 4 * i] = L.layer[i].y; out[6 + 4 * i] = L.layer[i].w; out[7 + 4 * i] = L.layer[i].h; }
    return pyr;
}
void orc_free(void* p) { free(p); }

/* cv::GaussianBlur(level, level, Size(7, 7), 2, 2, BORDER_REFLECT_101) on every level of the buffer, rings left as they are (orb.cpp:1183-1196) */
static void blurLevels(const Layout* L, uint8_t* pyr)
{
    const int pitch = L->bufW;
    double g[7];
    orc_getGaussianKernel(7, 2.0, g);
    for (int i = 0; i < 7; i++) g[i] = (double)(float)g[i];                  /* createGaussianKernels: CV_32F taps for an 8-bit image (smooth.dispatch.cpp:278) */
    for (int l = 0; l < L->nLevels; l++) {
        const RectI r = L->layer[l];
        uint8_t* im = pyr + (size_t)r.y * pitch + r.x;
        uint8_t* tmp = (uint8_t*)malloc((size_t)r.w * r.h);
        /* the neighbours of the submatrix are its own reflected border, so the isolated form gives the same pixels */
        orc_sepFilter2D(im, (size_t)pitch, tmp, (size_t)r.w, r.w, r.h, 1, 0, 0, r.w, r.h, 0, 0, g, 7, g, 7, -1, -1, 0.0, ORC_BORDER_REFLECT_101);
        for (int y = 0; y < r.h; y++) memcpy(im + (size_t)y * pitch, tmp + (size_t)y * r.w, (size_t)r.w);
        free(tmp);
    }
}

/* the buffer the descriptors are sampled from (every level smoothed), layout as orc_orbPyramid */
uint8_t* orc_orbPyramidBlurred(const uint8_t* img, size_t step, int w, int h, int nlevels, double scaleFactor, int edgeThreshold, int firstLevel, int patchSize, int* out)
{
    uint8_t* pyr = orc_orbPyramid(img, step, w, h, nlevels, scaleFactor, edgeThreshold, firstLevel, patchSize, out);
    if (!pyr) return NULL;
    Layout L;
    buildLayout(&L, w, h, nlevels, firstLevel, (double)scaleFactor, edgeThreshold, patchSize);
    blurLevels(&L, pyr);
    return pyr;
}
/* the sampling pattern of a parameter set as 2 * (512 or 128 * WTA_K) ints (orb.cpp:1205-1223); returns the count of ints */
int orc_orbPattern(int patchSize, int wta_k, int* pat)
{
    buildPattern(patchSize, wta_k, pat);
    return wta_k == 2 ? 1024 : 128 * wta_k * 2;
}

/* cv::ORB::detectAndCompute on a CV_8UC1 image, optional CV_8UC1 mask of the same size (NULL: none).  useProvided: nIn keypoints come in through kps
 * (the mask is then unused, as in the reference).  Returns the keypoint count (kps / desc are filled up to cap), -1 for parameters outside the restatement. */
int orc_ORBmask(const uint8_t* img, size_t step, int w, int h, const uint8_t* mask, size_t mstep, int nfeatures, double scaleFactor, int nlevels, int edgeThreshold,
                int firstLevel, int wta_k, int scoreType, int patchSize, int fastThreshold, int useProvided, void* kpsIO, int nIn, int cap, uint8_t* desc, int doDesc)
{
    /* scaleFactor: the double the reference keeps -- ORB::create fills it from a float, setScaleFactor from a double (orb.cpp:660, :1262) */
    KP* io = (KP*)kpsIO;
    if (patchSize < 2 || firstLevel < 0 || (wta_k != 2 && wta_k != 3 && wta_k != 4) || w <= 0 || h <= 0) return -1;
    int nLevels = nlevels;
    KP* all = NULL; int nAll = 0, sortedByLevel = 1;
    if (useProvided) {
        nLevels = 0;
        for (int i = 0; i < nIn; i++) {
            if (io[i].octave < 0) return -1;
            if (i > 0 && io[i].octave < io[i - 1].octave) sortedByLevel = 0;
            if (io[i].octave > nLevels) nLevels = io[i].octave;
        }
        nLevels++;
    }
    if (nLevels < 1 || nLevels > 64) return -1;
    Layout L;
    buildLayout(&L, w, h, nLevels, firstLevel, scaleFactor, edgeThreshold, patchSize);
    const int pitch = L.bufW;
    uint8_t* pyr = (uint8_t*)calloc((size_t)L.bufW * L.bufH, 1);
    uint8_t* mpyr = (mask && !useProvided) ? (uint8_t*)calloc((size_t)L.bufW * L.bufH, 1) : NULL;
    if (!pyr || (mask && !useProvided && !mpyr)) { free(pyr); free(mpyr); return -1; }
    if (buildPyramid(&L, pyr, img, step, w, h, firstLevel, mpyr, mask, mstep)) { free(pyr); free(mpyr); return -1; }

    if (!useProvided) {
        /* computeKeyPoints :775-1000 */
        int nfl[64];
        const float factor = (float)(1.0 / scaleFactor);
        float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nLevels));
        int sum = 0;
        for (int l = 0; l < nLevels - 1; l++) { nfl[l] = roundF(nd); sum += nfl[l]; nd *= factor; }
        nfl[nLevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
        const int half = patchSize / 2;
        int* umax = (int*)malloc(sizeof(int) * (half + 2));
        buildUmax(half, umax);
        int counters[64], capAll = 0;
        for (int l = 0; l < nLevels; l++) {
            const RectI r = L.layer[l];
            const uint8_t* im = pyr + (size_t)r.y * pitch + r.x;
            int fcap = r.w * r.h / 4 + 16;
            float* f = (float*)malloc(sizeof(float) * 3 * fcap);
            int n = orc_FAST(im, (size_t)pitch, r.w, r.h, fastThreshold, 1, 2, f, fcap);
            if (n < 0 || n > fcap) { free(f); free(umax); free(all); free(pyr); free(mpyr); return -1; }
            if (mpyr) {                                                          /* FastFeatureDetector::detect -> KeyPointsFilter::runByPixelsMask (fast.cpp:583, keypoint.cpp:146-165) */
                const uint8_t* mk = mpyr + (size_t)r.y * pitch + r.x;
                int m = 0;
                for (int i = 0; i < n; i++)
                    if (mk[(size_t)(int)(f[3 * i + 1] + 0.5f) * pitch + (int)(f[3 * i] + 0.5f)] != 0) { f[3 * m] = f[3 * i]; f[3 * m + 1] = f[3 * i + 1]; f[3 * m + 2] = f[3 * i + 2]; m++; }
                n = m;
            }
            KP* k = (KP*)malloc(sizeof(KP) * (n + 1));
            for (int i = 0; i < n; i++) { k[i].x = f[3 * i]; k[i].y = f[3 * i + 1]; k[i].size = 7.f; k[i].angle = -1.f; k[i].response = f[3 * i + 2]; k[i].octave = 0; k[i].class_id = -1; }
            free(f);
            n = runByImageBorder(k, n, r.w, r.h, edgeThreshold);
            n = retainBest(k, n, scoreType == 0 ? 2 * nfl[l] : nfl[l]);
            counters[l] = n;
            for (int i = 0; i < n; i++) { k[i].octave = l; k[i].size = patchSize * L.scale[l]; }
            if (nAll + n > capAll) { capAll = (nAll + n) * 2 + 64; all = (KP*)realloc(all, sizeof(KP) * capAll); }
            if (n) memcpy(all + nAll, k, sizeof(KP) * n);
            nAll += n;
            free(k);
        }
        if (nAll && scoreType == 0) {
            for (int i = 0; i < nAll; i++) {
                const RectI r = L.layer[all[i].octave];
                all[i].response = harrisAt(pyr, pitch, roundF(all[i].x) + r.x, roundF(all[i].y) + r.y, 0.04f);
            }
            int off = 0, m = 0;
            KP* kept = (KP*)malloc(sizeof(KP) * (nAll + 1));
            for (int l = 0; l < nLevels; l++) {
                const int n = retainBest(all + off, counters[l], nfl[l]);
                memcpy(kept + m, all + off, sizeof(KP) * n);
                m += n; off += counters[l];
            }
            free(all); all = kept; nAll = m;
        }
        for (int i = 0; i < nAll; i++) {
            const RectI r = L.layer[all[i].octave];
            all[i].angle = icAngleAt(pyr + (size_t)(roundF(all[i].y) + r.y) * pitch + roundF(all[i].x) + r.x, pitch, umax, half);
        }
        for (int i = 0; i < nAll; i++) { const float s = L.scale[all[i].octave]; all[i].x *= s; all[i].y *= s; }
        free(umax);
    } else {
        all = (KP*)malloc(sizeof(KP) * (nIn + 1));
        memcpy(all, io, sizeof(KP) * nIn);
        nAll = runByImageBorder(all, nIn, w, h, edgeThreshold);
        if (!sortedByLevel) {                                                    /* stable regrouping by level :1159-1172 */
            KP* t = (KP*)malloc(sizeof(KP) * (nAll + 1));
            int m = 0;
            for (int l = 0; l < nLevels; l++) for (int i = 0; i < nAll; i++) if (all[i].octave == l) t[m++] = all[i];
            free(all); all = t;
        }
    }

    if (doDesc && nAll) {
        int pat[1024];
        buildPattern(patchSize, wta_k, pat);
        blurLevels(&L, pyr);
        for (int j = 0; j < nAll && j < cap; j++) {
            const RectI r = L.layer[all[j].octave];
            const float scale = 1.f / L.scale[all[j].octave];
            const uint8_t* center = pyr + (size_t)(roundF(all[j].y * scale) + r.y) * pitch + roundF(all[j].x * scale) + r.x;
            describe(center, pitch, all[j].angle, pat, wta_k, desc + (size_t)j * 32);
        }
    }
    for (int i = 0; i < nAll && i < cap; i++) io[i] = all[i];
    free(all); free(pyr); free(mpyr);
    return nAll;
}

int orc_ORB(const uint8_t* img, size_t step, int w, int h, int nfeatures, double scaleFactor, int nlevels, int edgeThreshold, int firstLevel, int wta_k,
            int scoreType, int patchSize, int fastThreshold, int useProvided, void* kps, int nIn, int cap, uint8_t* desc, int doDesc)
{
    return orc_ORBmask(img, step, w, h, NULL, 0, nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, wta_k, scoreType, patchSize, fastThreshold, useProvided, kps, nIn, cap, desc, doDesc);
}
