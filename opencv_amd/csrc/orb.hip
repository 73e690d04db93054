// orb.hip -- SURVEY §8 f3, "features2d detectors": cv::ORB (modules/features2d/src/orb.cpp, ORB_Impl::detectAndCompute :1012-1255) as one call on a
// frame that stays in HBM.  The reference has no HAL hook for ORB; with the imgproc / features2d hooks alone its resize, FAST and blur calls each cross
// PCIe twice per pyramid level.  Here the pyramid buffer, the FAST score images, the smoothed pyramid and the pattern live in device scratch; the host
// sees the candidate lists (to cull them the way the reference does) and the final keypoints / descriptors.
//
//   pyramid      every level = cv::resize(INTER_LINEAR_EXACT) of the previous one (mi355cv_resize, warp.hip k_resize_exact), packed side by side in one
//                8-bit buffer with a BORDER_REFLECT_101 ring of max(edgeThreshold, ceil(halfPatch sqrt 2), 4) + 1 pixels (k_orb_border): the reference's
//                own layout (orb.cpp:1056-1095), so Harris / angle / descriptor reads near a level's edge see the same pixels
//   keypoints    FAST 9-16 with suppression on all levels in one launch per pass (fast.hip k_fast_*_levels), the candidates of all levels in raster order
//                by a row count / scan / write (no sort, no atomics; the mask and image-border tests of KeyPointsFilter are part of the candidate test),
//                culled on the host by KeyPointsFilter::retainBest -- std::nth_element + std::partition: the reference's output ORDER is that of the C++
//                library, so the host side calls the same two algorithms --, then ONE kernel gives every candidate its Harris response (7 x 7 block of
//                integer gradients) and its intensity-centroid angle (k_orb_score_angle: a wavefront per keypoint, exact integer sums, cv::fastAtan2's
//                polynomial in the reference's operation order); second cull per level on the host
//   descriptors  levels smoothed by cv::GaussianBlur(7 x 7, sigma 2): on a submatrix with a non-isolated border that is cv::sepFilter2D with float taps, not the
//                bit-exact 8-bit Gaussian (smooth.dispatch.cpp:656,829) -> mi355cv_sepFilter (seproll.hip) into a second pyramid buffer; k_orb_desc: a thread
//                per descriptor byte, pattern in LDS as signed bytes, rotation (cos, sin) computed on the host with the C library's cosf / sinf like the reference
#include "rt.h"
#include "orb_math.h"
#include "orb_host.h"
#include "fast_levels.h"
#include "gausskernel.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <vector>

namespace mi355 {
void fastLevelsScores(const uchar* pyr, size_t pitch, uchar* sc, uchar* sup, const FastLevels& L, hipStream_t st);                         // fast.hip
void fastLevelsCollect(const uchar* sup, size_t pitch, const uchar* mask, int thr, int edge, const FastLevels& L, unsigned* rowCount, unsigned* rowOff, unsigned* levelTotal,
                       unsigned long long* keys, hipStream_t st);
}

using namespace mi355;
using namespace orbh;

namespace {

static_assert(sizeof(KP) == sizeof(mi355cv_KeyPoint) && sizeof(KP) == 28, "cv::KeyPoint layout");
struct LayerTab { orbm::Layer l[MAX_LEVELS]; };
struct CandKp { int x, y, level, pad; };                 // a candidate in level coordinates
struct DescKp { int cx, cy; float a, b; };               // centre in buffer coordinates, (cos, sin) of the keypoint angle

// ---- kernels ------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_orb_border(uchar* __restrict__ pyr, int pitch, orbm::Layer r, int border, const uchar* __restrict__ src, size_t sstep, int g0, int ng, int row0, int nrows)
{
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (g >= ng || row >= nrows) return;
    orbm::borderDword(pyr, pitch, r, border, src, sstep, row0 + row, g0 + g);
}

// cv::threshold(mask, mask, 254, 0, THRESH_TOZERO) on a resized mask level (orb.cpp:1119): only 255 survives
__global__ __launch_bounds__(256) void k_orb_mask_tozero(uchar* __restrict__ m, int pitch, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uchar* p = m + (size_t)y * pitch + x;
    if (*p <= 254) *p = 0;
}

__device__ __forceinline__ int waveSum(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// a wavefront per candidate: Harris response and intensity-centroid angle
__global__ __launch_bounds__(256) void k_orb_score_angle(const uchar* __restrict__ pyr, int pitch, const CandKp* __restrict__ kp, int n, LayerTab tab,
                                                         const int* __restrict__ umax, int half, float harris_k, float2* __restrict__ out)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const CandKp k = kp[i];
    const orbm::Layer L = tab.l[k.level];
    const int cx = k.x + L.x, cy = k.y + L.y;
    int a, b, c, m01, m10;
    orbm::harrisLane(pyr, pitch, cx, cy, lane, a, b, c);
    orbm::angleLane(pyr + (size_t)cy * pitch + cx, pitch, umax, half, lane, m01, m10);
    a = waveSum(a); b = waveSum(b); c = waveSum(c); m01 = waveSum(m01); m10 = waveSum(m10);
    if (lane == 0) out[i] = make_float2(orbm::harrisFinish(a, b, c, harris_k), orbm::fastAtan2((float)m01, (float)m10));
}

// a thread per descriptor byte, 8 keypoints per workgroup; the pattern (<= 512 points as signed bytes) is staged in LDS
__global__ __launch_bounds__(256) void k_orb_desc(const uchar* __restrict__ pyr, int pitch, const DescKp* __restrict__ kp, int n, const signed char* __restrict__ pattern, int patBytes,
                                                  int wta_k, uchar* __restrict__ desc, size_t dstep)
{
    __shared__ signed char pat[1024];
    for (int t = threadIdx.x; t < patBytes; t += 256) pat[t] = pattern[t];
    __syncthreads();
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), byte = threadIdx.x & 31;
    if (i >= n) return;
    const DescKp k = kp[i];
    desc[(size_t)i * dstep + byte] = (uchar)orbm::descByte(pyr + (size_t)k.cy * pitch + k.cx, pitch, k.a, k.b, pat, wta_k, byte);
}

void launchBorder(uchar* pyr, const Layout& L, int level, const uchar* src, size_t sstep, hipStream_t st)
{
    const BorderGrid b = borderGrid(L, level);
    hipLaunchKernelGGL(k_orb_border, dim3(divUp(b.ng, 64), divUp(b.nrows, 4)), dim3(256), 0, st, pyr, L.pitch, L.layer[level], L.border, src, sstep, b.g0, b.ng, b.row0, b.nrows);
}

bool copyD2H(void* dst, const void* src, size_t bytes, hipStream_t st)
{
    return bytes == 0 || (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess);
}

} // namespace

extern "C" {

// cv::ORB::detectAndCompute (orb.cpp:1012) on a CV_8UC1 image (host or device resident), optional CV_8UC1 mask of the same size.
//   use_provided_keypoints == 0: detect (and describe when `descriptors` is not null); != 0: describe the nkeypoints_in keypoints in `keypoints`
//   (Feature2D::compute).  Keypoints leave in the reference's order, at most `capacity` of them (and of descriptor rows, 32 bytes each) are written.
// Returns the keypoint count (may exceed capacity: call again with larger arrays), -1 when the arguments are not served (nothing computed), -2 on a
// device failure.
MI355CV_API int mi355cv_ORB_detectAndCompute(const uchar* image, size_t step, int width, int height, const uchar* mask, size_t mask_step, const mi355cv_OrbParams* prm,
                                             int use_provided_keypoints, mi355cv_KeyPoint* keypoints, int nkeypoints_in, int capacity, uchar* descriptors, size_t descriptors_step)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || !image || !prm || width <= 0 || height <= 0 || capacity < 0 || (capacity > 0 && !keypoints)) return -1;
    const mi355cv_OrbParams p = *prm;
    const bool provided = use_provided_keypoints != 0, doDesc = descriptors != nullptr;
    if (p.patchSize < 2 || p.patchSize > 127 || p.firstLevel < 0 || (p.WTA_K != 2 && p.WTA_K != 3 && p.WTA_K != 4) || (p.scoreType != 0 && p.scoreType != 1) || !(p.scaleFactor > 0.0))
        { setError(MI355CV_NOT_IMPLEMENTED, "ORB: parameters outside the served range"); return -1; }
    if (provided && (nkeypoints_in < 0 || nkeypoints_in > capacity)) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: %d keypoints passed in, room for %d", nkeypoints_in, capacity); return -1; }
    if (doDesc && descriptors_step < 32) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: descriptor rows are 32 bytes"); return -1; }
    if ((long long)width * height > 0x3fffffffLL) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: image beyond 2^30 pixels"); return -1; }
    const double scaleFactor = p.scaleFactor;                               // the double the reference keeps (ORB::create fills it from a float, setScaleFactor from a double: orb.cpp:660,1262)

    std::vector<KP> all;
    int nLevels = p.nlevels;
    bool sortedByLevel = true;
    if (provided) {                                                         // orb.cpp:1042-1062
        all.assign(reinterpret_cast<const KP*>(keypoints), reinterpret_cast<const KP*>(keypoints) + nkeypoints_in);
        nLevels = 0;
        for (int i = 0; i < nkeypoints_in; i++) {
            if (all[i].octave < 0) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: keypoint with a negative octave"); return -1; }
            if (i > 0 && all[i].octave < all[i - 1].octave) sortedByLevel = false;
            nLevels = std::max(nLevels, all[i].octave);
        }
        nLevels++;
    }
    if (nLevels < 1 || nLevels > MAX_LEVELS) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: 1..%d pyramid levels are served", MAX_LEVELS); return -1; }
    Layout L;
    buildLayout(L, width, height, nLevels, p.firstLevel, scaleFactor, p.edgeThreshold, p.patchSize);
    for (int l = 0; l < nLevels; l++) if (L.layer[l].w < 1 || L.layer[l].h < 1) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: a pyramid level is empty"); return -1; }
    if ((long long)L.pitch * L.bufH > 0x7fffffffLL) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: the pyramid buffer (%d x %d bytes) exceeds 2 GiB", L.pitch, L.bufH); return -1; }

    Stager stg;                                  // outermost: the resize / sepFilter hooks called below leave synchronisation and scratch recycling to this one
    if (!ensureDevice()) return -1;
    if (hostImageTooSmall(image, (size_t)width * height, minPixels(HOST_HEAVY))) return -1;
    hipStream_t st = stream();
    size_t iss = 0, mss = 0;
    const uchar* dimg = stg.in(image, step, (size_t)width, height, &iss);
    const uchar* dmask = (mask && !provided) ? stg.in(mask, mask_step, (size_t)width, height, &mss) : nullptr;
    const size_t bufBytes = (size_t)L.pitch * L.bufH;
    uchar* pyr = (uchar*)stg.scratch(bufBytes);
    uchar* mpyr = dmask ? (uchar*)stg.scratch(bufBytes) : nullptr;
    if (!dimg || !pyr || (mask && !provided && (!dmask || !mpyr))) return -1;
    LayerTab tab; memset(&tab, 0, sizeof tab);
    for (int l = 0; l < nLevels; l++) tab.l[l] = L.layer[l];

    // ---- the pyramid (orb.cpp:1098-1143)
    {
        const uchar* prev = dimg; size_t pstep = iss; int pw = width, ph = height;
        const uchar* prevM = dmask; size_t pmstep = mss;
        for (int l = 0; l < nLevels; l++) {
            const orbm::Layer r = L.layer[l];
            uchar* cur = pyr + (size_t)r.y * L.pitch + r.x;
            uchar* curM = mpyr ? mpyr + (size_t)r.y * L.pitch + r.x : nullptr;
            if (l != p.firstLevel) {
                if (mi355cv_resize(MI355CV_8U, prev, pstep, pw, ph, cur, (size_t)L.pitch, r.w, r.h, (double)r.w / pw, (double)r.h / ph, MI355CV_INTER_LINEAR_EXACT) != MI355CV_OK) return -2;
                launchBorder(pyr, L, l, nullptr, 0, st);
                if (curM) {
                    if (mi355cv_resize(MI355CV_8U, prevM, pmstep, pw, ph, curM, (size_t)L.pitch, r.w, r.h, (double)r.w / pw, (double)r.h / ph, MI355CV_INTER_LINEAR_EXACT) != MI355CV_OK) return -2;
                    if (l > p.firstLevel) hipLaunchKernelGGL(k_orb_mask_tozero, dim3(divUp(r.w, 64), divUp(r.h, 4)), dim3(256), 0, st, curM, L.pitch, r.w, r.h);
                }
            } else {
                launchBorder(pyr, L, l, dimg, iss, st);
                if (curM && hipMemcpy2DAsync(curM, (size_t)L.pitch, dmask, mss, (size_t)r.w, (size_t)r.h, hipMemcpyDeviceToDevice, st) != hipSuccess) return -2;
            }
            if (l > p.firstLevel) { prev = cur; pstep = (size_t)L.pitch; pw = r.w; ph = r.h; prevM = curM; pmstep = (size_t)L.pitch; }
        }
    }

    if (!provided) {
        // ---- computeKeyPoints (orb.cpp:775-1000)
        std::vector<int> nfl(nLevels);
        {
            const float factor = (float)(1.0 / scaleFactor);
            float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nLevels));
            int sum = 0;
            for (int l = 0; l < nLevels - 1; l++) { nfl[l] = cvRoundF(nd); sum += nfl[l]; nd *= factor; }
            nfl[nLevels - 1] = std::max(p.nfeatures - sum, 0);
        }
        // FAST on every level: scores and suppression of all levels in one launch each (buffers of the pyramid's geometry), the candidates of all levels in
        // raster order by a row count, a scan and a write pass -- no sort, no atomics; KeyPointsFilter::runByPixelsMask / runByImageBorder
        // (keypoint.cpp:107-165), which follow FAST at once in the reference, are part of the candidate test, so nothing near the edge leaves the GPU
        static_assert(MAX_LEVELS <= FAST_MAX_LEVELS && MAX_LEVELS == mi355::lim::ORB_MAX_LEVELS, "level tables; mi355cv_limit(\"orb_max_levels\")");
        FastLevels FL; memset(&FL, 0, sizeof FL);
        FL.n = nLevels;
        size_t bound = 0;
        const int edge = p.edgeThreshold > 0 ? p.edgeThreshold : 0;
        for (int l = 0; l < nLevels; l++) {
            const orbm::Layer r = L.layer[l];
            FL.x[l] = r.x; FL.y[l] = r.y; FL.w[l] = r.w; FL.h[l] = r.h;
            FL.tile0[l + 1] = FL.tile0[l] + divUp(r.h, 4); FL.row0[l + 1] = FL.row0[l] + r.h;
            if (r.w > 2 * edge && r.h > 2 * edge) bound += ((size_t)(r.w - 2 * edge + 2) / 2) * ((size_t)(r.h - 2 * edge + 2) / 2);   // 3 x 3 strict maxima: at most one per 2 x 2 block
        }
        const int rows = FL.row0[nLevels];
        if (FL.tile0[nLevels] > 65535) { setError(MI355CV_NOT_IMPLEMENTED, "ORB: %d rows of pyramid exceed one launch", rows); return -1; }
        uchar* sc = (uchar*)stg.scratch(bufBytes);
        uchar* sup = (uchar*)stg.scratch(bufBytes);
        unsigned* rowCount = (unsigned*)stg.scratch(sizeof(unsigned) * (size_t)rows);
        unsigned* rowOff = (unsigned*)stg.scratch(sizeof(unsigned) * (size_t)rows);
        unsigned* levelTotal = (unsigned*)stg.scratch(sizeof(unsigned) * (MAX_LEVELS + 1));
        unsigned long long* keys = (unsigned long long*)stg.scratch((bound + 1) * 8);
        if (!sc || !sup || !rowCount || !rowOff || !levelTotal || !keys) return -2;
        int thr = p.fastThreshold < 0 ? 0 : p.fastThreshold > 255 ? 255 : p.fastThreshold;       // fast.cpp:81
        if (!thr) thr = 1;                                                                      // fast.cpp:467 (suppression is always on here)
        fastLevelsScores(pyr, (size_t)L.pitch, sc, sup, FL, st);
        fastLevelsCollect(sup, (size_t)L.pitch, mpyr, thr, edge, FL, rowCount, rowOff, levelTotal, keys, st);
        // counters and candidate lists land in page-locked memory (Stager::pinned): their size is the GPU's decision, and a copy into pageable memory would
        // go through the runtime's bounce buffer with the host blocked
        unsigned* cnt = (unsigned*)stg.pinned(sizeof(unsigned) * (MAX_LEVELS + 1));
        if (!cnt || !copyD2H(cnt, levelTotal, sizeof(unsigned) * (size_t)(nLevels + 1), st)) return -2;
        const size_t total = cnt[nLevels];
        std::vector<size_t> off(nLevels);
        {
            size_t run = 0;
            for (int l = 0; l < nLevels; l++) { off[l] = run; run += cnt[l]; }
            if (run != total || total > bound) { setError(MI355CV_ERROR_UNKNOWN, "ORB: FAST candidate counts inconsistent (%zu of at most %zu)", total, bound); return -2; }
        }
        const unsigned long long* hk = (const unsigned long long*)stg.pinned((total ? total : 1) * 8);
        if (!hk || !copyD2H(const_cast<unsigned long long*>(hk), keys, total * 8, st)) return -2;

        // the first cull, level by level; long lists (a 4K level 0 has 10^5 candidates) on threads of their own: the levels are independent
        std::vector<std::vector<Cand>> cands(nLevels);
        auto cull = [&](int l) {
            std::vector<Cand>& c = cands[l];
            const unsigned long long* k = hk + off[l];
            c.resize(cnt[l]);
            for (size_t i = 0; i < c.size(); i++) c[i] = {(float)((int)(unsigned)(k[i] & 0xffffffffu) - 1), 0xffffffffu - (unsigned)(k[i] >> 32)};
            retainBestCand(c, p.scoreType == 0 ? 2 * nfl[l] : nfl[l]);
        };
        {
            std::vector<std::future<void>> side;
            int big = 0;
            for (int l = 0; l < nLevels; l++) big += cnt[l] >= 8192u;
            std::vector<int> sideLevel;
            for (int l = nLevels - 1; l >= 0; l--) {                       // the longest list (level firstLevel or 0) on the calling thread, last
                bool spawned = false;
                if (cnt[l] >= 8192u && big > 1 && l != 0) {
                    try { side.push_back(std::async(std::launch::async, cull, l)); sideLevel.push_back(l); spawned = true; }
                    catch (...) { spawned = false; }                       // no thread to be had: cull here (nothing may cross the extern "C" boundary)
                }
                if (!spawned && l != 0) cull(l);
            }
            cull(0);
            for (size_t i = 0; i < side.size(); i++) { try { side[i].get(); } catch (...) { cull(sideLevel[i]); } }
        }
        std::vector<int> counts(nLevels);
        std::vector<KP> lvl;
        for (int l = 0; l < nLevels; l++) {
            const orbm::Layer r = L.layer[l];
            counts[l] = (int)cands[l].size();
            const float size = p.patchSize * L.scale[l];
            for (const Cand& c : cands[l]) all.push_back({(float)(c.idx % (unsigned)r.w), (float)(c.idx / (unsigned)r.w), size, -1.f, c.response, l, -1});
        }
        if (!all.empty()) {
            const int n = (int)all.size(), half = p.patchSize / 2;
            std::vector<CandKp> ck(n);
            for (int i = 0; i < n; i++) ck[i] = {cvRoundF(all[i].x), cvRoundF(all[i].y), all[i].octave, 0};
            std::vector<int> umax;
            buildUmax(half, umax);
            const CandKp* dk = (const CandKp*)stg.param(ck.data(), (size_t)n * sizeof(CandKp));
            const int* dum = (const int*)stg.param(umax.data(), umax.size() * sizeof(int));
            float2* dout = (float2*)stg.scratch((size_t)n * sizeof(float2));
            if (!dk || !dum || !dout) return -2;
            hipLaunchKernelGGL(k_orb_score_angle, dim3(divUp(n, 4)), dim3(256), 0, st, pyr, L.pitch, dk, n, tab, dum, half, 0.04f, dout);
            const float2* ho = (const float2*)stg.pinned((size_t)n * sizeof(float2));
            if (!ho || !copyD2H(const_cast<float2*>(ho), dout, (size_t)n * sizeof(float2), st)) return -2;
            for (int i = 0; i < n; i++) { all[i].angle = ho[i].y; if (p.scoreType == 0) all[i].response = ho[i].x; }
            if (p.scoreType == 0) {                                              // second cull per level on the Harris response (orb.cpp:941-961)
                std::vector<KP> kept;
                int off = 0;
                for (int l = 0; l < nLevels; l++) {
                    lvl.assign(all.begin() + off, all.begin() + off + counts[l]);
                    off += counts[l];
                    retainBest(lvl, nfl[l]);
                    kept.insert(kept.end(), lvl.begin(), lvl.end());
                }
                all.swap(kept);
            }
            for (KP& k : all) { const float s = L.scale[k.octave]; k.x *= s; k.y *= s; }
        }
    } else {
        runByImageBorder(all, width, height, p.edgeThreshold);                  // orb.cpp:1157
        if (!sortedByLevel) std::stable_sort(all.begin(), all.end(), [](const KP& a, const KP& b) { return a.octave < b.octave; });      // :1159-1172
    }

    const int nAll = (int)all.size(), take = std::min(nAll, capacity);
    if (doDesc && take > 0 && nAll <= capacity) {
        // ---- descriptors (orb.cpp:1175-1253): smoothed levels into a second buffer whose rings are the unsmoothed ones, like the reference's in-place blur
        uchar* blur = (uchar*)stg.scratch(bufBytes);
        if (!blur) return -2;
        if (hipMemcpyAsync(blur, pyr, bufBytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return -2;
        std::vector<double> g;
        if (!gaussianKernelBitExact(7, 2.0, g)) return -2;
        float gf[7];
        for (int i = 0; i < 7; i++) gf[i] = (float)g[i];                       // createGaussianKernels: CV_32F taps for an 8-bit image (smooth.dispatch.cpp:278)
        cvhalFilter2D* ctx = nullptr;
        if (mi355cv_sepFilterInit(&ctx, MI355CV_8U, MI355CV_8U, MI355CV_32F, (uchar*)gf, 7, (uchar*)gf, 7, -1, -1, 0.0, B_REFLECT_101) != MI355CV_OK) return -2;
        bool used[MAX_LEVELS] = {};
        for (const KP& k : all) used[k.octave] = true;
        int rc = MI355CV_OK;
        for (int l = 0; l < nLevels && rc == MI355CV_OK; l++) {
            if (!used[l]) continue;                                             // a level no keypoint lives on is never sampled
            const orbm::Layer r = L.layer[l];
            const size_t o = (size_t)r.y * L.pitch + r.x;
            rc = mi355cv_sepFilter(ctx, pyr + o, (size_t)L.pitch, blur + o, (size_t)L.pitch, r.w, r.h, r.w, r.h, 0, 0);
        }
        mi355cv_sepFilterFree(ctx);
        if (rc != MI355CV_OK) return -2;
        std::vector<DescKp> dk(nAll);
        for (int j = 0; j < nAll; j++) {
            const KP& k = all[j];
            const orbm::Layer r = L.layer[k.octave];
            const float scale = 1.f / L.scale[k.octave];
            float angle = k.angle;
            angle *= (float)(3.1415926535897932384626433832795 / 180.f);
            dk[j] = {cvRoundF(k.x * scale) + r.x, cvRoundF(k.y * scale) + r.y, cosf(angle), sinf(angle)};
            // a provided keypoint may point anywhere: the patch must stay inside the buffer (the reference reads whatever lies there)
            const int reach = (int)std::ceil((p.patchSize / 2) * 1.4142135623730951) + 1;
            if (dk[j].cx < reach || dk[j].cy < reach || dk[j].cx + reach >= L.bufW || dk[j].cy + reach >= L.bufH)
                { setError(MI355CV_NOT_IMPLEMENTED, "ORB: keypoint %d reaches outside the pyramid buffer", j); return -1; }
        }
        signed char pat[1024];
        const int patBytes = buildPattern(p.patchSize, p.WTA_K, pat);
        const DescKp* ddk = (const DescKp*)stg.param(dk.data(), (size_t)nAll * sizeof(DescKp));
        const signed char* dpat = (const signed char*)stg.param(pat, (size_t)patBytes);
        size_t dds = 0;
        uchar* dd = stg.out(descriptors, descriptors_step, 32, nAll, &dds);
        if (!ddk || !dpat || !dd) return -2;
        hipLaunchKernelGGL(k_orb_desc, dim3(divUp(nAll, 8)), dim3(256), 0, st, blur, L.pitch, ddk, nAll, dpat, patBytes, p.WTA_K, dd, dds);
    }
    if (take > 0 && nAll <= capacity) memcpy(keypoints, all.data(), (size_t)nAll * sizeof(KP));
    const int rc = stg.finish("ORB_detectAndCompute");
    return rc == MI355CV_OK ? nAll : -2;
}

} // extern "C"
