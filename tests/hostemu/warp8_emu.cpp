// tests/hostemu -- TEST INFRASTRUCTURE ONLY.  The LDS-tile warp kernel of opencv_amd/csrc/warp8.h run on the CPU: the same host plan, the same three
// phases, executed thread by thread with barriers replaced by loop boundaries, one "workgroup" at a time.  Pixels the kernel hands to the generic
// sampler are copied from `expect` (the pinned restatement's output) and counted -- the test compares every other pixel with the restatement.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include "warp8.h"

template <int CN, int LW, int NR>
static void leanTileT(const warp8::Args& a, const warp8::LBox& b, int x0, int y0, const unsigned char* src, unsigned char* lds, unsigned char* dst)
{
    static uint32_t v[256][NR];                                             // every thread's staging registers, between the two halves of the staging
    const bool rim = b.kind == warp8::LEAN_RIM;
    for (int tid = 0; tid < 256; tid++) { if (rim) warp8::leanLoad<CN, LW, NR, true>(a, b, src, tid, v[tid]); else warp8::leanLoad<CN, LW, NR, false>(a, b, src, tid, v[tid]); }
    for (int tid = 0; tid < 256; tid++) warp8::leanStore<CN, LW, NR>(a, b, lds, tid, v[tid]);
    for (int tid = 0; tid < 256; tid++) {
        warp8::LeanRowT rt; warp8::leanRowTerms<CN>(a, y0, tid, rt);
        if (rim) warp8::leanRows<CN, true>(a, b, x0, y0, lds, dst, tid, rt); else warp8::leanRows<CN, false>(a, b, x0, y0, lds, dst, tid, rt);
    }
}
template <int CN>
static void leanTileEmu(const warp8::Args& a, const warp8::LBox& b, int x0, int y0, const unsigned char* src, unsigned char* lds, unsigned char* dst)
{
#define LT(LW_, NR_) if (a.leanLW == LW_ && a.leanNR == NR_) return leanTileT<CN, LW_, NR_>(a, b, x0, y0, src, lds, dst)
#define LTS(LW_) LT(LW_, 6); LT(LW_, 10); LT(LW_, 14); LT(LW_, 20)
    LTS(16); LTS(32); LTS(64); LTS(128); LTS(256);
#undef LTS
#undef LT
}

template <int CN, int KIND, int FETCH, bool LEAN = false>
static void run(const warp8::Args& a, size_t ldsBytes, const unsigned char* src, unsigned char* dst, const short* tab, const unsigned char* expect, size_t estep, long long* stats)
{
    std::vector<unsigned char> lds(ldsBytes + 64);
    for (int ty = 0; ty < a.gy; ty++)
        for (int tx = 0; tx < a.gx; tx++) {
            const int x0 = tx * warp8::TW, y0 = ty * a.th;
            std::memset(lds.data(), 0xA5, lds.size());                      // stale LDS must never reach an output pixel
            if constexpr (LEAN && (CN == 1 || CN == 3) && KIND == 0) {                              // k_warp8_lean1 first; the general kernel then takes what it left
                const warp8::LBox lb = warp8::leanClassify<CN>(a, x0, y0);
                if (lb.kind == warp8::LEAN_OUTSIDE) {
                    for (int tid = 0; tid < 256; tid++) warp8::leanFill<CN>(a, x0, y0, dst, tid);
                    stats[4]++; stats[6]++;
                    continue;
                }
                if (lb.kind != warp8::LEAN_NO) {
                    leanTileEmu<CN>(a, lb, x0, y0, src, lds.data(), dst);
                    stats[4]++; stats[5] += lb.kind == warp8::LEAN_RIM;
                    continue;
                }
            }
            for (int tid = 0; tid < 256; tid++) warp8::phaseA<KIND>(a, x0, y0, lds.data(), tid);
            const warp8::Box b = warp8::boxFromTerms<CN, KIND>(a, reinterpret_cast<const int*>(lds.data() + warp8::OFF_TERMS));
            for (int tid = 0; tid < 256; tid++) warp8::phaseB<CN, KIND>(a, b, x0, y0, src, lds.data(), tid);
            stats[2] += b.all; stats[3] += b.cw == 0;
            for (int tid = 0; tid < 256; tid++) {
                const unsigned redo = warp8::phaseC<CN, KIND, FETCH>(a, b, x0, y0, lds.data(), dst, tid);
                warp8::redoGroups<CN, KIND>(a, b, redo, x0, y0, lds.data(), tid, [&](int x, int y, int X, int Y) {
                    // the coordinates handed to the generic sampler must be the pixel's own (checked against a direct evaluation)
                    int Xr, Yr;
                    if (KIND == 0) { Xr = (warp8::affRowX(a, y) + warp8::affColX(a, x)) >> 5; Yr = (warp8::affRowY(a, y) + warp8::affColY(a, x)) >> 5; }
                    else warp8::perspXY(a, x, y, Xr, Yr);
                    if (X != Xr || Y != Yr) stats[3] += 1000000;
                    std::memcpy(dst + (size_t)y * a.dstep + (size_t)x * CN, expect + (size_t)y * estep + (size_t)x * CN, CN);
                    stats[1]++;
                });
            }
        }
    stats[0] = (long long)a.dw * a.dh - stats[1];
}

// stats: [0] pixels produced from the LDS tile, [1] pixels left to the generic sampler, [2] tiles with an exact all-inside box, [3] tiles with nothing staged,
//        [4] tiles served by the lean path (fetch bit 2, with the term tables of bit 1), [5] of them rim tiles, [6] tiles wholly outside
extern "C" int emu_warp8(const unsigned char* src, size_t sstep, int sw, int sh, unsigned char* dst, size_t dstep, int dw, int dh, int cn, int kind, const double* M,
                         const short* tab, const unsigned char* expect, size_t estep, long long* stats, int constBorder, unsigned cval, int fetch)
{
    warp8::Args a; size_t ldsBytes = 0;
    int bh0 = dh < 16 ? dh : 16;
    const int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;
    if (!warp8::plan(a, cn, kind, M, sw, sh, dw, dh, sstep, dstep, src, dst, bw0, &ldsBytes)) return 1;
    a.constBorder = constBorder; a.cval = cval;
    std::vector<int> tt;
    if (kind == 0 && (fetch & 2)) {                                   // the per-call term tables the launch would build (k_warp8_terms)
        tt.resize(2 * (size_t)dw + 2 * (size_t)dh);
        for (int i = 0; i < dw; i++) { tt[i] = warp8::affColX(a, i); tt[dw + i] = warp8::affColY(a, i); }
        for (int i = 0; i < dh; i++) { tt[2 * dw + i] = warp8::affRowX(a, i); tt[2 * dw + dh + i] = warp8::affRowY(a, i); }
        a.colT = tt.data(); a.rowT = tt.data() + 2 * dw;
    }
    const bool lean = (fetch & 4) && a.colT;
    fetch &= 1;
    stats[0] = stats[1] = stats[2] = stats[3] = stats[4] = stats[5] = stats[6] = 0;
    if (lean && kind == 0 && cn == 1) { run<1, 0, 1, true>(a, ldsBytes, src, dst, tab, expect, estep, stats); return 0; }
    if (lean && kind == 0 && cn == 3) { run<3, 0, 1, true>(a, ldsBytes, src, dst, tab, expect, estep, stats); return 0; }
#define RUN(CN_, K_, F_) run<CN_, K_, F_>(a, ldsBytes, src, dst, tab, expect, estep, stats)
    if (kind == 0) { if (cn == 1) { if (fetch) RUN(1, 0, 1); else RUN(1, 0, 0); } else if (cn == 3) { if (fetch) RUN(3, 0, 1); else RUN(3, 0, 0); } else RUN(4, 0, 0); }
    else           { if (cn == 1) { if (fetch) RUN(1, 1, 1); else RUN(1, 1, 0); } else if (cn == 3) { if (fetch) RUN(3, 1, 1); else RUN(3, 1, 0); } else RUN(4, 1, 0); }
#undef RUN
    return 0;
}


// ---- bilinear resize on the lean machinery (k_resize8_lean): tables as k_resize8_terms builds them, every tile through rzClassify / leanLoad / leanStore / rzRows ----
template <int CN, int LW, int NR>
static void rzTileT(const warp8::Args& a, const warp8::LBox& b, int x0, int y0, const unsigned char* src, unsigned char* lds, unsigned char* dst)
{
    static uint32_t v[256][NR];
    const bool rim = b.kind == warp8::LEAN_RIM;
    for (int tid = 0; tid < 256; tid++) { if (rim) warp8::leanLoad<CN, LW, NR, true>(a, b, src, tid, v[tid]); else warp8::leanLoad<CN, LW, NR, false>(a, b, src, tid, v[tid]); }
    for (int tid = 0; tid < 256; tid++) warp8::leanStore<CN, LW, NR>(a, b, lds, tid, v[tid]);
    for (int tid = 0; tid < 256; tid++) { warp8::RzRowT rt; warp8::rzRowTerms<CN>(a, y0, tid, rt); warp8::rzRows<CN>(a, b, x0, y0, lds, dst, tid, rt); }
}
template <int CN>
static int rzRun(const warp8::Args& a, size_t ldsBytes, const unsigned char* src, unsigned char* dst, long long* stats)
{
    std::vector<unsigned char> lds(ldsBytes + 64);
    for (int ty = 0; ty < a.gy; ty++)
        for (int tx = 0; tx < a.gx; tx++) {
            const int x0 = tx * warp8::TW, y0 = ty * a.th;
            std::memset(lds.data(), 0xA5, lds.size());
            const warp8::LBox b = warp8::rzClassify<CN>(a, x0, y0);
            if (b.kind == warp8::LEAN_NO) { stats[1]++; continue; }
            stats[0]++; stats[2] += b.kind == warp8::LEAN_RIM;
#define LT(LW_, NR_) if (a.leanLW == LW_ && a.leanNR == NR_) rzTileT<CN, LW_, NR_>(a, b, x0, y0, src, lds.data(), dst)
#define LTS(LW_) LT(LW_, 6); LT(LW_, 10); LT(LW_, 14); LT(LW_, 20)
            LTS(16); LTS(32); LTS(64); LTS(128); LTS(256);
#undef LTS
#undef LT
        }
    return 0;
}
// stats: [0] tiles served, [1] tiles declined (their pixels stay untouched), [2] tiles through the predicated loader
extern "C" int emu_resize8(const unsigned char* src, size_t sstep, int sw, int sh, unsigned char* dst, size_t dstep, int dw, int dh, int cn,
                           double inv_scale_x, double inv_scale_y, int areaMode, long long* stats)
{
    warp8::Args a; size_t ldsBytes = 0;
    if (!warp8::planResize(a, cn, sw, sh, dw, dh, sstep, dstep, src, dst, 1. / inv_scale_x, inv_scale_x, 1. / inv_scale_y, inv_scale_y, areaMode, &ldsBytes)) return 1;
    std::vector<int> tt(2 * (size_t)dw + 4 * (size_t)dh);
    for (int i = 0; i < dw; i++) { int sx; uint32_t a01; warp8::rzColTerm(a, i, sx, a01); tt[i] = sx; tt[dw + i] = (int)a01; }
    for (int i = 0; i < dh; i++) { int y0, y1; uint32_t b0, b1; warp8::rzRowTerm(a, i, y0, y1, b0, b1); int* r = tt.data() + 2 * dw; r[i] = y0; r[dh + i] = y1; r[2 * dh + i] = (int)b0; r[3 * dh + i] = (int)b1; }
    a.colT = tt.data(); a.rowT = tt.data() + 2 * dw;
    stats[0] = stats[1] = stats[2] = 0;
    return cn == 1 ? rzRun<1>(a, ldsBytes, src, dst, stats) : rzRun<3>(a, ldsBytes, src, dst, stats);
}

// ---- the bicubic tile path (boxFromTerms4 / phaseC4 of warp8.h; kernel k_warp8_cubic): plan with the bicubic margin, per-call term tables, the Q15 table laid
// out as the kernel lays it out in LDS (12 dwords per entry, 8 used).  stats: [0] pixels from the LDS tile, [1] pixels left to the generic sampler (copied from
// `expect`), [2] tiles with an all-inside box, [3] tiles with nothing staged (+ 1e6 per coordinate mismatch handed to the sampler)
template <int CN>
static void cubicRun(const warp8::Args& a, size_t ldsBytes, const unsigned char* src, unsigned char* dst, const uint32_t* wq, const unsigned char* expect, size_t estep, long long* stats)
{
    std::vector<unsigned char> lds(ldsBytes + 64);
    for (int ty = 0; ty < a.gy; ty++)
        for (int tx = 0; tx < a.gx; tx++) {
            const int x0 = tx * warp8::TW, y0 = ty * a.th;
            std::memset(lds.data(), 0xA5, lds.size());
            for (int tid = 0; tid < 256; tid++) warp8::phaseA<0>(a, x0, y0, lds.data(), tid);
            const warp8::Box b = warp8::boxFromTerms4<CN>(a, reinterpret_cast<const int*>(lds.data() + warp8::OFF_TERMS));
            for (int tid = 0; tid < 256; tid++) warp8::phaseB<CN, 0>(a, b, x0, y0, src, lds.data(), tid);
            stats[2] += b.all; stats[3] += b.cw == 0;
            for (int tid = 0; tid < 256; tid++) {
                const unsigned redo = warp8::phaseC4<CN>(a, b, x0, y0, lds.data(), wq, 12, dst, tid);
                warp8::redoGroups<CN, 0>(a, b, redo, x0, y0, lds.data(), tid, [&](int x, int y, int X, int Y) {
                    const int Xr = (warp8::affRowX(a, y) + warp8::affColX(a, x)) >> 5, Yr = (warp8::affRowY(a, y) + warp8::affColY(a, x)) >> 5;
                    if (X != Xr || Y != Yr) stats[3] += 1000000;
                    std::memcpy(dst + (size_t)y * a.dstep + (size_t)x * CN, expect + (size_t)y * estep + (size_t)x * CN, CN);
                    stats[1]++;
                });
            }
        }
    stats[0] = (long long)a.dw * a.dh - stats[1];
}

extern "C" int emu_warp8_cubic(const unsigned char* src, size_t sstep, int sw, int sh, unsigned char* dst, size_t dstep, int dw, int dh, int cn, const double* M,
                               const short* tabQ15 /* [1024][16] */, const unsigned char* expect, size_t estep, long long* stats, int constBorder, unsigned cval)
{
    warp8::Args a; size_t ldsBytes = 0;
    if (cn != 1 && cn != 3) return 1;
    if (!warp8::plan(a, cn, 0, M, sw, sh, dw, dh, sstep, dstep, src, dst, 1, &ldsBytes, 3)) return 1;
    a.constBorder = constBorder; a.cval = cval;
    std::vector<int> tt(2 * (size_t)dw + 2 * (size_t)dh);
    for (int i = 0; i < dw; i++) { tt[i] = warp8::affColX(a, i); tt[dw + i] = warp8::affColY(a, i); }
    for (int i = 0; i < dh; i++) { tt[2 * dw + i] = warp8::affRowX(a, i); tt[2 * dw + dh + i] = warp8::affRowY(a, i); }
    a.colT = tt.data(); a.rowT = tt.data() + 2 * dw;
    std::vector<uint32_t> wq(1024 * 12, 0xDEADBEEFu);
    for (int i = 0; i < 1024 * 8; i++) std::memcpy(&wq[(size_t)(i >> 3) * 12 + (i & 7)], reinterpret_cast<const unsigned char*>(tabQ15) + 4 * (size_t)i, 4);
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (cn == 1) cubicRun<1>(a, ldsBytes, src, dst, wq.data(), expect, estep, stats); else cubicRun<3>(a, ldsBytes, src, dst, wq.data(), expect, estep, stats);
    return 0;
}
