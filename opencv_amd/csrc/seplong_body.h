// seplong_body.h -- the three phases of k_seplong (seplong.hip) as functions of the thread id.  Everything here is __host__ __device__ and free of wave
// intrinsics so that tests/hostemu/seplong_emu.cpp runs the very same staging, row-pass and column-pass lines on the CPU, thread by thread with the barriers
// replaced by loop boundaries, against the pinned restatement (the GPU adds only the launch).
//
// A workgroup of 256 lanes owns a strip of TP pixels and walks DOWN a segment of rows, RB = 16 source rows per step:
//   1. stage      RB source rows of the strip (+ nx - 1 pixels of halo, borders resolved here, channels de-interleaved into planes) into LDS as 4-byte values;
//   2. row pass   each lane filters 4 consecutive pixels of one plane of one row: a sliding 8-register window refilled by one 16-byte LDS read per 4 taps, taps
//                 from scalar loads (uniform index); the sums go to an LDS ring of NR >= ny - 1 + RB rows;
//   3. column pass  each lane owns 2 neighbouring pixels x 4 consecutive output rows: one 8-byte LDS read per ring row feeds 8 multiply-adds (windows sliding up
//                 and down the ring for the symmetric / anti-symmetric pair forms); results leave in the destination depth.
// The arithmetic -- the order of every chain, where the reference's vector body and scalar tail differ -- is that of the one-thread-per-output kernels this
// replaces (k_sepfilter_generic, k_sepfixed_generic), which tests/test_filters_gpu.py pins to the reference bit for bit:
//   mode 0  float: row s = kx[0] * S[0], s = fma(kx[i], S[i], s) (RowFilter, filter.simd.hpp:2386); column in the (anti)symmetric pair form
//           s = fma(ky[c], r[c], delta), s = fma(ky[c+k], r[c+k] +- r[c-k], s) (SymmColumnFilter :2679-2751) or the plain chain (ColumnFilter :2609)
//   mode 1  CV_8U -> CV_8U, taps x 2^8 as int32: exact row sums; column in FLOAT for the elements the reference's 16-lane loop reaches
//           (SymmColumnVec_32s8u :1011-1085), (v + 2^15) >> 16 for the row tail
//   mode 2  CV_8U -> CV_16S, integer taps: exact int32 sums, saturate
//   mode 3  CV_8U -> CV_8U, cv::GaussianBlur's Q8.8 taps with sum <= 256 per axis (fixedSmoothInvoker, smooth.simd.hpp:1926): (sum + 2^15) >> 16
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#if defined(__HIPCC__)
#define SL_HD __host__ __device__ __forceinline__
#else
#define SL_HD inline
#endif

namespace seplong {

enum { SL_8U = 0, SL_16U = 2, SL_16S = 3, SL_32F = 5 };          // CV depth codes
constexpr int RB = 16;                                            // source rows staged / output rows emitted per step

struct Geom {
    int W, H, sdepth, ddepth, fullW, fullH, offX, offY, border;
    int nx, ny, ax, ay, symY;
    int SP, RP, NR;                            // stage pitch and ring pitch per plane (4-byte elements), ring rows (a multiple of RB)
    int seg;                                   // output rows per segment
    int vecEnd;                                // mode 1: elements below this index take the float column form
    float deltaF; int deltaI;
    uint32_t bval[4];                          // modes 4 / 5 (erode / dilate): the value of elements outside the image under BORDER_CONSTANT, per channel
};

template <int CN> struct StripOf { static constexpr int TP = CN == 1 ? 128 : CN == 2 ? 64 : CN == 3 ? 40 : 32; };

SL_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
SL_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
SL_HD int imin(int a, int b) { return a < b ? a : b; }
SL_HD int imax(int a, int b) { return a > b ? a : b; }

// borderInterpolate (core/src/copy.cpp:748-793); -1 for BORDER_CONSTANT.  Border codes: 0 constant, 1 replicate, 2 reflect, 3 wrap, 4 reflect_101.
SL_HD int border(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == 1) return p < 0 ? 0 : len - 1;
    if (type == 2 || type == 4) {
        const int delta = type == 4;
        if (len == 1) return 0;
        do { if (p < 0) p = -p - 1 + delta; else p = len - 1 - (p - len) - delta; } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (type == 3) {
        if (p < 0) p -= ((p - len + 1) / len) * len;
        if (p >= len) p %= len;
        return p;
    }
    return -1;
}

SL_HD float ldSrcF(const unsigned char* row, int idx, int depth)
{
    switch (depth) {
    case SL_8U:  return (float)row[idx];
    case SL_16U: return (float)reinterpret_cast<const unsigned short*>(row)[idx];
    case SL_16S: return (float)reinterpret_cast<const short*>(row)[idx];
    default:     return reinterpret_cast<const float*>(row)[idx];
    }
}

// saturate_cast<DT>(float): cvRound (round-half-even) then clamp (core/saturate.hpp:103-142)
SL_HD void stDstF(unsigned char* row, int idx, int depth, float s)
{
    switch (depth) {
    case SL_8U:  { float r = rintf(s); r = fminf(fmaxf(r, 0.f), 255.f); row[idx] = (unsigned char)(int)r; break; }
    case SL_16U: { float r = rintf(s); r = fminf(fmaxf(r, 0.f), 65535.f); reinterpret_cast<unsigned short*>(row)[idx] = (unsigned short)(int)r; break; }
    case SL_16S: { float r = rintf(s); r = fminf(fmaxf(r, -32768.f), 32767.f); reinterpret_cast<short*>(row)[idx] = (short)(int)r; break; }
    default:     reinterpret_cast<float*>(row)[idx] = s;
    }
}

// integer multiply-add on the values of each mode: Q8.8 operands fit 24 bits (full-rate v_mad_u32_u24), the integer kernels of cv::sepFilter2D need all 32
template <int MODE> SL_HD uint32_t imad(uint32_t k, uint32_t v, uint32_t acc)
{
    if constexpr (MODE == 4) { (void)k; return v < acc ? v : acc; }               // erode: the "multiply-add" of a flat rectangle is a minimum
    else if constexpr (MODE == 5) { (void)k; return v > acc ? v : acc; }          // dilate
    else if constexpr (MODE == 3) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umul24(k, v) + acc;
#else
        return (k & 0xffffffu) * (v & 0xffffffu) + acc;
#endif
    } else return k * v + acc;
}

struct V2 { uint32_t x, y; };
struct V4 { uint32_t x, y, z, w; };
SL_HD V2 ld2(const uint32_t* p) { V2 v; __builtin_memcpy(&v, p, 8); return v; }           // 8-byte aligned by construction: one ds_read_b64
SL_HD V4 ld4(const uint32_t* p) { V4 v; __builtin_memcpy(&v, p, 16); return v; }          // 16-byte aligned: one ds_read_b128

// the per-segment constants every phase uses
template <int CN> struct Seg {
    int px0, y0, rows, nsrc, nsteps, spn, fx0; bool xin;
    SL_HD void init(const Geom& g, int strip, int segIdx)
    {
        constexpr int TP = StripOf<CN>::TP;
        px0 = strip * TP; y0 = segIdx * g.seg;
        rows = imin(g.seg, g.H - y0);
        nsrc = rows + g.ny - 1;                          // source rows this segment filters horizontally (row 0 = image row y0 - ay)
        nsteps = (nsrc + RB - 1) / RB;
        spn = (TP + g.nx - 1) * CN;                      // staged elements per row
        fx0 = px0 - g.ax + g.offX;                       // full-image x of the first staged pixel
        xin = fx0 >= 0 && fx0 + TP + g.nx - 1 <= g.fullW;
    }
};

// ---- 1. stage: a wave per row, lanes along the interleaved elements (coalesced), planes in LDS.  Two halves so that the loads of step j + 1 can be in flight while
// step j is filtered: stageLoad issues every load of the step into registers (RPW rows x MMAX elements per lane, nothing depends on them), stageStore parks them in LDS.
// (The first version loaded and stored element by element: 12 dependent global-load latencies per step and nothing else to do meanwhile -- 56 us per 4K frame for 19
// taps, profiles/r06_seplong_first_run_latency_bound.txt.)
constexpr int RPW = RB / 4;                                       // rows per wave and step
// elements per lane and row: the widest staged row is (TP + nx - 1) * CN elements; LONG = kernels beyond 33 taps
template <int CN, bool LONG> struct StageOf { static constexpr int NXMAX = LONG ? 129 : 33, MMAX = ((StripOf<CN>::TP + NXMAX - 1) * CN + 63) / 64; };

template <int MODE, int CN, int MMAX>
SL_HD void stageLoad(const Geom& g, const Seg<CN>& sg, int step, const unsigned char* src, size_t sstep, int tid, uint32_t (&v)[RPW * MMAX])
{
    const int wv = tid >> 6, ln = tid & 63;
    const int n0 = step * RB, rlim = imin(RB, sg.nsrc - n0);
#pragma unroll
    for (int i = 0; i < RPW; i++) {
        const int r = wv + 4 * i;
        const int yy = r < rlim ? border(sg.y0 - g.ay + n0 + r + g.offY, g.fullH, g.border) : -1;
        const unsigned char* row = src + (ptrdiff_t)((yy < 0 ? g.offY : yy) - g.offY) * (ptrdiff_t)sstep;
#pragma unroll
        for (int m = 0; m < MMAX; m++) {
            const int q = ln + 64 * m;
            const int po = q / CN, c = q - po * CN;
            uint32_t val = MODE >= 4 ? g.bval[c] : 0u;
            if (yy >= 0 && q < sg.spn) {
                const int fp = sg.fx0 + po;
                const int xx = sg.xin ? fp : border(fp, g.fullW, g.border);
                if (xx >= 0) {
                    const int idx = (xx - g.offX) * CN + c;
                    if constexpr (MODE == 0) val = f2u(ldSrcF(row, idx, g.sdepth));
                    else val = row[idx];
                }
            }
            v[i * MMAX + m] = val;
        }
    }
}

template <int CN, int MMAX>
SL_HD void stageStore(const Geom& g, const Seg<CN>& sg, int step, uint32_t* S, int tid, const uint32_t (&v)[RPW * MMAX])
{
    const int wv = tid >> 6, ln = tid & 63;
    const int n0 = step * RB, rlim = imin(RB, sg.nsrc - n0);
#pragma unroll
    for (int i = 0; i < RPW; i++) {
        const int r = wv + 4 * i;
        if (r >= rlim) continue;
        uint32_t* Sr = S + r * CN * g.SP;
#pragma unroll
        for (int m = 0; m < MMAX; m++) {
            const int q = ln + 64 * m;
            if (q < sg.spn) { const int po = q / CN, c = q - po * CN; Sr[c * g.SP + po] = v[i * MMAX + m]; }
        }
    }
}

// ---- 2. row pass: 4 consecutive pixels of one plane of one row per lane; taps = kx (float bits or ints)
template <int MODE, int CN>
SL_HD void rowPass(const Geom& g, const Seg<CN>& sg, int step, const uint32_t* S, uint32_t* ring, const uint32_t* kx, int tid)
{
    constexpr int Q4 = StripOf<CN>::TP / 4;
    const int n0 = step * RB, rlim = imin(RB, sg.nsrc - n0);
    const int slot0 = n0 % g.NR;                                           // NR is a multiple of RB: the step's rows are consecutive slots
    for (int t = tid; t < rlim * CN * Q4; t += 256) {
        const int x4 = t % Q4, rc = t / Q4, c = rc % CN, r = rc / CN;
        const uint32_t* Sp = S + (r * CN + c) * g.SP + 4 * x4;
        uint32_t w[8];
        { const V4 a = ld4(Sp); w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; }
        uint32_t acc[4];
        if constexpr (MODE == 0) {
            const float k0 = u2f(kx[0]);
#pragma unroll
            for (int o = 0; o < 4; o++) acc[o] = f2u(k0 * u2f(w[o]));                  // RowFilter: s = kx[0] * S[0], then s += kx[i] * S[i]
        } else {
#pragma unroll
            for (int o = 0; o < 4; o++) { if constexpr (MODE >= 4) acc[o] = w[o]; else acc[o] = imad<MODE>(kx[0], w[o], 0u); }
        }
        for (int g4 = 0; g4 < g.nx; g4 += 4) {
            { const V4 b = ld4(Sp + g4 + 4); w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; }
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const int i = g4 + tt;
                if (i >= 1 && i < g.nx) {
                    const uint32_t k = kx[i];
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        if constexpr (MODE == 0) acc[o] = f2u(__builtin_fmaf(u2f(k), u2f(w[o + tt]), u2f(acc[o])));
                        else acc[o] = imad<MODE>(k, w[o + tt], acc[o]);
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) w[o] = w[o + 4];
        }
        const V4 out = {acc[0], acc[1], acc[2], acc[3]};
        __builtin_memcpy(ring + (slot0 + r) * CN * g.RP + c * g.RP + 4 * x4, &out, 16);
    }
}

// output rows complete after `step`: those whose ny source rows are all in the ring
template <int CN> SL_HD int doneAfter(const Geom& g, const Seg<CN>& sg, int step)
{
    const int staged = imin((step + 1) * RB, sg.nsrc);
    return imin(sg.rows, imax(0, staged - g.ny + 1));
}

// ---- 3. column pass: output rows [done, newDone); ky = float bits or ints, kyS = mode 1's float(ky) * 2^-16
template <int MODE, int CN>
SL_HD void colPass(const Geom& g, const Seg<CN>& sg, int done, int newDone, const uint32_t* ring, const uint32_t* ky, const uint32_t* kyS,
                   unsigned char* dst, size_t dstep, int tid)
{
    constexpr int TP = StripOf<CN>::TP, PP = TP / 2, NPAIR = CN * PP;
    const int p = tid % NPAIR, part = tid / NPAIR;
    const int r0 = done + part * 4;
    if (part >= 4 || r0 >= newDone) return;
    const int NR = g.NR, rs = CN * g.RP;
    const int c = p / PP, q = p - c * PP;
    const uint32_t* rb = ring + c * g.RP + 2 * q;
    auto ld = [&](int slot) -> V2 { return ld2(rb + slot * rs); };
    auto inc = [&](int s) { return s + 1 == NR ? 0 : s + 1; };
    auto dec = [&](int s) { return s == 0 ? NR - 1 : s - 1; };
    const int px = sg.px0 + 2 * q;
    const int e0 = px * CN + c;                                    // element index of the lane's first pixel; the second is e0 + CN
    const bool has0 = px < g.W, has1 = px + 1 < g.W;
    const int nval = imin(4, newDone - r0);
    // which column form: the pair form on floats (mode 0 with symY, mode 1's vector body) and / or the plain chain
    bool pairForm, chain;
    if constexpr (MODE == 0) { pairForm = g.symY != 0; chain = !pairForm; }
    else if constexpr (MODE == 1) { pairForm = g.ny > 1 && e0 < g.vecEnd; chain = g.ny <= 1 || e0 + CN >= g.vecEnd; }
    else { pairForm = false; chain = true; }
    float fs[4][2];
    uint32_t is[4][2];
    if (pairForm) {
        const int half = g.ny / 2;
        V2 C[4], U[4], D[4];
        const int ic = (r0 + g.ay) % NR;
        int i = ic;
#pragma unroll
        for (int o = 0; o < 4; o++) { C[o] = ld(i); i = inc(i); }
        int iu = i, id = dec(ic);
        U[0] = C[1]; U[1] = C[2]; U[2] = C[3]; U[3] = ld(iu);
        D[1] = C[0]; D[2] = C[1]; D[3] = C[2]; D[0] = ld(id);
        const float kc = u2f(MODE == 1 ? kyS[g.ay] : ky[g.ay]);
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if constexpr (MODE == 1) {
                fs[o][0] = __builtin_fmaf((float)(int)C[o].x, kc, g.deltaF);
                fs[o][1] = __builtin_fmaf((float)(int)C[o].y, kc, g.deltaF);
            } else {
                fs[o][0] = g.symY == 1 ? __builtin_fmaf(kc, u2f(C[o].x), g.deltaF) : g.deltaF;
                fs[o][1] = g.symY == 1 ? __builtin_fmaf(kc, u2f(C[o].y), g.deltaF) : g.deltaF;
            }
        }
        const bool anti = MODE == 0 && g.symY == 2;
#pragma unroll 4
        for (int k = 1; k <= half; k++) {
            const float kk = u2f(MODE == 1 ? kyS[g.ay + k] : ky[g.ay + k]);
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if constexpr (MODE == 1) {
                    fs[o][0] = __builtin_fmaf((float)(int)(U[o].x + D[o].x), kk, fs[o][0]);
                    fs[o][1] = __builtin_fmaf((float)(int)(U[o].y + D[o].y), kk, fs[o][1]);
                } else {
                    const float ux = u2f(U[o].x), dx = u2f(D[o].x), uy = u2f(U[o].y), dy = u2f(D[o].y);
                    fs[o][0] = __builtin_fmaf(kk, anti ? ux - dx : ux + dx, fs[o][0]);
                    fs[o][1] = __builtin_fmaf(kk, anti ? uy - dy : uy + dy, fs[o][1]);
                }
            }
            iu = inc(iu); id = dec(id);
            U[0] = U[1]; U[1] = U[2]; U[2] = U[3]; U[3] = ld(iu);
            D[3] = D[2]; D[2] = D[1]; D[1] = D[0]; D[0] = ld(id);
        }
    }
    if (chain) {
        // s = delta, then s = fma(ky[j], r[j], s) for j = 0 .. ny - 1 per output (ColumnFilter: s = ky[0] * r[0] + delta first -- the same value); output o of the lane
        // takes ring row m as its tap m - o, so rows 3 .. ny - 1 feed all four outputs and only the two ends of the window are guarded
#pragma unroll
        for (int o = 0; o < 4; o++) { if constexpr (MODE == 0) fs[o][0] = fs[o][1] = g.deltaF; else if constexpr (MODE == 4) is[o][0] = is[o][1] = 0xffffffffu; else if constexpr (MODE == 5) is[o][0] = is[o][1] = 0u; else is[o][0] = is[o][1] = (uint32_t)g.deltaI; }      // (mode 1 may hold its pair-form floats in fs)
        int i = r0 % NR;
        auto tapRow = [&](int m, bool guarded) {
            const V2 v = ld(i);
            i = inc(i);
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int jj = m - o;
                if (!guarded || (jj >= 0 && jj < g.ny)) {
                    const uint32_t k = ky[jj];
                    if constexpr (MODE == 0) {
                        fs[o][0] = __builtin_fmaf(u2f(k), u2f(v.x), fs[o][0]);
                        fs[o][1] = __builtin_fmaf(u2f(k), u2f(v.y), fs[o][1]);
                    } else {
                        is[o][0] = imad<MODE>(k, v.x, is[o][0]);
                        is[o][1] = imad<MODE>(k, v.y, is[o][1]);
                    }
                }
            }
        };
        const int m1 = imin(3, g.ny + 3), m2 = imax(g.ny, m1);
        for (int m = 0; m < m1; m++) tapRow(m, true);
#pragma unroll 4
        for (int m = m1; m < g.ny; m++) tapRow(m, false);
        for (int m = m2; m < g.ny + 3; m++) tapRow(m, true);
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
        if (o >= nval) continue;
        unsigned char* drow = dst + (size_t)(sg.y0 + r0 + o) * dstep;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (!(h == 0 ? has0 : has1)) continue;
            const int e = e0 + h * CN;
            if constexpr (MODE == 0) stDstF(drow, e, g.ddepth, fs[o][h]);
            else if constexpr (MODE == 1) {
                if (g.ny > 1 && e < g.vecEnd) { const float r = rintf(fs[o][h]); drow[e] = (unsigned char)(int)fminf(fmaxf(r, 0.f), 255.f); }
                else { const int r = ((int)is[o][h] + (1 << 15)) >> 16; drow[e] = (unsigned char)(r < 0 ? 0 : r > 255 ? 255 : r); }
            } else if constexpr (MODE == 2) {
                const int a = (int)is[o][h];
                reinterpret_cast<short*>(drow)[e] = (short)(a < -32768 ? -32768 : a > 32767 ? 32767 : a);
            } else if constexpr (MODE >= 4) {
                drow[e] = (unsigned char)is[o][h];
            } else {
                drow[e] = (unsigned char)((is[o][h] + 0x8000u) >> 16);
            }
        }
    }
}

// what the host derives from the call (shared with the emulation): pitches, ring rows, rows per segment; false = more LDS than a CU has
inline bool plan(Geom& g, int cn, int nframes, size_t* ldsBytes, int* nstrips, int* nseg)
{
    const int TP = cn == 1 ? 128 : cn == 2 ? 64 : cn == 3 ? 40 : 32;
    g.SP = TP + ((g.nx + 3) & ~3) + 4;
    g.RP = TP;
    g.NR = (g.ny - 1 + RB + RB - 1) / RB * RB;
    g.vecEnd = (g.W * cn) & ~15;
    *ldsBytes = ((size_t)RB * cn * g.SP + (size_t)g.NR * cn * g.RP) * 4;
    if (*ldsBytes > 160 * 1024) return false;
    *nstrips = (g.W + TP - 1) / TP;
    // rows per segment: enough workgroups for ~4 per CU, but a segment repeats ny - 1 row passes at its top: at least 4 x ny rows where the image allows
    const long long per = (long long)*nstrips * nframes;
    long long want = (1024 + per - 1) / per;
    if (want < 1) want = 1;
    if (want > g.H) want = g.H;
    int seg = (g.H + (int)want - 1) / (int)want;
    const int floorRows = 4 * g.ny > 64 ? 4 * g.ny : 64;
    if (seg < floorRows) seg = floorRows;
    seg = (seg + RB - 1) / RB * RB;
    if (seg > g.H) seg = g.H;
    g.seg = seg;
    *nseg = (g.H + seg - 1) / seg;
    return true;
}

} // namespace seplong
