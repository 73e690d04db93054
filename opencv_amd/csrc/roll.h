// roll.h -- the register-rolling stencil skeleton shared by the HBM-bound 8-bit stencil kernels (measured design of the
// headline Gaussian, smooth.hip k_binomial_roll2):
//   * one WAVE = one work item: a strip of 64 chunks of CB bytes (CB = 16: 1 KiB of a row; CB = 8 for kernels whose per-pixel
//     register state is too large for 16 pixels per lane) x a segment of rows x a frame, work items
//     ordered strip -> segment -> frame so that resident waves sweep memory almost linearly;
//   * a lane owns CB consecutive bytes of a row (one dwordx4 / dwordx2 load) and walks the segment row by row,
//     keeping the rows it still needs in registers; loads run ahead through a ring (a slot is refilled right after use);
//   * +-RX*cn neighbour bytes: DPP wave_shr / wave_shl from the adjacent lane; the two lanes at the wave edge get theirs
//     from one 4-byte side load per row (two 64-byte sectors per wave); image-border halos are rebuilt from the lane's own
//     16 bytes with wave-uniform v_perm selectors; rows outside the image are resolved to scalars before the loop;
//   * vertically adjacent segments walk in opposite directions and sit on the same XCD (alt), so the rows they share are
//     in L2 when the second one asks.
//   * a row whose byte length is not a multiple of CB ("ragged"): the last chunk is loaded from the last CB bytes of the row
//     (so no byte outside the row is ever touched), its logical content -- the valid bytes followed by the border bytes its
//     neighbours and its own outputs need -- is rebuilt from that load with the same kind of v_perm selectors, and its store is
//     element-wise for the valid bytes only.  Everything else is unchanged, so aligned images pay nothing.
// Requirements (checked on the host): W*cn >= CB, W > RX, (RX+1)*cn <= CB, border in {CONSTANT,
// REPLICATE, REFLECT, REFLECT_101}.
#pragma once
#include "rt.h"
#include <cstdlib>

namespace roll {

template <int RX, int CN, int CB = 16> struct Cfg {
    static constexpr int HB = RX * CN;            // halo bytes per side
    static constexpr int HD = (HB + 3) / 4;       // halo dwords per side
    static constexpr int MD = CB / 4;             // dwords of the lane's own chunk
    static constexpr int NW = MD + 2 * HD;        // dwords of the assembled window
};

template <int HD, int MD = 4> struct Raw { uint32_t m[MD]; uint32_t side[HD]; };

// A window of the image (a cv::Mat submatrix with real pixels around it, the HAL's offset_x / offset_y / full_width / full_height contract):
// the kernel is handed the PARENT image -- chunks lie on the parent's rows, halos and borders are the parent's -- and produces only the
// window: chunks that overlap its byte range [x0b, x1b) of a row, rows [y0, y1); the first / last chunk store just their bytes inside it.
// `dst` is the window's own origin.  x1b < 0: the whole image.
struct Win { int x0b, x1b, y0, y1; };
__host__ __device__ inline Win wholeImage() { Win w = {0, -1, 0, -1}; return w; }

template <int HD, int MD = 4> struct Edge { uint32_t la[HD], lb[HD], lc[HD], ra[HD], rb[HD], rc[HD], oa[MD], ob[MD], oc[MD]; };

__device__ __forceinline__ void selSetByte(uint32_t& a, uint32_t& b, uint32_t& c, int j, int idx /* 0..15 or <0 */)
{
    const uint32_t sh = 8u * (uint32_t)j, clr = ~(0xffu << sh);
    uint32_t va = 0x0cu, vb = 0x0cu, vc = 0x0cu;
    if (idx >= 8)      { vb = (uint32_t)(idx - 8); vc = 4u + (uint32_t)j; }
    else if (idx >= 0) { va = (uint32_t)idx;       vc = (uint32_t)j; }
    a = (a & clr) | (va << sh); b = (b & clr) | (vb << sh); c = (c & clr) | (vc << sh);
}

// four bytes picked out of the lane's own chunk (selectors from selSetByte; an 8-byte chunk only needs the first perm)
template <int MD>
__device__ __forceinline__ uint32_t gatherOwn(const uint32_t (&m)[MD], uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t t1 = __builtin_amdgcn_perm(m[1], m[0], a);
    if (MD == 2) return t1;
    const uint32_t t2 = __builtin_amdgcn_perm(m[MD - 1], m[MD - 2], b);
    return __builtin_amdgcn_perm(t2, t1, c);
}

template <int MD> __device__ __forceinline__ void loadChunk(uint32_t (&m)[MD], const uchar* p)
{
    if (MD == 4) { const uint4 v = *reinterpret_cast<const uint4*>(p); m[0] = v.x; m[1] = v.y; m[MD - 2] = v.z; m[MD - 1] = v.w; }
    else { const uint2 v = *reinterpret_cast<const uint2*>(p); m[0] = v.x; m[1] = v.y; }
}

template <int RX, int RY, int CN, int CB = 16>
struct Ctx {
    static constexpr int HD = Cfg<RX, CN, CB>::HD, HB = Cfg<RX, CN, CB>::HB, NW = Cfg<RX, CN, CB>::NW, MD = CB / 4;
    typedef Raw<HD, MD> RawT;
    const uchar* src; size_t sstep;
    int H, lane, c, nchunks, mainOff, sideOff, y0, y1, nrows, up, frame;
    bool active, hasFirst, hasLast, isLastChunk, rag;
    int vb;                                            // valid bytes of the last chunk (CB unless the row is ragged)
    int wx0, wx1, wy0;                                 // window: byte range of a row that is stored, first row (whole image: 0, W*CN, 0)
    int tChunks, tByte0;                               // transposed store (below): chunks of this strip that are stored whole (0: the per-lane store is used), first byte of the strip
    uchar* tlds;                                       // this wave's LDS scratch for it (useLds), or nullptr
    Edge<HD, MD> es;
    int rowBelow[RY > 0 ? RY : 1], rowAbove[RY > 0 ? RY : 1];

    // decode the work item of this wave; false when the wave has nothing to do
    __device__ __forceinline__ bool init(const uchar* s, size_t ss, size_t sframe, int W, int H_, int nchunks_, int nstrips, int segRows, int nseg,
                                         int nframes, int border, int alt, Win win = wholeImage())
    {
        // nchunks_ = chunks of a PARENT row; nstrips / nseg = strips of 64 chunks over the window's chunk range / segments over its rows
        if (win.x1b < 0) { win.x0b = 0; win.x1b = W * CN; win.y0 = 0; win.y1 = H_; }
        wx0 = win.x0b; wx1 = win.x1b; wy0 = win.y0;
        const int cA = win.x0b / CB, cB = (win.x1b - 1) / CB;
        lane = threadIdx.x & 63;
        const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
        const int strip = wid % nstrips;
        const int t0 = wid / nstrips;
        int seg = t0 % nseg;
        frame = t0 / nseg;
        if (frame >= nframes) return false;
        up = 0;
        if (alt) {
            // alt = 1: partners 8 work items apart (4 strips per row: 32 waves = 8 workgroups = one turn of the XCD round-robin); alt = D > 1: D work items apart
            // (kernels with fewer strips per row pass 32 / nstrips)
            const int D = alt == 1 ? 8 : alt, G2 = 2 * D;
            const int g = seg / G2, j = seg - g * G2;
            if ((g + 1) * G2 <= nseg) seg = g * G2 + 2 * (j % D) + j / D;
            up = seg & 1;
        }
        H = H_; nchunks = nchunks_;
        src = s + (size_t)frame * sframe; sstep = ss;
        const int c0 = cA + strip * 64;
        c = c0 + lane;
        y0 = win.y0 + seg * segRows; y1 = min(win.y1, y0 + segRows); nrows = y1 - y0;
        const bool live = c < nchunks;                 // the lane has a chunk of its own in the parent row (it may lie right of the window: it then
                                                       // only feeds its left neighbour's halo)
        active = c <= cB; hasFirst = c0 == 0; hasLast = c0 + 64 >= nchunks; isLastChunk = c == nchunks - 1;
        vb = W * CN - CB * (nchunks - 1);
        rag = vb < CB;
        // the strip stores whole chunks only (no window edge inside it, no ragged last chunk): its row piece is min(64, chunks left) * CB * OUTB contiguous bytes
        tByte0 = CB * c0 - win.x0b;
        tChunks = (win.x0b % CB == 0 && (win.x1b % CB == 0 || c0 + 64 <= cB) && !(rag && hasLast)) ? min(64, cB + 1 - c0) : 0;
        tlds = nullptr;
        mainOff = CB * (live ? c : nchunks - 1);
        if (rag && (!live || isLastChunk)) mainOff = W * CN - CB;        // the last CB bytes of the row
        const int leftOff = c0 > 0 ? CB * c0 - 4 * HD : 0;
        const int rightOff = c0 + 64 < nchunks ? CB * (c0 + 64) : (rag ? W * CN - CB : CB * (nchunks - 1));   // (last strip: never used, only legal)
        sideOff = lane < 32 ? leftOff : rightOff;
#pragma unroll
        for (int d = 0; d < HD; d++) { es.la[d] = es.lb[d] = es.lc[d] = es.ra[d] = es.rb[d] = es.rc[d] = 0x0c0c0c0cu; }
        if (hasFirst) {
#pragma unroll
            for (int t = 0; t < HB; t++) {
                const int bt = t - HB;
                const int px = (bt - (CN - 1)) / CN;               // floor(bt / CN)
                const int sp = mi355_borderInterpolate(px, W, border);
                const int pos = 4 * HD - HB + t;
                selSetByte(es.la[pos >> 2], es.lb[pos >> 2], es.lc[pos >> 2], pos & 3, sp < 0 ? -1 : sp * CN + (bt - px * CN));
            }
        }
#pragma unroll
        for (int d = 0; d < MD; d++) { es.oa[d] = es.ob[d] = es.oc[d] = 0x0c0c0c0cu; }
        if (hasLast && !rag) {
#pragma unroll
            for (int t = 0; t < HB; t++) {
                const int sp = mi355_borderInterpolate(W + t / CN, W, border);
                selSetByte(es.ra[t >> 2], es.rb[t >> 2], es.rc[t >> 2], t & 3, sp < 0 ? -1 : sp * CN + (t % CN) - CB * (nchunks - 1));
            }
        }
        if (hasLast && rag) {
            // logical bytes j of the last chunk and its right halo: j < vb -> row byte CB*(nchunks-1)+j, then the border bytes;
            // all expressed as positions in the load of the row's last CB bytes (only the first HB border bytes can matter)
            const int base = W * CN - CB;
#pragma unroll
            for (int j = 0; j < CB + HB; j++) {
                int idx = -1;
                if (j < vb) idx = CB - vb + j;
                else if (j - vb < HB) {
                    const int t = j - vb;
                    const int sp = mi355_borderInterpolate(W + t / CN, W, border);
                    idx = sp < 0 ? -1 : sp * CN + (t % CN) - base;
                }
                if (j < CB) selSetByte(es.oa[j >> 2], es.ob[j >> 2], es.oc[j >> 2], j & 3, idx);
                else selSetByte(es.ra[(j - CB) >> 2], es.rb[(j - CB) >> 2], es.rc[(j - CB) >> 2], (j - CB) & 3, idx);
            }
        }
#pragma unroll
        for (int i = 0; i < RY; i++) { rowBelow[i] = mi355_borderInterpolate(H + i, H, border); rowAbove[i] = mi355_borderInterpolate(-1 - i, H, border); }
        return true;
    }

    // logical row j of the segment (walking order) -> image row
    __device__ __forceinline__ int gy(int j) const { return up ? y1 - 1 - j : y0 + j; }
    // image row g in [-RY, H+RY) -> source row index, or -1 for a BORDER_CONSTANT row
    __device__ __forceinline__ int rowIdx(int g) const
    {
        int ry = g;
#pragma unroll
        for (int i = 0; i < RY; i++) { ry = (g == H + i) ? rowBelow[i] : ry; ry = (g == -1 - i) ? rowAbove[i] : ry; }
        return ry;
    }
    // issue the loads of logical row j (clamped so that the address is always legal); valid = 0 for a constant-border row
    __device__ __forceinline__ void issue(RawT& r, int j, int& valid) const
    {
        const int ry = rowIdx(gy(min(j, nrows - 1 + RY)));
        valid = ry >= 0;
        const uchar* row = src + (size_t)max(ry, 0) * sstep;
        loadChunk<MD>(r.m, row + mainOff);
#pragma unroll
        for (int d = 0; d < HD; d++) r.side[d] = *reinterpret_cast<const uint32_t*>(row + sideOff + 4 * d);
    }
    // same, for an explicit image row g in [-RY, H+RY)
    __device__ __forceinline__ void issueImg(RawT& r, int g, int& valid) const
    {
        const int ry = rowIdx(g);
        valid = ry >= 0;
        const uchar* row = src + (size_t)max(ry, 0) * sstep;
        loadChunk<MD>(r.m, row + mainOff);
#pragma unroll
        for (int d = 0; d < HD; d++) r.side[d] = *reinterpret_cast<const uint32_t*>(row + sideOff + 4 * d);
    }
    // the lane's window of one row as dwords: X[0..HD) left halo, X[HD..HD+MD) own bytes, X[HD+MD..NW) right halo
    __device__ __forceinline__ void window(uint32_t (&X)[NW], const RawT& r) const
    {
        uint32_t mv[MD];
#pragma unroll
        for (int d = 0; d < MD; d++) mv[d] = r.m[d];
        if (hasLast && rag) {
#pragma unroll
            for (int d = 0; d < MD; d++) { const uint32_t g = gatherOwn<MD>(r.m, es.oa[d], es.ob[d], es.oc[d]); mv[d] = isLastChunk ? g : mv[d]; }
        }
        uint32_t hl[HD], hr[HD], hb[HD];
#pragma unroll
        for (int d = 0; d < HD; d++) { hl[d] = r.side[d]; hr[d] = r.side[d]; }
        if (hasFirst) {
#pragma unroll
            for (int d = 0; d < HD; d++) hl[d] = gatherOwn<MD>(r.m, es.la[d], es.lb[d], es.lc[d]);
        }
        if (hasLast) {
#pragma unroll
            for (int d = 0; d < HD; d++) hb[d] = gatherOwn<MD>(r.m, es.ra[d], es.rb[d], es.rc[d]);
        }
#pragma unroll
        for (int d = 0; d < HD; d++) {
            X[d] = __builtin_amdgcn_update_dpp(hl[d], mv[MD - HD + d], 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 keeps hl
            uint32_t rr = __builtin_amdgcn_update_dpp(hr[d], mv[d], 0x130, 0xf, 0xf, false);       // wave_shl:1, lane 63 keeps hr
            if (hasLast) rr = isLastChunk ? hb[d] : rr;
            X[HD + MD + d] = rr;
        }
#pragma unroll
        for (int d = 0; d < MD; d++) X[HD + d] = mv[d];
    }
    // the lane's outputs of image row y (OUTB bytes per source byte) -> memory; a ragged last chunk writes its valid elements only, the first / last
    // chunk of a window the elements inside it
    // LDS scratch of the transposed store: NPC = MD * OUTB / 4 pieces of 16 bytes per lane, one region of 64 pieces (+ 128 bytes of skew) per piece index
    template <int OUTB> static constexpr int tldsBytesPerWave() { return (MD * OUTB / 4) * (1024 + 128); }
    __device__ __forceinline__ void useLds(uchar* blockScratch, int bytesPerWave) { tlds = blockScratch + (threadIdx.x >> 6) * bytesPerWave; }

    template <int OUTB>
    __device__ __forceinline__ void store(uchar* __restrict__ dst, size_t dstep, int y, const uint32_t (&o)[MD * OUTB]) const
    {
        // A lane that owns more than 16 output bytes of a row (8U -> 16S / 32F: 32 or 64) used to write them as NPC 16-byte stores at a lane stride of 16 * NPC bytes: every
        // store instruction then touches each 64-byte line partially, and HBM delivers half of what contiguous stores get (profiles/r05_store_geometry_probe.txt: 1 : 4 mix
        // 0.45 of the roofline with 2 x 16 B per lane against 0.67 with 1 x 16 B; fill only 0.31 against 0.70).  So the wave TRANSPOSES its row piece through LDS first:
        // piece h of lane L goes to region h at 16 L (contiguous ds_write_b128), store instruction q then takes piece 64 q + L of the row piece -- piece (64 q + L) % NPC of
        // lane (64 q + L) / NPC -- back (ds_read_b128; the 128-byte skew between regions makes the 16-lane groups of the read conflict-free for NPC = 2) and writes 1 KiB of
        // the row contiguously.  The scratch is private to the wave and LDS operations of one wave execute in order: no barrier, no double buffer.
        if constexpr (MD * OUTB > 4) {
            if (transposes()) { storeT<OUTB>(dst, dstep, y, o); return; }
        }
        storeLanes<OUTB>(dst, dstep, y, o);
    }
    // wave-uniform: this wave's row pieces go through the transposed store (kernels may hoist the test and call storeT / storeLanes themselves)
    __device__ __forceinline__ bool transposes() const { return tlds != nullptr && tChunks > 0; }
    template <int OUTB>
    __device__ __forceinline__ void storeT(uchar* __restrict__ dst, size_t dstep, int y, const uint32_t (&o)[MD * OUTB]) const
    {
        constexpr int NPC = MD * OUTB / 4, RS = 1024 + 128;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h = 0; h < NPC; h++)
            *reinterpret_cast<u32x4*>(tlds + h * RS + 16 * lane) = u32x4{o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uchar* rowp = dst + (size_t)(y - wy0) * dstep + (ptrdiff_t)tByte0 * OUTB;
        const int np = tChunks * NPC;
#pragma unroll
        for (int q = 0; q < NPC; q++) {
            const int pc = 64 * q + lane;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tlds + (pc % NPC) * RS + 16 * (pc / NPC));
            if (pc < np) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(rowp) + pc);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next row's writes stay behind these reads
        __builtin_amdgcn_wave_barrier();
    }
    template <int OUTB>
    __device__ __forceinline__ void storeLanes(uchar* __restrict__ dst, size_t dstep, int y, const uint32_t (&o)[MD * OUTB]) const
    {
        if (!active) return;
        const int b0 = CB * c;                                                // first byte of the chunk in the parent row
        uchar* p = dst + (size_t)(y - wy0) * dstep + (ptrdiff_t)(b0 - wx0) * OUTB;
        const int lo = max(wx0 - b0, 0), hi = min(min(wx1 - b0, CB), (rag && isLastChunk) ? vb : CB);
        if (lo == 0 && hi == CB) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if constexpr ((MD * OUTB) % 4 == 0) {
#pragma unroll
                for (int q = 0; q < MD * OUTB / 4; q++)
                    __builtin_nontemporal_store(u32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]}, reinterpret_cast<u32x4*>(p) + q);
            } else
                __builtin_nontemporal_store(u32x2{o[0], o[1]}, reinterpret_cast<u32x2*>(p));
        } else {
#pragma unroll
            for (int e = 0; e < CB; e++) {
                if (e >= lo && e < hi) {
                    if constexpr (OUTB == 4) reinterpret_cast<uint32_t*>(p)[e] = o[e];
                    else if constexpr (OUTB == 2) reinterpret_cast<unsigned short*>(p)[e] = (unsigned short)(o[e >> 1] >> (16 * (e & 1)));
                    else p[e] = (uchar)(o[e >> 2] >> (8 * (e & 3)));
                }
            }
        }
    }
};

// ---- packed byte planes --------------------------------------------------------------------------------------------
// A window of NW dwords as two planes of packed u16 pairs: E[d] = (byte 4d, byte 4d+2), O[d] = (byte 4d+1, byte 4d+3) of
// window dword d (window dword HD is the lane's first own dword).  pairAt<Q,S,HD>(E,O,k) = the bytes at distance S from the
// output pair (plane Q, own dword k), i.e. window bytes (4(k+HD)+Q+S, +2), as a packed pair: a plane element or one alignbit.
constexpr int floordiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
constexpr int mod2(int a) { return ((a % 2) + 2) % 2; }
template <int NW> __device__ __forceinline__ void planes(uint32_t (&E)[NW], uint32_t (&O)[NW], const uint32_t (&X)[NW])
{
#pragma unroll
    for (int d = 0; d < NW; d++) { E[d] = X[d] & 0x00ff00ffu; O[d] = (X[d] >> 8) & 0x00ff00ffu; }
}
template <int Q, int S, int HD>
__device__ __forceinline__ uint32_t pairAt(const uint32_t* E, const uint32_t* O, int k)
{
    constexpr int q2 = mod2(Q + S);
    constexpr int f = floordiv2(Q + S);
    const uint32_t* P = q2 ? O : E;
    if constexpr (mod2(f) == 0) return P[k + f / 2 + HD];
    else { constexpr int lo = floordiv2(f - 1); return __builtin_amdgcn_alignbit(P[k + lo + 1 + HD], P[k + lo + HD], 16); }
}

// host side: eligibility of the rolling path and launch geometry
inline bool eligible(const void* s, size_t ss, size_t sf, const void* d, size_t ds, size_t df, int W, int cn, int rx, int border, int cb = 16)
{
    // row starts need no alignment: gfx950 serves unaligned dwordx4 global accesses (at some cost in sectors touched)
    (void)s; (void)ss; (void)sf; (void)d; (void)ds; (void)df;
    // halos come from the neighbouring chunk (rx*cn <= cb) and, at the image border, from pixels 0..rx of the lane's OWN chunk
    // (for a ragged row: from the row's last cb bytes)
    if (W * cn < cb || W <= rx || (rx + 1) * cn > cb) return false;
    const int nchunks = (W * cn + cb - 1) / cb, vb = W * cn - cb * (nchunks - 1);
    if (vb < cb && nchunks % 64 == 1 && vb < 4 * ((rx * cn + 3) / 4)) return false;   // the ragged chunk opens a strip: the previous strip's side load would cross the row end
    return border == mi355::B_CONSTANT || border == mi355::B_REPLICATE || border == mi355::B_REFLECT || border == mi355::B_REFLECT_101;
}

struct Geom { int nchunks, nstrips, seg, nseg; unsigned blocks; };
// `wantWaves`: work items (waves) the launch should at least consist of, when the image has enough rows for that at >= minSeg rows each.
// `win`: a window of the (W x H) image (see Win): strips cover the chunks that overlap it, segments its rows.
inline Geom geometry(int W, int H, int cn, int nframes, int bestSeg, int minSeg, int cb = 16, int wantWaves = 2048, Win win = wholeImage())
{
    Geom g;
    g.nchunks = mi355::divUp(W * cn, cb); g.nstrips = mi355::divUp(g.nchunks, 64);
    if (win.x1b >= 0) { g.nstrips = mi355::divUp((win.x1b - 1) / cb - win.x0b / cb + 1, 64); H = win.y1 - win.y0; }
    if (const char* e = std::getenv("MI355CV_ROLL_SEG")) { const int v = atoi(e); if (v > 0) bestSeg = v; }     // tuning experiments
    long long per = (long long)g.nstrips * nframes;
    if (const char* e = std::getenv("MI355CV_ROLL_WAVES")) { const int v = atoi(e); if (v > 0) wantWaves = v; }
    long long wantSeg = (wantWaves + per - 1) / per;
    int seg = (int)((H + wantSeg - 1) / wantSeg);
    if (seg > bestSeg) seg = bestSeg;
    if (seg < minSeg) seg = minSeg;
    if (seg > H) seg = H;
    g.seg = seg; g.nseg = mi355::divUp(H, seg);
    g.blocks = (unsigned)(((long long)g.nstrips * g.nseg * nframes + 3) / 4);
    return g;
}

} // namespace roll
