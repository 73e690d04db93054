// integral.hip -- cv::integral for 8-bit single-channel sources (cv_hal_integral, hal_replacement.hpp:977; sumpixels.simd.hpp) in two passes
// over the pixels instead of three passes over the 4-8x larger sum image.
//
//   S[y+1][x+1] = sum_{i<=y, j<=x} p[i][j],  row 0 and column 0 of S are zero.
//
// The image is cut into tiles of TH pixel rows x 256 S-columns; ONE WAVE owns a tile (a lane owns 4 consecutive S columns = one dword of
// pixels and one 16-byte store per row):
//   pass A  k_integral_tilesums  per tile: column sums of its pixels (one u32 per pixel column), row sums (one u32 per pixel row), total
//                                -> colsum[ty][x], rowsum[tx][y], tileTot[ty][tx]: (H/TH) x W + (W/256) x H words, ~2 MB for a 4K frame;
//   scan    k_integral_carries   exclusive scans of those along ty resp. tx, in place (W + H threads, <= H/TH resp. W/256 steps each), and
//                                the 2-D exclusive prefix of the tile totals (one workgroup, in LDS) = every tile's corner value;
//   pass B  k_integral_tiles     per tile: top edge = corner + prefix of its column carries, then row by row: lane-local prefix of 4 pixels
//                                + DPP wave scan + the row's carry, added into the running column accumulators, stored.
// Traffic: the pixels twice (the second time from L2 / Infinity Cache), the carries (~2 % of the output), the output once: within a few per
// cent of the compulsory 1 B read + 4 (8) B written per pixel, against ~4x that for the row pass + two column passes it replaces.
// All sums are exact integers (u32 partials, int32 / u64 accumulators); CV_64F outputs are those integers converted once, so they equal the
// reference's double sums bit for bit.
#include "rt.h"
#include "integral.h"

namespace mi355 {
namespace {

constexpr int ITH = 16;            // pixel rows per tile (a wave issues all of a tile's pixel loads before it consumes the first)
constexpr int ITW = 256;           // S columns per tile (64 lanes x 4)

typedef unsigned u32u __attribute__((aligned(1)));

// inclusive prefix sum over the 64 lanes of a wave (row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 / 31 across rows)
__device__ __forceinline__ unsigned waveScanIncl(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ unsigned long long waveScanIncl64(unsigned long long v)
{
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long a = __shfl_up(v, o);
        if ((int)(threadIdx.x & 63) >= o) v += a;
    }
    return v;
}

// the 4 pixels of a lane: S columns c..c+3 of the tile use pixel columns c-1..c+2; `x` = c - 1 may be -1 (S column 0) and the dword may run past
// the row's end -- those bytes read as 0.  Interior lanes take one (possibly unaligned) dword load.
__device__ __forceinline__ unsigned loadPix4(const uchar* __restrict__ row, int x, int W)
{
    if (x >= 0 && x + 4 <= W) return *reinterpret_cast<const u32u*>(row + x);
    unsigned w = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) if (x + b >= 0 && x + b < W) w |= (unsigned)row[x + b] << (8 * b);
    return w;
}

// ---- pass A: tile sums.  colsum / colsq indexed [ty][pixel x] (pitch Wp), rowsum / rowsq indexed [tx][pixel y] (pitch H)
template <bool SQ>
__global__ __launch_bounds__(256) void k_integral_tilesums(const uchar* __restrict__ src, size_t sstep, size_t sframe, int W, int H, int nTx, int nTy, int nframes,
                                                           unsigned* __restrict__ colsum, unsigned* __restrict__ colsq, unsigned* __restrict__ rowsum,
                                                           unsigned* __restrict__ rowsq, unsigned* __restrict__ tileTot, unsigned* __restrict__ tileTotQ,
                                                           int Wp, size_t auxFrame)
{
    const int lane = threadIdx.x & 63;
    const int wid = (int)xcdContiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);     // neighbouring tiles (shared output lines) behind the same L2
    const int tx = wid % nTx, ty = (wid / nTx) % nTy, f = wid / (nTx * nTy);
    if (f >= nframes) return;
    src += (size_t)f * sframe;
    colsum += (size_t)f * auxFrame; rowsum += (size_t)f * auxFrame; tileTot += (size_t)f * auxFrame;
    if (SQ) { colsq += (size_t)f * auxFrame; rowsq += (size_t)f * auxFrame; tileTotQ += (size_t)f * auxFrame; }
    // aligned pixel strips here: pixel columns [256 tx, 256 tx + 256); the shift by one S column is pass B's business
    const int x = tx * ITW + 4 * lane, y0 = ty * ITH, y1 = min(H, y0 + ITH);
    unsigned cs[4] = {0, 0, 0, 0}, cq[4] = {0, 0, 0, 0};
    unsigned myRow = 0, myRowQ = 0;
    unsigned wv[ITH];
#pragma unroll
    for (int r = 0; r < ITH; r++) wv[r] = (x < W && y0 + r < y1) ? loadPix4(src + (size_t)(y0 + r) * sstep, x, W) : 0u;
#pragma unroll
    for (int r = 0; r < ITH; r++) {
        const unsigned w = wv[r];
        unsigned s = 0, q = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned p = (w >> (8 * b)) & 255u;
            cs[b] += p; s += p;
            if (SQ) { cq[b] += p * p; q += p * p; }
        }
        const unsigned rs = (unsigned)__builtin_amdgcn_readlane((int)waveScanIncl(s), 63);      // the row's sum over the strip
        if (lane == r) myRow = rs;
        if (SQ) {
            const unsigned rq = (unsigned)__builtin_amdgcn_readlane((int)waveScanIncl(q), 63);
            if (lane == r) myRowQ = rq;
        }
    }
    if (x < W) {
#pragma unroll
        for (int b = 0; b < 4; b++) if (x + b < W) {
            colsum[(size_t)ty * Wp + x + b] = cs[b];
            if (SQ) colsq[(size_t)ty * Wp + x + b] = cq[b];
        }
    }
    if (lane < y1 - y0) {
        rowsum[(size_t)tx * H + y0 + lane] = myRow;
        if (SQ) rowsq[(size_t)tx * H + y0 + lane] = myRowQ;
    }
    // the tile's total (lanes >= rows hold 0), for the 2-D prefix over tiles that gives every tile its corner value
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)waveScanIncl(myRow), 63);
    if (lane == 0) tileTot[(size_t)ty * nTx + tx] = tot;
    if (SQ) {
        const unsigned totq = (unsigned)__builtin_amdgcn_readlane((int)waveScanIncl(myRowQ), 63);
        if (lane == 0) tileTotQ[(size_t)ty * nTx + tx] = totq;
    }
}

// ---- scan: colsum[ty][x] -> sum over tiles above (exclusive, along ty); rowsum[tx][y] -> sum over strips to the left (exclusive, along tx)
__global__ __launch_bounds__(256) void k_integral_carries(const unsigned* __restrict__ colsum, const unsigned* __restrict__ colsq, const unsigned* __restrict__ rowsum,
                                                          const unsigned* __restrict__ rowsq, unsigned* __restrict__ colcar, unsigned* __restrict__ colcarQ,
                                                          unsigned* __restrict__ rowcar, unsigned* __restrict__ rowcarQ,
                                                          const unsigned* __restrict__ tileTot, const unsigned* __restrict__ tileTotQ,
                                                          unsigned long long* __restrict__ corner, unsigned long long* __restrict__ cornerQ,
                                                          int W, int H, int Wp, int nTx, int nTy, size_t auxFrame, int sq)
{
    const size_t fo = (size_t)blockIdx.z * auxFrame;
    const int which = blockIdx.y;                                   // 0: sums, 1: squares
    if (which && !sq) return;
    if (blockIdx.x == gridDim.x - 1) {
        // last block: corner[ty][tx] = sum of the totals of the tiles above-left (t < ty, s < tx), in LDS.  Along tx first (one thread per tile
        // row, <= 64 steps), then along ty -- the long direction -- in three short phases: 16 chunks per strip column are summed in parallel,
        // the 16 chunk sums are scanned, and every chunk rewrites its entries as exclusive prefixes.
        extern __shared__ unsigned long long tt[];                  // nTy x nTx, then 16 x nTx chunk sums
        const unsigned* src = (which ? tileTotQ : tileTot) + fo;
        unsigned long long* dst = (unsigned long long*)((char*)(which ? cornerQ : corner) + fo * 4);      // the corner arrays are 2 words per entry
        const int n = nTx * nTy;
        unsigned long long* cs = tt + n;
        for (int i = threadIdx.x; i < n; i += 256) tt[i] = src[i];
        __syncthreads();
        for (int t = threadIdx.x; t < nTy; t += 256) {               // exclusive prefix along tx
            unsigned long long run = 0;
            for (int sx = 0; sx < nTx; sx++) { const unsigned long long v = tt[t * nTx + sx]; tt[t * nTx + sx] = run; run += v; }
        }
        __syncthreads();
        const int per = (nTy + 15) / 16;                             // tile rows per chunk
        for (int i = threadIdx.x; i < 16 * nTx; i += 256) {
            const int k = i / nTx, sx = i - k * nTx;
            unsigned long long sum = 0;
            for (int t = k * per; t < min(nTy, (k + 1) * per); t++) sum += tt[t * nTx + sx];
            cs[i] = sum;
        }
        __syncthreads();
        for (int sx = threadIdx.x; sx < nTx; sx += 256) {
            unsigned long long run = 0;
            for (int k = 0; k < 16; k++) { const unsigned long long v = cs[k * nTx + sx]; cs[k * nTx + sx] = run; run += v; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * nTx; i += 256) {
            const int k = i / nTx, sx = i - k * nTx;
            unsigned long long run = cs[i];
            for (int t = k * per; t < min(nTy, (k + 1) * per); t++) { const unsigned long long v = tt[t * nTx + sx]; dst[t * nTx + sx] = run; run += v; }
        }
        return;
    }
    const unsigned* cs = (which ? colsq : colsum) + fo;
    const unsigned* rs = (which ? rowsq : rowsum) + fo;
    unsigned* co = (which ? colcarQ : colcar) + fo;
    unsigned* ro = (which ? rowcarQ : rowcar) + fo;
    const int colBlocks = (W + 31) / 32;
    if ((int)blockIdx.x < colBlocks) {
        // column carries: a workgroup takes 32 columns x 8 chunks of tile rows.  Every thread loads its chunk at once (one memory round trip for
        // the whole scan instead of nTy dependent ones in a handful of waves), the chunk sums are exchanged through LDS, and the exclusive
        // prefixes are written from the registers.
        __shared__ unsigned part[8][33];
        const int c = threadIdx.x & 31, ck = threadIdx.x >> 5, col = blockIdx.x * 32 + c;
        const int per = (nTy + 7) / 8, t0 = ck * per, t1 = min(nTy, t0 + per);
        unsigned v[32];
        unsigned sum = 0;
        if (per <= 32) {
#pragma unroll
            for (int u = 0; u < 32; u++) { v[u] = (col < W && t0 + u < t1) ? cs[(size_t)(t0 + u) * Wp + col] : 0u; sum += v[u]; }
        } else {
            for (int t = t0; t < t1; t++) if (col < W) sum += cs[(size_t)t * Wp + col];
        }
        part[ck][c] = sum;
        __syncthreads();
        unsigned run = 0;
        for (int k = 0; k < ck; k++) run += part[k][c];
        if (col < W) {
            if (per <= 32) {
#pragma unroll
                for (int u = 0; u < 32; u++) if (t0 + u < t1) { co[(size_t)(t0 + u) * Wp + col] = run; run += v[u]; }
            } else {
                for (int t = t0; t < t1; t++) { const unsigned x = cs[(size_t)t * Wp + col]; co[(size_t)t * Wp + col] = run; run += x; }
            }
        }
        return;
    }
    const int y = ((int)blockIdx.x - colBlocks) * 256 + threadIdx.x;
    if (y < H) {
        unsigned run = 0;
#pragma unroll 8
        for (int tx = 0; tx < nTx; tx++) { const unsigned x = rs[(size_t)tx * H + y]; ro[(size_t)tx * H + y] = run; run += x; }
    }
}

// ---- pass B.  TS = int (CV_32S sums, wrapping like the reference's int arithmetic) or double (exact integers accumulated in u64).
template <typename TS> struct Acc;
template <> struct Acc<int>    { typedef unsigned T; static __device__ __forceinline__ int    out(unsigned v) { return (int)v; } };
template <> struct Acc<double> { typedef unsigned long long T; static __device__ __forceinline__ double out(unsigned long long v) { return (double)v; } };

template <typename TS>
__device__ __forceinline__ void storeRow4(TS* __restrict__ drow, int c, int Wc, const typename Acc<TS>::T (&a)[4], bool nt = false)
{
    if (c + 4 <= Wc) {
        if (sizeof(TS) == 4) {
            typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
            i4u v; v.x = (int)a[0]; v.y = (int)a[1]; v.z = (int)a[2]; v.w = (int)a[3];
            if (nt) __builtin_nontemporal_store(v, reinterpret_cast<i4u*>(drow + c)); else *reinterpret_cast<i4u*>(drow + c) = v;
        } else {
            typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
            d2u v0, v1; v0.x = (double)a[0]; v0.y = (double)a[1]; v1.x = (double)a[2]; v1.y = (double)a[3];
            *reinterpret_cast<d2u*>(drow + c) = v0; *reinterpret_cast<d2u*>(drow + c + 2) = v1;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (c + k < Wc) drow[c + k] = Acc<TS>::out(a[k]);
    }
}

template <typename TS, bool SQ>
__global__ __launch_bounds__(256) void k_integral_tiles(const uchar* __restrict__ src, size_t sstep, size_t sframe, int W, int H, int nTx, int nTy, int nframes,
                                                        TS* __restrict__ sum, size_t sumStep, size_t sumFrame, double* __restrict__ sq, size_t sqStep, size_t sqFrame,
                                                        const unsigned* __restrict__ colsum, const unsigned* __restrict__ colsq, const unsigned* __restrict__ rowsum,
                                                        const unsigned* __restrict__ rowsq, const unsigned long long* __restrict__ cornerArr,
                                                        const unsigned long long* __restrict__ cornerArrQ, int Wp, size_t auxFrame, int nt)
{
    typedef typename Acc<TS>::T A;
    const int lane = threadIdx.x & 63;
    const int wid = (int)xcdContiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);     // neighbouring tiles (shared output lines) behind the same L2
    const int tx = wid % nTx, ty = (wid / nTx) % nTy, f = wid / (nTx * nTy);
    if (f >= nframes) return;
    src += (size_t)f * sframe; sum += (size_t)f * sumFrame;
    colsum += (size_t)f * auxFrame; rowsum += (size_t)f * auxFrame;
    cornerArr = (const unsigned long long*)((const char*)cornerArr + (size_t)f * auxFrame * 4);
    if (SQ) { sq += (size_t)f * sqFrame; colsq += (size_t)f * auxFrame; rowsq += (size_t)f * auxFrame;
              cornerArrQ = (const unsigned long long*)((const char*)cornerArrQ + (size_t)f * auxFrame * 4); }
    const int Wc = W + 1;                                            // S columns
    const int X0 = tx * ITW, c = X0 + 4 * lane;                      // this lane's S columns c..c+3; its pixels are c-1..c+2
    const int y0 = ty * ITH, y1 = min(H, y0 + ITH);

    // ---- top edge of the tile: S[y0][cc] = sum of the column carries (sums of the tiles above) of the pixel columns j < cc, for cc = c..c+3
    A top[4]; unsigned long long topq[4] = {0, 0, 0, 0};
    {
        const unsigned* cc = colsum + (size_t)ty * Wp;
        const unsigned* cq = SQ ? colsq + (size_t)ty * Wp : nullptr;
        // corner = everything above-left of the tile: the 2-D prefix of the (aligned) tile totals covers the pixel columns < X0; the tile's own
        // first pixel column is X0 - 1, so its carry comes off again
        A corner = (A)cornerArr[(size_t)ty * nTx + tx];
        unsigned long long cornerq = SQ ? cornerArrQ[(size_t)ty * nTx + tx] : 0ull;
        if (X0 > 0) { corner -= (A)cc[X0 - 1]; if (SQ) cornerq -= cq[X0 - 1]; }
        // this lane's four carries: pixel columns c-1, c, c+1, c+2 (out of range -> 0); S column c+k takes k4[0..k]
        unsigned k4[4], q4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = c - 1 + k;
            const bool in = j >= 0 && j < W;
            k4[k] = in ? cc[j] : 0u;
            q4[k] = (SQ && in) ? cq[j] : 0u;
        }
        A loc[4]; unsigned long long locq[4];
        loc[0] = k4[0]; loc[1] = loc[0] + k4[1]; loc[2] = loc[1] + k4[2]; loc[3] = loc[2] + k4[3];
        locq[0] = q4[0]; locq[1] = locq[0] + q4[1]; locq[2] = locq[1] + q4[2]; locq[3] = locq[2] + q4[3];
        A excl; unsigned long long exclq = 0;
        if (sizeof(A) == 4) excl = (A)(waveScanIncl((unsigned)loc[3]) - (unsigned)loc[3]);
        else excl = (A)(waveScanIncl64((unsigned long long)loc[3]) - (unsigned long long)loc[3]);
        if (SQ) exclq = waveScanIncl64(locq[3]) - locq[3];
#pragma unroll
        for (int k = 0; k < 4; k++) { top[k] = corner + excl + loc[k]; if (SQ) topq[k] = cornerq + exclq + locq[k]; }
    }
    // row 0 of S is zero: written by the first row of tiles
    if (ty == 0 && c < Wc) {
        const A z[4] = {0, 0, 0, 0};
        const unsigned long long zq[4] = {0, 0, 0, 0};
        storeRow4<TS>(sum, c, Wc, z);
        if (SQ) storeRow4<double>(sq, c, Wc, zq);
    }
    // the row carries of this tile's rows (sum of the pixels left of pixel column X0 -- aligned strips, see pass A): lane r holds row y0 + r
    const unsigned rcAll = lane < y1 - y0 ? rowsum[(size_t)tx * H + y0 + lane] : 0u;
    const unsigned rqAll = (SQ && lane < y1 - y0) ? rowsq[(size_t)tx * H + y0 + lane] : 0u;
    // pixels X0 + 4 lane .. + 3 (aligned strip) of every row of the tile, all loads in flight before the first is consumed
    const int x = X0 + 4 * lane;
    unsigned wv[ITH];
#pragma unroll
    for (int r = 0; r < ITH; r++) wv[r] = (x < W && y0 + r < y1) ? loadPix4(src + (size_t)(y0 + r) * sstep, x, W) : 0u;
#pragma unroll
    for (int r = 0; r < ITH; r++) {
        if (y0 + r >= y1) break;
        const int y = y0 + r;
        // shifted by one pixel: q = {previous lane's p3, p0, p1, p2}; lane 0's first is the strip boundary pixel X0 - 1, which the row carry
        // already contains -> 0
        const unsigned w = wv[r];
        unsigned prev = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0 gets 0
        const unsigned p0 = prev >> 24, p1 = w & 255u, p2 = (w >> 8) & 255u, p3 = (w >> 16) & 255u;
        const unsigned a0 = p0, a1 = a0 + p1, a2 = a1 + p2, a3 = a2 + p3;
        const unsigned rc = (unsigned)__builtin_amdgcn_readlane((int)rcAll, r);
        const unsigned base = rc + (waveScanIncl(a3) - a3);
        top[0] += (A)(base + a0); top[1] += (A)(base + a1); top[2] += (A)(base + a2); top[3] += (A)(base + a3);
        if (c < Wc) storeRow4<TS>(sum + (size_t)(y + 1) * sumStep, c, Wc, top, nt != 0);
        if (SQ) {
            const unsigned b0 = p0 * p0, b1 = b0 + p1 * p1, b2 = b1 + p2 * p2, b3 = b2 + p3 * p3;
            const unsigned rq = (unsigned)__builtin_amdgcn_readlane((int)rqAll, r);
            const unsigned baseq = rq + (waveScanIncl(b3) - b3);
            topq[0] += baseq + b0; topq[1] += baseq + b1; topq[2] += baseq + b2; topq[3] += baseq + b3;
            if (c < Wc) storeRow4<double>(sq + (size_t)(y + 1) * sqStep, c, Wc, topq);
        }
    }
}

} // namespace

// aux layout per frame and per kind (sums / squares), in 4-byte words:
//   colsum nTy x W | rowsum nTx x H | colcarry nTy x W | rowcarry nTx x H | tileTot nTy x nTx | (pad to 8 bytes) | corner 2 x nTy x nTx
static size_t auxWordsPerFrame(int W, int H, int* nTxOut, int* nTyOut)
{
    const int nTx = divUp(W + 1, ITW), nTy = divUp(H, ITH);
    if (nTxOut) *nTxOut = nTx;
    if (nTyOut) *nTyOut = nTy;
    size_t w = 2 * ((size_t)nTy * W + (size_t)nTx * H) + (size_t)nTy * nTx;
    w = (w + 1) & ~(size_t)1;                                        // the u64 corner array starts 8-byte aligned
    return w + 2 * (size_t)nTy * nTx;
}

size_t integralTiledAuxBytes(int W, int H, int nframes, bool sq)
{
    return auxWordsPerFrame(W, H, nullptr, nullptr) * 4 * (sq ? 2 : 1) * (size_t)nframes + 16;
}

bool integralTiledU8(const uchar* src, size_t sstep, size_t sframe, int W, int H, int nframes, void* sum, size_t sumStepElems, size_t sumFrameElems, bool sumIsDouble,
                     double* sq, size_t sqStepElems, size_t sqFrameElems, void* aux, hipStream_t st)
{
    if (W < 1 || H < 1 || nframes < 1 || !aux || !sum) return false;
    // the squared partial sums (row / column carries of Q) are u32: 65025 * max(W, H) must stay below 2^32 (W, H <= 66051); beyond, the
    // general three-pass path with its 64-bit sums serves the call
    if (sq && (W > 66051 || H > 66051)) return false;
    int nTx, nTy;
    const size_t perFrame = auxWordsPerFrame(W, H, &nTx, &nTy);       // even: every array of every frame keeps its 8-byte alignment
    if (((size_t)nTx * nTy + 16 * (size_t)nTx) * 8 > 60 * 1024) return false;   // the 2-D prefix of the tile totals runs in one workgroup's LDS
    const int Wp = W;
    unsigned* base = (unsigned*)(((uintptr_t)aux + 7) & ~(uintptr_t)7);
    struct Arrays { unsigned *colsum, *rowsum, *colcar, *rowcar, *tileTot; unsigned long long* corner; };
    auto layout = [&](unsigned* b) {
        Arrays a;
        const size_t cw = (size_t)nTy * Wp, rw = (size_t)nTx * H;
        a.colsum = b; a.rowsum = a.colsum + cw; a.colcar = a.rowsum + rw; a.rowcar = a.colcar + cw; a.tileTot = a.rowcar + rw;
        size_t w = 2 * (cw + rw) + (size_t)nTy * nTx; w = (w + 1) & ~(size_t)1;
        a.corner = (unsigned long long*)(b + w);
        return a;
    };
    const Arrays S = layout(base);
    Arrays Q = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (sq) Q = layout(base + perFrame * nframes);
    const long long waves = (long long)nTx * nTy * nframes;
    const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    if (sq) hipLaunchKernelGGL((k_integral_tilesums<true>), grid, blk, 0, st, src, sstep, sframe, W, H, nTx, nTy, nframes, S.colsum, Q.colsum, S.rowsum, Q.rowsum, S.tileTot, Q.tileTot, Wp, perFrame);
    else    hipLaunchKernelGGL((k_integral_tilesums<false>), grid, blk, 0, st, src, sstep, sframe, W, H, nTx, nTy, nframes, S.colsum, Q.colsum, S.rowsum, Q.rowsum, S.tileTot, Q.tileTot, Wp, perFrame);
    hipLaunchKernelGGL(k_integral_carries, dim3(divUp(W, 32) + divUp(H, 256) + 1, sq ? 2 : 1, nframes), blk, ((size_t)nTx * nTy + 16 * (size_t)nTx) * 8, st,
                       S.colsum, Q.colsum, S.rowsum, Q.rowsum, S.colcar, Q.colcar, S.rowcar, Q.rowcar, S.tileTot, Q.tileTot, S.corner, Q.corner,
                       W, H, Wp, nTx, nTy, perFrame, sq ? 1 : 0);
    constexpr int nt = 0;                                          // (non-temporal sum stores measured 7-10 % slower, profiles/r04_integral_aligned_stores_ab.txt)
#define ITILES(TS_, SQ_) hipLaunchKernelGGL((k_integral_tiles<TS_, SQ_>), grid, blk, 0, st, src, sstep, sframe, W, H, nTx, nTy, nframes, (TS_*)sum, sumStepElems, sumFrameElems, \
                                            sq, sqStepElems, sqFrameElems, S.colcar, Q.colcar, S.rowcar, Q.rowcar, S.corner, Q.corner, Wp, perFrame, nt)
    if (sumIsDouble) { if (sq) ITILES(double, true); else ITILES(double, false); }
    else             { if (sq) ITILES(int, true); else ITILES(int, false); }
#undef ITILES
    return true;
}

} // namespace mi355
