// seplong.hip -- separable filtering with long kernels (10 .. lim::SEP_MAX_TAPS taps per axis), rows a1 / a4 of SURVEY.md section 8.
//
// What the reference does (filter.simd.hpp:2386 RowFilter, :2652 SymmColumnFilter, smooth.simd.hpp:954 hlineSmooth, :1629 vlineSmooth): every source row is
// filtered horizontally ONCE into a ring of ny intermediate rows, every output row is one vertical combination of that ring: nx + ny multiply-adds per element.
// The kernels this file replaces (k_sepfilter_generic<129>, k_sepfixed_generic beyond 9 taps) recomputed the ny row sums for every output: nx * ny gathers.
//
// k_seplong<MODE, CN>: a workgroup of 256 lanes owns a strip of TP pixels and walks DOWN a segment of rows, RB = 16 source rows per step:
//   1. stage   RB source rows of the strip (+ nx - 1 pixels of halo, borders resolved here, channels de-interleaved into planes) into LDS as 4-byte values;
//   2. row pass  each lane filters 4 consecutive pixels of one plane of one row: a sliding 8-register window refilled by one ds_read_b128 per 4 taps, taps from
//              scalar loads (uniform index); the sums go to an LDS ring of NR >= ny - 1 + RB rows;
//   3. column pass  each lane owns 2 neighbouring pixels x 4 consecutive output rows: one ds_read_b64 per ring row feeds 8 multiply-adds (sliding windows up and
//              down the ring for the symmetric / anti-symmetric pair forms); results leave in the destination depth.
// Two barriers per step; the ring is the only vertical state, so a segment costs ny - 1 extra row passes at its top.  The arithmetic (order of every chain, where
// the reference's vector body and scalar tail differ) is that of the kernels it replaces, which tests/test_filters_gpu.py pins to the reference bit for bit.
#include "seplong.h"
#include "seplong_body.h"
#include <climits>
#include <cstring>

using namespace mi355;

namespace {

typedef seplong::Geom LongGeom;
using seplong::RB;

template <int MODE, int CN, bool LONG>
__global__ __launch_bounds__(256) void k_seplong(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                 LongGeom g, const uint32_t* __restrict__ taps)
{
    constexpr int MMAX = seplong::StageOf<CN, LONG>::MMAX;
    extern __shared__ uint4 lds4[];
    uint32_t* S = reinterpret_cast<uint32_t*>(lds4);                       // [RB][CN][SP]
    uint32_t* ring = S + RB * CN * g.SP;                                   // [NR][CN][RP]
    const int tid = threadIdx.x;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    seplong::Seg<CN> sg;
    sg.init(g, blockIdx.x, blockIdx.y);
    const uint32_t* kx = taps;
    const uint32_t* ky = taps + g.nx;
    const uint32_t* kyS = taps + g.nx + g.ny;                              // mode 1: float(ky) * 2^-16
    uint32_t v[seplong::RPW * MMAX];                                       // the step's source elements on their way from HBM to LDS
    seplong::stageLoad<MODE, CN, MMAX>(g, sg, 0, src, sstep, tid, v);
    int done = 0;
    for (int j = 0; j < sg.nsteps; j++) {
        seplong::stageStore<CN, MMAX>(g, sg, j, S, tid, v);
        if (j + 1 < sg.nsteps) seplong::stageLoad<MODE, CN, MMAX>(g, sg, j + 1, src, sstep, tid, v);      // in flight while step j is filtered
        __syncthreads();
        seplong::rowPass<MODE, CN>(g, sg, j, S, ring, kx, tid);
        __syncthreads();                                                   // (also: every lane is done with S before the next stageStore)
        const int newDone = seplong::doneAfter<CN>(g, sg, j);
        seplong::colPass<MODE, CN>(g, sg, done, newDone, ring, ky, kyS, dst, dstep, tid);
        done = newDone;
    }
}

template <int MODE>
void launchLong(int cn, bool lng, dim3 grid, size_t lds, hipStream_t st, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, const LongGeom& g, const uint32_t* taps)
{
#define LAUNCH_(CN_, L_) do { \
        static bool attr[64] = {}; const int dv = activeDevice() & 63; \
        if (lds > 48 * 1024 && !attr[dv]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_seplong<MODE, CN_, L_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dv] = true; } \
        hipLaunchKernelGGL((k_seplong<MODE, CN_, L_>), grid, dim3(256), lds, st, src, sstep, sframe, dst, dstep, dframe, g, taps); } while (0)
#define LAUNCHC_(CN_) do { if (lng) LAUNCH_(CN_, true); else LAUNCH_(CN_, false); } while (0)
    switch (cn) { case 1: LAUNCHC_(1); break; case 2: LAUNCHC_(2); break; case 3: LAUNCHC_(3); break; default: LAUNCHC_(4); }
#undef LAUNCHC_
#undef LAUNCH_
}

} // namespace

namespace mi355 {

bool seplongRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
                int W, int H, int cn, int sdepth, int ddepth, int fullW, int fullH, int offX, int offY, int border, const SepLongTaps& t, hipStream_t st)
{
    if (cn < 1 || cn > 4 || t.nx < 1 || t.ny < 1 || t.nx > lim::SEP_MAX_TAPS || t.ny > lim::SEP_MAX_TAPS || W < 1 || H < 1 || nframes < 1) return false;
    if (t.ax < 0 || t.ax >= t.nx || t.ay < 0 || t.ay >= t.ny || border < 0 || border > B_REFLECT_101) return false;
    enum { D8U = MI355CV_8U, D16U = MI355CV_16U, D16S = MI355CV_16S, D32F = MI355CV_32F };
    if (t.mode == 0) {
        if (sdepth != D8U && sdepth != D16U && sdepth != D16S && sdepth != D32F) return false;
        if (ddepth != D8U && ddepth != D16U && ddepth != D16S && ddepth != D32F) return false;
        if (t.symY && (!(t.ny & 1) || t.ay != t.ny / 2)) return false;
    } else {
        if (sdepth != D8U || ddepth != (t.mode == 2 ? D16S : D8U)) return false;
        if (t.mode == 1 && t.ny > 1 && (!(t.ny & 1) || t.ay != t.ny / 2)) return false;       // the float column form is the symmetric pair form
    }
    LongGeom g;
    memset(&g, 0, sizeof g);
    g.W = W; g.H = H; g.sdepth = sdepth; g.ddepth = ddepth; g.fullW = fullW; g.fullH = fullH; g.offX = offX; g.offY = offY; g.border = border;
    g.nx = t.nx; g.ny = t.ny; g.ax = t.ax; g.ay = t.ay; g.symY = t.symY; g.deltaF = t.deltaF; g.deltaI = t.deltaI;
    for (int k = 0; k < 4; k++) g.bval[k] = t.bval[k];
    size_t lds = 0; int nstrips = 0, nseg = 0;
    if (!seplong::plan(g, cn, nframes, &lds, &nstrips, &nseg)) return false;
    const int seg = g.seg;
    if (nseg > 65535 || nframes > 65535) return false;
    // the taps: kx, ky (float bits or ints), then mode 1's float(ky) * 2^-16
    std::vector<uint32_t> tb((size_t)t.nx + 2 * (size_t)t.ny);
    for (int i = 0; i < t.nx; i++) { if (t.mode == 0) memcpy(&tb[i], &t.kxf[i], 4); else tb[i] = t.mode >= 4 ? 0u : (uint32_t)t.kxi[i]; }
    for (int i = 0; i < t.ny; i++) {
        if (t.mode == 0) memcpy(&tb[t.nx + i], &t.kyf[i], 4); else tb[t.nx + i] = t.mode >= 4 ? 0u : (uint32_t)t.kyi[i];
        const float s = t.mode == 0 || t.mode >= 4 ? 0.f : (float)t.kyi[i] * (1.0f / 65536.0f);
        memcpy(&tb[t.nx + t.ny + i], &s, 4);
    }
    const uint32_t* dt = static_cast<const uint32_t*>(stg.param(tb.data(), tb.size() * 4));
    if (!dt) return false;
    const dim3 grid(nstrips, nseg, nframes);
    const bool lng = t.nx > 33;                                            // the staged row is (TP + nx - 1) * cn elements wide: registers per lane follow the class
    switch (t.mode) {
    case 0:  launchLong<0>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 1:  launchLong<1>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 2:  launchLong<2>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 4:  launchLong<4>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 5:  launchLong<5>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    default: launchLong<3>(cn, lng, grid, lds, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    }
    noteKernel("k_seplong<%d,%d,%d> grid=%ux%ux%u x256 lds=%zu taps=%dx%d seg=%d", t.mode, cn, (int)lng, grid.x, grid.y, grid.z, lds, t.nx, t.ny, seg);
    return true;
}

} // namespace mi355
