// tests/hostemu -- TEST INFRASTRUCTURE ONLY.  The LDS-ring separable kernel of opencv_amd/csrc/seplong_body.h run on the CPU: the same host plan, the same
// three phases per step, executed thread by thread with the two barriers replaced by loop boundaries, one "workgroup" (strip x segment) at a time.  The LDS image is
// poisoned before every workgroup so that a stale or never-written ring slot reaching an output shows up as a mismatch against the pinned restatement.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include "seplong_body.h"

template <int MODE, int CN, bool LONG>
static void runT(const seplong::Geom& g, int nstrips, int nseg, size_t ldsBytes, const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep,
                 const uint32_t* kx, const uint32_t* ky, const uint32_t* kyS)
{
    std::vector<uint32_t> lds(ldsBytes / 4 + 16);
    uint32_t* S = lds.data();
    uint32_t* ring = S + seplong::RB * CN * g.SP;
    for (int sy = 0; sy < nseg; sy++)
        for (int sx = 0; sx < nstrips; sx++) {
            std::memset(lds.data(), 0xA5, lds.size() * 4);
            seplong::Seg<CN> sg;
            sg.init(g, sx, sy);
            constexpr int MMAX = seplong::StageOf<CN, LONG>::MMAX;
            static uint32_t v[256][seplong::RPW * MMAX];                     // every thread's staging registers between the two halves of the staging
            for (int tid = 0; tid < 256; tid++) seplong::stageLoad<MODE, CN, MMAX>(g, sg, 0, src, sstep, tid, v[tid]);
            int done = 0;
            for (int j = 0; j < sg.nsteps; j++) {
                for (int tid = 0; tid < 256; tid++) seplong::stageStore<CN, MMAX>(g, sg, j, S, tid, v[tid]);
                if (j + 1 < sg.nsteps) for (int tid = 0; tid < 256; tid++) seplong::stageLoad<MODE, CN, MMAX>(g, sg, j + 1, src, sstep, tid, v[tid]);
                for (int tid = 0; tid < 256; tid++) seplong::rowPass<MODE, CN>(g, sg, j, S, ring, kx, tid);
                const int newDone = seplong::doneAfter<CN>(g, sg, j);
                for (int tid = 0; tid < 256; tid++) seplong::colPass<MODE, CN>(g, sg, done, newDone, ring, ky, kyS, dst, dstep, tid);
                done = newDone;
            }
        }
}

// kx / ky: the taps as mi355cv_sepFilterDescribe reports them (float bits in mode 0, int32 otherwise); segRows > 0 overrides the planned rows per segment (a
// multiple of 16 or >= H) so that small images exercise several segments.  Returns 0, or 1 where seplongRun would refuse.
extern "C" int emu_seplong(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int W, int H, int cn, int sdepth, int ddepth,
                           int fullW, int fullH, int offX, int offY, int border, int mode, const uint32_t* kx, int nx, const uint32_t* ky, int ny,
                           int ax, int ay, int symY, float deltaF, int deltaI, int nframes, int segRows, int* planOut)
{
    if (cn < 1 || cn > 4 || nx < 1 || ny < 1 || nx > 129 || ny > 129) return 1;
    seplong::Geom g;
    std::memset(&g, 0, sizeof g);
    g.W = W; g.H = H; g.sdepth = sdepth; g.ddepth = ddepth; g.fullW = fullW; g.fullH = fullH; g.offX = offX; g.offY = offY; g.border = border;
    g.nx = nx; g.ny = ny; g.ax = ax; g.ay = ay; g.symY = symY; g.deltaF = deltaF; g.deltaI = deltaI;
    size_t lds = 0; int nstrips = 0, nseg = 0;
    if (!seplong::plan(g, cn, nframes, &lds, &nstrips, &nseg)) return 1;
    if (segRows > 0) { g.seg = segRows > H ? H : segRows; nseg = (H + g.seg - 1) / g.seg; }
    if (planOut) { planOut[0] = nstrips; planOut[1] = nseg; planOut[2] = g.seg; planOut[3] = (int)lds; planOut[4] = g.NR; }
    std::vector<uint32_t> kyS(ny);
    for (int i = 0; i < ny; i++) { const float s = mode == 0 ? 0.f : (float)(int)ky[i] * (1.0f / 65536.0f); std::memcpy(&kyS[i], &s, 4); }
#define RUN(M_, C_) if (nx > 33) runT<M_, C_, true>(g, nstrips, nseg, lds, src, sstep, dst, dstep, kx, ky, kyS.data()); else runT<M_, C_, false>(g, nstrips, nseg, lds, src, sstep, dst, dstep, kx, ky, kyS.data())
#define RUNC(M_) do { switch (cn) { case 1: RUN(M_, 1); break; case 2: RUN(M_, 2); break; case 3: RUN(M_, 3); break; default: RUN(M_, 4); } } while (0)
    switch (mode) { case 0: RUNC(0); break; case 1: RUNC(1); break; case 2: RUNC(2); break; default: RUNC(3); }
#undef RUNC
#undef RUN
    return 0;
}
