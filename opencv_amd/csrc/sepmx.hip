// sepmx.hip -- cv::GaussianBlur on CV_8U with 10 .. 129 taps per axis (row a1 of SURVEY.md section 8; any sigma beyond the 9 taps of the register-rolling kernels) with BOTH
// passes on the matrix cores.
//
// The reference (fixedSmoothInvoker, smooth.simd.hpp:1926; hlineSmooth :954, vlineSmooth :1629) filters every row once with Q8.8 taps into 16-bit row sums and combines ny of
// them per output row: nx + ny multiply-adds per byte -- 38 for sigma = 3, 258 for 129 taps.  On vector lanes that is what bounds the kernel (k_seplong: 32 us per 4K frame for
// 19 taps, 0.065 of the HBM roofline).  Both passes are products with a banded Toeplitz matrix, and 8-bit pixels x 7-bit taps are what v_mfma_i32_32x32x32_i8 multiplies:
//   row pass     R[r][x]  = sum_k S[r][k] * Bx[k][x],   Bx[k][x] = kx[(k - x - delta) / cn]           (A = 32 source rows x 32 KSX bytes from LDS, B = constants in registers)
//   column pass  D[y][x]  = sum_k Ay[y][k] * R[k][x],   Ay[y][k] = ky[k - y]                          (A = constants in registers, B = the row sums)
// The row sums are 16-bit: they enter the column pass as two int8 planes (high and low byte, each with its own accumulator); sepmx_body.h has the bias algebra that makes
// every operand signed and the result exact.  The result registers of the row pass (lane = column, 16 rows per lane) ARE the B operand layout of the column pass once the K
// index is permuted the same way in Ay -- the row sums never leave the lane that computed them: no ring in LDS, no transposition between the passes.
//
// A workgroup of 8 waves owns a strip of 256 bytes of the row (wave w: 32 of them) and walks DOWN a segment, 32 rows per step:
//   wait for this wave's share of the step's source rows | barrier | the PREVIOUS step's output tile leaves as 1 KiB per wave-instruction (16 bytes per lane, whole 256-byte
//   row pieces) | request the rows TWO steps ahead (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into a ring of three staged blocks, no registers) | row pass
//   of this step's rows (KSX MFMAs; the pixels become signed on their way from LDS: one v_xor per dword) | split into byte planes (20 VALU per 16 sums) | column pass over
//   the last KSY tiles of row sums (2 KSY MFMAs) | (accH << 8) + accL, byte 2 into the transposition block.
// One barrier per step, ~19 KB of source rows in flight per workgroup, two workgroups per CU.  The left / right border lives in the row pass' matrix (sepmx_body.h:
// buildRowB), the top / bottom border in the address of the staged row, so staging is a plain copy of aligned 16-byte chunks.  Rows whose pitch or address rule out aligned
// chunks go through registers instead (one step ahead; DMA = false).  HBM traffic: the source once (+ nx - 1 columns per strip, + 32 (KSY - 1) rows per segment), the
// destination once.
#include "sepmx.h"
#include "sepmx_body.h"
#include <vector>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>

using namespace mi355;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
using sepmx::Geom;
using sepmx::TR;

__device__ __forceinline__ void waitVm(int n)       // s_waitcnt vmcnt(n) alone (vector-memory operations return in order: at most n of the newest are still out)
{
    if (n >= 3) __builtin_amdgcn_s_waitcnt(0x0F73);          // (a wave issues at most three per step)
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0F71);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

template <int KSX, int KSY, bool DMA, bool BOX>
__global__ __launch_bounds__(512, ((DMA ? (KSX + KSY <= 6 || (KSX + KSY == 7 && KSY <= 3)) : KSX + KSY <= 5) ? 4 : 2)) void k_sepmx(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                                  Geom g, const int* __restrict__ bsel /* [strips][8] */, const int* __restrict__ seeds /* [classes][32] */,
                                                  const v4i* __restrict__ rowB /* [classes][2][KSX][64] */, const v4i* __restrict__ colA /* [KSY][64] */)
{
    constexpr int NW = sepmx::NWAVE, DEPTH = 2, TW = sepmx::TW, NT = 64 * NW, NCHUNK = (TW - 32 + 32 * KSX) / 16, PC = NCHUNK | 1, P = 16 * PC, NSLOT = DMA ? DEPTH + 1 : 2, NI = (TR * PC + NT - 1) / NT;
    static_assert(NI <= 3, "at most three chunks per lane and step");
    extern __shared__ uint4 lds16[];                     // NSLOT staged blocks of TR x P bytes, two transposition blocks of TR x TW, the column pass' A operand (KSY KB)
    uchar (*stage)[TR * P] = reinterpret_cast<uchar (*)[TR * P]>(lds16);
    uchar (*tr)[TR * TW] = reinterpret_cast<uchar (*)[TR * TW]>(reinterpret_cast<uchar*>(lds16) + NSLOT * TR * P);
    v4i* AyL = reinterpret_cast<v4i*>(reinterpret_cast<uchar*>(lds16) + NSLOT * TR * P + 2 * TR * TW);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, h = lane >> 5;
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd) {                                         // workgroups go to the XCDs round-robin by linear id: give each XCD a contiguous run of (frame, segment, strip)
        const unsigned N = gridDim.x * gridDim.y * gridDim.z, L = bx + gridDim.x * (by + gridDim.y * bz), j = (L & 7) * (N >> 3) + (L >> 3);
        bx = j % gridDim.x; by = (j / gridDim.x) % gridDim.y; bz = j / (gridDim.x * gridDim.y);
        // (the divisions run on the vector unit: without these the strip / segment / frame and every pointer derived from them would live in vector registers)
        bx = __builtin_amdgcn_readfirstlane(bx); by = __builtin_amdgcn_readfirstlane(by); bz = __builtin_amdgcn_readfirstlane(bz);
    }
    src += (size_t)bz * sframe;
    dst += (size_t)bz * dframe;
    const int X0 = (int)bx * TW - g.shift, y0 = by * g.seg;
    const int rows = min(g.seg, g.H - y0);
    const int nU = (rows + TR - 1) / TR, nT = nU + KSY - 1;

    const int sel = __builtin_amdgcn_readfirstlane(bsel[bx * NW + wave]), cls = sel & 0xffff;
    const bool twice = (sel >> 16) != 0;                 // some weight of this wave's matrix is beyond int8: a second product with the rest (rim waves under BORDER_REPLICATE)
    v4i Bx[KSX];
#pragma unroll
    for (int k = 0; k < KSX; k++) Bx[k] = rowB[(cls * 2 * KSX + k) * 64 + lane];
    // the column pass' A operand is the same for every wave: it waits in LDS and is read where it is used (KSY x 4 registers less per lane: what keeps two workgroups on a CU)
    for (int i = tid; i < KSY * 64; i += NT) AyL[i] = colA[i];
    const int seedC = seeds[cls * 32 + n];            // the column's constant of the column pass (sepmx_body.h: the arithmetic)

    // staging: a block is TR rows of PC 16-byte chunks, chunk q at byte 16 q (the last chunk of a row is padding: the pitch is 16 * odd); wave-instruction i of wave w covers
    // chunks 64 (w + 8 i) .. + 63
    auto bytesOf = [&](const uchar* p, long long rel) -> uint4 {                     // a chunk that reaches outside the parent's memory: the bytes inside, zeros for the rest
        unsigned wd[4] = {0, 0, 0, 0};
#pragma nounroll
        for (int b = 0; b < 16; b++) if (rel + b >= 0 && rel + b < g.span) wd[b >> 2] |= (unsigned)p[b] << (8 * (b & 3));
        return make_uint4(wd[0], wd[1], wd[2], wd[3]);
    };
    uint4 park[DMA ? 1 : NI];
    // request the rows of step t; returns the number of asynchronous wave-instructions issued (DMA)
    auto glds16 = [&](const uchar* p, uchar* slot) {                                // 16 bytes per lane from p into slot + 16 * lane, asynchronously (counts on vmcnt)
        unsigned keep;
        const unsigned ldsAddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)slot;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(ldsAddr) : "memory");
    };
    auto request = [&](int t, int sl) -> int {                                      // (sl = t mod NSLOT, carried by the caller: no division in the walk)
        int issued = 0;
        const int syA = y0 - g.ay + TR * t + g.offY;                                 // the step's first source row in the parent
        const bool inner = g.fast && syA >= 1 && syA + TR - 1 <= g.fullH - 2;        // (uniform)
#pragma unroll
        for (int i = 0; i < NI; i++) {
            if (64 * (wave + NW * i) >= TR * PC) continue;                            // (wave-uniform)
            const uchar* p = src; long long rel = 0;
            int ln = lane;
            asm volatile("" : "+v"(ln));                                              // (opaque: the chunk's row / column / base pointer are recomputed per step -- hoisted out
            const int q = 64 * (wave + NW * i) + ln, cr = q / PC, cc = q - cr * PC;    //  of the walk they are six more registers per lane, and a spilled register comes back
            const int e0 = X0 - g.ax * g.cn - g.delta + 16 * cc;                       //  through vector memory, behind every row piece in flight)
            const bool valid = q < TR * PC && cc < NCHUNK;
            const int sy = y0 - g.ay + TR * t + cr;
            uchar* slot = &stage[sl][16 * 64 * (wave + NW * i)];
            if (DMA && inner) {
                // every row of the step is a real row away from the parent's rim: no border, no rim test; a wave-instruction's 64 chunks span three rows, so some lane always
                // issues -- the count needs no ballot
                if (valid) glds16(src + (ptrdiff_t)sy * (ptrdiff_t)sstep + e0, slot);
                issued++;
                continue;
            }
            int kind = -1;
            if (valid) {
                if (inner) { kind = sepmx::CH_LOAD; p = src + (ptrdiff_t)sy * (ptrdiff_t)sstep + e0; }
                else kind = sepmx::chunkKind(g, src, sstep, sy, e0, &p, &rel);
            }
            if (DMA) {
                if (__ballot(kind == sepmx::CH_LOAD)) {
                    if (kind == sepmx::CH_LOAD) glds16(p, slot);
                    issued++;
                }
                if (kind == sepmx::CH_ZERO) *reinterpret_cast<uint4*>(slot + 16 * lane) = make_uint4(0, 0, 0, 0);
                if (kind == sepmx::CH_BYTES) *reinterpret_cast<uint4*>(slot + 16 * lane) = bytesOf(p, rel);
            } else {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kind == sepmx::CH_LOAD) __builtin_memcpy(&v, p, 16);
                if (kind == sepmx::CH_BYTES) v = bytesOf(p, rel);
                park[i] = v;
            }
        }
        return __builtin_amdgcn_readfirstlane(issued);
    };
    auto deposit = [&](int sl) {                                                      // (registers -> block; DMA = false)
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (64 * (wave + NW * i) + lane < TR * PC && (64 * (wave + NW * i) + lane) % PC < NCHUNK) *reinterpret_cast<uint4*>(&stage[sl][16 * (64 * (wave + NW * i) + lane)]) = park[DMA ? 0 : i];
    };
    // an output tile (rows y0 + 32 u ..) from its transposition block, 1 KiB (4 / 2 whole row pieces) per wave
    auto emit = [&](int u, int buf) {
        const int rr = (TR / NW) * wave + lane / (TW / 16), cc = 16 * (lane % (TW / 16));
        const int y = TR * u + rr, x = X0 + cc;
        if (y >= rows || x < 0 || x >= g.WE) return;
        const uint4 v = *reinterpret_cast<const uint4*>(&tr[buf][rr * TW + cc]);
        uchar* d = dst + (size_t)(y0 + y) * dstep + x;
        if (x + 16 <= g.WE) __builtin_memcpy(d, &v, 16);
        else {
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma nounroll
            for (int i = 0; i < g.WE - x; i++) d[i] = (uchar)(wd[i >> 2] >> (8 * (i & 3)));
        }
    };

    v4i ringH[KSY], ringL[KSY];
#pragma unroll
    for (int k = 0; k < KSY; k++) { ringH[k] = v4i{0, 0, 0, 0}; ringL[k] = v4i{0, 0, 0, 0}; }

    int newer = 0;                                       // asynchronous instructions this wave has issued after those of the step it is about to read
    int inflight[DEPTH];                                 // ... per step still ahead (two steps: three gained nothing, profiles/r06_sepmx.txt)
    if (DMA) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) inflight[d] = d < nT ? request(d, d) : 0;
    } else { (void)request(0, 0); deposit(0); }
    int cur = 0;                                         // t mod NSLOT
    for (int t = 0; t < nT; t++, cur = cur + 1 == NSLOT ? 0 : cur + 1) {
        const int ahead = DMA ? (cur == 0 ? NSLOT - 1 : cur - 1) : (cur ^ 1);                  // (t + DEPTH) mod NSLOT resp. (t + 1) mod 2
        if (DMA) {
            newer = 0;
#pragma unroll
            for (int d = 1; d < DEPTH; d++) newer += inflight[d];
            waitVm(newer);
        }
        __syncthreads();
        const int u = t - (KSY - 1);
        if (u >= 1) emit(u - 1, (t - 1) & 1);
        if (DMA) {
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; d++) inflight[d] = inflight[d + 1];
            inflight[DEPTH - 1] = t + DEPTH < nT ? request(t + DEPTH, ahead) : 0;
        }
        else if (t + 1 < nT) (void)request(t + 1, ahead);
        __builtin_amdgcn_sched_barrier(0);               // (phase fences: the scheduler otherwise hoists the next phase's LDS reads over this one and runs out of registers;
        // ---- row pass                                  //  a spilled register is reloaded through vector memory, behind every row piece in flight)
        v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const uchar* A = &stage[cur][n * P + 32 * wave + 16 * h];
#pragma unroll
        for (int k = 0; k < KSX; k++) {
            v4i a = *reinterpret_cast<const v4i*>(A + 32 * k);
            a ^= v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};                 // pixels - 128
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bx[k], acc, 0, 0, 0);
        }
        if (twice) {
#pragma unroll
            for (int k = 0; k < KSX; k++) {
                v4i a = *reinterpret_cast<const v4i*>(A + 32 * k);
                a ^= v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, rowB[((cls * 2 + 1) * KSX + k) * 64 + lane], acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 16 row sums of column n -> the two int8 planes in the column pass' B layout (byte i <-> regRow(h, i))
#pragma unroll
        for (int k = 0; k + 1 < KSY; k++) { ringH[k] = ringH[k + 1]; ringL[k] = ringL[k + 1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned t01 = __builtin_amdgcn_perm((unsigned)acc[4 * q + 1], (unsigned)acc[4 * q], 0x05040100u);
            const unsigned t23 = __builtin_amdgcn_perm((unsigned)acc[4 * q + 3], (unsigned)acc[4 * q + 2], 0x05040100u);
            ringH[KSY - 1][q] = (int)__builtin_amdgcn_perm(t23, t01, 0x07050301u);
            ringL[KSY - 1][q] = (int)(__builtin_amdgcn_perm(t23, t01, 0x06040200u) ^ 0x80808080u);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u >= 0) {
            // ---- column pass
            v16i aH, aL;
#pragma unroll
            for (int i = 0; i < 16; i++) { aH[i] = 0; aL[i] = seedC; }
#pragma unroll
            for (int k = 0; k < KSY; k++) {
                const v4i ay = AyL[k * 64 + lane];
                aH = __builtin_amdgcn_mfma_i32_32x32x32_i8(ay, ringH[k], aH, 0, 0, 0);
                aL = __builtin_amdgcn_mfma_i32_32x32x32_i8(ay, ringL[k], aL, 0, 0, 0);
            }
            uchar* T = &tr[t & 1][32 * wave + n];
            if (BOX) {
                // cv::boxFilter's normalisations on the exact window sum (box_filter.simd.hpp: ColumnSum<ushort, uchar> :429-455, ColumnSum<int, uchar> :340-385)
                const bool tail = g.box == 2 && X0 + 32 * wave + n >= g.tailStart;
                unsigned sum[16];
#pragma unroll
                for (int i = 0; i < 16; i++) sum[i] = ((unsigned)aH[i] << 8) + (unsigned)aL[i];
                if (g.box == 1) {
#pragma unroll
                    for (int i = 0; i < 16; i++) T[sepmx::regRow(h, i) * TW] = (uchar)(__umul24(sum[i] + (unsigned)g.divDelta, (unsigned)g.divScale) >> 23);     // both factors < 2^24, the product < 2^32
                } else if (g.box == 3) {
#pragma unroll
                    for (int i = 0; i < 16; i++) T[sepmx::regRow(h, i) * TW] = (uchar)(sum[i] > 255u ? 255u : sum[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) T[sepmx::regRow(h, i) * TW] = (uchar)(int)fminf(rintf((float)sum[i] * g.scaleF), 255.f);
                    if (tail) {                                                          // (the last (W * cn) % 8 elements of a row: at most one wave of the last strip comes here)
#pragma unroll
                        for (int i = 0; i < 16; i++) T[sepmx::regRow(h, i) * TW] = (uchar)(int)fmin(rint((double)sum[i] * g.scaleD), 255.0);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const unsigned v = ((unsigned)aH[i] << 8) + (unsigned)aL[i];
                    T[sepmx::regRow(h, i) * TW] = (uchar)(v >> 16);
                }
            }
        }
        if (!DMA && t + 1 < nT) deposit(ahead);
    }
    __syncthreads();
    emit(nU - 1, (nT - 1) & 1);
}

template <int KSX, bool DMA, bool BOX>
void launchY(int ksy, dim3 grid, hipStream_t st, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, const Geom& g,
             const int* bsel, const int* seeds, const v4i* rowB, const v4i* colA)
{
    constexpr int TW = sepmx::TW, PC = ((TW - 32 + 32 * KSX) / 16) | 1;
    constexpr size_t lds = (size_t)(DMA ? 3 : 2) * TR * 16 * PC + 2 * (size_t)TR * TW + 1024 * sepmx::MAXKS;
#define SEPMX_LAUNCH_(KSY_) do { \
        static bool attr[64] = {}; const int dv = activeDevice() & 63; \
        if (lds > 48 * 1024 && !attr[dv]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sepmx<KSX, KSY_, DMA, BOX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[dv] = true; } \
        hipLaunchKernelGGL((k_sepmx<KSX, KSY_, DMA, BOX>), grid, dim3(512), lds, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); } while (0)
    switch (ksy) {
    case 2:  SEPMX_LAUNCH_(2);  break;
    case 3:  SEPMX_LAUNCH_(3);  break;
    case 4:  SEPMX_LAUNCH_(4);  break;
    case 5:  SEPMX_LAUNCH_(5);  break;
    case 7:  SEPMX_LAUNCH_(7);  break;
    default: if constexpr (KSX < 13) SEPMX_LAUNCH_(9);  break;        // (13 x 9 does not fit 256 registers without spilling: plan() declines it)
    }
#undef SEPMX_LAUNCH_
}
template <bool DMA, bool BOX>
void launchX(int ksx, int ksy, dim3 grid, hipStream_t st, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, const Geom& g,
             const int* bsel, const int* seeds, const v4i* rowB, const v4i* colA)
{
    switch (ksx) {
    case 2:  launchY<2, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    case 3:  launchY<3, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    case 4:  launchY<4, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    case 5:  launchY<5, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    case 7:  launchY<7, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    case 9:  launchY<9, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    default: launchY<13, DMA, BOX>(ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, bsel, seeds, rowB, colA); break;
    }
}

} // namespace

namespace mi355 {

bool sepmxRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
              int W, int H, int cn, int fullW, int fullH, int offX, int offY, int border, const uint16_t* kx, int nx, int ax, const uint16_t* ky, int ny, int ay, hipStream_t st,
              const SepmxBox* box)
{
    if (nframes < 1 || nframes > 65535 || border < 0 || border > B_REFLECT_101 || ax < 0 || ax >= nx || ay < 0 || ay >= ny) return false;
    Geom g;
    memset(&g, 0, sizeof g);
    g.W = W; g.H = H; g.cn = cn; g.fullW = fullW; g.fullH = fullH; g.offX = offX; g.offY = offY; g.border = border; g.nx = nx; g.ny = ny; g.ax = ax; g.ay = ay;
    if (box) { g.box = box->mode; g.divScale = box->divScale; g.divDelta = box->divDelta; g.scaleF = box->scaleF; g.scaleD = box->scaleD; g.tailStart = (W * cn) & ~7; }
    static const int dmaEnv = std::getenv("MI355CV_SEPMX_DMA") ? atoi(std::getenv("MI355CV_SEPMX_DMA")) : -1;
    static const int segEnv = std::getenv("MI355CV_SEPMX_SEG") ? atoi(std::getenv("MI355CV_SEPMX_SEG")) : 0;
    static const int xcdEnv = std::getenv("MI355CV_SEPMX_XCD") ? atoi(std::getenv("MI355CV_SEPMX_XCD")) : 1;
    if (!sepmx::plan(g, kx, ky, (uintptr_t)src, sstep, nframes > 1 ? sframe : 0, nframes, segEnv, dmaEnv)) return false;
    constexpr int TW = sepmx::TW;
    const int nstrips = (g.WE + g.shift + TW - 1) / TW, nseg = (H + g.seg - 1) / g.seg;
    if (nseg > 65535) return false;
    // The operand block (bsel | seeds | rowB | colA, each part 16-byte aligned) depends on the taps and the row geometry only: callers filter frame after frame with the same
    // parameters, so the last few blocks stay resident on the device (building one costs ~0.1 ms of host time, a per-frame call is ~20 us)
    struct Block { std::vector<int> key; int dev; uchar* d; size_t o1, o2, o3; int ncls; unsigned long long stamp; };
    static std::mutex mu;
    static std::vector<Block> cache;
    static unsigned long long clock = 0;
    std::vector<int> key = {nx, ny, ax, ay, cn, W, fullW, offX, border, g.delta, g.shift, g.ksx, g.ksy, g.box != 0};
    key.insert(key.end(), kx, kx + nx); key.insert(key.end(), ky, ky + ny);
    const int dev = activeDevice();
    const uchar* d = nullptr; size_t o1 = 0, o2 = 0, o3 = 0;
    // one lock from the lookup to the launch: a block found here cannot be evicted (and freed) by another host thread before the kernel that reads it is in the stream, and
    // the eviction's hipFree waits for the device, i.e. for every kernel launched under this lock before it
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : cache) if (e.dev == dev && e.key == key) { e.stamp = ++clock; d = e.d; o1 = e.o1; o2 = e.o2; o3 = e.o3; g.ncls = e.ncls; break; }
    if (!d) {
        // the row pass' operand classes: 0 = the plain Toeplitz matrix, one more per wave whose columns reach a left / right border (or the ragged end of the row)
        const size_t tabB = (size_t)g.ksx * 64 * 16;                              // per class: the matrix, then the part of its weights beyond int8
        std::vector<int> bsel((size_t)nstrips * 8, 0), seeds(32, 0);
        std::vector<int8_t> rowB(2 * tabB, 0);
        bool haveInterior = false;
        int ncls = 1;
        for (int s = 0; s < nstrips; s++)
            for (int w = 0; w < 8; w++) {
                const int e0 = s * TW - g.shift + 32 * w, e1 = e0 + 31;
                if (e1 < 0 || e0 >= g.WE) continue;                                   // a wave without outputs: class 0, never stored
                const bool inside = e0 >= 0 && e1 < g.WE && e0 / cn + offX - ax >= 0 && e1 / cn + offX + (nx - 1 - ax) < fullW;
                if (inside && haveInterior) continue;
                static thread_local int8_t tab[sepmx::MAXKSX * 64 * 16], tab2[sepmx::MAXKSX * 64 * 16]; int sd[32]; bool interior = false, twice = false;
                if (!sepmx::buildRowB(g, kx, g.sumKy, s * TW - g.shift, w, tab, tab2, &twice, sd, &interior)) return false;
                if (interior) { memcpy(rowB.data(), tab, tabB); memcpy(seeds.data(), sd, sizeof sd); haveInterior = true; continue; }
                if (ncls >= 4096) return false;
                rowB.insert(rowB.end(), tab, tab + tabB); rowB.insert(rowB.end(), tab2, tab2 + tabB); seeds.insert(seeds.end(), sd, sd + 32);
                bsel[(size_t)s * 8 + w] = ncls++ | (twice ? 1 << 16 : 0);
            }
        g.ncls = ncls;
        std::vector<int8_t> colA((size_t)g.ksy * 64 * 16);
        sepmx::buildColA(g, ky, colA.data());
        auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
        o1 = up16(bsel.size() * 4); o2 = o1 + up16(seeds.size() * 4); o3 = o2 + up16(rowB.size());
        const size_t total = o3 + colA.size();
        std::vector<uchar> blob(total, 0);
        memcpy(blob.data(), bsel.data(), bsel.size() * 4); memcpy(blob.data() + o1, seeds.data(), seeds.size() * 4);
        memcpy(blob.data() + o2, rowB.data(), rowB.size()); memcpy(blob.data() + o3, colA.data(), colA.size());
        uchar* dd = nullptr;
        if (hipMalloc(&dd, total) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemcpy(dd, blob.data(), total, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(dd); return false; }
        if (cache.size() >= 16) {                                                     // drop the least recently used block (hipFree waits for the device: nothing still reads it)
            size_t old = 0;
            for (size_t i = 1; i < cache.size(); i++) if (cache[i].stamp < cache[old].stamp) old = i;
            (void)hipFree(cache[old].d); cache.erase(cache.begin() + old);
        }
        cache.push_back({key, dev, dd, o1, o2, o3, g.ncls, ++clock});
        d = dd;
    }
    const int ncls = g.ncls;
    const dim3 grid(nstrips, nseg, nframes);
    g.xcd = xcdEnv && ((size_t)nstrips * nseg * nframes) % 8 == 0;
    const int* dsel = reinterpret_cast<const int*>(d); const int* dseed = reinterpret_cast<const int*>(d + o1);
    const v4i* dB = reinterpret_cast<const v4i*>(d + o2); const v4i* dA = reinterpret_cast<const v4i*>(d + o3);
    if (g.box) {
        if (g.dma) launchX<true, true>(g.ksx, g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dsel, dseed, dB, dA);
        else       launchX<false, true>(g.ksx, g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dsel, dseed, dB, dA);
    } else {
        if (g.dma) launchX<true, false>(g.ksx, g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dsel, dseed, dB, dA);
        else       launchX<false, false>(g.ksx, g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dsel, dseed, dB, dA);
    }
    noteKernel("k_sepmx<%d,%d,%d%s> grid=%ux%ux%u x512 taps=%dx%d cn=%d delta=%d shift=%d classes=%d seg=%d", g.ksx, g.ksy, g.dma, g.box ? ",box" : "", grid.x, grid.y, grid.z, nx, ny, cn, g.delta, g.shift, ncls, g.seg);
    return true;
}

} // namespace mi355
