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
//   barrier | request the next 32 source rows (16-byte loads, 1-2 per lane, parked in registers for the whole step) | row pass of this step's rows from the staged block
//   (KSX MFMAs) | split into byte planes (20 VALU per 16 sums) | column pass over the last KSY tiles of row sums (2 KSY MFMAs) | (accH << 8) + accL, byte 2 into the
//   transposition block | the PREVIOUS step's output tile leaves as 1 KiB per wave-instruction (16 bytes per lane, whole 256-byte row pieces) | next rows into the other block.
// One barrier per step.  HBM traffic: the source once (+ nx - 1 columns per strip, + 32 (KSY - 1) rows per segment), the destination once.
#include "sepmx.h"
#include "sepmx_body.h"
#include <vector>

using namespace mi355;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
using sepmx::Geom;
using sepmx::TR;
using sepmx::TW;

template <int KSX, int KSY>
__global__ __launch_bounds__(512, 4) void k_sepmx(const uchar* __restrict__ src, size_t sstep, size_t sframe, uchar* __restrict__ dst, size_t dstep, size_t dframe,
                                               Geom g, const v4i* __restrict__ tabs /* row B [KSX][64], then column A [KSY][64] */)
{
    constexpr int NCHUNK = (TW - 32 + 32 * KSX) / 16, P = 16 * (NCHUNK | 1), NQ = (TR * NCHUNK + 511) / 512;
    __shared__ __attribute__((aligned(16))) uchar stage[2][TR * P];
    __shared__ __attribute__((aligned(16))) uchar tr[2][TR * TW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    src += (size_t)blockIdx.z * sframe;
    dst += (size_t)blockIdx.z * dframe;
    const int X0 = blockIdx.x * TW, y0 = blockIdx.y * g.seg;
    const int rows = min(g.seg, g.H - y0);
    const int nU = (rows + TR - 1) / TR, nT = nU + KSY - 1;

    v4i Bx[KSX], Ay[KSY];
#pragma unroll
    for (int k = 0; k < KSX; k++) Bx[k] = tabs[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < KSY; k++) Ay[k] = tabs[(KSX + k) * 64 + lane];

    // staging: chunk q = tid + 512 j of the step's TR x NCHUNK chunks
    uint4 park[NQ];
    auto request = [&](int t) {
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const int q = tid + 512 * j;
            if (q >= TR * NCHUNK) continue;
            const int r = q / NCHUNK, c = q - r * NCHUNK;
            const uchar* p = nullptr;
            uchar tmp[16];
            if (sepmx::stageChunk(g, src, sstep, X0, y0, t, r, c, &p, tmp)) __builtin_memcpy(&park[j], tmp, 16);
            else __builtin_memcpy(&park[j], p, 16);
        }
    };
    auto deposit = [&](int t) {
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const int q = tid + 512 * j;
            if (q >= TR * NCHUNK) continue;
            const int r = q / NCHUNK, c = q - r * NCHUNK;
            uint4 v = park[j];
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;          // pixels - 128
            *reinterpret_cast<uint4*>(&stage[t & 1][r * P + 16 * c]) = v;
        }
    };
    // the output tile of step t - 1 (rows y0 + 32 u ..), 4 rows of 256 bytes per wave
    auto emit = [&](int u, int buf) {
        const int rr = 4 * wave + (lane >> 4), cc = 16 * (lane & 15);
        const int y = TR * u + rr, x = X0 + cc;
        if (y >= rows || x >= g.WE) return;
        const uint4 v = *reinterpret_cast<const uint4*>(&tr[buf][rr * TW + cc]);
        uchar* d = dst + (size_t)(y0 + y) * dstep + x;
        if (x + 16 <= g.WE) __builtin_memcpy(d, &v, 16);
        else { uchar b[16]; __builtin_memcpy(b, &v, 16); for (int i = 0; i < g.WE - x; i++) d[i] = b[i]; }
    };

    v4i ringH[KSY], ringL[KSY];
#pragma unroll
    for (int k = 0; k < KSY; k++) { ringH[k] = v4i{0, 0, 0, 0}; ringL[k] = v4i{0, 0, 0, 0}; }

    request(0);
    deposit(0);
    for (int t = 0; t < nT; t++) {
        __syncthreads();
        if (t + 1 < nT) request(t + 1);
        // ---- row pass
        v16i acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = g.accR0;
        const uchar* A = &stage[t & 1][n * P + 32 * wave + 16 * h];
#pragma unroll
        for (int k = 0; k < KSX; k++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i*>(A + 32 * k), Bx[k], acc, 0, 0, 0);
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
        const int u = t - (KSY - 1);
        if (u >= 0) {
            // ---- column pass
            v16i aH, aL;
#pragma unroll
            for (int i = 0; i < 16; i++) { aH[i] = 0; aL[i] = g.accL0; }
#pragma unroll
            for (int k = 0; k < KSY; k++) {
                aH = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ay[k], ringH[k], aH, 0, 0, 0);
                aL = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ay[k], ringL[k], aL, 0, 0, 0);
            }
            uchar* T = &tr[t & 1][32 * wave + n];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned v = ((unsigned)aH[i] << 8) + (unsigned)aL[i];
                T[sepmx::regRow(h, i) * TW] = (uchar)(v >> 16);
            }
            if (u >= 1) emit(u - 1, (t - 1) & 1);
        }
        if (t + 1 < nT) deposit(t + 1);
    }
    __syncthreads();
    emit(nU - 1, (nT - 1) & 1);
}

template <int KSX>
void launchY(int ksy, dim3 grid, hipStream_t st, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, const Geom& g, const v4i* tabs)
{
    switch (ksy) {
    case 2:  hipLaunchKernelGGL((k_sepmx<KSX, 2>), grid, dim3(512), 0, st, src, sstep, sframe, dst, dstep, dframe, g, tabs); break;
    case 3:  hipLaunchKernelGGL((k_sepmx<KSX, 3>), grid, dim3(512), 0, st, src, sstep, sframe, dst, dstep, dframe, g, tabs); break;
    case 4:  hipLaunchKernelGGL((k_sepmx<KSX, 4>), grid, dim3(512), 0, st, src, sstep, sframe, dst, dstep, dframe, g, tabs); break;
    default: hipLaunchKernelGGL((k_sepmx<KSX, 5>), grid, dim3(512), 0, st, src, sstep, sframe, dst, dstep, dframe, g, tabs); break;
    }
}

} // namespace

namespace mi355 {

bool sepmxRun(Stager& stg, const uchar* src, size_t sstep, size_t sframe, uchar* dst, size_t dstep, size_t dframe, int nframes,
              int W, int H, int cn, int fullW, int fullH, int offX, int offY, int border, const uint16_t* kx, int nx, int ax, const uint16_t* ky, int ny, int ay, hipStream_t st)
{
    if (nframes < 1 || nframes > 65535 || border < 0 || border > B_REFLECT_101 || ax < 0 || ax >= nx || ay < 0 || ay >= ny) return false;
    Geom g;
    memset(&g, 0, sizeof g);
    g.W = W; g.H = H; g.cn = cn; g.fullW = fullW; g.fullH = fullH; g.offX = offX; g.offY = offY; g.border = border; g.nx = nx; g.ny = ny; g.ax = ax; g.ay = ay;
    if (!sepmx::plan(g, kx, ky, (uintptr_t)src, (sstep | sframe), nframes)) return false;
    const int nstrips = (g.WE + TW - 1) / TW, nseg = (H + g.seg - 1) / g.seg;
    if (nseg > 65535) return false;
    std::vector<int8_t> tab((size_t)(g.ksx + g.ksy) * 64 * 16);
    sepmx::buildRowB(g, kx, tab.data());
    sepmx::buildColA(g, ky, tab.data() + (size_t)g.ksx * 64 * 16);
    const v4i* dt = static_cast<const v4i*>(stg.param(tab.data(), tab.size()));
    if (!dt) return false;
    const dim3 grid(nstrips, nseg, nframes);
    switch (g.ksx) {
    case 2:  launchY<2>(g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 3:  launchY<3>(g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    case 4:  launchY<4>(g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    default: launchY<5>(g.ksy, grid, st, src, sstep, sframe, dst, dstep, dframe, g, dt); break;
    }
    noteKernel("k_sepmx<%d,%d> grid=%ux%ux%u x512 taps=%dx%d cn=%d delta=%d seg=%d", g.ksx, g.ksy, grid.x, grid.y, grid.z, nx, ny, cn, g.delta, g.seg);
    return true;
}

} // namespace mi355
