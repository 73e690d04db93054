// sepmx_emu.cpp -- the HOST half of k_sepmx (opencv_amd/csrc/sepmx_body.h: plan, the border-folded row-pass matrices with their second product, the column matrix, the
// seeds of the bias algebra) run on the CPU at matrix level: every product the kernel hands to v_mfma_i32_32x32x32_i8 is evaluated as a plain integer matrix product over the
// SAME tables the product uploads, columns outside the image are filled with garbage (their weight must be zero), and the byte-plane algebra / finish is the kernel's.  What
// this does not cover is the instruction's lane <-> element map (regRow), which only the GPU tests can pin.  Test infrastructure; built by tests/test_sepmx_emu.py.
#include "sepmx_body.h"
#include <vector>
#include <cmath>
#include <cstdio>

using namespace sepmx;

extern "C" int emu_sepmx(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int W, int H, int cn, int fullW, int fullH, int offX, int offY, int border,
                         const uint16_t* kx, int nx, int ax, const uint16_t* ky, int ny, int ay, int box, int divScale, int divDelta, float scaleF, double scaleD,
                         unsigned long long fakeAddr, int* info /* ksx, ksy, delta, shift, dma, classes */)
{
    Geom g; memset(&g, 0, sizeof g);
    g.W = W; g.H = H; g.cn = cn; g.fullW = fullW; g.fullH = fullH; g.offX = offX; g.offY = offY; g.border = border; g.nx = nx; g.ny = ny; g.ax = ax; g.ay = ay;
    g.box = box; g.divScale = divScale; g.divDelta = divDelta; g.scaleF = scaleF; g.scaleD = scaleD; g.tailStart = (W * cn) & ~7;
    if (!plan(g, kx, ky, (uintptr_t)fakeAddr, sstep, 0, 1)) return 1;
    const int nstrips = (g.WE + g.shift + TW - 1) / TW;
    std::vector<int8_t> colA((size_t)g.ksy * 64 * 16);
    buildColA(g, ky, colA.data());
    // the column matrix back from its operand layout: Ay[s][m][k] with k = regRow(h, i) + 16 * 0 ... : lane (m, h), byte i <-> tile row regRow(h, i)
    auto ayAt = [&](int s, int m, int krow) -> int { for (int h = 0; h < 2; h++) for (int i = 0; i < 16; i++) if (regRow(h, i) == krow) return colA[((size_t)s * 64 + (h * 32 + m)) * 16 + i]; return 0; };
    int ncls = 1;
    const int rowsR = H + ny - 1;                                    // row-sum rows rho = 0 .. H + ny - 2 <-> ROI source row rho - ay
    for (int s = 0; s < nstrips; s++)
        for (int w = 0; w < 8; w++) {
            const int X0 = s * TW - g.shift, e0 = X0 + 32 * w;
            if (e0 + 31 < 0 || e0 >= g.WE) continue;
            static int8_t tab[MAXKSX * 64 * 16], tab2[MAXKSX * 64 * 16]; int seed[32]; bool interior, twice;
            if (!buildRowB(g, kx, g.sumKy, X0, w, tab, tab2, &twice, seed, &interior)) return 2;
            if (!interior) ncls++;
            const int K = 32 * g.ksx, win0 = e0 - g.ax * g.cn - g.delta;
            // row pass for every row-sum row
            std::vector<int> acc((size_t)rowsR * 32);
            for (int rho = 0; rho < rowsR; rho++) {
                const int yy = borderIdx(rho - g.ay + g.offY, g.fullH, g.border);
                for (int n = 0; n < 32; n++) {
                    long long a = 0;
                    for (int k = 0; k < K; k++) {
                        const int lane = (k % 32) / 16 * 32 + n, i = k % 16, ks = k / 32;
                        const int wgt = tab[((size_t)ks * 64 + lane) * 16 + i] + tab2[((size_t)ks * 64 + lane) * 16 + i];
                        int sval;
                        if (yy < 0) sval = 0;
                        else {
                            const long long f = (long long)win0 + k + (long long)g.offX * g.cn;          // element in the parent's row
                            sval = (f >= 0 && f < (long long)g.fullW * g.cn) ? src[(ptrdiff_t)(yy - g.offY) * (ptrdiff_t)sstep + win0 + k] : 0xAB;      // garbage where the kernel reads whatever lies there
                        }
                        a += (long long)wgt * (sval - 128);
                    }
                    if (a < -32768 || a > 32767) return 3;                                               // the int16 claim of the bias algebra
                    acc[(size_t)rho * 32 + n] = (int)a;
                }
            }
            // column pass, tile by tile as the kernel does it (row-sum tile u + s against Ay[s])
            for (int y = 0; y < H; y++)
                for (int n = 0; n < 32; n++) {
                    const int xe = e0 + n;
                    if (xe < 0 || xe >= g.WE) continue;
                    const int u = y / 32, m = y % 32;
                    long long aH = 0, aL = seed[n];
                    for (int st = 0; st < g.ksy; st++)
                        for (int kr = 0; kr < 32; kr++) {
                            const int rho = 32 * (u + st) + kr;
                            const int wgt = ayAt(st, m, kr);
                            if (!wgt) continue;
                            if (rho >= rowsR) return 4;                                                   // a weight on a row the walk never computes
                            const int a = acc[(size_t)rho * 32 + n];
                            const int Hh = a >> 8, L = a & 255, l = L - 128;
                            aH += (long long)wgt * Hh; aL += (long long)wgt * l;
                        }
                    const unsigned v = (unsigned)((aH << 8) + aL);
                    unsigned r;
                    if (g.box == 0) { if (v >> 24) return 5; r = (v >> 16) & 255u; }
                    else if (g.box == 1) r = (((v + (unsigned)g.divDelta) & 0xffffffu) * (unsigned)(g.divScale & 0xffffff)) >> 23;
                    else if (g.box == 3) r = v > 255u ? 255u : v;
                    else r = xe >= g.tailStart ? (unsigned)(int)fmin(rint((double)v * g.scaleD), 255.0) : (unsigned)(int)fminf(rintf((float)v * g.scaleF), 255.f);
                    dst[(size_t)y * dstep + xe] = (unsigned char)r;
                }
        }
    info[0] = g.ksx; info[1] = g.ksy; info[2] = g.delta; info[3] = g.shift; info[4] = g.dma; info[5] = ncls;
    return 0;
}
