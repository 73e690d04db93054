// median5_emu.cpp -- opencv_amd/csrc/median5_math.h (the 5 x 5 median of k_median_roll<5, CN> per lane and output row) compiled for the CPU: every
// 16-byte chunk of every row is given the window a lane of the kernel sees (its own four dwords and HD halo dwords per side, BORDER_REPLICATE per
// channel), split into the even / odd byte planes, and run through the same lines.  tests/test_hostemu.py compares with the pinned restatement.
// Test infrastructure.
#include "median5_math.h"
#include <algorithm>
#include <cstddef>
#include <vector>

namespace {
template <int NW> struct RowP { uint32_t E[NW], O[NW]; };

template <int CN>
void run(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int W, int H)
{
    constexpr int HB = 2 * CN, HD = (HB + 3) / 4, NW = 4 + 2 * HD;
    const int rowBytes = W * CN, nchunks = (rowBytes + 15) / 16;
    auto at = [&](int y, long a) -> unsigned {          // byte a of row y with the border replicated per channel
        const int yy = std::min(std::max(y, 0), H - 1);
        long x = a >= 0 ? a / CN : -((-a + CN - 1) / CN);
        const int ch = (int)(a - x * CN);
        x = std::min<long>(std::max<long>(x, 0), W - 1);
        return src[(size_t)yy * sstep + (size_t)x * CN + ch];
    };
    for (int y = 0; y < H; y++)
        for (int c = 0; c < nchunks; c++) {
            RowP<NW> ring[5];
            for (int j = 0; j < 5; j++)
                for (int d = 0; d < NW; d++) {
                    const long a0 = 16L * c - 4 * HD + 4 * d;
                    const uint32_t X = at(y + j - 2, a0) | at(y + j - 2, a0 + 1) << 8 | at(y + j - 2, a0 + 2) << 16 | at(y + j - 2, a0 + 3) << 24;
                    ring[j].E[d] = X & 0x00ff00ffu; ring[j].O[d] = (X >> 8) & 0x00ff00ffu;
                }
            uint32_t o[4];
            if constexpr (CN == 1) med5::row1(ring, o); else med5::rowN<CN, HD, NW>(ring, o);
            for (int b = 0; b < 16 && 16 * c + b < rowBytes; b++) dst[(size_t)y * dstep + 16 * c + b] = (unsigned char)(o[b >> 2] >> (8 * (b & 3)));
        }
}
}

extern "C" int emu_median5(const unsigned char* src, size_t sstep, unsigned char* dst, size_t dstep, int W, int H, int cn)
{
    if (cn == 1) run<1>(src, sstep, dst, dstep, W, H);
    else if (cn == 3) run<3>(src, sstep, dst, dstep, W, H);
    else if (cn == 4) run<4>(src, sstep, dst, dstep, W, H);
    else return -1;
    return 0;
}
