// median5_math.h -- the 5 x 5 median of k_median_roll<5, CN> (median.hip) per lane and output row, on COLUMNS SORTED ONCE PER POSITION: a column of the
// window is shared by the five outputs that see it, so its five values are sorted once (9 exchanges) and the median of 25 is then selected from five
// sorted columns by merging instead of by a 113-exchange network on unordered values (networks generated and verified by tools/gen_median_net.py):
//   P = merge of two neighbouring sorted columns (13 exchanges, sorted 10)
//   Q = of two such pairs the ranks 7..12 of their 20 values (25 exchanges): an element with 13 or more of the 20 on one side cannot be the median of 25
//   median = rank 5 of Q and the fifth sorted column (10 exchanges)
// Pixels are packed two to a 32-bit word as 16-bit halves (planes E = even bytes, O = odd bytes of a window dword, roll.h), every exchange is one
// v_pk_min_u16 + one v_pk_max_u16.  For CN == 1 neighbouring columns are the E and O value of the same word, so P exists only for pairs that start at an even
// byte and Q only for quadruples that do: each P serves two Q, each Q two outputs (an even output takes Q to its left + the column two to its right, an odd
// one the column two to its left + Q from one to its left) -- 391 exchanges per lane row of 16 pixels against 904.  Other channel counts gather the sorted
// columns per output word (9 + 61 exchanges against 113).
// Plain C++ but for the two intrinsics: tests/hostemu/median5_emu.cpp compiles the same lines for the CPU and runs them against the pinned restatement.
#pragma once
#include <stdint.h>
#include "median_net.h"

#ifndef MI355_HD
#  if defined(__HIPCC__)
#    define MI355_HD __host__ __device__ __forceinline__
#  else
#    define MI355_HD inline
#  endif
#endif

namespace med5 {

#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
MI355_HD uint32_t pmin(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
MI355_HD uint32_t pmax(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }
MI355_HD uint32_t align16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#else
MI355_HD uint32_t pmin(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16); }
MI355_HD uint32_t pmax(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, bl = b & 0xffffu, ah = a >> 16, bh = b >> 16; return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16); }
MI355_HD uint32_t align16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }          // the upper half of lo below the lower half of hi
#endif

#define MED5_CE(a, b) { const uint32_t t_ = med5::pmin(v[a], v[b]); v[b] = med5::pmax(v[a], v[b]); v[a] = t_; }

// the packed pair at byte distance S from the output pair (plane Q, window word k + HD) in the planes E / O of one row: roll::pairAt's index arithmetic
constexpr int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
constexpr int pmod2(int a) { return ((a % 2) + 2) % 2; }
template <int Q, int S, int HD>
MI355_HD uint32_t pairAt(const uint32_t* E, const uint32_t* O, int k)
{
    constexpr int q2 = pmod2(Q + S);
    constexpr int f = fdiv2(Q + S);
    const uint32_t* P = q2 ? O : E;
    if constexpr (pmod2(f) == 0) return P[k + f / 2 + HD];
    else { constexpr int lo = fdiv2(f - 1); return align16(P[k + lo + 1 + HD], P[k + lo + HD]); }
}

// one channel, window of 6 dwords (HD = 1): ring[j].E[d] / .O[d] = the planes of row j of the window; o[k] = the four medians of own dword k
template <class RowP>
MI355_HD void row1(const RowP (&ring)[5], uint32_t (&o)[4])
{
    constexpr int M55[10] = MI355_MERGE55_OUT, M6[6] = MI355_MID6_OUT;
    uint32_t sE[6][5], sO[6][5];          // sorted columns of window words 0..5 (ascending)
    uint32_t p[6][10];                    // p[d]: columns (4d, 4d+1) | (4d+2, 4d+3) merged, ascending
    uint32_t q[5][6];                     // q[d]: ranks 7..12 of columns 4d..4d+3 | 4d+2..4d+5
#pragma unroll
    for (int d = 0; d < 6; d++) {
        {
            uint32_t v[5];
#pragma unroll
            for (int j = 0; j < 5; j++) v[j] = ring[j].E[d];
            MI355_SORT5(MED5_CE)
#pragma unroll
            for (int j = 0; j < 5; j++) sE[d][j] = v[j];
#pragma unroll
            for (int j = 0; j < 5; j++) v[j] = ring[j].O[d];
            MI355_SORT5(MED5_CE)
#pragma unroll
            for (int j = 0; j < 5; j++) sO[d][j] = v[j];
        }
        {
            uint32_t v[10];
#pragma unroll
            for (int j = 0; j < 5; j++) { v[j] = sE[d][j]; v[5 + j] = sO[d][j]; }
            MI355_MERGE55(MED5_CE)
#pragma unroll
            for (int i = 0; i < 10; i++) p[d][i] = v[M55[i]];
        }
        if (d >= 1) {
            uint32_t v[20];
#pragma unroll
            for (int i = 0; i < 10; i++) { v[i] = p[d - 1][i]; v[10 + i] = align16(p[d][i], p[d - 1][i]); }
            MI355_MID6(MED5_CE)
#pragma unroll
            for (int i = 0; i < 6; i++) q[d - 1][i] = v[M6[i]];
        }
        if (d >= 2) {
            const int w = d - 1;          // window word of own dword w - 1
            uint32_t v[11];
            // even outputs (bytes 4w, 4w + 2): the four columns starting two to the left, and the column two to the right
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = align16(q[w][i], q[w - 1][i]);
#pragma unroll
            for (int i = 0; i < 5; i++) v[6 + i] = align16(sE[w + 1][i], sE[w][i]);
            MI355_RANK5(MED5_CE)
            const uint32_t rE = v[MI355_RANK5_OUT];
            // odd outputs (bytes 4w + 1, 4w + 3): the column two to the left, and the four columns starting one to the left
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = q[w][i];
#pragma unroll
            for (int i = 0; i < 5; i++) v[6 + i] = align16(sO[w][i], sO[w - 1][i]);
            MI355_RANK5(MED5_CE)
            o[w - 1] = rE | (v[MI355_RANK5_OUT] << 8);
        }
    }
}

// any channel count: the columns of every window word sorted once into srt[r] (r-th smallest), then per output word the five sorted columns at
// distances -2 CN .. 2 CN are merged
// planes of the five rows held as window dwords X[NW] (the planes are split off where a column is sorted: 40 registers of state instead of 80)
template <int NW> struct RowX { uint32_t X[NW]; };
template <int NW> struct RowEO { uint32_t E[NW], O[NW]; };
template <int NW> MI355_HD uint32_t planeE(const RowX<NW>& r, int d) { return r.X[d] & 0x00ff00ffu; }
template <int NW> MI355_HD uint32_t planeO(const RowX<NW>& r, int d) { return (r.X[d] >> 8) & 0x00ff00ffu; }
template <class R> MI355_HD uint32_t planeE(const R& r, int d) { return r.E[d]; }
template <class R> MI355_HD uint32_t planeO(const R& r, int d) { return r.O[d]; }

template <int CN, int HD, int NW, class RowT>
MI355_HD void rowN(const RowT (&ring)[5], uint32_t (&o)[4])
{
    constexpr int M55[10] = MI355_MERGE55_OUT, M6[6] = MI355_MID6_OUT;
    static_assert(NW == 4 + 2 * HD && HD == 2, "window of 8 dwords: output word k reads the sorted columns of window words k .. k + 4");
    RowEO<NW> srt[5];
    auto sortWord = [&](int d) {
        uint32_t v[5];
#pragma unroll
        for (int j = 0; j < 5; j++) v[j] = planeE(ring[j], d);
        MI355_SORT5(MED5_CE)
#pragma unroll
        for (int j = 0; j < 5; j++) srt[j].E[d] = v[j];
#pragma unroll
        for (int j = 0; j < 5; j++) v[j] = planeO(ring[j], d);
        MI355_SORT5(MED5_CE)
#pragma unroll
        for (int j = 0; j < 5; j++) srt[j].O[d] = v[j];
    };
#pragma unroll
    for (int d = 0; d < 4; d++) sortWord(d);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sortWord(k + 4);                  // a sliding window of five sorted words: the words left of k are dead by now
        uint32_t r2[2];
#pragma unroll
        for (int qq = 0; qq < 2; qq++) {
            uint32_t c[5][5];
#pragma unroll
            for (int r = 0; r < 5; r++) {
                if (qq) { c[0][r] = pairAt<1, -2 * CN, HD>(srt[r].E, srt[r].O, k); c[1][r] = pairAt<1, -CN, HD>(srt[r].E, srt[r].O, k); c[2][r] = pairAt<1, 0, HD>(srt[r].E, srt[r].O, k);
                          c[3][r] = pairAt<1, CN, HD>(srt[r].E, srt[r].O, k); c[4][r] = pairAt<1, 2 * CN, HD>(srt[r].E, srt[r].O, k); }
                else    { c[0][r] = pairAt<0, -2 * CN, HD>(srt[r].E, srt[r].O, k); c[1][r] = pairAt<0, -CN, HD>(srt[r].E, srt[r].O, k); c[2][r] = pairAt<0, 0, HD>(srt[r].E, srt[r].O, k);
                          c[3][r] = pairAt<0, CN, HD>(srt[r].E, srt[r].O, k); c[4][r] = pairAt<0, 2 * CN, HD>(srt[r].E, srt[r].O, k); }
            }
            uint32_t pq[20];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                uint32_t v[10];
#pragma unroll
                for (int j = 0; j < 5; j++) { v[j] = c[2 * h][j]; v[5 + j] = c[2 * h + 1][j]; }
                MI355_MERGE55(MED5_CE)
#pragma unroll
                for (int i = 0; i < 10; i++) pq[10 * h + i] = v[M55[i]];
            }
            uint32_t v[11];
            {
                uint32_t* vv = pq;
#define MED5_CEQ(a, b) { const uint32_t t_ = med5::pmin(vv[a], vv[b]); vv[b] = med5::pmax(vv[a], vv[b]); vv[a] = t_; }
                MI355_MID6(MED5_CEQ)
#undef MED5_CEQ
            }
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = pq[M6[i]];
#pragma unroll
            for (int j = 0; j < 5; j++) v[6 + j] = c[4][j];
            MI355_RANK5(MED5_CE)
            r2[qq] = v[MI355_RANK5_OUT];
        }
        o[k] = r2[0] | (r2[1] << 8);
    }
}

} // namespace med5
