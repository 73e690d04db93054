/* oracle/integral.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): cv::integral restated channel by channel.
 *
 * Follows hal::integral / integral_<T, ST, QT> (modules/imgproc/src/sumpixels.dispatch.cpp:191-341: the plain sum :220-234, sum + squared sum :235-256, the tilted
 * branch :257-341) and the type table of :381-406.  Where the reference walks interleaved rows with pointer offsets, this file takes one channel at a time as a
 * W x H plane P[y][j] and states what every output element is, with the order of the floating-point additions kept:
 *
 *   row prefix      s_y[j]    = s_y[j-1] + P[y][j]                                  (s_y[-1] = 0: `0 + P`, which turns a -0 into +0)
 *   sum             S[y+1][j+1] = S[y][j+1] + s_y[j],  S[0][*] = S[*][0] = 0
 *   squared sum     the same with q_y[j] = q_y[j-1] + (QT)P * (QT)P
 *   tilted branch   row 0 stores s_0[j] / q_0[j] themselves (no "0 +": the sign of a zero survives), and with
 *       b_y[j]   = b_{y-1}[j+1] + P[y][j]  (j <= W-2),  b_y[W-1] = P[y][W-1],  b_0[j] = P[0][j]      (the reference's `buf`; b[1] reads as 0 when W == 1)
 *       R[y][j]  = the tilted sum's element Tt[y+1][j+1]:   R[0][j] = P[0][j]
 *       R[y][0]   = (R[y-1][0] + P[y][0]) + b_{y-1}[1]
 *       R[y][j]   = b_{y-1}[j] + ((b_{y-1}[j+1] + P[y][j]) + R[y-1][j-1])          1 <= j <= W-2
 *       R[y][W-1] = (P[y][W-1] + b_{y-1}[W-1]) + R[y-1][W-2]                        W > 1
 *       Tt[0][*] = 0,  Tt[1][0] = 0,  Tt[y+1][0] = Tt[y][1]  (y >= 1)
 *
 * The reference's vector bodies (sumpixels.simd.hpp: uchar -> int / float / double without squared or tilted sums) add exact integers, so they agree with the above
 * wherever the sums stay exactly representable -- always for int (wrapping) and double, for float while H * W * 255 < 2^24; beyond that their scalar tails recover the
 * row prefix by a subtraction (:528-533) and the result depends on the vector width of the CPU: tests/test_oracle_integral.py stays inside the exact domain there.
 * Pinned against oracle/_ref (cv::integral itself) by that test.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* depth codes as in oracle.h: 0 8U, 2 16U, 3 16S, 4 32S, 5 32F, 6 64F */
static double srcAt(const unsigned char* row, int depth, int e)
{
    switch (depth) {
    case 0: return row[e];
    case 2: return ((const unsigned short*)row)[e];
    case 3: return ((const short*)row)[e];
    case 5: return ((const float*)row)[e];
    default: return ((const double*)row)[e];
    }
}

/* one accumulator kind per output depth; int sums wrap like the reference's int arithmetic (two's complement) */
#define DEF_PLANE(NAME, ST, QT, ADDS, ADDQ, MULQ)                                                                                         \
static void NAME(const unsigned char* src, size_t sstep, int depth, int W, int H, int cn, int k,                                          \
                 ST* S, size_t sS, QT* Q, size_t sQ, ST* Tt, size_t sT)                                                                   \
{                                                                                                                                         \
    const int tilted = Tt != NULL;                                                                                                        \
    ST* b = tilted ? (ST*)calloc((size_t)W + 2, sizeof(ST)) : NULL;        /* b_{y-1} */                                                  \
    ST* bn = tilted ? (ST*)calloc((size_t)W + 2, sizeof(ST)) : NULL;                                                                      \
    for (int j = 0; j <= W; j++) { S[(size_t)j * cn + k] = 0; if (Q) Q[(size_t)j * cn + k] = 0; if (tilted) Tt[(size_t)j * cn + k] = 0; } \
    for (int y = 0; y < H; y++) {                                                                                                         \
        const unsigned char* row = src + (size_t)y * sstep;                                                                               \
        ST* Sp = S + (size_t)y * sS; ST* Sc = Sp + sS;                                                                                    \
        QT* Qp = Q ? Q + (size_t)y * sQ : NULL; QT* Qc = Q ? Qp + sQ : NULL;                                                              \
        ST s = 0; QT q = 0;                                                                                                               \
        Sc[k] = 0; if (Q) Qc[k] = 0;                                                                                                      \
        for (int j = 0; j < W; j++) {                                                                                                     \
            const double p = srcAt(row, depth, j * cn + k);                                                                               \
            const ST ps = (ST)p; const QT pq = (QT)p;                                                                                     \
            const int direct = tilted && y > 0 && j == 0;                   /* :300-301 `t0 = s = it`; everywhere else `s = 0; s += it` */ \
            s = direct ? ps : ADDS(s, ps);                                                                                                \
            q = direct ? MULQ(pq, pq) : ADDQ(q, MULQ(pq, pq));                                                                            \
            const size_t e = (size_t)(j + 1) * cn + k;                                                                                    \
            Sc[e] = (tilted && y == 0) ? s : ADDS(Sp[e], s);                                                                              \
            if (Q) Qc[e] = (tilted && y == 0) ? q : ADDQ(Qp[e], q);                                                                       \
        }                                                                                                                                 \
        if (!tilted) continue;                                                                                                            \
        ST* Tp = Tt + (size_t)y * sT; ST* Tc = Tp + sT;                      /* R[y-1][j] = Tp[(j+1) cn + k] */                           \
        if (y == 0) {                                                                                                                     \
            Tc[k] = 0;                                                                                                                    \
            for (int j = 0; j < W; j++) { const ST ps = (ST)srcAt(row, depth, j * cn + k); Tc[(size_t)(j + 1) * cn + k] = ps; b[j] = ps; } \
            b[W] = 0;                                                                                                                     \
            continue;                                                                                                                     \
        }                                                                                                                                 \
        Tc[k] = Tp[(size_t)cn + k];                                                                                                       \
        for (int j = 0; j < W; j++) {                                                                                                     \
            const ST ps = (ST)srcAt(row, depth, j * cn + k);                                                                              \
            const size_t e = (size_t)(j + 1) * cn + k;                                                                                    \
            ST r;                                                                                                                         \
            if (j == 0) r = ADDS(ADDS(Tp[e], ps), (W == 1 ? (ST)0 : b[1]));                                                               \
            else if (j < W - 1) r = ADDS(b[j], ADDS(ADDS(b[j + 1], ps), Tp[e - cn]));                                                     \
            else r = ADDS(ADDS(ps, b[j]), Tp[e - cn]);                                                                                    \
            Tc[e] = r;                                                                                                                    \
            bn[j] = j < W - 1 ? ADDS(b[j + 1], ps) : ps;                                                                                  \
        }                                                                                                                                 \
        { ST* t = b; b = bn; bn = t; }                                                                                                    \
    }                                                                                                                                     \
    free(b); free(bn);                                                                                                                    \
}

#define FADD(a, b) ((a) + (b))
#define FMUL(a, b) ((a) * (b))
static int iadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static int imul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

DEF_PLANE(plane_i_i, int, int, iadd, iadd, imul)
DEF_PLANE(plane_i_f, int, float, iadd, FADD, FMUL)
DEF_PLANE(plane_i_d, int, double, iadd, FADD, FMUL)
DEF_PLANE(plane_f_f, float, float, FADD, FADD, FMUL)
DEF_PLANE(plane_f_d, float, double, FADD, FADD, FMUL)
DEF_PLANE(plane_d_d, double, double, FADD, FADD, FMUL)

/* returns 0, or -1 for a depth triple outside the reference's table (sumpixels.dispatch.cpp:383-406).  Steps in bytes; sq / tilted may be NULL. */
int orc_integral(int depth, int sdepth, int sqdepth, const unsigned char* src, size_t sstep, unsigned char* sum, size_t sumstep,
                 unsigned char* sq, size_t sqstep, unsigned char* tilted, size_t tstep, int W, int H, int cn)
{
    const int key = depth * 100 + sdepth * 10 + sqdepth;
    switch (key) {
    case 46: case 45: case 44: case 56: case 55: case 66: case 266: case 366: case 556: case 555: case 566: case 666: break;
    default: return -1;
    }
    for (int k = 0; k < cn; k++) {
#define RUN(F, ST, QT) F(src, sstep, depth, W, H, cn, k, (ST*)sum, sumstep / sizeof(ST), (QT*)sq, sq ? sqstep / sizeof(QT) : 0, (ST*)tilted, tilted ? tstep / sizeof(ST) : 0)
        if (sdepth == 4 && sqdepth == 4) RUN(plane_i_i, int, int);
        else if (sdepth == 4 && sqdepth == 5) RUN(plane_i_f, int, float);
        else if (sdepth == 4) RUN(plane_i_d, int, double);
        else if (sdepth == 5 && sqdepth == 5) RUN(plane_f_f, float, float);
        else if (sdepth == 5) RUN(plane_f_d, float, double);
        else RUN(plane_d_d, double, double);
#undef RUN
    }
    return 0;
}
