/* oracle/gausskernel.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the reference's bit-exact Gaussian tap generation with native IEEE doubles
 * (built with -ffp-contract=off; softdouble == IEEE binary64 round-to-nearest-even):
 *   modules/core/src/softfloat.cpp:3535-3563   f64_exp (64-entry 2^(k/64) table x degree-5 polynomial)
 *   modules/imgproc/src/smooth.dispatch.cpp:81-198  getGaussianKernelBitExact
 *   modules/imgproc/src/smooth.dispatch.cpp:224-258 getGaussianKernelFixedPoint_ED */
#include "oracle.h"
#include <math.h>
#include <string.h>

static double raw64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static double orc_exp(double x)
{
    static double tab[64];
    static int init = 0;
    if (!init) { for (int i = 0; i < 64; i++) tab[i] = (double)powl(2.0L, (long double)i / 64.0L); init = 1; }
    if (isnan(x)) return x;
    if (isinf(x)) return x > 0 ? x : 0.0;
    const double c0 = raw64(0x3f83ce0f3e46f431ull);
    const double a5 = 1.0 / c0, a4 = raw64(0x3fe62e42fefa39f1ull) / c0, a3 = raw64(0x3fcebfbdff82a45aull) / c0,
                 a2 = raw64(0x3fac6b08d81fec75ull) / c0, a1 = raw64(0x3f83b2a72b4f3cd3ull) / c0,
                 a0 = raw64(0x3f55e7aa1566c2a4ull) / c0;
    const double pre = raw64(0x3ff71547652b82feull) * 64.0;
    uint64_t bits; memcpy(&bits, &x, 8);
    int ex = (int)((bits >> 52) & 0x7FF);
    double x0 = ex > 1033 ? (x < 0 ? -192000.0 : 192000.0) : x * pre;
    int v = (int)rint(x0);
    int t = (v >> 6) + 1023;
    if (t < 0) t = 0;
    if (t > 2047) t = 2047;
    double buf = raw64((uint64_t)t << 52);
    x0 = (x0 - rint(x0)) * (1.0 / 64.0);
    return buf * c0 * tab[v & 63] * (((((a0 * x0 + a1) * x0 + a2) * x0 + a3) * x0 + a4) * x0 + a5);
}

int orc_getGaussianKernel(int n, double sigma, double* r)
{
    if (n <= 0) return 1;
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7 || n == 9)) {
        uint16_t q[9];
        orc_binomialTapsQ8(n, q);
        for (int i = 0; i < n; i++) r[i] = q[i] / 256.0;        /* :89-145, all exactly representable */
        return 0;
    }
    double sigmaX = sigma > 0 ? sigma : fma((double)n, raw64(0x3fc3333333333333ull), raw64(0x3fd6666666666666ull));
    double scale2X = -0.125 / (sigmaX * sigmaX);
    int n2 = (n - 1) / 2;
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) { r[i] = orc_exp((double)(x * x) * scale2X); sum += r[i]; }
    sum *= 2.0;
    sum += 1.0;
    if (!(n & 1)) sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) { double t = r[i] * mul1; r[i] = t; r[n - 1 - i] = t; }
    r[n2] = 1.0 * mul1;
    if (!(n & 1)) r[n2 + 1] = r[n2];
    return 0;
}

int orc_getGaussianKernelQ(int n, double sigma, int fractionBits, int64_t* q)
{
    double k[256];
    if (n > 256 || !(n & 1) || orc_getGaussianKernel(n, sigma, k)) return 1;
    int64_t mult = (int64_t)1 << fractionBits, sum = 0;
    double err = 0.0;
    int n2 = n / 2;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * (double)mult + err;
        int64_t v0 = (int64_t)rint(adj);
        err = adj - (double)v0;
        q[i] = v0; q[n - 1 - i] = v0;
        sum += v0;
    }
    q[n2] = mult - 2 * sum;
    return 0;
}
