// gausskernel.hip -- host code only.  Restates the reference's bit-exact Gaussian tap generator.
//
// The reference evaluates everything in `softdouble` (software IEEE-754 binary64, round-to-nearest-
// even; modules/core/src/softfloat.cpp).  Native `double` arithmetic is the same function as long as
// no contraction/reassociation happens (this file is built with -ffp-contract=off), so the algorithm
// is restated with plain doubles:
//   * exp(): softfloat.cpp:3535-3563 f64_exp -- 2^(k/64) table x degree-5 polynomial;
//   * taps : smooth.dispatch.cpp:81-198 getGaussianKernelBitExact;
//   * Q8.8 / Q16.16 with error diffusion: smooth.dispatch.cpp:224-258 getGaussianKernelFixedPoint_ED.
#include "gausskernel.h"
#include <cmath>
#include <cstring>

namespace mi355 {
namespace {

double fromRaw(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// 2^(i/64), i = 0..63, correctly rounded (the reference tabulates the same numbers, softfloat.cpp:3435)
const double* expTab()
{
    static double tab[64];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 64; i++) tab[i] = (double)exp2l((long double)i / 64.0L);
        init = true;
    }
    return tab;
}

// softfloat.cpp:3535-3563
double sfExp(double x)
{
    if (std::isnan(x)) return x;
    if (std::isinf(x)) return x > 0 ? x : 0.0;
    const double A0c = fromRaw(0x3f83ce0f3e46f431ull);                 // EXPPOLY_32F_A0
    const double A5 = 1.0 / A0c,
                 A4 = fromRaw(0x3fe62e42fefa39f1ull) / A0c,
                 A3 = fromRaw(0x3fcebfbdff82a45aull) / A0c,
                 A2 = fromRaw(0x3fac6b08d81fec75ull) / A0c,
                 A1 = fromRaw(0x3f83b2a72b4f3cd3ull) / A0c,
                 A0 = fromRaw(0x3f55e7aa1566c2a4ull) / A0c;
    const double prescale = fromRaw(0x3ff71547652b82feull) * 64.0;     // 1/ln2 * 2^6
    const double postscale = 1.0 / 64.0;
    const double maxval = 3000.0 * 64.0;
    uint64_t raw; memcpy(&raw, &x, 8);
    int e = (int)((raw >> 52) & 0x7FF);
    double x0 = e > 1023 + 10 ? (x < 0 ? -maxval : maxval) : x * prescale;
    int val0 = (int)nearbyint(x0);                                      // cvRound: half-to-even
    int t = (val0 >> 6) + 1023;
    t = t < 0 ? 0 : (t > 2047 ? 2047 : t);
    double buf = fromRaw((uint64_t)t << 52);
    x0 = (x0 - nearbyint(x0)) * postscale;
    return buf * A0c * expTab()[val0 & 63] * (((((A0 * x0 + A1) * x0 + A2) * x0 + A3) * x0 + A4) * x0 + A5);
}

} // namespace

bool gaussianKernelBitExact(int n, double sigma, std::vector<double>& r)
{
    if (n <= 0) return false;
    if (sigma <= 0) {                                                   // :89-145 hard-coded tables
        static const double k1[] = {1.0}, k3[] = {0.25, 0.5, 0.25}, k5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
            k7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125},
            k9[] = {4 / 256., 13 / 256., 30 / 256., 51 / 256., 60 / 256., 51 / 256., 30 / 256., 13 / 256., 4 / 256.};
        const double* k = n == 1 ? k1 : n == 3 ? k3 : n == 5 ? k5 : n == 7 ? k7 : n == 9 ? k9 : nullptr;
        if (k) { r.assign(k, k + n); return true; }
    }
    const double sd015 = fromRaw(0x3fc3333333333333ull), sd035 = fromRaw(0x3fd6666666666666ull);
    const double sigmaX = sigma > 0 ? sigma : fma((double)n, sd015, sd035);   // mulAdd == fused (softfloat f64_mulAdd)
    const double scale2X = -0.125 / (sigmaX * sigmaX);
    const int n2 = (n - 1) / 2;
    std::vector<double> values(n2 + 1);
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = sfExp((double)(x * x) * scale2X);
        values[i] = t;
        sum += t;
    }
    sum *= 2.0;
    sum += 1.0;
    if ((n & 1) == 0) sum += 1.0;
    const double mul1 = 1.0 / sum;
    r.assign(n, 0.0);
    for (int i = 0; i < n2; i++) {
        double t = values[i] * mul1;
        r[i] = t;
        r[n - 1 - i] = t;
    }
    r[n2] = 1.0 * mul1;
    if ((n & 1) == 0) r[n2 + 1] = r[n2];
    return true;
}

bool gaussianKernelFixedQ(int n, double sigma, int fractionBits, std::vector<int64_t>& out)
{
    std::vector<double> k;
    if (!(n & 1) || !gaussianKernelBitExact(n, sigma, k)) return false;
    const int64_t mult = (int64_t)1 << fractionBits;
    const double multD = (double)mult;
    out.assign(n, 0);
    const int n2 = n / 2;
    double err = 0.0;
    int64_t sum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * multD + err;
        int64_t v0 = (int64_t)nearbyint(adj);                            // cvRound(softdouble)
        err = adj - (double)v0;
        out[i] = v0;
        out[n - 1 - i] = v0;
        sum += v0;
    }
    sum *= 2;
    out[n2] = mult - sum;
    return true;
}

} // namespace mi355
