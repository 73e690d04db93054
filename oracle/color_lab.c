/* oracle/color_lab.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restates the CV_8U CIE L*a*b* conversions behind cv_hal_cvtBGRtoLab / cv_hal_cvtLabtoBGR (hal_replacement.hpp:535-565), the bit-exact
 * integer paths the reference takes for 8-bit images:
 *   - forward   RGB2Lab_b      color_lab.cpp:1573-1892 (scalar tail :1840-1853; the vector body computes the same integers)
 *   - inverse   Lab2RGBinteger color_lab.cpp:2399-2691 (process() :2441-2493), which Lab2RGB_b :2713-2730 calls since enableBitExactness
 *   - tables    createLabTabs  color_lab.cpp:1234-1342 (gamma :1023-1040, initLUTforABXZ :1086-1110), constants :941-1020
 * The tables come from softfloat arithmetic in the reference.  softfloat's +, -, *, /, mulAdd and its conversions are IEEE-754 operations, so plain
 * C float / double arithmetic (compiled with -ffp-contract=off) gives the same bits.  Two functions are approximations and are restated as such:
 *   - cbrt(softfloat)      softfloat.cpp:3897-3931: Turkowski's quartic rational on the mantissa, evaluated in double, TRUNCATED to 23 bits;
 *   - pow(softdouble, ..)  softfloat.cpp:3961-3987 = exp(y log x) with table + polynomial, accurate to a few ulp of double.  Its result is rounded to
 *                          float and then to an integer table entry, so the C library's pow (also within an ulp) yields the same entries; the pin is
 *                          the exhaustive comparison of all 2^24 inputs against the reference in tests/test_oracle_lab.py. */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

enum { LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT, CBRT_TAB_B = 256 * 3 / 2 * (1 << GAMMA_SHIFT),
       INV_GAMMA_SHIFT = 12, INV_GAMMA_TAB = 1 << INV_GAMMA_SHIFT, LUT_BASE = 1 << 14, LAB_BASE = 1 << 14, MIN_AB = -8145, ABXZ_N = LAB_BASE * 9 / 4 };

static uint16_t sRGBGammaTab_b[256], linearGammaTab_b[256], LabCbrtTab_b[CBRT_TAB_B];
static uint16_t sRGBInvGammaTab_b[INV_GAMMA_TAB], linearInvGammaTab_b[INV_GAMMA_TAB], LabToYF_b[512];
static int abToXZ_b[ABXZ_N];
static int tablesReady;

static const double D65[3] = {0.950456, 1.0, 1.088754};
static const double sRGB2XYZ_D65[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
static const double XYZ2sRGB_D65[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};

/* f32_cbrt (softfloat.cpp:3897): x = fr * 2^(3 e'), 0.125 <= fr < 1; fr -> P(fr) / Q(fr) in double; the quotient's top 23 fraction bits are kept */
static float cbrtTurkowski(float x)
{
    uint32_t v; memcpy(&v, &x, 4);
    if ((v & 0x7fffffffu) == 0) return 0.f;
    const int s = (int)(v >> 31);
    int ex = (int)((v >> 23) & 255) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3 - 1;
    uint64_t fb = ((uint64_t)(shx + 1023) << 52) | ((uint64_t)(v & 0x7fffffu) << 29);
    double fr; memcpy(&fr, &fb, 8);
    const double num = (((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr
                        + 13.43250139086239872172837314) * fr + 0.1636161226585754240958355063;
    const double den = (((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr
                        + 33.9905941350215598754191872) * fr + 1.0;
    fr = num / den;
    memcpy(&fb, &fr, 8);
    const uint32_t out = ((uint32_t)s << 31) | ((uint32_t)(ex + 127) << 23) | (uint32_t)((fb & 0xfffffffffffffull) >> 29);
    float y; memcpy(&y, &out, 4);
    return y;
}

/* applyGamma / applyInvGamma (color_lab.cpp:1023-1040): double arithmetic on a float argument, the result rounded to float */
static float applyGamma(float x)
{
    const double xd = x, thr = 809.0 / 20000.0, lowScale = 323.0 / 25.0, power = 12.0 / 5.0, xshift = 11.0 / 200.0;
    return (float)(xd <= thr ? xd / lowScale : pow((xd + xshift) / (1.0 + xshift), power));
}
static float applyInvGamma(float x)
{
    const double xd = x, thr = 7827.0 / 2500000.0, lowScale = 323.0 / 25.0, power = 12.0 / 5.0, xshift = 11.0 / 200.0;
    return (float)(xd <= thr ? xd * lowScale : pow(xd, 1.0 / power) * (1.0 + xshift) - xshift);
}

static void buildTables(void)
{
    if (tablesReady) return;
    const float f255 = 255.f;
    const float intScale = (float)(255 * (1 << GAMMA_SHIFT));
    for (int i = 0; i < 256; i++) {
        const float x = (float)i / f255;
        sRGBGammaTab_b[i] = (uint16_t)lrintf(intScale * applyGamma(x));
        linearGammaTab_b[i] = (uint16_t)(i * (1 << GAMMA_SHIFT));
    }
    const float invScale = 1.f / (float)INV_GAMMA_TAB;
    for (int i = 0; i < INV_GAMMA_TAB; i++) {
        const float x = invScale * (float)i;
        sRGBInvGammaTab_b[i] = (uint16_t)lrintf(f255 * applyInvGamma(x));
        linearInvGammaTab_b[i] = (uint16_t)(int)(f255 * x);
    }
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    const float cbTabScale = 1.f / (f255 * (float)(1 << GAMMA_SHIFT)), lshift2 = (float)(1 << LAB_SHIFT2);
    for (int i = 0; i < CBRT_TAB_B; i++) {
        const float x = cbTabScale * (float)i;
        LabCbrtTab_b[i] = (uint16_t)lrintf(lshift2 * (x < lthresh ? fmaf(x, lscale, lbias) : cbrtTurkowski(x)));
    }
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) {
            y = (int)lrintf((float)(i * LUT_BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            const float t = 16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29);
            ify = (int)lrintf((float)LUT_BASE * t);
        } else {
            const float a = (float)(i * 100 * LUT_BASE) / (float)(255 * 116), b = (float)(16 * LUT_BASE) / 116.f;
            const float fy = a + b;
            ify = (int)lrintf(fy);
            const float f2 = fy * fy, f3 = f2 * fy;
            y = (int)lrintf(f3 / (float)(LUT_BASE * LUT_BASE));
        }
        LabToYF_b[i * 2] = (uint16_t)y;
        LabToYF_b[i * 2 + 1] = (uint16_t)ify;
    }
    for (int i = MIN_AB; i < ABXZ_N + MIN_AB; i++) {
        int v;
        if (i <= 3390) v = i * 108 / 841 - LUT_BASE * 16 / 116 * 108 / 841;
        else v = i * i / LUT_BASE * i / LUT_BASE;
        abToXZ_b[i - MIN_AB] = v;
    }
    tablesReady = 1;
}

static int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
static uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* RGB2Lab_b: coefficients :1590-1606 (rows divided by the white point, scaled by 2^12, rounded half to even), per pixel :1840-1853 */
void orc_cvtBGRtoLab8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue, int srgb)
{
    buildTables();
    const int blueIdx = swapBlue ? 2 : 0;
    int C[9];
    for (int i = 0; i < 3; i++) {
        const double c0 = sRGB2XYZ_D65[i * 3], c1 = sRGB2XYZ_D65[i * 3 + 1], c2 = sRGB2XYZ_D65[i * 3 + 2], ls = (double)(1 << LAB_SHIFT);
        C[i * 3 + (blueIdx ^ 2)] = (int)lrint(ls * c0 / D65[i]);
        C[i * 3 + 1] = (int)lrint(ls * c1 / D65[i]);
        C[i * 3 + blueIdx] = (int)lrint(ls * c2 / D65[i]);
    }
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const uint16_t* tab = srgb ? sRGBGammaTab_b : linearGammaTab_b;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstep;
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++, s += scn, d += 3) {
            const int R = tab[s[0]], G = tab[s[1]], B = tab[s[2]];
            const int fX = LabCbrtTab_b[descale(R * C[0] + G * C[1] + B * C[2], LAB_SHIFT)];
            const int fY = LabCbrtTab_b[descale(R * C[3] + G * C[4] + B * C[5], LAB_SHIFT)];
            const int fZ = LabCbrtTab_b[descale(R * C[6] + G * C[7] + B * C[8], LAB_SHIFT)];
            d[0] = sat8(descale(Lscale * fY + Lshift, LAB_SHIFT2));
            d[1] = sat8(descale(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
            d[2] = sat8(descale(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
        }
    }
}

/* Lab2RGBinteger: coefficients :2415-2437, per pixel process() :2441-2493 and the store order of operator() :2675-2686 */
void orc_cvtLabtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int srgb)
{
    buildTables();
    const int blueIdx = swapBlue ? 2 : 0, BASE = 1 << 14, shift = LAB_SHIFT + (14 - INV_GAMMA_SHIFT);
    int C[9];
    for (int i = 0; i < 3; i++) {
        const double c0 = XYZ2sRGB_D65[i], c1 = XYZ2sRGB_D65[i + 3], c2 = XYZ2sRGB_D65[i + 6], ls = (double)(1 << LAB_SHIFT);
        C[i + blueIdx * 3] = (int)lrint(ls * c0 * D65[i]);
        C[i + 3] = (int)lrint(ls * c1 * D65[i]);
        C[i + (blueIdx ^ 2) * 3] = (int)lrint(ls * c2 * D65[i]);
    }
    for (int yy = 0; yy < h; yy++) {
        const uint8_t* s = src + (size_t)yy * sstep;
        uint8_t* d = dst + (size_t)yy * dstep;
        for (int xx = 0; xx < w; xx++, s += 3, d += dcn) {
            const int LL = s[0], aa = s[1], bb = s[2];
            const int y = LabToYF_b[LL * 2], ify = LabToYF_b[LL * 2 + 1];
            const int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
            const int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
            const int x = abToXZ_b[ify + adiv - MIN_AB], z = abToXZ_b[ify - bdiv - MIN_AB];
            int ro = descale(C[0] * x + C[1] * y + C[2] * z, shift);
            int go = descale(C[3] * x + C[4] * y + C[5] * z, shift);
            int bo = descale(C[6] * x + C[7] * y + C[8] * z, shift);
            ro = ro < 0 ? 0 : ro > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : ro;
            go = go < 0 ? 0 : go > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : go;
            bo = bo < 0 ? 0 : bo > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : bo;
            if (srgb) { ro = sRGBInvGammaTab_b[ro]; go = sRGBInvGammaTab_b[go]; bo = sRGBInvGammaTab_b[bo]; }
            else { ro = ((ro << 8) - ro) >> INV_GAMMA_SHIFT; go = ((go << 8) - go) >> INV_GAMMA_SHIFT; bo = ((bo << 8) - bo) >> INV_GAMMA_SHIFT; }
            d[0] = sat8(bo); d[1] = sat8(go); d[2] = sat8(ro);
            if (dcn == 4) d[3] = 255;
        }
    }
}

/* ----------------------------------------------------------------------------------------------------------------- L*u*v*, CV_8U
 *   - forward (sRGB only)  RGB2Luvinterpolate color_lab.cpp:3276-3376: trilinear interpolation (trilinearInterpolate :1352-1392) in a 33^3 table of
 *                          (L, u, v) as 14-bit fixed point, built by initLUTforLABLUVs16 :1126-1232 in softfloat arithmetic; RGB2Luv_b :3378-3400 takes it
 *                          for sRGB sources (linear RGB goes through the float path, which is not restated: the hook declines it)
 *   - inverse              Luv2RGBinteger :3556-3915 (process() :3587-3646), tables from initLUTforLUV :1043-1084
 * The vector bodies of the reference replace the scalar code's divisions by 2^14 with arithmetic shifts; the two differ only for negative
 * intermediates, which the clamp to [0, 2] that follows maps to the same value -- tests/test_oracle_lab.py compares every colour. */
enum { LUT_DIM = 33, TRI_SHIFT = 4, TRI_BASE = 1 << TRI_SHIFT };
static int16_t RGB2LuvLUT[LUT_DIM * LUT_DIM * LUT_DIM * 3], RGB2LabLUT[LUT_DIM * LUT_DIM * LUT_DIM * 3];
static int LuToUp_b[256 * 256], LvToVp_b[256 * 256];
static int luvReady;

static float fmax2(float a, float b) { return a > b ? a : b; }          /* softfloat max(a, b) = a > b ? a : b */

static void buildLuvTables(void)
{
    if (luvReady) return;
    buildTables();
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    const float uLow = -134.f, uRange = 220.f - -134.f, vLow = -140.f, vRange = 122.f - -140.f, f255 = 255.f;
    float dd = (float)(D65[0] + D65[1] * 15.0 + D65[2] * 3.0);
    dd = 1.f / fmax2(dd, 1.1920928955078125e-7f);
    const float un = dd * 52.f * (float)D65[0], vn = dd * 117.f * (float)D65[1];
    float C[9];
    for (int i = 0; i < 3; i++) { C[i * 3 + 2] = (float)sRGB2XYZ_D65[i * 3]; C[i * 3 + 1] = (float)sRGB2XYZ_D65[i * 3 + 1]; C[i * 3] = (float)sRGB2XYZ_D65[i * 3 + 2]; }
    float S[9];                                                     /* the Lab rows are divided by the white point, in double, then rounded */
    {
        const double sw[3] = {1.0 / D65[0], 1.0, 1.0 / D65[2]};
        for (int i = 0; i < 3; i++) { S[i * 3] = (float)(sRGB2XYZ_D65[i * 3 + 2] * sw[i]); S[i * 3 + 1] = (float)(sRGB2XYZ_D65[i * 3 + 1] * sw[i]); S[i * 3 + 2] = (float)(sRGB2XYZ_D65[i * 3] * sw[i]); }
    }
    const float f9033 = (float)(29 * 29 * 29) / 27.f;
    const float lld = (float)(LUT_DIM - 1), lbase = (float)LAB_BASE, f9of4 = 9.f / 4.f;
    for (int p = 0; p < LUT_DIM; p++)
        for (int q = 0; q < LUT_DIM; q++)
            for (int r = 0; r < LUT_DIM; r++) {
                const float R = applyGamma((float)p / lld), G = applyGamma((float)q / lld), B = applyGamma((float)r / lld);
                {                                                   /* RGB -> Lab grid (:1171-1187) */
                    float u0 = R * S[0], u1 = G * S[1], u2 = B * S[2];
                    const float X = (u0 + u1) + u2;
                    u0 = R * S[3]; u1 = G * S[4]; u2 = B * S[5];
                    const float Y = (u0 + u1) + u2;
                    u0 = R * S[6]; u1 = G * S[7]; u2 = B * S[8];
                    const float Z = (u0 + u1) + u2;
                    const float FX = X > lthresh ? cbrtTurkowski(X) : fmaf(X, lscale, lbias);
                    const float FY = Y > lthresh ? cbrtTurkowski(Y) : fmaf(Y, lscale, lbias);
                    const float FZ = Z > lthresh ? cbrtTurkowski(Z) : fmaf(Z, lscale, lbias);
                    const float L = Y > lthresh ? (116.f * FY - 16.f) : (f9033 * Y);
                    const float a = 500.f * (FX - FY), b = 200.f * (FY - FZ);
                    int16_t* e = RGB2LabLUT + p * 3 + q * LUT_DIM * 3 + r * LUT_DIM * LUT_DIM * 3;
                    e[0] = (int16_t)lrintf((lbase * L) / 100.f);
                    e[1] = (int16_t)lrintf((lbase * (a + 128.f)) / 256.f);
                    e[2] = (int16_t)lrintf((lbase * (b + 128.f)) / 256.f);
                }
                float t0 = R * C[0], t1 = G * C[1], t2 = B * C[2];
                const float X = (t0 + t1) + t2;
                t0 = R * C[3]; t1 = G * C[4]; t2 = B * C[5];
                const float Y = (t0 + t1) + t2;
                t0 = R * C[6]; t1 = G * C[7]; t2 = B * C[8];
                const float Z = (t0 + t1) + t2;
                float L = Y < lthresh ? fmaf(Y, lscale, lbias) : cbrtTurkowski(Y);
                L = L * 116.f - 16.f;
                const float a15 = 15.f * Y, a3 = 3.f * Z;
                const float d = 52.f / fmax2((X + a15) + a3, 1.1920928955078125e-7f);
                const float xd = X * d, u = L * (xd - un);
                const float yd = (f9of4 * Y) * d, v = L * (yd - vn);
                int16_t* e = RGB2LuvLUT + p * 3 + q * LUT_DIM * 3 + r * LUT_DIM * LUT_DIM * 3;
                e[0] = (int16_t)lrintf((lbase * L) / 100.f);
                e[1] = (int16_t)lrintf((lbase * (u - uLow)) / uRange);
                e[2] = (int16_t)lrintf((lbase * (v - vLow)) / vRange);
            }
    for (int LL = 0; LL < 256; LL++) {
        const float L = (float)(LL * 100) / f255;
        for (int uu = 0; uu < 256; uu++) {
            const float u = ((float)uu * uRange) / f255 + uLow;
            const float up = 9.f * (u + L * un);
            LuToUp_b[LL * 256 + uu] = (int)lrintf(up * (float)(LUT_BASE / 1024));
        }
        for (int vv = 0; vv < 256; vv++) {
            const float v = ((float)vv * vRange) / f255 + vLow;
            float vp = 0.25f / (v + L * vn);
            if (vp > 0.25f) vp = 0.25f;
            if (vp < -0.25f) vp = -0.25f;
            LvToVp_b[LL * 256 + vv] = (int)lrintf(vp * (float)(LUT_BASE * 1024));
        }
    }
    luvReady = 1;
}

/* trilinearInterpolate :1352 on the plain 33^3 x 3 table: the packed table of the reference holds, for every cell, its eight corners with the upper
 * neighbours clamped to the last grid point (fill_one :1112) */
static void trilinear(const int16_t* LUT, int cx, int cy, int cz, int* a, int* b, int* c)
{
    const int tx = cx >> (14 - 5), ty = cy >> (14 - 5), tz = cz >> (14 - 5);
    const int x = (cx >> (14 - 8 - 1)) & (TRI_BASE - 1), y = (cy >> (14 - 8 - 1)) & (TRI_BASE - 1), z = (cz >> (14 - 8 - 1)) & (TRI_BASE - 1);
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < 8; i++) {
        const int dp = i >> 2, dq = (i >> 1) & 1, dr = i & 1;
        const int pp = tx + dp < LUT_DIM - 1 ? tx + dp : LUT_DIM - 1, qq = ty + dq < LUT_DIM - 1 ? ty + dq : LUT_DIM - 1, rr = tz + dr < LUT_DIM - 1 ? tz + dr : LUT_DIM - 1;
        const int16_t* e = LUT + pp * 3 + qq * LUT_DIM * 3 + rr * LUT_DIM * LUT_DIM * 3;
        const int w = (dp ? x : TRI_BASE - x) * (dq ? y : TRI_BASE - y) * (dr ? z : TRI_BASE - z);
        acc[0] += e[0] * w; acc[1] += e[1] * w; acc[2] += e[2] * w;
    }
    *a = descale(acc[0], TRI_SHIFT * 3); *b = descale(acc[1], TRI_SHIFT * 3); *c = descale(acc[2], TRI_SHIFT * 3);
}

void orc_cvtBGRtoLuv8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue)
{
    buildLuvTables();
    const int bIdx = swapBlue ? 2 : 0, baseDiv = LAB_BASE / 256;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstep;
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++, s += scn, d += 3) {
            int L, u, v;
            trilinear(RGB2LuvLUT, s[bIdx] * baseDiv, s[1] * baseDiv, s[bIdx ^ 2] * baseDiv, &L, &u, &v);
            d[0] = sat8(L / baseDiv); d[1] = sat8(u / baseDiv); d[2] = sat8(v / baseDiv);
        }
    }
}

void orc_cvtLuvtoBGR8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int dcn, int swapBlue, int srgb)
{
    buildLuvTables();
    const int blueIdx = swapBlue ? 2 : 0, BASE = 1 << 14, shift = LAB_SHIFT + (14 - INV_GAMMA_SHIFT);
    int C[9];
    for (int i = 0; i < 3; i++) {
        const double ls = (double)(1 << LAB_SHIFT);
        C[i + blueIdx * 3] = (int)lrint(ls * XYZ2sRGB_D65[i]);
        C[i + 3] = (int)lrint(ls * XYZ2sRGB_D65[i + 3]);
        C[i + (blueIdx ^ 2) * 3] = (int)lrint(ls * XYZ2sRGB_D65[i + 6]);
    }
    for (int yy = 0; yy < h; yy++) {
        const uint8_t* s = src + (size_t)yy * sstep;
        uint8_t* d = dst + (size_t)yy * dstep;
        for (int xx = 0; xx < w; xx++, s += 3, d += dcn) {
            const int LL = s[0], uu = s[1], vv = s[2];
            const int y = LabToYF_b[LL * 2];
            const int up = LuToUp_b[LL * 256 + uu], vp = LvToVp_b[LL * 256 + vv];
            const long long xv = (long long)up * (long long)vp;
            int x = (int)(xv / BASE);
            x = (int)((long long)y * x / BASE);
            const long long vpl = (12 * 13 * 100 * (LUT_BASE / 1024)) * (long long)(vp * LL);
            long long zp = vpl - xv * (255 / 3);
            zp /= BASE;
            const long long zq = zp - (long long)(5 * 255 * BASE);
            const int zm = (int)(y * zq / BASE);
            int z = zm / 256 + zm / 65536;
            x = x < 0 ? 0 : x > 2 * BASE ? 2 * BASE : x;
            z = z < 0 ? 0 : z > 2 * BASE ? 2 * BASE : z;
            int ro = descale(C[0] * x + C[1] * y + C[2] * z, shift);
            int go = descale(C[3] * x + C[4] * y + C[5] * z, shift);
            int bo = descale(C[6] * x + C[7] * y + C[8] * z, shift);
            ro = ro < 0 ? 0 : ro > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : ro;
            go = go < 0 ? 0 : go > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : go;
            bo = bo < 0 ? 0 : bo > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : bo;
            if (srgb) { ro = sRGBInvGammaTab_b[ro]; go = sRGBInvGammaTab_b[go]; bo = sRGBInvGammaTab_b[bo]; }
            else { ro = ((ro << 8) - ro) >> INV_GAMMA_SHIFT; go = ((go << 8) - go) >> INV_GAMMA_SHIFT; bo = ((bo << 8) - bo) >> INV_GAMMA_SHIFT; }
            d[0] = sat8(bo); d[1] = sat8(go); d[2] = sat8(ro);
            if (dcn == 4) d[3] = 255;
        }
    }
}

/* ----------------------------------------------------------------------------------------------------------------- L*a*b*, CV_32F
 *   - forward, sRGB     RGB2Lab_f color_lab.cpp:1895 with useInterpolation (:1905): clip to [0, 1], round to 14-bit fixed point, trilinear interpolation in
 *                       the 33^3 RGB -> Lab grid, back to float (:1955-2045)
 *   - forward, linear   the float branch (:2047-2160): XYZ rows, cube root through the cubic spline over LabCbrtTab (splineBuild :20, splineInterpolate :50)
 *   - inverse           Lab2RGBfloat :2169-2395 (Lab2RGB_f :2694 forwards to it), sRGB through the spline over sRGBInvGammaTab
 * written in the form of the reference's VECTOR bodies (products by reciprocal constants, a*b + c as two roundings -- the SSE3 baseline this file is
 * compiled for has no fused multiply-add) for the first 8 * (width / 8) pixels of a row, and in the form of its scalar tails (divisions, sums left to
 * right, cv::cubeRoot instead of the spline) for the last width % 8: tests/test_oracle_lab.py finds the result equal to the reference's bit for bit. */
static float LabCbrtSpline[1024 * 4], InvGammaSpline[1024 * 4], GammaSpline[1024 * 4];
static int splinesReady;

static void splineBuild(const float* f, int n, float* tab)
{
    float cn = 0.f;
    tab[0] = tab[1] = 0.f;
    for (int i = 1; i < n; i++) {
        const float t = ((f[i + 1] - f[i] * 2.f) + f[i - 1]) * 3.f;
        const float l = 1.f / (4.f - tab[(i - 1) * 4]);
        tab[i * 4] = l; tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
    }
    for (int j = 0; j < n; j++) {
        const int i = n - j - 1;
        const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
        const float b = (f[i + 1] - f[i]) - (cn + c * 2.f) / 3.f;
        const float d = (cn - c) / 3.f;
        tab[i * 4] = f[i]; tab[i * 4 + 1] = b; tab[i * 4 + 2] = c; tab[i * 4 + 3] = d;
        cn = c;
    }
}

static float splineAt(float x, const float* tab, int n)
{
    int ix = (int)x;
    ix = ix < 0 ? 0 : ix > n - 1 ? n - 1 : ix;
    x -= (float)ix;
    tab += ix * 4;
    float r = tab[3] * x + tab[2];
    r = r * x + tab[1];
    return r * x + tab[0];
}

static void buildSplines(void)
{
    if (splinesReady) return;
    static float f[1025], ig[1025], g[1025];
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    const float cbScale = 1.f / ((float)(1024 * 2) / 3.f), gScale = 1.f / 1024.f;
    for (int i = 0; i <= 1024; i++) {
        const float x = cbScale * (float)i;
        f[i] = x < lthresh ? fmaf(x, lscale, lbias) : cbrtTurkowski(x);
        ig[i] = applyInvGamma(gScale * (float)i);
        g[i] = applyGamma(gScale * (float)i);
    }
    splineBuild(f, 1024, LabCbrtSpline);
    splineBuild(ig, 1024, InvGammaSpline);
    splineBuild(g, 1024, GammaSpline);
    splinesReady = 1;
}

static float clip01(float v) { return v < 0.f ? 0.f : v <= 1.f ? v : 1.f; }

/* cv::cubeRoot (core/src/mathfuncs.cpp:104-140): the same rational as above, but on a float mantissa and ROUNDED to float */
static float cubeRootRounded(float value)
{
    uint32_t vi; memcpy(&vi, &value, 4);
    const uint32_t ix = vi & 0x7fffffffu, sgn = vi & 0x80000000u;
    int ex = (int)(ix >> 23) - 127, shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    uint32_t fi = (ix & ((1u << 23) - 1)) | ((uint32_t)(shx + 127) << 23);
    float frf; memcpy(&frf, &fi, 4);
    const double fr = frf;
    const double num = (((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr
                        + 13.43250139086239872172837314) * fr + 0.1636161226585754240958355063;
    const double den = (((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr
                        + 33.9905941350215598754191872) * fr + 1.0;
    const float q = (float)(num / den);
    uint32_t qi; memcpy(&qi, &q, 4);
    qi = (qi + ((uint32_t)ex << 23) + sgn) & ((vi << 1) != 0 ? 0xffffffffu : 0u);      /* the reference's (ex << 23) on a negative ex, as unsigned arithmetic */
    float out; memcpy(&out, &qi, 4);
    return out;
}

void orc_cvtBGRtoLab32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int srgb)
{
    buildLuvTables(); buildSplines();
    const int bIdx = swapBlue ? 2 : 0;
    float C[9];
    {
        const double sw[3] = {1.0 / D65[0], 1.0, 1.0 / D65[2]};
        for (int i = 0; i < 3; i++) {
            C[i * 3 + (bIdx ^ 2)] = (float)(sw[i] * sRGB2XYZ_D65[i * 3]);
            C[i * 3 + 1] = (float)(sw[i] * sRGB2XYZ_D65[i * 3 + 1]);
            C[i * 3 + bIdx] = (float)(sw[i] * sRGB2XYZ_D65[i * 3 + 2]);
        }
    }
    const float tabScale = (float)(1024 * 2) / 3.f;
    for (int y = 0; y < h; y++) {
        const float* s = (const float*)((const uint8_t*)src + (size_t)y * sstepBytes);
        float* d = (float*)((uint8_t*)dst + (size_t)y * dstepBytes);
        for (int x = 0; x < w; x++, s += scn, d += 3) {
            if (srgb) {
                const float R = clip01(s[bIdx]), G = clip01(s[1]), B = clip01(s[bIdx ^ 2]);
                int iL, ia, ib;
                trilinear(RGB2LabLUT, (int)lrintf(R * 16384.f), (int)lrintf(G * 16384.f), (int)lrintf(B * 16384.f), &iL, &ia, &ib);
                d[0] = (float)iL * (100.0f / 16384.f);
                float t = (float)ia * (256.0f / 16384.f); d[1] = t + -128.f;
                t = (float)ib * (256.0f / 16384.f); d[2] = t + -128.f;
            } else if (x >= (w & ~7)) {                        /* the scalar tail of a row (:2132-2157): the last w % 8 pixels */
                const float R = clip01(s[0]), G = clip01(s[1]), B = clip01(s[2]), a16 = 16.f / 116.f;
                float t0 = R * C[0], t1 = G * C[1], t2 = B * C[2]; const float X = (t0 + t1) + t2;
                t0 = R * C[3]; t1 = G * C[4]; t2 = B * C[5]; const float Y = (t0 + t1) + t2;
                t0 = R * C[6]; t1 = G * C[7]; t2 = B * C[8]; const float Z = (t0 + t1) + t2;
                float FX, FY, FZ, L;
                if (X > 0.008856f) FX = cubeRootRounded(X); else { FX = 7.787f * X; FX = FX + a16; }
                if (Y > 0.008856f) FY = cubeRootRounded(Y); else { FY = 7.787f * Y; FY = FY + a16; }
                if (Z > 0.008856f) FZ = cubeRootRounded(Z); else { FZ = 7.787f * Z; FZ = FZ + a16; }
                if (Y > 0.008856f) { L = 116.f * FY; L = L - 16.f; } else L = 903.3f * Y;
                d[0] = L; d[1] = 500.f * (FX - FY); d[2] = 200.f * (FY - FZ);
            } else {
                const float R = clip01(s[0]), G = clip01(s[1]), B = clip01(s[2]);
                float t2 = B * C[2], t1 = G * C[1] + t2; const float X = R * C[0] + t1;
                t2 = B * C[5]; t1 = G * C[4] + t2; const float Y = R * C[3] + t1;
                t2 = B * C[8]; t1 = G * C[7] + t2; const float Z = R * C[6] + t1;
                const float FX = splineAt(X * tabScale, LabCbrtSpline, 1024), FY = splineAt(Y * tabScale, LabCbrtSpline, 1024), FZ = splineAt(Z * tabScale, LabCbrtSpline, 1024);
                float L;
                if (Y > 0.008856f) { L = 116.f * FY; L = L + -16.f; } else L = 903.3f * Y;
                d[0] = L; d[1] = 500.f * (FX - FY); d[2] = 200.f * (FY - FZ);
            }
        }
    }
}

void orc_cvtLabtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int srgb)
{
    buildSplines();
    const int blueIdx = swapBlue ? 2 : 0;
    float C[9];
    for (int i = 0; i < 3; i++) {
        C[i + (blueIdx ^ 2) * 3] = (float)(XYZ2sRGB_D65[i] * D65[i]);
        C[i + 3] = (float)(XYZ2sRGB_D65[i + 3] * D65[i]);
        C[i + blueIdx * 3] = (float)(XYZ2sRGB_D65[i + 6] * D65[i]);
    }
    const float lThresh = 8.f, fThresh = 6.f / 29.f, c16_116 = 16.0f / 116.0f;
    const float inv903 = 1.f / 903.3f, inv116 = 1.f / 116.0f, pinv500 = 1.f / 500.f, ninv200 = -1.f / 200.f, inv7787 = 1.f / 7.787f;
    for (int yy = 0; yy < h; yy++) {
        const float* s = (const float*)((const uint8_t*)src + (size_t)yy * sstepBytes);
        float* d = (float*)((uint8_t*)dst + (size_t)yy * dstepBytes);
        for (int xx = 0; xx < w; xx++, s += 3, d += dcn) {
            const float li = s[0], ai = s[1], bi = s[2];
            const int tail = xx >= (w & ~7);                    /* the scalar tail of a row (:2343-2385) divides where the vector body multiplies by a reciprocal */
            float y, fy;
            if (li <= lThresh) { y = tail ? li / 903.3f : li * inv903; fy = 7.787f * y; fy = fy + c16_116; }
            else { fy = tail ? (li + 16.0f) / 116.0f : (li + 16.0f) * inv116; y = fy * fy; y = y * fy; }
            float fxz[2];
            if (tail) { fxz[0] = ai / 500.0f; fxz[0] = fxz[0] + fy; fxz[1] = bi / 200.0f; fxz[1] = fy - fxz[1]; }
            else { fxz[0] = ai * pinv500; fxz[0] = fxz[0] + fy; fxz[1] = bi * ninv200; fxz[1] = fxz[1] + fy; }
            for (int j = 0; j < 2; j++) {
                const float f = fxz[j];
                if (f <= fThresh) fxz[j] = tail ? (f - c16_116) / 7.787f : (f - c16_116) * inv7787;
                else { float t = f * f; fxz[j] = t * f; }
            }
            const float x = fxz[0], z = fxz[1];
            float ro, go, bo;
            if (tail) {
                float t0 = C[0] * x, t1 = C[1] * y, t2 = C[2] * z; ro = (t0 + t1) + t2;
                t0 = C[3] * x; t1 = C[4] * y; t2 = C[5] * z; go = (t0 + t1) + t2;
                t0 = C[6] * x; t1 = C[7] * y; t2 = C[8] * z; bo = (t0 + t1) + t2;
            } else {
                float t2 = C[2] * z, t1 = C[1] * y + t2; ro = C[0] * x + t1;
                t2 = C[5] * z; t1 = C[4] * y + t2; go = C[3] * x + t1;
                t2 = C[8] * z; t1 = C[7] * y + t2; bo = C[6] * x + t1;
            }
            ro = ro < 1.f ? ro : 1.f; ro = ro > 0.f ? ro : 0.f;                  /* v_max(zero, v_min(ro, one)) */
            go = go < 1.f ? go : 1.f; go = go > 0.f ? go : 0.f;
            bo = bo < 1.f ? bo : 1.f; bo = bo > 0.f ? bo : 0.f;
            if (srgb) {
                ro = splineAt(ro * 1024.f, InvGammaSpline, 1024);
                go = splineAt(go * 1024.f, InvGammaSpline, 1024);
                bo = splineAt(bo * 1024.f, InvGammaSpline, 1024);
            }
            d[0] = ro; d[1] = go; d[2] = bo;
            if (dcn == 4) d[3] = 1.f;
        }
    }
}

/* ----------------------------------------------------------------------------------------------------------------- L*u*v*, CV_32F (and CV_8U from linear RGB)
 *   - forward   RGB2Luvfloat color_lab.cpp:2868-3037 (vector body :2925-3003, scalar tail :3006-3032)
 *   - inverse   Luv2RGBfloat :3057-3254 (vector body :3113-3205, scalar tail :3208-3247 -- note its 1/903.3f where the body has 1/903.296296f)
 *   - CV_8U from linear RGB: RGB2Luv_b's float branch :3405-3545 = bytes / 255 -> RGB2Luvfloat -> scale, round, saturate
 * each in the form of the vector body for the first 8 * (width / 8) pixels of a row and of the scalar tail for the rest. */
static void luvWhite(float* un, float* vn, int inverse)
{
    float d = (float)(D65[0] + D65[1] * 15.0 + D65[2] * 3.0);
    d = 1.f / fmax2(d, 1.1920928955078125e-7f);
    if (inverse) { *un = (52.f * d) * (float)D65[0]; *vn = (117.f * d) * (float)D65[1]; }
    else { *un = (d * 52.f) * (float)D65[0]; *vn = (d * 117.f) * (float)D65[1]; }
}

static void luvForwardRow(const float* s, int scn, float* d, int w, const float* C, float un, float vn, int srgb)
{
    const float tabScale = (float)(1024 * 2) / 3.f;
    for (int x = 0; x < w; x++, s += scn, d += 3) {
        float R = s[0], G = s[1], B = s[2], L, u, v;
        if (x >= (w & ~7)) {                                /* scalar tail */
            R = clip01(R); G = clip01(G); B = clip01(B);
            if (srgb) { R = splineAt(R * 1024.f, GammaSpline, 1024); G = splineAt(G * 1024.f, GammaSpline, 1024); B = splineAt(B * 1024.f, GammaSpline, 1024); }
            float t0 = R * C[0], t1 = G * C[1], t2 = B * C[2]; const float X = (t0 + t1) + t2;
            t0 = R * C[3]; t1 = G * C[4]; t2 = B * C[5]; const float Y = (t0 + t1) + t2;
            t0 = R * C[6]; t1 = G * C[7]; t2 = B * C[8]; const float Z = (t0 + t1) + t2;
            L = splineAt(Y * tabScale, LabCbrtSpline, 1024);
            L = 116.f * L; L = L - 16.f;
            float den = 15 * Y; den = X + den; t0 = 3 * Z; den = den + t0;
            const float dd = 52.f / (den > 1.1920928955078125e-7f ? den : 1.1920928955078125e-7f);      /* std::max(a, b): a < b ? b : a */
            t0 = X * dd; u = L * (t0 - un);
            t0 = (9 * 0.25f) * Y; t0 = t0 * dd; v = L * (t0 - vn);
        } else {                                            /* vector body */
            R = R > 0.f ? R : 0.f; R = R < 1.f ? R : 1.f;   /* v_min(v_max(R, zero), one) */
            G = G > 0.f ? G : 0.f; G = G < 1.f ? G : 1.f;
            B = B > 0.f ? B : 0.f; B = B < 1.f ? B : 1.f;
            if (srgb) { R = splineAt(R * 1024.f, GammaSpline, 1024); G = splineAt(G * 1024.f, GammaSpline, 1024); B = splineAt(B * 1024.f, GammaSpline, 1024); }
            float t2 = B * C[2], t1 = G * C[1] + t2; const float X = R * C[0] + t1;
            t2 = B * C[5]; t1 = G * C[4] + t2; const float Y = R * C[3] + t1;
            t2 = B * C[8]; t1 = G * C[7] + t2; const float Z = R * C[6] + t1;
            L = splineAt(Y * tabScale, LabCbrtSpline, 1024);
            L = L * 116.f; L = L + -16.f;
            float den = Z * 3.f + X; den = Y * 15.f + den;
            const float dd = 52.f / (den > 1.1920928955078125e-7f ? den : 1.1920928955078125e-7f);
            float t0 = X * dd + -un; u = L * t0;
            t0 = (9.F * 0.25F) * Y; t0 = t0 * dd + -vn; v = L * t0;
        }
        d[0] = L; d[1] = u; d[2] = v;
    }
}

static void luvForwardCoeffs(float* C, int swapBlue)
{
    for (int i = 0; i < 9; i++) C[i] = (float)sRGB2XYZ_D65[i];
    if (!swapBlue) for (int i = 0; i < 3; i++) { const float t = C[i * 3]; C[i * 3] = C[i * 3 + 2]; C[i * 3 + 2] = t; }      /* blueIdx == 0 */
}

void orc_cvtBGRtoLuv32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int scn, int swapBlue, int srgb)
{
    buildSplines();
    float C[9], un, vn;
    luvForwardCoeffs(C, swapBlue); luvWhite(&un, &vn, 0);
    for (int y = 0; y < h; y++)
        luvForwardRow((const float*)((const uint8_t*)src + (size_t)y * sstepBytes), scn, (float*)((uint8_t*)dst + (size_t)y * dstepBytes), w, C, un, vn, srgb);
}

/* CV_8U, linear RGB (sRGB sources take the grid interpolation above): per row, bytes * (1 / 255.f) -> the float conversion -> L * 2.55, u, v into 0..255 */
void orc_cvtLBGRtoLuv8u(const uint8_t* src, size_t sstep, uint8_t* dst, size_t dstep, int w, int h, int scn, int swapBlue)
{
    buildSplines();
    float C[9], un, vn;
    luvForwardCoeffs(C, swapBlue); luvWhite(&un, &vn, 0);
    const float f255inv = 1.f / 255.f, fL = 255.f / 100.f, uLow = -134.f, uRange = 354.f, vLow = -140.f, vRange = 262.f;
    const float fu = 255.f / uRange, fv = 255.f / vRange, su = (-uLow * 255.f) / uRange, sv = (-vLow * 255.f) / vRange;
    float* in = (float*)malloc((size_t)w * 3 * sizeof(float));
    float* out = (float*)malloc((size_t)w * 3 * sizeof(float));
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstep;
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) for (int c = 0; c < 3; c++) in[x * 3 + c] = (float)s[x * scn + c] * f255inv;
        luvForwardRow(in, 3, out, w, C, un, vn, 0);
        for (int x = 0; x < w; x++) {
            float t = out[x * 3] * fL; d[x * 3] = sat8((int)lrintf(t));
            t = out[x * 3 + 1] * fu; t = t + su; d[x * 3 + 1] = sat8((int)lrintf(t));
            t = out[x * 3 + 2] * fv; t = t + sv; d[x * 3 + 2] = sat8((int)lrintf(t));
        }
    }
    free(in); free(out);
}

void orc_cvtLuvtoBGR32f(const float* src, size_t sstepBytes, float* dst, size_t dstepBytes, int w, int h, int dcn, int swapBlue, int srgb)
{
    buildSplines();
    const int blueIdx = swapBlue ? 2 : 0;
    float C[9], un, vn;
    for (int i = 0; i < 3; i++) { C[i + (blueIdx ^ 2) * 3] = (float)XYZ2sRGB_D65[i]; C[i + 3] = (float)XYZ2sRGB_D65[i + 3]; C[i + blueIdx * 3] = (float)XYZ2sRGB_D65[i + 6]; }
    luvWhite(&un, &vn, 1);
    for (int yy = 0; yy < h; yy++) {
        const float* s = (const float*)((const uint8_t*)src + (size_t)yy * sstepBytes);
        float* d = (float*)((uint8_t*)dst + (size_t)yy * dstepBytes);
        for (int x = 0; x < w; x++, s += 3, d += dcn) {
            const float L = s[0], u = s[1], v = s[2];
            float R, G, B;
            if (x >= (w & ~7)) {                            /* scalar tail */
                float Y;
                if (L >= 8) { Y = (L + 16.f) * (1.f / 116.f); float t = Y * Y; Y = t * Y; }
                else Y = L * (1.0f / 903.3f);
                float up = L * un; up = 3.f * (u + up);
                float vp = L * vn; vp = 0.25f / (v + vp);
                if (vp > 0.25f) vp = 0.25f;
                if (vp < -0.25f) vp = -0.25f;
                float X = Y * 3.f; X = X * up; X = X * vp;
                float Z = (12.f * 13.f) * L; Z = Z - up; Z = Z * vp; Z = Z - 5.f; Z = Y * Z;
                float t0 = X * C[0], t1 = Y * C[1], t2 = Z * C[2]; R = (t0 + t1) + t2;
                t0 = X * C[3]; t1 = Y * C[4]; t2 = Z * C[5]; G = (t0 + t1) + t2;
                t0 = X * C[6]; t1 = Y * C[7]; t2 = Z * C[8]; B = (t0 + t1) + t2;
                R = clip01(R); G = clip01(G); B = clip01(B);
            } else {                                        /* vector body */
                float Ylo = (L + 16.f) * (1.f / 116.f); { float t = Ylo * Ylo; Ylo = t * Ylo; }
                const float Yhi = L * (1.0f / 903.296296f);
                const float Y = L >= 8.f ? Ylo : Yhi;
                float up = L * un + u; up = 3.f * up;
                float vp = L * vn + v; vp = 0.25f / vp;
                vp = 0.25f < vp ? 0.25f : vp;               /* v_min(v4inv, vp): SSE min(a, b) = a < b ? a : b */
                vp = -0.25f > vp ? -0.25f : vp;             /* v_max(-0.25, .) */
                float X = 3.f * up; X = X * vp;
                float Z = L * (12.f * 13.f) + -up; Z = Z * vp + -5.f;
                float t = X * C[0] + C[1]; t = Z * C[2] + t; R = t * Y;
                t = X * C[3] + C[4]; t = Z * C[5] + t; G = t * Y;
                t = X * C[6] + C[7]; t = Z * C[8] + t; B = t * Y;
                R = R > 0.f ? R : 0.f; R = R < 1.f ? R : 1.f;
                G = G > 0.f ? G : 0.f; G = G < 1.f ? G : 1.f;
                B = B > 0.f ? B : 0.f; B = B < 1.f ? B : 1.f;
            }
            if (srgb) { R = splineAt(R * 1024.f, InvGammaSpline, 1024); G = splineAt(G * 1024.f, InvGammaSpline, 1024); B = splineAt(B * 1024.f, InvGammaSpline, 1024); }
            d[0] = R; d[1] = G; d[2] = B;
            if (dcn == 4) d[3] = 1.f;
        }
    }
}

/* the tables themselves, for a direct comparison with the library's (tests): which = 0 sRGBGamma (256), 1 LabCbrt (3072), 2 sRGBInvGamma (4096),
 * 3 LabToYF (512) as uint16; 4 abToXZ (36864) as int32; 5 the 33^3 x 3 RGB -> Luv table as int16; 6 / 7 LuToUp / LvToVp (65536) as int32 */
int orc_labTable(int which, void* out)
{
    buildTables();
    switch (which) {
    case 0: memcpy(out, sRGBGammaTab_b, sizeof sRGBGammaTab_b); return 256;
    case 1: memcpy(out, LabCbrtTab_b, sizeof LabCbrtTab_b); return CBRT_TAB_B;
    case 2: memcpy(out, sRGBInvGammaTab_b, sizeof sRGBInvGammaTab_b); return INV_GAMMA_TAB;
    case 3: memcpy(out, LabToYF_b, sizeof LabToYF_b); return 512;
    case 4: memcpy(out, abToXZ_b, sizeof abToXZ_b); return ABXZ_N;
    case 5: buildLuvTables(); memcpy(out, RGB2LuvLUT, sizeof RGB2LuvLUT); return LUT_DIM * LUT_DIM * LUT_DIM * 3;      /* int16 */
    case 6: buildLuvTables(); memcpy(out, LuToUp_b, sizeof LuToUp_b); return 65536;                                     /* int32 */
    case 7: buildLuvTables(); memcpy(out, LvToVp_b, sizeof LvToVp_b); return 65536;
    case 8: buildLuvTables(); memcpy(out, RGB2LabLUT, sizeof RGB2LabLUT); return LUT_DIM * LUT_DIM * LUT_DIM * 3;      /* int16 */
    case 9: buildSplines(); memcpy(out, LabCbrtSpline, sizeof LabCbrtSpline); return 4096;                               /* float */
    case 10: buildSplines(); memcpy(out, InvGammaSpline, sizeof InvGammaSpline); return 4096;
    case 11: buildSplines(); memcpy(out, GammaSpline, sizeof GammaSpline); return 4096;
    }
    return -1;
}
