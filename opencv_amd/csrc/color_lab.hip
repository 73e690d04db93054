// color_lab.hip -- SURVEY.md §8 f1: CIE L*a*b* and L*u*v* <-> BGR / RGB for CV_8U images behind cv_hal_cvtBGRtoLab / cv_hal_cvtLabtoBGR
// (hal_replacement.hpp:535-565; callers hal::cvtBGRtoLab color_lab.cpp:4230, hal::cvtLabtoBGR :4327).  The reference's 8-bit paths are pure integer
// arithmetic over small tables -- RGB2Lab_b (color_lab.cpp:1573) and Lab2RGBinteger (:2399, taken by Lab2RGB_b since enableBitExactness) -- so the
// results here are bit-identical by construction once the tables are:
//   gamma      256 x u16   sRGBGammaTab_b  = round(2040 * gamma(i / 255))                 createLabTabs :1258-1264
//   cbrt      3072 x u16   LabCbrtTab_b    = round(2^15 * f(i / 2040)), f = Lab's cube-root / linear ramp   :1273-1279
//   yf         256 x u32   LabToYF_b       = (y, ify) of every 8-bit L                    :1281-1306
//   invGamma  4096 x u16   sRGBInvGammaTab_b = round(255 * gamma^-1(i / 4096))            :1265-1271
// built once on the host (the reference's softfloat arithmetic restated with IEEE float / double operations, its Turkowski cube root
// (softfloat.cpp:3897) restated as written) and kept in HBM per device; a workgroup copies what its kernel needs into LDS (6.5 KB forward, 9 KB
// inverse) and converts 16 rows x 256 pixels, a lane owning 4 consecutive pixels of a row (dword traffic, pix4.h).  The a/b -> X/Z table of the
// reference (initLUTforABXZ :1086, 147 KB) is two integer formulas, evaluated instead of looked up.
// HBM-bound: (scn + 3) B per pixel forward, (3 + dcn) B inverse.
// L*a*b* on CV_32F images: the reference's float paths in the form of their vector bodies (RGB2Lab_f :1895 -- for sRGB the same 33^3 grid interpolation as
// L*u*v*, for linear RGB a cubic spline for the cube root --, Lab2RGBfloat :2169); the last width % 8 pixels of a row in the form of its scalar tails (cubeRoot(),
// divisions): the CPU restatement of the same form (tests) equals the reference bit for bit on every case tested.
// L*u*v* (isLab == false), CV_8U: sRGB -> Luv by trilinear interpolation in the reference's 33^3 fixed-point table (RGB2Luvinterpolate :3276), Luv ->
// sRGB / linear RGB by Luv2RGBinteger (:3556) -- tables restated the same way.  L*u*v* on CV_32F images, and on CV_8U images in LINEAR RGB, follow the
// reference's float paths (RGB2Luvfloat :2868, Luv2RGBfloat :3057) the same way as CV_32F L*a*b*.  Both hooks now serve every (depth, isLab, srgb) case.
#include "rt.h"
#include "pix4.h"
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

using namespace mi355;

namespace {

enum { LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT, N_CBRT = 256 * 3 / 2 * (1 << GAMMA_SHIFT), INV_GAMMA_SHIFT = 12,
       N_INVG = 1 << INV_GAMMA_SHIFT, LBASE = 1 << 14 };

struct LabTabs {
    uint16_t gamma[256];
    uint16_t cbrt[N_CBRT];
    uint32_t yf[256];                  // y | ify << 16
    uint16_t invGamma[N_INVG];
};

// softfloat.cpp:3897 f32_cbrt: |x| = fr * 8^k with 0.125 <= fr < 1, fr -> P(fr) / Q(fr) (quartic rational, double), the top 23 fraction bits kept
float cubeRootTurkowski(float x)
{
    uint32_t v; std::memcpy(&v, &x, 4);
    if (!(v & 0x7fffffffu)) return 0.f;
    int ex = (int)((v >> 23) & 255) - 127, shx = ex % 3;
    if (shx >= 0) shx -= 3;
    ex = (ex - shx) / 3 - 1;
    uint64_t b = ((uint64_t)(shx + 1023) << 52) | ((uint64_t)(v & 0x7fffffu) << 29);
    double fr; std::memcpy(&fr, &b, 8);
    const double p = (((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr
                      + 13.43250139086239872172837314) * fr + 0.1636161226585754240958355063;
    const double q = (((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr
                      + 33.9905941350215598754191872) * fr + 1.0;
    fr = p / q;
    std::memcpy(&b, &fr, 8);
    const uint32_t o = (v & 0x80000000u) | ((uint32_t)(ex + 127) << 23) | (uint32_t)((b & 0xfffffffffffffull) >> 29);
    float y; std::memcpy(&y, &o, 4);
    return y;
}

// color_lab.cpp:1011-1040: the sRGB transfer function and its inverse in double on a float argument, rounded to float
float gammaFwd(float x)
{
    const double xd = x, xshift = 11.0 / 200.0;
    return (float)(xd <= 809.0 / 20000.0 ? xd / (323.0 / 25.0) : std::pow((xd + xshift) / (1.0 + xshift), 12.0 / 5.0));
}
float gammaInv(float x)
{
    const double xd = x, xshift = 11.0 / 200.0;
    return (float)(xd <= 7827.0 / 2500000.0 ? xd * (323.0 / 25.0) : std::pow(xd, 1.0 / (12.0 / 5.0)) * (1.0 + xshift) - xshift);
}

LabTabs g_host;
std::once_flag g_hostOnce;

void buildHost()
{
    LabTabs& t = g_host;
    const float f255 = 255.f, intScale = (float)(255 * (1 << GAMMA_SHIFT));
    for (int i = 0; i < 256; i++) t.gamma[i] = (uint16_t)lrintf(intScale * gammaFwd((float)i / f255));
    const float invScale = 1.f / (float)N_INVG;
    for (int i = 0; i < N_INVG; i++) t.invGamma[i] = (uint16_t)lrintf(f255 * gammaInv(invScale * (float)i));
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    const float cbScale = 1.f / (f255 * (float)(1 << GAMMA_SHIFT)), lshift2 = (float)(1 << LAB_SHIFT2);
    for (int i = 0; i < N_CBRT; i++) {
        const float x = cbScale * (float)i;
        t.cbrt[i] = (uint16_t)lrintf(lshift2 * (x < lthresh ? fmaf(x, lscale, lbias) : cubeRootTurkowski(x)));
    }
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) {                                                                      // the linear part of L -> Y
            y = (int)lrintf((float)(i * LBASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            const float s = 16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29);
            ify = (int)lrintf((float)LBASE * s);
        } else {
            const float a = (float)(i * 100 * LBASE) / (float)(255 * 116), b = (float)(16 * LBASE) / 116.f, fy = a + b;
            ify = (int)lrintf(fy);
            const float f2 = fy * fy, f3 = f2 * fy;
            y = (int)lrintf(f3 / (float)(LBASE * LBASE));
        }
        t.yf[i] = (uint32_t)y | ((uint32_t)ify << 16);
    }
}

enum { LAB_MAX_DEV = 64 };
LabTabs* g_dev[LAB_MAX_DEV];
std::mutex g_devMu;

const LabTabs* deviceTabs()
{
    std::call_once(g_hostOnce, buildHost);
    const int dev = activeDevice();
    if (dev < 0 || dev >= LAB_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_devMu);
    if (!g_dev[dev]) {
        LabTabs* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(LabTabs)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemcpy(d, &g_host, sizeof(LabTabs), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_dev[dev] = d;
    }
    return g_dev[dev];
}

// the D65 white point and the sRGB primaries as the reference holds them (color_lab.cpp:103-128, :941: these decimals ARE its raw doubles)
const double kD65[3] = {0.950456, 1.0, 1.088754};
const double kRgb2Xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
const double kXyz2Rgb[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};

// ---- L*u*v* tables (initLUTforLABLUVs16 color_lab.cpp:1126-1232, initLUTforLUV :1043-1084), built on first use:
//   lut   33^3 grid points x (L, u, v, 0) as 14-bit fixed point int16 -- RGB2Luvinterpolate's table, one 8-byte entry per grid point (the reference
//         stores every cell's eight corners; the upper neighbours it clamps never reach the last plane for 8-bit inputs: cell index <= 31)
//   up    LuToUp_b [L][u] = round(16 * 9 (u + L un)),   vp   LvToVp_b [L][v] = round(2^24 * clamp(1 / (4 (v + L vn)), +-1/4))
enum { LUV_DIM = 33, LUV_GRID = LUV_DIM * LUV_DIM * LUV_DIM };
struct LuvTabs {
    short lut[LUV_GRID * 4];
    int up[256 * 256];
    int vp[256 * 256];
    // CV_32F L*a*b*: the RGB -> Lab grid of the same builder (L, a, b, 0) and the two cubic splines (1024 knots x (f, b, c, d), splineBuild color_lab.cpp:20)
    short labGrid[LUV_GRID * 4];
    float cbrtSpline[1024 * 4];
    float invGammaSpline[1024 * 4];
    float gammaSpline[1024 * 4];
};
LuvTabs* g_luvHost;                    // 0.8 MB: allocated when first needed
std::once_flag g_luvHostOnce;

inline float maxSoft(float a, float b) { return a > b ? a : b; }

void buildLuvHost()
{
    std::call_once(g_hostOnce, buildHost);
    LuvTabs* t = new LuvTabs;
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f, f255 = 255.f, eps = 1.1920928955078125e-7f;
    const float uLow = -134.f, uRange = 220.f - -134.f, vLow = -140.f, vRange = 122.f - -140.f;
    float dd = (float)(kD65[0] + kD65[1] * 15.0 + kD65[2] * 3.0);
    dd = 1.f / maxSoft(dd, eps);
    const float un = dd * 52.f * (float)kD65[0], vn = dd * 117.f * (float)kD65[1];
    float C[9], S[9];                                               // columns reversed: the table's first axis is blue; S = the Lab rows / white point
    const double sw[3] = {1.0 / kD65[0], 1.0, 1.0 / kD65[2]};
    for (int i = 0; i < 3; i++) {
        C[i * 3 + 2] = (float)kRgb2Xyz[i * 3]; C[i * 3 + 1] = (float)kRgb2Xyz[i * 3 + 1]; C[i * 3] = (float)kRgb2Xyz[i * 3 + 2];
        S[i * 3] = (float)(kRgb2Xyz[i * 3 + 2] * sw[i]); S[i * 3 + 1] = (float)(kRgb2Xyz[i * 3 + 1] * sw[i]); S[i * 3 + 2] = (float)(kRgb2Xyz[i * 3] * sw[i]);
    }
    const float f9033 = (float)(29 * 29 * 29) / 27.f;
    const float lld = (float)(LUV_DIM - 1), lbase = (float)LBASE, f9of4 = 9.f / 4.f;
    float gam[LUV_DIM];
    for (int p = 0; p < LUV_DIM; p++) gam[p] = gammaFwd((float)p / lld);
    for (int r = 0; r < LUV_DIM; r++)
        for (int q = 0; q < LUV_DIM; q++)
            for (int p = 0; p < LUV_DIM; p++) {
                const float R = gam[p], G = gam[q], B = gam[r];
                float t0 = R * C[0], t1 = G * C[1], t2 = B * C[2];
                const float X = (t0 + t1) + t2;
                t0 = R * C[3]; t1 = G * C[4]; t2 = B * C[5];
                const float Y = (t0 + t1) + t2;
                t0 = R * C[6]; t1 = G * C[7]; t2 = B * C[8];
                const float Z = (t0 + t1) + t2;
                float L = Y < lthresh ? fmaf(Y, lscale, lbias) : cubeRootTurkowski(Y);
                L = L * 116.f - 16.f;
                const float y15 = 15.f * Y, z3 = 3.f * Z;
                const float d = 52.f / maxSoft((X + y15) + z3, eps);
                const float xd = X * d, u = L * (xd - un);
                const float yd = (f9of4 * Y) * d, v = L * (yd - vn);
                short* e = t->lut + 4 * ((r * LUV_DIM + q) * LUV_DIM + p);
                e[0] = (short)lrintf((lbase * L) / 100.f);
                e[1] = (short)lrintf((lbase * (u - uLow)) / uRange);
                e[2] = (short)lrintf((lbase * (v - vLow)) / vRange);
                e[3] = 0;
                // the Lab grid point (initLUTforLABLUVs16 :1171-1187)
                float s0 = R * S[0], s1 = G * S[1], s2 = B * S[2];
                const float Xl = (s0 + s1) + s2;
                s0 = R * S[3]; s1 = G * S[4]; s2 = B * S[5];
                const float Yl = (s0 + s1) + s2;
                s0 = R * S[6]; s1 = G * S[7]; s2 = B * S[8];
                const float Zl = (s0 + s1) + s2;
                const float FX = Xl > lthresh ? cubeRootTurkowski(Xl) : fmaf(Xl, lscale, lbias);
                const float FY = Yl > lthresh ? cubeRootTurkowski(Yl) : fmaf(Yl, lscale, lbias);
                const float FZ = Zl > lthresh ? cubeRootTurkowski(Zl) : fmaf(Zl, lscale, lbias);
                const float Ll = Yl > lthresh ? (116.f * FY - 16.f) : (f9033 * Yl);
                const float al = 500.f * (FX - FY), bl = 200.f * (FY - FZ);
                short* g = t->labGrid + 4 * ((r * LUV_DIM + q) * LUV_DIM + p);
                g[0] = (short)lrintf((lbase * Ll) / 100.f);
                g[1] = (short)lrintf((lbase * (al + 128.f)) / 256.f);
                g[2] = (short)lrintf((lbase * (bl + 128.f)) / 256.f);
                g[3] = 0;
            }
    {
        // cubic splines over 1024 intervals (splineBuild color_lab.cpp:20-47, float arithmetic): Lab's f() on [0, 1.5] and the inverse sRGB transfer on [0, 1]
        std::vector<float> f(1025), ig(1025), gf(1025);
        const float cbScale = 1.f / ((float)(1024 * 2) / 3.f), gScale = 1.f / 1024.f;
        for (int i = 0; i <= 1024; i++) {
            const float x = cbScale * (float)i;
            f[i] = x < lthresh ? fmaf(x, lscale, lbias) : cubeRootTurkowski(x);
            ig[i] = gammaInv(gScale * (float)i);
            gf[i] = gammaFwd(gScale * (float)i);
        }
        auto build = [](const std::vector<float>& fv, float* tab) {
            const int n = 1024;
            float cn = 0.f;
            tab[0] = tab[1] = 0.f;
            for (int i = 1; i < n; i++) {
                const float tt = ((fv[i + 1] - fv[i] * 2.f) + fv[i - 1]) * 3.f;
                const float l = 1.f / (4.f - tab[(i - 1) * 4]);
                tab[i * 4] = l; tab[i * 4 + 1] = (tt - tab[(i - 1) * 4 + 1]) * l;
            }
            for (int j = 0; j < n; j++) {
                const int i = n - j - 1;
                const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
                const float b = (fv[i + 1] - fv[i]) - (cn + c * 2.f) / 3.f;
                const float d = (cn - c) / 3.f;
                tab[i * 4] = fv[i]; tab[i * 4 + 1] = b; tab[i * 4 + 2] = c; tab[i * 4 + 3] = d;
                cn = c;
            }
        };
        build(f, t->cbrtSpline);
        build(ig, t->invGammaSpline);
        build(gf, t->gammaSpline);
    }
    for (int LL = 0; LL < 256; LL++) {
        const float L = (float)(LL * 100) / f255;
        for (int uu = 0; uu < 256; uu++) {
            const float u = ((float)uu * uRange) / f255 + uLow;
            t->up[LL * 256 + uu] = (int)lrintf((9.f * (u + L * un)) * (float)(LBASE / 1024));
        }
        for (int vv = 0; vv < 256; vv++) {
            const float v = ((float)vv * vRange) / f255 + vLow;
            float vp = 0.25f / (v + L * vn);
            if (vp > 0.25f) vp = 0.25f;
            if (vp < -0.25f) vp = -0.25f;
            t->vp[LL * 256 + vv] = (int)lrintf(vp * (float)(LBASE * 1024));
        }
    }
    g_luvHost = t;
}

LuvTabs* g_luvDev[LAB_MAX_DEV];

const LuvTabs* deviceLuvTabs()
{
    std::call_once(g_luvHostOnce, buildLuvHost);
    const int dev = activeDevice();
    if (dev < 0 || dev >= LAB_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_devMu);
    if (!g_luvDev[dev]) {
        LuvTabs* d = nullptr;
        if (hipMalloc((void**)&d, sizeof(LuvTabs)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemcpy(d, g_luvHost, sizeof(LuvTabs), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_luvDev[dev] = d;
    }
    return g_luvDev[dev];
}

struct Coef9 { int c[9]; };

constexpr int ROWS_PER_BLOCK = 16;                  // 4 rows per wave: a 4K frame is 8160 waves, 8 per SIMD

__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// ---- forward: 4 pixels per lane, ROWS_PER_BLOCK rows per workgroup, tables in LDS
template <int SCN, bool SRGB>
__global__ __launch_bounds__(256) void k_bgr2lab_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int aligned,
                                                     const LabTabs* __restrict__ tabs, Coef9 k)
{
    __shared__ uint16_t cb[N_CBRT];
    __shared__ uint16_t gm[256];
    for (int i = threadIdx.x; i < N_CBRT / 2; i += 256) ((unsigned*)cb)[i] = ((const unsigned*)tabs->cbrt)[i];
    if (SRGB && threadIdx.x < 128) ((unsigned*)gm)[threadIdx.x] = ((const unsigned*)tabs->gamma)[threadIdx.x];
    __syncthreads();
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (x4 >= W) return;
    const int n = min(4, W - x4);
    const bool fast = n == 4 && aligned;
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const int yEnd = min(H, (int)(blockIdx.y + 1) * ROWS_PER_BLOCK);
    auto loadPx = [&](pix4::Px<SCN>& in, int y) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x4 * SCN;
        if (fast) {
#pragma unroll
            for (int i = 0; i < SCN; i++) in.w[i] = ((const unsigned*)s)[i];
        } else {
            in.clear();
#pragma unroll
            for (int i = 0; i < 4 * SCN; i++) if (i < n * SCN) in.put(i, s[i]);
        }
    };
    auto convert = [&](const pix4::Px<SCN>& in, int y) {
        pix4::Px<3> out;
        out.clear();
#pragma unroll
        for (int p = 0; p < 4; p++) {
            int R = in.get(p * SCN), G = in.get(p * SCN + 1), B = in.get(p * SCN + 2);               // channel order is folded into the coefficients
            if (SRGB) { R = gm[R]; G = gm[G]; B = gm[B]; }
            else { R <<= GAMMA_SHIFT; G <<= GAMMA_SHIFT; B <<= GAMMA_SHIFT; }
            // every factor fits 24 bits (R, G, B <= 2040, coefficients < 2^13, table values < 2^16): v_mad_u32_u24 / v_mad_i32_i24, not the quarter-rate
            // 32-bit multiply -- the products and sums are the reference's ints
            const int fX = cb[descale(__mul24(R, k.c[0]) + __mul24(G, k.c[1]) + __mul24(B, k.c[2]), LAB_SHIFT)];
            const int fY = cb[descale(__mul24(R, k.c[3]) + __mul24(G, k.c[4]) + __mul24(B, k.c[5]), LAB_SHIFT)];
            const int fZ = cb[descale(__mul24(R, k.c[6]) + __mul24(G, k.c[7]) + __mul24(B, k.c[8]), LAB_SHIFT)];
            out.put(p * 3, sat8(descale(__mul24(Lscale, fY) + Lshift, LAB_SHIFT2)));
            out.put(p * 3 + 1, sat8(descale(__mul24(500, fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)));
            out.put(p * 3 + 2, sat8(descale(__mul24(200, fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)));
        }
        uchar* d = dst + (size_t)y * dstep + (size_t)x4 * 3;
        if (fast) {
#pragma unroll
            for (int i = 0; i < 3; i++) ((unsigned*)d)[i] = out.w[i];
        } else {
#pragma unroll
            for (int i = 0; i < 12; i++) if (i < n * 3) d[i] = (uchar)out.get(i);
        }
    };
    // two rows per step, both loads issued before the first is consumed (a workgroup's waves are few: the load latency would otherwise sit in the loop)
    for (int y = blockIdx.y * ROWS_PER_BLOCK + (threadIdx.x >> 6); y < yEnd; y += 8) {
        const bool two = y + 4 < yEnd;
        pix4::Px<SCN> a, b;
        loadPx(a, y);
        if (two) loadPx(b, y + 4);
        convert(a, y);
        if (two) convert(b, y + 4);
    }
}

// initLUTforABXZ (color_lab.cpp:1086-1110) as arithmetic: the linear ramp below 6/29 (C division truncates toward zero, i may be negative), the cube above
__device__ __forceinline__ int abToXZ(int i)
{
    if (i <= 3390) return __mul24(i, 108) / 841 - LBASE * 16 / 116 * 108 / 841;
    return __mul24(__mul24(i, i) >> 14, i) >> 14;                     // |i| < 2^15, i^2 >> 14 < 2^16
}

// ---- inverse
template <int DCN, bool SRGB>
__global__ __launch_bounds__(256) void k_lab2bgr_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int aligned,
                                                     const LabTabs* __restrict__ tabs, Coef9 k)
{
    __shared__ uint32_t yf[256];
    __shared__ uint16_t ig[SRGB ? N_INVG : 2];
    yf[threadIdx.x] = tabs->yf[threadIdx.x];
    if (SRGB) for (int i = threadIdx.x; i < N_INVG / 2; i += 256) ((unsigned*)ig)[i] = ((const unsigned*)tabs->invGamma)[i];
    __syncthreads();
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (x4 >= W) return;
    const int n = min(4, W - x4);
    const bool fast = n == 4 && aligned;
    constexpr int shift = LAB_SHIFT + (14 - INV_GAMMA_SHIFT);
    const int yEnd = min(H, (int)(blockIdx.y + 1) * ROWS_PER_BLOCK);
    auto loadPx = [&](pix4::Px<3>& in, int y) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x4 * 3;
        if (fast) {
#pragma unroll
            for (int i = 0; i < 3; i++) in.w[i] = ((const unsigned*)s)[i];
        } else {
            in.clear();
#pragma unroll
            for (int i = 0; i < 12; i++) if (i < n * 3) in.put(i, s[i]);
        }
    };
    auto convert = [&](const pix4::Px<3>& in, int y) {
        pix4::Px<DCN> out;
        out.clear();
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int LL = in.get(p * 3), aa = in.get(p * 3 + 1), bb = in.get(p * 3 + 2);
            const unsigned t = yf[LL];
            const int yv = (int)(t & 0xffffu), ify = (int)(t >> 16);
            const int adiv = ((__mul24(5 * aa, 53687) + (1 << 7)) >> 13) - 128 * LBASE / 500;
            const int bdiv = ((__mul24(bb, 41943) + (1 << 4)) >> 9) - 128 * LBASE / 200 + 1;
            const int xv = abToXZ(ify + adiv), zv = abToXZ(ify - bdiv);
            // |coefficient| < 2^15, -1335 <= x, z <= 88231, y <= 2^14: 24-bit factors, products below 2^31 (the reference's ints)
            int ro = descale(__mul24(k.c[0], xv) + __mul24(k.c[1], yv) + __mul24(k.c[2], zv), shift);
            int go = descale(__mul24(k.c[3], xv) + __mul24(k.c[4], yv) + __mul24(k.c[5], zv), shift);
            int bo = descale(__mul24(k.c[6], xv) + __mul24(k.c[7], yv) + __mul24(k.c[8], zv), shift);
            ro = max(0, min(N_INVG - 1, ro)); go = max(0, min(N_INVG - 1, go)); bo = max(0, min(N_INVG - 1, bo));
            if (SRGB) { ro = ig[ro]; go = ig[go]; bo = ig[bo]; }
            else { ro = ((ro << 8) - ro) >> INV_GAMMA_SHIFT; go = ((go << 8) - go) >> INV_GAMMA_SHIFT; bo = ((bo << 8) - bo) >> INV_GAMMA_SHIFT; }
            out.put(p * DCN, sat8(bo)); out.put(p * DCN + 1, sat8(go)); out.put(p * DCN + 2, sat8(ro));
            if (DCN == 4) out.put(p * 4 + 3, 255);
        }
        uchar* d = dst + (size_t)y * dstep + (size_t)x4 * DCN;
        if (fast) {
#pragma unroll
            for (int i = 0; i < DCN; i++) ((unsigned*)d)[i] = out.w[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4 * DCN; i++) if (i < n * DCN) d[i] = (uchar)out.get(i);
        }
    };
    for (int y = blockIdx.y * ROWS_PER_BLOCK + (threadIdx.x >> 6); y < yEnd; y += 8) {        // two rows per step, loads first (see the forward kernel)
        const bool two = y + 4 < yEnd;
        pix4::Px<3> a, b;
        loadPx(a, y);
        if (two) loadPx(b, y + 4);
        convert(a, y);
        if (two) convert(b, y + 4);
    }
}

// ---- L*u*v* forward (sRGB): trilinear interpolation in the 33^3 table (trilinearInterpolate color_lab.cpp:1352-1392).  A channel value c selects
// cell c >> 3 and the weight (2 c) & 15 of 16; the two corners along the first (blue) axis are adjacent entries: one 16-byte load, four per pixel.
template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2luv_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int aligned,
                                                     const LuvTabs* __restrict__ tabs, int bIdx)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= W || y >= H) return;
    const int n = min(4, W - x4);
    const bool fast = n == 4 && aligned;
    const uchar* s = src + (size_t)y * sstep + (size_t)x4 * SCN;
    uchar* d = dst + (size_t)y * dstep + (size_t)x4 * 3;
    pix4::Px<SCN> in; pix4::Px<3> out;
    out.clear();
    if (fast) {
#pragma unroll
        for (int i = 0; i < SCN; i++) in.w[i] = ((const unsigned*)s)[i];
    } else {
        in.clear();
#pragma unroll
        for (int i = 0; i < 4 * SCN; i++) if (i < n * SCN) in.put(i, s[i]);
    }
    typedef short s16x8 __attribute__((ext_vector_type(8), aligned(8)));
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int c0 = in.get(p * SCN), c1 = in.get(p * SCN + 1), c2 = in.get(p * SCN + 2);
        const int cb = bIdx ? c2 : c0, cg = c1, cr = bIdx ? c0 : c2;                     // table axes: blue, green, red
        const int tx = cb >> 3, ty = cg >> 3, tz = cr >> 3, fx = (2 * cb) & 15, fy = (2 * cg) & 15, fz = (2 * cr) & 15;
        const short* cell = tabs->lut + 4 * ((tz * LUV_DIM + ty) * LUV_DIM + tx);
        int aL = 0, aU = 0, aV = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {                                                     // k = 2 dq + dr (green, red); both blue corners per load
            const int dq = k >> 1, dr = k & 1;
            const s16x8 e = *reinterpret_cast<const s16x8*>(cell + 4 * ((dr * LUV_DIM + dq) * LUV_DIM));
            const int wyz = __mul24(dq ? fy : 16 - fy, dr ? fz : 16 - fz);
            const int w0 = __mul24(16 - fx, wyz), w1 = __mul24(fx, wyz);
            aL += __mul24(e[0], w0) + __mul24(e[4], w1);
            aU += __mul24(e[1], w0) + __mul24(e[5], w1);
            aV += __mul24(e[2], w0) + __mul24(e[6], w1);
        }
        // CV_DESCALE(., 12), then / (LAB_BASE / 256) -- the sums are not negative
        out.put(p * 3, sat8(descale(aL, 12) >> 6)); out.put(p * 3 + 1, sat8(descale(aU, 12) >> 6)); out.put(p * 3 + 2, sat8(descale(aV, 12) >> 6));
    }
    if (fast) {
#pragma unroll
        for (int i = 0; i < 3; i++) ((unsigned*)d)[i] = out.w[i];
    } else {
#pragma unroll
        for (int i = 0; i < 12; i++) if (i < n * 3) d[i] = (uchar)out.get(i);
    }
}

// ---- L*u*v* inverse: Luv2RGBinteger::process (color_lab.cpp:3587-3646) as written (64-bit intermediates, C division)
template <int DCN, bool SRGB>
__global__ __launch_bounds__(256) void k_luv2bgr_u8(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int aligned,
                                                     const LabTabs* __restrict__ tabs, const LuvTabs* __restrict__ luv, Coef9 k)
{
    __shared__ uint32_t yf[256];
    __shared__ uint16_t ig[SRGB ? N_INVG : 2];
    yf[threadIdx.x] = tabs->yf[threadIdx.x];
    if (SRGB) for (int i = threadIdx.x; i < N_INVG / 2; i += 256) ((unsigned*)ig)[i] = ((const unsigned*)tabs->invGamma)[i];
    __syncthreads();
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (x4 >= W) return;
    const int n = min(4, W - x4);
    const bool fast = n == 4 && aligned;
    constexpr int shift = LAB_SHIFT + (14 - INV_GAMMA_SHIFT);
    const int yEnd = min(H, (int)(blockIdx.y + 1) * ROWS_PER_BLOCK);
    for (int y = blockIdx.y * ROWS_PER_BLOCK + (threadIdx.x >> 6); y < yEnd; y += 4) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x4 * 3;
        uchar* d = dst + (size_t)y * dstep + (size_t)x4 * DCN;
        pix4::Px<3> in; pix4::Px<DCN> out;
        out.clear();
        if (fast) {
#pragma unroll
            for (int i = 0; i < 3; i++) in.w[i] = ((const unsigned*)s)[i];
        } else {
            in.clear();
#pragma unroll
            for (int i = 0; i < 12; i++) if (i < n * 3) in.put(i, s[i]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int LL = in.get(p * 3), uu = in.get(p * 3 + 1), vv = in.get(p * 3 + 2);
            const int yv = (int)(yf[LL] & 0xffffu);
            const int up = luv->up[LL * 256 + uu], vp = luv->vp[LL * 256 + vv];
            const long long xv = (long long)up * (long long)vp;
            int xi = (int)(xv / LBASE);
            xi = (int)((long long)yv * xi / LBASE);
            const long long vpl = (12 * 13 * 100 * (LBASE / 1024)) * (long long)(vp * LL);
            long long zp = vpl - xv * (255 / 3);
            zp /= LBASE;
            const long long zq = zp - (long long)(5 * 255 * LBASE);
            const int zm = (int)(yv * zq / LBASE);
            int zi = zm / 256 + zm / 65536;
            xi = max(0, min(2 * LBASE, xi)); zi = max(0, min(2 * LBASE, zi));
            int ro = descale(k.c[0] * xi + k.c[1] * yv + k.c[2] * zi, shift);
            int go = descale(k.c[3] * xi + k.c[4] * yv + k.c[5] * zi, shift);
            int bo = descale(k.c[6] * xi + k.c[7] * yv + k.c[8] * zi, shift);
            ro = max(0, min(N_INVG - 1, ro)); go = max(0, min(N_INVG - 1, go)); bo = max(0, min(N_INVG - 1, bo));
            if (SRGB) { ro = ig[ro]; go = ig[go]; bo = ig[bo]; }
            else { ro = ((ro << 8) - ro) >> INV_GAMMA_SHIFT; go = ((go << 8) - go) >> INV_GAMMA_SHIFT; bo = ((bo << 8) - bo) >> INV_GAMMA_SHIFT; }
            out.put(p * DCN, sat8(bo)); out.put(p * DCN + 1, sat8(go)); out.put(p * DCN + 2, sat8(ro));
            if (DCN == 4) out.put(p * 4 + 3, 255);
        }
        if (fast) {
#pragma unroll
            for (int i = 0; i < DCN; i++) ((unsigned*)d)[i] = out.w[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4 * DCN; i++) if (i < n * DCN) d[i] = (uchar)out.get(i);
        }
    }
}

// ---------------------------------------------------------------------------------- CV_32F L*a*b* (one pixel per lane)
// splineInterpolate (color_lab.cpp:50-58): the knot below x, then Horner on its four coefficients (one 16-byte load)
__device__ __forceinline__ float splineAt(float x, const float* __restrict__ tab)
{
    int ix = (int)x;
    ix = max(0, min(1023, ix));
    x -= (float)ix;
    const float4 t = *reinterpret_cast<const float4*>(tab + 4 * ix);
    float r = t.w * x + t.z;
    r = r * x + t.y;
    return r * x + t.x;
}

struct CoefF9 { float c[9]; };

// cv::cubeRoot (core/src/mathfuncs.cpp:104-140), which the scalar row tails of RGB2Lab_f call: Turkowski's rational on the float mantissa in double,
// ROUNDED to float (the softfloat version used for the tables truncates)
__device__ __forceinline__ float cubeRootRounded(float value)
{
    const uint32_t vi = __float_as_uint(value), ix = vi & 0x7fffffffu, sgn = vi & 0x80000000u;
    int ex = (int)(ix >> 23) - 127, shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    const double fr = __uint_as_float((ix & ((1u << 23) - 1)) | ((uint32_t)(shx + 127) << 23));
    const double num = (((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr
                        + 13.43250139086239872172837314) * fr + 0.1636161226585754240958355063;
    const double den = (((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr
                        + 33.9905941350215598754191872) * fr + 1.0;
    const float q = (float)(num / den);
    const uint32_t qi = (uint32_t)((int)__float_as_uint(q) + (ex << 23) + (int)sgn) & ((vi << 1) != 0 ? 0xffffffffu : 0u);
    return __uint_as_float(qi);
}

// RGB2Lab_f, sRGB (useInterpolation, color_lab.cpp:1955-2045): clip, 14-bit fixed point, trilinear interpolation in the RGB -> Lab grid, back to float
template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2lab_f32_grid(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                           const LuvTabs* __restrict__ tabs, int bIdx)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * SCN;
    float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * 3;
    auto clip = [](float v) { return v < 0.f ? 0.f : v <= 1.f ? v : 1.f; };
    const int cx = (int)__builtin_rintf(clip(s[bIdx]) * 16384.f), cy = (int)__builtin_rintf(clip(s[1]) * 16384.f), cz = (int)__builtin_rintf(clip(s[bIdx ^ 2]) * 16384.f);
    const int tx = cx >> 9, ty = cy >> 9, tz = cz >> 9, fx = (cx >> 5) & 15, fy = (cy >> 5) & 15, fz = (cz >> 5) & 15;
    int aL = 0, aA = 0, aB = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int dp = i >> 2, dq = (i >> 1) & 1, dr = i & 1;
        const int pp = min(tx + dp, LUV_DIM - 1), qq = min(ty + dq, LUV_DIM - 1), rr = min(tz + dr, LUV_DIM - 1);       // the last grid plane repeats (fill_one :1112)
        const short4 e = *reinterpret_cast<const short4*>(tabs->labGrid + 4 * ((rr * LUV_DIM + qq) * LUV_DIM + pp));
        const int w = __mul24(__mul24(dp ? fx : 16 - fx, dq ? fy : 16 - fy), dr ? fz : 16 - fz);
        aL += __mul24(e.x, w); aA += __mul24(e.y, w); aB += __mul24(e.z, w);
    }
    const int iL = descale(aL, 12), ia = descale(aA, 12), ib = descale(aB, 12);
    d[0] = (float)iL * (100.0f / 16384.f);
    float t = (float)ia * (256.0f / 16384.f); d[1] = t + -128.f;
    t = (float)ib * (256.0f / 16384.f); d[2] = t + -128.f;
}

// RGB2Lab_f, linear RGB (the float branch, vector body :2062-2130): XYZ, f() through the spline, L / a / b
template <int SCN>
__global__ __launch_bounds__(256) void k_bgr2lab_f32_lin(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                          const LuvTabs* __restrict__ tabs, CoefF9 k)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * SCN;
    float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * 3;
    auto clip = [](float v) { return v < 0.f ? 0.f : v <= 1.f ? v : 1.f; };
    const float R = clip(s[0]), G = clip(s[1]), B = clip(s[2]);
    if (x >= (W & ~7)) {
        // the last W % 8 pixels of a row are the reference's scalar tail (:2132-2157): sums left to right, cubeRoot() instead of the spline
        const float a16 = 16.f / 116.f;
        float t0 = R * k.c[0], t1 = G * k.c[1], t2 = B * k.c[2]; const float X = (t0 + t1) + t2;
        t0 = R * k.c[3]; t1 = G * k.c[4]; t2 = B * k.c[5]; const float Y = (t0 + t1) + t2;
        t0 = R * k.c[6]; t1 = G * k.c[7]; t2 = B * k.c[8]; const float Z = (t0 + t1) + t2;
        float FX, FY, FZ, L;
        if (X > 0.008856f) FX = cubeRootRounded(X); else { FX = 7.787f * X; FX = FX + a16; }
        if (Y > 0.008856f) FY = cubeRootRounded(Y); else { FY = 7.787f * Y; FY = FY + a16; }
        if (Z > 0.008856f) FZ = cubeRootRounded(Z); else { FZ = 7.787f * Z; FZ = FZ + a16; }
        if (Y > 0.008856f) { L = 116.f * FY; L = L - 16.f; } else L = 903.3f * Y;
        d[0] = L; d[1] = 500.f * (FX - FY); d[2] = 200.f * (FY - FZ);
        return;
    }
    float t2 = B * k.c[2], t1 = G * k.c[1] + t2; const float X = R * k.c[0] + t1;
    t2 = B * k.c[5]; t1 = G * k.c[4] + t2; const float Y = R * k.c[3] + t1;
    t2 = B * k.c[8]; t1 = G * k.c[7] + t2; const float Z = R * k.c[6] + t1;
    const float tabScale = (float)(1024 * 2) / 3.f;
    const float FX = splineAt(X * tabScale, tabs->cbrtSpline), FY = splineAt(Y * tabScale, tabs->cbrtSpline), FZ = splineAt(Z * tabScale, tabs->cbrtSpline);
    float L;
    if (Y > 0.008856f) { L = 116.f * FY; L = L + -16.f; } else L = 903.3f * Y;
    d[0] = L; d[1] = 500.f * (FX - FY); d[2] = 200.f * (FY - FZ);
}

// Lab2RGBfloat (vector body :2216-2325): L -> Y, a / b -> X / Z (products by reciprocal constants), XYZ -> RGB, clip, inverse sRGB transfer through the spline
template <int DCN, bool SRGB>
__global__ __launch_bounds__(256) void k_lab2bgr_f32(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                      const LuvTabs* __restrict__ tabs, CoefF9 k)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || yy >= H) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)yy * sstep) + (size_t)x * 3;
    float* d = reinterpret_cast<float*>(dst + (size_t)yy * dstep) + (size_t)x * DCN;
    const float li = s[0], ai = s[1], bi = s[2];
    const float fThresh = 6.f / 29.f, c16_116 = 16.0f / 116.0f;
    const float inv903 = 1.f / 903.3f, inv116 = 1.f / 116.0f, pinv500 = 1.f / 500.f, ninv200 = -1.f / 200.f, inv7787 = 1.f / 7.787f;
    const bool tail = x >= (W & ~7);                  // the reference's scalar row tail (:2343-2385) divides where its vector body multiplies by a reciprocal
    float y, fy;
    if (li <= 8.f) { y = tail ? li / 903.3f : li * inv903; fy = 7.787f * y; fy = fy + c16_116; }
    else { fy = tail ? (li + 16.0f) / 116.0f : (li + 16.0f) * inv116; y = fy * fy; y = y * fy; }
    float fxz[2];
    if (tail) { fxz[0] = ai / 500.0f; fxz[0] = fxz[0] + fy; fxz[1] = bi / 200.0f; fxz[1] = fy - fxz[1]; }
    else { fxz[0] = ai * pinv500; fxz[0] = fxz[0] + fy; fxz[1] = bi * ninv200; fxz[1] = fxz[1] + fy; }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float f = fxz[j];
        if (f <= fThresh) fxz[j] = tail ? (f - c16_116) / 7.787f : (f - c16_116) * inv7787;
        else { const float t = f * f; fxz[j] = t * f; }
    }
    const float xv = fxz[0], zv = fxz[1];
    float ro, go, bo;
    if (tail) {
        float t0 = k.c[0] * xv, t1 = k.c[1] * y, t2 = k.c[2] * zv; ro = (t0 + t1) + t2;
        t0 = k.c[3] * xv; t1 = k.c[4] * y; t2 = k.c[5] * zv; go = (t0 + t1) + t2;
        t0 = k.c[6] * xv; t1 = k.c[7] * y; t2 = k.c[8] * zv; bo = (t0 + t1) + t2;
    } else {
        float t2 = k.c[2] * zv, t1 = k.c[1] * y + t2; ro = k.c[0] * xv + t1;
        t2 = k.c[5] * zv; t1 = k.c[4] * y + t2; go = k.c[3] * xv + t1;
        t2 = k.c[8] * zv; t1 = k.c[7] * y + t2; bo = k.c[6] * xv + t1;
    }
    ro = ro < 1.f ? ro : 1.f; ro = ro > 0.f ? ro : 0.f;
    go = go < 1.f ? go : 1.f; go = go > 0.f ? go : 0.f;
    bo = bo < 1.f ? bo : 1.f; bo = bo > 0.f ? bo : 0.f;
    if (SRGB) {
        ro = splineAt(ro * 1024.f, tabs->invGammaSpline);
        go = splineAt(go * 1024.f, tabs->invGammaSpline);
        bo = splineAt(bo * 1024.f, tabs->invGammaSpline);
    }
    d[0] = ro; d[1] = go; d[2] = bo;
    if (DCN == 4) d[3] = 1.f;
}

// ---------------------------------------------------------------------------------- CV_32F L*u*v*, and CV_8U L*u*v* from linear RGB
struct LuvF { float c[9]; float un, vn; };

// RGB2Luvfloat (color_lab.cpp:2868-3037): vector body for the first 8 * (W / 8) pixels of a row, scalar tail for the rest.  U8: RGB2Luv_b's float
// branch (:3405-3545) -- bytes / 255 in, (L * 2.55, u, v scaled into 0..255) rounded and saturated out
template <int SCN, bool SRGB, bool U8>
__global__ __launch_bounds__(256) void k_bgr2luv_f32(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                      const LuvTabs* __restrict__ tabs, LuvF k)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float R, G, B;
    if (U8) {
        const uchar* s = src + (size_t)y * sstep + (size_t)x * SCN;
        const float f255inv = 1.f / 255.f;
        R = (float)s[0] * f255inv; G = (float)s[1] * f255inv; B = (float)s[2] * f255inv;
    } else {
        const float* s = reinterpret_cast<const float*>(src + (size_t)y * sstep) + (size_t)x * SCN;
        R = s[0]; G = s[1]; B = s[2];
    }
    const float tabScale = (float)(1024 * 2) / 3.f, eps = 1.1920928955078125e-7f;
    float L, u, v;
    if (x >= (W & ~7)) {
        R = R < 0.f ? 0.f : R <= 1.f ? R : 1.f; G = G < 0.f ? 0.f : G <= 1.f ? G : 1.f; B = B < 0.f ? 0.f : B <= 1.f ? B : 1.f;
        if (SRGB) { R = splineAt(R * 1024.f, tabs->gammaSpline); G = splineAt(G * 1024.f, tabs->gammaSpline); B = splineAt(B * 1024.f, tabs->gammaSpline); }
        float t0 = R * k.c[0], t1 = G * k.c[1], t2 = B * k.c[2]; const float X = (t0 + t1) + t2;
        t0 = R * k.c[3]; t1 = G * k.c[4]; t2 = B * k.c[5]; const float Y = (t0 + t1) + t2;
        t0 = R * k.c[6]; t1 = G * k.c[7]; t2 = B * k.c[8]; const float Z = (t0 + t1) + t2;
        L = splineAt(Y * tabScale, tabs->cbrtSpline);
        L = 116.f * L; L = L - 16.f;
        float den = 15 * Y; den = X + den; t0 = 3 * Z; den = den + t0;
        const float dd = 52.f / (den > eps ? den : eps);
        t0 = X * dd; u = L * (t0 - k.un);
        t0 = (9 * 0.25f) * Y; t0 = t0 * dd; v = L * (t0 - k.vn);
    } else {
        R = R > 0.f ? R : 0.f; R = R < 1.f ? R : 1.f;
        G = G > 0.f ? G : 0.f; G = G < 1.f ? G : 1.f;
        B = B > 0.f ? B : 0.f; B = B < 1.f ? B : 1.f;
        if (SRGB) { R = splineAt(R * 1024.f, tabs->gammaSpline); G = splineAt(G * 1024.f, tabs->gammaSpline); B = splineAt(B * 1024.f, tabs->gammaSpline); }
        float t2 = B * k.c[2], t1 = G * k.c[1] + t2; const float X = R * k.c[0] + t1;
        t2 = B * k.c[5]; t1 = G * k.c[4] + t2; const float Y = R * k.c[3] + t1;
        t2 = B * k.c[8]; t1 = G * k.c[7] + t2; const float Z = R * k.c[6] + t1;
        L = splineAt(Y * tabScale, tabs->cbrtSpline);
        L = L * 116.f; L = L + -16.f;
        float den = Z * 3.f + X; den = Y * 15.f + den;
        const float dd = 52.f / (den > eps ? den : eps);
        float t0 = X * dd + -k.un; u = L * t0;
        t0 = (9.F * 0.25F) * Y; t0 = t0 * dd + -k.vn; v = L * t0;
    }
    if (U8) {
        uchar* d = dst + (size_t)y * dstep + (size_t)x * 3;
        const float fL = 255.f / 100.f, uRange = 354.f, vRange = 262.f, fu = 255.f / uRange, fv = 255.f / vRange, su = (134.f * 255.f) / uRange, sv = (140.f * 255.f) / vRange;
        float t = L * fL; d[0] = (uchar)sat8((int)__builtin_rintf(t));
        t = u * fu; t = t + su; d[1] = (uchar)sat8((int)__builtin_rintf(t));
        t = v * fv; t = t + sv; d[2] = (uchar)sat8((int)__builtin_rintf(t));
    } else {
        float* d = reinterpret_cast<float*>(dst + (size_t)y * dstep) + (size_t)x * 3;
        d[0] = L; d[1] = u; d[2] = v;
    }
}

// Luv2RGBfloat (color_lab.cpp:3057-3254): vector body / scalar tail as above (the tail has 1 / 903.3f where the body has 1 / 903.296296f)
template <int DCN, bool SRGB>
__global__ __launch_bounds__(256) void k_luv2bgr_f32(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H,
                                                      const LuvTabs* __restrict__ tabs, LuvF k)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), yy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || yy >= H) return;
    const float* s = reinterpret_cast<const float*>(src + (size_t)yy * sstep) + (size_t)x * 3;
    float* d = reinterpret_cast<float*>(dst + (size_t)yy * dstep) + (size_t)x * DCN;
    const float L = s[0], u = s[1], v = s[2];
    float R, G, B;
    if (x >= (W & ~7)) {
        float Y;
        if (L >= 8) { Y = (L + 16.f) * (1.f / 116.f); const float t = Y * Y; Y = t * Y; }
        else Y = L * (1.0f / 903.3f);
        float up = L * k.un; up = 3.f * (u + up);
        float vp = L * k.vn; vp = 0.25f / (v + vp);
        if (vp > 0.25f) vp = 0.25f;
        if (vp < -0.25f) vp = -0.25f;
        float X = Y * 3.f; X = X * up; X = X * vp;
        float Z = (12.f * 13.f) * L; Z = Z - up; Z = Z * vp; Z = Z - 5.f; Z = Y * Z;
        float t0 = X * k.c[0], t1 = Y * k.c[1], t2 = Z * k.c[2]; R = (t0 + t1) + t2;
        t0 = X * k.c[3]; t1 = Y * k.c[4]; t2 = Z * k.c[5]; G = (t0 + t1) + t2;
        t0 = X * k.c[6]; t1 = Y * k.c[7]; t2 = Z * k.c[8]; B = (t0 + t1) + t2;
        R = R < 0.f ? 0.f : R <= 1.f ? R : 1.f; G = G < 0.f ? 0.f : G <= 1.f ? G : 1.f; B = B < 0.f ? 0.f : B <= 1.f ? B : 1.f;
    } else {
        float Ylo = (L + 16.f) * (1.f / 116.f); { const float t = Ylo * Ylo; Ylo = t * Ylo; }
        const float Yhi = L * (1.0f / 903.296296f);
        const float Y = L >= 8.f ? Ylo : Yhi;
        float up = L * k.un + u; up = 3.f * up;
        float vp = L * k.vn + v; vp = 0.25f / vp;
        vp = 0.25f < vp ? 0.25f : vp;
        vp = -0.25f > vp ? -0.25f : vp;
        float X = 3.f * up; X = X * vp;
        float Z = L * (12.f * 13.f) + -up; Z = Z * vp + -5.f;
        float t = X * k.c[0] + k.c[1]; t = Z * k.c[2] + t; R = t * Y;
        t = X * k.c[3] + k.c[4]; t = Z * k.c[5] + t; G = t * Y;
        t = X * k.c[6] + k.c[7]; t = Z * k.c[8] + t; B = t * Y;
        R = R > 0.f ? R : 0.f; R = R < 1.f ? R : 1.f;
        G = G > 0.f ? G : 0.f; G = G < 1.f ? G : 1.f;
        B = B > 0.f ? B : 0.f; B = B < 1.f ? B : 1.f;
    }
    if (SRGB) { R = splineAt(R * 1024.f, tabs->invGammaSpline); G = splineAt(G * 1024.f, tabs->invGammaSpline); B = splineAt(B * 1024.f, tabs->invGammaSpline); }
    d[0] = R; d[1] = G; d[2] = B;
    if (DCN == 4) d[3] = 1.f;
}

// un, vn of the D65 white point as RGB2Luvfloat / Luv2RGBfloat derive them (:2902-2907, :3088-3093): the sum in double, everything after in float
void luvWhitePoint(float* un, float* vn)
{
    float d = (float)(kD65[0] + kD65[1] * 15.0 + kD65[2] * 3.0);
    d = 1.f / maxSoft(d, 1.1920928955078125e-7f);
    *un = (d * 52.f) * (float)kD65[0]; *vn = (d * 117.f) * (float)kD65[1];
}

} // namespace

extern "C" {

// replaces hal_ni_cvtBGRtoLab (hal_replacement.hpp:535-548): CV_8U, L*a*b*, sRGB or linear RGB; everything else is declined
MI355CV_API int mi355cv_cvtBGRtoLab(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int scn, bool swapBlue, bool isLab, bool srgb)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (scn != 3 && scn != 4) || width <= 0 || height <= 0");
    if (depth == MI355CV_32F && isLab) {
        Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
        const LuvTabs* ft = deviceLuvTabs();
        if (!ft) return setError(MI355CV_NOT_IMPLEMENTED, "cvtBGRtoLab: no device memory for the tables");
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn * 4, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 12, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        const dim3 grid(divUp(width, 64), divUp(height, 4));
        const int bIdx = swapBlue ? 2 : 0;
        if (srgb) {
            if (scn == 3) hipLaunchKernelGGL((k_bgr2lab_f32_grid<3>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, bIdx);
            else          hipLaunchKernelGGL((k_bgr2lab_f32_grid<4>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, bIdx);
        } else {
            // RGB2Lab_f's constructor (color_lab.cpp:1907-1930): rows scaled by 1 / white point in double, rounded to float, channel order folded in
            CoefF9 kf; const double sw[3] = {1.0 / kD65[0], 1.0, 1.0 / kD65[2]};
            for (int i = 0; i < 3; i++) {
                kf.c[i * 3 + (bIdx ^ 2)] = (float)(sw[i] * kRgb2Xyz[i * 3]);
                kf.c[i * 3 + 1]          = (float)(sw[i] * kRgb2Xyz[i * 3 + 1]);
                kf.c[i * 3 + bIdx]       = (float)(sw[i] * kRgb2Xyz[i * 3 + 2]);
            }
            if (scn == 3) hipLaunchKernelGGL((k_bgr2lab_f32_lin<3>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, kf);
            else          hipLaunchKernelGGL((k_bgr2lab_f32_lin<4>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, kf);
        }
        noteKernel("k_bgr2lab_f32_%s<%d> grid=%ux%u x256", srgb ? "grid" : "lin", scn, grid.x, grid.y);
        return stg.finish("cvtBGRtoLab");
    }
    if (depth != MI355CV_8U && depth != MI355CV_32F) return mi355::declined(__func__, __LINE__, "depth != MI355CV_8U && depth != MI355CV_32F");
    if (!isLab && (depth == MI355CV_32F || !srgb)) {
        // L*u*v* in float: CV_32F images, and CV_8U images in linear RGB (RGB2Luv_b color_lab.cpp:3389-3392 interpolates in the grid for sRGB only)
        const int e = depth == MI355CV_32F ? 4 : 1;
        Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % e) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % e");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
        const LuvTabs* ft = deviceLuvTabs();
        if (!ft) return setError(MI355CV_NOT_IMPLEMENTED, "cvtBGRtoLab: no device memory for the tables");
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn * e, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 3 * e, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        LuvF kf;
        for (int i = 0; i < 9; i++) kf.c[i] = (float)kRgb2Xyz[i];
        if (!swapBlue) for (int i = 0; i < 3; i++) std::swap(kf.c[i * 3], kf.c[i * 3 + 2]);            // blueIdx == 0 (:2891)
        luvWhitePoint(&kf.un, &kf.vn);
        const dim3 grid(divUp(width, 64), divUp(height, 4));
#define LAUNCH(SCN_, SRGB_, U8_) hipLaunchKernelGGL((k_bgr2luv_f32<SCN_, SRGB_, U8_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, kf)
        if (depth == MI355CV_8U) { if (scn == 3) LAUNCH(3, false, true); else LAUNCH(4, false, true); }
        else if (scn == 3) { if (srgb) LAUNCH(3, true, false); else LAUNCH(3, false, false); }
        else               { if (srgb) LAUNCH(4, true, false); else LAUNCH(4, false, false); }
#undef LAUNCH
        noteKernel("k_bgr2luv_f32<%d,%s,%s> grid=%ux%u x256", scn, srgb ? "srgb" : "linear", e == 1 ? "u8" : "f32", grid.x, grid.y);
        return stg.finish("cvtBGRtoLab");
    }
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const LabTabs* tabs = isLab ? deviceTabs() : nullptr;
    const LuvTabs* luv = isLab ? nullptr : deviceLuvTabs();
    if (!tabs && !luv) return setError(MI355CV_NOT_IMPLEMENTED, "cvtBGRtoLab: no device memory for the tables");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * scn, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * 3, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    if (!isLab) {
        const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
        const dim3 grid(divUp(divUp(width, 4), 64), divUp(height, 4));
        if (scn == 3) hipLaunchKernelGGL((k_bgr2luv_u8<3>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, al, luv, swapBlue ? 2 : 0);
        else          hipLaunchKernelGGL((k_bgr2luv_u8<4>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, al, luv, swapBlue ? 2 : 0);
        noteKernel("k_bgr2luv_u8<%d> grid=%ux%u x256", scn, grid.x, grid.y);
        return stg.finish("cvtBGRtoLab");
    }
    // RGB2Lab_b's constructor (color_lab.cpp:1590-1606): rows of sRGB -> XYZ divided by the white point, 2^12 fixed point, the channel order folded in
    Coef9 k; const int blueIdx = swapBlue ? 2 : 0;
    for (int i = 0; i < 3; i++) {
        const double ls = (double)(1 << LAB_SHIFT);
        k.c[i * 3 + (blueIdx ^ 2)] = (int)lrint(ls * kRgb2Xyz[i * 3] / kD65[i]);
        k.c[i * 3 + 1]             = (int)lrint(ls * kRgb2Xyz[i * 3 + 1] / kD65[i]);
        k.c[i * 3 + blueIdx]       = (int)lrint(ls * kRgb2Xyz[i * 3 + 2] / kD65[i]);
    }
    const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    const dim3 grid(divUp(divUp(width, 4), 64), divUp(height, ROWS_PER_BLOCK));
    hipStream_t st = stream();
#define LAUNCH(SCN_, SRGB_) hipLaunchKernelGGL((k_bgr2lab_u8<SCN_, SRGB_>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, al, tabs, k)
    if (scn == 3) { if (srgb) LAUNCH(3, true); else LAUNCH(3, false); }
    else          { if (srgb) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
    noteKernel("k_bgr2lab_u8<%d,%s> grid=%ux%u x256", scn, srgb ? "srgb" : "linear", grid.x, grid.y);
    return stg.finish("cvtBGRtoLab");
}

// replaces hal_ni_cvtLabtoBGR (hal_replacement.hpp:550-565)
MI355CV_API int mi355cv_cvtLabtoBGR(const uchar* src_data, size_t src_step, uchar* dst_data, size_t dst_step, int width, int height,
                                    int depth, int dcn, bool swapBlue, bool isLab, bool srgb)
{
    mi355::EntryGuard entry_(__func__);
    if (disabled() || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0) return mi355::declined(__func__, __LINE__, "disabled() || (dcn != 3 && dcn != 4) || width <= 0 || height <= 0");
    if (depth == MI355CV_32F && isLab) {
        Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
        const LuvTabs* ft = deviceLuvTabs();
        if (!ft) return setError(MI355CV_NOT_IMPLEMENTED, "cvtLabtoBGR: no device memory for the tables");
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * 12, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * 4, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        // Lab2RGBfloat's constructor (color_lab.cpp:2188-2200): XYZ -> RGB columns times the white point, in double, rounded to float
        CoefF9 kf; const int bi = swapBlue ? 2 : 0;
        for (int i = 0; i < 3; i++) {
            kf.c[i + (bi ^ 2) * 3] = (float)(kXyz2Rgb[i] * kD65[i]);
            kf.c[i + 3]            = (float)(kXyz2Rgb[i + 3] * kD65[i]);
            kf.c[i + bi * 3]       = (float)(kXyz2Rgb[i + 6] * kD65[i]);
        }
        const dim3 grid(divUp(width, 64), divUp(height, 4));
#define LAUNCH(DCN_, SRGB_) hipLaunchKernelGGL((k_lab2bgr_f32<DCN_, SRGB_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, kf)
        if (dcn == 3) { if (srgb) LAUNCH(3, true); else LAUNCH(3, false); }
        else          { if (srgb) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
        noteKernel("k_lab2bgr_f32<%d,%s> grid=%ux%u x256", dcn, srgb ? "srgb" : "linear", grid.x, grid.y);
        return stg.finish("cvtLabtoBGR");
    }
    if (depth == MI355CV_32F) {                             // L*u*v*, CV_32F: Luv2RGBfloat
        Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
        if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data) || ((uintptr_t)src_data | src_step | (uintptr_t)dst_data | dst_step) % 4");
        if (hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels(HOST_HEAVY))");
        const LuvTabs* ft = deviceLuvTabs();
        if (!ft) return setError(MI355CV_NOT_IMPLEMENTED, "cvtLabtoBGR: no device memory for the tables");
        size_t dss, dds;
        const uchar* ds = stg.in(src_data, src_step, (size_t)width * 12, height, &dss);
        uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn * 4, height, &dds);
        if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
        LuvF kf; const int bi = swapBlue ? 2 : 0;
        for (int i = 0; i < 3; i++) { kf.c[i + (bi ^ 2) * 3] = (float)kXyz2Rgb[i]; kf.c[i + 3] = (float)kXyz2Rgb[i + 3]; kf.c[i + bi * 3] = (float)kXyz2Rgb[i + 6]; }
        luvWhitePoint(&kf.un, &kf.vn);
        const dim3 grid(divUp(width, 64), divUp(height, 4));
#define LAUNCH(DCN_, SRGB_) hipLaunchKernelGGL((k_luv2bgr_f32<DCN_, SRGB_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, ft, kf)
        if (dcn == 3) { if (srgb) LAUNCH(3, true); else LAUNCH(3, false); }
        else          { if (srgb) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
        noteKernel("k_luv2bgr_f32<%d,%s> grid=%ux%u x256", dcn, srgb ? "srgb" : "linear", grid.x, grid.y);
        return stg.finish("cvtLabtoBGR");
    }
    if (depth != MI355CV_8U) return mi355::declined(__func__, __LINE__, "depth != MI355CV_8U");
    Stager stg;                                  // first: a declined call must also put the host's device back (~Stager)
    if (!ensureDevice() || inPlaceOnDevice(src_data, dst_data)) return mi355::declined(__func__, __LINE__, "!ensureDevice() || inPlaceOnDevice(src_data, dst_data)");
    if (hostImageTooSmall(src_data, (size_t)width * height, minPixels())) return mi355::declined(__func__, __LINE__, "hostImageTooSmall(src_data, (size_t)width * height, minPixels())");
    const LabTabs* tabs = deviceTabs();
    const LuvTabs* luv = isLab ? nullptr : deviceLuvTabs();
    if (!tabs || (!isLab && !luv)) return setError(MI355CV_NOT_IMPLEMENTED, "cvtLabtoBGR: no device memory for the tables");
    size_t dss, dds;
    const uchar* ds = stg.in(src_data, src_step, (size_t)width * 3, height, &dss);
    uchar* dd = stg.out(dst_data, dst_step, (size_t)width * dcn, height, &dds);
    if (!ds || !dd) return mi355::declined(__func__, __LINE__, "!ds || !dd");
    if (!isLab) {
        // Luv2RGBinteger's constructor (color_lab.cpp:3567-3584): XYZ -> sRGB in 2^12 fixed point, no white point (it is folded into the tables)
        Coef9 kq; const int bi = swapBlue ? 2 : 0;
        for (int i = 0; i < 3; i++) {
            const double ls = (double)(1 << LAB_SHIFT);
            kq.c[i + bi * 3]       = (int)lrint(ls * kXyz2Rgb[i]);
            kq.c[i + 3]            = (int)lrint(ls * kXyz2Rgb[i + 3]);
            kq.c[i + (bi ^ 2) * 3] = (int)lrint(ls * kXyz2Rgb[i + 6]);
        }
        const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
        const dim3 grid(divUp(divUp(width, 4), 64), divUp(height, ROWS_PER_BLOCK));
#define LAUNCH(DCN_, SRGB_) hipLaunchKernelGGL((k_luv2bgr_u8<DCN_, SRGB_>), grid, dim3(256), 0, stream(), ds, dss, dd, dds, width, height, al, tabs, luv, kq)
        if (dcn == 3) { if (srgb) LAUNCH(3, true); else LAUNCH(3, false); }
        else          { if (srgb) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
        noteKernel("k_luv2bgr_u8<%d,%s> grid=%ux%u x256", dcn, srgb ? "srgb" : "linear", grid.x, grid.y);
        return stg.finish("cvtLabtoBGR");
    }
    // Lab2RGBinteger's constructor (color_lab.cpp:2415-2437): columns of XYZ -> sRGB times the white point; stored B, G, R, so the first row is blue's
    Coef9 k; const int blueIdx = swapBlue ? 2 : 0;
    for (int i = 0; i < 3; i++) {
        const double ls = (double)(1 << LAB_SHIFT);
        k.c[i + blueIdx * 3]       = (int)lrint(ls * kXyz2Rgb[i] * kD65[i]);
        k.c[i + 3]                 = (int)lrint(ls * kXyz2Rgb[i + 3] * kD65[i]);
        k.c[i + (blueIdx ^ 2) * 3] = (int)lrint(ls * kXyz2Rgb[i + 6] * kD65[i]);
    }
    const int al = ((((uintptr_t)ds | dss | (uintptr_t)dd | dds) & 3) == 0) ? 1 : 0;
    const dim3 grid(divUp(divUp(width, 4), 64), divUp(height, ROWS_PER_BLOCK));
    hipStream_t st = stream();
#define LAUNCH(DCN_, SRGB_) hipLaunchKernelGGL((k_lab2bgr_u8<DCN_, SRGB_>), grid, dim3(256), 0, st, ds, dss, dd, dds, width, height, al, tabs, k)
    if (dcn == 3) { if (srgb) LAUNCH(3, true); else LAUNCH(3, false); }
    else          { if (srgb) LAUNCH(4, true); else LAUNCH(4, false); }
#undef LAUNCH
    noteKernel("k_lab2bgr_u8<%d,%s> grid=%ux%u x256", dcn, srgb ? "srgb" : "linear", grid.x, grid.y);
    return stg.finish("cvtLabtoBGR");
}

// diagnostics (tests): the host tables behind the two hooks.  which = 0 gamma (256 x u16), 1 cbrt (3072 x u16), 2 invGamma (4096 x u16),
// 3 yf (256 x u32: y | ify << 16), 4 the 33^3 x (L, u, v, 0) int16 table of RGB -> Luv, 5 / 6 LuToUp / LvToVp (65536 x i32).  Returns the entry
// count; needs no GPU.
MI355CV_API int mi355cv_labTable(int which, void* out)
{
    mi355::EntryGuard entry_(__func__);
    std::call_once(g_hostOnce, buildHost);
    switch (which) {
    case 0: std::memcpy(out, g_host.gamma, sizeof g_host.gamma); return 256;
    case 1: std::memcpy(out, g_host.cbrt, sizeof g_host.cbrt); return N_CBRT;
    case 2: std::memcpy(out, g_host.invGamma, sizeof g_host.invGamma); return N_INVG;
    case 3: std::memcpy(out, g_host.yf, sizeof g_host.yf); return 256;
    case 4: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->lut, sizeof g_luvHost->lut); return LUV_GRID * 4;
    case 5: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->up, sizeof g_luvHost->up); return 65536;
    case 6: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->vp, sizeof g_luvHost->vp); return 65536;
    case 7: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->labGrid, sizeof g_luvHost->labGrid); return LUV_GRID * 4;
    case 8: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->cbrtSpline, sizeof g_luvHost->cbrtSpline); return 4096;
    case 9: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->invGammaSpline, sizeof g_luvHost->invGammaSpline); return 4096;
    case 10: std::call_once(g_luvHostOnce, buildLuvHost); std::memcpy(out, g_luvHost->gammaSpline, sizeof g_luvHost->gammaSpline); return 4096;
    }
    return -1;
}

} // extern "C"
