// probe (round 5): what bounds cfg3c (warpAffine 8K CV_32F bilinear) at 0.45 when even a pure shift runs there?  The same bilinear-tap traffic (a (+3, +2) pixel shift with
// fractional weights, 4 taps per destination pixel) on 7680 x 4320 float frames in four geometries; loads, the 4-tap blend and the store only.
//   V1  one pixel per lane, 8 rows per wave in flight, 8-byte tap-pair loads from two rows, 4-byte stores; workgroup = 64 columns x 32 rows   (k_warp_lin's geometry)
//   V2  as V1, workgroup = 256 columns x 8 rows (a workgroup's 4 waves side by side: 1 KiB of a row per workgroup-row)
//   V3  four pixels per lane: 16-byte + 4-byte loads from two rows, 4 rows in flight, one 16-byte nt store; wave = 256 columns; workgroup = 256 columns x 16 rows
//   V4  V3 walking down a strip of SEG rows, every source row loaded once (the lower taps of row y are the upper taps of row y + 1)
// hipcc -O3 --offload-arch=gfx950 shift.hip -o shift && ./shift
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int W = 7680, H = 4320, DX = 3, DY = 2;
__device__ __forceinline__ float blend(float a, float b, float c, float d) { return a * 0.28125f + b * 0.09375f + c * 0.46875f + d * 0.15625f; }

// VR: k_warp_lin's gather under a ROTATION (7 degrees x 0.95 about the centre, float coordinates -- timing only): one pixel per lane, 8 pixels per lane in flight, 8-byte tap
// pairs from two rows, 4-byte stores, with the wave's 64 lanes arranged LW wide x 64 / LW high.  A 64 x 1 row of destination pixels crosses ~9 source rows (26 cache lines per
// load instruction, profiles/r02_warp_pmc.txt); 32 x 2 and 16 x 4 footprints are compact (14 / 7 lines) but store 128- / 64-byte row pieces.
template <int LW>
__global__ __launch_bounds__(256) void vr(const float* __restrict__ src, float* __restrict__ dst, int nframes)
{
    constexpr int LH = 64 / LW, G = 8;                              // a wave: LW columns x (LH * G) rows; a workgroup: 4 waves side by side
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tilesX = W / (4 * LW);
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int x = (tx * 4 + wv) * LW + (lane % LW);
    const int yb = ty * (LH * G) + lane / LW;
    src += (size_t)blockIdx.y * W * H; dst += (size_t)blockIdx.y * W * H;
    const float c = 0.992546f / 0.95f, sn = 0.121869f / 0.95f, cx = W * 0.5f, cy = H * 0.5f;
    f2u a[G], b[G]; float fx[G], fy[G];
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int y = yb + i * LH;
        const float X = c * (x - cx) + sn * (y - cy) + cx, Y = -sn * (x - cx) + c * (y - cy) + cy;
        const int sx = min(max((int)floorf(X), 0), W - 2), sy = min(max((int)floorf(Y), 0), H - 2);
        fx[i] = X - floorf(X); fy[i] = Y - floorf(Y);
        a[i] = *reinterpret_cast<const f2u*>(src + (size_t)sy * W + sx);
        b[i] = *reinterpret_cast<const f2u*>(src + (size_t)(sy + 1) * W + sx);
    }
#pragma unroll
    for (int i = 0; i < G; i++) {
        const float t0 = a[i].x + fx[i] * (a[i].y - a[i].x), t1 = b[i].x + fx[i] * (b[i].y - b[i].x);
        dst[(size_t)(yb + i * LH) * W + x] = t0 + fy[i] * (t1 - t0);
    }
}

template <int BW>   // BW = 64: workgroup 64 x 32;  BW = 256: workgroup 256 x 8
__global__ __launch_bounds__(256) void v1(const float* __restrict__ src, float* __restrict__ dst, int nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tilesX = W / BW;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int x = BW == 64 ? tx * 64 + lane : tx * 256 + wv * 64 + lane;
    const int yb = BW == 64 ? (ty * 4 + wv) * 8 : ty * 8;
    src += (size_t)blockIdx.y * W * H; dst += (size_t)blockIdx.y * W * H;
    f2u a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int sy = min(yb + i + DY, H - 2), sx = min(x + DX, W - 2);
        a[i] = *reinterpret_cast<const f2u*>(src + (size_t)sy * W + sx);
        b[i] = *reinterpret_cast<const f2u*>(src + (size_t)(sy + 1) * W + sx);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) dst[(size_t)(yb + i) * W + x] = blend(a[i].x, a[i].y, b[i].x, b[i].y);
}

__global__ __launch_bounds__(256) void v3(const float* __restrict__ src, float* __restrict__ dst, int nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tilesX = W / 256;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int x = tx * 256 + lane * 4;
    const int yb = (ty * 4 + wv) * 4;
    src += (size_t)blockIdx.y * W * H; dst += (size_t)blockIdx.y * W * H;
    f4u a[4], b[4]; float a4[4], b4[4];
    const int sx = min(x + DX, W - 8);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sy = min(yb + i + DY, H - 2);
        a[i] = *reinterpret_cast<const f4u*>(src + (size_t)sy * W + sx); a4[i] = src[(size_t)sy * W + sx + 4];
        b[i] = *reinterpret_cast<const f4u*>(src + (size_t)(sy + 1) * W + sx); b4[i] = src[(size_t)(sy + 1) * W + sx + 4];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f4 o = {blend(a[i].x, a[i].y, b[i].x, b[i].y), blend(a[i].y, a[i].z, b[i].y, b[i].z), blend(a[i].z, a[i].w, b[i].z, b[i].w), blend(a[i].w, a4[i], b[i].w, b4[i])};
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dst + (size_t)(yb + i) * W + x));
    }
}

// V5: four pixels per lane like V3, but every pixel GATHERS its two 8-byte tap pairs on its own (what a rotated map needs); G rows in flight; one 16-byte store per row
template <int G, bool NT>
__global__ __launch_bounds__(256) void v5(const float* __restrict__ src, float* __restrict__ dst, int nframes)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tilesX = W / 256;
    const int tx = blockIdx.x % tilesX, ty = blockIdx.x / tilesX;
    const int x = tx * 256 + lane * 4;
    const int yb = (ty * 4 + wv) * G;
    src += (size_t)blockIdx.y * W * H; dst += (size_t)blockIdx.y * W * H;
    f2u a[G][4], b[G][4];
#pragma unroll
    for (int i = 0; i < G; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sy = min(yb + i + DY, H - 2), sx = min(x + k + DX, W - 2);
            a[i][k] = *reinterpret_cast<const f2u*>(src + (size_t)sy * W + sx);
            b[i][k] = *reinterpret_cast<const f2u*>(src + (size_t)(sy + 1) * W + sx);
        }
#pragma unroll
    for (int i = 0; i < G; i++) {
        f4 o = {blend(a[i][0].x, a[i][0].y, b[i][0].x, b[i][0].y), blend(a[i][1].x, a[i][1].y, b[i][1].x, b[i][1].y),
                blend(a[i][2].x, a[i][2].y, b[i][2].x, b[i][2].y), blend(a[i][3].x, a[i][3].y, b[i][3].x, b[i][3].y)};
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dst + (size_t)(yb + i) * W + x));
        else *reinterpret_cast<f4*>(dst + (size_t)(yb + i) * W + x) = o;
    }
}

template <int D>     // rows in flight
__global__ __launch_bounds__(256) void v4(const float* __restrict__ src, float* __restrict__ dst, int nframes, int SEG)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nstrips = W / 256, nseg = (H + SEG - 1) / SEG;
    const int strip = wid % nstrips, seg = (wid / nstrips) % nseg, frame = wid / (nstrips * nseg);
    if (frame >= nframes) return;
    src += (size_t)frame * W * H; dst += (size_t)frame * W * H;
    const int x = strip * 256 + lane * 4, sx = min(x + DX, W - 8);
    const int y0 = seg * SEG, y1 = min(H, y0 + SEG);
    f4u r[D]; float r4[D];
    f4u prev = *reinterpret_cast<const f4u*>(src + (size_t)min(y0 + DY, H - 1) * W + sx); float prev4 = src[(size_t)min(y0 + DY, H - 1) * W + sx + 4];
#pragma unroll
    for (int u = 0; u < D; u++) { const size_t o = (size_t)min(y0 + u + DY + 1, H - 1) * W + sx; r[u] = *reinterpret_cast<const f4u*>(src + o); r4[u] = src[o + 4]; }
    for (int y = y0; y < y1; y += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const f4u b = r[u]; const float b4 = r4[u];
            { const size_t o = (size_t)min(y + u + D + DY + 1, H - 1) * W + sx; r[u] = *reinterpret_cast<const f4u*>(src + o); r4[u] = src[o + 4]; }
            f4 o = {blend(prev.x, prev.y, b.x, b.y), blend(prev.y, prev.z, b.y, b.z), blend(prev.z, prev.w, b.z, b.w), blend(prev.w, prev4, b.w, b4)};
            if (y + u < y1) __builtin_nontemporal_store(o, reinterpret_cast<f4*>(dst + (size_t)(y + u) * W + x));
            prev = b; prev4 = b4;
        }
    }
}

template <class F> void timeit(const char* name, F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float sum = 0, best = 1e9f;
    for (int rep = 0; rep < 10; rep++) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = 8.0 * W * H * 8, ms = sum / 7;
    printf("%-78s %7.3f ms  %6.1f us / frame  %.3f of 8 TB/s  (best %.3f)\n", name, ms, ms * 1000 / 8, bytes / ms / 1e6 / 8000, bytes / best / 1e6 / 8000);
}
int main()
{
    float *s, *d; const int NF = 8;
    hipMalloc(&s, (size_t)NF * W * H * 4); hipMalloc(&d, (size_t)NF * W * H * 4); hipMemset(s, 0, (size_t)NF * W * H * 4);
    timeit("VR rotation 7 deg, gather, wave = 64 x 1 lanes (k_warp_lin)", [&] { hipLaunchKernelGGL(vr<64>, dim3((W / 256) * (H / 8), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("VR rotation 7 deg, gather, wave = 32 x 2 lanes", [&] { hipLaunchKernelGGL(vr<32>, dim3((W / 128) * (H / 16), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("VR rotation 7 deg, gather, wave = 16 x 4 lanes", [&] { hipLaunchKernelGGL(vr<16>, dim3((W / 64) * (H / 32), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("VR rotation 7 deg, gather, wave = 8 x 8 lanes", [&] { hipLaunchKernelGGL(vr<8>, dim3((W / 32) * (H / 64), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V1 1 px / lane, 8 rows in flight, workgroup 64 x 32 (k_warp_lin's geometry)", [&] { hipLaunchKernelGGL(v1<64>, dim3((W / 64) * (H / 32), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V2 1 px / lane, 8 rows in flight, workgroup 256 x 8", [&] { hipLaunchKernelGGL(v1<256>, dim3((W / 256) * (H / 8), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V3 4 px / lane, 4 rows in flight, 16 B nt stores, workgroup 256 x 16", [&] { hipLaunchKernelGGL(v3, dim3((W / 256) * (H / 16), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V5 4 px / lane, each pixel gathers its own 2 x 8 B tap pairs, 2 rows in flight, 16 B nt store", [&] { hipLaunchKernelGGL((v5<2, true>), dim3((W / 256) * (H / 8), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V5 the same, plain store", [&] { hipLaunchKernelGGL((v5<2, false>), dim3((W / 256) * (H / 8), NF), dim3(256), 0, 0, s, d, NF); });
    timeit("V5 4 rows in flight, nt store", [&] { hipLaunchKernelGGL((v5<4, true>), dim3((W / 256) * (H / 16), NF), dim3(256), 0, 0, s, d, NF); });
    for (int seg : {32}) {
        char nm[120];
        snprintf(nm, sizeof nm, "V4 4 px / lane, walking %d rows, 4 rows in flight, every source row loaded once", seg);
        timeit(nm, [&] { hipLaunchKernelGGL(v4<4>, dim3((unsigned)(((long long)(W / 256) * ((H + seg - 1) / seg) * NF + 3) / 4)), dim3(256), 0, 0, s, d, NF, seg); });
        snprintf(nm, sizeof nm, "V4 4 px / lane, walking %d rows, 8 rows in flight", seg);
        timeit(nm, [&] { hipLaunchKernelGGL(v4<8>, dim3((unsigned)(((long long)(W / 256) * ((H + seg - 1) / seg) * NF + 3) / 4)), dim3(256), 0, 0, s, d, NF, seg); });
    }
    return 0;
}
