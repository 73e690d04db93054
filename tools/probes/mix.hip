// probe: what does HBM deliver for the read : write MIX of each rolling kernel, with nothing but the loads and the stores?  A wave owns a strip of 64 lanes and walks a segment of
// rows like roll.h does; per step it loads RR source rows of RB bytes per lane and stores one row of WB bytes per lane (the stored value depends on every load, so nothing is
// dropped).  The fractions it prints are the ceilings the kernels' own fractions are to be read against:
//   Gaussian / box / 8U->8U filters   RB 16, RR 1, WB 16      (1 : 1)
//   pyrDown                           RB 16, RR 2, WB 8       (4 : 1)
//   Sobel 8U -> 16S                   RB 16, RR 1, WB 32      (1 : 2)
//   cornerHarris 8U -> 32F            RB 8,  RR 1, WB 32      (1 : 4)
//   read only / write only            the two extremes
// hipcc -O3 --offload-arch=gfx950 mix.hip -o mix && ./mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
template <int NB> struct Vec { unsigned d[NB / 4]; };
template <int NB> __device__ __forceinline__ Vec<NB> ld(const unsigned char* p)
{
    Vec<NB> v;
    if constexpr (NB == 16) { const v4 t = *reinterpret_cast<const v4*>(p); v.d[0] = t.x; v.d[1] = t.y; v.d[2] = t.z; v.d[3] = t.w; }
    else if constexpr (NB == 8) { const uint2 t = *reinterpret_cast<const uint2*>(p); v.d[0] = t.x; v.d[1] = t.y; }
    else v.d[0] = *reinterpret_cast<const unsigned*>(p);
    return v;
}
template <int NB> __device__ __forceinline__ void st(unsigned char* p, unsigned x)
{
    if constexpr (NB >= 16) {
#pragma unroll
        for (int q = 0; q < NB / 16; q++) __builtin_nontemporal_store(v4{x, x + 1, x + 2, x + 3}, reinterpret_cast<v4*>(p) + q);
    } else if constexpr (NB == 8) { typedef unsigned v2 __attribute__((ext_vector_type(2))); __builtin_nontemporal_store(v2{x, x + 1}, reinterpret_cast<v2*>(p)); }
    else if constexpr (NB == 4) *reinterpret_cast<unsigned*>(p) = x;
}
// source rows of `sstep` bytes, `H` of them per frame; a segment is SEG source rows; D source rows in flight per wave
template <int RB, int RR, int WB, int D>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t sstep, size_t sframe, size_t dstep, size_t dframe,
                                         int wbytes, int H, int SEG, int nstrips, int nseg, int nframes, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int strip = wid % nstrips, t0 = wid / nstrips, seg = t0 % nseg, frame = t0 / nseg;
    if (frame >= nframes) return;
    const int xb = (strip * 64 + lane) * RB;
    const bool live = xb < wbytes;
    const int y0 = seg * SEG, y1 = min(H, y0 + SEG);
    const unsigned char* s = src + (size_t)frame * sframe + (live ? xb : 0);
    unsigned char* d = dst + (size_t)frame * dframe + (size_t)(strip * 64 + lane) * WB;
    Vec<RB> ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = ld<RB>(s + (size_t)min(y0 + u, H - 1) * sstep);
    unsigned acc = 0;
    for (int y = y0; y < y1; y += D) {
#pragma unroll
        for (int u = 0; u < D; u += RR) {
            unsigned x = 0;
#pragma unroll
            for (int r = 0; r < RR; r++) {
#pragma unroll
                for (int q = 0; q < RB / 4; q++) x += ring[u + r].d[q];
                ring[u + r] = ld<RB>(s + (size_t)min(y + u + r + D, H - 1) * sstep);
            }
            if (WB > 0) { if (live && y + u < y1) st<WB>(d + (size_t)((y + u) / RR) * dstep, x); }
            else acc += x;
        }
    }
    if (WB == 0 && acc == 0x12345678u) *sink = acc;
}
template <int RB, int RR, int WB, int D>
void run(const char* name, const unsigned char* s, unsigned char* d, int W /* source bytes per row */, int H, int nf, int SEG, unsigned* sink)
{
    const size_t sstep = W, sframe = sstep * H;
    const size_t dstep = WB ? (size_t)W / RB * WB : 0, dframe = dstep * (H / RR);
    const int nstrips = (W / RB + 63) / 64, nseg = (H + SEG - 1) / SEG;
    const unsigned blocks = (unsigned)(((long long)nstrips * nseg * nf + 3) / 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 12; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<RB, RR, WB, D>), dim3(blocks), dim3(256), 0, 0, s, d, sstep, sframe, dstep, dframe, W, H, SEG, nstrips, nseg, nf, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 4) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)(sframe + dframe) * nf, ms = sum / 8;
    if (name[0] != '(') printf("%-34s %5d x %4d x %3d  seg %3d  ring %2d: %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s   (best %.3f)\n", name, W, H, nf, SEG, D, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, bytes / best / 1e6 / 8000);
}
int main()
{
    unsigned char *s, *d; unsigned* sink;
    const size_t cap = (size_t)3 << 30;
    hipMalloc(&s, cap); hipMalloc(&d, cap); hipMalloc(&sink, 4); hipMemset(s, 1, cap);
    for (int i = 0; i < 40; i++) run<16, 1, 16, 8>("(warm-up)", s, d, 3840, 2160, 64, 32, sink);
    puts("--- 4K, 64 frames");
    run<16, 1, 16, 8>("1 : 1  (Gaussian, box)", s, d, 3840, 2160, 64, 32, sink);
    run<16, 1, 32, 8>("1 : 2  (Sobel 8U -> 16S)", s, d, 3840, 2160, 64, 32, sink);
    run<16, 1, 0, 8>("read only", s, d, 3840, 2160, 64, 32, sink);
    puts("--- 1080p, 256 frames");
    run<16, 1, 16, 8>("1 : 1  (Gaussian, box)", s, d, 1920, 1080, 256, 32, sink);
    run<16, 2, 8, 8>("4 : 1  (pyrDown)", s, d, 1920, 1080, 256, 32, sink);
    run<16, 2, 8, 8>("4 : 1  (pyrDown) seg 64", s, d, 1920, 1080, 256, 64, sink);
    run<16, 2, 8, 16>("4 : 1  (pyrDown) ring 16", s, d, 1920, 1080, 256, 32, sink);
    run<8, 1, 32, 4>("1 : 4  (cornerHarris, 8 B chunks)", s, d, 1920, 1080, 256, 36, sink);
    run<8, 1, 32, 8>("1 : 4  (cornerHarris) ring 8", s, d, 1920, 1080, 256, 36, sink);
    run<16, 1, 0, 8>("read only", s, d, 1920, 1080, 256, 32, sink);
    puts("--- pyramid levels 1 -> 2 -> 3 -> 4 (960, 480, 240 wide), 256 frames");
    run<16, 2, 8, 8>("4 : 1  960 x 540", s, d, 960, 540, 256, 32, sink);
    run<16, 2, 8, 8>("4 : 1  480 x 270", s, d, 480, 270, 256, 32, sink);
    run<16, 2, 8, 8>("4 : 1  240 x 135 (width 240: 15 lanes)", s, d, 240, 135, 256, 32, sink);
    return 0;
}
