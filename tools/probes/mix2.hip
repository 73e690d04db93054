// probe (round 5): the write-heavy mixes (cornerHarris 1 : 4, Sobel 8U -> 16S 1 : 2, integral 1 : 4) under different STORE GEOMETRIES, loads and stores only.
// Same row-walking skeleton as mix.hip (a wave owns 64 chunks of a row and walks a segment of rows), plus:
//   xcd = 1   each XCD takes a contiguous eighth of the work items (block b -> XCD b % 8 is the observed dispatch rule; the remap makes XCD x process items [x N/8, (x+1) N/8))
//   nt  = 0   plain stores instead of non-temporal ones
//   RB  = 0   no loads at all (the row-walking fill ceiling)
//   wgrows    the 4 waves of a workgroup take 4 consecutive SEGMENTS of one strip instead of 4 consecutive strips of one segment
// hipcc -O3 --offload-arch=gfx950 mix2.hip -o mix2 && ./mix2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));
template <int NB> struct Vec { unsigned d[NB >= 4 ? NB / 4 : 1]; };
template <int NB> __device__ __forceinline__ Vec<NB> ld(const unsigned char* p)
{
    Vec<NB> v;
    if constexpr (NB == 16) { const v4 t = *reinterpret_cast<const v4*>(p); v.d[0] = t.x; v.d[1] = t.y; v.d[2] = t.z; v.d[3] = t.w; }
    else if constexpr (NB == 8) { const v2 t = *reinterpret_cast<const v2*>(p); v.d[0] = t.x; v.d[1] = t.y; }
    else if constexpr (NB == 4) v.d[0] = *reinterpret_cast<const unsigned*>(p);
    else v.d[0] = 0;
    return v;
}
template <int NB, bool NT> __device__ __forceinline__ void st(unsigned char* p, unsigned x)
{
#pragma unroll
    for (int q = 0; q < NB / 16; q++) {
        if (NT) __builtin_nontemporal_store(v4{x, x + 1, x + 2, x + 3}, reinterpret_cast<v4*>(p) + q);
        else reinterpret_cast<v4*>(p)[q] = v4{x, x + 1, x + 2, x + 3};
    }
}
template <int RB, int WB, int D, bool NT>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t sstep, size_t sframe, size_t dstep, size_t dframe,
                                         int nchunks, int H, int SEG, int nstrips, int nseg, int nframes, int xcd, int wgrows, unsigned nblocks)
{
    const int lane = threadIdx.x & 63;
    unsigned b = blockIdx.x;
    if (xcd) { const unsigned per = nblocks / 8; if (b < per * 8) b = (b % 8) * per + b / 8; }
    int wid = __builtin_amdgcn_readfirstlane((int)(b * 4 + (threadIdx.x >> 6)));
    int strip, seg, frame;
    if (wgrows) {                                   // work items ordered segment (4 consecutive in a workgroup) -> strip -> frame
        seg = wid % nseg; const int t0 = wid / nseg; strip = t0 % nstrips; frame = t0 / nstrips;
    } else { strip = wid % nstrips; const int t0 = wid / nstrips; seg = t0 % nseg; frame = t0 / nseg; }
    if (frame >= nframes) return;
    const int c = strip * 64 + lane;
    const bool live = c < nchunks;
    const int y0 = seg * SEG, y1 = min(H, y0 + SEG);
    const unsigned char* s = src + (size_t)frame * sframe + (size_t)(live ? c : 0) * (RB ? RB : 1);
    unsigned char* d = dst + (size_t)frame * dframe + (size_t)c * WB;
    Vec<RB> ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = ld<RB>(s + (size_t)min(y0 + u, H - 1) * sstep);
    for (int y = y0; y < y1; y += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            unsigned x = y;
#pragma unroll
            for (int q = 0; q < (RB >= 4 ? RB / 4 : 1); q++) x += ring[u].d[q];
            if (RB) ring[u] = ld<RB>(s + (size_t)min(y + u + D, H - 1) * sstep);
            if (live && y + u < y1) st<WB, NT>(d + (size_t)(y + u) * dstep, x);
        }
    }
}
// pyrDown's mix (4 : 1): two source rows of 16 bytes per lane per output row of 8 bytes per lane.  HALF = the same bytes stored by the lower 32 lanes as 16 bytes each
// (what an LDS transposition of the wave's 512-byte row piece would issue), against 64 lanes x 8 bytes
template <int D, bool HALF>
__global__ __launch_bounds__(256) void kp(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t sstep, size_t sframe, size_t dstep, size_t dframe,
                                          int nchunks, int H, int SEG, int nstrips, int nseg, int nframes)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int strip = wid % nstrips, t0 = wid / nstrips, seg = t0 % nseg, frame = t0 / nseg;
    if (frame >= nframes) return;
    const int c = strip * 64 + lane;
    const bool live = c < nchunks;
    const int y0 = seg * SEG, y1 = min(H, y0 + SEG);
    const unsigned char* s = src + (size_t)frame * sframe + (size_t)(live ? c : 0) * 16;
    unsigned char* d = dst + (size_t)frame * dframe + (size_t)strip * 512;
    Vec<16> ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = ld<16>(s + (size_t)min(y0 + u, H - 1) * sstep);
    for (int y = y0; y < y1; y += D) {
#pragma unroll
        for (int u = 0; u < D; u += 2) {
            unsigned x = y;
#pragma unroll
            for (int r = 0; r < 2; r++) {
#pragma unroll
                for (int q = 0; q < 4; q++) x += ring[u + r].d[q];
                ring[u + r] = ld<16>(s + (size_t)min(y + u + r + D, H - 1) * sstep);
            }
            if (live && y + u < y1) {
                unsigned char* row = d + (size_t)((y + u) >> 1) * dstep;
                if (HALF) { if (lane < 32) __builtin_nontemporal_store(v4{x, x + 1, x + 2, x + 3}, reinterpret_cast<v4*>(row) + lane); }
                else __builtin_nontemporal_store(v2{x, x + 1}, reinterpret_cast<v2*>(row) + lane);
            }
        }
    }
}
template <int D, bool HALF>
void runp(const char* name, const unsigned char* s, unsigned char* d, int W, int H, int nf, int SEG)
{
    const size_t sstep = W, sframe = sstep * H, dstep = W / 2, dframe = dstep * (H / 2);
    const int nchunks = W / 16, nstrips = (nchunks + 63) / 64, nseg = (H + SEG - 1) / SEG;
    const unsigned blocks = (unsigned)(((long long)nstrips * nseg * nf + 3) / 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 12; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((kp<D, HALF>), dim3(blocks), dim3(256), 0, 0, s, d, sstep, sframe, dstep, dframe, nchunks, H, SEG, nstrips, nseg, nf);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 4) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)(sframe + dframe) * nf, ms = sum / 8;
    printf("%-58s seg %3d              : %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s  (best %.3f)\n", name, SEG, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, bytes / best / 1e6 / 8000);
}

template <int RB, int WB, int D, bool NT>
void run(const char* name, const unsigned char* s, unsigned char* d, int W /* pixels = source bytes per row */, int H, int nf, int SEG, int xcd, int wgrows, int PXL /* pixels per lane */)
{
    const size_t sstep = W, sframe = sstep * H;
    const size_t dstep = (size_t)W / PXL * WB, dframe = dstep * H;
    const int nchunks = W / PXL, nstrips = (nchunks + 63) / 64, nseg = (H + SEG - 1) / SEG;
    const unsigned blocks = (unsigned)(((long long)nstrips * nseg * nf + 3) / 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 12; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<RB, WB, D, NT>), dim3(blocks), dim3(256), 0, 0, s, d, sstep, sframe, dstep, dframe, nchunks, H, SEG, nstrips, nseg, nf, xcd, wgrows, blocks);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 4) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)((RB ? sframe : 0) + dframe) * nf, ms = sum / 8;
    if (name[0] != '(') printf("%-58s seg %3d xcd %d wgrows %d: %7.3f ms  %6.0f GB/s  %.3f of 8 TB/s  (best %.3f)\n", name, SEG, xcd, wgrows, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000, bytes / best / 1e6 / 8000);
}
int main()
{
    unsigned char *s, *d;
    const size_t cap = (size_t)3 << 30;
    hipMalloc(&s, cap); hipMalloc(&d, cap); hipMemset(s, 1, cap);
    for (int i = 0; i < 40; i++) run<8, 32, 4, true>("(warm-up)", s, d, 1920, 1080, 256, 36, 0, 0, 8);
    puts("--- 1 : 4  cornerHarris 1080p 8U -> 32F, 256 frames");
    for (int xcd = 0; xcd < 2; xcd++) {
        run<8, 32, 4, true>("8 px / lane: 8 B load, 2 x 16 B nt stores (today)", s, d, 1920, 1080, 256, 36, xcd, 0, 8);
        run<8, 32, 4, false>("8 px / lane, plain stores", s, d, 1920, 1080, 256, 36, xcd, 0, 8);
        run<4, 16, 8, true>("4 px / lane: 4 B load, 1 x 16 B nt store", s, d, 1920, 1080, 256, 36, xcd, 0, 4);
        run<4, 16, 8, false>("4 px / lane, plain store", s, d, 1920, 1080, 256, 36, xcd, 0, 4);
        run<16, 64, 4, true>("16 px / lane: 16 B load, 4 x 16 B nt stores", s, d, 1920, 1080, 256, 36, xcd, 0, 16);
        run<8, 32, 4, true>("8 px / lane, workgroup = 4 segments of a strip", s, d, 1920, 1080, 256, 36, xcd, 1, 8);
        run<4, 16, 8, true>("4 px / lane, workgroup = 4 segments of a strip", s, d, 1920, 1080, 256, 36, xcd, 1, 4);
        run<4, 16, 8, true>("4 px / lane, seg 72", s, d, 1920, 1080, 256, 72, xcd, 0, 4);
        run<4, 16, 8, true>("4 px / lane, seg 18", s, d, 1920, 1080, 256, 18, xcd, 0, 4);
        run<0, 32, 4, true>("fill only, 2 x 16 B nt stores per lane", s, d, 1920, 1080, 256, 36, xcd, 0, 8);
        run<0, 16, 8, true>("fill only, 1 x 16 B nt store per lane", s, d, 1920, 1080, 256, 36, xcd, 0, 4);
        run<0, 16, 8, false>("fill only, 1 x 16 B plain store per lane", s, d, 1920, 1080, 256, 36, xcd, 0, 4);
        run<8, 0, 4, true>("(skip)", s, d, 1920, 1080, 256, 36, xcd, 0, 8);
    }
    puts("--- 1 : 2  Sobel 4K 8U -> 16S, 64 frames");
    for (int xcd = 0; xcd < 2; xcd++) {
        run<16, 32, 8, true>("16 px / lane: 16 B load, 2 x 16 B nt stores (today)", s, d, 3840, 2160, 64, 32, xcd, 0, 16);
        run<8, 16, 8, true>("8 px / lane: 8 B load, 1 x 16 B nt store", s, d, 3840, 2160, 64, 32, xcd, 0, 8);
        run<8, 16, 8, false>("8 px / lane, plain store", s, d, 3840, 2160, 64, 32, xcd, 0, 8);
        run<8, 16, 8, true>("8 px / lane, workgroup = 4 segments of a strip", s, d, 3840, 2160, 64, 32, xcd, 1, 8);
    }
    puts("--- 4 : 1  pyrDown 1080p -> 960 x 540, 256 frames");
    runp<8, false>("2 x 16 B loads, 8 B nt store per lane (today)", s, d, 1920, 1080, 256, 32);
    runp<8, true>("2 x 16 B loads, 16 B nt store by the lower 32 lanes", s, d, 1920, 1080, 256, 32);
    runp<16, false>("the same, 16 source rows in flight, 8 B stores", s, d, 1920, 1080, 256, 32);
    runp<16, true>("16 source rows in flight, 16 B stores by 32 lanes", s, d, 1920, 1080, 256, 32);
    puts("--- 1 : 1  Gaussian 4K, 64 frames (the reference point)");
    for (int xcd = 0; xcd < 2; xcd++) run<16, 16, 8, true>("16 px / lane: 16 B load, 16 B nt store", s, d, 3840, 2160, 64, 32, xcd, 0, 16);
    return 0;
}
