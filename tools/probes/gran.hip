// probe: does the bytes-per-lane of a row-walking copy bound it?  A wave owns a strip of 64 lanes x B bytes of a 7680 x 4320 float image (30 720-byte rows) and walks
// R rows down, loading and storing B bytes per lane and row -- the access pattern of the warp kernels (B = 4: k_warp_lin / k_warp32_tile) against that of the rolling
// kernels (B = 16).  hipcc -O3 --offload-arch=gfx950 gran.hip -o gran && ./gran
#include <hip/hip_runtime.h>
#include <cstdio>
template <int B, int R>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t step, size_t frame, int W, int H)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (blockIdx.x * 64 + lane) * B;                      // byte column
    const int y0 = (blockIdx.y * 4 + wave) * R;
    if (x >= W || y0 >= H) return;
    const unsigned char* s = src + (size_t)blockIdx.z * frame + (size_t)y0 * step + x;
    unsigned char* d = dst + (size_t)blockIdx.z * frame + (size_t)y0 * step + x;
    typedef unsigned v4 __attribute__((ext_vector_type(4)));
    typedef unsigned v2 __attribute__((ext_vector_type(2)));
    if (B == 4) { unsigned v[R];
#pragma unroll
        for (int i = 0; i < R; i++) v[i] = *reinterpret_cast<const unsigned*>(s + (size_t)i * step);
#pragma unroll
        for (int i = 0; i < R; i++) *reinterpret_cast<unsigned*>(d + (size_t)i * step) = v[i];
    } else if (B == 8) { v2 v[R];
#pragma unroll
        for (int i = 0; i < R; i++) v[i] = *reinterpret_cast<const v2*>(s + (size_t)i * step);
#pragma unroll
        for (int i = 0; i < R; i++) *reinterpret_cast<v2*>(d + (size_t)i * step) = v[i];
    } else { v4 v[R];
#pragma unroll
        for (int i = 0; i < R; i++) v[i] = *reinterpret_cast<const v4*>(s + (size_t)i * step);
#pragma unroll
        for (int i = 0; i < R; i++) *reinterpret_cast<v4*>(d + (size_t)i * step) = v[i];
    }
}
template <int B, int R> void run(const unsigned char* s, unsigned char* d, size_t step, size_t frame, int W, int H, int nf)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid((W / B + 63) / 64, (H + 4 * R - 1) / (4 * R), nf);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k<B, R>), grid, dim3(256), 0, 0, s, d, step, frame, W, H);
    hipEventRecord(a);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<B, R>), grid, dim3(256), 0, 0, s, d, step, frame, W, H);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%2d bytes per lane, %2d rows per wave: %7.3f ms  %6.0f GB/s  (%.3f of 8 TB/s)\n", B, R, ms, 2.0 * frame * nf / ms / 1e6, 2.0 * frame * nf / ms / 1e6 / 8000);
}
int main()
{
    const int W = 7680 * 4, H = 4320, nf = 16;
    const size_t step = W, frame = step * H;
    unsigned char *s, *d;
    hipMalloc(&s, frame * nf); hipMalloc(&d, frame * nf); hipMemset(s, 1, frame * nf);
    run<4, 8>(s, d, step, frame, W, H, nf); run<4, 16>(s, d, step, frame, W, H, nf); run<4, 32>(s, d, step, frame, W, H, nf);
    run<8, 8>(s, d, step, frame, W, H, nf); run<8, 16>(s, d, step, frame, W, H, nf);
    run<16, 8>(s, d, step, frame, W, H, nf); run<16, 16>(s, d, step, frame, W, H, nf);
    return 0;
}
