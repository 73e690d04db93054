// write-only bandwidth on gfx950: what a store-dominated kernel (cv::integral: 1 B read, 4 B written per pixel) can hope for.  (tools/gpu_call12.sh)
//   aligned    every lane one 16-byte store, addresses 16-byte aligned, consecutive
//   rows3841   rows of 3841 ints (cv::integral's 4K output pitch, rows only 4-byte aligned): a wave writes 256 consecutive ints of a row with 16-byte stores
//   rows3841d  the same bytes with 4-byte stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_aligned(uint4* p, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = make_uint4(i, 1, 2, 3); }
template <bool NT> __global__ __launch_bounds__(256) void k_rows(int* p, int W, int H, int frames)
{
    // grid: (ceil(W / 256), H / 4, frames); wave w of the block = row 4 * blockIdx.y + w; lane = 4 consecutive ints
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    if (row >= H || c >= W) return;
    int* q = p + ((size_t)blockIdx.z * H + row) * W + c;
    typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
    if (c + 4 <= W) { i4u v = {c, row, 2, 3}; if (NT) __builtin_nontemporal_store(v, reinterpret_cast<i4u*>(q)); else *reinterpret_cast<i4u*>(q) = v; }
    else for (int k = 0; c + k < W; k++) q[k] = k;
}
__global__ __launch_bounds__(256) void k_rowsd(int* p, int W, int H, int frames)
{
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (row >= H) return;
    int* q = p + ((size_t)blockIdx.z * H + row) * W;
    for (int k = 0; k < 4; k++) { const int c = blockIdx.x * 256 + k * 64 + (threadIdx.x & 63); if (c < W) q[c] = c; }
}
template <class F> static float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main()
{
    const int W = 3841, H = 2161, F = 64; const size_t bytes = (size_t)W * H * F * 4;
    int* d; if (hipMalloc(&d, bytes + 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const size_t n16 = bytes / 16;
    float ms = timeit([&] { k_aligned<<<dim3((unsigned)((n16 + 255) / 256)), 256>>>((uint4*)d, n16); });
    printf("aligned 16-byte stores:           %.3f ms for %.2f GB = %.0f GB/s\n", ms, bytes / 1e9, bytes / ms / 1e6);
    dim3 g((W + 255) / 256, (H + 3) / 4, F);
    ms = timeit([&] { k_rows<false><<<g, 256>>>(d, W, H, F); });
    printf("rows of 3841 ints, 16-byte stores: %.3f ms = %.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { k_rows<true><<<g, 256>>>(d, W, H, F); });
    printf("rows of 3841 ints, 16-byte nontemporal stores: %.3f ms = %.0f GB/s\n", ms, bytes / ms / 1e6);
    ms = timeit([&] { k_rowsd<<<g, 256>>>(d, W, H, F); });
    printf("rows of 3841 ints, 4-byte stores:  %.3f ms = %.0f GB/s\n", ms, bytes / ms / 1e6);
    return 0;
}
