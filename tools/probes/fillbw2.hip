// why do rows of 3841 ints write at 3.1 TB/s against 6.8 TB/s for aligned stores (fillbw.hip)?  Hypothesis: the 128-byte lines two neighbouring segments share are
// written half by one workgroup and half by another -- on another XCD, behind another L2 -- and each half goes to memory as a partial line.  Variants keep the
// bytes written the same and change only who writes what:
//   v1  wave = 256 ints of a row, block = 4 rows x 256 columns (the baseline of fillbw.hip)
//   v2  block = 1024 consecutive ints of one row
//   v3  v2, with block ids remapped so that each XCD (id % 8) owns one contiguous eighth of the buffer
//   v4  1024-thread block = a whole row of 3841 ints (4 ints per lane), rows in order
//   v5  v4 with the XCD remap (each XCD writes a contiguous run of rows)
//   v6  flat: the buffer as one long array, block = 1024 consecutive ints (misalignment only at the very ends) -- the upper bound for this address range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void seg(int* row, int c, int W) { if (c + 4 <= W) { i4u v = {c, 1, 2, 3}; *reinterpret_cast<i4u*>(row + c) = v; } else for (int k = 0; c + k < W; k++) row[c + k] = k; }
__global__ __launch_bounds__(256) void v1(int* p, int W, int H) { const int row = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.x * 256 + (threadIdx.x & 63) * 4; if (row < H && c < W) seg(p + ((size_t)blockIdx.z * H + row) * W, c, W); }
template <bool XCD> __global__ __launch_bounds__(256) void v2(int* p, int W, long long rows, unsigned nblk)
{
    unsigned id = blockIdx.x; if (XCD) { const unsigned per = nblk / 8; id = (id % 8) * per + id / 8; if (blockIdx.x >= per * 8) id = blockIdx.x; }
    const unsigned bpr = (W + 1023) / 1024; const long long row = id / bpr; const int c = (id % bpr) * 1024 + threadIdx.x * 4;
    if (row < rows && c < W) seg(p + (size_t)row * W, c, W);
}
template <bool XCD> __global__ __launch_bounds__(1024) void v4(int* p, int W, long long rows, unsigned nblk)
{
    unsigned id = blockIdx.x; if (XCD) { const unsigned per = nblk / 8; id = (id % 8) * per + id / 8; if (blockIdx.x >= per * 8) id = blockIdx.x; }
    const int c = threadIdx.x * 4; if (id < rows && c < W) seg(p + (size_t)id * W, c, W);
}
__global__ __launch_bounds__(256) void v6(int* p, size_t n) { const size_t c = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; if (c + 4 <= n) { i4u v = {1, 1, 2, 3}; *reinterpret_cast<i4u*>(p + c) = v; } }
template <class F> static float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main()
{
    const int W = 3841, H = 2161, F = 64; const long long rows = (long long)H * F; const size_t n = (size_t)W * rows, bytes = n * 4;
    int* d; if (hipMalloc(&d, bytes + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    d += 1;                                                   // the integral's images do not start on a line either
    auto rep = [&](const char* name, float ms) { printf("%-72s %.3f ms = %5.0f GB/s\n", name, ms, bytes / ms / 1e6); };
    rep("v1 block = 4 rows x 256 ints", timeit([&] { v1<<<dim3((W + 255) / 256, (H + 3) / 4, F), 256>>>(d, W, H); }));
    const unsigned nb2 = (unsigned)(rows * ((W + 1023) / 1024));
    rep("v2 block = 1024 consecutive ints of a row", timeit([&] { v2<false><<<nb2, 256>>>(d, W, rows, nb2); }));
    rep("v3 v2 + each XCD a contiguous eighth", timeit([&] { v2<true><<<nb2, 256>>>(d, W, rows, nb2); }));
    rep("v4 1024-thread block = one whole row", timeit([&] { v4<false><<<(unsigned)rows, 1024>>>(d, W, rows, (unsigned)rows); }));
    rep("v5 v4 + each XCD a contiguous run of rows", timeit([&] { v4<true><<<(unsigned)rows, 1024>>>(d, W, rows, (unsigned)rows); }));
    rep("v6 flat array, 16-byte stores at 4-byte alignment", timeit([&] { v6<<<(unsigned)((n / 4 + 255) / 256), 256>>>(d, n); }));
    return 0;
}
