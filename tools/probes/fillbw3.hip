// does it matter HOW a wave's lanes cover a contiguous output row?  The rolling kernels give a lane 16 pixels: 16 bytes of CV_8U output (one dwordx4 store, the
// wave writes 1 KB contiguous), but 32 bytes of CV_16S (two dwordx4 stores 16 bytes apart: each instruction covers every other 16-byte piece of 2 KB) and 64
// bytes of CV_32F (four).  Same bytes, aligned rows, write-only:
//   S pieces per lane, strided:    instruction k writes piece S * lane + k      (what the kernels do)
//   S pieces per lane, contiguous: instruction k writes piece 64 * k + lane     (what a cross-lane exchange before the store would give)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int S, bool CONTIG> __global__ __launch_bounds__(256) void k(uint4* p, size_t npieces)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); const int lane = threadIdx.x & 63;
    const size_t base = wave * 64 * S;
    if (base + 64 * S > npieces) return;
#pragma unroll
    for (int k2 = 0; k2 < S; k2++) { const size_t i = base + (CONTIG ? 64 * k2 + lane : S * lane + k2); p[i] = make_uint4((unsigned)i, 1, 2, 3); }
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
    const size_t bytes = 2ull << 30, np = bytes / 16;
    uint4* d; if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
#define RUN(S_, C_) { const unsigned nb = (unsigned)(np / (256 * S_)); float ms = timeit([&] { k<S_, C_><<<nb, 256>>>(d, np); }); \
    printf("%d x 16 bytes per lane, %-10s %.3f ms = %5.0f GB/s\n", S_, C_ ? "contiguous" : "strided", ms, bytes / ms / 1e6); }
    RUN(1, true); RUN(2, false); RUN(2, true); RUN(4, false); RUN(4, true);
    return 0;
}
