// pix4.h -- the launch shape shared by the per-pixel colour kernels whose channels are bytes: every lane owns FOUR consecutive pixels of a
// row, i.e. 4*SB source bytes and 4*DB destination bytes moved as whole dwords (1..4 per lane, so a wave reads and writes contiguous
// 256..1024-byte runs), and the conversion itself works on bytes picked out of those registers.  Rows whose base or pitch is not a
// multiple of four, and the last (partial) group of a row, fall back to byte accesses.
#pragma once
#include "rt.h"

namespace pix4 {

template <int NB>                                             // NB bytes per pixel, four pixels
struct Px {
    unsigned w[NB];
    __device__ __forceinline__ int get(int i) const { return (int)((w[i >> 2] >> ((i & 3) * 8)) & 255u); }
    __device__ __forceinline__ void put(int i, int v) { w[i >> 2] |= (unsigned)v << ((i & 3) * 8); }     // v in 0..255, target byte still zero
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < NB; i++) w[i] = 0;
    }
};

template <int SB, int DB, class Op>
__global__ __launch_bounds__(256) void k_pix4(const uchar* __restrict__ src, size_t sstep, uchar* __restrict__ dst, size_t dstep, int W, int H, int aligned, Op op)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= W || y >= H) return;
    const uchar* s = src + (size_t)y * sstep + (size_t)x4 * SB;
    uchar* d = dst + (size_t)y * dstep + (size_t)x4 * DB;
    const int n = min(4, W - x4);
    const bool fast = n == 4 && aligned;
    Px<SB> in; Px<DB> out;
    out.clear();
    if (fast) {
#pragma unroll
        for (int i = 0; i < SB; i++) in.w[i] = ((const unsigned*)s)[i];
    } else {
        in.clear();
#pragma unroll
        for (int i = 0; i < 4 * SB; i++) if (i < n * SB) in.put(i, s[i]);
    }
    op(in, out);
    if (fast) {
#pragma unroll
        for (int i = 0; i < DB; i++) ((unsigned*)d)[i] = out.w[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4 * DB; i++) if (i < n * DB) d[i] = (uchar)out.get(i);
    }
}

template <int SB, int DB, class Op>
inline void launch(hipStream_t st, const uchar* s, size_t ss, uchar* d, size_t ds, int W, int H, const Op& op)
{
    const int al = ((((uintptr_t)s | ss | (uintptr_t)d | ds) & 3) == 0) ? 1 : 0;
    hipLaunchKernelGGL((k_pix4<SB, DB, Op>), dim3(mi355::divUp(mi355::divUp(W, 4), 64), mi355::divUp(H, 4)), dim3(256), 0, st, s, ss, d, ds, W, H, al, op);
}

} // namespace pix4
