// micro-benchmark: issue rate of the i8 MFMA shapes on gfx950 (one wave per SIMD, 8 independent accumulators)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(int iters, int* out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i acc[8]; v4i acc4[8];
    for (int i = 0; i < 8; i++) { for (int j = 0; j < 16; j++) acc[i][j] = 0; acc4[i] = v4i{0, 0, 0, 0}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[i], 0, 0, 0);
        }
    }
    int s = 0;
    for (int i = 0; i < 8; i++) { for (int j = 0; j < 16; j++) s += acc[i][j]; s += acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int shape = 0; shape < 2; shape++)
        for (int blocks : {1, 256, 512}) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, out);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) {
                    const double nmfma = (double)iters * 8;                      // per wave
                    const double macs = shape == 0 ? 32768.0 : 16384.0;
                    const double wavesPerSimd = blocks <= 256 ? 1.0 : 2.0;
                    printf("shape %s blocks %d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz), chip %.0f TOPS\n", shape == 0 ? "32x32x32" : "16x16x64",
                           blocks, ms, ms * 1e6 / (nmfma * wavesPerSimd), ms * 1e6 / (nmfma * wavesPerSimd) * 2.4,
                           2.0 * macs * nmfma * 4 * blocks / (ms * 1e-3) / 1e12);
                }
            }
        }
    return 0;
}
