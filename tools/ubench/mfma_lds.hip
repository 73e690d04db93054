// micro-benchmark: the inner loop of the matchTemplate MFMA kernel in isolation (one wave per SIMD, 4 waves per block, 130 KB LDS):
// per iteration 40 v_mfma_i32_32x32x32_i8 fed by 16 ds_read_b128 (A) and 25 ds_read_b32 + 20 v_alignbyte (B), double buffered.
// Variants: 0 MFMA only; 1 + A loads; 2 + A + B loads (no align); 3 + align; 4 = 3 with sched_group_barrier interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int PPITCH = 272, TPITCH = 200;

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(int iters, int* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* P = smem; unsigned char* T = smem + 383 * PPITCH;
    for (int i = threadIdx.x; i < (383 * PPITCH + 128 * TPITCH) / 4; i += 256) reinterpret_cast<int*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    int bOff[5], bSh[5];
    for (int ks = 0; ks < 5; ks++) { const int o = 32 + 32 * ks + 16 * h - m; bOff[ks] = o & ~3; bSh[ks] = o & 3; }
    v16i acc[2][4];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 4; b++) for (int i = 0; i < 16; i++) acc[a][b][i] = 0;
    const unsigned char* Pw = P + (wave * 64 + m) * PPITCH + 16 * h;
    struct Frag { v4i A[2][8]; unsigned R[5][5]; v4i B[5]; };
    auto load = [&](Frag& f, int r) {
        if (V >= 1) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int cb = 0; cb < 8; cb++) f.A[mt][cb] = *reinterpret_cast<const v4i*>(Pw + (32 * mt + r) * PPITCH + 32 * cb);
        }
        if (V >= 2) {
            const unsigned char* Tr = T + r * TPITCH;
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                const unsigned* tp = reinterpret_cast<const unsigned*>(Tr + bOff[ks]);
#pragma unroll
                for (int d = 0; d < 5; d++) f.R[ks][d] = tp[d];
            }
        }
    };
    auto compute = [&](Frag& f) {
#pragma unroll
        for (int ks = 0; ks < 5; ks++) {
            v4i b = f.B[ks];
            if (V == 2) { b.x = f.R[ks][0]; b.y = f.R[ks][1]; b.z = f.R[ks][2]; b.w = f.R[ks][3] + f.R[ks][4]; }
            if (V >= 3) {
                b.x = __builtin_amdgcn_alignbyte(f.R[ks][1], f.R[ks][0], bSh[ks]); b.y = __builtin_amdgcn_alignbyte(f.R[ks][2], f.R[ks][1], bSh[ks]);
                b.z = __builtin_amdgcn_alignbyte(f.R[ks][3], f.R[ks][2], bSh[ks]); b.w = __builtin_amdgcn_alignbyte(f.R[ks][4], f.R[ks][3], bSh[ks]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.A[mt][nt + ks], b, acc[mt][nt], 0, 0, 0);
        }
    };
    Frag F0, F1;
    for (int mt = 0; mt < 2; mt++) for (int cb = 0; cb < 8; cb++) { F0.A[mt][cb] = v4i{lane, cb, mt, 1}; F1.A[mt][cb] = v4i{lane, cb, mt, 2}; }
    for (int ks = 0; ks < 5; ks++) { F0.B[ks] = v4i{ks, lane, 3, 4}; F1.B[ks] = v4i{ks, lane, 5, 6}; for (int d = 0; d < 5; d++) { F0.R[ks][d] = d; F1.R[ks][d] = d + 1; } }
    load(F0, 0);
    for (int it = 0; it < iters; it += 2) {
        const int r = it & 127;
        load(F1, r + 1 > 127 ? 0 : r + 1); compute(F0);
        if (V == 4) {
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
                for (int q = 0; q < 8; q++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load(F0, r + 2 > 127 ? 0 : r + 2); compute(F1);
        if (V == 4) {
#pragma unroll
            for (int ks = 0; ks < 5; ks++) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
                for (int q = 0; q < 8; q++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    int s = 0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 4; b++) for (int i = 0; i < 16; i++) s += acc[a][b][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V> void run(int blocks, int iters, int* out)
{
    const size_t lds = 383 * PPITCH + 128 * TPITCH;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("variant %d blocks %d: %.3f ms, %.1f ns per iteration (40 MFMA) = %.1f cycles @2.4GHz per MFMA; err=%d\n", V, blocks, ms, ms * 1e6 / iters,
           ms * 1e6 / iters / 40 * 2.4, (int)hipGetLastError());
}
int main()
{
    int* out; hipMalloc(&out, 1024 * 256 * 4);
    for (int blocks : {1, 256}) { run<0>(blocks, 12800, out); run<1>(blocks, 12800, out); run<2>(blocks, 12800, out); run<3>(blocks, 12800, out); run<4>(blocks, 12800, out); }
    return 0;
}
