// pkrate.hip -- issue rate of the 16-bit packed min / max the median kernels are made of, against full-rate 32-bit forms: N waves per SIMD run
// chains of one instruction; cycles per wave instruction per SIMD = elapsed * clock / (instructions per wave * waves per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/pkrate.hip -o tools/probes/pkrate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t seed)
{
    uint32_t a[CHAINS], b[CHAINS];
    for (int i = 0; i < CHAINS; i++) { a[i] = seed * (threadIdx.x + 1) + i * 0x01010101u; b[i] = seed ^ (0x00ff00ffu * (i + 1)); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 1) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 3) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 4) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 5) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 9) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            if (OP == 10) asm volatile("v_max3_u16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < CHAINS; i++) s ^= a[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
}

template <int OP> void run(const char* name, uint32_t* d, double ghz, int cus)
{
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = cus * wavesPerSimd;                      // 256 threads = 4 waves = one per SIMD of a CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 7u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 7u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)ITERS * CHAINS * wavesPerSimd;
        printf("%-16s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave instruction per SIMD (at %.2f GHz)\n", name, wavesPerSimd, ms, ms * 1e-3 * ghz * 1e9 / instr, ghz);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, %.2f GHz\n", p.gcnArchName, cus, ghz);
    uint32_t* d; hipMalloc(&d, 4096);
    run<0>("v_pk_min_u16", d, ghz, cus); run<1>("v_pk_max_u16", d, ghz, cus); run<9>("v_pk_min_i16", d, ghz, cus); run<7>("v_pk_add_u16", d, ghz, cus);
    run<2>("v_min_u32", d, ghz, cus); run<3>("v_min_u16", d, ghz, cus); run<4>("v_min3_u32", d, ghz, cus); run<5>("v_med3_u32", d, ghz, cus); run<10>("v_max3_u16", d, ghz, cus);
    run<6>("v_alignbit_b32", d, ghz, cus); run<8>("v_perm_b32", d, ghz, cus);
    return 0;
}
