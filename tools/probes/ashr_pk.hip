// Probe for the v_ashr_pk_u8_i32 trap (GPU call r04f): four int32 values per lane are shifted right by 15, clamped to 0..255 and packed into
// one dword, once as plain C (hipcc of ROCm 7.2 folds the first two into v_ashr_pk_u8_i32 and ORs the other two into the result) and once with the shifted
// value kept opaque (no fold).  Prints how many of the lanes' packed dwords differ from the host's.  Written after the round's GPU budget: run it first next round
//   hipcc -O3 --offload-arch=gfx950 ashr_pk.hip -o /tmp/ashr_pk && /tmp/ashr_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned clampByte(int v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); }

template <bool OPAQUE>
__global__ void k_pack(const int4* __restrict__ in, unsigned* __restrict__ out, int n, unsigned* __restrict__ junk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned acc = junk[i & 63];                      // something in the destination register's neighbourhood beforehand
    const int4 v = in[i];
    int a = v.x >> 15, b = v.y >> 15, c = v.z >> 15, d = v.w >> 15;
    if (OPAQUE) { asm volatile("" : "+v"(a)); asm volatile("" : "+v"(b)); asm volatile("" : "+v"(c)); asm volatile("" : "+v"(d)); }
    unsigned o = 0;
    o |= clampByte(a); o |= clampByte(b) << 8; o |= clampByte(c) << 16; o |= clampByte(d) << 24;
    out[i] = o; junk[(i & 63) + 64] = acc;
}

int main()
{
    const int n = 1 << 20;
    std::vector<int4> h(n); std::vector<unsigned> want(n), got(n), junkH(128, 0xffffffffu);
    srand(7);
    for (int i = 0; i < n; i++) {
        int* p = reinterpret_cast<int*>(&h[i]);
        for (int k = 0; k < 4; k++) p[k] = (int)((unsigned)rand() * 2654435761u) >> (rand() & 7);
        unsigned o = 0;
        for (int k = 0; k < 4; k++) { const int s = p[k] >> 15; o |= (unsigned)(s < 0 ? 0 : s > 255 ? 255 : s) << (8 * k); }
        want[i] = o;
    }
    int4* dIn; unsigned *dOut, *dJunk;
    if (hipMalloc(&dIn, n * sizeof(int4)) != hipSuccess || hipMalloc(&dOut, n * 4) != hipSuccess || hipMalloc(&dJunk, 128 * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemcpy(dIn, h.data(), n * sizeof(int4), hipMemcpyHostToDevice);
    for (int opaque = 0; opaque < 2; opaque++) {
        hipMemcpy(dJunk, junkH.data(), 128 * 4, hipMemcpyHostToDevice);
        if (opaque) hipLaunchKernelGGL(k_pack<true>, dim3(n / 256), dim3(256), 0, 0, dIn, dOut, n, dJunk);
        else        hipLaunchKernelGGL(k_pack<false>, dim3(n / 256), dim3(256), 0, 0, dIn, dOut, n, dJunk);
        hipMemcpy(got.data(), dOut, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, badHigh = 0;
        for (int i = 0; i < n; i++) if (got[i] != want[i]) { bad++; badHigh += (got[i] & 0xffffu) == (want[i] & 0xffffu); }
        printf("%s: %ld of %d packed dwords differ from the host (%ld of them only in the upper 16 bits)\n", opaque ? "shifted values kept opaque" : "plain C (folded by the compiler)", bad, n, badHigh);
    }
    return 0;
}
