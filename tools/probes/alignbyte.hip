// what does v_alignbyte_b32 do with a shift operand above 3 on gfx950, and does the SDWA byte write behave as warp8.h assumes?  (tools/gpu_call10.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out)
{
    const uint32_t hi = 0x77665544u, lo = 0x33221100u;
    for (int s = 0; s < 8; s++) { uint32_t sh = s + out[15]; out[s] = __builtin_amdgcn_alignbyte(hi, lo, sh); }
    uint32_t acc = 0xAABBCCDDu, v = (0x5Au << 10) | 0x3ffu | (3u << 18); const uint32_t ten = 10u;
    asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "s"(ten), "v"(v));
    out[8] = acc;
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64); k<<<1, 1>>>(d); uint32_t h[9]; (void)hipMemcpy(h, d, 36, hipMemcpyDeviceToHost);
    for (int s = 0; s < 8; s++) printf("alignbyte(0x77665544, 0x33221100, %d) = %08x\n", s, h[s]);
    printf("sdwa byte1 of AABBCCDD <- (v >> 10): %08x (v >> 10 = %x)\n", h[8], ((0x5Au << 10) | 0x3ffu | (3u << 18)) >> 10);
    return 0;
}
