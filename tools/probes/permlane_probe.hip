#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r[0]; p[threadIdx.x + 64] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    p[threadIdx.x + 128] = r2[0]; p[threadIdx.x + 192] = r2[1];
    unsigned d = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x128, 0xf, 0xC, false);
    p[threadIdx.x + 256] = d;
}
int main() {
    unsigned* d; hipMalloc(&d, 320 * 4);
    k<<<1, 64>>>(d);
    unsigned h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[5] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1", "dpp ror8 bank C"};
    for (int s = 0; s < 5; ++s) { printf("%s:", nm[s]); for (int i = 0; i < 64; ++i) printf(" %u", h[s * 64 + i]); printf("\n"); }
    return 0;
}
