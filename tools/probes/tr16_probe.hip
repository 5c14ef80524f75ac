// Semantics probe for ds_read_b64_tr_b16 (gfx950): every lane supplies the address of 4 contiguous 16-bit elements; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out, int stride_elems) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const int l = threadIdx.x;
    // lane i of a 16-lane group: row (i >> 2) of a [4][16] block whose rows are stride_elems apart, columns (i & 3) * 4 .. + 3; groups 1024 elements apart
    const int off = (l >> 4) * 1024 + ((l & 15) >> 2) * stride_elems + (l & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
    for (int stride : {16, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            const int want = (l >> 4) * 1024 + j * stride + (l & 15);      // element (row j, column l & 15) of the group's block
            if (h[l * 4 + j] != want) ++bad;
        }
        printf("stride %d: lane 0: %d %d %d %d  lane 5: %d %d %d %d  lane 17: %d %d %d %d  mismatches vs (row j, col l&15): %d\n", stride,
               h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23], h[68], h[69], h[70], h[71], bad);
    }
    return 0;
}
