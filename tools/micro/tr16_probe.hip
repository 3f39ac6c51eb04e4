// tr16_probe.hip -- what does ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) deliver?  LDS holds M[row][col] = row * 256 + col
// (pitch 72 elements); lane l of 16-lane group G = l >> 4, g = l & 15, passes the address of M[R0 + (g >> 2)][C0(G) + 4 (g & 3)] -- a
// [4 rows][16 cols] block per group -- and prints the four 16-bit values it receives as (row, col) pairs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tr16_probe.hip -o tools/micro/tr16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int P = 72;
__global__ void k(short* out) {
    __shared__ short lds[64 * P];
    for (int i = threadIdx.x; i < 64 * P; i += 64) lds[i] = (short)((i / P) * 256 + (i % P));
    __syncthreads();
    const int l = threadIdx.x, G = l >> 4, g = l & 15;
    const int row = 8 + (g >> 2), col = 16 * G + 4 * (g & 3);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + row * P + col));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d;
    hipMalloc((void**)&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d (group %d, g %2d; passed row %d col %2d):", l, l >> 4, l & 15, 8 + ((l & 15) >> 2), 16 * (l >> 4) + 4 * (l & 3));
        for (int j = 0; j < 4; ++j) printf("  (r%d,c%2d)", (unsigned short)h[l * 4 + j] / 256, (unsigned short)h[l * 4 + j] % 256);
        printf("\n");
    }
    return 0;
}
