// wave_simd_map.hip -- where do the 8 wavefronts of a 512-thread workgroup land?  rollout_split_kernel (csrc/ppo.hip) pairs
// wave w (actor wave) with wave w + 4 (critic wave) and wants each pair on ONE SIMD.  Prints, for a few workgroups, the SIMD id
// of every wave (HW_REG_HW_ID bits 5:4) and checks wave w / w + 4 share a SIMD in every workgroup of a full-chip launch.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_simd_map.hip -o tools/micro/wave_simd_map.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512, 1) void probe(int* out, int spin) {
    const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_ID[5:4]
    const int cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);    // HW_ID[11:8]
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = fmaf(x, 1.0001f, 0.5f);  // keep the workgroup resident while the others start
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = simd | (cu << 8) | ((x == 0.123f) << 30);
}

int main() {
    const int nb = 256;
    int* d;
    hipMalloc(&d, nb * 8 * sizeof(int));
    hipMemset(d, 0xff, nb * 8 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, d, 20000);
    hipDeviceSynchronize();
    int* h = (int*)malloc(nb * 8 * sizeof(int));
    hipMemcpy(h, d, nb * 8 * sizeof(int), hipMemcpyDeviceToHost);
    int paired = 0, per_simd_two = 0;
    for (int b = 0; b < nb; ++b) {
        int ok = 1, cnt[4] = {0, 0, 0, 0};
        for (int w = 0; w < 8; ++w) cnt[h[b * 8 + w] & 3]++;
        for (int w = 0; w < 4; ++w) ok &= ((h[b * 8 + w] & 3) == (h[b * 8 + w + 4] & 3));
        paired += ok;
        per_simd_two += (cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2);
        if (b < 4) {
            printf("workgroup %d simd of waves 0..7:", b);
            for (int w = 0; w < 8; ++w) printf(" %d", h[b * 8 + w] & 3);
            printf("\n");
        }
    }
    printf("workgroups with two waves on every SIMD: %d / %d;  with wave w and w + 4 on the same SIMD: %d / %d\n", per_simd_two, nb,
           paired, nb);
    return (paired == nb && per_simd_two == nb) ? 0 : 1;  // rollout_split_kernel stays correct either way; it gets slower
}
