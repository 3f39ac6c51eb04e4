// tanh_sel.hip -- is the branch-free tanh of the two-layer PPO tile (csrc/ppo_grad_tile.h: tanh_sel) the SAME bits as the
// library's tanhf on gfx950?  Every float bit pattern is tried (2^32 inputs, 2^22 threads x 1024 each); NaNs compare as NaN.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ../../include -o tanh_sel.bin tanh_sel.hip && ./tanh_sel.bin
#include "../../reinforcementlearning.jl_amd/csrc/ppo_grad_tile.h"

#include <stdio.h>

__global__ void k(unsigned long long* bad, unsigned int* first_bad) {
    const unsigned int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0;
    for (unsigned int i = 0; i < 1024u; ++i) {
        const unsigned int bits = t * 1024u + i;
        const float x = __uint_as_float(bits);
        const float a = tanhf(x), b = rlhip::tanh_sel(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
        if (!same) {
            ++nb;
            atomicMin(first_bad, bits);
        }
    }
    if (nb) atomicAdd(bad, nb);
}

int main() {
    unsigned long long* bad;
    unsigned int* fb;
    hipMalloc(&bad, 8);
    hipMalloc(&fb, 4);
    hipMemset(bad, 0, 8);
    hipMemset(fb, 0xff, 4);
    hipLaunchKernelGGL(k, dim3(1 << 14), dim3(256), 0, 0, bad, fb);
    unsigned long long h = 0;
    unsigned int hf = 0;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, fb, 4, hipMemcpyDeviceToHost);
    printf("tanh_sel vs tanhf over all 2^32 bit patterns: %llu differ (first 0x%08x)\n", h, hf);
    return h != 0;
}
