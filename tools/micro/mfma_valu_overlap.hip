// mfma_valu_overlap.hip -- does an f32 MFMA run BESIDE the f32 VALU on gfx950, or on it?
// 1024-thread workgroups, one per CU (4 waves per SIMD, like the PPO tile).  Per loop iteration a wave issues
//   V: 32 independent v_fma_f32 (16 accumulators x 2)        = 32 x 64 x 2   =   4 096 flop
//   M: 2 independent v_mfma_f32_32x32x2_f32                   = 2 x 4 096     =   8 192 flop
//   B: 2 independent v_mfma_f32_32x32x16_bf16                 = 2 x 32 768    =  65 536 flop
// Modes: V only | M only | V + M in every wave | V in two of a SIMD's four waves, M in the other two (waves w with (w >> 2) & 1: wave w sits on SIMD w % 4) | B only |
// V + B in every wave.
// If the matrix pipe ran beside the VALU of the same SIMD, "V + M" would take max(t_V, t_M); measured on MI355X (round 3):
//   V only 2.78 ms (121 TFLOP/s) | M only 4.32 ms (155 TFLOP/s) | V + M in every wave 8.44 ms = 1.19 x (t_V + t_M) |
//   2 V + 2 M waves per SIMD 4.25 ms (sum / 2 = 3.55, max / 2 = 2.16) | B only 2.28 ms (2 357 TFLOP/s) | V + B 4.97 ms = 0.98 x sum
// i.e. at four waves per SIMD a SIMD's VALU stream and its MFMA stream ADD, for f32 and for bf16 MFMAs alike, whether
// they come from the same wave or from different waves of the SIMD (profiles/r03_tile_mfma.md).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_valu_overlap.bin mfma_valu_overlap.hip && ./mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s) {
    const int w = threadIdx.x >> 6;
    const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && ((w >> 2) & 1) == 0) || MODE == 5;
    const bool do_b = MODE == 4 || MODE == 5;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) {
        ba[i] = (__bf16)(0.001f * (threadIdx.x & 7) + s);
        bb[i] = (__bf16)(0.002f * i);
    }
    const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && ((w >> 2) & 1) == 1);
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-6f + i;
    f32x16 c0, c1;
    for (int q = 0; q < 16; ++q) {
        c0[q] = q * 1e-3f;
        c1[q] = q * 2e-3f;
    }
    const float x = s, y = s * 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (do_v) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], x, y);
        }
        if (do_b) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ba, c1, 0, 0, 0);
        }
        if (do_m) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0);
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a[i] + c0[i] + c1[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int MODE>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    const float tv = run<0>(d, iters), tm = run<1>(d, iters), tb = run<2>(d, iters), tx = run<3>(d, iters);
    const float t16 = run<4>(d, iters), tv16 = run<5>(d, iters);
    const double fv = 256.0 * 16 * iters * 4096.0, fm = 256.0 * 16 * iters * 8192.0;
    printf("V only                 %8.3f ms  %6.1f TFLOP/s (v_fma_f32)\n", tv, fv / tv / 1e9);
    printf("M only                 %8.3f ms  %6.1f TFLOP/s (v_mfma_f32_32x32x2_f32)\n", tm, fm / tm / 1e9);
    printf("V + M in every wave    %8.3f ms  = %.2f x (t_V + t_M), %.2f x max(t_V, t_M)\n", tb, tb / (tv + tm), tb / (tv > tm ? tv : tm));
    printf("2 V + 2 M waves / SIMD  %8.3f ms  (half the work of each: t_V / 2 + t_M / 2 = %.3f, max = %.3f)\n", tx, (tv + tm) / 2,
           (tv > tm ? tv : tm) / 2);
    printf("B only                 %8.3f ms  %6.1f TFLOP/s (v_mfma_f32_32x32x16_bf16)\n", t16, 256.0 * 16 * iters * 65536.0 / t16 / 1e9);
    printf("V + B in every wave    %8.3f ms  = %.2f x (t_V + t_B), %.2f x max(t_V, t_B)\n", tv16, tv16 / (tv + t16),
           tv16 / (tv > t16 ? tv : t16));
    return 0;
}
