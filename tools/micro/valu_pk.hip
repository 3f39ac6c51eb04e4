// valu_pk.hip -- micro-benchmark: f32 VALU throughput of v_fma_f32 vs v_pk_fma_f32 on gfx950 at 4 waves per SIMD
// (1024-thread workgroups, one per CU), the occupancy of the PPO learner tiles.  Same number of f32 FMAs in both kernels.
//   hipcc --offload-arch=gfx950 -O3 -o valu_pk valu_pk.hip && ./valu_pk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k_fma(float* out, int iters, float a, float b) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_pk(float* out, int iters, float a, float b) {
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a2), "v"(b2));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mixed: SGPR-pair operand (the weight pair) and a broadcast half (op_sel) as in the learner tile
__global__ __launch_bounds__(1024) void k_pk_sgpr(float* out, int iters, float a, float b) {
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 x2 = {a, b};
    unsigned long long w = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    w = __builtin_amdgcn_readfirstlane((unsigned)w) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(w >> 32)) << 32);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(w), "v"(x2));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static float time_ms(K k, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, iters, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, iters, 0.999f, 1e-3f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 20000;
    const double fmas = 256.0 * 1024 * 16 * iters;
    float t1 = time_ms(k_fma, out, iters), t2 = time_ms(k_pk, out, iters), t3 = time_ms(k_pk_sgpr, out, iters);
    printf("v_fma_f32              : %.3f ms  %.1f TFLOP/s\n", t1, 2 * fmas / (t1 * 1e-3) / 1e12);
    printf("v_pk_fma_f32           : %.3f ms  %.1f TFLOP/s\n", t2, 2 * fmas / (t2 * 1e-3) / 1e12);
    printf("v_pk_fma_f32 sgpr+opsel: %.3f ms  %.1f TFLOP/s\n", t3, 2 * fmas / (t3 * 1e-3) / 1e12);
    return 0;
}
