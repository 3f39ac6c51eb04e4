// pk_fma_mfma_repro.hip -- ONE bounded attempt at a stand-alone reproducer of the round-1/2 miscomputes:
// SLP-packed `v_pk_fma_f32` whose multiplier is the HIGH half of a register pair freshly filled by an LDS load
// (op_sel / op_sel_hi = 1 on src0), issued between MFMAs of the same wave (profiles/attic/ppo3p_kernel.h was the only
// in-product reproducer).  Each lane accumulates  acc.xy = fma(w.y, x.xy, acc.xy)  over K steps, w = an 8-byte LDS read per
// step, with 0 .. 3 independent MFMAs between the LDS wait and the packed FMA and 0 / 1 / 2 waves per SIMD beside it; the
// reference is the same chain with two scalar v_fma_f32.  Any bit difference, or any run-to-run difference, is a hit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pk_fma_mfma_repro.bin pk_fma_mfma_repro.hip && ./pk_fma_mfma_repro.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NMFMA, bool PACKED>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w_in, const float* __restrict__ x_in, float* __restrict__ out, int K) {
    __shared__ f32x2 l_w[512 * 8];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512 * 8; i += 512) l_w[i] = f32x2{w_in[2 * i], w_in[2 * i + 1]};
    __syncthreads();
    f32x2 acc = {0.0f, 0.0f};
    f32x16 c[3];
    for (int j = 0; j < 3; ++j)
        for (int q = 0; q < 16; ++q) c[j][q] = 0.001f * q;
    float ma = 1.0f + tid * 1e-3f, mb = 0.5f;
    asm volatile("" : "+v"(ma), "+v"(mb));
    for (int kk = 0; kk < K; ++kk) {
        const f32x2 x = {x_in[(kk * 512 + tid) * 2 % 4096], x_in[((kk * 512 + tid) * 2 + 1) % 4096]};
        f32x2 w;
        const unsigned addr = (unsigned)(size_t)(&l_w[(tid * 8 + (kk & 7)) % (512 * 8)]);
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
#pragma unroll
        for (int j = 0; j < NMFMA; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, c[j], 0, 0, 0);
        if (PACKED) {
            // low = fma(w.hi, x.lo, acc.lo), high = fma(w.hi, x.hi, acc.hi): the operand form of the sightings
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x));
        } else {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc.x) : "v"(w.y), "v"(x.x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc.y) : "v"(w.y), "v"(x.y));
        }
    }
    float r = 0.f;
    for (int j = 0; j < 3; ++j) r += c[j][0] * 0.0f;  // keep the MFMAs alive without touching the result
    out[(blockIdx.x * 512 + tid) * 2] = acc.x + r;
    out[(blockIdx.x * 512 + tid) * 2 + 1] = acc.y + r;
}

template <int NMFMA>
static int check(const float* w, const float* x, float* o1, float* o2, float* o3, float* h1, float* h2, float* h3, int blocks, int K) {
    const size_t n = (size_t)blocks * 512 * 2;
    hipLaunchKernelGGL((k<NMFMA, false>), dim3(blocks), dim3(512), 0, 0, w, x, o1, K);
    hipLaunchKernelGGL((k<NMFMA, true>), dim3(blocks), dim3(512), 0, 0, w, x, o2, K);
    hipLaunchKernelGGL((k<NMFMA, true>), dim3(blocks), dim3(512), 0, 0, w, x, o3, K);
    hipMemcpy(h1, o1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2, o2, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h3, o3, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, flaky = 0;
    for (size_t i = 0; i < n; ++i) {
        bad += memcmp(&h1[i], &h2[i], 4) != 0;
        flaky += memcmp(&h2[i], &h3[i], 4) != 0;
    }
    printf("MFMAs between the LDS wait and the packed FMA: %d, %d workgroups of 8 waves, K = %d: packed != scalar in %zu of %zu values, "
           "packed run 1 != run 2 in %zu\n", NMFMA, blocks, K, bad, n, flaky);
    return bad || flaky;
}

int main() {
    float *w, *x, *o1, *o2, *o3;
    const int maxb = 512;
    hipMalloc(&w, 512 * 8 * 2 * 4);
    hipMalloc(&x, 4096 * 4);
    hipMalloc(&o1, maxb * 512 * 2 * 4);
    hipMalloc(&o2, maxb * 512 * 2 * 4);
    hipMalloc(&o3, maxb * 512 * 2 * 4);
    float* hw = new float[512 * 16];
    float* hx = new float[4096];
    for (int i = 0; i < 512 * 16; ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    for (int i = 0; i < 4096; ++i) hx[i] = (float)((i * 40503u) % 1999) / 999.0f - 1.0f;
    hipMemcpy(w, hw, 512 * 16 * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx, 4096 * 4, hipMemcpyHostToDevice);
    float *h1 = new float[maxb * 1024], *h2 = new float[maxb * 1024], *h3 = new float[maxb * 1024];
    int hits = 0;
    for (int blocks : {256, 512})  // 2 and 4 waves per SIMD (8-wave workgroups, 1 or 2 per CU)
        for (int K : {64, 1000}) {
            hits += check<0>(w, x, o1, o2, o3, h1, h2, h3, blocks, K);
            hits += check<1>(w, x, o1, o2, o3, h1, h2, h3, blocks, K);
            hits += check<2>(w, x, o1, o2, o3, h1, h2, h3, blocks, K);
            hits += check<3>(w, x, o1, o2, o3, h1, h2, h3, blocks, K);
        }
    printf(hits ? "REPRODUCED: %d configurations differ\n" : "not reproduced: %d configurations differ\n", hits);
    return 0;
}
