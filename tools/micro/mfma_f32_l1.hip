// mfma_f32_l1.hip -- is layer 1 (K = ns <= 4) on v_mfma_f32_32x32x2_f32 the SAME bits as the oracle's fmaf chain
//     z = b1;  z = fmaf(W1[u][0], x[0], z);  ...;  z = fmaf(W1[u][3], x[3], z)      (oracle/rlo_mlp3.c, mlp3_device.h)?
// One wave, a 32-sample x 32-unit block, two k-steps (k = 0, 1 then 2, 3), the bias as the accumulator's initial value.
// Operand images assumed (and checked here): A lane l = A[m = l & 31][k = l >> 5], B lane l = B[k = l >> 5][n = l & 31],
// D register q of lane l = D[m = (q & 3) + 8 (q >> 2) + 4 (l >> 5)][n = l & 31].
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_f32_l1.bin mfma_f32_l1.hip && ./mfma_f32_l1.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// form 0: D[sample][unit] (A = x, B = W1^T): lane = unit, registers = samples   (what the backward kernels need)
// form 1: D[unit][sample] (A = W1, B = x^T): lane = sample, registers = units   (what the forward kernels need)
__global__ void k(const float* x /*[4][32]*/, const float* W1 /*[32][4] unit-major*/, const float* b1, float* out0 /*[32 s][32 u]*/,
                  float* out1 /*[32 u][32 s]*/, int ns) {
    const int l = threadIdx.x, c = l & 31, kb = l >> 5;
    {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = b1[c];  // bias of this lane's unit (column)
        for (int ks = 0; ks < 2; ++ks) {
            const int kk = 2 * ks + kb;
            const float a = kk < ns ? x[kk * 32 + c] : 0.0f;       // A[m = sample c][k = kk]
            const float b = kk < ns ? W1[c * 4 + kk] : 0.0f;       // B[k = kk][n = unit c]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int q = 0; q < 16; ++q) out0[((q & 3) + 8 * (q >> 2) + 4 * kb) * 32 + c] = acc[q];
    }
    {
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = b1[(q & 3) + 8 * (q >> 2) + 4 * kb];  // bias of the register's unit (row)
        for (int ks = 0; ks < 2; ++ks) {
            const int kk = 2 * ks + kb;
            const float a = kk < ns ? W1[c * 4 + kk] : 0.0f;       // A[m = unit c][k = kk]
            const float b = kk < ns ? x[kk * 32 + c] : 0.0f;       // B[k = kk][n = sample c]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int q = 0; q < 16; ++q) out1[((q & 3) + 8 * (q >> 2) + 4 * kb) * 32 + c] = acc[q];
    }
}

int main() {
    float hx[128], hW[128], hb[32], o0[1024], o1[1024];
    float *dx, *dW, *db, *d0, *d1;
    hipMalloc(&dx, 512); hipMalloc(&dW, 512); hipMalloc(&db, 128); hipMalloc(&d0, 4096); hipMalloc(&d1, 4096);
    int bad_total = 0;
    for (int ns = 2; ns <= 4; ++ns)
        for (int trial = 0; trial < 200; ++trial) {
            srand(1000 * ns + trial);
            for (int i = 0; i < 128; ++i) {
                hx[i] = ((float)rand() / RAND_MAX - 0.5f) * (trial % 3 == 0 ? 40.0f : 2.0f);
                hW[i] = ((float)rand() / RAND_MAX - 0.5f) * 1.3f;
            }
            for (int i = 0; i < 32; ++i) hb[i] = ((float)rand() / RAND_MAX - 0.5f) * 0.7f;
            hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice); hipMemcpy(dW, hW, 512, hipMemcpyHostToDevice);
            hipMemcpy(db, hb, 128, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dW, db, d0, d1, ns);
            hipMemcpy(o0, d0, 4096, hipMemcpyDeviceToHost); hipMemcpy(o1, d1, 4096, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int s = 0; s < 32; ++s)
                for (int u = 0; u < 32; ++u) {
                    float z = hb[u];
                    for (int kk = 0; kk < ns; ++kk) z = fmaf(hW[u * 4 + kk], hx[kk * 32 + s], z);
                    if (memcmp(&z, &o0[s * 32 + u], 4) != 0 && !(z == 0.0f && o0[s * 32 + u] == 0.0f)) ++bad;
                    if (memcmp(&z, &o1[u * 32 + s], 4) != 0 && !(z == 0.0f && o1[u * 32 + s] == 0.0f)) ++bad;
                }
            bad_total += bad;
            if (bad && trial < 3) printf("ns %d trial %d: %d of 2048 differ\n", ns, trial, bad);
        }
    printf("mfma_f32_32x32x2 layer 1 vs fmaf chain: %d mismatching values over 3 x 200 trials x 2048 (0 = bit-identical, both forms)\n",
           bad_total);
    return bad_total != 0;
}
