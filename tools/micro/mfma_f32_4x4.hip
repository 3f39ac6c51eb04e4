// mfma_f32_4x4.hip -- v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, K = 1) as phase 2 of the two-layer PPO
// tile wants to use it (csrc/ppo_grad_tile.h), lane = hidden unit over all 64 lanes:
//   z    D[b][i][j] = b1[unit 4b+j] + sum_k x[sample i][k] W1[unit 4b+j][k]     four chained instructions, k = 0..3
//   dW1  D[b][i][j] += x[sample s][k = i] dz[s][unit 4b+j]                         one instruction per sample, chained over s
// Operand images assumed (and checked here): A lane l = A[block l >> 2][i = l & 3], B lane l = B[block l >> 2][j = l & 3],
// D register i of lane l = D[block l >> 2][i][j = l & 3].  Checked: bit equality with the fmaf chains of the oracle.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_f32_4x4.bin mfma_f32_4x4.hip && ./mfma_f32_4x4.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// x [NSAMP][4], W1 [64][4], b1 [64], dz [NSAMP][64]; z_out [NSAMP][64], dw1_out [64][4]
template <int NSAMP>
__global__ void k(const float* x, const float* W1, const float* b1, const float* dz, float* z_out, float* dw1_out) {
    const int l = threadIdx.x, i = l & 3;
    float w1[4];
    for (int kk = 0; kk < 4; ++kk) w1[kk] = W1[4 * l + kk];
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < NSAMP; s0 += 4) {
        f32x4 z = {b1[l], b1[l], b1[l], b1[l]};
        for (int kk = 0; kk < 4; ++kk) z = __builtin_amdgcn_mfma_f32_4x4x1f32(x[4 * (s0 + i) + kk], w1[kk], z, 0, 0, 0);
        for (int ii = 0; ii < 4; ++ii) {
            z_out[(s0 + ii) * 64 + l] = z[ii];
            g = __builtin_amdgcn_mfma_f32_4x4x1f32(x[4 * (s0 + ii) + i], dz[(s0 + ii) * 64 + l], g, 0, 0, 0);
        }
    }
    for (int kk = 0; kk < 4; ++kk) dw1_out[4 * l + kk] = g[kk];
}

static float frand(unsigned* st) {
    *st = *st * 1664525u + 1013904223u;
    return ((int)(*st >> 8) - (1 << 23)) / (float)(1 << 22);
}

int main() {
    constexpr int NSAMP = 32;
    float hx[NSAMP * 4], hw[256], hb[64], hdz[NSAMP * 64], hz[NSAMP * 64], hg[256];
    float *dx, *dw, *db, *ddz, *dzo, *dg;
    hipMalloc(&dx, sizeof hx); hipMalloc(&dw, sizeof hw); hipMalloc(&db, sizeof hb); hipMalloc(&ddz, sizeof hdz);
    hipMalloc(&dzo, sizeof hz); hipMalloc(&dg, sizeof hg);
    unsigned st = 12345u;
    long bad_z = 0, bad_g = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const float sc = trial % 3 == 0 ? 1e-3f : trial % 3 == 1 ? 1.0f : 37.0f;
        for (float& v : hx) v = sc * frand(&st);
        for (float& v : hw) v = frand(&st);
        for (float& v : hb) v = 0.1f * frand(&st);
        for (float& v : hdz) v = (trial & 1) ? frand(&st) : (frand(&st) > 0 ? frand(&st) : 0.0f);
        hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice);
        hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(ddz, hdz, sizeof hdz, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k<NSAMP>, dim3(1), dim3(64), 0, 0, dx, dw, db, ddz, dzo, dg);
        hipMemcpy(hz, dzo, sizeof hz, hipMemcpyDeviceToHost); hipMemcpy(hg, dg, sizeof hg, hipMemcpyDeviceToHost);
        for (int s = 0; s < NSAMP; ++s)
            for (int u = 0; u < 64; ++u) {
                float z = hb[u];
                for (int kk = 0; kk < 4; ++kk) z = fmaf(hx[4 * s + kk], hw[4 * u + kk], z);
                bad_z += memcmp(&z, &hz[s * 64 + u], 4) != 0;
                ++total;
            }
        for (int u = 0; u < 64; ++u)
            for (int kk = 0; kk < 4; ++kk) {
                float g = 0.f;
                for (int s = 0; s < NSAMP; ++s) g = fmaf(hx[4 * s + kk], hdz[s * 64 + u], g);
                bad_g += memcmp(&g, &hg[4 * u + kk], 4) != 0;
            }
    }
    printf("v_mfma_f32_4x4x1_16b_f32: z chain %ld of %ld values differ from the fmaf chain; dW1 accumulation %ld of %d differ\n",
           bad_z, total, bad_g, 200 * 256);
    return (bad_z || bad_g) ? 1 : 0;
}
