// log_sampling.hip -- log_f64_sampling (csrc/select_device.h: the Float64 log of the log-sum-exp, fdlibm-style reduction, < 1 ulp)
// against the host libm on the GPU:
//   (a) EVERY Float32 in [1, 64] (the log-sum-exp's argument for <= LOG_SAMPLING_MAX_NA = 64 actions, the range in which
//       categorical_select1 uses it; 6 x 2^23 + 1 arguments): (float) log_f64_sampling((double) x) must equal
//       (float) log((double) x) bit for bit -- the oracle's rounding, so the rollout's log-probabilities stay bit-identical to it;
//   (b) its accuracy class on 2^24 doubles in (0, 40): error against a long-double reference <= 1 ulp (informative: no caller
//       uses it outside (a)'s range).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/micro/log_sampling.hip -o tools/micro/log_sampling.bin
#include "../../reinforcementlearning.jl_amd/csrc/select_device.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void eval_f32(float* out, uint32_t b0, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)rlhip::log_f64_sampling((double)__uint_as_float(b0 + i));
}
__global__ void eval_f64(double* out, const double* in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rlhip::log_f64_sampling(in[i]);
}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main() {
    const uint32_t b0 = 0x3f800000u, n = 0x42800000u - 0x3f800000u + 1u;  // 1.0f .. 64.0f
    static_assert(rlhip::LOG_SAMPLING_MAX_NA == 64, "enumerate the range categorical_select1 uses");
    float *d, *h = (float*)malloc(sizeof(float) * n);
    HC(hipMalloc(&d, sizeof(float) * n));
    hipLaunchKernelGGL(eval_f32, dim3((n + 255) / 256), dim3(256), 0, 0, d, b0, n);
    HC(hipMemcpy(h, d, sizeof(float) * n, hipMemcpyDeviceToHost));
    long bad = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t b = b0 + i;
        float x;
        memcpy(&x, &b, 4);
        const float ref = (float)log((double)x);
        if (memcmp(&ref, &h[i], 4) != 0 && bad++ < 5) printf("x = %.9g: device %.9g, libm %.9g\n", x, h[i], ref);
    }
    printf("(a) %u floats in [1, 64]: %ld Float32 roundings differ from libm\n", n, bad);
    const uint32_t m = 1u << 24;
    double *hin = (double*)malloc(sizeof(double) * m), *hout = (double*)malloc(sizeof(double) * m), *din, *dout;
    uint64_t st = 88172645463325252ull;
    for (uint32_t i = 0; i < m; ++i) {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        double u = (double)((st >> 11) | 1) * 0x1p-53;
        hin[i] = (i & 1) ? -log(u) : u;
    }
    HC(hipMalloc(&din, sizeof(double) * m));
    HC(hipMalloc(&dout, sizeof(double) * m));
    HC(hipMemcpy(din, hin, sizeof(double) * m, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_f64, dim3((m + 255) / 256), dim3(256), 0, 0, dout, din, m);
    HC(hipMemcpy(hout, dout, sizeof(double) * m, hipMemcpyDeviceToHost));
    double maxulp = 0.0;
    long differ = 0;
    for (uint32_t i = 0; i < m; ++i) {
        const double c = log(hin[i]);
        const long double t = logl((long double)hin[i]);
        const double ulp = nextafter(fabs(c), INFINITY) - fabs(c);
        const double err = fabs((double)(((long double)hout[i] - t) / (long double)ulp));
        if (err > maxulp) maxulp = err;
        differ += hout[i] != c;
    }
    printf("(b) %u doubles: max error %.3f ulp against long double; %ld differ from libm in the last place\n", m, maxulp, differ);
    return (bad == 0 && maxulp <= 1.0) ? 0 : 1;
}
