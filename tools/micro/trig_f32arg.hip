// trig_f32arg.hip -- sincos_f32arg and jl_mod_2pi_f32arg (csrc/env_device.h: sin / cos / mod 2 pi of a Float32 ARGUMENT evaluated in
// Float64; kernel polynomials up to pi/4, a two-term Cody-Waite reduction with a carried tail up to 2^16, ocml beyond) against the
// host libm on the GPU, for the Float32 values with |x| <= 2^16 -- `trig_f32arg.bin full` walks ALL 2 399 141 890 of them (3.6 minutes
// on a 32-thread box: the host's libm calls are the cost; done once, 0 / 0 / 0 differences, profiles/r04_pmc.md), the default
// takes every 13th bit pattern (17 s) --:
//   (float) sin, (float) cos must equal (float) sin((double) x), (float) cos((double) x) bit for bit (the oracle's roundings;
//   the sign of a zero result is not compared: sin(-0) is +0 here);  mod(x, 2 pi) must equal the fmod-based jl_mod bit for bit.
// The envs' parity with the oracle over this range is then a matter of enumeration, not of two libms' accuracy classes.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include tools/micro/trig_f32arg.hip -o tools/micro/trig_f32arg.bin -pthread
#include "../../reinforcementlearning.jl_amd/csrc/env_device.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

__global__ void eval(float* fs, float* fc, double* fm, uint32_t b0, uint32_t n, uint32_t sign, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float((b0 + i * stride) | sign);
    double s, c;
    rlhip::sincos_f32arg(x, &s, &c);
    fs[i] = (float)s;
    fc[i] = (float)c;
    fm[i] = rlhip::jl_mod_2pi_f32arg(x);
}
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static double jl_mod_host(double x, double y) {
    double r = fmod(x, y);
    if (r == 0) return copysign(r, y);
    if ((r > 0) != (y > 0)) return r + y;
    return r;
}

int main(int argc, char** argv) {
    const uint32_t top = 0x47800000u;  // 65536.0f
    const uint32_t stride = (argc > 1 && strcmp(argv[1], "full") == 0) ? 1u : 13u;
    const uint32_t CH = 1u << 25;
    float *ds, *dc, *hs = (float*)malloc(4ull * CH), *hcv = (float*)malloc(4ull * CH);
    double *dm, *hm = (double*)malloc(8ull * CH);
    HC(hipMalloc(&ds, 4ull * CH));
    HC(hipMalloc(&dc, 4ull * CH));
    HC(hipMalloc(&dm, 8ull * CH));
    const unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    long bad_s = 0, bad_c = 0, bad_m = 0, total = 0;
    const double two_pi = 2.0 * 3.14159265358979323846;
    for (uint32_t sign = 0; sign <= 0x80000000u; sign += 0x80000000u) {
        for (uint64_t b0 = 0; b0 <= top; b0 += (uint64_t)CH * stride) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(CH, ((uint64_t)top - b0) / stride + 1);
            hipLaunchKernelGGL(eval, dim3((n + 255) / 256), dim3(256), 0, 0, ds, dc, dm, (uint32_t)b0, n, sign, stride);
            HC(hipMemcpy(hs, ds, 4ull * n, hipMemcpyDeviceToHost));
            HC(hipMemcpy(hcv, dc, 4ull * n, hipMemcpyDeviceToHost));
            HC(hipMemcpy(hm, dm, 8ull * n, hipMemcpyDeviceToHost));
            std::vector<long> bs(nt, 0), bc(nt, 0), bm(nt, 0);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t]() {
                    for (uint32_t i = t; i < n; i += nt) {
                        const uint32_t bits = ((uint32_t)b0 + i * stride) | sign;
                        float x;
                        memcpy(&x, &bits, 4);
                        const float rs = (float)sin((double)x), rc = (float)cos((double)x);
                        const double rm = jl_mod_host((double)x, two_pi);
                        bs[t] += !(hs[i] == rs) || ((rs != 0.0f) && memcmp(&hs[i], &rs, 4) != 0);
                        bc[t] += memcmp(&hcv[i], &rc, 4) != 0;
                        bm[t] += memcmp(&hm[i], &rm, 8) != 0;
                    }
                });
            for (auto& t : th) t.join();
            for (unsigned t = 0; t < nt; ++t) bad_s += bs[t], bad_c += bc[t], bad_m += bm[t];
            total += n;
        }
        if (sign == 0x80000000u) break;
    }
    printf("%ld Float32 arguments with |x| <= 65536 (every %u-th bit pattern): sin roundings that differ from libm %ld, cos %ld, mod 2 pi %ld (%u host threads)\n",
           total, stride, bad_s, bad_c, bad_m, nt);
    return (bad_s == 0 && bad_c == 0 && bad_m == 0) ? 0 : 1;
}
