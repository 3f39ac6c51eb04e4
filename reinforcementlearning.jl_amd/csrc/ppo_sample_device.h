// ppo_sample_device.h -- per-lane action sampling from the actor head (shared by ppo.hip and ppo3.hip).
#pragma once
#include "common.h"
#include "mlp_device.h"
#include "select_device.h"

namespace rlhip {

constexpr float LOG2PI_F = 1.8378770664093453f;  // log(2f0 * pi) as Float32 (RLCore/utils/distributions.jl:9)

struct RegLogits {
    const float* l;
    __device__ __forceinline__ float operator()(int k) const { return l[k]; }
};

// normlogpdf(mu, sigma, x; eps = 1f-8)   RLCore/utils/distributions.jl:18-21
__device__ __forceinline__ float normlogpdf1(float mu, float sigma, float x) {
    float se = sigma + 1.0e-8f;
    float z = (x - mu) / se;
    return -(z * z + LOG2PI_F) / 2.0f - logf(se);
}

// Box-Muller standard normal from the NORMAL stream, evaluated in Float64 and rounded once (so that
// CPU libm and GPU ocml agree); the `randn(rng, Float32, ...)` stand-in of networks.jl:70.
__device__ __forceinline__ float normal_draw(uint64_t seed, uint32_t id, uint32_t step, int k) {
    u32x4 w = philox4x32_10(seed, id, (uint32_t)(k >> 1), step, TAG_NORMAL);
    double u1 = (double)((w.x >> 8) + 1u) * 0x1p-24;
    double u2 = (double)(w.y >> 8) * 0x1p-24;
    double r = ::sqrt(-2.0 * ::log(u1));
    double a = 6.283185307179586 * u2;
    return (float)((k & 1) ? r * ::sin(a) : r * ::cos(a));
}

// The random part of policy_sample for (id, step): nothing in it depends on the actor output, so the rollout kernels
// evaluate it ahead of the sequential env loop, spread over the lanes of an env.
//   discrete:   noise[k] = Gumbel noise of action k (k < na <= MAXO)
//   continuous: noise[0] = the standard normal draw (as a double holding the Float32 value)
__device__ __forceinline__ void policy_noise(int cont, int na, uint64_t seed, uint32_t id, uint32_t step,
                                             double noise[MAXO]) {
    if (!cont) gumbel_noise(na, seed, id, step, noise);
    else noise[0] = (double)normal_draw(seed, id, step, 0);
}

struct NoiseRegs {
    const double* g;
    __device__ __forceinline__ double operator()(int k) const { return g[k]; }
};

// Sample an action from the actor head output `oa` given the pre-drawn noise.
//   discrete:   oa = logits (na);  Gumbel-max, logp = logsoftmax(logits)[a]
//   continuous: oa = (mu, log sigma) for a 1-D action;  z = mu + exp(log sigma) * noise,
//               logp = normlogpdf(mu, sigma, z)   (GaussianNetwork, networks.jl:64-82, squash = identity)
__device__ __forceinline__ void policy_select(int cont, int na, const float oa[MAXO], const double noise[MAXO],
                                              int32_t& ai, float& af, float& logp) {
    if (!cont) {
        ai = categorical_select1(RegLogits{oa}, NoMask{}, na, NoiseRegs{noise}, &logp);
        af = 0.0f;
    } else {
        float mu = oa[0], sg = expf(oa[1]);
        float z = mu + sg * (float)noise[0];
        logp = normlogpdf1(mu, sg, z);
        af = z;
        ai = 0;
    }
}

// noise + selection on the spot (single-step plan! kernels)
__device__ __forceinline__ void policy_sample(int cont, int na, const float oa[MAXO], uint64_t seed,
                                              uint32_t id, uint32_t step, int32_t& ai, float& af,
                                              float& logp) {
    double noise[MAXO];
    policy_noise(cont, na, seed, id, step, noise);
    policy_select(cont, na, oa, noise, ai, af, logp);
}

}  // namespace rlhip
