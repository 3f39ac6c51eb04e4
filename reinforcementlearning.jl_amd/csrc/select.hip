// select.hip -- integer action selection for n envs per launch (one lane per env).
//
// Replaces:
//   plan!(::EpsilonGreedyExplorer, values[, mask])  RLCore/policies/explorers/epsilon_greedy_explorer.jl:102-131
//   plan!(::GreedyExplorer, ...)                    :200-205   (eps = 0)
//   findmax / findmax_masked / find_all_max         RLCore/utils/basic.jl:91-120
//   sample_categorical (Gumbel-max)                 RLCore/utils/networks.jl:425-432, masking :466-468
// which the reference calls once per env per step from `_run` (RLCore/core/run.jl:57).
//
// Integer outputs are bit-exact against the oracle given the same values and the same Philox
// draws: comparisons are done exactly as the reference does them (first maximal index, NaN is
// maximal, masked entries = typemin), the eps test is `u >= eps` in Float64.
#include "common.h"
#include "select_device.h"

namespace rlhip {

__global__ __launch_bounds__(256) void eps_greedy_kernel(const float* __restrict__ values, int64_t na,
                                                         int64_t n, int64_t ks, int64_t is,
                                                         const uint8_t* __restrict__ mask, double eps,
                                                         int is_break_tie, uint64_t seed,
                                                         uint32_t env_id_base, uint32_t step,
                                                         int32_t* __restrict__ actions) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StridedValues q{values + i * is, ks};
    StridedMask mk{mask ? mask + i * is : nullptr, ks};
    actions[i] = eps_greedy_select1(q, mk, (int)na, eps, is_break_tie != 0, seed,
                                    env_id_base + (uint32_t)i, step);
}

__global__ __launch_bounds__(256) void categorical_kernel(const float* __restrict__ logits, int64_t na,
                                                          int64_t n, int64_t ks, int64_t is,
                                                          const uint8_t* __restrict__ mask,
                                                          uint64_t seed, uint32_t env_id_base,
                                                          uint32_t step, int32_t* __restrict__ actions,
                                                          float* __restrict__ logp_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StridedValues l{logits + i * is, ks};
    StridedMask mk{mask ? mask + i * is : nullptr, ks};
    float lp;
    actions[i] = categorical_sample1(l, mk, (int)na, seed, env_id_base + (uint32_t)i, step, &lp);
    if (logp_out) logp_out[i] = lp;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

double rlhip_get_eps(int32_t kind, double eps_stable, double eps_init, int64_t warmup_steps,
                     int64_t decay_steps, int64_t step) {
    // get_eps  epsilon_greedy_explorer.jl:69-88 (host scalar, Float64)
    if (kind == 0) {  // :linear
        if (step <= warmup_steps) return eps_init;
        if (step >= warmup_steps + decay_steps) return eps_stable;
        int64_t steps_left = warmup_steps + decay_steps - step;
        return eps_stable + (double)steps_left / (double)decay_steps * (eps_init - eps_stable);
    }
    if (step <= warmup_steps) return eps_init;  // :exp
    int64_t n = step - warmup_steps;
    double scale = eps_init - eps_stable;
    return eps_stable + scale * exp(-1.0 * (double)n / (double)decay_steps);
}

int32_t rlhip_eps_greedy_select_f32(const float* values, int64_t na, int64_t n, int64_t k_stride,
                                    int64_t i_stride, const uint8_t* mask, double eps,
                                    int32_t is_break_tie, uint64_t seed, uint32_t env_id_base,
                                    uint32_t step, int32_t* actions, rlhip_stream_t stream) {
    RLHIP_REQUIRE(values != nullptr && actions != nullptr, "NULL array");
    RLHIP_REQUIRE(na >= 1 && na <= 0x7FFFFFFF && n >= 0, "bad shape");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(eps_greedy_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       values, na, n, k_stride, i_stride, mask, eps, is_break_tie, seed, env_id_base,
                       step, actions);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_categorical_sample_f32(const float* logits, int64_t na, int64_t n, int64_t k_stride,
                                     int64_t i_stride, const uint8_t* mask, uint64_t seed,
                                     uint32_t env_id_base, uint32_t step, int32_t* actions,
                                     float* logp_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(logits != nullptr && actions != nullptr, "NULL array");
    RLHIP_REQUIRE(na >= 1 && na <= 0x7FFFFFFF && n >= 0, "bad shape");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(categorical_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       logits, na, n, k_stride, i_stride, mask, seed, env_id_base, step, actions,
                       logp_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
