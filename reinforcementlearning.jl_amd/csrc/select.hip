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

// prob(s::EpsilonGreedyExplorer, values[, mask])  epsilon_greedy_explorer.jl:141-194 for n envs: the Float64 probability vector
// the reference wraps in `Categorical(probs; check_args = false)` -- eps / n_legal on every legal action (0.0 on masked ones),
// plus (1 - eps) on findmax (first maximal index; :164, :192) or (1 - eps) / c on each of the c tied maxima (:145-147, :181-184).
// Same operation order as the oracle (one division, one addition per entry), hence the same bits.
__global__ __launch_bounds__(256) void eps_greedy_prob_kernel(const float* __restrict__ values, int64_t na, int64_t n,
                                                              int64_t ks, int64_t is, const uint8_t* __restrict__ mask,
                                                              double eps, int is_break_tie, double* __restrict__ probs) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StridedValues q{values + i * is, ks};
    StridedMask mk{mask ? mask + i * is : nullptr, ks};
    double* out = probs + i * is;
    int nlegal = 0;
    for (int k = 0; k < (int)na; ++k) nlegal += (!mk.has() || mk(k)) ? 1 : 0;
    const double base = eps / (double)nlegal;
    if (!is_break_tie) {
        const int best = findmax_first(q, mk, (int)na);
        for (int k = 0; k < (int)na; ++k) {
            double pk = (!mk.has() || mk(k)) ? base : 0.0;
            if (k == best) pk += 1 - eps;
            out[(int64_t)k * ks] = pk;
        }
        return;
    }
    bool have = false;  // find_all_max (basic.jl:91-114): the legal maximum, NaN propagating; ties by == (never true for NaN)
    float v = 0.f;
    for (int k = 0; k < (int)na; ++k) {
        if (mk.has() && !mk(k)) continue;
        const float x = q(k);
        if (!have) {
            v = x;
            have = true;
        } else if (v == v && (x != x || x > v)) {
            v = x;
        }
    }
    int c = 0;
    for (int k = 0; k < (int)na; ++k)
        if (!(mk.has() && !mk(k)) && q(k) == v) ++c;
    for (int k = 0; k < (int)na; ++k) {
        const bool legal = !mk.has() || mk(k);
        double pk = legal ? base : 0.0;
        if (legal && have && q(k) == v) pk += (1 - eps) / (double)c;
        out[(int64_t)k * ks] = pk;
    }
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

// (net::CategoricalNetwork)(state[, mask]; is_sampling, is_return_log_prob)  RLCore/src/utils/networks.jl:405-432 and the
// masked methods :459-472 in ONE launch on (na, n) component-major logits: masked logits = logits + ifelse(mask, 0, typemin)
// (:461); the Gumbel-max draw (sample_categorical :425-432) from them; z = Flux.onehotbatch(draw, 1:na).
__global__ __launch_bounds__(256) void categorical_network_kernel(const float* __restrict__ logits, int na, int64_t n,
                                                                  const uint8_t* __restrict__ mask, uint64_t seed,
                                                                  uint32_t env_id_base, uint32_t step,
                                                                  float* __restrict__ masked_out,
                                                                  int32_t* __restrict__ actions, float* __restrict__ onehot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (masked_out) {
        for (int k = 0; k < na; ++k) {
            const float l = logits[(int64_t)k * n + i];
            const bool m = mask == nullptr || mask[(int64_t)k * n + i] != 0;
            masked_out[(int64_t)k * n + i] = l + (m ? 0.0f : -__builtin_inff());
        }
    }
    if (actions) {
        StridedValues l{logits + i, n};
        StridedMask mk{mask ? mask + i : nullptr, n};
        float lp;
        const int a = categorical_sample1(l, mk, na, seed, env_id_base + (uint32_t)i, step, &lp);
        actions[i] = a;
        if (onehot)
            for (int k = 0; k < na; ++k) onehot[(int64_t)k * n + i] = k == a ? 1.0f : 0.0f;
    }
}

// ---- the remaining explorers, one lane per env (BatchExplorer: "apply the inner explorer to each column") ----
//   WeightedExplorer{is_normalized}   RLCore/src/policies/explorers/weighted_explorer.jl:19-33
//       sample(rng, Weights(values[, 1])): t = rand(rng) * sum; walk the cumulative weights while cw < t
//       (StatsBase.sample, un-vendored: published algorithm restated); mask: values[.!mask] .= 0
//   WeightedSoftmaxExplorer           .../weighted_softmax_explorer.jl:21-27: Weights(softmax(values), 1);
//       mask: values[.!mask] .= typemin(T)
//   GumbelSoftmaxExplorer             .../gumbel_softmax_explorer.jl:11-24:
//       argmax(logsoftmax(v) .- log.(-log.(rand(rng, T, n))))   all in T = Float32
// exp / log are evaluated in Float64 and rounded once (libm-independent, as everywhere in this library).
constexpr int EXPL_MAXNA = 64;

template <int KIND>
__global__ __launch_bounds__(256) void explorer_kernel(const float* __restrict__ values, int na, int64_t n, int64_t ks,
                                                       int64_t is, const uint8_t* __restrict__ mask,
                                                       int is_normalized, uint64_t seed, uint32_t env_id_base,
                                                       uint32_t step, int32_t* __restrict__ actions) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = values + i * is;
    const uint8_t* mk = mask ? mask + i * is : nullptr;
    const uint32_t id = env_id_base + (uint32_t)i;
    if (KIND == 0 || KIND == 1) {
        float w[EXPL_MAXNA];
        float sum = 1.0f;
        if (KIND == 0) {
            float acc = 0.0f;
            for (int k = 0; k < na; ++k) {
                w[k] = (mk && !mk[(int64_t)k * ks]) ? 0.0f : v[(int64_t)k * ks];
                acc += w[k];
            }
            if (!is_normalized) sum = acc;
        } else {
            float mx = -INFINITY;
            for (int k = 0; k < na; ++k) {
                w[k] = (mk && !mk[(int64_t)k * ks]) ? -INFINITY : v[(int64_t)k * ks];
                if (w[k] > mx) mx = w[k];
            }
            float se = 0.0f;
            for (int k = 0; k < na; ++k) {
                w[k] = (float)::exp((double)(w[k] - mx));
                se += w[k];
            }
            for (int k = 0; k < na; ++k) w[k] = w[k] / se;
        }
        u32x4 r = philox4x32_10(seed, id, 0, step, TAG_EXPLORE);
        double t = u01_f64(r.x, r.y) * (double)sum;
        int a = 0;
        float cw = w[0];
        while ((double)cw < t && a < na - 1) {
            ++a;
            cw += w[a];
        }
        actions[i] = a;
    } else {
        float mx = -INFINITY;
        for (int k = 0; k < na; ++k) {
            float x = (mk && !mk[(int64_t)k * ks]) ? -INFINITY : v[(int64_t)k * ks];
            if (x > mx) mx = x;
        }
        float se = 0.0f;
        for (int k = 0; k < na; ++k) {
            float x = (mk && !mk[(int64_t)k * ks]) ? -INFINITY : v[(int64_t)k * ks];
            se += (float)::exp((double)(x - mx));
        }
        float lse = (float)::log((double)se);
        int best = 0;
        float bg = 0.0f;
        u32x4 r = {0, 0, 0, 0};
        for (int k = 0; k < na; ++k) {
            if ((k & 3) == 0) r = philox4x32_10(seed, id, 0x8000u + (uint32_t)(k >> 2), step, TAG_GUMBEL);
            uint32_t word = (k & 3) == 0 ? r.x : (k & 3) == 1 ? r.y : (k & 3) == 2 ? r.z : r.w;
            float u = u01_f32(word);
            float x = (mk && !mk[(int64_t)k * ks]) ? -INFINITY : v[(int64_t)k * ks];
            float logit = (x - mx) - lse;
            float inner = (float)::log((double)u);
            float gum = (float)::log((double)(-inner));
            float g = logit - gum;
            if (k == 0 || g > bg || (g != g && bg == bg)) {  // argmax: first maximal index, NaN is maximal
                bg = g;
                best = k;
            }
        }
        actions[i] = best;
    }
}

// UCBExplorer  RLCore/src/policies/explorers/UCB_explorer.jl:24-30 (Float64): per env instance its own action counts
//   v, inds = find_all_max(values .+ c * sqrt(log(step + 1) / actioncounts)); a = rand(rng, inds); counts[a] += 1
__global__ __launch_bounds__(256) void ucb_kernel(const float* __restrict__ values, int na, int64_t n, int64_t ks,
                                                  int64_t is, double c, double* __restrict__ counts, int64_t step,
                                                  uint64_t seed, uint32_t env_id_base, int32_t* __restrict__ actions) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = values + i * is;
    const double lg = ::log((double)(step + 1));
    double best = 0.0;
    int cnt = 0;
    for (int k = 0; k < na; ++k) {
        double x = (double)v[(int64_t)k * ks] + c * ::sqrt(lg / counts[(int64_t)k * n + i]);
        if (k == 0 || (best == best && (x != x || x > best))) {
            best = x;
            cnt = 1;
        } else if (x == best) {
            ++cnt;
        }
    }
    // recount exactly as find_all_max does (entries equal to the maximum; NaN == NaN is false)
    cnt = 0;
    for (int k = 0; k < na; ++k) {
        double x = (double)v[(int64_t)k * ks] + c * ::sqrt(lg / counts[(int64_t)k * n + i]);
        if (x == best) ++cnt;
    }
    int a = 0;
    if (cnt > 0) {
        u32x4 r = philox4x32_10(seed, env_id_base + (uint32_t)i, 0, (uint32_t)step, TAG_EXPLORE);
        int j = (int)randint32(r.x, (uint32_t)cnt);
        for (int k = 0; k < na; ++k) {
            double x = (double)v[(int64_t)k * ks] + c * ::sqrt(lg / counts[(int64_t)k * n + i]);
            if (x == best) {
                if (j == 0) {
                    a = k;
                    break;
                }
                --j;
            }
        }
    }
    counts[(int64_t)a * n + i] += 1.0;
    actions[i] = a;
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

int32_t rlhip_eps_greedy_prob_f32(const float* values, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride,
                                  const uint8_t* mask, double eps, int32_t is_break_tie, double* probs,
                                  rlhip_stream_t stream) {
    RLHIP_REQUIRE(values != nullptr && probs != nullptr, "NULL array");
    RLHIP_REQUIRE(na >= 1 && na <= 0x7FFFFFFF && n >= 0, "bad shape");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(eps_greedy_prob_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), values, na, n,
                       k_stride, i_stride, mask, eps, is_break_tie, probs);
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

int32_t rlhip_categorical_network_f32(const float* logits, int64_t na, int64_t n, const uint8_t* mask, uint64_t seed,
                                      uint32_t env_id_base, uint32_t step, float* masked_logits_out, int32_t* actions,
                                      float* onehot_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(logits != nullptr && (masked_logits_out != nullptr || actions != nullptr), "NULL array");
    RLHIP_REQUIRE(onehot_out == nullptr || actions != nullptr, "the one-hot output needs the action output");
    RLHIP_REQUIRE(na >= 1 && na <= 0x7FFFFFFF && n >= 0, "bad shape");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(categorical_network_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), logits,
                       (int)na, n, mask, seed, env_id_base, step, masked_logits_out, actions, onehot_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_explorer_select_f32(int32_t kind, const float* values, int64_t na, int64_t n, int64_t k_stride,
                                  int64_t i_stride, const uint8_t* mask, int32_t is_normalized, uint64_t seed,
                                  uint32_t env_id_base, uint32_t step, int32_t* actions, rlhip_stream_t stream) {
    RLHIP_REQUIRE(values != nullptr && actions != nullptr, "NULL array");
    RLHIP_REQUIRE(kind >= 0 && kind <= 2, "kind: 0 weighted, 1 weighted-softmax, 2 gumbel-softmax");
    RLHIP_REQUIRE(na >= 1 && na <= EXPL_MAXNA && n >= 0, "bad shape (na <= 64)");
    if (n == 0) return RLHIP_OK;
    dim3 grid((int)((n + 255) / 256));
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        hipLaunchKernelGGL((explorer_kernel<0>), grid, dim3(256), 0, s, values, (int)na, n, k_stride, i_stride, mask,
                           is_normalized, seed, env_id_base, step, actions);
    else if (kind == 1)
        hipLaunchKernelGGL((explorer_kernel<1>), grid, dim3(256), 0, s, values, (int)na, n, k_stride, i_stride, mask,
                           is_normalized, seed, env_id_base, step, actions);
    else
        hipLaunchKernelGGL((explorer_kernel<2>), grid, dim3(256), 0, s, values, (int)na, n, k_stride, i_stride, mask,
                           is_normalized, seed, env_id_base, step, actions);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ucb_select_f32(const float* values, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride, double c,
                             double* action_counts, int64_t step, uint64_t seed, uint32_t env_id_base,
                             int32_t* actions, rlhip_stream_t stream) {
    RLHIP_REQUIRE(values && action_counts && actions, "NULL array");
    RLHIP_REQUIRE(na >= 1 && na <= 0x7FFFFFFF && n >= 0 && step >= 1, "bad shape / step");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(ucb_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), values, (int)na, n,
                       k_stride, i_stride, c, action_counts, step, seed, env_id_base, actions);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
