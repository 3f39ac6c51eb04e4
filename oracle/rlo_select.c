/* rlo_select.c -- integer action selection: find_all_max, findmax, eps schedules, eps-greedy,
 * Gumbel-max categorical.  TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 * Pinned on RLCore/test/utils/base.jl:2-14 and
 * RLCore/test/policies/explorers/epsilon_greedy_explorer.jl:7-19,45-57,60-73
 * (tests/golden/select.json).
 */
#include "rl_oracle.h"
#include <math.h>
#include <stdlib.h>

/* find_all_max(x) / find_all_max(x, mask)  RLCore/utils/basic.jl:91-114 */
int64_t rlo_find_all_max_f64(const double* x, int64_t n, const uint8_t* mask, double* vmax,
                             int64_t* idx_out) {
    int have = 0;
    double v = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue; /* :112 maximum(view(x, mask)) */
        if (!have) {
            v = x[i];
            have = 1;
        } else if (!isnan(v) && (isnan(x[i]) || x[i] > v)) {
            v = x[i]; /* Julia's maximum propagates NaN */
        }
    }
    if (!have) return 0;
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        if (x[i] == v) idx_out[c++] = i; /* :96,:113 (NaN == NaN is false, as in Julia) */
    }
    if (vmax) *vmax = v;
    return c;
}

/* findmax(A)[2]: first index of the maximum; NaN counts as maximal (Base.findmax semantics,
 * "NaN is treated as greater than all other values").  findmax_masked (:117-118) replaces
 * masked-out entries by typemin(T) = -Inf before the search. */
#define DEFINE_FINDMAX(T, SFX)                                                \
    int64_t rlo_findmax_##SFX(const T* x, int64_t n, const uint8_t* mask) {   \
        int64_t best = 0;                                                     \
        T bv = (mask && !mask[0]) ? (T)-INFINITY : x[0];                      \
        for (int64_t i = 1; i < n; ++i) {                                     \
            T xi = (mask && !mask[i]) ? (T)-INFINITY : x[i];                  \
            if (isnan(bv)) break;                                             \
            if (isnan(xi) || xi > bv) {                                       \
                bv = xi;                                                      \
                best = i;                                                     \
            }                                                                 \
        }                                                                     \
        return best;                                                          \
    }
DEFINE_FINDMAX(double, f64)
DEFINE_FINDMAX(float, f32)

/* get_eps  RLCore/policies/explorers/epsilon_greedy_explorer.jl:69-88 (all Float64) */
double rlo_get_eps(int kind, double eps_stable, double eps_init, int64_t warmup_steps,
                   int64_t decay_steps, int64_t step) {
    if (kind == 0) { /* :linear  :69-78 */
        if (step <= warmup_steps) return eps_init;
        if (step >= warmup_steps + decay_steps) return eps_stable;
        int64_t steps_left = warmup_steps + decay_steps - step;
        return eps_stable + (double)steps_left / (double)decay_steps * (eps_init - eps_stable);
    }
    /* :exp  :80-88 */
    if (step <= warmup_steps) return eps_init;
    int64_t n = step - warmup_steps;
    double scale = eps_init - eps_stable;
    return eps_stable + scale * exp(-1.0 * (double)n / (double)decay_steps);
}

/* plan!(s::EpsilonGreedyExplorer, values[, mask])  :102-131, one column per env.
 * Draw order of the reference: u = rand(rng) first, then (only on the branch taken) an index draw.
 * Counter-based stand-in: one Philox block per (env, step): (w0,w1) -> u, w2 -> random branch index,
 * w3 -> tie-break index. */
int rlo_eps_greedy_select_f32(const float* values, int64_t na, int64_t n, const uint8_t* mask,
                              double eps, int is_break_tie, uint64_t seed, uint32_t env_id_base,
                              uint32_t step, int32_t* actions) {
    double* tmp = (double*)malloc(sizeof(double) * (size_t)na);
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)na);
    if (!tmp || !idx) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const float* q = values + i * na;
        const uint8_t* mk = mask ? mask + i * na : 0;
        uint32_t w[4];
        rlo_philox4x32_10(seed, env_id_base + (uint32_t)i, 0, step, RLO_TAG_EXPLORE, w);
        double u = rlo_u01_f64(w[0], w[1]);
        int32_t a;
        if (u >= eps) { /* greedy branch  :105,:111,:121,:130 */
            if (is_break_tie) {
                for (int64_t k = 0; k < na; ++k) tmp[k] = (double)q[k];
                int64_t c = rlo_find_all_max_f64(tmp, na, mk, 0, idx);
                a = (c > 0) ? (int32_t)idx[rlo_randint(w[3], (uint32_t)c)] : 0;
            } else {
                a = (int32_t)rlo_findmax_f32(q, na, mk);
            }
        } else { /* random branch: rand(rng, 1:n) or rand(rng, findall(mask)) */
            if (mk) {
                int64_t c = 0;
                for (int64_t k = 0; k < na; ++k)
                    if (mk[k]) idx[c++] = k;
                a = (c > 0) ? (int32_t)idx[rlo_randint(w[2], (uint32_t)c)] : 0;
            } else {
                a = (int32_t)rlo_randint(w[2], (uint32_t)na);
            }
        }
        actions[i] = a;
    }
    free(tmp);
    free(idx);
    return 0;
}

/* prob(s, values[, mask])  :141-194 */
int rlo_eps_greedy_prob_f64(const double* values, int64_t na, const uint8_t* mask, double eps,
                            int is_break_tie, double* probs) {
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)na);
    if (!idx) return -1;
    int64_t nlegal = 0;
    for (int64_t k = 0; k < na; ++k) nlegal += (!mask || mask[k]) ? 1 : 0;
    for (int64_t k = 0; k < na; ++k)
        probs[k] = (!mask || mask[k]) ? eps / (double)nlegal : 0.0; /* :143,:163,:179-180 */
    if (is_break_tie) {
        int64_t c = rlo_find_all_max_f64(values, na, mask, 0, idx);
        for (int64_t j = 0; j < c; ++j) probs[idx[j]] += (1 - eps) / (double)c; /* :145-147 */
    } else {
        probs[rlo_findmax_f64(values, na, mask)] += 1 - eps; /* :164,:192 */
    }
    free(idx);
    return 0;
}

/* sample_categorical  RLCore/utils/networks.jl:425-432 (mask: logits .+= ifelse(mask, 0, typemin) :466-468)
 *   log_probs = logsoftmax(logits, dims = 1)            (Float32, NNlib: x - max - log(sum(exp(x - max))))
 *   gumbels   = -log.(-log.(rand(rng, size...))) .+ log_probs   (rand -> Float64, so Float64)
 *   z         = argmax over dim 1 (first maximal index)
 */
int rlo_categorical_sample_f32(const float* logits, int64_t na, int64_t n, const uint8_t* mask,
                               uint64_t seed, uint32_t env_id_base, uint32_t step,
                               int32_t* actions, float* logp_out) {
    float* lp = (float*)malloc(sizeof(float) * (size_t)na);
    if (!lp) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const float* l = logits + i * na;
        const uint8_t* mk = mask ? mask + i * na : 0;
        float mx = -INFINITY;
        for (int64_t k = 0; k < na; ++k) {
            lp[k] = (mk && !mk[k]) ? -INFINITY : l[k];
            if (lp[k] > mx) mx = lp[k];
        }
        float se = 0.0f;
        /* exp / log evaluated in Float64 and rounded once: the (almost always) correctly rounded Float32
         * value, so that CPU libm and GPU ocml agree bit for bit */
        for (int64_t k = 0; k < na; ++k) se += (float)exp((double)(lp[k] - mx));
        float lse = (float)log((double)se);
        for (int64_t k = 0; k < na; ++k) lp[k] = (lp[k] - mx) - lse;
        int64_t best = 0;
        double bg = 0;
        for (int64_t k = 0; k < na; ++k) {
            uint32_t w[4];
            rlo_philox4x32_10(seed, env_id_base + (uint32_t)i, (uint32_t)(k / 2), step, RLO_TAG_GUMBEL, w);
            double u = (k & 1) ? rlo_u01_f64(w[2], w[3]) : rlo_u01_f64(w[0], w[1]);
            double g = -log(-log(u)) + (double)lp[k];
            if (k == 0 || g > bg) {
                bg = g;
                best = k;
            }
        }
        actions[i] = (int32_t)best;
        if (logp_out) logp_out[i] = lp[best];
    }
    free(lp);
    return 0;
}

/* ----------------------------------------------------------------- the remaining explorers --
 * One column of `values` per env instance (BatchExplorer, RLCore/src/policies/explorers/batch_explorer.jl:14-21).
 *   kind 0  WeightedExplorer{is_normalized}   weighted_explorer.jl:19-33: sample(rng, Weights(values[, 1]));
 *           masked: values[.!mask] .= 0.  StatsBase.sample(rng, wv) (un-vendored; published algorithm):
 *               t = rand(rng) * sum(wv); i = 1; cw = wv[1]; while cw < t && i < n: i += 1; cw += wv[i]
 *   kind 1  WeightedSoftmaxExplorer           weighted_softmax_explorer.jl:21-27: Weights(softmax(values), 1);
 *           masked: values[.!mask] .= typemin(T)
 *   kind 2  GumbelSoftmaxExplorer             gumbel_softmax_explorer.jl:11-24:
 *           argmax(logsoftmax(v) .- log.(-log.(rand(rng, T, n))))  in T = Float32
 * values (na x n) column-major: values[k + na * i].  exp / log in Float64, rounded once. */
int rlo_explorer_select_f32(int kind, const float* values, int64_t na, int64_t n, const uint8_t* mask,
                            int is_normalized, uint64_t seed, uint32_t env_id_base, uint32_t step,
                            int32_t* actions) {
    float* w = (float*)malloc(sizeof(float) * (size_t)na);
    if (!w) return -1;
    for (int64_t i = 0; i < n; ++i) {
        const float* v = values + i * na;
        const uint8_t* mk = mask ? mask + i * na : 0;
        uint32_t id = env_id_base + (uint32_t)i;
        if (kind == 0 || kind == 1) {
            float sum = 1.0f;
            if (kind == 0) {
                float acc = 0.0f;
                for (int64_t k = 0; k < na; ++k) {
                    w[k] = (mk && !mk[k]) ? 0.0f : v[k];
                    acc += w[k];
                }
                if (!is_normalized) sum = acc;
            } else {
                float mx = -INFINITY;
                for (int64_t k = 0; k < na; ++k) {
                    w[k] = (mk && !mk[k]) ? -INFINITY : v[k];
                    if (w[k] > mx) mx = w[k];
                }
                float se = 0.0f;
                for (int64_t k = 0; k < na; ++k) {
                    w[k] = (float)exp((double)(w[k] - mx));
                    se += w[k];
                }
                for (int64_t k = 0; k < na; ++k) w[k] = w[k] / se;
            }
            uint32_t r[4];
            rlo_philox4x32_10(seed, id, 0, step, RLO_TAG_EXPLORE, r);
            double t = rlo_u01_f64(r[0], r[1]) * (double)sum;
            int64_t a = 0;
            float cw = w[0];
            while ((double)cw < t && a < na - 1) {
                ++a;
                cw += w[a];
            }
            actions[i] = (int32_t)a;
        } else {
            float mx = -INFINITY;
            for (int64_t k = 0; k < na; ++k) {
                w[k] = (mk && !mk[k]) ? -INFINITY : v[k];
                if (w[k] > mx) mx = w[k];
            }
            float se = 0.0f;
            for (int64_t k = 0; k < na; ++k) se += (float)exp((double)(w[k] - mx));
            float lse = (float)log((double)se);
            int64_t best = 0;
            float bg = 0.0f;
            uint32_t r[4] = {0, 0, 0, 0};
            for (int64_t k = 0; k < na; ++k) {
                if ((k & 3) == 0) rlo_philox4x32_10(seed, id, 0x8000u + (uint32_t)(k >> 2), step, RLO_TAG_GUMBEL, r);
                float u = rlo_u01_f32(r[k & 3]);
                float logit = (w[k] - mx) - lse;
                float inner = (float)log((double)u);
                float gum = (float)log((double)(-inner));
                float g = logit - gum;
                if (k == 0 || g > bg || (isnan(g) && !isnan(bg))) {
                    bg = g;
                    best = k;
                }
            }
            actions[i] = (int32_t)best;
        }
    }
    free(w);
    return 0;
}

/* UCBExplorer  UCB_explorer.jl:24-30.  counts (na x n): counts[k * n + i] (one counter set per env instance),
 * initialised by the caller to eps = 1e-10 (:21-22); tie-break rand(rng, inds) = randint(word 0, #ties). */
int rlo_ucb_select_f32(const float* values, int64_t na, int64_t n, double c, double* counts, int64_t step,
                       uint64_t seed, uint32_t env_id_base, int32_t* actions) {
    double* x = (double*)malloc(sizeof(double) * (size_t)na);
    int64_t* inds = (int64_t*)malloc(sizeof(int64_t) * (size_t)na);
    if (!x || !inds) return -1;
    double lg = log((double)(step + 1));
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = 0; k < na; ++k) x[k] = (double)values[k + na * i] + c * sqrt(lg / counts[k * n + i]);
        double vmax;
        int64_t cnt = rlo_find_all_max_f64(x, na, 0, &vmax, inds);
        int64_t a = 0;
        if (cnt > 0) {
            uint32_t r[4];
            rlo_philox4x32_10(seed, env_id_base + (uint32_t)i, 0, (uint32_t)step, RLO_TAG_EXPLORE, r);
            a = inds[rlo_randint(r[0], (uint32_t)cnt)];
        }
        counts[a * n + i] += 1.0;
        actions[i] = (int32_t)a;
    }
    free(x);
    free(inds);
    return 0;
}
