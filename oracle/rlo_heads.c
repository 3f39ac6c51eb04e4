/* rlo_heads.c -- CPU restatement of the stochastic Gaussian policy heads.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows RLCore/src/utils/networks.jl:
 *   GaussianNetwork      call with sampling :64-82, K samples per state :90-100, (state, action) :110-116,
 *                        logpdfcorrection (tanh) :39-40, inversesquash :41-42
 *   SoftGaussianNetwork  :147-165, K samples :173-182, (state, action) :192-198
 * and RLCore/src/utils/distributions.jl:18-21 (normlogpdf), :31-34 (diagnormlogpdf).  The heads' inputs are the
 * OUTPUTS of the mu / sigma sub-networks (d x n, column-major); `randn(rng, Float32, ...)` is the shared Philox NORMAL
 * stream: element (k, j) of env i at step t is draw k + d*j of (seed, env_id_base + i, t).  softplus is NNlib's
 * (un-vendored): log1p(exp(-abs(x))) + relu(x).  Reductions over the action dimension run in index order. */
#include "rl_oracle.h"
#include <math.h>

static float clampf(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
static float softplusf(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.0f ? x : 0.0f); }

static float head_noise(uint64_t seed, uint32_t id, uint32_t step, int64_t q) {
    uint32_t w[4];
    float z0, z1;
    rlo_philox4x32_10(seed, id, (uint32_t)(q / 2), step, RLO_TAG_NORMAL, w);
    rlo_normal_pair_f32(w[0], w[1], &z0, &z1);
    return (q & 1) ? z1 : z0;
}

/* log-probability of the pre-squash sample z (d values) under N(mu, sigma) of state i */
static float head_logp(const float* mu, const float* sg, const float* z, int64_t d, int squash, int soft) {
    const float eps = 1.0e-8f, log2pi = 1.8378770664093453f;
    if (soft) { /* :156 sum(normlogpdf(mu, sigma, z) .- (2f0 .* (log(2f0) .- z .- softplus.(-2f0 .* z))), dims = 1) */
        float acc = 0.0f;
        for (int64_t k = 0; k < d; ++k) {
            float nl = rlo_normlogpdf_f32(mu[k], sg[k], z[k]);
            float corr = 2.0f * ((0.6931472f - z[k]) - softplusf(-2.0f * z[k]));
            acc += nl - corr;
        }
        return acc;
    }
    float prod = 1.0f, sum = 0.0f, corr = 0.0f; /* :74 diagnormlogpdf(mu, sigma, z) .+ logpdfcorrection(z, squash) */
    for (int64_t k = 0; k < d; ++k) {
        float s = sg[k] + eps, v = s * s, dx = z[k] - mu[k];
        prod *= v;
        sum += (dx * dx) / v;
        if (squash) {
            float t = tanhf(z[k]);
            corr += logf(1.0f - t * t); /* :39 -sum(log.(1 .- tanh.(z).^2), dims = 1) */
        }
    }
    float lp = -0.5f * ((logf(prod) + sum) + (float)d * log2pi);
    return squash ? lp + (-corr) : lp;
}

int rlo_gaussian_head_sample_f32(const float* mu, const float* raw_sigma, int64_t d, int64_t n, int64_t K,
                                 float min_sigma, float max_sigma, int squash, int soft, uint64_t seed,
                                 uint32_t env_id_base, uint32_t step, float* action_out, float* logp_out) {
    if (d < 1 || d > 64 || K < 1) return -1;
    for (int64_t i = 0; i < n; ++i) {
        float sg[64], z[64];
        for (int64_t k = 0; k < d; ++k) sg[k] = clampf(raw_sigma[i * d + k], min_sigma, max_sigma);
        for (int64_t j = 0; j < K; ++j) {
            for (int64_t k = 0; k < d; ++k) {
                float noise = head_noise(seed, env_id_base + (uint32_t)i, step, k + d * j);
                z[k] = mu[i * d + k] + sg[k] * noise;
                action_out[(i * K + j) * d + k] = (squash || soft) ? tanhf(z[k]) : z[k];
            }
            if (logp_out) logp_out[i * K + j] = head_logp(mu + i * d, sg, z, d, squash, soft);
        }
    }
    return 0;
}

int rlo_gaussian_head_logp_f32(const float* mu, const float* raw_sigma, const float* action, int64_t d, int64_t n,
                               int64_t K, float min_sigma, float max_sigma, int squash, int soft, float* logp_out) {
    if (d < 1 || d > 64 || K < 1) return -1;
    for (int64_t i = 0; i < n; ++i) {
        float sg[64], z[64];
        for (int64_t k = 0; k < d; ++k) sg[k] = clampf(raw_sigma[i * d + k], min_sigma, max_sigma);
        for (int64_t j = 0; j < K; ++j) {
            for (int64_t k = 0; k < d; ++k) {
                float a = action[(i * K + j) * d + k];
                z[k] = (squash || soft) ? atanhf(a) : a; /* inversesquash :41-42, atanh.(action) :196 */
            }
            logp_out[i * K + j] = head_logp(mu + i * d, sg, z, d, squash, soft);
        }
    }
    return 0;
}
