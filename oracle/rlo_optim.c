/* rlo_optim.c -- parameter-update and loss primitives.  TEST INFRASTRUCTURE ONLY (rl_oracle.h).
 *
 * In-tree reference functions:
 *   Polyak / hard target sync      RLCore/policies/learners/target_network.jl:70-88
 *   clip_by_global_norm!           RLCore/utils/basic.jl:19-29
 *   normlogpdf / diagnormlogpdf    RLCore/utils/distributions.jl:9,18-21,31-34
 * Un-vendored third-party (formula restated from the published packages, PARITY UNPINNED,
 * cross-checked against torch on the CPU in tests/):
 *   Adam          Optimisers.jl (Flux 0.14-0.16 dependency; call site flux_approximator.jl:46)
 *   huber_loss    Flux.Losses (blog config a_practical_introduction_to_RL.jl/index.html:15134)
 *   TD target     removed Zoo DQNLearner (docs/src/rlcore.md:28)
 */
#include "rl_oracle.h"
#include <math.h>

/* dest .= rho .* dest .+ (1 - rho) .* src   target_network.jl:81-82 (Float32 throughout) */
void rlo_polyak_f32(float* dst, const float* src, int64_t n, float rho) {
    float om = 1.0f - rho;
    for (int64_t i = 0; i < n; ++i) dst[i] = rho * dst[i] + om * src[i];
}

/* tn.n_optimise += 1; if tn.n_optimise % tn.sync_freq == 0 ... tn.n_optimise = 0   :74-86 */
int rlo_target_sync_due(int64_t* n_optimise, int64_t sync_freq) {
    *n_optimise += 1;
    if (*n_optimise % sync_freq == 0) {
        *n_optimise = 0;
        return 1;
    }
    return 0;
}

/* global_norm = sqrt(sum of squares)  :19; if clip_norm <= gn: g .*= clip_norm / max(clip_norm, gn)  :23-26.
 * The reference sums in Float32 with Julia's pairwise mapreduce (order not reproducible bit for bit);
 * the oracle sums in Float64 and rounds once -- compare with tolerance. */
float rlo_clip_by_global_norm_f32(float* g, int64_t n, float clip_norm) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) acc += (double)g[i] * (double)g[i];
    float gn = (float)sqrt(acc);
    if (clip_norm <= gn) {
        float scale = clip_norm / fmaxf(clip_norm, gn);
        for (int64_t i = 0; i < n; ++i) g[i] *= scale;
    }
    return gn;
}

/* Optimisers.Adam(eta, (b1, b2), eps):  mt = b1*mt + (1-b1)*dx;  vt = b2*vt + (1-b2)*dx^2;
 * dx' = mt / (1 - b1^t) / (sqrt(vt / (1 - b2^t)) + eps) * eta;  x -= dx'.  The running powers b^t are
 * Float32 products carried in the optimiser state (bt = bt .* b after every step). */
void rlo_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t t) {
    float b1t = beta1, b2t = beta2;
    for (int64_t k = 1; k < t; ++k) {
        b1t *= beta1;
        b2t *= beta2;
    }
    float c1 = 1.0f - b1t, c2 = 1.0f - b2t;
    float om1 = 1.0f - beta1, om2 = 1.0f - beta2;
    for (int64_t i = 0; i < n; ++i) {
        float gi = g[i];
        float mi = beta1 * m[i] + om1 * gi;
        float vi = beta2 * v[i] + om2 * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        float d = mi / c1 / (sqrtf(vi / c2) + eps) * lr;
        p[i] = p[i] - d;
    }
}

/* const log2pi = log(2.0f0 * pi)   distributions.jl:9  (Float32) */
static inline float rlo_log2pi(void) { return logf(6.2831855f); }

/* normlogpdf(mu, sigma, x; eps = 1f-8)   distributions.jl:18-21 */
float rlo_normlogpdf_f32(float mu, float sigma, float x) {
    const float eps = 1.0e-8f;
    float z = (x - mu) / (sigma + eps);
    return -(z * z + rlo_log2pi()) / 2.0f - logf(sigma + eps);
}

/* diagnormlogpdf(mu, sigma, x; eps = 1f-8)   distributions.jl:31-34; arrays (d x n) column-major */
void rlo_diagnormlogpdf_f32(const float* mu, const float* sigma, const float* x, int64_t d,
                            int64_t n, float* out) {
    const float eps = 1.0e-8f;
    for (int64_t i = 0; i < n; ++i) {
        float prod = 1.0f, sum = 0.0f;
        for (int64_t k = 0; k < d; ++k) {
            float s = sigma[i * d + k] + eps;
            float v = s * s;
            float dx = x[i * d + k] - mu[i * d + k];
            prod *= v;
            sum += (dx * dx) / v;
        }
        out[i] = -0.5f * (logf(prod) + sum + (float)d * rlo_log2pi());
    }
}

/* Flux.Losses.huber_loss(q, target; delta) = mean(((e^2) * [e < delta]) * 0.5 + delta * (e - 0.5 * delta) * [e >= delta]),
 * e = |q - target|.  dq (optional) receives dL/dq. */
float rlo_huber_f32(const float* q, const float* target, int64_t n, float delta, float* dq) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float d = q[i] - target[i];
        float e = fabsf(d);
        float l = (e < delta) ? (e * e) * 0.5f : delta * (e - 0.5f * delta);
        acc += (double)l;
        if (dq) {
            float gi = (e < delta) ? d : (d > 0.0f ? delta : (d < 0.0f ? -delta : 0.0f));
            dq[i] = gi / (float)n;
        }
    }
    return (float)(acc / (double)n);
}

/* G = r + gamma * (1 - terminal) * max_a' Qt(s', a')   (Qt column-major na x n) */
void rlo_td_target_f32(const float* qt_next, int64_t na, int64_t n, const float* r,
                       const uint8_t* terminal, float gamma, float* target) {
    for (int64_t i = 0; i < n; ++i) {
        float mx = qt_next[i * na];
        for (int64_t k = 1; k < na; ++k)
            if (qt_next[i * na + k] > mx) mx = qt_next[i * na + k];
        float cont = terminal[i] ? 0.0f : 1.0f;
        target[i] = r[i] + gamma * cont * mx;
    }
}
