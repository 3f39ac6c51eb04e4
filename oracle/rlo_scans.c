/* rlo_scans.c -- discount_rewards / discount_rewards_reduced / generalized_advantage_estimation.
 * TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 * Follows RLCore/utils/basic.jl:138-235 (discount_rewards), :237-319 (reduced), :334-417 (GAE).
 * Pinned on the reference's own known-answer tests RLCore/test/utils/base.jl:22-152
 * (tests/golden/scans.json).
 *
 * `dims` semantics (:155,169,184,202,247,365,385): the matrix drivers remap
 * dims = ndims - dims + 1 and iterate eachslice(…, dims = remapped), so the user's `dims` is the
 * axis the scan RUNS ALONG: dims = 1 scans down each column, dims = 2 scans along each row.
 * `x * false` in Julia is a strong zero (copysign(0, x), also for NaN/Inf).
 */
#include "rl_oracle.h"
#include <math.h>

#define DEFINE_SCANS(T, SFX, COPYSIGN)                                                              \
    /* _discount_rewards!(new_rewards, rewards, gamma, terminal, init)  :227-235 */                 \
    static T discount_vec_##SFX(T* out, const T* r, int64_t len, int64_t stride, T gamma,           \
                                const uint8_t* term, T init) {                                      \
        T gain = init;                                                                              \
        for (int64_t i = len - 1; i >= 0; --i) {                                                    \
            int is_continue = term ? !term[i * stride] : 1; /* :230 */                              \
            T gg = gamma * gain;                                                                    \
            T carried = is_continue ? gg : COPYSIGN((T)0, gg);                                      \
            gain = r[i * stride] + carried; /* :231 */                                              \
            if (out) out[i * stride] = gain; /* :232 */                                             \
        }                                                                                           \
        return gain;                                                                                \
    }                                                                                               \
    /* _generalized_advantage_estimation!(adv, r, v, gamma, lambda, terminal)  :408-417 */          \
    static void gae_vec_##SFX(T* adv, const T* r, const T* v, int64_t len, int64_t stride,          \
                              T gamma, T lambda, const uint8_t* term) {                             \
        T gae = (T)0; /* :409 */                                                                    \
        for (int64_t i = len - 1; i >= 0; --i) {                                                    \
            int is_continue = term ? !term[i * stride] : 1; /* :411 */                              \
            T gv = gamma * v[(i + 1) * stride];                                                     \
            T boot = is_continue ? gv : COPYSIGN((T)0, gv);                                         \
            T delta = r[i * stride] + boot - v[i * stride]; /* :412 */                              \
            T gl = gamma * lambda;                                                                  \
            T glc = is_continue ? gl : COPYSIGN((T)0, gl);                                          \
            gae = delta + glc * gae; /* :413 */                                                     \
            adv[i * stride] = gae;   /* :414 */                                                     \
        }                                                                                           \
    }                                                                                               \
    int rlo_discount_rewards_##SFX(T* out, const T* r, int64_t n1, int64_t n2, T gamma,             \
                                   const uint8_t* terminal, const T* init, int dims) {              \
        if (dims == 0) { /* vector: dims = Colon  :214-221; init is a scalar (or NULL -> zero) */   \
            if (n2 != 1) return -1; /* MethodError for a matrix without dims (test :45) */          \
            discount_vec_##SFX(out, r, n1, 1, gamma, terminal, init ? init[0] : (T)0);              \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 1) { /* slices = columns, scan down dim 1 */                                    \
            for (int64_t j = 0; j < n2; ++j)                                                        \
                discount_vec_##SFX(out + j * n1, r + j * n1, n1, 1, gamma,                          \
                                   terminal ? terminal + j * n1 : 0, init ? init[j] : (T)0);        \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 2) { /* slices = rows, scan along dim 2 (stride n1) */                          \
            for (int64_t i = 0; i < n1; ++i)                                                        \
                discount_vec_##SFX(out + i, r + i, n2, n1, gamma, terminal ? terminal + i : 0,      \
                                   init ? init[i] : (T)0);                                          \
            return 0;                                                                               \
        }                                                                                           \
        return -1;                                                                                  \
    }                                                                                               \
    /* discount_rewards_reduced  :237-263 (vector), :240-251 + :274-319 (matrix) */                 \
    int rlo_discount_rewards_reduced_##SFX(T* out, const T* r, int64_t n1, int64_t n2, T gamma,     \
                                           const uint8_t* terminal, const T* init, int dims) {      \
        if (dims == 0) {                                                                            \
            if (n2 != 1) return -1;                                                                 \
            out[0] = discount_vec_##SFX(0, r, n1, 1, gamma, terminal, init ? init[0] : (T)0);       \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 1) {                                                                            \
            for (int64_t j = 0; j < n2; ++j)                                                        \
                out[j] = discount_vec_##SFX(0, r + j * n1, n1, 1, gamma,                            \
                                            terminal ? terminal + j * n1 : 0,                       \
                                            init ? init[j] : (T)0);                                 \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 2) {                                                                            \
            for (int64_t i = 0; i < n1; ++i)                                                        \
                out[i] = discount_vec_##SFX(0, r + i, n2, n1, gamma, terminal ? terminal + i : 0,   \
                                            init ? init[i] : (T)0);                                 \
            return 0;                                                                               \
        }                                                                                           \
        return -1;                                                                                  \
    }                                                                                               \
    /* generalized_advantage_estimation  :334-404.  values: (n1+1) x n2 for dims = 1,               \
     * n1 x (n2+1) for dims = 2, length n1+1 for a vector. */                                       \
    int rlo_gae_##SFX(T* adv, const T* r, const T* v, int64_t n1, int64_t n2, T gamma, T lambda,    \
                      const uint8_t* terminal, int dims) {                                          \
        if (dims == 0) {                                                                            \
            if (n2 != 1) return -1; /* MethodError (test :129) */                                   \
            gae_vec_##SFX(adv, r, v, n1, 1, gamma, lambda, terminal);                               \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 1) {                                                                            \
            for (int64_t j = 0; j < n2; ++j)                                                        \
                gae_vec_##SFX(adv + j * n1, r + j * n1, v + j * (n1 + 1), n1, 1, gamma, lambda,     \
                              terminal ? terminal + j * n1 : 0);                                    \
            return 0;                                                                               \
        }                                                                                           \
        if (dims == 2) {                                                                            \
            for (int64_t i = 0; i < n1; ++i)                                                        \
                gae_vec_##SFX(adv + i, r + i, v + i, n2, n1, gamma, lambda,                         \
                              terminal ? terminal + i : 0);                                         \
            return 0;                                                                               \
        }                                                                                           \
        return -1;                                                                                  \
    }

DEFINE_SCANS(double, f64, copysign)
DEFINE_SCANS(float, f32, copysignf)
