// select_device.h -- per-lane action selection (device inline), shared by the stand-alone selection
// kernels (select.hip) and the fused plan / rollout kernels (ppo.hip, dqn.hip).
//
//   findmax(A)[2]: first maximal index, NaN is maximal (Base.findmax)
//   findmax_masked: masked-out entries become typemin(T) = -Inf   RLCore/utils/basic.jl:117-118
//   find_all_max + rand(rng, inds): tie-break variant             RLCore/utils/basic.jl:91-114,
//                                                                 epsilon_greedy_explorer.jl:102-106,118-123
//   eps-greedy draw order: u = rand(rng) first; `u >= eps ? greedy : rand(1:n)`   :108-112, :127-131
//   Gumbel-max categorical                                        RLCore/utils/networks.jl:425-432
#pragma once
#include "common.h"

namespace rlhip {

struct StridedValues {
    const float* p;
    int64_t ks;
    __device__ __forceinline__ float operator()(int k) const { return p[(int64_t)k * ks]; }
};
struct StridedMask {
    const uint8_t* p;
    int64_t ks;
    __device__ __forceinline__ bool has() const { return p != nullptr; }
    __device__ __forceinline__ bool operator()(int k) const { return p[(int64_t)k * ks] != 0; }
};
struct NoMask {
    __device__ __forceinline__ bool has() const { return false; }
    __device__ __forceinline__ bool operator()(int) const { return true; }
};

template <class V, class M>
__device__ __forceinline__ int findmax_first(const V& q, const M& mk, int na) {
    int best = 0;
    float bv = (mk.has() && !mk(0)) ? -INFINITY : q(0);
    for (int k = 1; k < na; ++k) {
        float x = (mk.has() && !mk(k)) ? -INFINITY : q(k);
        if (bv != bv) break;  // first NaN wins
        if (x != x || x > bv) {
            bv = x;
            best = k;
        }
    }
    return best;
}

// index (0-based) of the j-th (0-based) legal entry equal to the legal maximum
template <class V, class M>
__device__ __forceinline__ int tie_break_pick(const V& q, const M& mk, int na, uint32_t w) {
    bool have = false;
    float v = 0.f;
    for (int k = 0; k < na; ++k) {
        if (mk.has() && !mk(k)) continue;
        float x = q(k);
        if (!have) {
            v = x;
            have = true;
        } else if (v == v && (x != x || x > v)) {
            v = x;  // maximum propagates NaN
        }
    }
    int c = 0;
    for (int k = 0; k < na; ++k)
        if (!(mk.has() && !mk(k)) && q(k) == v) ++c;
    if (c == 0) return 0;
    int j = (int)randint32(w, (uint32_t)c);
    for (int k = 0; k < na; ++k)
        if (!(mk.has() && !mk(k)) && q(k) == v) {
            if (j == 0) return k;
            --j;
        }
    return 0;
}

template <class V, class M>
__device__ __forceinline__ int32_t eps_greedy_select1(const V& q, const M& mk, int na, double eps,
                                                      bool is_break_tie, uint64_t seed, uint32_t id,
                                                      uint32_t step) {
    u32x4 w = philox4x32_10(seed, id, 0, step, TAG_EXPLORE);
    double u = u01_f64(w.x, w.y);
    if (u >= eps) {  // greedy branch
        return is_break_tie ? tie_break_pick(q, mk, na, w.w) : findmax_first(q, mk, na);
    }
    if (mk.has()) {  // rand(rng, findall(mask))
        int c = 0;
        for (int k = 0; k < na; ++k) c += mk(k) ? 1 : 0;
        if (c == 0) return 0;
        int j = (int)randint32(w.z, (uint32_t)c);
        for (int k = 0; k < na; ++k)
            if (mk(k)) {
                if (j == 0) return k;
                --j;
            }
        return 0;
    }
    return (int32_t)randint32(w.z, (uint32_t)na);  // rand(rng, 1:n)
}

// log(x) in Float64 for the log-sum-exp of the sampling path, whose result is rounded to Float32 once: the classic argument
// reduction x = 2^k (1 + f), sqrt(1/2) < 1 + f <= sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the
// degree-7 minimax R of Sun's fdlibm e_log.c (coefficients Lg1..Lg7 and the ln2 split below are that file's; "freely
// distributable"), error < 1 ulp.  tools/micro/log_sampling.hip runs it on the GPU for EVERY Float32 in [1, 64] (the sum of <= 64
// exponentials whose largest is 1): the Float32 rounding equals the host libm's -- the oracle's -- in all 6 x 2^23 + 1 cases, so the
// log-probabilities are bit for bit what ocml's log gave.  ocml's log(double) is double-double arithmetic: ~85 Float64
// instructions against ~45 here, Float64 VALU ops run at half rate, and the log-sum-exp sits on the rollout's dependent chain
// (profiles/r04_rollout.md).  The Gumbel noise (compared as Float64, off the chain) keeps ocml's log.
__device__ __forceinline__ double log_f64_sampling(double x) {
    if (!(x >= 0x1p-1022 && x < __builtin_inf())) return ::log(x);  // zero / subnormal / inf / nan / negative: never on this path
    int k = __builtin_amdgcn_frexp_exp(x);       // x = m 2^k, 0.5 <= m < 1
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * ::fma(w, ::fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * ::fma(w, ::fma(w, ::fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return ::fma(dk, 6.93147180369123816490e-01, -((hfsq - ::fma(dk, 1.90821492927058770002e-10, s * (hfsq + R))) - f));
}

// Gumbel noise of action k of (id, step): -log(-log(u_k)), Float64; depends on nothing but the counters, so a rollout
// kernel may evaluate it for many steps ahead, spread over the lanes that cooperate on an env (ppo.hip)
__device__ __forceinline__ void gumbel_noise(int na, uint64_t seed, uint32_t id, uint32_t step, double* gn) {
    u32x4 w = {0, 0, 0, 0};
    for (int k = 0; k < na; ++k) {
        if ((k & 1) == 0) w = philox4x32_10(seed, id, (uint32_t)(k >> 1), step, TAG_GUMBEL);
        double u = (k & 1) ? u01_f64(w.z, w.w) : u01_f64(w.x, w.y);
        gn[k] = -::log(-::log(u));
    }
}

// logsoftmax (NNlib: x - max - log(sum(exp(x - max)))), Float32; Gumbel-max over log-probability + noise in Float64.
// GN: functor k -> the Gumbel noise of action k
// SHORT_LOG: the log of the log-sum-exp by log_f64_sampling -- bit-equal to the host libm's (the oracle's) after the Float32
// rounding for EVERY Float32 argument in [1, LOG_SAMPLING_MAX_NA] (tools/micro/log_sampling.hip enumerates them on the GPU), i.e.
// for up to LOG_SAMPLING_MAX_NA actions; callers with more actions take ocml's log (the claim was enumerated for
// <= 4 actions only while select.hip's entry points accept any `na`).
constexpr int LOG_SAMPLING_MAX_NA = 64;
template <bool SHORT_LOG = true, class V, class M, class GN>
__device__ __forceinline__ int32_t categorical_select1(const V& l, const M& mk, int na, const GN& gn, float* logp_out) {
    float mx = -INFINITY;
    for (int k = 0; k < na; ++k) {
        float x = (mk.has() && !mk(k)) ? -INFINITY : l(k);
        if (x > mx) mx = x;
    }
    float se = 0.f;
    if (na == 2 && !mk.has()) {
        // two actions: the term of the maximum is exp(0) = 1 exactly, so ONE exponential gives the same sum bit for bit
        // (0 + e + 1 and 0 + 1 + e are both RN(1 + e))
        const float x0 = l(0), x1 = l(1);
        const float other = (x1 > x0) ? x0 : x1;
        se = 1.0f + (float)::exp((double)(other - mx));
    } else {
        for (int k = 0; k < na; ++k) {
            float x = (mk.has() && !mk(k)) ? -INFINITY : l(k);
            se += (float)::exp((double)(x - mx));  // Float64 eval, rounded once (libm-independent)
        }
    }
    float lse = SHORT_LOG ? (float)log_f64_sampling((double)se) : (float)::log((double)se);
    int best = 0;
    double bg = 0.0;
    float blp = 0.f;
    for (int k = 0; k < na; ++k) {
        float x = (mk.has() && !mk(k)) ? -INFINITY : l(k);
        float lp = (x - mx) - lse;
        double g = gn(k) + (double)lp;
        if (k == 0 || g > bg) {
            bg = g;
            best = k;
            blp = lp;
        }
    }
    *logp_out = blp;
    return best;
}

struct GumbelInline {  // the noise evaluated on the spot, one Philox block per two actions
    uint64_t seed;
    uint32_t id, step;
    mutable u32x4 w;
    __device__ __forceinline__ double operator()(int k) const {
        if ((k & 1) == 0) w = philox4x32_10(seed, id, (uint32_t)(k >> 1), step, TAG_GUMBEL);
        const double u = (k & 1) ? u01_f64(w.z, w.w) : u01_f64(w.x, w.y);
        return -::log(-::log(u));
    }
};

template <class V, class M>
__device__ __forceinline__ int32_t categorical_sample1(const V& l, const M& mk, int na, uint64_t seed,
                                                       uint32_t id, uint32_t step, float* logp_out) {
    if (na <= LOG_SAMPLING_MAX_NA) return categorical_select1<true>(l, mk, na, GumbelInline{seed, id, step, u32x4{0, 0, 0, 0}}, logp_out);
    return categorical_select1<false>(l, mk, na, GumbelInline{seed, id, step, u32x4{0, 0, 0, 0}}, logp_out);
}

}  // namespace rlhip
