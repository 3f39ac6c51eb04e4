// gae_device.h -- the GAE recurrence of one env instance over a time-major trajectory, for fusing into the tail of
// the rollout kernels (the stand-alone scan is scans.hip; both follow RLCore/src/utils/basic.jl:408-417 operation for
// operation, so the fused result is bit-identical to rlhip_gae_returns_f32 on the same arrays).
#pragma once
#include "common.h"

namespace rlhip {

template <typename T>
__device__ __forceinline__ T strong_zero_mul(T x, bool keep) {
    return keep ? x : (T)copysign((T)0, x);  // Julia: x * false == copysign(0, x), also for NaN / Inf
}

// adv / ret / r / term: (T, n) time-major, v: (T + 1, n); one lane scans env `i` backwards in chunks of 16 steps whose
// loads are all issued before the first dependent add (the lane reads back what it wrote during the rollout: L2 hits)
__device__ __forceinline__ void gae_scan_lane(float* __restrict__ adv, float* __restrict__ ret,
                                              const float* __restrict__ r, const float* __restrict__ v,
                                              const uint8_t* __restrict__ term, int64_t n, int T, int64_t i, float gamma,
                                              float lambda) {
    constexpr int CH = 16;
    float gae = 0.0f;                    // :409
    float vnext = v[(int64_t)T * n + i];  // V[T+1]
    const float gl = gamma * lambda;
    for (int hi = T; hi > 0; hi -= CH) {
        const int cnt = hi < CH ? hi : CH;
        float r_[CH], v_[CH];
        uint8_t t_[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int64_t t = hi - 1 - c;
            if (c < cnt) {
                r_[c] = r[t * n + i];
                v_[c] = v[t * n + i];
                t_[c] = term[t * n + i];
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c < cnt) {
                const int64_t t = hi - 1 - c;
                const bool is_continue = !t_[c];                              // :411
                const float vi = v_[c];
                const float boot = strong_zero_mul(gamma * vnext, is_continue);
                const float delta = r_[c] + boot - vi;                        // :412
                const float glc = strong_zero_mul(gl, is_continue);
                gae = delta + glc * gae;                                      // :413
                adv[t * n + i] = gae;                                         // :414
                ret[t * n + i] = gae + vi;
                vnext = vi;
            }
        }
    }
}

}  // namespace rlhip
