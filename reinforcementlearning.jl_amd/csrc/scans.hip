// scans.hip -- reverse-time scans: discount_rewards, discount_rewards_reduced, GAE (+ fused returns).
//
// Replaces RLCore/utils/basic.jl:138-235 (discount_rewards[!]), :237-319 (discount_rewards_reduced[!]),
// :334-417 (generalized_advantage_estimation[!]) whose matrix drivers loop `eachslice` serially over
// the envs (stride-N row views of a column-major matrix) and serially over time.
//
// Mapping: one lane per slice (env), time serial inside the lane -- the recurrence
//   gain_t = r_t + (gamma * gain_{t+1}) * c_t          (:231)
//   gae_t  = (r_t + (gamma * V_{t+1}) * c_t - V_t) + ((gamma * lambda) * c_t) * gae_{t+1}   (:412-413)
// is evaluated in exactly the reference's operation order (no FMA contraction, `x * false` is a
// strong zero), so Float32/Float64 results are bit-identical to a sequential CPU evaluation.
// With the PPO layout (rewards (n_env, T), dims = 2) consecutive lanes read consecutive addresses at
// every t: fully coalesced, 13 B (17 B with fused returns) of algorithmic traffic per (env, t).
// The loads do not depend on the recurrence, so the unrolled loop keeps several time steps of loads
// in flight per lane; the dependent chain is 3-4 VALU ops per step.
#include "gae_device.h"

namespace rlhip {

// out may be null (reduced form).  One lane per slice.
template <typename T, bool REDUCED>
__global__ __launch_bounds__(256) void discount_kernel(T* __restrict__ out, const T* __restrict__ r,
                                                       const uint8_t* __restrict__ term,
                                                       const T* __restrict__ init, int64_t n_slices,
                                                       int64_t len, int64_t elem_stride,
                                                       int64_t slice_stride, T gamma,
                                                       int init_per_slice) {
    int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_slices) return;
    const T* rp = r + sl * slice_stride;
    const uint8_t* tp = term ? term + sl * slice_stride : nullptr;
    T gain = init ? init[init_per_slice ? sl : 0] : (T)0;  // :225,:254 zero(eltype) default
#pragma unroll 4
    for (int64_t i = len - 1; i >= 0; --i) {
        bool is_continue = tp ? !tp[i * elem_stride] : true;  // :230
        T gg = gamma * gain;
        gain = rp[i * elem_stride] + strong_zero_mul(gg, is_continue);  // :231
        if (!REDUCED) out[sl * slice_stride + i * elem_stride] = gain;  // :232
    }
    if (REDUCED) out[sl] = gain;  // :262
}

// values: slice stride v_slice_stride (n1 + 1 for dims = 1), same element stride as rewards.
// The time axis is walked in register-staged chunks of CH steps: all loads of a chunk are issued
// back to back (they do not depend on the recurrence), then the chunk is scanned -- one memory
// latency per CH steps instead of one per step (the first version: 20 us for T = 32 on 64 waves).
// NT: non-temporal loads / stores, chosen by the host when the scan streams far more than the L2 holds (every element
// is touched once); the PPO hot path (0.5 MB trajectory, consumed by the gradient kernel next) keeps ordinary accesses.
template <typename T, bool WITH_RETURNS, int CH, bool NT>
__global__ __launch_bounds__(256) void gae_kernel(T* __restrict__ adv, T* __restrict__ ret,
                                                  const T* __restrict__ r, const T* __restrict__ v,
                                                  const uint8_t* __restrict__ term, int64_t n_slices,
                                                  int64_t len, int64_t elem_stride,
                                                  int64_t slice_stride, int64_t v_slice_stride,
                                                  T gamma, T lambda) {
    int64_t sl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_slices) return;
    const T* rp = r + sl * slice_stride;
    const T* vp = v + sl * v_slice_stride;
    const uint8_t* tp = term ? term + sl * slice_stride : nullptr;
    T gae = (T)0;                         // :409
    T vnext = vp[len * elem_stride];      // V[T+1]
    const T gl = gamma * lambda;
    for (int64_t hi = len; hi > 0; hi -= CH) {
        const int cnt = (int)((hi < CH) ? hi : CH);
        T r_[CH], v_[CH];
        uint8_t t_[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int64_t i = hi - 1 - c;
            if (c < cnt) {
                if (NT) {
                    r_[c] = __builtin_nontemporal_load(rp + i * elem_stride);
                    v_[c] = __builtin_nontemporal_load(vp + i * elem_stride);
                    t_[c] = tp ? __builtin_nontemporal_load(tp + i * elem_stride) : (uint8_t)0;
                } else {
                    r_[c] = rp[i * elem_stride];
                    v_[c] = vp[i * elem_stride];
                    t_[c] = tp ? tp[i * elem_stride] : (uint8_t)0;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c < cnt) {
                const int64_t i = hi - 1 - c;
                bool is_continue = !t_[c];                              // :411
                T vi = v_[c];
                T boot = strong_zero_mul(gamma * vnext, is_continue);
                T delta = r_[c] + boot - vi;                            // :412
                T glc = strong_zero_mul(gl, is_continue);
                gae = delta + glc * gae;                                // :413
                // (NT: non-temporal LOADS only -- write streams take ordinary stores, see gae_vec4_kernel)
                adv[sl * slice_stride + i * elem_stride] = gae;         // :414
                if (WITH_RETURNS) ret[sl * slice_stride + i * elem_stride] = gae + vi;
                vnext = vi;
            }
        }
    }
}

// Float32, env-major layout (dims = 2: element (env, t) at env + n t), streaming sizes: FOUR envs per lane, so every
// access is 16 bytes per lane (1 KB per wave instruction instead of 256 B) and the four recurrences interleave; the
// arithmetic of each env is the scalar kernel's, operation for operation.
template <bool WITH_RETURNS, int CH>
__global__ __launch_bounds__(256) void gae_vec4_kernel(float* __restrict__ adv, float* __restrict__ ret,
                                                       const float* __restrict__ r, const float* __restrict__ v,
                                                       const uint8_t* __restrict__ term, int64_t n, int64_t len,
                                                       float gamma, float lambda) {
    const int64_t sl = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (sl >= n) return;
    auto ld4 = [](const float* p) {
        nt_u32x4 u = nt_load16(p);
        return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
    };
    auto st4 = [](float* p, float a, float b, float c, float d) {
        // the outputs are pure write streams: ordinary stores (the L2 writes whole lines back) -- non-temporal stores
        // measured 106 us against 87 us per 2^20 x 32 scan (5.4 -> 6.6 TB/s; profiles/r03_store_policy.md)
        nt_u32x4 u = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
        *reinterpret_cast<nt_u32x4*>(p) = u;
    };
    float gae[4] = {0.f, 0.f, 0.f, 0.f};
    float vnext[4];
    {
        const float4 q = ld4(v + len * n + sl);
        vnext[0] = q.x, vnext[1] = q.y, vnext[2] = q.z, vnext[3] = q.w;
    }
    const float gl = gamma * lambda;
    for (int64_t hi = len; hi > 0; hi -= CH) {
        const int cnt = (int)((hi < CH) ? hi : CH);
        float4 r_[CH], v_[CH];
        uint32_t t_[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int64_t i = hi - 1 - c;
            if (c < cnt) {
                r_[c] = ld4(r + i * n + sl);
                v_[c] = ld4(v + i * n + sl);
                t_[c] = term ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(term + i * n + sl)) : 0u;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c < cnt) {
                const int64_t i = hi - 1 - c;
                const float rr[4] = {r_[c].x, r_[c].y, r_[c].z, r_[c].w};
                const float vv[4] = {v_[c].x, v_[c].y, v_[c].z, v_[c].w};
                float rt[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool is_continue = ((t_[c] >> (8 * e)) & 0xFFu) == 0u;   // :411
                    const float boot = strong_zero_mul(gamma * vnext[e], is_continue);
                    const float delta = rr[e] + boot - vv[e];                      // :412
                    const float glc = strong_zero_mul(gl, is_continue);
                    gae[e] = delta + glc * gae[e];                                 // :413
                    rt[e] = gae[e] + vv[e];
                    vnext[e] = vv[e];
                }
                st4(adv + i * n + sl, gae[0], gae[1], gae[2], gae[3]);            // :414
                if (WITH_RETURNS) st4(ret + i * n + sl, rt[0], rt[1], rt[2], rt[3]);
            }
        }
    }
}

struct ScanGeom {
    int64_t n_slices, len, elem_stride, slice_stride, v_slice_stride;
};

static int32_t scan_geom(int64_t n1, int64_t n2, int32_t dims, ScanGeom* g) {
    RLHIP_REQUIRE(n1 >= 0 && n2 >= 0, "negative size");
    if (dims == 0) {
        // a matrix without `dims` is a MethodError in the reference (test utils/base.jl:45,129)
        RLHIP_REQUIRE(n2 == 1, "matrix input requires dims = 1 or 2 (MethodError in the reference)");
        *g = {1, n1, 1, n1, n1 + 1};
        return RLHIP_OK;
    }
    if (dims == 1) {  // scan down each column
        *g = {n2, n1, 1, n1, n1 + 1};
        return RLHIP_OK;
    }
    if (dims == 2) {  // scan along each row (the coalesced PPO layout)
        *g = {n1, n2, n1, 1, 1};
        return RLHIP_OK;
    }
    set_error("dims must be 0 (vector), 1 or 2");
    return RLHIP_EINVAL;
}

template <typename T, bool REDUCED>
static int32_t discount_impl(T* out, const T* r, int64_t n1, int64_t n2, T gamma, const uint8_t* term,
                             const T* init, int32_t dims, hipStream_t s) {
    ScanGeom g;
    int32_t rc = scan_geom(n1, n2, dims, &g);
    if (rc) return rc;
    // empty inputs are a no-op (an empty device array has a NULL base pointer): checked before the NULL test
    if (g.n_slices == 0 || (g.len == 0 && !REDUCED)) return RLHIP_OK;
    RLHIP_REQUIRE(out != nullptr && r != nullptr, "NULL array");
    hipLaunchKernelGGL((discount_kernel<T, REDUCED>), dim3((int)((g.n_slices + 255) / 256)), dim3(256), 0,
                       s, out, r, term, init, g.n_slices, g.len, g.elem_stride, g.slice_stride, gamma,
                       dims == 0 ? 0 : 1);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

template <typename T>
static int32_t gae_impl(T* adv, T* ret, const T* r, const T* v, int64_t n1, int64_t n2, T gamma, T lambda,
                        const uint8_t* term, int32_t dims, hipStream_t s) {
    ScanGeom g;
    int32_t rc = scan_geom(n1, n2, dims, &g);
    if (rc) return rc;
    if (g.n_slices == 0 || g.len == 0) return RLHIP_OK;  // empty trajectory: nothing to write
    RLHIP_REQUIRE(adv != nullptr && r != nullptr && v != nullptr, "NULL array");
    dim3 grid((int)((g.n_slices + 255) / 256));
    const bool streaming = g.n_slices * g.len >= ((int64_t)1 << 22);
    if constexpr (sizeof(T) == 4) {
        const bool env_major = g.slice_stride == 1 && g.v_slice_stride == 1 && g.elem_stride == g.n_slices;
        const uintptr_t al = (uintptr_t)adv | (uintptr_t)r | (uintptr_t)v | (uintptr_t)(ret ? ret : adv);
        if (streaming && env_major && g.n_slices % 4 == 0 && (al & 15) == 0 && (!term || ((uintptr_t)term & 3) == 0)) {
            dim3 g4((int)((g.n_slices / 4 + 255) / 256));
            if (ret)
                hipLaunchKernelGGL((gae_vec4_kernel<true, 8>), g4, dim3(256), 0, s, (float*)adv, (float*)ret,
                                   (const float*)r, (const float*)v, term, g.n_slices, g.len, (float)gamma, (float)lambda);
            else
                hipLaunchKernelGGL((gae_vec4_kernel<false, 8>), g4, dim3(256), 0, s, (float*)adv, (float*)nullptr,
                                   (const float*)r, (const float*)v, term, g.n_slices, g.len, (float)gamma, (float)lambda);
            RLHIP_LAUNCH_CHECK();
            return RLHIP_OK;
        }
    }
#define LAUNCH_GAE(WR_, NT_)                                                                                          \
    hipLaunchKernelGGL((gae_kernel<T, WR_, 128 / sizeof(T), NT_>), grid, dim3(256), 0, s, adv, ret, r, v, term,       \
                       g.n_slices, g.len, g.elem_stride, g.slice_stride, g.v_slice_stride, gamma, lambda)
    if (ret) {
        if (streaming) LAUNCH_GAE(true, true);
        else LAUNCH_GAE(true, false);
    } else {
        if (streaming) LAUNCH_GAE(false, true);
        else LAUNCH_GAE(false, false);
    }
#undef LAUNCH_GAE
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_discount_rewards_f32(float* out, const float* r, int64_t n1, int64_t n2, float gamma,
                                   const uint8_t* terminal, const float* init, int32_t dims,
                                   rlhip_stream_t stream) {
    return discount_impl<float, false>(out, r, n1, n2, gamma, terminal, init, dims, as_stream(stream));
}
int32_t rlhip_discount_rewards_f64(double* out, const double* r, int64_t n1, int64_t n2, double gamma,
                                   const uint8_t* terminal, const double* init, int32_t dims,
                                   rlhip_stream_t stream) {
    return discount_impl<double, false>(out, r, n1, n2, gamma, terminal, init, dims, as_stream(stream));
}
int32_t rlhip_discount_rewards_reduced_f32(float* out, const float* r, int64_t n1, int64_t n2,
                                           float gamma, const uint8_t* terminal, const float* init,
                                           int32_t dims, rlhip_stream_t stream) {
    return discount_impl<float, true>(out, r, n1, n2, gamma, terminal, init, dims, as_stream(stream));
}
int32_t rlhip_discount_rewards_reduced_f64(double* out, const double* r, int64_t n1, int64_t n2,
                                           double gamma, const uint8_t* terminal, const double* init,
                                           int32_t dims, rlhip_stream_t stream) {
    return discount_impl<double, true>(out, r, n1, n2, gamma, terminal, init, dims, as_stream(stream));
}
int32_t rlhip_gae_f32(float* adv, const float* r, const float* v, int64_t n1, int64_t n2, float gamma,
                      float lambda, const uint8_t* terminal, int32_t dims, rlhip_stream_t stream) {
    return gae_impl<float>(adv, nullptr, r, v, n1, n2, gamma, lambda, terminal, dims, as_stream(stream));
}
int32_t rlhip_gae_f64(double* adv, const double* r, const double* v, int64_t n1, int64_t n2,
                      double gamma, double lambda, const uint8_t* terminal, int32_t dims,
                      rlhip_stream_t stream) {
    return gae_impl<double>(adv, nullptr, r, v, n1, n2, gamma, lambda, terminal, dims, as_stream(stream));
}
int32_t rlhip_gae_returns_f32(float* adv, float* ret, const float* r, const float* v,
                              const uint8_t* terminal, int64_t n_env, int64_t T, float gamma,
                              float lambda, rlhip_stream_t stream) {
    return gae_impl<float>(adv, ret, r, v, n_env, T, gamma, lambda, terminal, 2, as_stream(stream));
}

}  // extern "C"
