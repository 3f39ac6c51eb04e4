// heads.hip -- stochastic Gaussian policy heads on the device.
//
// Reference: RLCore/src/utils/networks.jl -- GaussianNetwork (:64-116: sampling, K samples per state, (state, action)
// evaluation, logpdfcorrection for the tanh squash :39-42) and SoftGaussianNetwork (:147-198); normlogpdf /
// diagnormlogpdf of RLCore/src/utils/distributions.jl:18-34.  Inputs are the outputs of the mu / sigma sub-networks
// (d x n, column-major); `randn(rng, Float32, ...)` is the shared Philox NORMAL stream (element (k, j) of env i at
// step t = draw k + d*j of (seed, env_id_base + i, t)).  One lane per (sample j, state i): it walks the d action
// components in index order (the reductions over dims = 1), so a launch streams 2 d + (d + 1) K floats per state --
// HBM-bound elementwise work; consecutive lanes touch consecutive d-vectors.
#include "ppo_sample_device.h"

namespace rlhip {

constexpr int HEAD_MAX_D = 64;

__device__ __forceinline__ float clampj(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
// NNlib.softplus (un-vendored): log1p(exp(-abs(x))) + relu(x)
__device__ __forceinline__ float softplus_nnlib(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.0f ? x : 0.0f); }

// running log-probability of one pre-squash sample, fed one component at a time
template <int SOFT>
struct HeadLogp {
    float prod = 1.0f, sum = 0.0f, corr = 0.0f, acc = 0.0f;
    __device__ __forceinline__ void add(float mu, float sg, float z, int squash) {
        if (SOFT) {  // :156
            const float nl = normlogpdf1(mu, sg, z);
            const float c = 2.0f * ((0.6931472f - z) - softplus_nnlib(-2.0f * z));
            acc += nl - c;
        } else {  // :74 with distributions.jl:31-34 and networks.jl:39
            const float s = sg + 1.0e-8f, v = s * s, dx = z - mu;
            prod *= v;
            sum += (dx * dx) / v;
            if (squash) {
                const float t = tanhf(z);
                corr += logf(1.0f - t * t);
            }
        }
    }
    __device__ __forceinline__ float result(int d, int squash) const {
        if (SOFT) return acc;
        const float lp = -0.5f * ((logf(prod) + sum) + (float)d * LOG2PI_F);
        return squash ? lp + (-corr) : lp;
    }
};

template <int SOFT, int EVAL>
__global__ __launch_bounds__(256) void gaussian_head_kernel(const float* __restrict__ mu,
                                                            const float* __restrict__ raw_sigma,
                                                            const float* __restrict__ action_in, int d, int64_t n,
                                                            int K, float min_sigma, float max_sigma, int squash,
                                                            uint64_t seed, uint32_t env_id_base, uint32_t step,
                                                            float* __restrict__ action_out,
                                                            float* __restrict__ logp_out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * K) return;
    const int64_t i = gid / K;
    const int j = (int)(gid - i * K);
    const bool sq = SOFT || squash;
    HeadLogp<SOFT> lp;
    for (int k = 0; k < d; ++k) {
        const float m = mu[i * d + k];
        const float sg = clampj(raw_sigma[i * d + k], min_sigma, max_sigma);  // :67
        float z;
        if (EVAL) {
            const float a = action_in[gid * d + k];
            z = sq ? atanhf(a) : a;  // inversesquash :41-42 / atanh.(action) :196
        } else {
            z = m + sg * normal_draw(seed, env_id_base + (uint32_t)i, step, k + d * j);  // :69-71
            action_out[gid * d + k] = sq ? tanhf(z) : z;
        }
        if (logp_out) lp.add(m, sg, z, squash);
    }
    if (logp_out) logp_out[gid] = lp.result(d, squash);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_gaussian_head_sample_f32(const float* mu, const float* raw_sigma, int64_t d, int64_t n, int64_t K,
                                       float min_sigma, float max_sigma, int32_t squash, int32_t soft, uint64_t seed,
                                       uint32_t env_id_base, uint32_t step, float* action_out, float* logp_out,
                                       rlhip_stream_t stream) {
    RLHIP_REQUIRE(mu && raw_sigma && action_out, "NULL argument");
    RLHIP_REQUIRE(d >= 1 && d <= HEAD_MAX_D && n >= 0 && K >= 1 && n * K <= 0x7FFFFFFFll * 256, "bad shape");
    RLHIP_REQUIRE((squash == 0 || squash == 1) && (soft == 0 || soft == 1), "squash / soft must be 0 or 1");
    if (n == 0) return RLHIP_OK;
    const dim3 grid((unsigned)((n * K + 255) / 256)), block(256);
    if (soft)
        hipLaunchKernelGGL((gaussian_head_kernel<1, 0>), grid, block, 0, as_stream(stream), mu, raw_sigma, nullptr, (int)d,
                           n, (int)K, min_sigma, max_sigma, 1, seed, env_id_base, step, action_out, logp_out);
    else
        hipLaunchKernelGGL((gaussian_head_kernel<0, 0>), grid, block, 0, as_stream(stream), mu, raw_sigma, nullptr, (int)d,
                           n, (int)K, min_sigma, max_sigma, squash, seed, env_id_base, step, action_out, logp_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_gaussian_head_logp_f32(const float* mu, const float* raw_sigma, const float* action, int64_t d, int64_t n,
                                     int64_t K, float min_sigma, float max_sigma, int32_t squash, int32_t soft,
                                     float* logp_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(mu && raw_sigma && action && logp_out, "NULL argument");
    RLHIP_REQUIRE(d >= 1 && d <= HEAD_MAX_D && n >= 0 && K >= 1 && n * K <= 0x7FFFFFFFll * 256, "bad shape");
    RLHIP_REQUIRE((squash == 0 || squash == 1) && (soft == 0 || soft == 1), "squash / soft must be 0 or 1");
    if (n == 0) return RLHIP_OK;
    const dim3 grid((unsigned)((n * K + 255) / 256)), block(256);
    if (soft)
        hipLaunchKernelGGL((gaussian_head_kernel<1, 1>), grid, block, 0, as_stream(stream), mu, raw_sigma, action, (int)d, n,
                           (int)K, min_sigma, max_sigma, 1, 0ull, 0u, 0u, nullptr, logp_out);
    else
        hipLaunchKernelGGL((gaussian_head_kernel<0, 1>), grid, block, 0, as_stream(stream), mu, raw_sigma, action, (int)d, n,
                           (int)K, min_sigma, max_sigma, squash, 0ull, 0u, 0u, nullptr, logp_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
