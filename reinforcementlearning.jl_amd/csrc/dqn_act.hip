// dqn_act.hip -- plan! + act! + push! of one DQN vec-step in ONE kernel (the first three arrows of `_run`'s loop
// body, RLCore/src/core/run.jl:52-66): per env instance
//     q = forward(learner, state(env))                         flux_approximator.jl:43
//     a = plan!(EpsilonGreedyExplorer, q)                      epsilon_greedy_explorer.jl:108-112
//     act!(env, a)  (+ auto-reset, MultiThreadEnv protocol)    CartPoleEnv.jl:112-140 etc.
//     push!(trajectory, (state = s', action, reward, terminal))  agent_base.jl:56-59
// It calls the same device functions as dqn_plan_wide_kernel (net_forward, eps_greedy_select1), env_step_kernel
// (env_step1 / env_reset1 / env_obs1) and the ring push (same slots), so every output is bit-identical to the three
// separate launches -- it only removes two of them and the HBM round trip of the action / observation arrays.
// L = H / 16 lanes per env instance, weights in registers (as the PPO rollout kernel, ppo.hip).
#include "act_device.h"
#include "mlp_device.h"
#include "select_device.h"

namespace rlhip {


#ifdef RLHIP_DQN_TIMING  // start / end of workgroup 0, thread 0 (tools/dqn_timeline.py)
__device__ long long g_act_stamps[4];
#define ACT_STAMP(k)                                                                 \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_act_stamps[k] = wall_clock64();   \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
#else
#define ACT_STAMP(k) \
    do {             \
    } while (0)
#endif

struct RegQa {
    const float* q;
    __device__ __forceinline__ float operator()(int k) const { return q[k]; }
};

template <class P, int H, int L, int ACT>
__global__ __launch_bounds__(256, 1) void dqn_act_kernel(P p, EnvArrays<float> st, int64_t n, const float* __restrict__ params,
                                                         int na, double eps, uint64_t explorer_seed, uint32_t step,
                                                         uint64_t env_seed, uint32_t env_id_base, ActRing rb,
                                                         int32_t* __restrict__ actions, float* __restrict__ q_out,
                                                         float* __restrict__ obs_out, float* __restrict__ last_obs) {
    constexpr int NS = P::ODIM;
    constexpr int HPL = H / L;
    ACT_STAMP(0);
    int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t env = gl / L;
    const int sub = (int)(gl % L);
    const bool active = env < n;
    if (!active) env = n - 1;
    const bool writer = active && sub == 0;
    const uint32_t id = env_id_base + (uint32_t)env;
    NetRegs<NS, HPL> Q;
    load_net<NS, HPL>(Q, params, H, na, sub, L);
    LaneState<float> e;
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
    e.t = st.t[env];
    e.episode = st.episode[env];
    float x[4];
    env_obs1(p, e, x);
    float q[MAXO];
    net_forward<NS, HPL, L, ACT>(Q, x, q);
    const int32_t a = eps_greedy_select1(RegQa{q}, NoMask{}, na, eps, false, explorer_seed, id, step);
    float r;
    bool d;
    env_step1(p, e, a, 0.0f, r, d);
    float lo[4] = {0.f, 0.f, 0.f, 0.f};
    if (last_obs) env_obs1(p, e, lo);
    if (d) env_reset1(p, e, env_seed, id);
    float xn[4];
    env_obs1(p, e, xn);
    if (!writer) return;
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
    st.t[env] = e.t;
    if (d) st.episode[env] = e.episode;
    st.reward[env] = r;
    st.done[env] = (uint8_t)d;
    actions[env] = a;
    if (q_out)
        for (int o = 0; o < na; ++o) q_out[(int64_t)o * n + env] = q[o];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (obs_out) obs_out[(int64_t)k * n + env] = xn[k];
        if (last_obs) last_obs[(int64_t)k * n + env] = lo[k];
    }
#pragma unroll
    for (int k = NS; k < 4; ++k) xn[k] = 0.f;
    ring_push_transition(rb.rec, rb.state_slot, rb.prev_slot, n, env, xn, a, r, d ? 1u : 0u);
    ACT_STAMP(1);
}

template <class P>
static int32_t act_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, const float* params, int h,
                        int na, int act, double eps, uint64_t explorer_seed, uint32_t step, uint64_t env_seed,
                        uint32_t env_id_base, ActRing rb, int32_t* actions, float* q_out, float* obs_out, float* last_obs,
                        hipStream_t s) {
    RLHIP_REQUIRE(st && st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    typename P::cfg_t c2 = *cfg;
    c2.continuous = 0;
    P p = P::make(c2);
    EnvArrays<float> a = EnvArrays<float>::from(*st);
#define LAUNCH_A(H_, L_)                                                                                            \
    do {                                                                                                            \
        if (act == 0)                                                                                               \
            hipLaunchKernelGGL((dqn_act_kernel<P, H_, L_, 0>), dim3((int)((n * L_ + 255) / 256)), dim3(256), 0, s, p, a, n, \
                               params, na, eps, explorer_seed, step, env_seed, env_id_base, rb, actions, q_out, obs_out, \
                               last_obs);                                                                           \
        else                                                                                                        \
            hipLaunchKernelGGL((dqn_act_kernel<P, H_, L_, 1>), dim3((int)((n * L_ + 255) / 256)), dim3(256), 0, s, p, a, n, \
                               params, na, eps, explorer_seed, step, env_seed, env_id_base, rb, actions, q_out, obs_out, \
                               last_obs);                                                                           \
    } while (0)
    if (h == 256) LAUNCH_A(256, 16);
    else if (h == 128) LAUNCH_A(128, 8);
    else LAUNCH_A(64, 4);
#undef LAUNCH_A
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// act! + push! in one launch for policies whose plan! is a separate kernel (the MFMA Q-network): one lane per env
// instance, the same device functions and the same ring slots as env_step_kernel + push_transition_kernel.
template <class P>
__global__ __launch_bounds__(256) void env_act_push_kernel(P p, EnvArrays<float> st, int64_t n,
                                                           const int32_t* __restrict__ actions, uint64_t env_seed,
                                                           uint32_t env_id_base, ActRing rb, float* __restrict__ obs_out,
                                                           float* __restrict__ last_obs) {
    const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    env_act_push1(p, st, n, env, actions[env], env_seed, env_id_base, rb, obs_out, last_obs);
}

template <class P>
static int32_t act_push_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, const int32_t* actions,
                             uint64_t env_seed, uint32_t env_id_base, ActRing rb, float* obs_out, float* last_obs,
                             hipStream_t s) {
    RLHIP_REQUIRE(st && st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    typename P::cfg_t c2 = *cfg;
    c2.continuous = 0;
    P p = P::make(c2);
    EnvArrays<float> a = EnvArrays<float>::from(*st);
    hipLaunchKernelGGL((env_act_push_kernel<P>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, p, a, n, actions, env_seed,
                       env_id_base, rb, obs_out, last_obs);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

#ifdef RLHIP_DQN_TIMING
extern "C" int32_t rlhip_debug_act_stamps(long long* out4) {
    RLHIP_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(rlhip::g_act_stamps), sizeof(long long) * 4));
    return RLHIP_OK;
}
#endif

extern "C" int32_t rlhip_dqn_act_supported(int32_t kind, int64_t n, int64_t h) {
    return (kind >= 0 && kind <= 2 && (h == 256 || h == 128 || h == 64) && n >= 1 && n * 16 <= ((int64_t)1 << 22)) ? 1 : 0;
}

extern "C" int32_t rlhip_dqn_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                                     const float* params, int64_t h, int64_t na, int32_t act, double eps,
                                     uint64_t explorer_seed, uint32_t explorer_step, uint64_t env_seed,
                                     uint32_t env_id_base, rlhip_ring* rb, int32_t* actions, float* q_out,
                                     float* obs_out, float* last_obs, rlhip_stream_t stream) {
    RLHIP_REQUIRE(env_cfg && st && params && rb && actions, "NULL argument");
    RLHIP_REQUIRE(rlhip_dqn_act_supported(kind, n, h), "unsupported (kind, n, hidden) for the fused act kernel");
    RLHIP_REQUIRE(na >= 1 && na <= MAXO && (act == 0 || act == 1), "bad network description");
    RLHIP_REQUIRE(rb->elem_bytes == 4 && rb->n_env == n && rb->obs_dim == (kind == 0 ? 4 : (kind == 1 ? 3 : 2)),
                  "ring geometry does not match the env");
    RLHIP_REQUIRE(rb->len_sa >= 1, "push the first state before the first transition");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "the fused act + push kernels write a record ring (rlhip_ring_init, ABI 2)");
    ActRing ar = claim_slots(rb);
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return act_impl<CartPoleParams<float>>((const rlhip_cartpole_cfg*)env_cfg, st, n, params, (int)h, (int)na, act, eps,
                                               explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out,
                                               obs_out, last_obs, s);
    if (kind == 1)
        return act_impl<PendulumParams<float>>((const rlhip_pendulum_cfg*)env_cfg, st, n, params, (int)h, (int)na, act, eps,
                                               explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out,
                                               obs_out, last_obs, s);
    return act_impl<MountainCarParams<float>>((const rlhip_mountaincar_cfg*)env_cfg, st, n, params, (int)h, (int)na, act,
                                              eps, explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out,
                                              obs_out, last_obs, s);
}

/* act!(env, actions) + push!(trajectory, (state = s', action, reward, terminal)) in one launch (discrete Float32 envs;
 * actions: i32[n] 0-based, already planned).  Same outputs as rlhip_env_step (auto-reset) followed by
 * rlhip_ring_push_transition. */
extern "C" int32_t rlhip_env_act_push_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                                          const int32_t* actions, uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb,
                                          float* obs_out, float* last_obs, rlhip_stream_t stream) {
    RLHIP_REQUIRE(env_cfg && st && rb && actions, "NULL argument");
    RLHIP_REQUIRE(kind >= 0 && kind <= 2 && n >= 1, "kind must be 0 (cartpole), 1 (pendulum) or 2 (mountaincar)");
    RLHIP_REQUIRE(rb->elem_bytes == 4 && rb->n_env == n && rb->obs_dim == (kind == 0 ? 4 : (kind == 1 ? 3 : 2)),
                  "ring geometry does not match the env");
    RLHIP_REQUIRE(rb->len_sa >= 1, "push the first state before the first transition");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "the fused act + push kernels write a record ring (rlhip_ring_init, ABI 2)");
    ActRing ar = claim_slots(rb);
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return act_push_impl<CartPoleParams<float>>((const rlhip_cartpole_cfg*)env_cfg, st, n, actions, env_seed, env_id_base,
                                                    ar, obs_out, last_obs, s);
    if (kind == 1)
        return act_push_impl<PendulumParams<float>>((const rlhip_pendulum_cfg*)env_cfg, st, n, actions, env_seed, env_id_base,
                                                    ar, obs_out, last_obs, s);
    return act_push_impl<MountainCarParams<float>>((const rlhip_mountaincar_cfg*)env_cfg, st, n, actions, env_seed,
                                                   env_id_base, ar, obs_out, last_obs, s);
}
