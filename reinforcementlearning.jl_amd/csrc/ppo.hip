// ppo.hip -- the vectorised PPO hot path: policy forward + action sampling, fused T-step rollout,
// clipped-surrogate loss + gradient, update loop.
//
// What it replaces in the reference (see DESIGN.md for the full map):
//   * the vector-env run loop  `_run(policy, env::MultiThreadEnv, ...)`
//       docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:351-374
//     i.e. per vec-step: plan! (RLCore/policies/learners/flux_approximator.jl:43 forward +
//     RLCore/utils/networks.jl:425-432 Gumbel-max sample or :64-82 gaussian sample), PreAct push
//     (state, action, action_log_prob), act! (RLEnvs/*Env.jl _step! + auto reset), PostAct push
//     (reward, terminal)  -- RLCore/core/run.jl:52-67 is the scalar form of the same loop;
//   * PPOPolicy's update (removed Zoo; blog a_practical_introduction_to_RL.jl/index.html:15257-15287,
//     formulas SURVEY.md Appendix B): generalized_advantage_estimation (RLCore/utils/basic.jl:334-417,
//     scans.hip), shuffled micro-batches, clipped-surrogate / value / entropy loss, Zygote backward
//     through the two Chains, clip_by_global_norm! (RLCore/utils/basic.jl:19-29) and
//     Flux.Optimise.update! with Adam (flux_approximator.jl:46) -- optim.hip.
//
// Kernel design (gfx950):
//   rollout_split_kernel  L = h/16 lanes cooperate on one env: each lane owns 16 hidden units with their
//                         weights in registers for the whole T-step rollout; the <= 4 output sums are
//                         combined by DPP adds inside the wavefront.  Two wavefronts per env group on one
//                         SIMD: the actor wave runs the dependent chain (actor, selection, env step), the
//                         critic wave everything else (critic, trajectory stores, sampling noise, GAE).
//                         4096 envs x 16 lanes x 2 = 2048 wavefronts = two per SIMD of the chip; no
//                         inter-workgroup communication at all (envs are independent while the weights
//                         are frozen), so ONE launch covers T vec-steps.
//   rollout_scalar_kernel one lane per env, hidden units walked with wave-uniform (scalar-loaded)
//                         weights: the variant for very large n or unusual h.
//   ppo_grad_kernel       per 64-sample tile: phase 1 = lane per sample, the 4 waves split the hidden
//                         units (wave-uniform weights -> scalar loads), partial sums combined through LDS,
//                         softmax / ratio / clip / value / entropy derivative per sample; phase 2 =
//                         lane per hidden unit, samples walked from LDS (broadcast reads), weight
//                         gradients accumulated in registers -- no atomics, no cross-lane reductions.
//                         Per-workgroup partial gradients are summed in a fixed order by
//                         reduce_partials_kernel, so the gradient is run-to-run deterministic and
//                         replicas on different GPUs stay bit-identical after the all-reduce.
//   The epoch shuffle is a keyed bijection evaluated inline (common.h permute) -- no shuffle kernel,
//   no index array in HBM.
#include "env_device.h"
#include "mlp_device.h"
#include "select_device.h"
#include "ppo_common.h"
#include "ppo_sample_device.h"

extern "C" int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow,
                                       int64_t n, float grad_scale, float clip_norm, float lr, float beta1,
                                       float beta2, float eps, float* gn_out, rlhip_stream_t stream);
extern "C" int32_t rlhip_gae_returns_f32(float* advantages, float* returns, const float* rewards,
                                         const float* values, const uint8_t* terminal, int64_t n_env,
                                         int64_t T, float gamma, float lambda, rlhip_stream_t stream);

namespace rlhip {

__global__ void counters_advance_kernel(uint32_t* ctr, uint32_t d0, uint32_t d1) {
    ctr[0] += d0;  // vec-step counter
    ctr[1] += d1;  // update counter
}

// ---------------------------------------------------------------------------------- rollout ----
// L = h / 16 lanes cooperate on one env (each owns 16 hidden units, weights in registers for the whole rollout), and the work
// of an env group is split over TWO wavefronts that share a SIMD (512 threads per workgroup: wave w = actor wave, wave w + 4 =
// critic wave of the same 256 / L envs; the hardware places wave w and w + 4 of a workgroup on one SIMD:
// tools/micro/wave_simd_map.hip).  Only obs -> actor -> select -> env step -> obs is a dependent chain; a lone wavefront issues
// it at ~7 cycles per instruction (56 % VALU-active, profiles/r02_summary.md), so everything that is NOT on the chain sits in
// the second wave and fills the first one's issue gaps:
//   actor wave : actor forward, action selection, env step (+ auto reset), one 32-byte step record per env into LDS
//   critic wave: critic forward, ALL trajectory stores, the Gumbel / normal noise of the next 16-step chunk (Philox + Float64
//                log / sqrt / sin / cos: depends on (env, step) only, one step per lane of the env's group -- same operations
//                on the same operands as policy_sample), the bootstrap value and the GAE + returns scan
// One workgroup barrier per vec-step hands the double-buffered step record over (the critic wave trails by one step).
// Measured (headline rollout, 4096 CartPole envs x T = 32, same box): one wave doing everything 64.0 us -> 60.4 us with the
// head known at compile time -> 49.4 us split (profiles/r04_rollout.md).
// NOA: actor outputs evaluated (2: two actions or (mu, log sigma); MAXO otherwise); the critic has one.
// HEAD: the head known at compile time -- 2 / 3 = categorical over 2 / 3 actions, 1 = one-dimensional Gaussian, 0 = read
// pd.cont / pd.na at run time.  With the run-time form the selection is three loops over `na` with a compare-and-select
// chain per register-array access, and both heads' code sits in the step loop; the compile-time form is the same operations
// on the same operands, unrolled.
// per-phase cycle sums of workgroup 0 (one actor wave, one critic wave), printed at the end of the launch: build with
// RLHIP_EXTRA_FLAGS=-DRLHIP_ROLLOUT_TIMING (tools/rollout_one.py; proportions -- the stamps cost a few cycles each)
#ifdef RLHIP_ROLLOUT_TIMING
#define RT_DECL long long rt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt_last_ = 0
#define RT_START() rt_last_ = (long long)__builtin_amdgcn_s_memtime()
#define RT_STAMP(k)                                                   \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
        rt_[k] += now_ - rt_last_;                                    \
        rt_last_ = now_;                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define RT_PRINT(tag)                                                                                             \
    do {                                                                                                          \
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0)                                                          \
            printf("%s T=%d: %lld %lld %lld %lld %lld %lld %lld %lld\n", tag, T, rt_[0], rt_[1], rt_[2], rt_[3], rt_[4], \
                   rt_[5], rt_[6], rt_[7]);                                                                       \
    } while (0)
#else
#define RT_DECL
#define RT_START()
#define RT_STAMP(k)
#define RT_PRINT(tag)
#endif

template <class P, int H, int L, int ACT, int NOA, int HEAD>
__global__ __launch_bounds__(512, 1) void rollout_split_kernel(P p, EnvArrays<float> st, int64_t n, int T,
                                                               PolicyDesc pd, const float* __restrict__ params,
                                                               uint64_t seed, uint32_t env_id_base,
                                                               uint32_t vec_step0_in, const uint32_t* __restrict__ ctr, TrajPtrs tr,
                                                               int store_state) {
    const uint32_t vec_step0 = vec_step0_in + (ctr ? ctr[0] : 0u);  // device-resident counter (graph replay)
    constexpr int NS = P::ODIM;
    constexpr int HPL = H / L;
    constexpr int EPB = 256 / L;  // envs per workgroup
    constexpr int NOISE_CH = 16;
    const int cont = HEAD == 0 ? pd.cont : (HEAD == 1 ? 1 : 0);
    const int na = HEAD == 0 ? pd.na : (HEAD == 1 ? 1 : HEAD);
    if (HEAD != 0) p.continuous = cont;
    const int role = (int)(threadIdx.x >> 8);  // 0: actor wave, 1: critic wave
    const int tl = (int)(threadIdx.x & 255);
    int64_t gl = (int64_t)blockIdx.x * 256 + tl;
    int64_t env = gl / L;
    const int sub = (int)(gl % L);
    const int eg = tl / L;
    const bool active = env < n;
    if (!active) env = n - 1;  // keep the whole wave on valid data (shuffles stay uniform); no stores
    const bool writer = active && sub == 0;
    const uint32_t id = env_id_base + (uint32_t)env;

    __shared__ double l_noise[2][EPB][NOISE_CH][MAXO];  // noise of two 16-step chunks (the critic wave fills the next one)
    __shared__ float4 l_step[2][EPB][2];                // {x0..x3}, {logp, action bits, reward, terminal} of step t (t & 1)

    if (role == 0) {
        __builtin_amdgcn_s_setprio(3);  // the chain goes first whenever both waves of the SIMD are ready
        NetRegs<NS, HPL> A;
        load_net<NS, HPL>(A, params, H, pd.nout_a, sub, L);
        LaneState<float> e;
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
        e.t = st.t[env];
        e.episode = st.episode[env];
        float last_r = 0.0f;
        bool last_d = false;
        RT_DECL;
        __syncthreads();  // the noise of chunk 0
        RT_START();
        for (int t = 0; t < T; ++t) {
            double nz[MAXO];  // requested before the forward pass, consumed after it
#pragma unroll
            for (int k = 0; k < MAXO; ++k) nz[k] = (k < na) ? l_noise[(t / NOISE_CH) & 1][eg][t & (NOISE_CH - 1)][k] : 0.0;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            env_obs1(p, e, x);  // state(env) at PreActStage (post auto-reset)
            RT_STAMP(0);  // noise read issued + obs
            float oa[MAXO];
            net_forward<NS, HPL, L, ACT, NOA>(A, x, oa);
            RT_STAMP(1);  // actor forward
            int32_t ai;
            float af, lp;
            policy_select(cont, na, oa, nz, ai, af, lp);
            RT_STAMP(2);  // selection
            env_step1(p, e, ai, af, last_r, last_d);
            if (last_d) env_reset1(p, e, seed, id);  // MultiThreadEnv auto-reset
            RT_STAMP(3);  // env step
            if (sub == 0) {
                l_step[t & 1][eg][0] = make_float4(x[0], x[1], x[2], x[3]);
                l_step[t & 1][eg][1] = make_float4(lp, cont ? af : __int_as_float(ai), last_r, last_d ? 1.0f : 0.0f);
            }
            RT_STAMP(4);  // record
            __syncthreads();
            RT_STAMP(5);  // barrier
        }
        RT_PRINT("actor wave: obs/noise | forward | select | env step | record | barrier");
        {
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            env_obs1(p, e, x);
            if (sub == 0) l_step[T & 1][eg][0] = make_float4(x[0], x[1], x[2], x[3]);
            __syncthreads();
        }
        if (writer && store_state) {
#pragma unroll
            for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
            st.t[env] = e.t;
            st.episode[env] = e.episode;
            if (T > 0) {
                st.reward[env] = last_r;
                st.done[env] = (uint8_t)last_d;
            }
        }
    } else {
        NetRegs<NS, HPL> C;
        load_net<NS, HPL>(C, params + pd.np_a, H, 1, sub, L);
        // noise of steps [c0, c0 + NOISE_CH): one step per lane of the env's group (same operations as policy_sample)
        auto fill_noise = [&](int c0) {
            for (int i = sub; i < NOISE_CH && c0 + i < T; i += L) {
                double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
                policy_noise(cont, na, seed, id, vec_step0 + (uint32_t)(c0 + i), nz);
#pragma unroll
                for (int k = 0; k < MAXO; ++k) l_noise[(c0 / NOISE_CH) & 1][eg][i][k] = nz[k];
            }
        };
        fill_noise(0);
        RT_DECL;
        __syncthreads();
        RT_START();
        for (int t = 0; t < T; ++t) {
            // the actor wave is in step t, reading chunk t / 16: the other buffer (last read in step t - 1) takes the next chunk
            if ((t & (NOISE_CH - 1)) == 0 && t + NOISE_CH < T) fill_noise(t + NOISE_CH);
            RT_STAMP(0);  // noise of a later chunk
            __syncthreads();  // step t's record
            RT_STAMP(1);  // barrier
            const float4 xv = l_step[t & 1][eg][0];
            const float x[4] = {xv.x, xv.y, xv.z, xv.w};
            float oc[MAXO];
            net_forward<NS, HPL, L, ACT, 1>(C, x, oc);
            RT_STAMP(2);  // critic forward
            if (writer) {
                const float4 rec = l_step[t & 1][eg][1];
#pragma unroll
                for (int k = 0; k < NS; ++k) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
                tr.value[(int64_t)t * n + env] = oc[0];
                tr.logp[(int64_t)t * n + env] = rec.x;
                if (cont) tr.action_f[(int64_t)t * n + env] = rec.y;
                else tr.action_i[(int64_t)t * n + env] = __float_as_int(rec.y);
                tr.reward[(int64_t)t * n + env] = rec.z;
                tr.terminal[(int64_t)t * n + env] = (uint8_t)(rec.w != 0.0f);
            }
            RT_STAMP(3);  // trajectory stores issued
        }
        RT_PRINT("critic wave: noise | barrier | forward | stores");
        __syncthreads();  // the state after the last step
        {
            const float4 xv = l_step[T & 1][eg][0];
            const float x[4] = {xv.x, xv.y, xv.z, xv.w};
            float oc[MAXO];
            net_forward<NS, HPL, L, ACT, 1>(C, x, oc);
            if (writer) {
#pragma unroll
                for (int k = 0; k < NS; ++k) tr.obs[((int64_t)T * NS + k) * n + env] = x[k];
                tr.value[(int64_t)T * n + env] = oc[0];
            }
        }
        // generalized_advantage_estimation + returns for this env, from the values / rewards this lane just wrote
        if (writer && T > 0 && tr.adv && tr.ret)
            gae_scan_lane(tr.adv, tr.ret, tr.reward, tr.value, tr.terminal, n, T, env, pd.gamma, pd.lambda);
    }
}

template <class P, int ACT>
__global__ __launch_bounds__(256) void rollout_scalar_kernel(P p, EnvArrays<float> st, int64_t n, int T,
                                                             PolicyDesc pd, const float* __restrict__ params,
                                                             uint64_t seed, uint32_t env_id_base,
                                                             uint32_t vec_step0_in, const uint32_t* __restrict__ ctr, TrajPtrs tr,
                                                              int store_state) {
    const uint32_t vec_step0 = vec_step0_in + (ctr ? ctr[0] : 0u);  // device-resident counter (graph replay)
    constexpr int NS = P::ODIM;
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    uint32_t id = env_id_base + (uint32_t)env;
    LaneState<float> e;
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
    e.t = st.t[env];
    e.episode = st.episode[env];
    float last_r = 0.0f;
    bool last_d = false;
    for (int t = 0; t < T; ++t) {
        float x[4];
        env_obs1(p, e, x);
        float oa[MAXO], oc[MAXO];
        net_forward_scalar<NS, ACT>(params, pd.h, pd.nout_a, x, oa);
        net_forward_scalar<NS, ACT>(params + pd.np_a, pd.h, 1, x, oc);
        int32_t ai;
        float af, lp;
        policy_sample(pd.cont, pd.na, oa, seed, id, vec_step0 + (uint32_t)t, ai, af, lp);
#pragma unroll
        for (int k = 0; k < NS; ++k) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
        tr.value[(int64_t)t * n + env] = oc[0];
        tr.logp[(int64_t)t * n + env] = lp;
        if (pd.cont) tr.action_f[(int64_t)t * n + env] = af;
        else tr.action_i[(int64_t)t * n + env] = ai;
        env_step1(p, e, ai, af, last_r, last_d);
        if (last_d) env_reset1(p, e, seed, id);
        tr.reward[(int64_t)t * n + env] = last_r;
        tr.terminal[(int64_t)t * n + env] = (uint8_t)last_d;
    }
    {
        float x[4], oc[MAXO];
        env_obs1(p, e, x);
        net_forward_scalar<NS, ACT>(params + pd.np_a, pd.h, 1, x, oc);
#pragma unroll
        for (int k = 0; k < NS; ++k) tr.obs[((int64_t)T * NS + k) * n + env] = x[k];
        tr.value[(int64_t)T * n + env] = oc[0];
    }
    if (T > 0 && tr.adv && tr.ret)
        gae_scan_lane(tr.adv, tr.ret, tr.reward, tr.value, tr.terminal, n, T, env, pd.gamma, pd.lambda);
    if (store_state) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
        st.t[env] = e.t;
        st.episode[env] = e.episode;
        if (T > 0) {
            st.reward[env] = last_r;
            st.done[env] = (uint8_t)last_d;
        }
    }
}


template <class P>
static int32_t rollout_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                            const PolicyDesc& pd, const float* params, uint64_t seed, uint32_t env_id_base,
                            uint32_t vec_step0, const uint32_t* ctr, const rlhip_ppo_traj* traj, hipStream_t s) {
    RLHIP_REQUIRE(st && st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    typename P::cfg_t c2 = *cfg;
    c2.continuous = pd.cont;  // the policy head decides the action type
    P p = P::make(c2);
    EnvArrays<float> a = EnvArrays<float>::from(*st);
    TrajPtrs tr = TrajPtrs::from(*traj);
    // wide variant while the chip is not yet full of one-lane-per-env wavefronts
    bool wide = (pd.h == 256 || pd.h == 128 || pd.h == 64) && n * 16 <= (int64_t)1 << 22;
#define LAUNCH_WIDE_AS(H, L, ACT_, NOA_, HEAD_)                                                                  \
    hipLaunchKernelGGL((rollout_split_kernel<P, H, L, ACT_, NOA_, HEAD_>), dim3((int)((n * L + 255) / 256)), dim3(512), \
                       0, s, p, a, n, (int)T, pd, params, seed, env_id_base, vec_step0, ctr, tr, 1)
#define LAUNCH_WIDE(H, L)                                                                     \
    do {                                                                                      \
        if (pd.act == 0 && !pd.cont && pd.na == 2) LAUNCH_WIDE_AS(H, L, 0, 2, 2);             \
        else if (pd.act == 0 && pd.cont && pd.na == 1) LAUNCH_WIDE_AS(H, L, 0, 2, 1);         \
        else if (pd.act == 0 && !pd.cont && pd.na == 3) LAUNCH_WIDE_AS(H, L, 0, MAXO, 3);     \
        else if (pd.act == 0) LAUNCH_WIDE_AS(H, L, 0, MAXO, 0);                               \
        else if (!pd.cont && pd.na == 2) LAUNCH_WIDE_AS(H, L, 1, 2, 2);                       \
        else if (pd.cont && pd.na == 1) LAUNCH_WIDE_AS(H, L, 1, 2, 1);                        \
        else LAUNCH_WIDE_AS(H, L, 1, MAXO, 0);                                                \
    } while (0)
    if (wide && pd.h == 256) LAUNCH_WIDE(256, 16);  // 32 lanes per env (8 units each; the serial part replicated on twice the lanes) is slower
    else if (wide && pd.h == 128) LAUNCH_WIDE(128, 8);
    else if (wide && pd.h == 64) LAUNCH_WIDE(64, 4);
    else if (pd.act == 0)
        hipLaunchKernelGGL((rollout_scalar_kernel<P, 0>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, p, a, n,
                           (int)T, pd, params, seed, env_id_base, vec_step0, ctr, tr, 1);
    else
        hipLaunchKernelGGL((rollout_scalar_kernel<P, 1>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, p, a, n,
                           (int)T, pd, params, seed, env_id_base, vec_step0, ctr, tr, 1);
#undef LAUNCH_WIDE_AS
#undef LAUNCH_WIDE
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// ------------------------------------------------------------------------- single-step plan ----
// plan!(policy, env) for the vector env: forward + sample for n observations (SoA ns x n).  Uses the
// same device functions (and the same lane split) as the fused rollout, so both paths are bit-identical.
template <int NS, int H, int L, int ACT>
__global__ __launch_bounds__(256, 1) void plan_wide_kernel(int64_t n, PolicyDesc pd,
                                                           const float* __restrict__ params,
                                                           const float* __restrict__ obs, uint64_t seed,
                                                           uint32_t env_id_base, uint32_t step,
                                                           int32_t* __restrict__ action_i,
                                                           float* __restrict__ action_f,
                                                           float* __restrict__ logp, float* __restrict__ value) {
    constexpr int HPL = H / L;
    int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t env = gl / L;
    int sub = (int)(gl % L);
    bool active = env < n;
    if (!active) env = n - 1;
    NetRegs<NS, HPL> A, C;
    load_net<NS, HPL>(A, params, H, pd.nout_a, sub, L);
    load_net<NS, HPL>(C, params + pd.np_a, H, 1, sub, L);
    float x[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = obs[(int64_t)k * n + env];
    float oa[MAXO], oc[MAXO];
    net_forward<NS, HPL, L, ACT>(A, x, oa);
    net_forward<NS, HPL, L, ACT>(C, x, oc);
    int32_t ai;
    float af, lp;
    policy_sample(pd.cont, pd.na, oa, seed, env_id_base + (uint32_t)env, step, ai, af, lp);
    if (active && sub == 0) {
        if (pd.cont) action_f[env] = af;
        else action_i[env] = ai;
        if (logp) logp[env] = lp;
        if (value) value[env] = oc[0];
    }
}

template <int NS, int ACT>
__global__ __launch_bounds__(256) void plan_scalar_kernel(int64_t n, PolicyDesc pd,
                                                          const float* __restrict__ params,
                                                          const float* __restrict__ obs, uint64_t seed,
                                                          uint32_t env_id_base, uint32_t step,
                                                          int32_t* __restrict__ action_i,
                                                          float* __restrict__ action_f, float* __restrict__ logp,
                                                          float* __restrict__ value) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    float x[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = obs[(int64_t)k * n + env];
    float oa[MAXO], oc[MAXO];
    net_forward_scalar<NS, ACT>(params, pd.h, pd.nout_a, x, oa);
    net_forward_scalar<NS, ACT>(params + pd.np_a, pd.h, 1, x, oc);
    int32_t ai;
    float af, lp;
    policy_sample(pd.cont, pd.na, oa, seed, env_id_base + (uint32_t)env, step, ai, af, lp);
    if (pd.cont) action_f[env] = af;
    else action_i[env] = ai;
    if (logp) logp[env] = lp;
    if (value) value[env] = oc[0];
}

template <int NS>
static int32_t plan_impl(int64_t n, const PolicyDesc& pd, const float* params, const float* obs, uint64_t seed,
                         uint32_t env_id_base, uint32_t step, int32_t* ai, float* af, float* logp, float* value,
                         hipStream_t s) {
    bool wide = (pd.h == 256 || pd.h == 128 || pd.h == 64) && n * 16 <= (int64_t)1 << 22;
#define LAUNCH_PW(H, L)                                                                                        \
    do {                                                                                                       \
        if (pd.act == 0)                                                                                       \
            hipLaunchKernelGGL((plan_wide_kernel<NS, H, L, 0>), dim3((int)((n * L + 255) / 256)), dim3(256), 0, s, n, \
                               pd, params, obs, seed, env_id_base, step, ai, af, logp, value);                \
        else                                                                                                   \
            hipLaunchKernelGGL((plan_wide_kernel<NS, H, L, 1>), dim3((int)((n * L + 255) / 256)), dim3(256), 0, s, n, \
                               pd, params, obs, seed, env_id_base, step, ai, af, logp, value);                \
    } while (0)
    if (wide && pd.h == 256) LAUNCH_PW(256, 16);
    else if (wide && pd.h == 128) LAUNCH_PW(128, 8);
    else if (wide && pd.h == 64) LAUNCH_PW(64, 4);
    else if (pd.act == 0)
        hipLaunchKernelGGL((plan_scalar_kernel<NS, 0>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, n, pd, params,
                           obs, seed, env_id_base, step, ai, af, logp, value);
    else
        hipLaunchKernelGGL((plan_scalar_kernel<NS, 1>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, n, pd, params,
                           obs, seed, env_id_base, step, ai, af, logp, value);
#undef LAUNCH_PW
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// generic forward (any n_in <= 16, n_out <= 32): one lane per sample, scalar-loaded weights
__global__ __launch_bounds__(256) void mlp2_forward_kernel(const float* __restrict__ p, int n_in, int h,
                                                           int n_out, int act, const float* __restrict__ x,
                                                           int64_t batch, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    const float* W1 = p;
    const float* b1 = W1 + h * n_in;
    const float* W2 = b1 + h;
    const float* b2 = W2 + n_out * h;
    float xin[16], acc[32];
    for (int k = 0; k < n_in; ++k) xin[k] = x[(int64_t)k * batch + i];
    for (int o = 0; o < n_out; ++o) acc[o] = b2[o];
    for (int j = 0; j < h; ++j) {
        float z = b1[j];
        for (int k = 0; k < n_in; ++k) z = fmaf(W1[j + h * k], xin[k], z);
        float hv = act_fwd(act, z);
        for (int o = 0; o < n_out; ++o) acc[o] = fmaf(W2[o + n_out * j], hv, acc[o]);
    }
    for (int o = 0; o < n_out; ++o) out[(int64_t)o * batch + i] = acc[o];
}

__global__ __launch_bounds__(256) void mlp2_init_kernel(float* __restrict__ p, int n_in, int h, int n_out,
                                                        uint64_t seed, uint32_t net_id) {
    int64_t np = mlp2_nparams(n_in, h, n_out);
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= np) return;
    int64_t o1 = (int64_t)h * n_in, o2 = o1 + h, o3 = o2 + (int64_t)n_out * h;
    float val = 0.0f;  // Flux Dense bias default: zeros
    if (q < o1) {
        float s1 = sqrtf(6.0f / (float)(n_in + h));
        u32x4 w = philox4x32_10(seed, (uint32_t)q, 0, net_id * 4u + 0u, TAG_INIT);
        val = (2.0f * u01_f32(w.x) - 1.0f) * s1;
    } else if (q >= o2 && q < o3) {
        float s2 = sqrtf(6.0f / (float)(h + n_out));
        u32x4 w = philox4x32_10(seed, (uint32_t)(q - o2), 0, net_id * 4u + 2u, TAG_INIT);
        val = (2.0f * u01_f32(w.x) - 1.0f) * s2;
    }
    p[q] = val;
}

// per-step protocol: one env per lane, grid-stride
__global__ __launch_bounds__(256) void ppo_push_preact_kernel(TrajPtrs tr, int64_t t, int ns, int64_t n,
                                                              const float* __restrict__ obs, const float* __restrict__ value,
                                                              const float* __restrict__ logp,
                                                              const int32_t* __restrict__ a_i, const float* __restrict__ a_f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        for (int k = 0; k < ns; ++k) tr.obs[(t * ns + k) * n + i] = obs[(int64_t)k * n + i];
        tr.value[t * n + i] = value[i];
        if (logp) tr.logp[t * n + i] = logp[i];
        if (a_i) tr.action_i[t * n + i] = a_i[i];
        if (a_f) tr.action_f[t * n + i] = a_f[i];
    }
}
__global__ __launch_bounds__(256) void ppo_push_postact_kernel(TrajPtrs tr, int64_t t, int64_t n,
                                                               const float* __restrict__ reward,
                                                               const uint8_t* __restrict__ done) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        tr.reward[t * n + i] = reward[i];
        tr.terminal[t * n + i] = done[i];
    }
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_ppo_default(rlhip_ppo_cfg* c) {
    RLHIP_REQUIRE(c != nullptr, "cfg is NULL");
    // blog a_practical_introduction_to_RL.jl/index.html:15257-15278
    c->gamma = 0.99f;
    c->lambda = 0.95f;
    c->clip_range = 0.1f;
    c->max_grad_norm = 0.5f;
    c->actor_loss_weight = 1.0f;
    c->critic_loss_weight = 0.5f;
    c->entropy_loss_weight = 0.001f;
    c->lr = 1e-3f;
    c->beta1 = 0.9f;
    c->beta2 = 0.999f;
    c->adam_eps = 1e-8f;
    c->n_epochs = 4;
    c->n_microbatches = 4;
    c->hidden = 256;
    c->act = 0;
    c->continuous = 0;
    c->normalize_advantage = 0;
    c->layers = 2;
    return RLHIP_OK;
}

int64_t rlhip_ppo_nparams(int32_t kind, const rlhip_ppo_cfg* c) {
    if (is_layers3(c)) return ppo3_nparams(kind, c);
    PolicyDesc pd;
    if (make_desc(kind, c, &pd)) return -1;
    int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    return pd.np_a + mlp2_nparams(ns, pd.h, 1);
}

int64_t rlhip_mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out) { return mlp2_nparams(n_in, h, n_out); }

int32_t rlhip_mlp2_forward_f32(const float* params, int64_t n_in, int64_t h, int64_t n_out, int32_t act,
                               const float* x, int64_t batch, float* out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && x && out, "NULL array");
    RLHIP_REQUIRE(n_in >= 1 && n_in <= 16 && n_out >= 1 && n_out <= 32 && h >= 1, "unsupported layer sizes");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    if (batch == 0) return RLHIP_OK;
    hipLaunchKernelGGL(mlp2_forward_kernel, dim3((int)((batch + 255) / 256)), dim3(256), 0, as_stream(stream),
                       params, (int)n_in, (int)h, (int)n_out, act, x, batch, out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_mlp2_init_f32(float* params, int64_t n_in, int64_t h, int64_t n_out, uint64_t seed,
                            uint32_t net_id, rlhip_stream_t stream) {
    RLHIP_REQUIRE(params != nullptr && n_in >= 1 && h >= 1 && n_out >= 1, "bad arguments");
    int64_t np = mlp2_nparams(n_in, h, n_out);
    hipLaunchKernelGGL(mlp2_init_kernel, dim3((int)((np + 255) / 256)), dim3(256), 0, as_stream(stream), params,
                       (int)n_in, (int)h, (int)n_out, seed, net_id);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ppo_plan_f32(int32_t kind, const rlhip_ppo_cfg* cfg, const float* params, const float* obs,
                           int64_t n, uint64_t seed, uint32_t env_id_base, uint32_t vec_step,
                           int32_t* action_i, float* action_f, float* logp, float* value,
                           rlhip_stream_t stream) {
    RLHIP_REQUIRE(!is_layers3(cfg), "layers = 3: use the fused rollout (rlhip_ppo_rollout_f32), the per-step plan! is "
                                    "not built for the MFMA actor / critic");
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
    RLHIP_REQUIRE(params && obs && n >= 0, "bad arguments");
    RLHIP_REQUIRE(pd.cont ? (action_f != nullptr) : (action_i != nullptr), "action output is NULL");
    if (n == 0) return RLHIP_OK;
    hipStream_t s = as_stream(stream);
    if (kind == 0) return plan_impl<4>(n, pd, params, obs, seed, env_id_base, vec_step, action_i, action_f, logp, value, s);
    if (kind == 1) return plan_impl<3>(n, pd, params, obs, seed, env_id_base, vec_step, action_i, action_f, logp, value, s);
    return plan_impl<2>(n, pd, params, obs, seed, env_id_base, vec_step, action_i, action_f, logp, value, s);
}

/* The per-step protocol's pushes into slot t of the time-major PPO traces, one launch each (the fused rollout writes the
 * same slots itself).  Agent push protocol: RLCore/src/policies/agent/agent_base.jl:45-59 -- PreActStage: state, action,
 * action_log_prob (+ the critic's value); PostActStage: reward, terminal.  logp == NULL and no action: the bootstrap
 * push of (state, value) into slot T after the last step. */
int32_t rlhip_ppo_push_preact_f32(const rlhip_ppo_traj* traj, int64_t t, int64_t ns, int64_t n, const float* obs,
                                  const float* value, const float* logp, const int32_t* action_i, const float* action_f,
                                  rlhip_stream_t stream) {
    RLHIP_REQUIRE(traj && traj->obs && traj->value && obs && value, "NULL argument");
    RLHIP_REQUIRE(t >= 0 && ns >= 1 && ns <= 8 && n >= 0, "bad slot / shape");
    RLHIP_REQUIRE(!logp || traj->logp, "trajectory has no logp trace");
    RLHIP_REQUIRE((!action_i || traj->action_i) && (!action_f || traj->action_f), "trajectory has no action trace");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(ppo_push_preact_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), TrajPtrs::from(*traj),
                       t, (int)ns, n, obs, value, logp, action_i, action_f);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ppo_push_postact_f32(const rlhip_ppo_traj* traj, int64_t t, int64_t n, const float* reward,
                                   const uint8_t* done, rlhip_stream_t stream) {
    RLHIP_REQUIRE(traj && traj->reward && traj->terminal && reward && done, "NULL argument");
    RLHIP_REQUIRE(t >= 0 && n >= 0, "bad slot / shape");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(ppo_push_postact_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream),
                       TrajPtrs::from(*traj), t, n, reward, done);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

static int32_t rollout_entry(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                             const rlhip_ppo_cfg* cfg, const float* params, uint64_t seed, uint32_t env_id_base,
                             uint32_t vec_step0, const uint32_t* ctr, const rlhip_ppo_traj* traj,
                             rlhip_stream_t stream) {
    if (is_layers3(cfg)) {
        RLHIP_REQUIRE(ctr == nullptr, "layers = 3: the device-counter (graph replay) variant is not built");
        return ppo3_rollout(kind, env_cfg, st, n, T, cfg, params, seed, env_id_base, vec_step0, traj, stream);
    }
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
    RLHIP_REQUIRE(env_cfg && st && params && traj, "NULL argument");
    RLHIP_REQUIRE(n >= 1 && n <= 0x7FFFFFFFll && T >= 0 && T <= 0x7FFFFFFFll, "bad n / T");
    RLHIP_REQUIRE(traj->obs && traj->logp && traj->value && traj->reward && traj->terminal, "trajectory array is NULL");
    RLHIP_REQUIRE(pd.cont ? (traj->action_f != nullptr) : (traj->action_i != nullptr), "action trace is NULL");
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return rollout_impl<CartPoleParams<float>>((const rlhip_cartpole_cfg*)env_cfg, st, n, T, pd, params, seed, env_id_base, vec_step0, ctr, traj, s);
    if (kind == 1)
        return rollout_impl<PendulumParams<float>>((const rlhip_pendulum_cfg*)env_cfg, st, n, T, pd, params, seed, env_id_base, vec_step0, ctr, traj, s);
    return rollout_impl<MountainCarParams<float>>((const rlhip_mountaincar_cfg*)env_cfg, st, n, T, pd, params, seed, env_id_base, vec_step0, ctr, traj, s);
}

int32_t rlhip_ppo_rollout_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                              int64_t T, const rlhip_ppo_cfg* cfg, const float* params, uint64_t seed,
                              uint32_t env_id_base, uint32_t vec_step0, const rlhip_ppo_traj* traj,
                              rlhip_stream_t stream) {
    return rollout_entry(kind, env_cfg, st, n, T, cfg, params, seed, env_id_base, vec_step0, nullptr, traj, stream);
}

int32_t rlhip_ppo_rollout_dc_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                                 int64_t T, const rlhip_ppo_cfg* cfg, const float* params, uint64_t seed,
                                 uint32_t env_id_base, const uint32_t* counters, const rlhip_ppo_traj* traj,
                                 rlhip_stream_t stream) {
    RLHIP_REQUIRE(counters != nullptr, "counters is NULL");
    return rollout_entry(kind, env_cfg, st, n, T, cfg, params, seed, env_id_base, 0, counters, traj, stream);
}

int32_t rlhip_counters_advance(uint32_t* counters, uint32_t d_vec_step, uint32_t d_update, rlhip_stream_t stream) {
    RLHIP_REQUIRE(counters != nullptr, "counters is NULL");
    hipLaunchKernelGGL(counters_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), counters, d_vec_step, d_update);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ppo_gae_f32(const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                          rlhip_stream_t stream) {
    RLHIP_REQUIRE(cfg && traj && traj->adv && traj->ret && traj->reward && traj->value && traj->terminal,
                  "NULL argument");
    return rlhip_gae_returns_f32(traj->adv, traj->ret, traj->reward, traj->value, traj->terminal, n, T,
                                 cfg->gamma, cfg->lambda, stream);
}


}  // extern "C"
