/*
 * rlhip.h -- C ABI of librlhip.so: the MI355X-native rollout + learner hot path for
 * ReinforcementLearning.jl (vectorised classic-control env step -> replay ring push/gather ->
 * GAE / TD target / Huber / Adam -> gradient hand-off to the all-reduce).
 *
 * The reference has NO foreign-function boundary (it is 100 % Julia, zero `ccall`; SURVEY.md
 * section 0), so each entry point below names the Julia function(s) it replaces; the `ccall`
 * stubs a maintainer would add are in INTEGRATION.md and reinforcementlearning.jl_amd/julia/RLHip.jl.
 * Reference paths are relative to /root/reference/src/ :
 *   RLEnvs = ReinforcementLearningEnvironments/src/environments/examples
 *   RLCore = ReinforcementLearningCore/src
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs.  No torch / HIP types in any signature
 *     (a stream is an opaque `void*` holding a hipStream_t; NULL = the default stream).
 *   - every function returns int32: 0 = RLHIP_OK, < 0 = error; message via rlhip_last_error().
 *   - all data pointers are DEVICE pointers unless the name ends in `_host`.  Calls enqueue on the
 *     given stream and return immediately (no implicit sync) unless documented otherwise.
 *   - indices are 0-based (Julia glue adds/subtracts 1: actions 1..na <-> 0..na-1).
 *   - layouts are SoA: a per-env quantity q with D components is stored as q[d * n + i]
 *     (component-major, env index i contiguous) so that one wavefront lane per env reads coalesced.
 *     Time-major trajectories: x[(t * D + d) * n + i].
 *   - random draws follow the Philox4x32-10 specification in DESIGN.md (ctr = {idx, blk, t, tag},
 *     key = seed); `env_id_base` offsets idx so shards on different GPUs own disjoint streams and
 *     results do not depend on the number of GPUs.
 */
#ifndef RLHIP_H
#define RLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLHIP_OK 0
#define RLHIP_EINVAL (-1) /* bad argument (the reference would throw AssertionError / ArgumentError / MethodError) */
#define RLHIP_EHIP (-2)   /* a HIP runtime call failed */
#define RLHIP_ENODEV (-3) /* no gfx950 device visible */
#define RLHIP_ETIMEOUT (-4) /* a peer never arrived at a gradient exchange; the result was overwritten with NaN */
#define RLHIP_ECOMM (-5)    /* RCCL is missing or one of its calls failed / no transport for this exchange */

/* 2 (round 5): rlhip_ring rings of Float32 observations with <= 4 components store 64-byte transition records (s, a, r, t, s'
 *    per (slot, env)); `rlhip_ring.layout` names the layout, rlhip_ring_init takes NULL action / reward / terminal for
 *    them; rlhip_ppo_workspace_init + the capacity check of rlhip_ppo_update_f32; rlhip_eps_greedy_prob_f32.  ABI 1 stored those
 *    states transition-major with three separate traces (and round 3 component-major): a host built against an older header
 *    fails rlhip_abi_version() == RLHIP_ABI_VERSION and, if it skips that check, rlhip_ring_init (RLHIP_EINVAL). */
#define RLHIP_ABI_VERSION 2

typedef void* rlhip_stream_t; /* hipStream_t */
typedef void* rlhip_event_t;  /* hipEvent_t  */

/* ------------------------------------------------------------------------------ runtime -- */
int32_t rlhip_abi_version(void);
const char* rlhip_last_error(void);
int32_t rlhip_device_count(int32_t* n_out);
int32_t rlhip_set_device(int32_t device);
/* device arch name, e.g. "gfx950:sramecc+:xnack-" (host buffer) */
int32_t rlhip_device_name(int32_t device, char* name_host, int32_t cap);

/* Device memory / streams / events for hosts that own no GPU allocator (the Julia glue).  A host
 * that already has one (PyTorch-ROCm here) passes its own pointers and stream instead. */
int32_t rlhip_malloc(void** ptr_out, size_t bytes);
int32_t rlhip_free(void* ptr);
int32_t rlhip_memset(void* ptr, int32_t value, size_t bytes, rlhip_stream_t stream);
int32_t rlhip_memcpy_h2d(void* dst, const void* src_host, size_t bytes, rlhip_stream_t stream); /* syncs stream */
int32_t rlhip_memcpy_d2h(void* dst_host, const void* src, size_t bytes, rlhip_stream_t stream); /* syncs stream */
int32_t rlhip_memcpy_d2d(void* dst, const void* src, size_t bytes, rlhip_stream_t stream);
int32_t rlhip_stream_create(rlhip_stream_t* stream_out);
int32_t rlhip_stream_destroy(rlhip_stream_t stream);
int32_t rlhip_stream_sync(rlhip_stream_t stream);
int32_t rlhip_event_create(rlhip_event_t* event_out);
int32_t rlhip_event_destroy(rlhip_event_t event);
int32_t rlhip_event_record(rlhip_event_t event, rlhip_stream_t stream);
/* syncs on `stop`, returns milliseconds between the two recorded events */
int32_t rlhip_event_elapsed_ms(rlhip_event_t start, rlhip_event_t stop, float* ms_out);

/* Philox uniforms U[0,1) (24-bit, Float32) -- synthetic data / weight init helper.
 * out[i] = u01(word (i % 4) of Philox(seed, idx = i / 4, blk = 0, t, tag)). */
int32_t rlhip_fill_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t t, uint32_t tag,
                               rlhip_stream_t stream);
/* keyed permutation of [0, n) (the `shuffle(rng, 1:n)` stand-in): out[i] = perm_epoch(i) */
int32_t rlhip_permutation(uint32_t* out, uint32_t n, uint64_t seed, uint32_t epoch,
                          rlhip_stream_t stream);

/* --------------------------------------------------------------------------------- envs -- */
#define RLHIP_ENV_CARTPOLE 0    /* RLEnvs/CartPoleEnv.jl */
#define RLHIP_ENV_PENDULUM 1    /* RLEnvs/PendulumEnv.jl */
#define RLHIP_ENV_MOUNTAINCAR 2 /* RLEnvs/MountainCarEnv.jl */
#define RLHIP_ENV_ACROBOT 3     /* RLEnvs/src/environments/3rd_party/AcrobotEnv.jl (stand-alone env kernels only) */

/* CartPoleEnv(; kwargs...)  RLEnvs/CartPoleEnv.jl:22-32,74-79 -- Float64 as typed by the caller */
typedef struct {
    double gravity, masscart, masspole, halflength, forcemag, dt, thetathreshold_deg, xthreshold;
    int64_t max_steps;
    int32_t continuous; /* 0: action int32 in {0,1} (Julia 1,2); 1: action of type T in [-1,1] */
} rlhip_cartpole_cfg;

/* PendulumEnv(; kwargs...)  RLEnvs/PendulumEnv.jl:41-53 */
typedef struct {
    double max_speed, max_torque, g, m, l, dt;
    int64_t max_steps;
    int32_t continuous; /* 1: torque of type T; 0: int32 index in 0..n_actions-1 */
    int32_t n_actions;
} rlhip_pendulum_cfg;

/* MountainCarEnv(; kwargs...)  RLEnvs/MountainCarEnv.jl:19-40,67-81 */
typedef struct {
    double min_pos, max_pos, max_speed, goal_pos, goal_velocity, power, gravity;
    int64_t max_steps;
    int32_t continuous;
} rlhip_mountaincar_cfg;

/* AcrobotEnv(; kwargs...)  3rd_party/AcrobotEnv.jl:22-40.  PARITY UNPINNED: the reference integrates one act! with
 * OrdinaryDiffEq.solve(ode, RK4()) (:128-129), an un-vendored adaptive driver whose step-size controller decides the
 * arithmetic; this library takes ONE classic RK4 step of length dt over the reference's dsdt (:147-199) in Float64
 * (the "python gym" scheme the file cites), wraps / bounds (:135-138) and stores the state as T.  Discrete actions
 * 0..2 (torque -1, 0, +1); max_torque_noise > 0 draws one uniform per act! from the Philox stream ENVNOISE.
 * reward(env) is -1 after reset! (:99). */
typedef struct {
    double link_length_a, link_length_b, link_mass_a, link_mass_b, link_com_pos_a, link_com_pos_b, link_moi,
        max_torque_noise, max_vel_a, max_vel_b, g, dt;
    int64_t max_steps;
    int32_t nips; /* book_or_nips: 0 "book" (default), 1 "nips" */
} rlhip_acrobot_cfg;

int32_t rlhip_cartpole_default(rlhip_cartpole_cfg* cfg_host);
int32_t rlhip_acrobot_default(rlhip_acrobot_cfg* cfg_host);
int32_t rlhip_pendulum_default(rlhip_pendulum_cfg* cfg_host);
int32_t rlhip_mountaincar_default(rlhip_mountaincar_cfg* cfg_host, int32_t continuous);

/* SoA state of n env instances (device arrays, caller-owned).
 *   s[k]    T[n]   state component k (cartpole: x, xdot, theta, thetadot; pendulum: theta, thetadot;
 *                  mountaincar: x, v; acrobot: theta1, theta2, dtheta1, dtheta2)
 *   t       i32[n] step counter of the running episode
 *   done    u8[n]  is_terminated(env) after the LAST act!
 *   reward  T[n]   reward(env) after the last act!
 *   episode u32[n] number of resets so far (Philox time counter of the next reset)
 * PACKED MODE (episode == NULL; rlhip_env_reset / rlhip_env_step / rlhip_env_obs only): the reset counter lives in
 * the bits of t[i] that max_steps leaves free -- t[i] = step | episode << tbits, tbits = the smallest b >= 1 with
 * 2^b > max_steps + 1 (CartPole default 200: 8 bits of step, 24 bits = 16.7 M episodes; the counter SATURATES there --
 * rlhip_env_packed_episode_capacity -- so a host that may step an instance that often must keep episode[]).
 * Same states, rewards, flags and reset draws as the separate array; an auto-reset then touches no memory that the
 * step kernel does not stream anyway (with episode[] every reset is a scattered 4-byte read-modify-write: +9 % HBM
 * traffic at 2^24 CartPole envs under a random policy).  max_steps < 2^20 - 1.                  */
typedef struct {
    void* s[4];
    int32_t* t;
    uint8_t* done;
    void* reward;
    uint32_t* episode;
} rlhip_env_state;

int64_t rlhip_env_packed_episode_capacity(int64_t max_steps); /* 2^(32 - tbits) - 1, or -1 (max_steps out of range) */
int32_t rlhip_env_obs_dim(int32_t kind);   /* cartpole 4, pendulum 3 (sin, cos, thetadot), mountaincar 2, acrobot 6 */
int32_t rlhip_env_state_dim(int32_t kind); /* cartpole 4, pendulum 2, mountaincar 2, acrobot 4 */

/* reset!(env)   CartPoleEnv.jl:98-104, PendulumEnv.jl:84-92, MountainCarEnv.jl:99-105.
 * mask == NULL: reset every env (reset!(env; is_force = true) of the vector env); otherwise only
 * where mask[i] != 0 (reset!(env) of the vector env = "reset the terminated ones", pass st->done). */
int32_t rlhip_env_reset(int32_t kind, int32_t is_f64, const void* cfg_host,
                        const rlhip_env_state* st_host, int64_t n, uint64_t seed,
                        uint32_t env_id_base, const uint8_t* mask, rlhip_stream_t stream);

/* act!(env, a) = _step! + reward + is_terminated for every env in one launch
 *   CartPoleEnv.jl:106-140,:84-85; PendulumEnv.jl:94-122; MountainCarEnv.jl:107-135,:95.
 * actions: i32[n] (discrete, 0-based) or T[n] (continuous).
 * auto_reset != 0: the MultiThreadEnv protocol -- reward/done of the finished step stay visible and a
 *   terminated env immediately starts a fresh episode inside the same kernel.
 * last_obs (nullable): T[obs_dim * n] receives the observation BEFORE the auto-reset.
 * obs_out  (nullable): T[obs_dim * n] receives state(env) AFTER the step (and auto-reset).       */
int32_t rlhip_env_step(int32_t kind, int32_t is_f64, const void* cfg_host,
                       const rlhip_env_state* st_host, int64_t n, const void* actions,
                       int32_t auto_reset, uint64_t seed, uint32_t env_id_base, void* last_obs,
                       void* obs_out, rlhip_stream_t stream);

/* state(env)  CartPoleEnv.jl:86, PendulumEnv.jl:70,82, MountainCarEnv.jl:97 -> T[obs_dim * n] */
int32_t rlhip_env_obs(int32_t kind, int32_t is_f64, const rlhip_env_state* st_host, int64_t n,
                      void* obs_out, rlhip_stream_t stream);

/* -------------------------------------------------------------------------------- scans -- */
/* discount_rewards / discount_rewards_reduced / generalized_advantage_estimation
 *   RLCore/utils/basic.jl:138-235, :237-319, :334-417.
 * Matrices are column-major n1 x n2 like Julia.  dims = 0: vector (n2 must be 1);
 * dims = 1 / 2: the Julia `dims` keyword = the axis the scan runs along.  A matrix with dims = 0 is
 * RLHIP_EINVAL (the reference throws MethodError).  terminal / init may be NULL (`nothing`).
 * values has one more entry than rewards along the scan axis.
 * dims = 2 is the coalesced PPO layout (rewards (n_env, T)): one lane per env, T serial.        */
int32_t rlhip_discount_rewards_f32(float* out, const float* rewards, int64_t n1, int64_t n2,
                                   float gamma, const uint8_t* terminal, const float* init,
                                   int32_t dims, rlhip_stream_t stream);
int32_t rlhip_discount_rewards_f64(double* out, const double* rewards, int64_t n1, int64_t n2,
                                   double gamma, const uint8_t* terminal, const double* init,
                                   int32_t dims, rlhip_stream_t stream);
int32_t rlhip_discount_rewards_reduced_f32(float* out, const float* rewards, int64_t n1, int64_t n2,
                                           float gamma, const uint8_t* terminal, const float* init,
                                           int32_t dims, rlhip_stream_t stream);
int32_t rlhip_discount_rewards_reduced_f64(double* out, const double* rewards, int64_t n1,
                                           int64_t n2, double gamma, const uint8_t* terminal,
                                           const double* init, int32_t dims, rlhip_stream_t stream);
int32_t rlhip_gae_f32(float* advantages, const float* rewards, const float* values, int64_t n1,
                      int64_t n2, float gamma, float lambda, const uint8_t* terminal, int32_t dims,
                      rlhip_stream_t stream);
int32_t rlhip_gae_f64(double* advantages, const double* rewards, const double* values, int64_t n1,
                      int64_t n2, double gamma, double lambda, const uint8_t* terminal, int32_t dims,
                      rlhip_stream_t stream);
/* PPO fusion: advantages AND returns = advantages + values[:, 1:T] in one pass over (n_env, T)
 * time-major arrays (dims = 2 semantics). returns may be NULL. */
int32_t rlhip_gae_returns_f32(float* advantages, float* returns, const float* rewards,
                              const float* values, const uint8_t* terminal, int64_t n_env, int64_t T,
                              float gamma, float lambda, rlhip_stream_t stream);

/* ---------------------------------------------------------------------------- selection -- */
/* element (k, i) of a (na x n) value array = values[k * k_stride + i * i_stride]:
 *   Julia (na, N) column-major: k_stride = 1, i_stride = na;  SoA: k_stride = n, i_stride = 1.  */

/* plan!(::EpsilonGreedyExplorer, values[, mask]) for n envs
 *   RLCore/policies/explorers/epsilon_greedy_explorer.jl:102-131 (+ findmax / find_all_max
 *   RLCore/utils/basic.jl:91-120); eps from rlhip_get_eps; GreedyExplorer (:200-205) = eps 0.
 * mask (nullable) u8, same strides as values.  actions: i32[n], 0-based.
 * draws: Philox(seed, idx = env_id_base + i, blk 0, t = step, EXPLORE): (w0,w1) -> u, w2 -> random index,
 * w3 -> tie-break index. */
int32_t rlhip_eps_greedy_select_f32(const float* values, int64_t na, int64_t n, int64_t k_stride,
                                    int64_t i_stride, const uint8_t* mask, double eps,
                                    int32_t is_break_tie, uint64_t seed, uint32_t env_id_base,
                                    uint32_t step, int32_t* actions, rlhip_stream_t stream);
/* prob(::EpsilonGreedyExplorer, values[, mask]) for n envs  epsilon_greedy_explorer.jl:141-194 (the vector the reference wraps
 * in Categorical(probs; check_args = false); pinned by RLCore/test/policies/explorers/epsilon_greedy_explorer.jl:45-73):
 *   probs[k] = eps / n_legal on legal actions, 0 on masked ones; + (1 - eps) on findmax(values[, mask]) (is_break_tie = 0),
 *   or + (1 - eps) / c on each of the c entries of find_all_max(values[, mask]) (is_break_tie = 1).
 * probs: f64, same (k_stride, i_stride) addressing as values.  Float64 like the reference (eps is a Float64). */
int32_t rlhip_eps_greedy_prob_f32(const float* values, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride,
                                  const uint8_t* mask, double eps, int32_t is_break_tie, double* probs,
                                  rlhip_stream_t stream);
/* get_eps  epsilon_greedy_explorer.jl:69-88 (host-side scalar; kind 0 = linear, 1 = exp) */
double rlhip_get_eps(int32_t kind, double eps_stable, double eps_init, int64_t warmup_steps,
                     int64_t decay_steps, int64_t step);
/* sample_categorical (Gumbel-max)  RLCore/utils/networks.jl:425-432, masking :466-468.
 * logp_out (nullable) f32[n] = logsoftmax(logits)[action]. */
int32_t rlhip_categorical_sample_f32(const float* logits, int64_t na, int64_t n, int64_t k_stride,
                                     int64_t i_stride, const uint8_t* mask, uint64_t seed,
                                     uint32_t env_id_base, uint32_t step, int32_t* actions,
                                     float* logp_out, rlhip_stream_t stream);

/* (net::CategoricalNetwork)(state[, mask]; is_sampling, is_return_log_prob)  RLCore/utils/networks.jl:405-432, masked
 * methods :459-472, on (na, n) component-major logits in one launch: masked_logits_out (nullable) = logits +
 * ifelse(mask, 0, typemin) (:461; mask NULL = all true); actions (nullable) = the Gumbel-max draw from them (0-based, same
 * draws as rlhip_categorical_sample_f32); onehot_out (nullable, needs actions) = Flux.onehotbatch(draw, 1:na). */
int32_t rlhip_categorical_network_f32(const float* logits, int64_t na, int64_t n, const uint8_t* mask, uint64_t seed,
                                      uint32_t env_id_base, uint32_t step, float* masked_logits_out, int32_t* actions,
                                      float* onehot_out, rlhip_stream_t stream);

/* The remaining explorers as batched kernels (BatchExplorer semantics: the inner explorer applied to each column,
 * RLCore/src/policies/explorers/batch_explorer.jl:14-21).  values / mask addressing as rlhip_eps_greedy_select.
 *   kind 0  WeightedExplorer{is_normalized}   weighted_explorer.jl:19-33  (mask: weight 0)
 *   kind 1  WeightedSoftmaxExplorer           weighted_softmax_explorer.jl:21-27  (mask: typemin)
 *   kind 2  GumbelSoftmaxExplorer             gumbel_softmax_explorer.jl:11-24  (Float32 Gumbel noise)
 * Draws: Philox(seed, idx = env_id_base + i, t = step, EXPLORE / GUMBEL); na <= 64. */
int32_t rlhip_explorer_select_f32(int32_t kind, const float* values, int64_t na, int64_t n, int64_t k_stride,
                                  int64_t i_stride, const uint8_t* mask, int32_t is_normalized, uint64_t seed,
                                  uint32_t env_id_base, uint32_t step, int32_t* actions, rlhip_stream_t stream);
/* UCBExplorer  UCB_explorer.jl:24-30: argmax of values + c sqrt(log(step + 1) / counts) with a uniform pick among
 * ties; action_counts: f64 (na, n) device, counts[k * n + i], initialised to eps (1e-10), incremented here. */
int32_t rlhip_ucb_select_f32(const float* values, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride,
                             double c, double* action_counts, int64_t step, uint64_t seed, uint32_t env_id_base,
                             int32_t* actions, rlhip_stream_t stream);

/* ---------------------------------------------------------------------- parameter updates -- */
/* TargetNetwork sync: dest = rho * dest + (1 - rho) * src   target_network.jl:76-85 (rho = 0: hard copy) */
int32_t rlhip_polyak_f32(float* dst, const float* src, int64_t n, float rho, rlhip_stream_t stream);
/* clip_by_global_norm!(gs, ps, clip_norm)  RLCore/utils/basic.jl:19-29 over one flat gradient.
 * gn_out: f32[1] device (the returned norm). */
int32_t rlhip_clip_by_global_norm_f32(float* grad, int64_t n, float clip_norm, float* gn_out,
                                      rlhip_stream_t stream);
/* Flux.Optimise.update!(opt_state, model, grad)  flux_approximator.jl:46 with Optimisers.Adam.
 * beta_pow: f32[2] device = running (beta1^t, beta2^t), initialised to (beta1, beta2); updated here. */
int32_t rlhip_adam_f32(float* params, const float* grad, float* m, float* v, float* beta_pow,
                       int64_t n, float lr, float beta1, float beta2, float eps,
                       rlhip_stream_t stream);
/* fused: [global norm -> clip] -> Adam in ONE single-workgroup launch (n <= ~1M).  clip_norm <= 0
 * disables clipping.  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow,
                            int64_t n, float grad_scale, float clip_norm, float lr, float beta1,
                            float beta2, float eps, float* gn_out, rlhip_stream_t stream);
/* normlogpdf / diagnormlogpdf  RLCore/utils/distributions.jl:18-21, :31-34 (eps = 1f-8).
 * diag: arrays (d x n) column-major, out f32[n]. */
int32_t rlhip_normlogpdf_f32(const float* mu, const float* sigma, const float* x, float* out,
                             int64_t n, rlhip_stream_t stream);
int32_t rlhip_diagnormlogpdf_f32(const float* mu, const float* sigma, const float* x, int64_t d,
                                 int64_t n, float* out, rlhip_stream_t stream);
/* Stochastic Gaussian policy heads  RLCore/src/utils/networks.jl: GaussianNetwork :64-116 (logpdfcorrection /
 * inversesquash :39-42), SoftGaussianNetwork :147-198.  mu, raw_sigma: outputs of the mu / sigma sub-networks,
 * f32 (d x n) column-major; sigma = clamp(raw_sigma, min_sigma, max_sigma).  K samples per state
 * (K = 1: the `is_sampling = true` call; K > 1: the `(state, action_samples::Int)` call): z = mu + sigma * noise with
 * noise from the Philox NORMAL stream (draw k + d*j of (seed, env_id_base + i, step)).
 *   squash 0 identity / 1 tanh (GaussianNetwork.squash); soft 1 = SoftGaussianNetwork (tanh, its own logp form)
 *   action_out f32 (d x K x n) = squash(z); logp_out f32 (K x n), nullable (`is_return_log_prob = false`)
 * rlhip_gaussian_head_logp_f32 is the `(model)(state, action)` call: log-probability of given (squashed) actions. */
int32_t rlhip_gaussian_head_sample_f32(const float* mu, const float* raw_sigma, int64_t d, int64_t n, int64_t K,
                                       float min_sigma, float max_sigma, int32_t squash, int32_t soft,
                                       uint64_t seed, uint32_t env_id_base, uint32_t step, float* action_out,
                                       float* logp_out, rlhip_stream_t stream);
int32_t rlhip_gaussian_head_logp_f32(const float* mu, const float* raw_sigma, const float* action, int64_t d,
                                     int64_t n, int64_t K, float min_sigma, float max_sigma, int32_t squash,
                                     int32_t soft, float* logp_out, rlhip_stream_t stream);
/* Flux.Losses.huber_loss(q, target; delta) mean-aggregated -> loss_out f32[1]; dq (nullable) = dL/dq */
int32_t rlhip_huber_f32(const float* q, const float* target, int64_t n, float delta, float* loss_out,
                        float* dq, rlhip_stream_t stream);
/* DQN target G = r + gamma * (1 - terminal) * max_a' Qt(s', a')  (qt_next strides as in selection) */
int32_t rlhip_td_target_f32(const float* qt_next, int64_t na, int64_t n, int64_t k_stride,
                            int64_t i_stride, const float* reward, const uint8_t* terminal,
                            float gamma, float* target, rlhip_stream_t stream);

/* n-step form (R = r + gamma^n (1 - t) max_a' Qt(s_{i+n}, a')): reward = the n-step returns, terminal = any terminal
 * inside the window (both from rlhip_ring_fold_nstep), gamma^n = rlhip_gamma_pow(gamma, n_step) */
int32_t rlhip_td_target_n_f32(const float* qt_next, int64_t na, int64_t n, int64_t k_stride,
                              int64_t i_stride, const float* reward, const uint8_t* terminal,
                              float gamma, int32_t n_step, float* target, rlhip_stream_t stream);
/* gamma^n as the n-step learners use it: evaluated in Float64 and rounded once (Julia's `gamma^n` for a Float32 gamma) */
float rlhip_gamma_pow(float gamma, int32_t n);

/* -------------------------------------------------------------------------- replay ring -- */
/* CircularArraySARTSTraces(; capacity, state = Float32 => (obs_dim, n_env), action = Int32 => (n_env,),
 * reward = Float32 => (n_env,), terminal = Bool => (n_env,)) resident in HBM
 *   (un-vendored ReinforcementLearningTrajectories 0.4; call sites RLCore/policies/agent/agent_base.jl:45-59,
 *    RLCore/test/policies/q_based_policy.jl:41-47).
 * One frame = one vec-step.  state has capacity+1 frames (next_state[i] = state[i+1]), the other traces
 * capacity frames.  Head/length counters are HOST fields updated by the push calls (they are pure
 * functions of the push count, so no device sync is ever needed to know them).
 * elem_bytes: 4 (Float32 observations) or 1 (UInt8 frames, e.g. 84x84x4 Atari stacks).
 *
 * Two storage layouts (`layout`, set by rlhip_ring_init; rlhip_ring_layout() returns it):
 *   RLHIP_RING_FRAMES   every trace as pushed: state[(slot * obs_dim + k) * n_env + e] (capacity + 1 slots), action / reward /
 *                       terminal[slot * n_env + e] (capacity slots) -- UInt8 frames and Float32 observations with > 4 components.
 *   RLHIP_RING_RECORDS  Float32 observations with <= 4 components (the classic-control envs; what the fused DQN learners take):
 *                       `state` holds (capacity + 1) * n_env RECORDS of 64 bytes (one cache line = one fabric request),
 *                           record[slot * n_env + e] = { float s[4]; int32 action; float reward; uint32 terminal; uint32 spare;
 *                                                        float s_next[4]; uint32 pad[4] },
 *                       the whole transition (s, a, r, t, s') that LEAVES the state of that slot: a sampled transition is one
 *                       line (round 4: five; measured in csrc/ring_device.h).  push!(trajectory, (state = s', action, reward,
 *                       terminal)) completes the previous slot's record and opens the next (s = s': a state is stored twice).
 *                       Logical transition i lives in slot (head_sa + i) mod (capacity + 1); the newest slot holds a state only.
 *                       `action`, `reward`, `terminal` are NULL (a strided view for a host: words 4 / 5 / 6 of each 16-word
 *                       record, s_next at words 8..11).  s[k >= obs_dim] = s_next[k >= obs_dim] = 0.
 * rlhip_ring_state_bytes() is the size of the `state` allocation for either layout (64-byte aligned for records). */
#define RLHIP_RING_FRAMES 0
#define RLHIP_RING_RECORDS 2 /* (1 was ABI 1's transition-major state trace) */
typedef struct {
    int64_t capacity, n_env, obs_dim;
    int64_t head_sa, len_sa, head_rt, len_rt; /* host-side ring counters (head_rt / len_rt count the LOGICAL action / reward /
                                               * terminal traces in both layouts: lengths, sampler range, sum-tree keys) */
    int32_t elem_bytes;
    int32_t layout;    /* RLHIP_RING_FRAMES | RLHIP_RING_RECORDS */
    void* state;       /* FRAMES: (capacity + 1) * obs_dim * n_env elements; RECORDS: (capacity + 1) * n_env * 64 bytes */
    int32_t* action;   /* FRAMES: capacity * n_env; RECORDS: NULL */
    float* reward;     /* FRAMES: capacity * n_env; RECORDS: NULL */
    uint8_t* terminal; /* FRAMES: capacity * n_env; RECORDS: NULL */
} rlhip_ring;

int64_t rlhip_ring_state_bytes(int64_t capacity, int64_t n_env, int64_t obs_dim, int32_t elem_bytes);
int32_t rlhip_ring_layout(const rlhip_ring* rb_host);
/* action / reward / terminal: device arrays for RLHIP_RING_FRAMES, NULL for RLHIP_RING_RECORDS (= elem_bytes 4 and obs_dim <= 4;
 * anything else is RLHIP_EINVAL: a host written for ABI 1 fails here instead of reading transposed data) */
int32_t rlhip_ring_init(rlhip_ring* rb_host, int64_t capacity, int64_t n_env, int64_t obs_dim,
                        int32_t elem_bytes, void* state, int32_t* action, float* reward,
                        uint8_t* terminal);
/* push!(trajectory, (state = s,))   Agent PreEpisodeStage  agent_base.jl:45-47.
 * Protocol (checked before any counter of rb_host moves: a rejected call leaves the ring unchanged): the FIRST push is a state, every
 * later push a transition -- rlhip_ring_push_transition[_maxpool] without an open state and rlhip_ring_push_state[_maxpool] while
 * one is open (len_sa == len_rt + 1) return RLHIP_EINVAL.  The reference's EpisodesBuffer accepts a second PreEpisodeStage push by
 * padding the (a, r, t) slot between the two episodes and excluding it from sampling; this ring has no such mask.  Vector envs
 * auto-reset (one PreEpisode push per run); a single-instance host pushes the post-reset observation as s' of the terminal
 * transition (which `terminal = 1` cuts from the TD target and from stacked histories). */
int32_t rlhip_ring_push_state(rlhip_ring* rb_host, const void* obs, rlhip_stream_t stream);
/* push!(trajectory, (state = s', action = a, reward = r, terminal = t))   PostActStage  :56-59 */
int32_t rlhip_ring_push_transition(rlhip_ring* rb_host, const void* next_obs, const int32_t* action,
                                   const float* reward, const uint8_t* terminal,
                                   rlhip_stream_t stream);
/* length(trajectory.container) */
int64_t rlhip_ring_length(const rlhip_ring* rb_host);
/* BatchSampler(batch): flat indices j = frame_logical * n_env + env, uniform with replacement.
 * idx_out: i64[batch] device.  Philox(seed, idx = b, blk 0, t = draw_ctr, SAMPLER): ((w0:w1) * total) >> 64 */
int32_t rlhip_ring_sample_indices(const rlhip_ring* rb_host, int64_t batch, uint64_t seed,
                                  uint32_t draw_ctr, int64_t* idx_out, rlhip_stream_t stream);
/* `for batch in trajectory`: gather (s, a, r, t, s') for flat indices.  Outputs SoA: s, s_next
 * (obs_dim x batch) elements; a i32[batch]; r f32[batch]; term u8[batch].  LDS-staged index tile. */
/* Output layout of s / s_next: component-major SoA (obs_dim x batch) in general; FRAME-major
 * (batch x obs_dim, every frame contiguous) when rlhip_ring_gather_is_frame_major() != 0, i.e. for
 * n_env == 1 rings whose frame is >= 1024 bytes and a multiple of 16 bytes (image observations). */
int32_t rlhip_ring_gather_is_frame_major(const rlhip_ring* rb_host);
int32_t rlhip_ring_gather(const rlhip_ring* rb_host, const int64_t* idx, int64_t batch, void* s,
                          int32_t* a, float* r, uint8_t* term, void* s_next,
                          rlhip_stream_t stream);

/* n-step transitions -- NStepBatchSampler(n, gamma, batchsize) of RLTrajectories 0.4 (un-vendored: PARITY UNPINNED; the published
 * algorithm is restated in oracle/rlo_buffer.c).  Record rings only (Float32 observations with <= 4 components).
 *   rlhip_ring_sample_indices_nstep   inds = rand(rng, 1:(length - n + 1), batchsize) per env: flat logical START indices
 *                                     (li * n_env + e with li <= length - n_step), same Philox draw as rlhip_ring_sample_indices.
 *   rlhip_ring_fold_nstep             for every start index the window li .. li + ns - 1 (ns = n_step, or up to and including the
 *                                     first terminal step) folded into ONE transition {s_li, a_li, R, any(terminal), s_{li + ns}},
 *                                     R = discount_rewards_reduced(rewards[window], gamma) (RLCore/src/utils/basic.jl:237-319:
 *                                     r_0 + gamma (r_1 + gamma (...)), Float32), written as the `batch` records of slot 0 of `folded`
 *                                     -- a record ring from rlhip_ring_init(capacity >= 1, n_env = batch, obs_dim) whose host
 *                                     counters are set to "one stored vec-step".  Every DQN gradient entry point then runs unchanged:
 *                                         rlhip_dqn_grad_idx_f32(folded, ..., idx = iota_out, gamma = rlhip_gamma_pow(gamma, n_step), ...)
 *                                     (likewise _idx_w_, rlhip_dqn3_grad[_w]_f32 with idx).  iota_out (nullable): 0 .. batch - 1.
 *                                     n_step in 1..32; n_step = 1 reproduces the stored transitions bit for bit. */
int32_t rlhip_ring_sample_indices_nstep(const rlhip_ring* rb_host, int64_t batch, int32_t n_step, uint64_t seed,
                                        uint32_t draw_ctr, int64_t* idx_out, rlhip_stream_t stream);
int32_t rlhip_ring_fold_nstep(const rlhip_ring* rb_host, const int64_t* idx, int64_t batch, int32_t n_step, float gamma,
                              rlhip_ring* folded_host, int64_t* iota_out, rlhip_stream_t stream);

/* Debugging aid ("a debug build that bounds-checks gather indices"; the reference's `traces[inds]` throws a
 * BoundsError): how many of the flat logical indices idx[0 .. batch) lie outside [0, length(trajectory) * n_env), and the position
 * of the first one (-1 if none).  One launch and a stream synchronisation: n_bad / first_bad are HOST pointers (first_bad may be
 * NULL).  A library built with -DRLHIP_BOUNDS_CHECK (RLHIP_EXTRA_FLAGS=-DRLHIP_BOUNDS_CHECK python .../build.py --force;
 * rlhip_ring_bounds_checked_build() = 1) runs this check inside every entry point that takes caller-supplied indices
 * (rlhip_ring_gather, rlhip_ring_gather_stacked, rlhip_dqn_grad_idx[_w]_f32, rlhip_dqn3_grad[_w]_f32 with idx) and returns
 * RLHIP_EINVAL instead of reading outside the ring; the default build does not pay the synchronisation. */
int32_t rlhip_ring_check_indices(const rlhip_ring* rb_host, const int64_t* idx, int64_t batch, int64_t* n_bad,
                                 int64_t* first_bad, rlhip_stream_t stream);
int32_t rlhip_ring_bounds_checked_build(void);

/* ------------------------------------------------------- device-side episode hooks -- */
/* TotalRewardPerEpisode / BatchStepsPerEpisode / StepsPerEpisode (RLCore/src/core/hooks.jl:64-101, 146-196,
 * 202-231) without a per-step host read: one launch per vec-step (PostActStage) updates the per-instance
 * (steps, return) accumulators and appends one record per finished episode to a device log.
 * log_count keeps counting past log_capacity (records beyond it are dropped; the host can detect that). */
typedef struct {
    uint32_t vec_step; /* the vec-step in which the episode ended */
    uint32_t env;      /* instance index */
    int32_t steps;     /* episode length */
    int32_t pad;
    double total_reward;
} rlhip_episode_record;
int32_t rlhip_hook_episode_stats(const float* reward, const uint8_t* done, int64_t n, uint32_t vec_step,
                                 int32_t* steps_acc, double* return_acc, rlhip_episode_record* log,
                                 uint32_t log_capacity, uint32_t* log_count, rlhip_stream_t stream);

/* --------------------------------------- frame stacking at sample time, max-pool push -- */
/* StackFrames (RLCore/src/utils/stack_frames.jl:11-44) moved from the way in to the way out: the ring stores
 * single frames (n_env == 1), the gather assembles the n_stack-deep observations
 *   state stack of transition i = frames i-n+1 .. i,   next stack = i-n+2 .. i+1   (oldest first, newest last),
 * with all-zero frames before the episode start / before the oldest stored frame (what StackFrames holds after
 * reset!).  Outputs (batch, n_stack, frame) contiguous; a, r, term as rlhip_ring_gather.  n_stack <= 8. */
int32_t rlhip_ring_gather_stacked(const rlhip_ring* rb_host, const int64_t* idx, int64_t batch, int32_t n_stack,
                                  void* s, int32_t* a, float* r, uint8_t* term, void* s_next,
                                  rlhip_stream_t stream);
/* AtariEnv.act! 2-frame max-pool `screens[1] .= max.(screens[1], screens[2])`
 * (RLEnvs/src/environments/3rd_party/atari.jl:104-107) fused into the push of a UInt8 frame. */
int32_t rlhip_ring_push_state_maxpool(rlhip_ring* rb_host, const void* screen1, const void* screen2,
                                      rlhip_stream_t stream);
int32_t rlhip_ring_push_transition_maxpool(rlhip_ring* rb_host, const void* screen1, const void* screen2,
                                           const int32_t* action, const float* reward,
                                           const uint8_t* terminal, rlhip_stream_t stream);

/* plan! + act! + push! of one DQN vec-step in ONE launch for the 2-layer Q-network (dqn_act.hip): bit-identical
 * to rlhip_dqn_plan_f32 -> rlhip_env_step (auto-reset) -> rlhip_ring_push_transition, whose ring counters it
 * advances the same way.  rlhip_dqn_act_supported: hidden in {64, 128, 256}, n <= 2^18, Float32 discrete env. */
int32_t rlhip_dqn_act_supported(int32_t kind, int64_t n, int64_t h);
int32_t rlhip_dqn_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                          const float* params, int64_t h, int64_t na, int32_t act, double eps,
                          uint64_t explorer_seed, uint32_t explorer_step, uint64_t env_seed, uint32_t env_id_base,
                          rlhip_ring* rb_host, int32_t* actions, float* q_out, float* obs_out, float* last_obs,
                          rlhip_stream_t stream);
/* act! + push! in one launch for a policy whose plan! ran separately (the MFMA Q-network): actions i32[n] 0-based.
 * Bit-identical to rlhip_env_step(auto_reset = 1, last_obs, obs_out) followed by rlhip_ring_push_transition; the ring
 * counters advance as in that call.  Discrete Float32 CartPole / Pendulum / MountainCar. */
int32_t rlhip_env_act_push_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n,
                               const int32_t* actions, uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb,
                               float* obs_out, float* last_obs, rlhip_stream_t stream);

/* ------------------------------------- one-shot peer-to-peer all-reduce (multi-GPU learner) -- */
/* The exchange step of the sharded learner: SUM of the small flat gradient over the ranks, inside ONE
 * kernel on the caller's stream.  Each rank owns a comm buffer (uncached device memory, rlhip_p2p_comm_bytes(cap)
 * bytes, zero-initialised) which every peer maps through HIP IPC; rlhip_p2p_allreduce_f32 publishes the local vector,
 * waits for every rank's sequence flag and sums the buffers in rank order (bit-identical results on all ranks).
 * seq = 1, 2, 3, ... must advance by one per call on every rank.  A wait that exceeds timeout_polls sets
 * status_dev[0] = 1 (device or host-pinned memory) and overwrites `data` with NaN: an unreduced gradient must not
 * reach the optimiser silently.  These are the building blocks; hosts use rlhip_comm_* / rlhip_allreduce_grads below,
 * which own the buffers, validate the path on every rank and fall back to RCCL otherwise. */
int32_t rlhip_p2p_alloc(int64_t bytes, void** out);
int32_t rlhip_p2p_free(void* p);
int32_t rlhip_p2p_export(void* p, uint8_t handle_out[64]);
int32_t rlhip_p2p_import(const uint8_t handle[64], void** out);
int32_t rlhip_p2p_close(void* p);
int32_t rlhip_p2p_can_access(int32_t peer_device);                                  /* hipDeviceCanAccessPeer */
int32_t rlhip_p2p_probe(const void* p, int64_t byte_offset, uint32_t* value_out);   /* 4-byte D2H copy from a peer */
int64_t rlhip_p2p_comm_bytes(int64_t cap);
int32_t rlhip_p2p_allreduce_f32(float* data, int64_t n, int64_t cap, int32_t rank, int32_t world,
                                void* const* comm_bufs_host, uint32_t seq, int64_t timeout_polls,
                                int32_t* status_dev, rlhip_stream_t stream);

/* ------------------------------------ the sharded learner's collective behind the ABI -- */
/* SURVEY.md 8b/8e.  Replaces nothing in the reference (it has no Distributed / MPI / NCCL code); it is the exchange a
 * data-parallel `optimise!(::FluxApproximator, grad)` (RLCore/src/policies/learners/flux_approximator.jl:46) needs:
 * the SUM of the flat gradient over the ranks BEFORE clip_by_global_norm! (RLCore/src/utils/basic.jl:19-29).
 * The host moves two small blobs between its ranks with whatever byte transport it has (csrc/comm.hip header):
 *   rank 0: rlhip_comm_unique_id -> broadcast 128 B;  all: rlhip_comm_init;  all: rlhip_comm_export -> all-gather
 *   (64 B handle, device id);  all: rlhip_p2p_setup;  then rlhip_allreduce_grads per optimiser step.
 * rlhip_comm_init is a collective when unique_id_host != NULL (ncclCommInitRank on the CURRENT device; RCCL is
 * dlopen'ed here, not at library load).  unique_id_host = NULL creates a communicator without RCCL (the peer-to-peer
 * path only; e.g. several ranks sharing one GPU, which RCCL refuses).  cap = floats of the largest vector that takes
 * the peer-to-peer path.  rlhip_p2p_setup is a collective as well: it maps every peer's buffer, runs an exact
 * self-test and agrees on the verdict across ranks -- *active_out is the same on every rank; when it is 0
 * rlhip_comm_info().why says why and rlhip_allreduce_grads uses ncclAllReduce on the caller's stream.
 * rlhip_allreduce_grads: in-place SUM, enqueued on `stream`, bit-identical on every rank on the peer-to-peer path
 * (rank-order summation).  A peer that does not arrive within the timeout makes the call's result NaN and
 * rlhip_comm_check return RLHIP_ETIMEOUT (it reads a host-pinned word: no synchronisation).
 * Tear-down, two phases with a host-side barrier before each: barrier -> rlhip_comm_unmap (closes this rank's mappings of
 * the peers' buffers and its RCCL communicator) -> barrier -> rlhip_comm_destroy (frees the own, IPC-exported buffer: no
 * peer may still have it mapped).  rlhip_comm_destroy alone does both (world = 1, error paths). */
typedef void* rlhip_comm_t;
typedef struct rlhip_comm_desc {
    int32_t rank, world, device, p2p_active, rccl_active;
    uint32_t seq;          /* last peer-to-peer sequence number used */
    int64_t cap, timeout_polls;
    int32_t* status;       /* host-pinned, device-visible: [0] != 0 after a timeout */
    void* bufs[16];        /* every rank's exchange buffer as mapped in this process (rlhip_p2p_allreduce_f32 layout) */
    char why[256];         /* why the peer-to-peer path is not active ("ok" when it is) */
    char rccl_path[256];   /* the RCCL the process bound to ("" before the first use) */
} rlhip_comm_desc;
int32_t rlhip_comm_unique_id(uint8_t id_out_host[128]);
int32_t rlhip_comm_init(int32_t rank, int32_t world, const uint8_t* unique_id_host, int64_t cap,
                        rlhip_comm_t* comm_out);
int32_t rlhip_comm_export(rlhip_comm_t comm, uint8_t handle_out_host[64], int32_t* device_out);
int32_t rlhip_p2p_setup(rlhip_comm_t comm, const uint8_t* handles_host /* world x 64 B */,
                        const int32_t* devices_host /* world */, int32_t* active_out);
/* Switch the peer-to-peer path of this rank off (`why_host` is recorded for rlhip_comm_info): for a host that learns over its own
 * transport that another rank's set-up FAILED WITH AN ERROR (rlhip_comm_export / rlhip_p2p_setup returned non-zero there, so that
 * rank took no part in the agreement) -- every rank must then use the same transport.  Not needed after a clean
 * rlhip_p2p_setup: its verdict is already agreed across the ranks. */
int32_t rlhip_comm_disable_p2p(rlhip_comm_t comm, const char* why_host);
int32_t rlhip_allreduce_grads(rlhip_comm_t comm, float* grad, int64_t n, rlhip_stream_t stream);
int32_t rlhip_comm_check(rlhip_comm_t comm);
int32_t rlhip_comm_info(rlhip_comm_t comm, rlhip_comm_desc* out_host);
int32_t rlhip_comm_set_timeout(rlhip_comm_t comm, int64_t timeout_polls);
int32_t rlhip_comm_advance_seq(rlhip_comm_t comm, uint32_t n_steps);
int32_t rlhip_comm_unmap(rlhip_comm_t comm);
int32_t rlhip_comm_destroy(rlhip_comm_t comm);

/* ------------------------------------------------- one DQN vec-step as a single call -- */
/* One trip round the body of `_run` (RLCore/src/core/run.jl:52-70) for Agent{QBasedPolicy{DQN}} on the vector
 * env: plan! (q_based_policy.jl:30-32) -> act! -> push!(agent, PostActStage) (agent_base.jl:56-59) ->
 * optimise! (q_based_policy.jl:49; flux_approximator.jl:46; target_network.jl:70-88).  Enqueues exactly the
 * kernels of the per-step entry points, in the same order, so results are bit-identical to them.
 * The counters (explorer_step, draw_ctr, do_update, do_sync) are the host's: pure functions of the step count. */
typedef struct {
    int32_t kind;              /* 0 CartPole, 1 Pendulum (discrete), 2 MountainCar; Float32 state */
    const void* env_cfg;       /* rlhip_*_cfg (host) */
    const rlhip_env_state* st; /* device state arrays */
    int64_t n;                 /* env instances */
    uint64_t env_seed;
    uint32_t env_id_base;
    float* obs;                /* (obs_dim, n): current observation in, next observation out */
    float* last_obs;           /* (obs_dim, n) pre-reset observation of the step, may be NULL */
    rlhip_ring* ring;          /* host struct; its counters advance */
    int32_t layers;            /* 2: ns -> h -> na (dqn.hip); 3: ns -> h -> h -> na, h = 128 or 256 (dqn3.hip / ppo3w.hip, MFMA) */
    int64_t h, na;
    int32_t act;               /* 0 relu, 1 tanh */
    float* params;
    uint16_t* packed;          /* layers == 3 */
    float* target;
    uint16_t* target_packed;   /* layers == 3 */
    float *m, *v, *beta_pow;   /* Adam state */
    float lr, beta1, beta2, adam_eps, max_grad_norm, grad_scale;
    double eps;                /* get_eps(explorer, explorer_step) */
    uint64_t explorer_seed;
    uint32_t explorer_step;
    int64_t batch;
    float gamma, huber_delta;
    uint64_t sampler_seed;
    uint32_t draw_ctr;
    int32_t do_update;         /* run optimise! this step */
    int32_t do_sync;           /* target sync after this update */
    float rho;
    void* workspace;
    float *grad, *loss, *gn;   /* loss / gn: f32[1] device */
    int32_t* actions;          /* i32[n] out (0-based) */
    float* q;                  /* (na, n) out, may be NULL */
} rlhip_dqn_step_args;
int32_t rlhip_dqn_vec_step_f32(rlhip_dqn_step_args* args, rlhip_stream_t stream);

/* ------------------------------------------------------ 3-layer Q-network on the MFMA -- */
/* Chain(Dense(ns, 128, act), Dense(128, 128, act), Dense(128, na)) -- the blog's DQN model
 * (docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15126-15128), forward(learner, x) =
 * model(x) (RLCore/src/policies/learners/flux_approximator.jl:43).  Flat f32 master parameters in
 * Flux.destructure order W1 (h x ns) | b1 | W2 (h x h) | b2 | W3 (na x h) | b3.  The hidden x hidden layer runs
 * on v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate) from a packed bf16 copy of W2 in both operand
 * orders: uint16[rlhip_mlp3_packed_elems(h)], 16-byte aligned, refreshed by rlhip_mlp3_pack_bf16 after every
 * parameter update.  hidden must be 128 (dqn3.hip) or 256 (the streaming kernels of ppo3w.hip). */
int64_t rlhip_mlp3_nparams(int64_t ns, int64_t h, int64_t na);
int64_t rlhip_mlp3_packed_elems(int64_t h);
int32_t rlhip_mlp3_init_f32(float* params, int64_t ns, int64_t h, int64_t na, uint64_t seed, uint32_t net_id,
                            rlhip_stream_t stream);
int32_t rlhip_mlp3_pack_bf16(const float* params, int64_t ns, int64_t h, int64_t na, uint16_t* packed,
                             rlhip_stream_t stream);
/* plan!(QBasedPolicy, env) = forward + EpsilonGreedyExplorer (no tie-break), as rlhip_dqn_plan_f32.
 * actions may be NULL (pure forward: q_out (na x n)), q_out may be NULL. */
int32_t rlhip_dqn3_plan_f32(const float* params, const uint16_t* packed, int64_t ns, int64_t h, int64_t na,
                            int32_t act, const float* obs, int64_t n, double eps, uint64_t seed,
                            uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                            rlhip_stream_t stream);
/* plan! + act! + push! of one DQN vec-step of the 3-layer Q-network (hidden 128) in ONE launch: rlhip_dqn3_plan_f32 followed by
 * rlhip_env_act_push_f32, bit for bit (the lane that selects env e's action goes on with its env step, auto-reset and ring push);
 * the 3-layer counterpart of rlhip_dqn_act_f32, used by rlhip_dqn_vec_step_f32.  obs: (ns, n) device -- this step's observation on
 * entry, the next one's on return.  Supported: the three classic-control envs with their discrete action sets, n <= 32768. */
int32_t rlhip_dqn3_act_supported(int32_t kind, int64_t n, int64_t h, int64_t na);
int32_t rlhip_dqn3_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, const float* params,
                           const uint16_t* packed, int64_t h, int64_t na, int32_t act, double eps, uint64_t explorer_seed,
                           uint32_t explorer_step, uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb, int32_t* actions,
                           float* q_out, float* obs, float* last_obs, rlhip_stream_t stream);
int64_t rlhip_dqn3_workspace_bytes(int64_t ns, int64_t h, int64_t na, int64_t batch);
/* optimise!(learner, batch) up to the gradient, as rlhip_dqn_grad_f32.  idx: optional explicit flat logical
 * indices (e.g. from rlhip_ring_sample_prioritized); NULL = the uniform BatchSampler draw of
 * rlhip_ring_sample_indices(seed, draw_ctr) evaluated inline.  td_out: optional |Q(s,a) - y| per sample
 * (priority write-back).  loss_out: mean Huber loss. */
int32_t rlhip_dqn3_grad_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act, const float* params,
                            const uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                            int64_t batch, const int64_t* idx, float gamma, float huber_delta, uint64_t seed,
                            uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out, float* td_out,
                            rlhip_stream_t stream);
/* the same with importance-sampling weights (rlhip_per_is_weights_f32): loss = mean(weights .* huber(Q(s,a) - y)),
 * every sample's gradient scaled by its weight; td_out stays the unweighted |Q(s,a) - y| */
int32_t rlhip_dqn3_grad_w_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act, const float* params,
                              const uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                              int64_t batch, const int64_t* idx, const float* weights, float gamma, float huber_delta,
                              void* workspace, float* grad_out, float* loss_out, float* td_out, rlhip_stream_t stream);
/* optimise!(learner, batch) of the 3-layer learner complete, in two launches: the gradient (as rlhip_dqn3_grad_f32
 * with the inline uniform draw), then reduce + clip-by-global-norm + Adam + the bf16 re-pack of W2 in one launch;
 * bit-identical to rlhip_dqn3_grad_f32, rlhip_clip_adam_f32, rlhip_mlp3_pack_bf16 in sequence.  `packed` is updated
 * in place.  The tail of `workspace` (rlhip_dqn3_workspace_bytes) holds the launch's counters: zero before the
 * first call, re-armed by every call. */
int32_t rlhip_dqn3_update_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act, float* params,
                              uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                              int64_t batch, float gamma, float huber_delta, uint64_t seed, uint32_t draw_ctr,
                              void* workspace, float* grad_out, float* loss_out, float* m, float* v,
                              float* beta_pow, float grad_scale, float max_grad_norm, float lr, float beta1,
                              float beta2, float adam_eps, float* gn_out, rlhip_stream_t stream);

/* -------------------------------------------------------------------- priority sum-tree -- */
/* Prioritized replay: CircularArrayBuffers.SumTree (compat 0.1.12, RLCore/Project.toml:30) behind
 * ReinforcementLearningTrajectories 0.4 `CircularPrioritizedTraces` + the prioritized BatchSampler method
 * (`inds, priorities = rand(rng, sumtree, batchsize)`, `trajectory[:priority, keys] = p`) -- un-vendored, parity
 * unpinned; BASELINE.json configs[4] ("prioritized sampling gather").
 * tree: float[rlhip_sumtree_nodes(n_leaves)] device, ZERO-INITIALISED by the caller; implicit heap (root 1,
 * leaf k at P + k, P = next pow2 >= n_leaves).  Internal nodes are recomputed as left + right (drift-free).
 * tree[0] is not a heap node: rlhip_sumtree_update uses it as its arrival counter and leaves it 0.
 * Leaves are keyed by PHYSICAL ring position slot * n_env + env. */
int64_t rlhip_sumtree_nodes(int64_t n_leaves);
/* t[start+1 : start+count] .= value */
int32_t rlhip_sumtree_fill_range(float* tree, int64_t n_leaves, int64_t start, int64_t count, float value,
                                 rlhip_stream_t stream);
/* for (k, p) in zip(keys, prio): t[k] = p   (sequential semantics: the last duplicate wins).  0-based keys. */
int32_t rlhip_sumtree_update(float* tree, int64_t n_leaves, const int64_t* leaf, const float* prio, int64_t n,
                             rlhip_stream_t stream);
/* rand(rng, t, batch): v = u01_f32(Philox(seed, idx = b, blk 0, t = draw_ctr, SAMPLER).w2) * t.tree[1], descent
 * `v <= left ? left : (v -= left; right)` that never enters a zero-sum subtree.  prio_out may be NULL. */
int32_t rlhip_sumtree_sample(const float* tree, int64_t n_leaves, int64_t batch, uint64_t seed,
                             uint32_t draw_ctr, int64_t* leaf_out, float* prio_out, rlhip_stream_t stream);
/* PrioritizedDQN priority write-back value: out = (|td| + eps)^alpha (power in Float64, rounded once) */
int32_t rlhip_per_priority_f32(const float* td, int64_t n, float eps, float alpha, float* out,
                               rlhip_stream_t stream);
/* importance-sampling weights of the sampled batch (PrioritizedDQN, removed Zoo; Schaul et al. 2016):
 *   w = 1 ./ ((priorities .+ 1f-10) .^ beta);  w ./= maximum(w)      (powers in Float64, rounded once)
 * -- the N and total-priority factors of (N P(i))^-beta cancel in the normalisation.  PARITY UNPINNED. */
int32_t rlhip_per_is_weights_f32(const float* prio, int64_t n, float beta, float* w_out, rlhip_stream_t stream);
/* after rlhip_ring_push_transition: the n_env leaves of the newest transition frame := priority
 * (CircularPrioritizedTraces `default_priority`) */
int32_t rlhip_ring_push_priority(const rlhip_ring* rb_host, float* tree, float priority,
                                 rlhip_stream_t stream);
/* prioritized BatchSampler over the ring: idx_out = logical flat indices for rlhip_ring_gather, key_out =
 * physical leaf keys for rlhip_sumtree_update (may be NULL), prio_out = their priorities (may be NULL). */
int32_t rlhip_ring_sample_prioritized(const rlhip_ring* rb_host, const float* tree, int64_t batch,
                                      uint64_t seed, uint32_t draw_ctr, int64_t* idx_out, int64_t* key_out,
                                      float* prio_out, rlhip_stream_t stream);
/* The prioritized BatchSampler AND the gather of its batch in ONE launch (round 4): the draw of sample b is made by the
 * workgroup / lane that gathers it (same Philox draw, same descent: idx_out / key_out / prio_out and the gathered batch are
 * bit-identical to rlhip_ring_sample_prioritized followed by rlhip_ring_gather).  Frame-major rings (n_env = 1, frames of
 * >= 1 KB) and Float32 rings with <= 4 components; other layouts run the two launches.  Outputs as rlhip_ring_gather. */
int32_t rlhip_ring_sample_gather_prioritized(const rlhip_ring* rb_host, const float* tree, int64_t batch, uint64_t seed,
                                             uint32_t draw_ctr, int64_t* idx_out, int64_t* key_out, float* prio_out,
                                             void* s, int32_t* a, float* r, uint8_t* term, void* s_next,
                                             rlhip_stream_t stream);

/* The priority write-back of the PREVIOUS batch, the prioritized draw of the next one and the gather of its frames in ONE launch
 * (round 6): `trajectory[:priority, upd_key] = upd_prio` (as rlhip_sumtree_update: the last duplicate wins) is applied by one
 * wavefront of the launch before any workgroup draws; results -- the tree, idx / key / priority, the gathered batch -- are
 * bit-identical to rlhip_sumtree_update followed by rlhip_ring_sample_gather_prioritized, which is also what runs when the
 * combination has no fused form (n_upd > 64, n_upd == 0, rings that are not frame-major, sync == NULL).  `sync`: two device
 * words owned by the caller, zero before the first call and re-armed by every call (one pair per concurrently used stream).
 * The DQN-Atari step then is: plan! / act! / push! -> THIS -> gradient (which leaves the new priorities for the next call). */
int32_t rlhip_ring_update_sample_gather_prioritized(const rlhip_ring* rb_host, float* tree, const int64_t* upd_key,
                                                    const float* upd_prio, int64_t n_upd, int64_t batch, uint64_t seed,
                                                    uint32_t draw_ctr, int64_t* idx_out, int64_t* key_out, float* prio_out,
                                                    void* s, int32_t* a, float* r, uint8_t* term, void* s_next,
                                                    uint32_t* sync, rlhip_stream_t stream);

/* ---------------------------------------------------------------------------------- MLP -- */
/* Chain(Dense(n_in, h, act), Dense(h, n_out)) with flat parameters in Flux.destructure order:
 *   W1 (h x n_in col-major) | b1 (h) | W2 (n_out x h col-major) | b2 (n_out).   act: 0 relu, 1 tanh.
 * FluxApproximator.forward  RLCore/policies/learners/flux_approximator.jl:43.
 * x: SoA (n_in x batch), out: SoA (n_out x batch). */
int64_t rlhip_mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out);
int32_t rlhip_mlp2_forward_f32(const float* params, int64_t n_in, int64_t h, int64_t n_out,
                               int32_t act, const float* x, int64_t batch, float* out,
                               rlhip_stream_t stream);
/* glorot_uniform stand-in (Philox INIT stream; biases zero); net_id separates actor / critic / q */
int32_t rlhip_mlp2_init_f32(float* params, int64_t n_in, int64_t h, int64_t n_out, uint64_t seed,
                            uint32_t net_id, rlhip_stream_t stream);

/* ------------------------------------------------------------------------------ PPO path -- */
/* PPOPolicy hyper-parameters (removed Zoo; blog a_practical_introduction_to_RL.jl/index.html:15257-15278) */
typedef struct {
    float gamma, lambda, clip_range, max_grad_norm;
    float actor_loss_weight, critic_loss_weight, entropy_loss_weight;
    float lr, beta1, beta2, adam_eps;
    int32_t n_epochs, n_microbatches;
    int32_t hidden, act;
    int32_t continuous;          /* 0 categorical actor; 1 gaussian actor (mu, log sigma) */
    int32_t normalize_advantage; /* reserved, must be 0 */
    int32_t layers;              /* 2 (default; 0 means 2): ns -> hidden -> nout on the VALU (ppo.hip / ppo_grad.hip);
                                  * 3: ns -> hidden -> hidden -> nout actor and critic with the hidden x hidden layer on the
                                  * bf16 MFMA, hidden = 128 (ppo3.hip) or 256 (ppo3w.hip), BASELINE configs[2]
                                  * "actor/critic MLP in bf16 MFMA"; CartPole (categorical) and Pendulum (Gaussian) */
} rlhip_ppo_cfg;
int32_t rlhip_ppo_default(rlhip_ppo_cfg* cfg_host);
/* parameter count of ActorCritic(actor = ns->hidden->na_out, critic = ns->hidden->1), flat = [actor | critic] */
int64_t rlhip_ppo_nparams(int32_t kind, const rlhip_ppo_cfg* cfg_host);

/* PPOTrajectory of one update period, time-major SoA device arrays (caller-owned):
 *   obs (T+1, ns, n) f32 | action_i (T, n) i32 | action_f (T, na, n) f32 | logp (T, n) f32 |
 *   value (T+1, n) f32 | reward (T, n) f32 | terminal (T, n) u8 | adv (T, n) f32 | ret (T, n) f32 */
typedef struct {
    float* obs;
    float* logp;
    float* value;
    float* reward;
    float* adv;
    float* ret;
    float* action_f;
    int32_t* action_i;
    uint8_t* terminal;
} rlhip_ppo_traj;

/* The blog's `_run(policy, env::MultiThreadEnv, ...)` inner loop (index.md:351-374) for T vec-steps in
 * ONE launch: per step  state(env) -> actor/critic forward -> sample action (+ log-prob) -> trajectory
 * write -> act! (+ auto-reset) -> reward / terminal write; finally obs[T], value[T].
 * vec_step0: global vec-step counter at entry (Philox t of the sampling streams). */
int32_t rlhip_ppo_rollout_f32(int32_t kind, const void* env_cfg_host, const rlhip_env_state* st_host,
                              int64_t n, int64_t T, const rlhip_ppo_cfg* cfg_host,
                              const float* params, uint64_t seed, uint32_t env_id_base,
                              uint32_t vec_step0, const rlhip_ppo_traj* traj_host,
                              rlhip_stream_t stream);
/* plan!(policy, env) of the vector env as ONE launch (the per-step form of the rollout above, same
 * device code and lane split, hence bit-identical to it): actor/critic forward on obs (SoA ns x n),
 * action sample (Gumbel-max categorical networks.jl:425-432, or gaussian :64-82), log-prob, value.
 * action_i (discrete) or action_f (continuous) must be non-NULL; logp / value are nullable. */
int32_t rlhip_ppo_plan_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, const float* params,
                           const float* obs, int64_t n, uint64_t seed, uint32_t env_id_base,
                           uint32_t vec_step, int32_t* action_i, float* action_f, float* logp,
                           float* value, rlhip_stream_t stream);
/* The per-step protocol's pushes into slot t of the time-major PPO traces, one launch each (agent_base.jl:45-59):
 * PreActStage: state (ns x n SoA), value, action_log_prob, action (action_i or action_f, the other NULL);
 * logp == NULL and no action = the bootstrap push of (state, value) into slot T.  PostActStage: reward, terminal. */
int32_t rlhip_ppo_push_preact_f32(const rlhip_ppo_traj* traj_host, int64_t t, int64_t ns, int64_t n, const float* obs,
                                  const float* value, const float* logp, const int32_t* action_i, const float* action_f,
                                  rlhip_stream_t stream);
int32_t rlhip_ppo_push_postact_f32(const rlhip_ppo_traj* traj_host, int64_t t, int64_t n, const float* reward,
                                   const uint8_t* done, rlhip_stream_t stream);
/* generalized_advantage_estimation(reward, values, gamma, lambda; dims = 2, terminal) + returns */
int32_t rlhip_ppo_gae_f32(const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                          const rlhip_ppo_traj* traj_host, rlhip_stream_t stream);
/* workspace (device) needed by rlhip_ppo_grad_f32 / rlhip_ppo_update_f32, in bytes, for trajectories of up to n x T entries
 * (two-layer nets: partial gradient rows, the unit-record image, and 32 bytes per
 * trajectory entry for the sample records an update call packs once -- none above 2^24 entries).  It must be zero-initialised
 * once after allocation (it holds counters and epoch words that the kernels maintain themselves): rlhip_ppo_workspace_init
 * does that and registers its size. */
int64_t rlhip_ppo_workspace_bytes(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T);
/* ABI 2: REGISTER a workspace before its first use: zero-fills `bytes` bytes on `stream` (the counters and epoch words the
 * kernels maintain) and records (pointer -> bytes) in a host-side table.  Every rlhip_ppo_grad* / rlhip_ppo_apply_f32 /
 * rlhip_ppo_update* call compares rlhip_ppo_workspace_bytes(kind, cfg, n, T) of THAT call with the registered size and returns
 * RLHIP_EINVAL when it is larger -- or when the workspace was never registered -- instead of writing sample records past the
 * allocation (ABI 1: "the ABI carries no size to check").  Re-registering a pointer replaces its entry;
 * rlhip_ppo_workspace_release drops it (call it before freeing the memory). */
int32_t rlhip_ppo_workspace_init(void* workspace, int64_t bytes, rlhip_stream_t stream);
int32_t rlhip_ppo_workspace_release(void* workspace);
/* loss + flat gradient of micro-batch `mb` of epoch `epoch_ctr` (samples = keyed permutation of the
 * T*n transitions).  grad_out: f32[nparams]; losses_out (nullable): f32[4] = loss, actor, critic, entropy */
int32_t rlhip_ppo_grad_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                           const rlhip_ppo_traj* traj_host, const float* params, uint64_t seed,
                           uint32_t epoch_ctr, int32_t mb, void* workspace, float* grad_out,
                           float* losses_out, rlhip_stream_t stream);
/* Multi-GPU optimiser step: after the host's all-reduce of grad_out, rlhip_ppo_apply_f32 does [grad_scale] ->
 * clip_by_global_norm! -> Adam AND refreshes the learner's internal weight records in one launch, so that the next
 * micro-batch can call rlhip_ppo_grad_fresh_f32 (= rlhip_ppo_grad_f32 without its re-pack launch).  The first
 * gradient of an update call must use rlhip_ppo_grad_f32 (the parameters may have been changed by anyone). */
int32_t rlhip_ppo_grad_fresh_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                                 const rlhip_ppo_traj* traj_host, const float* params, uint64_t seed,
                                 uint32_t epoch_ctr, int32_t mb, void* workspace, float* grad_out,
                                 float* losses_out, rlhip_stream_t stream);
int32_t rlhip_ppo_apply_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T, float* params,
                            float* grad, float* m, float* v, float* beta_pow, float grad_scale, void* workspace,
                            float* gn_out, rlhip_stream_t stream);
/* optimise!(policy) of a policy sharded over `world` GPUs in ONE call: per optimiser step { rlhip_ppo_grad[_fresh]_f32
 * -> rlhip_p2p_allreduce_f32 -> rlhip_ppo_apply_f32(grad_scale = 1 / world) }, 4 launches on one stream, no host work
 * in between.  comm_bufs_host / comm_cap / status_dev / timeout_polls as rlhip_p2p_allreduce_f32; seq0 = the last
 * sequence number used (the call consumes seq0 + 1 .. seq0 + n_epochs * n_microbatches). */
int32_t rlhip_ppo_update_p2p_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                                 const rlhip_ppo_traj* traj_host, float* params, float* m, float* v,
                                 float* beta_pow, uint64_t seed, uint32_t update_ctr, void* workspace,
                                 float* grad_scratch, float* losses_out, int32_t rank, int32_t world,
                                 void* const* comm_bufs_host, int64_t comm_cap, uint32_t seq0,
                                 int64_t timeout_polls, int32_t* status_dev, rlhip_stream_t stream);
/* The same with a communicator (rlhip_comm_init): the fused peer-to-peer kernels when rlhip_p2p_setup activated them,
 * otherwise per optimiser step { gradient -> rlhip_allreduce_grads (ncclAllReduce on `stream`) -> rlhip_ppo_apply_f32 }.
 * world = 1 communicators without an RCCL side run rlhip_ppo_update_f32 (with one, the general sequence over a
 * one-rank ncclAllReduce: same bits).  Sequence numbers are kept inside the communicator. */
int32_t rlhip_ppo_update_comm_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                                  const rlhip_ppo_traj* traj_host, float* params, float* m, float* v,
                                  float* beta_pow, uint64_t seed, uint32_t update_ctr, void* workspace,
                                  float* grad_scratch, float* losses_out, rlhip_comm_t comm, rlhip_stream_t stream);
/* n_epochs x n_microbatches of { grad -> clip_by_global_norm! -> Adam } (single-GPU optimise!; multi-GPU hosts call
 * rlhip_ppo_update_comm_f32, or rlhip_ppo_grad_f32, all-reduce, rlhip_ppo_apply_f32).  update_ctr = number of previous
 * update calls.  Two-layer networks: two launches per optimiser step (gradient tiles -- the first one of a call also builds its
 * weight records from `params` and writes the sample records; then partial reduction + norm exchange + clip + Adam + record refresh).  WORKSPACE SIZE: the call writes 32 bytes per trajectory entry
 * (n * T of THIS call) of sample records behind the fixed part of the workspace -- the workspace must have been sized by
 * rlhip_ppo_workspace_bytes(kind, cfg, n, T) for the LARGEST n * T it is ever used with and registered with
 * rlhip_ppo_workspace_init; a call that needs more than was registered returns RLHIP_EINVAL (ABI 2). */
int32_t rlhip_ppo_update_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                             const rlhip_ppo_traj* traj_host, float* params, float* m, float* v,
                             float* beta_pow, uint64_t seed, uint32_t update_ctr, void* workspace,
                             float* grad_scratch, float* losses_out, rlhip_stream_t stream);

/* Device-resident counters: the same three calls with the vec-step counter (counters[0]) and the update
 * counter (counters[1]) read from device memory by the kernels instead of being baked into the launch
 * arguments, so that one whole iteration (rollout -> GAE -> update [-> all-reduces]) can be captured in a
 * HIP graph and replayed: the Philox streams still advance.  rlhip_counters_advance is the last node.
 * `epoch_local` = epoch index inside the current update (0 .. n_epochs-1). */
int32_t rlhip_ppo_rollout_dc_f32(int32_t kind, const void* env_cfg_host, const rlhip_env_state* st_host,
                                 int64_t n, int64_t T, const rlhip_ppo_cfg* cfg_host, const float* params,
                                 uint64_t seed, uint32_t env_id_base, const uint32_t* counters,
                                 const rlhip_ppo_traj* traj_host, rlhip_stream_t stream);
int32_t rlhip_ppo_grad_dc_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                              const rlhip_ppo_traj* traj_host, const float* params, uint64_t seed,
                              uint32_t epoch_local, const uint32_t* counters, int32_t mb, void* workspace,
                              float* grad_out, float* losses_out, rlhip_stream_t stream);
int32_t rlhip_ppo_update_dc_f32(int32_t kind, const rlhip_ppo_cfg* cfg_host, int64_t n, int64_t T,
                                const rlhip_ppo_traj* traj_host, float* params, float* m, float* v,
                                float* beta_pow, uint64_t seed, const uint32_t* counters, void* workspace,
                                float* grad_scratch, float* losses_out, rlhip_stream_t stream);
int32_t rlhip_counters_advance(uint32_t* counters, uint32_t d_vec_step, uint32_t d_update,
                               rlhip_stream_t stream);

/* ------------------------------------------------------------------------------ DQN path -- */
/* BasicDQN / DQN learner (removed Zoo; spec docs/src/rlcore.md:28, blog index.html:15121-15147):
 * q-net and target-net = mlp2 (ns -> h -> na).  One launch samples `batch` transitions from the ring
 * (BatchSampler draw `draw_ctr`), computes the TD target with the target net, the Huber loss and the
 * flat gradient.  workspace bytes: rlhip_dqn_workspace_bytes (allocate it zeroed: see rlhip_dqn_update_f32). */
int64_t rlhip_dqn_workspace_bytes(int64_t ns, int64_t h, int64_t na, int64_t batch);
int32_t rlhip_dqn_grad_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act,
                           const float* params, const float* target_params, int64_t batch,
                           float gamma, float huber_delta, uint64_t seed, uint32_t draw_ctr,
                           void* workspace, float* grad_out, float* loss_out,
                           rlhip_stream_t stream);
/* optimise!(learner, batch) complete -- gradient, then reduce + clip-by-global-norm + Adam -- in ONE launch up to 2048
 * samples (the gradient workgroup that departs last folds the partial rows, clips and steps), in two launches beyond;
 * bit-identical to rlhip_dqn_grad_f32 followed by rlhip_clip_adam_f32 (which is what runs for > 4096 parameters).
 * The last 64 bytes of `workspace` are its departure counters: zero before the first call, re-armed by every call. */
int32_t rlhip_dqn_update_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act, float* params,
                             const float* target_params, int64_t batch, float gamma, float huber_delta,
                             uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out,
                             float* m, float* v, float* beta_pow, float grad_scale, float max_grad_norm, float lr,
                             float beta1, float beta2, float adam_eps, float* gn_out, rlhip_stream_t stream);
/* The same learner step on explicit flat logical indices (e.g. from rlhip_ring_sample_prioritized, the prioritized
 * BatchSampler of RLTrajectories 0.4) with |Q(s,a) - y| per sample returned for the priority write-back. */
int32_t rlhip_dqn_grad_idx_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act,
                               const float* params, const float* target_params, int64_t batch,
                               const int64_t* idx, float gamma, float huber_delta, void* workspace,
                               float* grad_out, float* loss_out, float* td_out, rlhip_stream_t stream);
/* ... and with importance-sampling weights (prioritized replay with beta > 0): loss = mean(weights .* huber) */
int32_t rlhip_dqn_grad_idx_w_f32(const rlhip_ring* rb_host, int64_t h, int64_t na, int32_t act,
                                 const float* params, const float* target_params, int64_t batch,
                                 const int64_t* idx, const float* weights, float gamma, float huber_delta,
                                 void* workspace, float* grad_out, float* loss_out, float* td_out,
                                 rlhip_stream_t stream);
/* plan!(QBasedPolicy, env) for the vector env in one launch: q = forward(learner, state(env)) then
 * eps-greedy selection (q_based_policy.jl:30-32, abstract_learner.jl:37-39, epsilon_greedy_explorer.jl:108-112).
 * q_out (nullable): SoA (na x n). */
int32_t rlhip_dqn_plan_f32(const float* params, int64_t ns, int64_t h, int64_t na, int32_t act,
                           const float* obs, int64_t n, double eps, uint64_t seed,
                           uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                           rlhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RLHIP_H */
