/*
 * rl_oracle.h -- CPU parity oracle for the rlhip hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product path (reinforcementlearning.jl_amd/) never links, imports or falls
 * back to it.
 *
 * What it is: a plain-C restatement of the ReinforcementLearning.jl functions
 * on the hot path, each function citing the reference
 * file:line it follows.  Paths are relative to /root/reference/src/ :
 *   RLEnvs = ReinforcementLearningEnvironments/src/environments/examples
 *   RLCore = ReinforcementLearningCore/src
 *
 * Pinning status (see oracle/README.md and DESIGN.md):
 *   PINNED on the reference's own known-answer tests (tests/golden/, JSON files):
 *     discount_rewards / discount_rewards_reduced / generalized_advantage_
 *     estimation, find_all_max, findmax first-index rule, get_eps schedules,
 *     eps-greedy prob vectors, GreedyExplorer, TargetNetwork sync counter,
 *     normlogpdf / diagnormlogpdf (closed-form Normal).
 *   PARITY UNPINNED (the reference holds no golden vectors and cannot be run
 *   here -- no julia binary, un-vendored deps): the three env physics steps
 *   (reference tests are interface-only), everything whose arithmetic lives in
 *   un-vendored third-party packages (ring buffer / samplers =
 *   ReinforcementLearningTrajectories 0.4 + CircularArrayBuffers 0.1.12;
 *   Adam / huber_loss / Dense = Flux 0.14-0.16 -> Optimisers.jl / NNlib),
 *   the removed Zoo learners (DQN / PPO loss), and Julia's RNG streams
 *   (replaced by the Philox4x32-10 specification below, shared with the HIP
 *   kernels by specification, not by code).
 *
 * Build: `make -C oracle` -> oracle/_build/librl_oracle.so
 *        (gcc -O2 -ffp-contract=off -fno-fast-math: no FMA contraction, so the
 *        Float32/Float64 promotion quirks of the Julia source are reproduced.)
 *
 * Conventions: all indices in this C API are 0-based; Julia's 1-based action /
 * index values are (c_value + 1).  Matrices are column-major like Julia.
 */
#ifndef RL_ORACLE_H
#define RL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG -- */
/* Philox4x32-10 (Salmon et al., SC'11).  ctr = {idx, blk, t, tag}:
 *   idx = element index (global env id, sample slot, parameter index)
 *   blk = sub-block when one draw needs more than 4 words
 *   t   = time counter (episode number, explorer step, vec-step, epoch)
 *   tag = stream tag (RLO_TAG_*)
 * key = {seed_lo, seed_hi}. */
enum {
    RLO_TAG_RESET = 0,   /* env reset draws               */
    RLO_TAG_EXPLORE = 1, /* eps-greedy draws              */
    RLO_TAG_GUMBEL = 2,  /* categorical (Gumbel-max)      */
    RLO_TAG_NORMAL = 3,  /* gaussian policy noise         */
    RLO_TAG_SAMPLER = 4, /* replay batch indices          */
    RLO_TAG_SHUFFLE = 5, /* epoch permutation keys        */
    RLO_TAG_INIT = 6,    /* weight init                   */
    RLO_TAG_SYNTH = 7,   /* synthetic bench / test data   */
    RLO_TAG_ENVNOISE = 8 /* per-step env noise (Acrobot torque noise) */
};
void rlo_philox4x32_10(uint64_t seed, uint32_t idx, uint32_t blk, uint32_t t, uint32_t tag,
                       uint32_t out[4]);
/* rand(Float32) stand-in: 24 bits, [0,1).  rand(Float64) stand-in: 53 bits, [0,1). */
float rlo_u01_f32(uint32_t w);
double rlo_u01_f64(uint32_t w_hi, uint32_t w_lo);
/* rand(1:n) stand-in (0-based result): floor(w * n / 2^32). */
uint32_t rlo_randint(uint32_t w, uint32_t n);
/* Keyed bijection on [0, n): 6-round Feistel on the enclosing power of 4 with
 * cycle walking.  Stand-in for `shuffle(rng, 1:n)[i]` (a permutation: every index once). */
uint32_t rlo_permute(uint64_t seed, uint32_t epoch, uint32_t n, uint32_t i);
void rlo_fill_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t t, uint32_t tag);
/* Box-Muller standard normals from the NORMAL stream (f32). */
void rlo_normal_pair_f32(uint32_t w0, uint32_t w1, float* z0, float* z1);

/* ----------------------------------------------------------------- envs -- */
/* kwargs of CartPoleEnv(; ...) (RLEnvs/CartPoleEnv.jl:22-32,74-79), Float64 as typed by the user */
typedef struct {
    double gravity, masscart, masspole, halflength, forcemag, dt, thetathreshold_deg, xthreshold;
    int64_t max_steps;
    int32_t continuous; /* 0: action in {0,1} (Julia 1,2); 1: action in [-1,1] of type T */
} rlo_cartpole_cfg;
void rlo_cartpole_default(rlo_cartpole_cfg* c);

/* kwargs of PendulumEnv(; ...) (RLEnvs/PendulumEnv.jl:41-53) */
typedef struct {
    double max_speed, max_torque, g, m, l, dt;
    int64_t max_steps;
    int32_t continuous; /* 1: action f32/f64 torque; 0: action index in 0..n_actions-1 */
    int32_t n_actions;
} rlo_pendulum_cfg;
void rlo_pendulum_default(rlo_pendulum_cfg* c);

/* kwargs of MountainCarEnv(; ...) (RLEnvs/MountainCarEnv.jl:19-40,67-81) */
typedef struct {
    double min_pos, max_pos, max_speed, goal_pos, goal_velocity, power, gravity;
    int64_t max_steps;
    int32_t continuous;
} rlo_mountaincar_cfg;
void rlo_mountaincar_default(rlo_mountaincar_cfg* c, int continuous);

/* AcrobotEnv(; kwargs...)  RLEnvs/src/environments/3rd_party/AcrobotEnv.jl:22-40.  PARITY UNPINNED for this env: the
 * reference integrates one act! with OrdinaryDiffEq.solve(ode, RK4()) (:128-129), an un-vendored ADAPTIVE driver; this
 * restatement takes ONE classic RK4 step of length dt over the reference's own dsdt (:147-199) in Float64 (what the
 * "python gym" implementation the file cites does), then wraps / bounds (:135-138) and stores the state as T. */
typedef struct {
    double link_length_a, link_length_b, link_mass_a, link_mass_b, link_com_pos_a, link_com_pos_b, link_moi,
        max_torque_noise, max_vel_a, max_vel_b, g, dt;
    int64_t max_steps;
    int32_t nips; /* book_or_nips: 0 "book" (default), 1 "nips" */
} rlo_acrobot_cfg;
void rlo_acrobot_default(rlo_acrobot_cfg* c);

/* SoA vector-env state, one entry per env instance.
 * s: state arrays, s[k][i] = component k of env i (cartpole k<4; pendulum, mountaincar k<2)
 * t: step counter; done: terminal flag of the LAST act!; reward: reward(env) after the last act!
 * episode: number of resets performed so far (Philox time counter of the next reset)        */
typedef struct {
    void* s[4];
    int32_t* t;
    uint8_t* done;
    void* reward;
    uint32_t* episode;
} rlo_env_state;

/* kind: 0 cartpole, 1 pendulum, 2 mountaincar, 3 acrobot; is_f64: element type T of the env.
 * reset(mask == NULL): reset all; else reset where mask[i] != 0.
 * step: act!(env, a) for every env (auto_reset != 0: a terminated env is reset right after
 *       reward/done were recorded -- the MultiThreadEnv protocol); last_obs (optional, may be NULL)
 *       receives the pre-reset observation (obs_dim x n, SoA).
 * actions: int32 (discrete, 0-based) or T (continuous).                                      */
int rlo_env_reset(int kind, int is_f64, const void* cfg, rlo_env_state* st, int64_t n,
                  uint64_t seed, uint32_t env_id_base, const uint8_t* mask);
int rlo_env_step(int kind, int is_f64, const void* cfg, rlo_env_state* st, int64_t n,
                 const void* actions, int auto_reset, uint64_t seed, uint32_t env_id_base,
                 void* last_obs);
/* state(env): obs (obs_dim x n, SoA: obs[k*n + i]); cartpole 4, pendulum 3, mountaincar 2, acrobot 6 */
int rlo_env_obs(int kind, int is_f64, const rlo_env_state* st, int64_t n, void* obs);
int rlo_env_obs_dim(int kind);
int rlo_env_state_dim(int kind);

/* ---------------------------------------------------------------- stochastic heads -- */
/* GaussianNetwork / SoftGaussianNetwork heads  RLCore/src/utils/networks.jl:64-116, 147-198 (rlo_heads.c).
 * mu, raw_sigma (d x n); K samples per state; action (d x K x n), logp (K x n); squash 0 identity / 1 tanh;
 * soft 1 = SoftGaussianNetwork (always tanh).  d <= 64. */
int rlo_gaussian_head_sample_f32(const float* mu, const float* raw_sigma, int64_t d, int64_t n, int64_t K,
                                 float min_sigma, float max_sigma, int squash, int soft, uint64_t seed,
                                 uint32_t env_id_base, uint32_t step, float* action_out, float* logp_out);
int rlo_gaussian_head_logp_f32(const float* mu, const float* raw_sigma, const float* action, int64_t d, int64_t n,
                               int64_t K, float min_sigma, float max_sigma, int squash, int soft, float* logp_out);

/* ---------------------------------------------------------------- scans -- */
/* RLCore/utils/basic.jl:138-235 discount_rewards, :237-319 discount_rewards_reduced,
 * :334-417 generalized_advantage_estimation.  Matrices column-major (n1 x n2).
 * dims = 0: vector input (n2 must be 1); dims = 1 or 2: the Julia `dims` keyword (scan axis).
 * terminal / init may be NULL (= nothing).                                                  */
int rlo_discount_rewards_f64(double* out, const double* r, int64_t n1, int64_t n2, double gamma,
                             const uint8_t* terminal, const double* init, int dims);
int rlo_discount_rewards_f32(float* out, const float* r, int64_t n1, int64_t n2, float gamma,
                             const uint8_t* terminal, const float* init, int dims);
int rlo_discount_rewards_reduced_f64(double* out, const double* r, int64_t n1, int64_t n2,
                                     double gamma, const uint8_t* terminal, const double* init,
                                     int dims);
int rlo_discount_rewards_reduced_f32(float* out, const float* r, int64_t n1, int64_t n2,
                                     float gamma, const uint8_t* terminal, const float* init,
                                     int dims);
/* values has one more entry than rewards along the scan axis */
int rlo_gae_f64(double* adv, const double* r, const double* v, int64_t n1, int64_t n2, double gamma,
                double lambda, const uint8_t* terminal, int dims);
int rlo_gae_f32(float* adv, const float* r, const float* v, int64_t n1, int64_t n2, float gamma,
                float lambda, const uint8_t* terminal, int dims);

/* ------------------------------------------------------------ selection -- */
/* RLCore/utils/basic.jl:91-120.  Returns count; idx_out (0-based) must hold n entries; mask may be NULL */
int64_t rlo_find_all_max_f64(const double* x, int64_t n, const uint8_t* mask, double* vmax,
                             int64_t* idx_out);
/* findmax(A)[2] (0-based): first maximal index, NaN is maximal.  mask: findmax_masked (:117-118) */
int64_t rlo_findmax_f64(const double* x, int64_t n, const uint8_t* mask);
int64_t rlo_findmax_f32(const float* x, int64_t n, const uint8_t* mask);
/* get_eps, RLCore/policies/explorers/epsilon_greedy_explorer.jl:69-88. kind 0 linear, 1 exp */
double rlo_get_eps(int kind, double eps_stable, double eps_init, int64_t warmup_steps,
                   int64_t decay_steps, int64_t step);
/* plan!(::EpsilonGreedyExplorer, values[, mask]) (:102-131) for a batch: values (na x n) column-major,
 * one column per env; draws come from Philox(seed, idx = env_id_base + i, t = step, EXPLORE):
 * w0,w1 -> u (f64), w2 -> random index.  is_break_tie selects the :102-106/:118-123 variant
 * (greedy branch draws w3 over find_all_max).  Actions out are 0-based.                       */
int rlo_eps_greedy_select_f32(const float* values, int64_t na, int64_t n, const uint8_t* mask,
                              double eps, int is_break_tie, uint64_t seed, uint32_t env_id_base,
                              uint32_t step, int32_t* actions);
/* prob(::EpsilonGreedyExplorer, values[, mask]) (:141-194) */
int rlo_eps_greedy_prob_f64(const double* values, int64_t na, const uint8_t* mask, double eps,
                            int is_break_tie, double* probs);
/* Gumbel-max categorical, RLCore/utils/networks.jl:425-432 (+ masking :466-468).
 * logits (na x n) f32; u ~ Philox(seed, idx, blk = k/2, t = step, GUMBEL) f64 per (k, i).
 * actions 0-based; logp_out (optional) = logsoftmax(logits)[a].                              */
int rlo_categorical_sample_f32(const float* logits, int64_t na, int64_t n, const uint8_t* mask,
                               uint64_t seed, uint32_t env_id_base, uint32_t step,
                               int32_t* actions, float* logp_out);

/* --------------------------------------------------- parameter updates -- */
/* TargetNetwork.optimise! tail, RLCore/policies/learners/target_network.jl:70-88 */
void rlo_polyak_f32(float* dst, const float* src, int64_t n, float rho);
/* returns 1 when a sync must run on this call; updates *n_optimise like :74,:86 */
int rlo_target_sync_due(int64_t* n_optimise, int64_t sync_freq);
/* clip_by_global_norm!, RLCore/utils/basic.jl:19-29; returns gn */
float rlo_clip_by_global_norm_f32(float* g, int64_t n, float clip_norm);
/* Optimisers.jl Adam (un-vendored; formula in SURVEY.md Appendix B); t = 1-based step count */
void rlo_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t t);
/* RLCore/utils/distributions.jl:9,18-21 and :31-34 */
float rlo_normlogpdf_f32(float mu, float sigma, float x);
void rlo_diagnormlogpdf_f32(const float* mu, const float* sigma, const float* x, int64_t d,
                            int64_t n, float* out);
/* Flux.Losses.huber_loss(q, target; delta) with mean aggregation + dL/dq */
float rlo_huber_f32(const float* q, const float* target, int64_t n, float delta, float* dq);
/* DQN target: G = r + gamma * (1 - terminal) * max_a' Qt(s')   (Qt: na x n) */
void rlo_td_target_f32(const float* qt_next, int64_t na, int64_t n, const float* r,
                       const uint8_t* terminal, float gamma, float* target);

/* remaining explorers (rlo_select.c): kind 0 weighted, 1 weighted-softmax, 2 gumbel-softmax; UCB */
int rlo_explorer_select_f32(int kind, const float* values, int64_t na, int64_t n, const uint8_t* mask,
                            int is_normalized, uint64_t seed, uint32_t env_id_base, uint32_t step,
                            int32_t* actions);
int rlo_ucb_select_f32(const float* values, int64_t na, int64_t n, double c, double* counts, int64_t step,
                       uint64_t seed, uint32_t env_id_base, int32_t* actions);

/* ---------------------------------------------------------- replay ring -- */
/* CircularArraySARTSTraces over a vector env: each slot holds one vec-step (n_env transitions).
 * state is a multiplexed trace of capacity+1 frames (next_state[i] = state[i+1]);
 * action/reward/terminal have capacity frames.  SoA inside a frame: state[(slot*obs_dim + k)*n_env + i]. */
typedef struct {
    int64_t capacity, n_env, obs_dim;
    int64_t head_sa; /* physical slot of the OLDEST state frame                        */
    int64_t len_sa;  /* number of state frames stored (<= capacity + 1)               */
    int64_t head_rt; /* physical slot of the oldest action/reward/terminal frame      */
    int64_t len_rt;  /* number of transition frames stored (<= capacity)              */
    float* state;    /* (capacity+1) * obs_dim * n_env */
    int32_t* action; /* capacity * n_env               */
    float* reward;   /* capacity * n_env               */
    uint8_t* terminal;
} rlo_ring;
void rlo_ring_init(rlo_ring* rb, int64_t capacity, int64_t n_env, int64_t obs_dim, float* state,
                   int32_t* action, float* reward, uint8_t* terminal);
/* push!(traj, (state = s,))  (Agent PreEpisodeStage, RLCore/policies/agent/agent_base.jl:45-47) */
void rlo_ring_push_state(rlo_ring* rb, const float* obs);
/* push!(traj, (state = s', action, reward, terminal))  (PostActStage, :56-59) */
void rlo_ring_push_transition(rlo_ring* rb, const float* next_obs, const int32_t* action,
                              const float* reward, const uint8_t* terminal);
int64_t rlo_ring_length(const rlo_ring* rb); /* number of complete transitions (frames) */
/* BatchSampler: batch flat indices in [0, length*n_env): j = slot_logical * n_env + env,
 * drawn with Philox(seed, idx = b, t = draw_ctr, SAMPLER): ((w0:w1) * length*n_env) >> 64. */
void rlo_ring_sample_indices(const rlo_ring* rb, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                             int64_t* flat_idx);
/* gather (s, a, r, t, s') for flat indices; outputs SoA (obs_dim x batch), batch */
void rlo_ring_gather(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, float* s,
                     int32_t* a, float* r, uint8_t* term, float* s_next);

/* ------------------------------------------------ 3-layer bf16 Q-network -- */
/* Chain(Dense(ns, h, act), Dense(h, h, act), Dense(h, na)) (blog DQN net, index.html:15126-15128); the
 * hidden x hidden layer in bf16 with f32 accumulate (rlo_mlp3.c). */
float rlo_bf16_round_f32(float f);
int64_t rlo_mlp3_nparams(int64_t ns, int64_t h, int64_t na);
void rlo_mlp3_init_f32(float* p, int64_t ns, int64_t h, int64_t na, uint64_t seed, uint32_t net_id);
void rlo_mlp3_forward_f32(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x,
                          int64_t batch, float* out);
/* single-sample forward / backward of the 3-layer bf16 network (shared with the PPO oracle, rlo_learn.c).
 * caches: 4 h floats z1 | h1 | z2 | h2; scratch: dz2b (h floats), dh1 (h doubles); ga: Float64 accumulators */
void rlo_mlp3_forward1(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x, int64_t xstride,
                       float* out, int64_t ostride, float* z1, float* h1, float* z2, float* h2);
void rlo_mlp3_backward1(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x, int64_t xstride,
                        const float* dout, double* ga, const float* z1, const float* h1, const float* z2,
                        const float* h2, float* dz2b, double* dh1);
float rlo_dqn3_loss_grad_f32(int64_t ns, int64_t h, int64_t na, int act, const float* params,
                             const float* target_params, const float* s, const int32_t* a, const float* r,
                             const uint8_t* term, const float* s_next, int64_t b, float gamma, float huber_delta,
                             float* grad, float* q_out, const float* isw /* nullable: importance-sampling weights */);

/* ---------------------------------------------------- priority sum-tree -- */
/* CircularArrayBuffers.SumTree (0.1.12) / RLTrajectories 0.4 prioritized BatchSampler, un-vendored: PARITY
 * UNPINNED.  Implicit heap float tree[2P], P = next pow2 >= n_leaves, leaf k at P + k (the reference's
 * nparents + k); internal node = left + right recomputed from the children (drift-free variant of the
 * reference's `tree[parent] += change` walk); descent `v <= left ? left : (v -= left; right)` made robust
 * against zero-sum subtrees.  Draw: v = u01_f32(Philox(seed, b, 0, draw_ctr, SAMPLER)[2]) * tree[1]. */
int64_t rlo_sumtree_nodes(int64_t n_leaves);
void rlo_sumtree_fill_range(float* tree, int64_t n_leaves, int64_t start, int64_t count, float value);
void rlo_sumtree_update(float* tree, int64_t n_leaves, const int64_t* leaf, const float* prio, int64_t n);
void rlo_sumtree_sample(const float* tree, int64_t n_leaves, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                        int64_t* leaf_out, float* prio_out);
void rlo_per_priority_f32(const float* td, int64_t n, float eps, float alpha, float* out);
/* importance-sampling weights of the sampled batch (removed Zoo PrioritizedDQN, from memory -- PARITY UNPINNED):
 * w = 1 ./ ((priorities .+ 1f-10) .^ beta); w ./= maximum(w)   (powers in Float64, rounded once) */
void rlo_per_is_weights_f32(const float* prio, int64_t n, float beta, float* out);
/* leaf of the newest transition frame gets `priority`; prioritized draw mapped to logical flat indices */
void rlo_ring_push_priority(const rlo_ring* rb, float* tree, float priority);
void rlo_ring_sample_prioritized(const rlo_ring* rb, const float* tree, int64_t batch, uint64_t seed,
                                 uint32_t draw_ctr, int64_t* flat_idx, int64_t* key_out, float* prio_out);

/* stack-at-sample gather for single-env frame rings (StackFrames semantics, rlo_buffer.c) */
/* n-step transitions (NStepBatchSampler; rlo_buffer.c) */
void rlo_ring_sample_indices_nstep(const rlo_ring* rb, int64_t batch, int64_t n_step, uint64_t seed, uint32_t draw_ctr,
                                   int64_t* flat_idx);
void rlo_ring_gather_nstep(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, int64_t n_step, float gamma, float* s,
                           int32_t* a, float* r, uint8_t* term, float* s_next);
float rlo_gamma_pow(float gamma, int64_t n);
void rlo_ring_gather_stacked(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, int64_t n_stack, float* s,
                             int32_t* a, float* r, uint8_t* term, float* s_next);

/* ------------------------------------------------------------------ MLP -- */
/* Chain(Dense(n_in, h, act), Dense(h, n_out)) flat parameters in Flux.destructure order:
 * W1 (h x n_in, col-major), b1 (h), W2 (n_out x h, col-major), b2 (n_out).  act: 0 relu, 1 tanh */
int64_t rlo_mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out);
/* x SoA (n_in x batch: x[k*batch + i]); out SoA (n_out x batch).  Sequential f32 accumulation with
 * fmaf over k then over j (the kernels' order is different; compare with tolerance).         */
void rlo_mlp2_forward_f32(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                          const float* x, int64_t batch, float* out);
/* accumulates dL/dp into g (same layout as p) given dL/dout (n_out x batch SoA) */
void rlo_mlp2_backward_f32(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                           const float* x, int64_t batch, const float* dout, float* g);
/* glorot_uniform(rng) stand-in: U(-s, s), s = sqrt(6/(fan_in+fan_out)), from Philox(seed, idx=param
 * index within the layer tensor, t = layer_id, INIT); biases zero.                            */
void rlo_mlp2_init_f32(float* p, int64_t n_in, int64_t h, int64_t n_out, uint64_t seed,
                       uint32_t net_id);

/* -------------------------------------------------------------- learners -- */
typedef struct {
    float gamma, lambda, clip_range, max_grad_norm;
    float actor_loss_weight, critic_loss_weight, entropy_loss_weight;
    float lr, beta1, beta2, adam_eps;
    int32_t n_epochs, n_microbatches;
    int32_t hidden, act;   /* hidden width of actor and critic, activation */
    int32_t continuous;    /* 0: categorical actor (na logits); 1: gaussian actor (mu, log sigma), 1-D */
    int32_t normalize_advantage;
    int32_t layers; /* 2 (default, 0 also means 2): ns -> h -> nout; 3: ns -> 128 -> 128 -> nout, bf16 hidden layer */
} rlo_ppo_cfg;
void rlo_ppo_default(rlo_ppo_cfg* c);

/* PPO loss + gradient over one micro-batch (removed Zoo PPOPolicy; spec in SURVEY.md App. B).
 * obs SoA (ns x bm).  For discrete: action int32 (0-based); continuous: action f32 (act_f).
 * params = [actor | critic]; grad (zero-initialised by the callee) same layout.
 * losses_out[4] = {loss, actor_loss, critic_loss, entropy_loss}.                              */
void rlo_ppo_loss_grad_f32(const rlo_ppo_cfg* c, int64_t ns, int64_t na, const float* params,
                           const float* obs, const int32_t* act_i, const float* act_f,
                           const float* logp_old, const float* adv, const float* ret, int64_t bm,
                           float* grad, float* losses_out);

/* DQN (BasicDQN / DQNLearner of the removed Zoo): loss + gradient over one batch.
 * q-net and target-net are mlp2 (ns -> h -> na); returns loss; grad zero-initialised by callee. */
float rlo_dqn_loss_grad_f32(int64_t ns, int64_t h, int64_t na, int act, const float* params,
                            const float* target_params, const float* s, const int32_t* a,
                            const float* r, const uint8_t* term, const float* s_next, int64_t b,
                            float gamma, float huber_delta, float* grad,
                            const float* isw /* nullable: importance-sampling weights, loss = mean(w .* huber) */);

/* Whole vectorised PPO iteration on the CPU (rollout of T vec-steps with the MultiThreadEnv
 * protocol, GAE, n_epochs x n_microbatches updates).  Used as the cpu_baseline ("port") and
 * as the end-to-end checker.  Trajectory buffers (caller-owned):
 *   obs (T+1, obs_dim, n) ; action (T, n) int32 or f32 ; logp (T, n) ; value (T+1, n) ;
 *   reward (T, n) ; terminal (T, n) ; adv (T, n) ; ret (T, n)
 * vec_step0 = global vec-step counter at entry (Philox t for sampling), opt_step = Adam t so far. */
typedef struct {
    float *obs, *logp, *value, *reward, *adv, *ret, *action_f;
    int32_t* action_i;
    uint8_t* terminal;
} rlo_ppo_traj;
int rlo_ppo_rollout_f32(int kind, const void* env_cfg, rlo_env_state* st, int64_t n, int64_t T,
                        const rlo_ppo_cfg* c, const float* params, uint64_t seed,
                        uint32_t env_id_base, uint32_t vec_step0, rlo_ppo_traj* tr);
/* GAE + returns over the trajectory (time-major (T, n) layout: scan along T) */
void rlo_ppo_gae_f32(const rlo_ppo_cfg* c, int64_t n, int64_t T, rlo_ppo_traj* tr);
/* n_epochs x n_microbatches of loss/grad -> clip -> Adam; m, v Adam state; *opt_step advanced.
 * update_ctr = number of previous update calls (Philox t of the epoch permutations).       */
int rlo_ppo_update_f32(int kind, const rlo_ppo_cfg* c, int64_t n, int64_t T, rlo_ppo_traj* tr,
                       float* params, float* m, float* v, int64_t* opt_step, uint64_t seed,
                       uint32_t update_ctr, float* last_losses);
int64_t rlo_ppo_nparams(int kind, const rlo_ppo_cfg* c);

#ifdef __cplusplus
}
#endif
#endif /* RL_ORACLE_H */
