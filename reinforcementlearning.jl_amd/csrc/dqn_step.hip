// dqn_step.hip -- one whole vec-step of the DQN agent loop as ONE C-ABI call (host-side composition of the
// kernels in dqn.hip / dqn3.hip / envs.hip / ring.hip / optim.hip; no new device code).
//
// What it replaces: one trip round the body of `_run` (RLCore/src/core/run.jl:52-70) for
// `Agent{QBasedPolicy{DQNLearner}}` on the vector env --
//     action = plan!(policy, env)                      q_based_policy.jl:30-32 -> explorer :108-112
//     act!(env, action)                                CartPoleEnv.jl:112-140 (+ auto-reset, MultiThreadEnv protocol)
//     push!(agent, PostActStage(), env, action)        agent_base.jl:56-59
//     optimise!(agent, PostActStage())                 q_based_policy.jl:49 -> learner -> flux_approximator.jl:46,
//                                                      target_network.jl:70-88
// The per-step drop-in protocol costs one ccall (and, from Python, ~8 us of interpreter time) per arrow; at 4096
// envs every kernel is a few microseconds, so the loop was host-paced (72 us per vec-step).  This entry point
// enqueues the same kernels in the same order with the same arguments -- results are bit-identical to the
// per-step protocol (tests/test_gpu_run.py) -- from one call.
#include "common.h"

#include <stdlib.h>

extern "C" {
int32_t rlhip_dqn_plan_f32(const float*, int64_t, int64_t, int64_t, int32_t, const float*, int64_t, double, uint64_t,
                           uint32_t, uint32_t, int32_t*, float*, rlhip_stream_t);
int32_t rlhip_dqn3_plan_f32(const float*, const uint16_t*, int64_t, int64_t, int64_t, int32_t, const float*, int64_t,
                            double, uint64_t, uint32_t, uint32_t, int32_t*, float*, rlhip_stream_t);
int32_t rlhip_dqn_update_f32(const rlhip_ring*, int64_t, int64_t, int32_t, float*, const float*, int64_t, float, float,
                             uint64_t, uint32_t, void*, float*, float*, float*, float*, float*, float, float, float, float,
                             float, float, float*, rlhip_stream_t);
int32_t rlhip_dqn_grad_f32(const rlhip_ring*, int64_t, int64_t, int32_t, const float*, const float*, int64_t, float,
                           float, uint64_t, uint32_t, void*, float*, float*, rlhip_stream_t);
int32_t rlhip_dqn3_grad_f32(const rlhip_ring*, int64_t, int64_t, int32_t, const float*, const uint16_t*, const float*,
                            const uint16_t*, int64_t, const int64_t*, float, float, uint64_t, uint32_t, void*, float*,
                            float*, float*, rlhip_stream_t);
int32_t rlhip_dqn3_update_f32(const rlhip_ring*, int64_t, int64_t, int32_t, float*, uint16_t*, const float*, const uint16_t*,
                              int64_t, float, float, uint64_t, uint32_t, void*, float*, float*, float*, float*, float*,
                              float, float, float, float, float, float, float*, rlhip_stream_t);
int32_t rlhip_env_act_push_f32(int32_t, const void*, const rlhip_env_state*, int64_t, const int32_t*, uint64_t, uint32_t,
                               rlhip_ring*, float*, float*, rlhip_stream_t);
int32_t rlhip_mlp3_pack_bf16(const float*, int64_t, int64_t, int64_t, uint16_t*, rlhip_stream_t);
int64_t rlhip_mlp2_nparams(int64_t, int64_t, int64_t);
int64_t rlhip_mlp3_nparams(int64_t, int64_t, int64_t);
int32_t rlhip_env_obs_dim(int32_t kind);
int32_t rlhip_dqn_act_supported(int32_t kind, int64_t n, int64_t h);
int32_t rlhip_dqn3_act_supported(int32_t kind, int64_t n, int64_t h, int64_t na);
int32_t rlhip_dqn3_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, const float* params,
                           const uint16_t* packed, int64_t h, int64_t na, int32_t act, double eps, uint64_t explorer_seed,
                           uint32_t explorer_step, uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb, int32_t* actions,
                           float* q_out, float* obs, float* last_obs, rlhip_stream_t stream);
int32_t rlhip_dqn_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, const float* params,
                          int64_t h, int64_t na, int32_t act, double eps, uint64_t explorer_seed, uint32_t explorer_step,
                          uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb, int32_t* actions, float* q_out,
                          float* obs_out, float* last_obs, rlhip_stream_t stream);
}

extern "C" int32_t rlhip_dqn_vec_step_f32(rlhip_dqn_step_args* a, rlhip_stream_t stream) {
    RLHIP_REQUIRE(a != nullptr, "args is NULL");
    RLHIP_REQUIRE(a->env_cfg && a->st && a->obs && a->ring && a->params && a->actions, "NULL argument");
    RLHIP_REQUIRE(a->layers == 2 || a->layers == 3, "layers must be 2 or 3");
    RLHIP_REQUIRE(a->layers == 2 || (a->packed && a->target_packed), "3-layer network needs the packed weights");
    RLHIP_REQUIRE(!a->do_update || (a->target && a->m && a->v && a->beta_pow && a->workspace && a->grad),
                  "learner buffers missing");
    const int64_t ns = rlhip_env_obs_dim(a->kind);
    RLHIP_REQUIRE(ns == a->ring->obs_dim && a->n == a->ring->n_env, "ring geometry does not match the env");
    int32_t rc;
    if (a->layers == 2 && rlhip_dqn_act_supported(a->kind, a->n, a->h)) {
        // plan! + act! + push! in one launch (dqn_act.hip): same device functions, same slots, bit-identical
        rc = rlhip_dqn_act_f32(a->kind, a->env_cfg, a->st, a->n, a->params, a->h, a->na, a->act, a->eps,
                               a->explorer_seed, a->explorer_step, a->env_seed, a->env_id_base, a->ring, a->actions,
                               a->q, a->obs, a->last_obs, stream);
        if (rc) return rc;
    } else if (a->layers == 3 && rlhip_dqn3_act_supported(a->kind, a->n, a->h, a->na)) {
        // the same for the 3-layer Q-network (dqn3.hip: the plan kernel's selecting lane goes on with act! + push!)
        rc = rlhip_dqn3_act_f32(a->kind, a->env_cfg, a->st, a->n, a->params, a->packed, a->h, a->na, a->act, a->eps,
                                a->explorer_seed, a->explorer_step, a->env_seed, a->env_id_base, a->ring, a->actions, a->q,
                                a->obs, a->last_obs, stream);
        if (rc) return rc;
    } else {
    // plan!(policy, env): Q forward + eps-greedy on the current observation
        if (a->layers == 2)
            rc = rlhip_dqn_plan_f32(a->params, ns, a->h, a->na, a->act, a->obs, a->n, a->eps, a->explorer_seed,
                                    a->env_id_base, a->explorer_step, a->actions, a->q, stream);
        else
            rc = rlhip_dqn3_plan_f32(a->params, a->packed, ns, a->h, a->na, a->act, a->obs, a->n, a->eps, a->explorer_seed,
                                     a->env_id_base, a->explorer_step, a->actions, a->q, stream);
        if (rc) return rc;
        // act!(env, action) with auto-reset + push!(trajectory, (state = s', action, reward, terminal)): one launch
        // (dqn_act.hip; same device functions and slots as rlhip_env_step + rlhip_ring_push_transition)
        rc = rlhip_env_act_push_f32(a->kind, a->env_cfg, a->st, a->n, a->actions, a->env_seed, a->env_id_base, a->ring,
                                    a->obs, a->last_obs, stream);
        if (rc) return rc;
    }
    if (!a->do_update) return RLHIP_OK;
    // optimise!(learner, trajectory): sample + TD target + Huber + gradient, then clip + Adam
    int64_t np;
    if (a->layers == 2) {
        // gradient partials, then reduce + clip + Adam in one launch (bit-identical to the two calls below)
        np = rlhip_mlp2_nparams(ns, a->h, a->na);
        rc = rlhip_dqn_update_f32(a->ring, a->h, a->na, a->act, a->params, a->target, a->batch, a->gamma, a->huber_delta,
                                  a->sampler_seed, a->draw_ctr, a->workspace, a->grad, a->loss, a->m, a->v, a->beta_pow,
                                  a->grad_scale, a->max_grad_norm, a->lr, a->beta1, a->beta2, a->adam_eps, a->gn, stream);
        if (rc) return rc;
    } else if (a->layers == 3) {
        // gradient, then reduce + clip + Adam + bf16 re-pack in one launch (bit-identical to the calls below)
        np = rlhip_mlp3_nparams(ns, a->h, a->na);
        rc = rlhip_dqn3_update_f32(a->ring, a->h, a->na, a->act, a->params, a->packed, a->target, a->target_packed,
                                   a->batch, a->gamma, a->huber_delta, a->sampler_seed, a->draw_ctr, a->workspace, a->grad,
                                   a->loss, a->m, a->v, a->beta_pow, a->grad_scale, a->max_grad_norm, a->lr, a->beta1,
                                   a->beta2, a->adam_eps, a->gn, stream);
        if (rc) return rc;
    } else {
        if (a->layers == 2) {
            np = rlhip_mlp2_nparams(ns, a->h, a->na);
            rc = rlhip_dqn_grad_f32(a->ring, a->h, a->na, a->act, a->params, a->target, a->batch, a->gamma,
                                    a->huber_delta, a->sampler_seed, a->draw_ctr, a->workspace, a->grad, a->loss, stream);
        } else {
            np = rlhip_mlp3_nparams(ns, a->h, a->na);
            rc = rlhip_dqn3_grad_f32(a->ring, a->h, a->na, a->act, a->params, a->packed, a->target, a->target_packed,
                                     a->batch, nullptr, a->gamma, a->huber_delta, a->sampler_seed, a->draw_ctr,
                                     a->workspace, a->grad, a->loss, nullptr, stream);
        }
        if (rc) return rc;
        rc = rlhip_clip_adam_f32(a->params, a->grad, a->m, a->v, a->beta_pow, np, a->grad_scale, a->max_grad_norm, a->lr,
                                 a->beta1, a->beta2, a->adam_eps, a->gn, stream);
        if (rc) return rc;
        if (a->layers == 3) {
            rc = rlhip_mlp3_pack_bf16(a->params, ns, a->h, a->na, a->packed, stream);
            if (rc) return rc;
        }
    }
    if (a->do_sync) {  // TargetNetwork: dest = rho * dest + (1 - rho) * src
        rc = rlhip_polyak_f32(a->target, a->params, np, a->rho, stream);
        if (rc) return rc;
        if (a->layers == 3) rc = rlhip_mlp3_pack_bf16(a->target, ns, a->h, a->na, a->target_packed, stream);
    }
    return rc;
}
