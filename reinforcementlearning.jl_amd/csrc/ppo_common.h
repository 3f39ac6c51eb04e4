// ppo_common.h -- small host/device structs shared by ppo.hip (rollout / plan) and ppo_grad.hip.
#pragma once
#include "common.h"
#include "mlp_device.h"
#include "gae_device.h"

namespace rlhip {

struct TrajPtrs {
    float* obs;
    float* logp;
    float* value;
    float* reward;
    float* adv;
    float* ret;
    float* action_f;
    int32_t* action_i;
    uint8_t* terminal;
    static TrajPtrs from(const rlhip_ppo_traj& t) {
        return {t.obs, t.logp, t.value, t.reward, t.adv, t.ret, t.action_f, t.action_i, t.terminal};
    }
};

struct PolicyDesc {
    int h, act, cont, na, nout_a;
    int64_t np_a;
    float gamma, lambda;  // for the GAE scan fused into the tail of the rollout kernels
};

// cfg.layers == 3: actor / critic ns -> h -> h -> nout with the MFMA hidden layer, h = 128 (ppo3.hip) or 256 (ppo3w.hip)
static inline bool is_layers3(const rlhip_ppo_cfg* c) { return c != nullptr && c->layers == 3; }
int64_t ppo3_nparams(int32_t kind, const rlhip_ppo_cfg* c);
int64_t ppo3_workspace_bytes(int32_t kind, const rlhip_ppo_cfg* c, int64_t n, int64_t T);
int32_t ppo3_rollout(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                     const rlhip_ppo_cfg* cfg, const float* params, uint64_t seed, uint32_t env_id_base,
                     uint32_t vec_step0, const rlhip_ppo_traj* traj, rlhip_stream_t stream);
int32_t ppo3_grad(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                  const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace, float* grad_out,
                  float* losses_out, bool do_pack, rlhip_stream_t stream);
int32_t ppo3_update(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                    float* params, float* m, float* v, float* beta_pow, uint64_t seed, uint32_t update_ctr,
                    void* workspace, float* grad_scratch, float* losses_out, rlhip_stream_t stream);

static inline int64_t env_na(int kind, int cont) { return cont ? 1 : (kind == 0 ? 2 : 3); }

static inline int32_t make_desc(int32_t kind, const rlhip_ppo_cfg* c, PolicyDesc* pd) {
    RLHIP_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0, 1 or 2");
    RLHIP_REQUIRE(c != nullptr, "ppo cfg is NULL");
    RLHIP_REQUIRE(c->hidden >= 4 && c->hidden % 4 == 0, "hidden must be a positive multiple of 4");
    RLHIP_REQUIRE(c->act == 0 || c->act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE(c->normalize_advantage == 0, "normalize_advantage is not supported yet");
    RLHIP_REQUIRE(c->layers == 0 || c->layers == 2 || c->layers == 3, "layers must be 2 or 3");
    int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    pd->h = c->hidden;
    pd->act = c->act;
    pd->cont = c->continuous ? 1 : 0;
    pd->na = (int)env_na(kind, pd->cont);
    pd->nout_a = pd->cont ? 2 * pd->na : pd->na;
    pd->np_a = mlp2_nparams(ns, c->hidden, pd->nout_a);
    pd->gamma = c->gamma;
    pd->lambda = c->lambda;
    return RLHIP_OK;
}

}  // namespace rlhip
