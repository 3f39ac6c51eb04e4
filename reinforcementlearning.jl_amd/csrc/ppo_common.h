// ppo_common.h -- small host/device structs shared by ppo.hip (rollout / plan) and ppo_grad.hip.
#pragma once
#include "common.h"
#include "mlp_device.h"

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
};

static inline int64_t env_na(int kind, int cont) { return cont ? 1 : (kind == 0 ? 2 : 3); }

static inline int32_t make_desc(int32_t kind, const rlhip_ppo_cfg* c, PolicyDesc* pd) {
    RLHIP_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0, 1 or 2");
    RLHIP_REQUIRE(c != nullptr, "ppo cfg is NULL");
    RLHIP_REQUIRE(c->hidden >= 4 && c->hidden % 4 == 0, "hidden must be a positive multiple of 4");
    RLHIP_REQUIRE(c->act == 0 || c->act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE(c->normalize_advantage == 0, "normalize_advantage is not supported yet");
    int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    pd->h = c->hidden;
    pd->act = c->act;
    pd->cont = c->continuous ? 1 : 0;
    pd->na = (int)env_na(kind, pd->cont);
    pd->nout_a = pd->cont ? 2 * pd->na : pd->na;
    pd->np_a = mlp2_nparams(ns, c->hidden, pd->nout_a);
    return RLHIP_OK;
}

}  // namespace rlhip
