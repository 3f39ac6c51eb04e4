/* rlo_envs.c -- CartPole / Pendulum / MountainCar CPU restatement, Float32 and Float64 element
 * types.  TEST INFRASTRUCTURE ONLY (see rl_oracle.h).  Bodies in rlo_envs_impl.h. */
#define _GNU_SOURCE
#include "rl_oracle.h"
#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

/* ---- Float32 instantiation ---- */
#define T float
#define IS_F64 0
#define NAME(x) CAT(x, _f32)
#define SIN sinf
#define COS cosf
#define FABS fabsf
#include "rlo_envs_impl.h"
#undef T
#undef IS_F64
#undef NAME
#undef SIN
#undef COS
#undef FABS

/* ---- Float64 instantiation ---- */
#define T double
#define IS_F64 1
#define NAME(x) CAT(x, _f64)
#define SIN sin
#define COS cos
#define FABS fabs
#include "rlo_envs_impl.h"
#undef T
#undef IS_F64
#undef NAME
#undef SIN
#undef COS
#undef FABS

void rlo_cartpole_default(rlo_cartpole_cfg* c) {
    /* RLEnvs/CartPoleEnv.jl:22-32 */
    c->gravity = 9.8;
    c->masscart = 1.0;
    c->masspole = 0.1;
    c->halflength = 0.5;
    c->forcemag = 10.0;
    c->dt = 0.02;
    c->thetathreshold_deg = 12.0;
    c->xthreshold = 2.4;
    c->max_steps = 200;
    c->continuous = 0;
}

void rlo_pendulum_default(rlo_pendulum_cfg* c) {
    /* RLEnvs/PendulumEnv.jl:41-53 */
    c->max_speed = 8;
    c->max_torque = 2;
    c->g = 10;
    c->m = 1;
    c->l = 1;
    c->dt = 0.05;
    c->max_steps = 200;
    c->continuous = 1;
    c->n_actions = 3;
}

void rlo_mountaincar_default(rlo_mountaincar_cfg* c, int continuous) {
    /* RLEnvs/MountainCarEnv.jl:19-29; continuous overrides :74 */
    c->min_pos = -1.2;
    c->max_pos = 0.6;
    c->max_speed = 0.07;
    c->goal_pos = continuous ? 0.45 : 0.5;
    c->goal_velocity = 0.0;
    c->power = continuous ? 0.0015 : 0.001;
    c->gravity = 0.0025;
    c->max_steps = 200;
    c->continuous = continuous;
}

void rlo_acrobot_default(rlo_acrobot_cfg* c) {
    /* RLEnvs/src/environments/3rd_party/AcrobotEnv.jl:22-40 */
    c->link_length_a = 1.0;
    c->link_length_b = 1.0;
    c->link_mass_a = 1.0;
    c->link_mass_b = 1.0;
    c->link_com_pos_a = 0.5;
    c->link_com_pos_b = 0.5;
    c->link_moi = 1.0;
    c->max_torque_noise = 0.0;
    c->max_vel_a = 4 * M_PI;
    c->max_vel_b = 9 * M_PI;
    c->g = 9.8;
    c->dt = 0.2;
    c->max_steps = 200;
    c->nips = 0;
}

int rlo_env_obs_dim(int kind) { return kind == 0 ? 4 : (kind == 1 ? 3 : (kind == 2 ? 2 : 6)); }
int rlo_env_state_dim(int kind) { return (kind == 0 || kind == 3) ? 4 : 2; }

int rlo_env_reset(int kind, int is_f64, const void* cfg, rlo_env_state* st, int64_t n,
                  uint64_t seed, uint32_t env_id_base, const uint8_t* mask) {
    if (kind < 0 || kind > 3) return -1;
    return is_f64 ? env_reset_f64(kind, cfg, st, n, seed, env_id_base, mask)
                  : env_reset_f32(kind, cfg, st, n, seed, env_id_base, mask);
}

int rlo_env_step(int kind, int is_f64, const void* cfg, rlo_env_state* st, int64_t n,
                 const void* actions, int auto_reset, uint64_t seed, uint32_t env_id_base,
                 void* last_obs) {
    if (kind < 0 || kind > 3) return -1;
    return is_f64 ? env_step_f64(kind, cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs)
                  : env_step_f32(kind, cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs);
}

int rlo_env_obs(int kind, int is_f64, const rlo_env_state* st, int64_t n, void* obs) {
    if (kind < 0 || kind > 3) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (is_f64) write_obs1_f64(kind, st, n, i, (double*)obs);
        else write_obs1_f32(kind, st, n, i, (float*)obs);
    }
    return 0;
}
