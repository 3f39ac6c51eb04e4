/* rlo_envs_impl.h -- type-generic body of the classic-control envs (+ Acrobot).  Included twice by
 * rlo_envs.c with T = float (SFX f32) and T = double (SFX f64).  TEST INFRASTRUCTURE ONLY.
 *
 * Julia promotion rules that matter: Float32 op Float64 -> Float64;
 * Int op Float32 -> Float32; Float32 + pi -> Float32 but 2 * pi -> Float64; `a*b*c` folds left;
 * no FMA contraction; stores into Vector{T} / ::T fields round to T.  With T = double every cast
 * below is a no-op, so one body serves both element types.
 */

/* ---- CartPoleEnv --------------------------------------------------------------------------- */
typedef struct {
    T gravity, masscart, masspole, totalmass, halflength, polemasslength, forcemag, dt,
        thetathreshold, xthreshold;
    int64_t max_steps;
    int continuous;
} NAME(cartpole_params);

/* CartPoleEnvParams{T}(; kwargs...)  RLEnvs/CartPoleEnv.jl:22-46: derived fields are computed in
 * Float64 from the Float64 kwargs and only then converted to T by the struct constructor. */
static void NAME(cartpole_make)(const rlo_cartpole_cfg* c, NAME(cartpole_params) * p) {
    p->gravity = (T)c->gravity;
    p->masscart = (T)c->masscart;
    p->masspole = (T)c->masspole;
    p->totalmass = (T)(c->masscart + c->masspole);
    p->halflength = (T)c->halflength;
    p->polemasslength = (T)(c->masspole * c->halflength);
    p->forcemag = (T)c->forcemag;
    p->dt = (T)c->dt;
    p->thetathreshold = (T)(c->thetathreshold_deg * M_PI / 180);
    p->xthreshold = (T)c->xthreshold;
    p->max_steps = c->max_steps;
    p->continuous = c->continuous;
}

/* reset!  RLEnvs/CartPoleEnv.jl:98-104:  state[:] = T(0.1) * rand(rng, T, 4) .- T(0.05); t = 0;
 * (one more draw for env.action -- moot for a counter-based RNG); done = false */
static void NAME(cartpole_reset1)(rlo_env_state* st, int64_t i, uint64_t seed, uint32_t env_id) {
    T u[4];
    uint32_t w[4];
    uint32_t ep = st->episode[i];
#if IS_F64
    rlo_philox4x32_10(seed, env_id, 0, ep, RLO_TAG_RESET, w);
    u[0] = rlo_u01_f64(w[0], w[1]);
    u[1] = rlo_u01_f64(w[2], w[3]);
    rlo_philox4x32_10(seed, env_id, 1, ep, RLO_TAG_RESET, w);
    u[2] = rlo_u01_f64(w[0], w[1]);
    u[3] = rlo_u01_f64(w[2], w[3]);
#else
    rlo_philox4x32_10(seed, env_id, 0, ep, RLO_TAG_RESET, w);
    for (int k = 0; k < 4; ++k) u[k] = rlo_u01_f32(w[k]);
#endif
    for (int k = 0; k < 4; ++k) ((T*)st->s[k])[i] = (T)0.1 * u[k] - (T)0.05;
    st->t[i] = 0;
    st->episode[i] = ep + 1;
}

/* act! + _step!  RLEnvs/CartPoleEnv.jl:106-140; reward :84 */
static void NAME(cartpole_step1)(const NAME(cartpole_params) * p, rlo_env_state* st, int64_t i,
                                 const void* actions) {
    T* S0 = (T*)st->s[0];
    T* S1 = (T*)st->s[1];
    T* S2 = (T*)st->s[2];
    T* S3 = (T*)st->s[3];
    T a;
    if (p->continuous) {
        a = ((const T*)actions)[i]; /* :106-110 force = a * forcemag, a of type T in the vec env */
    } else {
        a = (((const int32_t*)actions)[i] == 1) ? (T)1 : (T)-1; /* :115  a == 2 ? 1 : -1 (0-based: 1) */
    }
    st->t[i] += 1;                                   /* :119 */
    T force = a * p->forcemag;                       /* :120 */
    T x = S0[i], xdot = S1[i], theta = S2[i], thetadot = S3[i]; /* :121 pre-step values */
    T costheta = COS(theta);                         /* :122 */
    T sintheta = SIN(theta);                         /* :123 */
    T tmp = (force + p->polemasslength * (thetadot * thetadot) * sintheta) / p->totalmass; /* :124 */
    /* :125-129 -- the literal 4 / 3 is Float64, so denominator, thetaacc and xacc are Float64 */
    T num = p->gravity * sintheta - costheta * tmp;
    T frac = p->masspole * (costheta * costheta) / p->totalmass;
    double den = (double)p->halflength * (4.0 / 3.0 - (double)frac);
    double thetaacc = (double)num / den;
    double xacc =
        (double)tmp - (double)p->polemasslength * thetaacc * (double)costheta / (double)p->totalmass; /* :130 */
    S0[i] = x + p->dt * xdot;                                 /* :131 pure T */
    S1[i] = (T)((double)xdot + (double)p->dt * xacc);         /* :132 Float64 sum, stored as T */
    S2[i] = theta + p->dt * thetadot;                         /* :133 */
    S3[i] = (T)((double)thetadot + (double)p->dt * thetaacc); /* :134 */
    int done = FABS(S0[i]) > p->xthreshold || FABS(S2[i]) > p->thetathreshold ||
               (int64_t)st->t[i] > p->max_steps;              /* :135-138 strict > */
    st->done[i] = (uint8_t)done;
    ((T*)st->reward)[i] = done ? (T)0 : (T)1;                 /* :84 */
}

/* ---- PendulumEnv --------------------------------------------------------------------------- */
typedef struct {
    T max_speed, max_torque, g, m, l, dt;
    int64_t max_steps;
    int continuous, n_actions;
} NAME(pendulum_params);

static void NAME(pendulum_make)(const rlo_pendulum_cfg* c, NAME(pendulum_params) * p) {
    p->max_speed = (T)c->max_speed;
    p->max_torque = (T)c->max_torque;
    p->g = (T)c->g;
    p->m = (T)c->m;
    p->l = (T)c->l;
    p->dt = (T)c->dt;
    p->max_steps = c->max_steps;
    p->continuous = c->continuous;
    p->n_actions = c->n_actions;
}

/* Julia mod(x::Float64, y::Float64) (floored modulo built on the exact rem = fmod) */
static double NAME(jl_mod)(double x, double y) {
    double r = fmod(x, y);
    if (r == 0) return copysign(r, y);
    if ((r > 0) != (y > 0)) return r + y;
    return r;
}

/* reset!  RLEnvs/PendulumEnv.jl:84-92 */
static void NAME(pendulum_reset1)(rlo_env_state* st, int64_t i, uint64_t seed, uint32_t env_id) {
    uint32_t w[4];
    uint32_t ep = st->episode[i];
    rlo_philox4x32_10(seed, env_id, 0, ep, RLO_TAG_RESET, w);
#if IS_F64
    T u0 = rlo_u01_f64(w[0], w[1]), u1 = rlo_u01_f64(w[2], w[3]);
#else
    T u0 = rlo_u01_f32(w[0]), u1 = rlo_u01_f32(w[1]);
#endif
    /* :85  2 * pi * (rand(T) - 1): (2*pi) is Float64, product Float64, stored as T */
    ((T*)st->s[0])[i] = (T)((2.0 * M_PI) * (double)(u0 - (T)1));
    /* :86  2 * (rand(T) - 1): pure T */
    ((T*)st->s[1])[i] = (T)2 * (u1 - (T)1);
    st->t[i] = 0;
    st->episode[i] = ep + 1;
}

/* act! + torque + _step!  RLEnvs/PendulumEnv.jl:94-122 */
static void NAME(pendulum_step1)(const NAME(pendulum_params) * p, rlo_env_state* st, int64_t i,
                                 const void* actions) {
    T* TH = (T*)st->s[0];
    T* THD = (T*)st->s[1];
    T a;
    if (p->continuous) {
        a = ((const T*)actions)[i]; /* :122 torque(env, a) = a */
    } else {
        /* :120-121 (4 / (n - 1)) * (a - (n - 1) / 2 - 1) in Float64, a 1-based; env.action::T rounds */
        double a1 = (double)(((const int32_t*)actions)[i] + 1);
        double nm1 = (double)(p->n_actions - 1);
        a = (T)((4.0 / nm1) * (a1 - nm1 / 2.0 - 1.0));
    }
    st->t[i] += 1; /* :101 */
    T th = TH[i], thdot = THD[i]; /* :102 */
    /* :103 clamp(a, -max_torque, max_torque) */
    a = (a > p->max_torque) ? p->max_torque : ((a < -p->max_torque) ? -p->max_torque : a);
    /* :104 costs = angle_normalize(th)^2 + 0.1 * thdot^2 + 0.001 * a^2  -- Float64:
     * angle_normalize(x) = mod(x + pi, 2 * pi) - pi (:71): x + pi is T, 2 * pi is Float64 */
    T thpi = th + (T)M_PI;
    double an = NAME(jl_mod)((double)thpi, 2.0 * M_PI) - M_PI;
    double costs = an * an + 0.1 * (double)(thdot * thdot) + 0.001 * (double)(a * a);
    /* :105-110 pure T; sin(th + pi) literally */
    T newthdot = thdot + ((T)-3 * p->g / ((T)2 * p->l) * SIN(th + (T)M_PI) +
                          (T)3 * a / (p->m * (p->l * p->l))) *
                             p->dt;
    th = th + newthdot * p->dt; /* :111 unclamped newthdot */
    newthdot = (newthdot > p->max_speed) ? p->max_speed
                                         : ((newthdot < -p->max_speed) ? -p->max_speed : newthdot); /* :112 */
    TH[i] = th;
    THD[i] = newthdot;
    st->done[i] = (uint8_t)((int64_t)st->t[i] >= p->max_steps); /* :115 */
    ((T*)st->reward)[i] = (T)(-costs);                            /* :116, field ::T */
}

/* ---- MountainCarEnv ------------------------------------------------------------------------ */
typedef struct {
    T min_pos, max_pos, max_speed, goal_pos, goal_velocity, power, gravity;
    int64_t max_steps;
    int continuous;
} NAME(mountaincar_params);

static void NAME(mountaincar_make)(const rlo_mountaincar_cfg* c, NAME(mountaincar_params) * p) {
    p->min_pos = (T)c->min_pos;
    p->max_pos = (T)c->max_pos;
    p->max_speed = (T)c->max_speed;
    p->goal_pos = (T)c->goal_pos;
    p->goal_velocity = (T)c->goal_velocity;
    p->power = (T)c->power;
    p->gravity = (T)c->gravity;
    p->max_steps = c->max_steps;
    p->continuous = c->continuous;
}

/* reset!  RLEnvs/MountainCarEnv.jl:99-105: x = 0.2 * rand(T) - 0.6 (Float64 arithmetic), v = 0 */
static void NAME(mountaincar_reset1)(rlo_env_state* st, int64_t i, uint64_t seed, uint32_t env_id) {
    uint32_t w[4];
    uint32_t ep = st->episode[i];
    rlo_philox4x32_10(seed, env_id, 0, ep, RLO_TAG_RESET, w);
#if IS_F64
    T u0 = rlo_u01_f64(w[0], w[1]);
#else
    T u0 = rlo_u01_f32(w[0]);
#endif
    ((T*)st->s[0])[i] = (T)(0.2 * (double)u0 - 0.6);
    ((T*)st->s[1])[i] = (T)0;
    st->t[i] = 0;
    st->episode[i] = ep + 1;
}

/* act! + _step!  RLEnvs/MountainCarEnv.jl:107-135; reward :95 */
static void NAME(mountaincar_step1)(const NAME(mountaincar_params) * p, rlo_env_state* st, int64_t i,
                                    const void* actions) {
    T* X = (T*)st->s[0];
    T* V = (T*)st->s[1];
    T force;
    if (p->continuous) {
        force = ((const T*)actions)[i]; /* :107-111, action of type T in the vec env */
    } else {
        force = (T)(((const int32_t*)actions)[i] - 1); /* :117 a - 2 (1-based) == a0 - 1 */
    }
    st->t[i] += 1; /* :120 */
    T x = X[i], v = V[i];
    v = v + (force * p->power + COS((T)3 * x) * (-p->gravity)); /* :122 */
    v = (v > p->max_speed) ? p->max_speed : ((v < -p->max_speed) ? -p->max_speed : v); /* :123 */
    x = x + v;                                                    /* :124 */
    x = (x > p->max_pos) ? p->max_pos : ((x < p->min_pos) ? p->min_pos : x); /* :125 */
    if (x == p->min_pos && v < (T)0) v = (T)0;                  /* :126-128 */
    int done = (x >= p->goal_pos && v >= p->goal_velocity) || (int64_t)st->t[i] >= p->max_steps; /* :129-131 */
    X[i] = x;
    V[i] = v;
    st->done[i] = (uint8_t)done;
    ((T*)st->reward)[i] = done ? (T)0 : (T)-1; /* :95 */
}

/* ---- AcrobotEnv (3rd_party/AcrobotEnv.jl) -- parity unpinned, see rl_oracle.h ------------------------------- */
typedef struct {
    double m1, m2, l1, lc1, lc2, I1, I2, g, dt;
    T max_vel_a, max_vel_b, noise;
    int64_t max_steps;
    int nips;
} NAME(acrobot_params);

static void NAME(acrobot_make)(const rlo_acrobot_cfg* c, NAME(acrobot_params) * p) {
    /* AcrobotEnvParams{T}: every field is stored as T (:42-56); dsdt reads them back (:149-156) */
    p->m1 = (double)(T)c->link_mass_a;
    p->m2 = (double)(T)c->link_mass_b;
    p->l1 = (double)(T)c->link_length_a;
    p->lc1 = (double)(T)c->link_com_pos_a;
    p->lc2 = (double)(T)c->link_com_pos_b;
    p->I1 = (double)(T)c->link_moi;
    p->I2 = (double)(T)c->link_moi;
    p->g = (double)(T)c->g;
    p->dt = (double)(T)c->dt;
    p->max_vel_a = (T)c->max_vel_a;
    p->max_vel_b = (T)c->max_vel_b;
    p->noise = (T)c->max_torque_noise;
    p->max_steps = c->max_steps;
    p->nips = c->nips;
}

/* reset!  :94-101: state = T(0.1) * rand(rng, T, 4) .- T(0.05); t = 0; done = false; reward = -1 */
static void NAME(acrobot_reset1)(rlo_env_state* st, int64_t i, uint64_t seed, uint32_t env_id) {
    NAME(cartpole_reset1)(st, i, seed, env_id); /* the same four draws and the same expression */
}

/* dsdt  :147-199 (du[5] = 0: the torque a is constant over the step) */
static void NAME(acrobot_dsdt)(const NAME(acrobot_params) * p, const double s[4], double a, double du[4]) {
    const double m1 = p->m1, m2 = p->m2, l1 = p->l1, lc1 = p->lc1, lc2 = p->lc2, I1 = p->I1, I2 = p->I2, g = p->g;
    const double theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
    const double c2 = cos(theta2), s2 = sin(theta2);
    double d1 = ((m1 * (lc1 * lc1) + m2 * ((l1 * l1 + lc2 * lc2) + ((2 * l1) * lc2) * c2)) + I1) + I2; /* :171 */
    double d2 = m2 * (lc2 * lc2 + (l1 * lc2) * c2) + I2;                                              /* :172 */
    double phi2 = ((m2 * lc2) * g) * cos((theta1 + theta2) - M_PI / 2.0);                             /* :173 */
    double phi1 = (((((-m2) * l1) * lc2) * (dtheta2 * dtheta2)) * s2 -
                   (((((2 * m2) * l1) * lc2) * dtheta2) * dtheta1) * s2) +
                  ((m1 * lc1 + m2 * l1) * g) * cos(theta1 - M_PI / 2);                                /* :174-179 */
    phi1 = phi1 + phi2;
    double ddtheta1 = 0.0, ddtheta2;
    if (p->nips) {
        ddtheta2 = ((a + (d2 / d1) * phi1) - phi2) / ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1); /* :183 */
    } else {
        ddtheta2 = (((a + (d2 / d1) * phi1) - (((m2 * l1) * lc2) * (dtheta1 * dtheta1)) * s2) - phi2) /
                   ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1);                                    /* :187-190 */
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;                                                  /* :191 */
    }
    du[0] = dtheta1;
    du[1] = dtheta2;
    du[2] = ddtheta1;
    du[3] = ddtheta2;
}

static double NAME(acrobot_wrap)(double x, double m, double M) { /* :204-222 */
    double diff = M - m;
    while (x > M) x = x - diff;
    while (x < m) x = x + diff;
    return x;
}

/* act!  :104-145 */
static void NAME(acrobot_step1)(const NAME(acrobot_params) * p, rlo_env_state* st, int64_t i, const void* actions,
                                uint64_t seed, uint32_t env_id) {
    st->t[i] += 1;                                             /* :106 */
    T torque = (T)(((const int32_t*)actions)[i] - 1);          /* :107 avail_torque = [-1, 0, 1] */
    if (p->noise > (T)0) {                                     /* :110-113 */
        uint32_t w[4];
        rlo_philox4x32_10(seed, env_id, (uint32_t)st->t[i], st->episode[i], RLO_TAG_ENVNOISE, w);
#if IS_F64
        T u = rlo_u01_f64(w[0], w[1]);
#else
        T u = rlo_u01_f32(w[0]);
#endif
        torque = (torque + (T)(2.0 * (double)p->noise) * u) - p->noise;
    }
    double y[4], k1[4], k2[4], k3[4], k4[4], yt[4];
    for (int k = 0; k < 4; ++k) y[k] = (double)((const T*)st->s[k])[i];
    const double a = (double)torque, h = p->dt, h2 = h / 2.0;
    NAME(acrobot_dsdt)(p, y, a, k1);
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h2 * k1[k];
    NAME(acrobot_dsdt)(p, yt, a, k2);
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h2 * k2[k];
    NAME(acrobot_dsdt)(p, yt, a, k3);
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h * k3[k];
    NAME(acrobot_dsdt)(p, yt, a, k4);
    double ns[4];
    for (int k = 0; k < 4; ++k) ns[k] = y[k] + (h / 6.0) * (((k1[k] + 2 * k2[k]) + 2 * k3[k]) + k4[k]);
    ns[0] = NAME(acrobot_wrap)(ns[0], -M_PI, M_PI);            /* :135-136 */
    ns[1] = NAME(acrobot_wrap)(ns[1], -M_PI, M_PI);
    double va = (double)p->max_vel_a, vb = (double)p->max_vel_b;
    ns[2] = fmin(fmax(ns[2], -va), va);                        /* :137-138 bound */
    ns[3] = fmin(fmax(ns[3], -vb), vb);
    T sT[4];
    for (int k = 0; k < 4; ++k) {
        sT[k] = (T)ns[k];
        ((T*)st->s[k])[i] = sT[k];
    }
    int succeeded = (-cos((double)sT[0]) - cos((double)sT[1] + (double)sT[0])) > 1.0; /* :141 */
    int done = succeeded || (int64_t)st->t[i] > p->max_steps;                          /* :142 */
    st->done[i] = (uint8_t)done;
    ((T*)st->reward)[i] = succeeded ? (T)0 : (T)-1;                                    /* :143 */
}

/* ---- drivers ------------------------------------------------------------------------------- */
static void NAME(write_obs1)(int kind, const rlo_env_state* st, int64_t n, int64_t i, T* obs) {
    if (kind == 0) {
        for (int k = 0; k < 4; ++k) obs[k * n + i] = ((const T*)st->s[k])[i]; /* CartPoleEnv.jl:86 */
    } else if (kind == 1) {
        T th = ((const T*)st->s[0])[i];
        obs[0 * n + i] = SIN(th); /* PendulumEnv.jl:70 */
        obs[1 * n + i] = COS(th);
        obs[2 * n + i] = ((const T*)st->s[1])[i];
    } else if (kind == 2) {
        for (int k = 0; k < 2; ++k) obs[k * n + i] = ((const T*)st->s[k])[i]; /* MountainCarEnv.jl:97 */
    } else {
        T a = ((const T*)st->s[0])[i], b = ((const T*)st->s[1])[i];
        obs[0 * n + i] = COS(a); /* acrobot_observation  AcrobotEnv.jl:73 */
        obs[1 * n + i] = SIN(a);
        obs[2 * n + i] = COS(b);
        obs[3 * n + i] = SIN(b);
        obs[4 * n + i] = ((const T*)st->s[2])[i];
        obs[5 * n + i] = ((const T*)st->s[3])[i];
    }
}

static int NAME(env_reset)(int kind, const void* cfg, rlo_env_state* st, int64_t n, uint64_t seed,
                           uint32_t env_id_base, const uint8_t* mask) {
    (void)cfg;
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        uint32_t id = env_id_base + (uint32_t)i;
        if (kind == 0) NAME(cartpole_reset1)(st, i, seed, id);
        else if (kind == 1) NAME(pendulum_reset1)(st, i, seed, id);
        else if (kind == 2) NAME(mountaincar_reset1)(st, i, seed, id);
        else NAME(acrobot_reset1)(st, i, seed, id);
        st->done[i] = 0;
        ((T*)st->reward)[i] = kind == 3 ? (T)-1 : (T)0; /* AcrobotEnv.jl:99 reward = -1 */
    }
    return 0;
}

static int NAME(env_step)(int kind, const void* cfg, rlo_env_state* st, int64_t n,
                          const void* actions, int auto_reset, uint64_t seed, uint32_t env_id_base,
                          void* last_obs) {
    NAME(cartpole_params) cp;
    NAME(pendulum_params) pp;
    NAME(mountaincar_params) mp;
    NAME(acrobot_params) ap;
    if (kind == 3) NAME(acrobot_make)((const rlo_acrobot_cfg*)cfg, &ap);
    else if (kind == 0) NAME(cartpole_make)((const rlo_cartpole_cfg*)cfg, &cp);
    else if (kind == 1) NAME(pendulum_make)((const rlo_pendulum_cfg*)cfg, &pp);
    else NAME(mountaincar_make)((const rlo_mountaincar_cfg*)cfg, &mp);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        if (kind == 0) NAME(cartpole_step1)(&cp, st, i, actions);
        else if (kind == 1) NAME(pendulum_step1)(&pp, st, i, actions);
        else if (kind == 2) NAME(mountaincar_step1)(&mp, st, i, actions);
        else NAME(acrobot_step1)(&ap, st, i, actions, seed, env_id_base + (uint32_t)i);
        if (last_obs) NAME(write_obs1)(kind, st, n, i, (T*)last_obs);
        if (auto_reset && st->done[i]) {
            /* MultiThreadEnv protocol: reward/terminal of the finished step stay visible, the
             * state is replaced by a fresh episode start */
            uint32_t id = env_id_base + (uint32_t)i;
            if (kind == 0) NAME(cartpole_reset1)(st, i, seed, id);
            else if (kind == 1) NAME(pendulum_reset1)(st, i, seed, id);
            else if (kind == 2) NAME(mountaincar_reset1)(st, i, seed, id);
            else NAME(acrobot_reset1)(st, i, seed, id);
        }
    }
    return 0;
}
