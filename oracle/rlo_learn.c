/* rlo_learn.c -- MLP forward/backward, PPO and DQN loss/gradient, whole PPO iteration on the CPU.
 * TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 *
 * None of the learners is in the reference tree any more (removed ReinforcementLearningZoo; SURVEY.md
 * section 0).  What IS in tree and is followed here:
 *   ActorCritic container                 RLCore/utils/networks.jl:15-20
 *   CategoricalNetwork Gumbel-max sample  RLCore/utils/networks.jl:425-432   (rlo_select.c)
 *   GaussianNetwork sample / logpdf       RLCore/utils/networks.jl:64-82, distributions.jl:18-21
 *   generalized_advantage_estimation      RLCore/utils/basic.jl:334-417      (rlo_scans.c)
 *   clip_by_global_norm!                  RLCore/utils/basic.jl:19-29        (rlo_optim.c)
 *   FluxApproximator.optimise! call site  RLCore/policies/learners/flux_approximator.jl:46
 *   vector-env run loop                   docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:351-374
 *   PPO hyper-parameters / net shapes     docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15238-15287
 * The PPO / DQN loss formulas are the published RLZoo v0.10 ones as recorded in SURVEY.md
 * Appendix B -- PARITY UNPINNED; gradients are checked against torch autograd in tests/.
 */
#include "rl_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>

int64_t rlo_mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out) {
    return h * n_in + h + n_out * h + n_out;
}

static inline float act_fwd(int act, float z) { return act == 0 ? (z > 0.0f ? z : 0.0f) : tanhf(z); }
/* derivative given pre-activation z and activation value hv */
static inline float act_bwd(int act, float z, float hv) {
    return act == 0 ? (z > 0.0f ? 1.0f : 0.0f) : (1.0f - hv * hv);
}

/* one sample; hid (h) and zbuf (h) optional scratch outputs */
static void mlp2_forward1(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                          const float* x, int64_t xstride, float* out, int64_t ostride, float* hid,
                          float* zbuf) {
    const float* W1 = p;
    const float* b1 = W1 + h * n_in;
    const float* W2 = b1 + h;
    const float* b2 = W2 + n_out * h;
    for (int64_t o = 0; o < n_out; ++o) out[o * ostride] = b2[o];
    for (int64_t j = 0; j < h; ++j) {
        float z = b1[j];
        for (int64_t k = 0; k < n_in; ++k) z = fmaf(W1[j + h * k], x[k * xstride], z);
        float hv = act_fwd(act, z);
        if (hid) hid[j] = hv;
        if (zbuf) zbuf[j] = z;
        for (int64_t o = 0; o < n_out; ++o)
            out[o * ostride] = fmaf(W2[o + n_out * j], hv, out[o * ostride]);
    }
}

void rlo_mlp2_forward_f32(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                          const float* x, int64_t batch, float* out) {
    for (int64_t i = 0; i < batch; ++i)
        mlp2_forward1(p, n_in, h, n_out, act, x + i, batch, out + i, batch, 0, 0);
}

/* accumulate gradient of one sample into double accumulators ga (same layout as p) */
static void mlp2_backward1(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                           const float* x, int64_t xstride, const float* dout, int64_t dstride,
                           double* ga, float* hid, float* zbuf) {
    const float* W1 = p;
    const float* b1 = W1 + h * n_in;
    const float* W2 = b1 + h;
    (void)b1;
    double* gW1 = ga;
    double* gb1 = gW1 + h * n_in;
    double* gW2 = gb1 + h;
    double* gb2 = gW2 + n_out * h;
    for (int64_t o = 0; o < n_out; ++o) gb2[o] += (double)dout[o * dstride];
    for (int64_t j = 0; j < h; ++j) {
        float dh = 0.0f;
        for (int64_t o = 0; o < n_out; ++o) {
            gW2[o + n_out * j] += (double)(dout[o * dstride] * hid[j]);
            dh = fmaf(dout[o * dstride], W2[o + n_out * j], dh);
        }
        float dz = dh * act_bwd(act, zbuf[j], hid[j]);
        gb1[j] += (double)dz;
        for (int64_t k = 0; k < n_in; ++k) gW1[j + h * k] += (double)(dz * x[k * xstride]);
    }
}

void rlo_mlp2_backward_f32(const float* p, int64_t n_in, int64_t h, int64_t n_out, int act,
                           const float* x, int64_t batch, const float* dout, float* g) {
    int64_t np = rlo_mlp2_nparams(n_in, h, n_out);
    double* ga = (double*)calloc((size_t)np, sizeof(double));
    float* hid = (float*)malloc(sizeof(float) * (size_t)h * 2);
    float* zb = hid + h;
    float outtmp[64];
    for (int64_t i = 0; i < batch; ++i) {
        mlp2_forward1(p, n_in, h, n_out, act, x + i, batch, outtmp, 1, hid, zb);
        mlp2_backward1(p, n_in, h, n_out, act, x + i, batch, dout + i, batch, ga, hid, zb);
    }
    for (int64_t q = 0; q < np; ++q) g[q] += (float)ga[q];
    free(ga);
    free(hid);
}

/* glorot_uniform stand-in (Flux: U(-s, s), s = sqrt(24 / (fan_in + fan_out)) / 2 ... = sqrt(6/(in+out))).
 * tensor ids: net_id*4 + {0: W1, 2: W2}; biases zero (Flux Dense default bias = zeros). */
void rlo_mlp2_init_f32(float* p, int64_t n_in, int64_t h, int64_t n_out, uint64_t seed,
                       uint32_t net_id) {
    float* W1 = p;
    float* b1 = W1 + h * n_in;
    float* W2 = b1 + h;
    float* b2 = W2 + n_out * h;
    float s1 = sqrtf(6.0f / (float)(n_in + h));
    float s2 = sqrtf(6.0f / (float)(h + n_out));
    for (int64_t q = 0; q < h * n_in; ++q) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)q, 0, net_id * 4u + 0u, RLO_TAG_INIT, w);
        W1[q] = (2.0f * rlo_u01_f32(w[0]) - 1.0f) * s1;
    }
    for (int64_t q = 0; q < h; ++q) b1[q] = 0.0f;
    for (int64_t q = 0; q < n_out * h; ++q) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)q, 0, net_id * 4u + 2u, RLO_TAG_INIT, w);
        W2[q] = (2.0f * rlo_u01_f32(w[0]) - 1.0f) * s2;
    }
    for (int64_t q = 0; q < n_out; ++q) b2[q] = 0.0f;
}

/* ------------------------------------------------------------------------------------- PPO -- */
void rlo_ppo_default(rlo_ppo_cfg* c) {
    /* a_practical_introduction_to_RL.jl/index.html:15257-15278 */
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
}

static int64_t ppo_actor_nout(const rlo_ppo_cfg* c, int64_t na) { return c->continuous ? 2 * na : na; }

/* actor / critic network of the PPO policy: cfg.layers = 2 (ns -> h -> nout, f32) or 3 (ns -> 128 -> 128 -> nout with
 * the bf16 hidden layer of rlo_mlp3.c).  Scratch per sample: NET_SCRATCH(h) floats + h doubles. */
#define NET_SCRATCH(h) (5 * (h))
static int net_layers(const rlo_ppo_cfg* c) { return c->layers == 3 ? 3 : 2; }
static int64_t net_nparams(const rlo_ppo_cfg* c, int64_t ns, int64_t nout) {
    return net_layers(c) == 3 ? rlo_mlp3_nparams(ns, c->hidden, nout) : rlo_mlp2_nparams(ns, c->hidden, nout);
}
static void net_forward1(const rlo_ppo_cfg* c, const float* p, int64_t ns, int64_t nout, const float* x,
                         int64_t xstride, float* out, float* scr) {
    int64_t h = c->hidden;
    if (net_layers(c) == 3)
        rlo_mlp3_forward1(p, ns, h, nout, c->act, x, xstride, out, 1, scr, scr + h, scr + 2 * h, scr + 3 * h);
    else
        mlp2_forward1(p, ns, h, nout, c->act, x, xstride, out, 1, scr, scr + h);
}
static void net_backward1(const rlo_ppo_cfg* c, const float* p, int64_t ns, int64_t nout, const float* x,
                          int64_t xstride, const float* dout, double* ga, float* scr, double* dscr) {
    int64_t h = c->hidden;
    if (net_layers(c) == 3)
        rlo_mlp3_backward1(p, ns, h, nout, c->act, x, xstride, dout, ga, scr, scr + h, scr + 2 * h, scr + 3 * h,
                           scr + 4 * h, dscr);
    else
        mlp2_backward1(p, ns, h, nout, c->act, x, xstride, dout, 1, ga, scr, scr + h);
}

int64_t rlo_ppo_nparams(int kind, const rlo_ppo_cfg* c) {
    int64_t ns = rlo_env_obs_dim(kind);
    int64_t na = (kind == 0) ? 2 : (kind == 1 ? (c->continuous ? 1 : 3) : (c->continuous ? 1 : 3));
    return net_nparams(c, ns, ppo_actor_nout(c, na)) + net_nparams(c, ns, 1);
}

static inline float log2pi_f32(void) { return logf(6.2831855f); }

/* one sample of the PPO loss: forward, loss terms, backward into the Float64 accumulators ga */
static void ppo_sample(const rlo_ppo_cfg* c, int64_t ns, int64_t na, int64_t h, int64_t nout_a, int64_t np_a,
                       const float* pa, const float* pc, const float* obs, const int32_t* act_i, const float* act_f,
                       const float* logp_old, const float* adv, const float* ret, int64_t bm, int64_t i, float lo,
                       float hi, float inv_b, float min_logp, double* ga, float* hid, float* zb, double* actor_acc_p,
                       double* critic_acc_p, double* ent_acc_p) {
    float out[64], dout[64];
    double actor_acc = 0, critic_acc = 0, ent_acc = 0;
    (void)h;
    {
        /* ---- actor ---- */
        net_forward1(c, pa, ns, nout_a, obs + i, bm, out, hid);
        float lp_old = logp_old[i] < min_logp ? min_logp : logp_old[i];
        float lp_new, ent;
        if (!c->continuous) {
            float mx = out[0];
            for (int64_t k = 1; k < na; ++k)
                if (out[k] > mx) mx = out[k];
            float se = 0.0f;
            for (int64_t k = 0; k < na; ++k) se += expf(out[k] - mx);
            float lse = logf(se);
            float logp[64], pr[64];
            ent = 0.0f;
            for (int64_t k = 0; k < na; ++k) {
                logp[k] = (out[k] - mx) - lse;
                pr[k] = expf(logp[k]);
                ent -= pr[k] * logp[k];
            }
            int32_t a = act_i[i];
            lp_new = logp[a];
            float ratio = expf(lp_new - lp_old);
            float A = adv[i];
            float surr1 = ratio * A;
            float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
            float surr2 = rc * A;
            actor_acc += (double)(surr1 < surr2 ? surr1 : surr2);
            int inside = (ratio >= lo && ratio <= hi);
            float dobj_dratio = (inside || surr1 < surr2) ? A : 0.0f;
            float dL_dlp = -c->actor_loss_weight * inv_b * dobj_dratio * ratio;
            for (int64_t k = 0; k < na; ++k) {
                float dlp = ((k == a) ? 1.0f : 0.0f) - pr[k];
                float dent = -pr[k] * (logp[k] + ent); /* dH/dl_k */
                dout[k] = dL_dlp * dlp - c->entropy_loss_weight * inv_b * dent;
            }
        } else {
            /* actor head: out[0..na) = mu, out[na..2na) = log sigma; sigma = exp(log sigma) */
            const float eps = 1.0e-8f;
            lp_new = 0.0f;
            float sum_ls = 0.0f;
            float dmu[32], dls[32];
            for (int64_t k = 0; k < na; ++k) {
                float mu = out[k], ls = out[na + k];
                float sg = expf(ls);
                float z = act_f[i + k * bm];
                float se = sg + eps;
                float zz = (z - mu) / se;
                lp_new += -(zz * zz + log2pi_f32()) / 2.0f - logf(se);
                sum_ls += ls;
                dmu[k] = (z - mu) / (se * se);
                dls[k] = ((z - mu) * (z - mu) / (se * se * se) - 1.0f / se) * sg;
            }
            ent = ((float)na * (log2pi_f32() + 1.0f) + sum_ls) / 2.0f;
            float ratio = expf(lp_new - lp_old);
            float A = adv[i];
            float surr1 = ratio * A;
            float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
            float surr2 = rc * A;
            actor_acc += (double)(surr1 < surr2 ? surr1 : surr2);
            int inside = (ratio >= lo && ratio <= hi);
            float dobj_dratio = (inside || surr1 < surr2) ? A : 0.0f;
            float dL_dlp = -c->actor_loss_weight * inv_b * dobj_dratio * ratio;
            for (int64_t k = 0; k < na; ++k) {
                dout[k] = dL_dlp * dmu[k];
                dout[na + k] = dL_dlp * dls[k] - c->entropy_loss_weight * inv_b * 0.5f;
            }
        }
        ent_acc += (double)ent;
        net_backward1(c, pa, ns, nout_a, obs + i, bm, dout, ga, hid, (double*)zb);
        /* ---- critic ---- */
        float v;
        net_forward1(c, pc, ns, 1, obs + i, bm, &v, hid);
        float dv = ret[i] - v;
        critic_acc += (double)(dv * dv);
        float dvout = -2.0f * c->critic_loss_weight * inv_b * dv;
        net_backward1(c, pc, ns, 1, obs + i, bm, &dvout, ga + np_a, hid, (double*)zb);
    }
    *actor_acc_p += actor_acc;
    *critic_acc_p += critic_acc;
    *ent_acc_p += ent_acc;
}

void rlo_ppo_loss_grad_f32(const rlo_ppo_cfg* c, int64_t ns, int64_t na, const float* params,
                           const float* obs, const int32_t* act_i, const float* act_f,
                           const float* logp_old, const float* adv_in, const float* ret, int64_t bm,
                           float* grad, float* losses_out) {
    int64_t h = c->hidden;
    int64_t nout_a = ppo_actor_nout(c, na);
    int64_t np_a = net_nparams(c, ns, nout_a);
    int64_t np_c = net_nparams(c, ns, 1);
    const float* pa = params;
    const float* pc = params + np_a;
    double* ga = (double*)calloc((size_t)(np_a + np_c), sizeof(double));
    /* per-sample scratch: NET_SCRATCH(h) floats of caches, then h doubles (3-layer backward) */
    float* hid = (float*)malloc(sizeof(float) * (size_t)NET_SCRATCH(h) + sizeof(double) * (size_t)h);
    float* zb = hid + NET_SCRATCH(h);
    float* adv = (float*)malloc(sizeof(float) * (size_t)bm);
    memcpy(adv, adv_in, sizeof(float) * (size_t)bm);
    if (c->normalize_advantage) { /* (A - mean) / clamp(std, 1e-8, 1000) with the corrected std */
        double mu = 0, s2 = 0;
        for (int64_t i = 0; i < bm; ++i) mu += adv[i];
        mu /= (double)bm;
        for (int64_t i = 0; i < bm; ++i) s2 += (adv[i] - mu) * (adv[i] - mu);
        double sd = sqrt(s2 / (double)(bm > 1 ? bm - 1 : 1));
        if (sd < 1e-8) sd = 1e-8;
        if (sd > 1000.0) sd = 1000.0;
        for (int64_t i = 0; i < bm; ++i) adv[i] = (float)((adv[i] - mu) / sd);
    }
    const float lo = 1.0f - c->clip_range, hi = 1.0f + c->clip_range;
    const float inv_b = 1.0f / (float)bm;
    const float min_logp = (float)log(1e-8); /* clamp!(log_p, log(1e-8), Inf) */
    /* samples are independent: with OpenMP (the all-cores CPU baseline build, -fopenmp) each thread owns a
     * contiguous chunk, its own Float64 gradient accumulator and scratch; accumulators are combined in thread
     * order.  Without OpenMP this is the single sequential pass it always was. */
    double actor_acc = 0, critic_acc = 0, ent_acc = 0;
#ifdef _OPENMP
    int64_t npar = np_a + np_c;
    int nthr = omp_get_max_threads();
    if (nthr > 1 && bm >= 4 * (int64_t)nthr) {
        double* gt = (double*)calloc((size_t)npar * (size_t)nthr, sizeof(double));
        double* accs = (double*)calloc((size_t)nthr * 3, sizeof(double));
#pragma omp parallel num_threads(nthr)
        {
            int tid = omp_get_thread_num();
            int64_t i0 = bm * tid / nthr, i1 = bm * (tid + 1) / nthr;
            float* hid_t = (float*)malloc(sizeof(float) * (size_t)NET_SCRATCH(h) + sizeof(double) * (size_t)h);
            double aa = 0, ca = 0, ea = 0;
            for (int64_t i = i0; i < i1; ++i)
                ppo_sample(c, ns, na, h, nout_a, np_a, pa, pc, obs, act_i, act_f, logp_old, adv, ret, bm, i, lo, hi,
                           inv_b, min_logp, gt + (size_t)npar * tid, hid_t, hid_t + NET_SCRATCH(h), &aa, &ca, &ea);
            accs[3 * tid] = aa;
            accs[3 * tid + 1] = ca;
            accs[3 * tid + 2] = ea;
            free(hid_t);
        }
        for (int t = 0; t < nthr; ++t) {
            for (int64_t q = 0; q < npar; ++q) ga[q] += gt[(size_t)npar * t + q];
            actor_acc += accs[3 * t];
            critic_acc += accs[3 * t + 1];
            ent_acc += accs[3 * t + 2];
        }
        free(gt);
        free(accs);
    } else
#endif
    {
        for (int64_t i = 0; i < bm; ++i)
            ppo_sample(c, ns, na, h, nout_a, np_a, pa, pc, obs, act_i, act_f, logp_old, adv, ret, bm, i, lo, hi, inv_b,
                       min_logp, ga, hid, zb, &actor_acc, &critic_acc, &ent_acc);
    }
    for (int64_t q = 0; q < np_a + np_c; ++q) grad[q] = (float)ga[q];
    float actor_loss = (float)(-actor_acc / (double)bm);
    float critic_loss = (float)(critic_acc / (double)bm);
    float entropy_loss = (float)(ent_acc / (double)bm);
    if (losses_out) {
        losses_out[0] = c->actor_loss_weight * actor_loss + c->critic_loss_weight * critic_loss -
                        c->entropy_loss_weight * entropy_loss;
        losses_out[1] = actor_loss;
        losses_out[2] = critic_loss;
        losses_out[3] = entropy_loss;
    }
    free(ga);
    free(hid);
    free(adv);
}

/* ------------------------------------------------------------------------------------- DQN -- */
float rlo_dqn_loss_grad_f32(int64_t ns, int64_t h, int64_t na, int act, const float* params,
                            const float* target_params, const float* s, const int32_t* a,
                            const float* r, const uint8_t* term, const float* s_next, int64_t b,
                            float gamma, float huber_delta, float* grad, const float* isw) {
    int64_t np = rlo_mlp2_nparams(ns, h, na);
    double* ga = (double*)calloc((size_t)np, sizeof(double));
    float* hid = (float*)malloc(sizeof(float) * (size_t)h * 2);
    float* zb = hid + h;
    float q[64], qn[64], dout[64];
    double acc = 0;
    for (int64_t i = 0; i < b; ++i) {
        mlp2_forward1(target_params, ns, h, na, act, s_next + i, b, qn, 1, 0, 0);
        float mx = qn[0];
        for (int64_t k = 1; k < na; ++k)
            if (qn[k] > mx) mx = qn[k];
        float cont = term[i] ? 0.0f : 1.0f;
        float G = r[i] + gamma * cont * mx;
        mlp2_forward1(params, ns, h, na, act, s + i, b, q, 1, hid, zb);
        float d = q[a[i]] - G;
        float e = fabsf(d);
        float l = (e < huber_delta) ? (e * e) * 0.5f : huber_delta * (e - 0.5f * huber_delta);
        float gi = (e < huber_delta) ? d : (d > 0.0f ? huber_delta : (d < 0.0f ? -huber_delta : 0.0f));
        gi = gi / (float)b;
        if (isw) { /* PrioritizedDQN: loss = mean(w .* huber(td)); every sample's gradient scaled by its weight */
            gi *= isw[i];
            l *= isw[i];
        }
        acc += (double)l;
        for (int64_t k = 0; k < na; ++k) dout[k] = 0.0f;
        dout[a[i]] = gi;
        mlp2_backward1(params, ns, h, na, act, s + i, b, dout, 1, ga, hid, zb);
    }
    for (int64_t qq = 0; qq < np; ++qq) grad[qq] = (float)ga[qq];
    free(ga);
    free(hid);
    return (float)(acc / (double)b);
}

/* ------------------------------------------------------------------------ PPO whole iteration -- */
static int64_t env_na(int kind, const rlo_ppo_cfg* c) {
    if (kind == 0) return c->continuous ? 1 : 2;
    if (kind == 1) return c->continuous ? 1 : 3;
    return c->continuous ? 1 : 3;
}

int rlo_ppo_rollout_f32(int kind, const void* env_cfg, rlo_env_state* st, int64_t n, int64_t T,
                        const rlo_ppo_cfg* c, const float* params, uint64_t seed,
                        uint32_t env_id_base, uint32_t vec_step0, rlo_ppo_traj* tr) {
    int64_t ns = rlo_env_obs_dim(kind);
    int64_t na = env_na(kind, c);
    int64_t nout_a = ppo_actor_nout(c, na);
    int64_t np_a = net_nparams(c, ns, nout_a);
    const float* pa = params;
    const float* pc = params + np_a;
    for (int64_t t = 0; t <= T; ++t) {
        float* obs_t = tr->obs + t * ns * n;
        rlo_env_obs(kind, 0, st, n, obs_t); /* state(env) at PreActStage (post auto-reset) */
        /* env instances are independent: the loops over i run on all cores in the -fopenmp baseline build
         * (bit-identical results: nothing is reduced across instances) */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            float scr[NET_SCRATCH(256)];
            net_forward1(c, pc, ns, 1, obs_t + i, n, tr->value + t * n + i, scr);
        }
        if (t == T) break;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            float out[64];
            float scr[NET_SCRATCH(256)];
            net_forward1(c, pa, ns, nout_a, obs_t + i, n, out, scr);
            if (!c->continuous) {
                rlo_categorical_sample_f32(out, na, 1, 0, seed, env_id_base + (uint32_t)i,
                                           vec_step0 + (uint32_t)t, tr->action_i + t * n + i,
                                           tr->logp + t * n + i);
            } else {
                /* GaussianNetwork sampling  networks.jl:68-76: z = mu + sigma * randn(Float32) ;
                 * logp = sum_k normlogpdf(mu, sigma, z) */
                float lp = 0.0f;
                for (int64_t k = 0; k < na; ++k) {
                    uint32_t w[4];
                    rlo_philox4x32_10(seed, env_id_base + (uint32_t)i, (uint32_t)(k / 2),
                                      vec_step0 + (uint32_t)t, RLO_TAG_NORMAL, w);
                    float z0, z1;
                    rlo_normal_pair_f32(w[0], w[1], &z0, &z1);
                    float noise = (k & 1) ? z1 : z0;
                    float mu = out[k], sg = expf(out[na + k]);
                    float z = mu + sg * noise;
                    tr->action_f[(t * na + k) * n + i] = z;
                    lp += rlo_normlogpdf_f32(mu, sg, z);
                }
                tr->logp[t * n + i] = lp;
            }
        }
        const void* acts = c->continuous ? (const void*)(tr->action_f + t * na * n)
                                         : (const void*)(tr->action_i + t * n);
        rlo_env_step(kind, 0, env_cfg, st, n, acts, 1, seed, env_id_base, 0);
        memcpy(tr->reward + t * n, st->reward, sizeof(float) * (size_t)n);
        memcpy(tr->terminal + t * n, st->done, (size_t)n);
    }
    return 0;
}

void rlo_ppo_gae_f32(const rlo_ppo_cfg* c, int64_t n, int64_t T, rlo_ppo_traj* tr) {
    /* time-major (T, n) storage == column-major matrix (n x T): generalized_advantage_estimation(
     * rewards, values, gamma, lambda; dims = 2, terminal = terminal) */
    rlo_gae_f32(tr->adv, tr->reward, tr->value, n, T, c->gamma, c->lambda, tr->terminal, 2);
    for (int64_t q = 0; q < n * T; ++q) tr->ret[q] = tr->adv[q] + tr->value[q];
}

int rlo_ppo_update_f32(int kind, const rlo_ppo_cfg* c, int64_t n, int64_t T, rlo_ppo_traj* tr,
                       float* params, float* m, float* v, int64_t* opt_step, uint64_t seed,
                       uint32_t update_ctr, float* last_losses) {
    int64_t ns = rlo_env_obs_dim(kind);
    int64_t na = env_na(kind, c);
    int64_t np = rlo_ppo_nparams(kind, c);
    int64_t total = n * T;
    int64_t bm = total / c->n_microbatches;
    float* grad = (float*)malloc(sizeof(float) * (size_t)np);
    float* obs = (float*)malloc(sizeof(float) * (size_t)(ns * bm));
    float* lp = (float*)malloc(sizeof(float) * (size_t)bm * 3);
    float* adv = lp + bm;
    float* ret = adv + bm;
    int32_t* ai = (int32_t*)malloc(sizeof(int32_t) * (size_t)bm);
    float* af = (float*)malloc(sizeof(float) * (size_t)(bm * na));
    for (int32_t e = 0; e < c->n_epochs; ++e) {
        uint32_t epoch_ctr = update_ctr * (uint32_t)c->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < c->n_microbatches; ++mb) {
#pragma omp parallel for schedule(static)
            for (int64_t b = 0; b < bm; ++b) {
                uint32_t f = rlo_permute(seed, epoch_ctr, (uint32_t)total, (uint32_t)(mb * bm + b));
                int64_t t = f / n, i = f % n;
                for (int64_t k = 0; k < ns; ++k) obs[k * bm + b] = tr->obs[(t * ns + k) * n + i];
                lp[b] = tr->logp[f];
                adv[b] = tr->adv[f];
                ret[b] = tr->ret[f];
                if (c->continuous)
                    for (int64_t k = 0; k < na; ++k) af[k * bm + b] = tr->action_f[(t * na + k) * n + i];
                else
                    ai[b] = tr->action_i[f];
            }
            rlo_ppo_loss_grad_f32(c, ns, na, params, obs, ai, af, lp, adv, ret, bm, grad, last_losses);
            rlo_clip_by_global_norm_f32(grad, np, c->max_grad_norm);
            *opt_step += 1;
            rlo_adam_f32(params, grad, m, v, np, c->lr, c->beta1, c->beta2, c->adam_eps, *opt_step);
        }
    }
    free(grad);
    free(obs);
    free(lp);
    free(ai);
    free(af);
    return 0;
}
