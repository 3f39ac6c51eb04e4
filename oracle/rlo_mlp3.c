/* rlo_mlp3.c -- three-layer Q-network ns -> h -> h -> na with a bf16 hidden x hidden layer, and the DQN
 * loss + gradient over it.  TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 *
 * Network: the blog's DQN model `Chain(Dense(ns, 128, relu), Dense(128, 128, relu), Dense(128, na))`
 * (docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15126-15128), run through
 * `forward(learner, x) = model(x)` (RLCore/src/policies/learners/flux_approximator.jl:43); parameters flat in
 * Flux.destructure order W1 (h x ns, col-major) | b1 | W2 (h x h) | b2 | W3 (na x h) | b3.
 * Loss: the removed Zoo DQN learner, same as rlo_dqn_loss_grad_f32 (rlo_learn.c).
 *
 * Mixed precision restated here (BASELINE config "MLP in bf16 MFMA", f32 accumulate, f32 master weights):
 *   layer 1 and the head are f32 exactly as in mlp2 (fmaf chains);
 *   layer 2 multiplies bf16(W2) by bf16(h1) (round-to-nearest-even) and accumulates in f32 -- here in double,
 *   rounded once, which bounds the difference to the MFMA's internal summation order (tests state the tolerance);
 *   backward: dz2 is rounded to bf16 for the two GEMMs dW2 = dz2^T h1 and dh1 = dz2 W2; db2 uses the f32 dz2.
 */
#include "rl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static float bf16r(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) {
        u = (u | 0x00400000u) & 0xFFFF0000u;
    } else {
        u += 0x7FFFu + ((u >> 16) & 1u);
        u &= 0xFFFF0000u;
    }
    memcpy(&f, &u, 4);
    return f;
}

float rlo_bf16_round_f32(float f) { return bf16r(f); }

static float a_fwd(int act, float z) { return act == 0 ? (z > 0.0f ? z : 0.0f) : tanhf(z); }
static float a_bwd(int act, float z, float h) { return act == 0 ? (z > 0.0f ? 1.0f : 0.0f) : 1.0f - h * h; }

int64_t rlo_mlp3_nparams(int64_t ns, int64_t h, int64_t na) { return h * ns + h + h * h + h + na * h + na; }

/* glorot_uniform stand-in as rlo_mlp2_init_f32; tensor ids net_id * 4 + {0: W1, 1: W2, 2: W3}; zero biases */
void rlo_mlp3_init_f32(float* p, int64_t ns, int64_t h, int64_t na, uint64_t seed, uint32_t net_id) {
    int64_t sizes[3] = {h * ns, h * h, na * h};
    int64_t fan[3][2] = {{ns, h}, {h, h}, {h, na}};
    int64_t bias[3] = {h, h, na};
    for (int t = 0; t < 3; ++t) {
        float s = sqrtf(6.0f / (float)(fan[t][0] + fan[t][1]));
        for (int64_t q = 0; q < sizes[t]; ++q) {
            uint32_t w[4];
            rlo_philox4x32_10(seed, (uint32_t)q, 0, net_id * 4u + (uint32_t)t, RLO_TAG_INIT, w);
            p[q] = (2.0f * rlo_u01_f32(w[0]) - 1.0f) * s;
        }
        p += sizes[t];
        for (int64_t q = 0; q < bias[t]; ++q) p[q] = 0.0f;
        p += bias[t];
    }
}

typedef struct {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
} mlp3_view;

static mlp3_view view3(const float* p, int64_t ns, int64_t h, int64_t na) {
    mlp3_view v;
    v.W1 = p;
    v.b1 = v.W1 + h * ns;
    v.W2 = v.b1 + h;
    v.b2 = v.W2 + h * h;
    v.W3 = v.b2 + h;
    v.b3 = v.W3 + na * h;
    return v;
}

/* caches (each h floats, may be NULL together): z1, h1 (f32), z2, h2 */
void rlo_mlp3_forward1(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x,
                          int64_t xstride, float* out, int64_t ostride, float* z1, float* h1, float* z2, float* h2) {
    mlp3_view v = view3(p, ns, h, na);
    for (int64_t j = 0; j < h; ++j) {
        float z = v.b1[j];
        for (int64_t k = 0; k < ns; ++k) z = fmaf(v.W1[j + h * k], x[k * xstride], z);
        z1[j] = z;
        h1[j] = a_fwd(act, z);
    }
    for (int64_t j = 0; j < h; ++j) {
        double acc = 0.0;
        for (int64_t k = 0; k < h; ++k) acc += (double)bf16r(v.W2[j + h * k]) * (double)bf16r(h1[k]);
        z2[j] = (float)acc + v.b2[j];
        h2[j] = a_fwd(act, z2[j]);
    }
    for (int64_t o = 0; o < na; ++o) {
        double acc = 0.0;
        for (int64_t j = 0; j < h; ++j) acc += (double)v.W3[o + na * j] * (double)h2[j];
        out[o * ostride] = (float)acc + v.b3[o];
    }
}

void rlo_mlp3_forward_f32(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x,
                          int64_t batch, float* out) {
    float* buf = (float*)malloc(sizeof(float) * (size_t)h * 4);
    for (int64_t i = 0; i < batch; ++i)
        rlo_mlp3_forward1(p, ns, h, na, act, x + i, batch, out + i, batch, buf, buf + h, buf + 2 * h, buf + 3 * h);
    free(buf);
}

void rlo_mlp3_backward1(const float* p, int64_t ns, int64_t h, int64_t na, int act, const float* x,
                           int64_t xstride, const float* dout, double* ga, const float* z1, const float* h1,
                           const float* z2, const float* h2, float* dz2b, double* dh1) {
    mlp3_view v = view3(p, ns, h, na);
    double* gW1 = ga;
    double* gb1 = gW1 + h * ns;
    double* gW2 = gb1 + h;
    double* gb2 = gW2 + h * h;
    double* gW3 = gb2 + h;
    double* gb3 = gW3 + na * h;
    for (int64_t o = 0; o < na; ++o) gb3[o] += (double)dout[o];
    for (int64_t j = 0; j < h; ++j) {
        float dh = 0.0f;
        for (int64_t o = 0; o < na; ++o) {
            gW3[o + na * j] += (double)dout[o] * (double)h2[j];
            dh = fmaf(dout[o], v.W3[o + na * j], dh);
        }
        float dz = dh * a_bwd(act, z2[j], h2[j]);
        gb2[j] += (double)dz;
        dz2b[j] = bf16r(dz);
    }
    for (int64_t k = 0; k < h; ++k) dh1[k] = 0.0;
    for (int64_t k = 0; k < h; ++k) {
        float hb = bf16r(h1[k]);
        for (int64_t j = 0; j < h; ++j) {
            gW2[j + h * k] += (double)dz2b[j] * (double)hb;
            dh1[k] += (double)dz2b[j] * (double)bf16r(v.W2[j + h * k]);
        }
    }
    for (int64_t k = 0; k < h; ++k) {
        float dz = (float)dh1[k] * a_bwd(act, z1[k], h1[k]);
        gb1[k] += (double)dz;
        for (int64_t i = 0; i < ns; ++i) gW1[k + h * i] += (double)dz * (double)x[i * xstride];
    }
}

/* one sample of the DQN loss: adds its gradient into ga (Float64), returns its (weighted) Huber loss */
static double dqn3_sample(int64_t ns, int64_t h, int64_t na, int act, const float* params, const float* target_params,
                          const float* s, const int32_t* a, const float* r, const uint8_t* term, const float* s_next,
                          int64_t b, int64_t i, float gamma, float huber_delta, double* ga, float* q_out, const float* isw,
                          float* buf, double* dh1) {
    float q[64], qn[64], dout[64];
    rlo_mlp3_forward1(target_params, ns, h, na, act, s_next + i, b, qn, 1, buf, buf + h, buf + 2 * h, buf + 3 * h);
    float mx = qn[0];
    for (int64_t k = 1; k < na; ++k)
        if (qn[k] > mx) mx = qn[k];
    float cont = term[i] ? 0.0f : 1.0f;
    float G = r[i] + gamma * cont * mx;
    rlo_mlp3_forward1(params, ns, h, na, act, s + i, b, q, 1, buf, buf + h, buf + 2 * h, buf + 3 * h);
    if (q_out)
        for (int64_t k = 0; k < na; ++k) q_out[k * b + i] = q[k];
    float d = q[a[i]] - G;
    float e = fabsf(d);
    float l = (e < huber_delta) ? (e * e) * 0.5f : huber_delta * (e - 0.5f * huber_delta);
    float gi = (e < huber_delta) ? d : (d > 0.0f ? huber_delta : (d < 0.0f ? -huber_delta : 0.0f));
    gi = gi / (float)b;
    if (isw) { /* importance-sampling weights of prioritized replay: mean(w .* huber(td)) */
        gi *= isw[i];
        l *= isw[i];
    }
    for (int64_t k = 0; k < na; ++k) dout[k] = 0.0f;
    dout[a[i]] = gi;
    rlo_mlp3_backward1(params, ns, h, na, act, s + i, b, dout, ga, buf, buf + h, buf + 2 * h, buf + 3 * h, buf + 4 * h, dh1);
    return (double)l;
}

/* y = r + gamma (1 - t) max_a' Qt(s', a'); Huber(delta) on Q(s, a) - y, mean over the batch.  Returns the
 * loss; grad (nparams floats) is overwritten.  q_out (na x b) optional: the online Q(s, .) values.
 * Samples are independent: with OpenMP (the all-cores build, oracle.use_all_cores) each thread owns a contiguous
 * chunk and its own Float64 accumulator, combined in thread order (as rlo_ppo_loss_grad_f32 in rlo_learn.c);
 * without OpenMP this is the single sequential pass it always was. */
float rlo_dqn3_loss_grad_f32(int64_t ns, int64_t h, int64_t na, int act, const float* params,
                             const float* target_params, const float* s, const int32_t* a, const float* r,
                             const uint8_t* term, const float* s_next, int64_t b, float gamma, float huber_delta,
                             float* grad, float* q_out, const float* isw) {
    int64_t np = rlo_mlp3_nparams(ns, h, na);
    double* ga = (double*)calloc((size_t)np, sizeof(double));
    double acc = 0;
#ifdef _OPENMP
    int nthr = omp_get_max_threads();
    if (nthr > 1 && b >= 4 * (int64_t)nthr) {
        double* gt = (double*)calloc((size_t)np * (size_t)nthr, sizeof(double));
        double* accs = (double*)calloc((size_t)nthr, sizeof(double));
#pragma omp parallel num_threads(nthr)
        {
            int tid = omp_get_thread_num();
            int64_t i0 = b * tid / nthr, i1 = b * (tid + 1) / nthr;
            float* buf_t = (float*)malloc(sizeof(float) * (size_t)h * 5);
            double* dh1_t = (double*)malloc(sizeof(double) * (size_t)h);
            double la = 0;
            for (int64_t i = i0; i < i1; ++i)
                la += dqn3_sample(ns, h, na, act, params, target_params, s, a, r, term, s_next, b, i, gamma, huber_delta,
                                  gt + (size_t)np * tid, q_out, isw, buf_t, dh1_t);
            accs[tid] = la;
            free(buf_t);
            free(dh1_t);
        }
        for (int t = 0; t < nthr; ++t) {
            for (int64_t qq = 0; qq < np; ++qq) ga[qq] += gt[(size_t)np * t + qq];
            acc += accs[t];
        }
        free(gt);
        free(accs);
    } else
#endif
    {
        float* buf = (float*)malloc(sizeof(float) * (size_t)h * 5);
        double* dh1 = (double*)malloc(sizeof(double) * (size_t)h);
        for (int64_t i = 0; i < b; ++i)
            acc += dqn3_sample(ns, h, na, act, params, target_params, s, a, r, term, s_next, b, i, gamma, huber_delta, ga,
                               q_out, isw, buf, dh1);
        free(buf);
        free(dh1);
    }
    for (int64_t qq = 0; qq < np; ++qq) grad[qq] = (float)ga[qq];
    free(ga);
    return (float)(acc / (double)b);
}
