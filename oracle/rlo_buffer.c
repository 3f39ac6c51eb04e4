/* rlo_buffer.c -- CircularArraySARTSTraces-style replay ring + BatchSampler index draw.
 * TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 *
 * The arithmetic lives in un-vendored packages (ReinforcementLearningTrajectories compat "0.4",
 * CircularArrayBuffers compat "0.1.12": RLCore/Project.toml:9,20,30,40) -- PARITY UNPINNED.
 * Anchors in the reference tree: constructor form RLCore/test/policies/q_based_policy.jl:41-47,
 * push protocol RLCore/policies/agent/agent_base.jl:45-59, trace order agent_srt_cache.jl:30-40,
 * length semantics RLCore/test/policies/agent.jl:27-34 (0 after the first state, 1 after the
 * first transition), multiplexed next_state = state[i+1] (q_based_policy.jl:62-94).
 *
 * Published algorithm restated: each trace is a ring over its last dimension; `push!` writes the
 * frame after the newest, overwriting the oldest when full; logical index i (0 = oldest) lives at
 * physical (head + i) mod frames.  The state trace has capacity+1 frames, action/reward/terminal
 * have capacity frames; transition i = (state[i], action[i], reward[i], terminal[i], state[i+1]).
 * Vector-env extension: one frame = one vec-step of n_env transitions, stored SoA.
 */
#include "rl_oracle.h"
#include <math.h>
#include <string.h>

void rlo_ring_init(rlo_ring* rb, int64_t capacity, int64_t n_env, int64_t obs_dim, float* state,
                   int32_t* action, float* reward, uint8_t* terminal) {
    rb->capacity = capacity;
    rb->n_env = n_env;
    rb->obs_dim = obs_dim;
    rb->head_sa = rb->len_sa = rb->head_rt = rb->len_rt = 0;
    rb->state = state;
    rb->action = action;
    rb->reward = reward;
    rb->terminal = terminal;
}

static void push_state_frame(rlo_ring* rb, const float* obs) {
    int64_t frames = rb->capacity + 1;
    int64_t fsz = rb->obs_dim * rb->n_env;
    int64_t phys;
    if (rb->len_sa < frames) {
        phys = (rb->head_sa + rb->len_sa) % frames;
        rb->len_sa += 1;
    } else {
        phys = rb->head_sa; /* overwrite the oldest, then it becomes the newest */
        rb->head_sa = (rb->head_sa + 1) % frames;
    }
    memcpy(rb->state + phys * fsz, obs, sizeof(float) * (size_t)fsz);
}

void rlo_ring_push_state(rlo_ring* rb, const float* obs) { push_state_frame(rb, obs); }

void rlo_ring_push_transition(rlo_ring* rb, const float* next_obs, const int32_t* action,
                              const float* reward, const uint8_t* terminal) {
    int64_t frames = rb->capacity;
    int64_t n = rb->n_env;
    int64_t phys;
    if (rb->len_rt < frames) {
        phys = (rb->head_rt + rb->len_rt) % frames;
        rb->len_rt += 1;
    } else {
        phys = rb->head_rt;
        rb->head_rt = (rb->head_rt + 1) % frames;
    }
    memcpy(rb->action + phys * n, action, sizeof(int32_t) * (size_t)n);
    memcpy(rb->reward + phys * n, reward, sizeof(float) * (size_t)n);
    memcpy(rb->terminal + phys * n, terminal, (size_t)n);
    push_state_frame(rb, next_obs);
}

int64_t rlo_ring_length(const rlo_ring* rb) { return rb->len_rt; }

/* BatchSampler: inds = rand(rng, 1:length(traces), batchsize) (with replacement).  Stand-in draw:
 * 64-bit word (w0:w1) of Philox(seed, idx = b, t = draw_ctr, SAMPLER) scaled by multiply-high. */
void rlo_ring_sample_indices(const rlo_ring* rb, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                             int64_t* flat_idx) {
    uint64_t total = (uint64_t)rb->len_rt * (uint64_t)rb->n_env;
    for (int64_t b = 0; b < batch; ++b) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)b, 0, draw_ctr, RLO_TAG_SAMPLER, w);
        uint64_t x = ((uint64_t)w[0] << 32) | (uint64_t)w[1];
        flat_idx[b] = (int64_t)(((unsigned __int128)x * (unsigned __int128)total) >> 64);
    }
}

void rlo_ring_gather(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, float* s,
                     int32_t* a, float* r, uint8_t* term, float* s_next) {
    int64_t n = rb->n_env, d = rb->obs_dim;
    for (int64_t b = 0; b < batch; ++b) {
        int64_t li = flat_idx[b] / n, e = flat_idx[b] % n;
        int64_t ps = (rb->head_sa + li) % (rb->capacity + 1);
        int64_t pn = (rb->head_sa + li + 1) % (rb->capacity + 1);
        int64_t pt = (rb->head_rt + li) % rb->capacity;
        for (int64_t k = 0; k < d; ++k) {
            s[k * batch + b] = rb->state[(ps * d + k) * n + e];
            s_next[k * batch + b] = rb->state[(pn * d + k) * n + e];
        }
        a[b] = rb->action[pt * n + e];
        r[b] = rb->reward[pt * n + e];
        term[b] = rb->terminal[pt * n + e];
    }
}

/* ------------------------------------------------------------------ n-step transitions --
 * NStepBatchSampler(n, gamma, batchsize) of RLTrajectories 0.4 (un-vendored; PARITY UNPINNED -- spec: SURVEY.md row L2
 * `R = r + gamma^n (1 - t) max Qt(s')`, blog an_introduction_.../index.md:320-331).  Published algorithm restated:
 *   valid start indices are those with n transitions ahead of them: inds = rand(rng, 1:(length - n + 1), batchsize);
 *   state, action        of the start step;
 *   the window           steps i .. i + ns - 1, ns = n unless a terminal flag cuts it short (its step is the last one);
 *   reward               foldr((x, y) -> x + gamma * y, rewards[window])  = discount_rewards_reduced over the window
 *                        (RLCore/src/utils/basic.jl:237-319: gain = r[i] + gamma * gain from the window's end, Float32);
 *   terminal             any(terminal[window]);
 *   next_state           state[i + ns].
 * Vector-env ring: start li in [0, len_rt - n], env e; the draw is the uniform sampler's over (len_rt - n + 1) * n_env. */
void rlo_ring_sample_indices_nstep(const rlo_ring* rb, int64_t batch, int64_t n_step, uint64_t seed, uint32_t draw_ctr,
                                   int64_t* flat_idx) {
    uint64_t total = (uint64_t)(rb->len_rt - n_step + 1) * (uint64_t)rb->n_env;
    for (int64_t b = 0; b < batch; ++b) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)b, 0, draw_ctr, RLO_TAG_SAMPLER, w);
        uint64_t x = ((uint64_t)w[0] << 32) | (uint64_t)w[1];
        flat_idx[b] = (int64_t)(((unsigned __int128)x * (unsigned __int128)total) >> 64);
    }
}

void rlo_ring_gather_nstep(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, int64_t n_step, float gamma, float* s,
                           int32_t* a, float* r, uint8_t* term, float* s_next) {
    int64_t n = rb->n_env, d = rb->obs_dim;
    for (int64_t b = 0; b < batch; ++b) {
        int64_t li = flat_idx[b] / n, e = flat_idx[b] % n;
        int64_t ns = 0;
        uint8_t t = 0;
        while (ns < n_step && li + ns < rb->len_rt) { /* (a window never runs past the newest stored transition) */
            t = rb->terminal[((rb->head_rt + li + ns) % rb->capacity) * n + e];
            ns += 1;
            if (t) break;
        }
        float gain = 0.0f;
        for (int64_t k = ns - 1; k >= 0; --k) gain = rb->reward[((rb->head_rt + li + k) % rb->capacity) * n + e] + gamma * gain;
        int64_t ps = (rb->head_sa + li) % (rb->capacity + 1), pn = (rb->head_sa + li + ns) % (rb->capacity + 1);
        for (int64_t k = 0; k < d; ++k) {
            s[k * batch + b] = rb->state[(ps * d + k) * n + e];
            s_next[k * batch + b] = rb->state[(pn * d + k) * n + e];
        }
        a[b] = rb->action[((rb->head_rt + li) % rb->capacity) * n + e];
        r[b] = gain;
        term[b] = t;
    }
}

/* gamma^n of the n-step target: Julia's `gamma^n` for Float32 gamma and Int n is evaluated through Float64 and rounded once */
float rlo_gamma_pow(float gamma, int64_t n) { return (float)pow((double)gamma, (double)n); }

/* ---------------------------------------------------------------- priority sum-tree --
 * Published algorithm of CircularArrayBuffers.SumTree (un-vendored, compat "0.1.12", RLCore/Project.toml:30):
 *   tree = zeros(nparents + capacity), nparents = 2^ceil(log2(capacity)) - 1; leaf i at nparents + i;
 *   setindex!: write the leaf, add the change to every ancestor;   get(t, v): walk down from the root,
 *   `v <= tree[left] ? left : (v -= tree[left]; right)`;   rand(rng, t) = get(t, rand(rng, Float32) * tree[1]).
 * and of the RLTrajectories 0.4 prioritized sampler: `inds, priorities = rand(rng, sumtree, batchsize)`.
 * Restated with the two documented differences of rl_oracle.h (recomputed parents, robust descent). */
static int64_t st_pow2(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

int64_t rlo_sumtree_nodes(int64_t n_leaves) { return n_leaves < 1 ? 0 : 2 * st_pow2(n_leaves); }

static void st_fix_up(float* tree, int64_t node) {
    for (node >>= 1; node >= 1; node >>= 1) tree[node] = tree[2 * node] + tree[2 * node + 1];
}

void rlo_sumtree_fill_range(float* tree, int64_t n_leaves, int64_t start, int64_t count, float value) {
    int64_t P = st_pow2(n_leaves);
    for (int64_t i = 0; i < count; ++i) tree[P + start + i] = value;
    if (count <= 0) return;
    for (int64_t lo = (P + start) >> 1, hi = (P + start + count - 1) >> 1; lo >= 1; lo >>= 1, hi >>= 1)
        for (int64_t node = lo; node <= hi; ++node) tree[node] = tree[2 * node] + tree[2 * node + 1];
}

void rlo_sumtree_update(float* tree, int64_t n_leaves, const int64_t* leaf, const float* prio, int64_t n) {
    int64_t P = st_pow2(n_leaves);
    for (int64_t i = 0; i < n; ++i) { /* in order: the last occurrence of a key wins */
        if (leaf[i] < 0 || leaf[i] >= n_leaves) continue;
        tree[P + leaf[i]] = prio[i];
        st_fix_up(tree, P + leaf[i]);
    }
}

static int64_t st_descend(const float* tree, int64_t P, float v) {
    int64_t node = 1;
    while (node < P) {
        float l = tree[2 * node], r = tree[2 * node + 1];
        int right = (v > l && r > 0.0f) || l == 0.0f;
        if (right) v = v - l;
        node = 2 * node + right;
    }
    return node - P;
}

void rlo_sumtree_sample(const float* tree, int64_t n_leaves, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                        int64_t* leaf_out, float* prio_out) {
    int64_t P = st_pow2(n_leaves);
    for (int64_t b = 0; b < batch; ++b) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)b, 0, draw_ctr, RLO_TAG_SAMPLER, w);
        float v = rlo_u01_f32(w[2]) * tree[1];
        int64_t k = st_descend(tree, P, v);
        if (k >= n_leaves) k = n_leaves - 1;
        leaf_out[b] = k;
        if (prio_out) prio_out[b] = tree[P + k];
    }
}

void rlo_ring_push_priority(const rlo_ring* rb, float* tree, float priority) {
    int64_t newest = (rb->head_rt + rb->len_rt - 1) % rb->capacity;
    rlo_sumtree_fill_range(tree, rb->capacity * rb->n_env, newest * rb->n_env, rb->n_env, priority);
}

void rlo_ring_sample_prioritized(const rlo_ring* rb, const float* tree, int64_t batch, uint64_t seed,
                                 uint32_t draw_ctr, int64_t* flat_idx, int64_t* key_out, float* prio_out) {
    int64_t n_leaves = rb->capacity * rb->n_env;
    int64_t* keys = key_out ? key_out : flat_idx;
    rlo_sumtree_sample(tree, n_leaves, batch, seed, draw_ctr, keys, prio_out);
    for (int64_t b = 0; b < batch; ++b) {
        int64_t pt = keys[b] / rb->n_env, e = keys[b] % rb->n_env;
        int64_t li = pt - rb->head_rt;
        if (li < 0) li += rb->capacity;
        int64_t k = keys[b];
        flat_idx[b] = li * rb->n_env + e;
        if (key_out) key_out[b] = k;
    }
}

/* PrioritizedDQN write-back value (removed Zoo learner): p = (|td| + eps)^alpha, power in double, rounded once */
void rlo_per_priority_f32(const float* td, int64_t n, float eps, float alpha, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        float x = fabsf(td[i]) + eps;
        out[i] = (alpha == 1.0f) ? x : (float)pow((double)x, (double)alpha);
    }
}

void rlo_per_is_weights_f32(const float* prio, int64_t n, float beta, float* out) {
    float mx = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        out[i] = (float)(1.0 / pow((double)(prio[i] + 1e-10f), (double)beta));
        if (out[i] > mx) mx = out[i];
    }
    for (int64_t i = 0; i < n; ++i) out[i] = out[i] / mx;
}

/* ------------------------------------------------------------- stack-at-sample gather --
 * StackFrames (RLCore/src/utils/stack_frames.jl:11-44) keeps the latest n frames in a CircularArrayBuffer that
 * starts zero-filled (:22-26) and is zero-filled again by reset! (:33-36); the newest frame is the last slice
 * (test RLCore/test/utils/stack_frames.jl:7-9).  With single frames in the ring (n_env == 1) the same stacks are
 * rebuilt at sample time: going back from the newest frame, a frame is part of the stack until an episode
 * boundary (terminal flag of the transition being crossed) or the oldest stored frame is passed; the rest is 0.
 * s, s_next: (batch, n_stack, obs_dim), oldest frame first. */
void rlo_ring_gather_stacked(const rlo_ring* rb, const int64_t* flat_idx, int64_t batch, int64_t n_stack, float* s,
                             int32_t* a, float* r, uint8_t* term, float* s_next) {
    int64_t d = rb->obs_dim;
    for (int64_t b = 0; b < batch; ++b) {
        int64_t li = flat_idx[b];
        int64_t pt = (rb->head_rt + li) % rb->capacity;
        a[b] = rb->action[pt];
        r[b] = rb->reward[pt];
        term[b] = rb->terminal[pt];
        for (int which = 0; which < 2; ++which) { /* 0: state stack ends at frame li, 1: next stack at li + 1 */
            float* out = which ? s_next : s;
            int64_t newest = li + which;
            int ok = 1;
            for (int64_t k = 0; k < n_stack; ++k) {
                int64_t f = newest - k;
                if (k >= 1) { /* stepping back from frame f + 1 to f crosses transition f */
                    if (f < 0) ok = 0;
                    else if (rb->terminal[(rb->head_rt + f) % rb->capacity]) ok = 0;
                }
                float* dst = out + (b * n_stack + (n_stack - 1 - k)) * d;
                if (ok) {
                    const float* src = rb->state + ((rb->head_sa + f) % (rb->capacity + 1)) * d;
                    memcpy(dst, src, sizeof(float) * (size_t)d);
                } else {
                    memset(dst, 0, sizeof(float) * (size_t)d);
                }
            }
        }
    }
}
