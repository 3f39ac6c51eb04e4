// act_device.h -- act! + push! of one env instance as a device function, and the host-side slot claim of the push, shared by
// dqn_act.hip (its own kernels) and dqn3.hip (the 3-layer plan kernel with the act + push tail).
#pragma once
#include "env_device.h"
#include "ring_device.h"

namespace rlhip {

struct ActRing {
    void* rec;           // record ring (ring_device.h)
    int64_t state_slot;  // physical slot opened by s'
    int64_t prev_slot;   // the slot before it: its record is completed with (a, r, t, s')
};

// act!(env, a) with auto-reset + push!(trajectory, (state = s', action, reward, terminal)) of ONE env instance: the body of
// env_act_push_kernel, shared with the plan kernels that append it to their own last step (dqn3.hip: mlp3_plan32_kernel<..., ActTail>)
template <class P>
__device__ __forceinline__ void env_act_push1(const P& p, const EnvArrays<float>& st, int64_t n, int64_t env, int32_t a,
                                              uint64_t env_seed, uint32_t env_id_base, const ActRing& rb,
                                              float* __restrict__ obs_out, float* __restrict__ last_obs) {
    constexpr int NS = P::ODIM;
    const uint32_t id = env_id_base + (uint32_t)env;
    LaneState<float> e;
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][env];
    e.t = st.t[env];
    e.episode = 0;
    float r;
    bool d;
    env_step1(p, e, a, 0.0f, r, d);
    float lo[4] = {0.f, 0.f, 0.f, 0.f};
    if (last_obs) env_obs1(p, e, lo);
    if (d) {
        e.episode = st.episode[env];
        env_reset1(p, e, env_seed, id);
        st.episode[env] = e.episode;
    }
    float xn[4];
    env_obs1(p, e, xn);
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
    st.t[env] = e.t;
    st.reward[env] = r;
    st.done[env] = (uint8_t)d;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (obs_out) obs_out[(int64_t)k * n + env] = xn[k];
        if (last_obs) last_obs[(int64_t)k * n + env] = lo[k];
    }
#pragma unroll
    for (int k = NS; k < 4; ++k) xn[k] = 0.f;
    ring_push_transition(rb.rec, rb.state_slot, rb.prev_slot, n, env, xn, a, r, d ? 1u : 0u);
}

// the slots push!(trajectory, (state = s', action, reward, terminal)) writes (ring.hip); advances the ring counters
static inline ActRing claim_slots(rlhip_ring* rb) {
    ActRing ar;
    ar.rec = rb->state;
    const int64_t frames = rb->capacity;
    if (rb->len_rt < frames) rb->len_rt += 1;  // the logical action / reward / terminal traces (lengths, sum-tree keys)
    else rb->head_rt = (rb->head_rt + 1) % frames;
    const int64_t sframes = rb->capacity + 1;
    if (rb->len_sa < sframes) {
        ar.state_slot = (rb->head_sa + rb->len_sa) % sframes;
        rb->len_sa += 1;
    } else {
        ar.state_slot = rb->head_sa;
        rb->head_sa = (rb->head_sa + 1) % sframes;
    }
    ar.prev_slot = (ar.state_slot + sframes - 1) % sframes;
    return ar;
}

}  // namespace rlhip
