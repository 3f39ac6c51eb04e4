// ring_device.h -- the RECORD layout of a replay ring with small Float32 observations (round 5).
//
// A ring whose observations are Float32 with <= 4 components (the classic-control envs: everything the fused DQN learners take)
// stores ONE 32-byte record per (state slot, env):
//
//     record[slot * n_env + e] = { float s[4];  int32 action;  float reward;  uint32 terminal;  uint32 spare }
//
// where (action, reward, terminal) belong to the transition that ARRIVED at this state -- i.e. the record is exactly the named
// tuple the agent pushes at PostActStage, `push!(trajectory, (state = s', action = a, reward = r, terminal = t))`
// (RLCore/src/policies/agent/agent_base.jl:56-59), so a push is one full-record write and a sampled transition
// (s, a, r, t, s') is TWO 32-byte sectors: the state half of record(slot of s) and the whole record(slot of s').
// The layouts before it cost 11 (round 3: component-major frames) and 5 (round 4: transition-major states + three traces)
// 64-byte lines per 82-byte CartPole sample (profiles/r04_pmc.md: 3.55 x the algorithmic bytes).
//
// Slot arithmetic is the state trace's (capacity + 1 slots, head_sa): logical transition li has
//     s  in slot ps = (head_sa + li)     mod (capacity + 1)
//     s', a, r, t in slot pn = (head_sa + li + 1) mod (capacity + 1)
// The action / reward / terminal traces of RLTrajectories' `CircularArraySARTSTraces` (capacity frames, head_rt) still exist
// LOGICALLY -- rlhip_ring.head_rt / len_rt keep counting them, the sum-tree keys stay `physical rt slot * n_env + e` -- only their
// storage moved into the records.  s[k >= obs_dim] and `spare` are zero.
#pragma once
#include "common.h"

namespace rlhip {

constexpr int RING_REC_BYTES = 32;

__host__ __device__ inline bool ring_records(int64_t obs_dim, int32_t elem_bytes) { return elem_bytes == 4 && obs_dim <= 4; }

struct RingRecs {  // device view of a record ring
    const uint8_t* rec;
    int64_t capacity, n_env, head_sa;
};

struct RingTransition {
    float s[4], sn[4];
    int32_t a;
    float r;
    uint32_t t;  // 0 / 1
};

// the three 16-byte loads of one sampled transition, all issued before the first use
__device__ __forceinline__ RingTransition ring_load_transition(const RingRecs& rb, int64_t fj) {
    const int64_t li = fj / rb.n_env, e = fj - li * rb.n_env;
    const int64_t ps = (rb.head_sa + li) % (rb.capacity + 1);
    const int64_t pn = (ps == rb.capacity) ? 0 : ps + 1;
    const uint8_t* r0 = rb.rec + (ps * rb.n_env + e) * RING_REC_BYTES;
    const uint8_t* r1 = rb.rec + (pn * rb.n_env + e) * RING_REC_BYTES;
    const float4 s = *reinterpret_cast<const float4*>(r0);
    const float4 sn = *reinterpret_cast<const float4*>(r1);
    const int4 w = *reinterpret_cast<const int4*>(r1 + 16);
    RingTransition t;
    t.s[0] = s.x, t.s[1] = s.y, t.s[2] = s.z, t.s[3] = s.w;
    t.sn[0] = sn.x, t.sn[1] = sn.y, t.sn[2] = sn.z, t.sn[3] = sn.w;
    t.a = w.x;
    t.r = __int_as_float(w.y);
    t.t = ((uint32_t)w.z & 0xffu) ? 1u : 0u;
    return t;
}

// one whole record (the pushed tuple); x[k >= OD] must be 0
__device__ __forceinline__ void ring_store_record(void* rec, int64_t slot, int64_t n_env, int64_t e, const float x[4], int32_t a,
                                                  float r, uint32_t t) {
    uint8_t* p = (uint8_t*)rec + (slot * n_env + e) * RING_REC_BYTES;
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<int4*>(p + 16) = make_int4(a, __float_as_int(r), (int)(t ? 1u : 0u), 0);
}

// the state half only: push!(trajectory, (state = s,)) at PreEpisodeStage -- no transition arrives at this state
__device__ __forceinline__ void ring_store_state_only(void* rec, int64_t slot, int64_t n_env, int64_t e, const float x[4]) {
    uint8_t* p = (uint8_t*)rec + (slot * n_env + e) * RING_REC_BYTES;
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<int4*>(p + 16) = make_int4(0, 0, 0, 0);
}

}  // namespace rlhip
