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

// byte offsets of record(s) and record(s') of flat logical index fj = li * n_env + e.  No 64-bit division where 32 bits do (a
// 64-bit div / mod pair is ~300 VALU instructions on this chip, a 32-bit one ~40), and no modulo at all for the ring wrap:
// head_sa <= capacity and li < capacity, so head_sa + li wraps at most once.
__device__ __forceinline__ void ring_record_offsets(const RingRecs& rb, int64_t fj, int64_t& off_s, int64_t& off_n) {
    int64_t li, e;
    if (((uint64_t)fj | (uint64_t)rb.n_env) >> 32) {
        li = fj / rb.n_env;
        e = fj - li * rb.n_env;
    } else {
        const uint32_t q = (uint32_t)fj / (uint32_t)rb.n_env;
        li = q;
        e = (uint32_t)fj - q * (uint32_t)rb.n_env;
    }
    int64_t ps = rb.head_sa + li;
    if (ps > rb.capacity) ps -= rb.capacity + 1;
    const int64_t pn = (ps == rb.capacity) ? 0 : ps + 1;
    off_s = (ps * rb.n_env + e) * RING_REC_BYTES;
    off_n = (pn * rb.n_env + e) * RING_REC_BYTES;
}

union RingChunk {  // one 16-byte half of a record
    nt_u32x4 u;
    float f[4];
};

// the three 16-byte loads of one sampled transition, all issued before the first use
template <bool NT = false>
__device__ __forceinline__ RingTransition ring_load_transition(const RingRecs& rb, int64_t fj) {
    int64_t o0, o1;
    ring_record_offsets(rb, fj, o0, o1);
    const uint8_t* r0 = rb.rec + o0;
    const uint8_t* r1 = rb.rec + o1;
    RingChunk s, sn, w;
    if (NT) {
        s.u = nt_load16(r0);
        sn.u = nt_load16(r1);
        w.u = nt_load16(r1 + 16);
    } else {
        s.u = *reinterpret_cast<const nt_u32x4*>(r0);
        sn.u = *reinterpret_cast<const nt_u32x4*>(r1);
        w.u = *reinterpret_cast<const nt_u32x4*>(r1 + 16);
    }
    RingTransition t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.s[k] = s.f[k], t.sn[k] = sn.f[k];
    t.a = (int32_t)w.u[0];
    t.r = w.f[1];
    t.t = (w.u[2] & 0xffu) ? 1u : 0u;
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
