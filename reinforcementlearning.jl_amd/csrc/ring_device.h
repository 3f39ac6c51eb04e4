// ring_device.h -- the RECORD layout of a replay ring with small Float32 observations (round 5).
//
// A ring whose observations are Float32 with <= 4 components (the classic-control envs: everything the fused DQN learners take)
// stores ONE 64-byte record -- one cache line, one fabric request -- per (state slot, env):
//
//     record[slot * n_env + e] = { float s[4];  int32 action;  float reward;  uint32 terminal;  uint32 spare;
//                                  float s_next[4];  uint32 pad[4] }
//
// i.e. the whole transition (s, a, r, t, s') that LEAVES the state of this slot.  A sampled transition is three 16-byte loads
// out of one line; the push of `(state = s', action, reward, terminal)` (RLCore/src/policies/agent/agent_base.jl:56-59)
// completes the record of the previous slot (32 bytes: a, r, t, s') and opens the next one (its s = s').
//
// Why a whole line.  Measured on the 2^20-sample gather (profiles/r05_pmc.md): this kernel is bound by the number of 64-byte
// requests the L2 sends to the fabric (TCC_EA0_RDREQ: ~45 G requests / s, the same request rate a streaming kernel reaches with
// 128-byte requests), not by bytes and not by load instructions:
//     round 3  component-major frames + three traces          11 lines / sample   164 - 168 us
//     round 4  transition-major states + three traces          5 lines            79 us
//     round 5a 32-byte records {s', a, r, t}, s in the slot before   2 lines (1.86 requests measured)   43.5 us
//     round 5b this layout                                      1 line             26.5 us
// The price: the state is stored twice (as s of its slot and as s' of the slot before) -- 64 instead of 41 bytes per transition.
//
// Slot arithmetic is the state trace's (capacity + 1 slots, head_sa): logical transition li lives in slot
// (head_sa + li) mod (capacity + 1); the newest slot holds only a state (its transition is not complete yet).  The action /
// reward / terminal traces of RLTrajectories' `CircularArraySARTSTraces` (capacity frames, head_rt) still exist LOGICALLY --
// rlhip_ring.head_rt / len_rt keep counting them, the sum-tree keys stay `physical rt slot * n_env + e` -- only their storage
// moved into the records.  s[k >= obs_dim], s_next[k >= obs_dim], `spare` and `pad` are zero.
#pragma once
#include "common.h"

namespace rlhip {

constexpr int RING_REC_BYTES = 64;

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

// byte offset of the record of flat logical index fj = li * n_env + e.  No 64-bit division where 32 bits do (a 64-bit div / mod
// pair is ~300 VALU instructions on this chip, a 32-bit one ~40), and no modulo at all for the ring wrap: head_sa <= capacity
// and li < capacity, so head_sa + li wraps at most once.
__device__ __forceinline__ int64_t ring_record_offset(const RingRecs& rb, int64_t fj) {
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
    return (ps * rb.n_env + e) * RING_REC_BYTES;
}

union RingChunk {  // one 16-byte quarter of a record
    nt_u32x4 u;
    float f[4];
};

// the three 16-byte loads of one sampled transition (one cache line), all issued before the first use
__device__ __forceinline__ RingTransition ring_load_transition(const RingRecs& rb, int64_t fj) {
    const uint8_t* r0 = rb.rec + ring_record_offset(rb, fj);
    RingChunk s, w, sn;
    s.u = *reinterpret_cast<const nt_u32x4*>(r0);
    w.u = *reinterpret_cast<const nt_u32x4*>(r0 + 16);
    sn.u = *reinterpret_cast<const nt_u32x4*>(r0 + 32);
    RingTransition t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t.s[k] = s.f[k], t.sn[k] = sn.f[k];
    t.a = (int32_t)w.u[0];
    t.r = w.f[1];
    t.t = (w.u[2] & 0xffu) ? 1u : 0u;
    return t;
}

// push!(trajectory, (state = s', action, reward, terminal)) for one env: completes the record of the previous slot (a, r, t, s':
// 32 bytes) and opens the record of the new slot (s = s', the rest zero: 64 bytes).  x[k >= OD] must be 0.
__device__ __forceinline__ void ring_push_transition(void* rec, int64_t slot_new, int64_t slot_prev, int64_t n_env, int64_t e,
                                                     const float x[4], int32_t a, float r, uint32_t t) {
    const nt_u32x4 xs = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    const nt_u32x4 z = {0u, 0u, 0u, 0u};
    uint8_t* pp = (uint8_t*)rec + (slot_prev * n_env + e) * RING_REC_BYTES;
    *reinterpret_cast<nt_u32x4*>(pp + 16) = nt_u32x4{(uint32_t)a, __float_as_uint(r), t ? 1u : 0u, 0u};
    *reinterpret_cast<nt_u32x4*>(pp + 32) = xs;
    uint8_t* pn = (uint8_t*)rec + (slot_new * n_env + e) * RING_REC_BYTES;
    *reinterpret_cast<nt_u32x4*>(pn) = xs;
    *reinterpret_cast<nt_u32x4*>(pn + 16) = z;
    *reinterpret_cast<nt_u32x4*>(pn + 32) = z;
    *reinterpret_cast<nt_u32x4*>(pn + 48) = z;
}

// push!(trajectory, (state = s,)) at PreEpisodeStage: opens a record, no transition is completed
__device__ __forceinline__ void ring_push_state(void* rec, int64_t slot, int64_t n_env, int64_t e, const float x[4]) {
    const nt_u32x4 z = {0u, 0u, 0u, 0u};
    uint8_t* p = (uint8_t*)rec + (slot * n_env + e) * RING_REC_BYTES;
    *reinterpret_cast<nt_u32x4*>(p) = nt_u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    *reinterpret_cast<nt_u32x4*>(p + 16) = z;
    *reinterpret_cast<nt_u32x4*>(p + 32) = z;
    *reinterpret_cast<nt_u32x4*>(p + 48) = z;
}

}  // namespace rlhip
