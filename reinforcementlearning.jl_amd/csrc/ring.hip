// ring.hip -- CircularArraySARTSTraces-style replay ring resident in HBM: batched push + gather.
//
// Replaces (un-vendored ReinforcementLearningTrajectories 0.4 / CircularArrayBuffers 0.1.12; call sites
// RLCore/policies/agent/agent_base.jl:45-59, docs/src/How_to_implement_a_new_algorithm.md:70-74,90-112):
//   push!(trajectory, (state = s,))                         -> rlhip_ring_push_state
//   push!(trajectory, (state = s', action, reward, terminal)) -> rlhip_ring_push_transition
//   BatchSampler: inds = rand(rng, 1:length, batchsize)      -> rlhip_ring_sample_indices
//   traces[inds] (`for batch in trajectory`)                 -> rlhip_ring_gather
// In the reference the trajectory always lives in host RAM and the gather is a per-sample strided
// memcpy followed by an H2D copy of the batch; here the ring never leaves HBM.
//
// Layout: one frame per slot.  Float32 observations with <= 4 components: one 64-byte RECORD per (state slot, env) holding the
// whole transition (s, a, r, t, s') that leaves that state -- ring_device.h; everything else as pushed, state[(slot * obs_dim + k) * n_env + e] with action /
// reward / terminal [slot * n_env + e].  One push = one contiguous frame per trace (coalesced 16 B/lane copies).
// Gather: the index tile of a workgroup is staged in LDS once (flat index -> physical state slot,
// next slot, transition slot, env), then
//   * small observations (CartPole: 4 floats): one lane per (sample, component) pair;
//   * large contiguous frames (n_env == 1, e.g. 84x84x4 u8 = 28 224 B): one workgroup per sample
//     streams the two frames with 16 B/lane loads -- the HBM-bandwidth stress of BASELINE config 5.
// Algorithmic bytes per sample: 2 * (2 * obs_bytes + 9)  (SURVEY.md 8d).
#include "common.h"
#include "ring_device.h"
#include "sumtree_device.h"
#include <cmath>

namespace rlhip {

__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                     int64_t n16) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void copy1_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                    int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// streaming launches: one 16-byte chunk per thread (a persistent grid-stride loop measured 20 - 40 % slower on the Adam stream
// at 2^26 parameters, csrc/optim.hip); the loops in the kernels only serve sizes beyond the cap
constexpr int STREAM_GRID_CAP = 1 << 20;

static int32_t copy_bytes(void* dst, const void* src, int64_t bytes, hipStream_t s) {
    if (bytes == 0) return RLHIP_OK;
    if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes) & 15) == 0) {
        int64_t n16 = bytes / 16;
        hipLaunchKernelGGL(copy16_kernel, dim3(grid_for(n16, 256, STREAM_GRID_CAP)), dim3(256), 0, s, (uint4*)dst,
                           (const uint4*)src, n16);
    } else {
        hipLaunchKernelGGL(copy1_kernel, dim3(grid_for(bytes, 256)), dim3(256), 0, s, (uint8_t*)dst,
                           (const uint8_t*)src, bytes);
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// fused push of the three transition traces (one launch instead of three copies)
__global__ __launch_bounds__(256) void push_art_kernel(int32_t* __restrict__ a_dst, float* __restrict__ r_dst,
                                                       uint8_t* __restrict__ t_dst,
                                                       const int32_t* __restrict__ a, const float* __restrict__ r,
                                                       const uint8_t* __restrict__ t, int64_t n) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        a_dst[i] = a[i];
        r_dst[i] = r[i];
        t_dst[i] = t[i];
    }
}

// push!(trajectory, (state = s', action, reward, terminal)) in ONE launch: the state frame (16-byte chunks) and the three
// per-env traces
__global__ __launch_bounds__(256) void push_transition_kernel(uint4* __restrict__ s_dst, const uint4* __restrict__ s_src,
                                                              int64_t n16, int32_t* __restrict__ a_dst,
                                                              float* __restrict__ r_dst, uint8_t* __restrict__ t_dst,
                                                              const int32_t* __restrict__ a, const float* __restrict__ r,
                                                              const uint8_t* __restrict__ t, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, m = n16 > n ? n16 : n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        if (i < n16) s_dst[i] = s_src[i];
        if (i < n) {
            a_dst[i] = a[i];
            r_dst[i] = r[i];
            t_dst[i] = t[i];
        }
    }
}

__global__ __launch_bounds__(256) void sample_indices_kernel(int64_t* __restrict__ out, int64_t batch,
                                                             uint64_t total, uint64_t seed,
                                                             uint32_t draw_ctr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    u32x4 w = philox4x32_10(seed, (uint32_t)b, 0, draw_ctr, TAG_SAMPLER);
    uint64_t x = ((uint64_t)w.x << 32) | (uint64_t)w.y;
    out[b] = (int64_t)__umul64hi(x, total);
}

// debug aid ("a debug build that bounds-checks gather indices"): how many of the flat logical indices lie
// outside [0, total); the first offender's position is kept in out[1] (smallest b)
__global__ __launch_bounds__(256) void check_indices_kernel(const int64_t* __restrict__ idx, int64_t batch, int64_t total,
                                                            unsigned long long* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int64_t j = idx[b];
    if (j < 0 || j >= total) {
        atomicAdd(&out[0], 1ull);
        atomicMin(&out[1], (unsigned long long)b);
    }
}

// Record rings (ring_device.h): the env's observation buffer is component-major (obs_dim x n_env), so the push transposes --
// lane = env, OD coalesced reads, then the record writes of ring_push_transition / ring_push_state (16-byte stores; a wave
// covers 4 KB of contiguous records).  a == NULL: push!(trajectory, (state = s,)).
template <int OD>
__global__ __launch_bounds__(256) void push_record_kernel(void* __restrict__ rec, int64_t slot, int64_t slot_prev,
                                                          const float* __restrict__ obs, int64_t n,
                                                          const int32_t* __restrict__ a, const float* __restrict__ r,
                                                          const uint8_t* __restrict__ t) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < OD; ++k) v[k] = obs[(int64_t)k * n + e];
    if (a) ring_push_transition(rec, slot, slot_prev, n, e, v, a[e], r[e], (uint32_t)t[e]);
    else ring_push_state(rec, slot, n, e, v);
}

static int32_t push_record(void* rec, int64_t slot, int64_t slot_prev, const float* obs, int64_t n, int64_t od, const int32_t* a,
                           const float* r, const uint8_t* t, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256));
#define RLHIP_PUSH_REC(OD_) \
    hipLaunchKernelGGL((push_record_kernel<OD_>), grid, dim3(256), 0, s, rec, slot, slot_prev, obs, n, a, r, t)
    if (od == 4) RLHIP_PUSH_REC(4);
    else if (od == 3) RLHIP_PUSH_REC(3);
    else if (od == 2) RLHIP_PUSH_REC(2);
    else RLHIP_PUSH_REC(1);
#undef RLHIP_PUSH_REC
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

struct RingView {
    int64_t capacity, n_env, obs_dim, head_sa, head_rt;
    const void* state;
    const int32_t* action;
    const float* reward;
    const uint8_t* terminal;
};

constexpr int GATHER_TILE = 256;

// small observations: E = element type (float or uint8_t)
template <typename E>
__global__ __launch_bounds__(256) void gather_small_kernel(RingView rb, const int64_t* __restrict__ idx,
                                                           int64_t batch, E* __restrict__ s,
                                                           int32_t* __restrict__ a, float* __restrict__ r,
                                                           uint8_t* __restrict__ term, E* __restrict__ sn) {
    __shared__ int64_t l_ps[GATHER_TILE], l_pn[GATHER_TILE];
    int64_t b0 = (int64_t)blockIdx.x * GATHER_TILE;
    int tile = (int)((batch - b0 < GATHER_TILE) ? (batch - b0) : GATHER_TILE);
    // stage the index tile: decode each flat index once
    if ((int)threadIdx.x < tile) {
        int64_t j = idx[b0 + threadIdx.x];
        int64_t li = j / rb.n_env, e = j - li * rb.n_env;
        int64_t ps = (rb.head_sa + li) % (rb.capacity + 1);
        int64_t pn = (rb.head_sa + li + 1) % (rb.capacity + 1);
        int64_t pt = (rb.head_rt + li) % rb.capacity;
        l_ps[threadIdx.x] = ps * rb.obs_dim * rb.n_env + e;
        l_pn[threadIdx.x] = pn * rb.obs_dim * rb.n_env + e;
        int64_t o = pt * rb.n_env + e;
        a[b0 + threadIdx.x] = rb.action[o];
        r[b0 + threadIdx.x] = rb.reward[o];
        term[b0 + threadIdx.x] = rb.terminal[o];
    }
    __syncthreads();
    const E* st = (const E*)rb.state;
    int64_t work = (int64_t)tile * rb.obs_dim;
    for (int64_t w = threadIdx.x; w < work; w += blockDim.x) {
        int64_t k = w / tile;
        int bl = (int)(w - k * tile);  // consecutive lanes -> consecutive samples: coalesced writes
        s[k * batch + b0 + bl] = st[l_ps[bl] + k * rb.n_env];
        sn[k * batch + b0 + bl] = st[l_pn[bl] + k * rb.n_env];
    }
}

// PRIO (round 4): the prioritized BatchSampler's draw happens HERE -- thread 0 of the sample's workgroup walks the sum-tree
// (the same Philox draw and descent as sumtree_sample_kernel: bit-identical indices), writes idx / key / priority and goes
// on; the 20 dependent 8-byte reads of a descent (~7 us) hide behind the frame streams of the other resident workgroups, so
// "sample + gather" is one launch instead of two.
struct PrioDraw {
    const float* tree;  // NULL: indices come from idx (uniform sampler / caller-supplied)
    int64_t P, n_leaves;
    uint64_t seed;
    uint32_t draw_ctr;
    int64_t* idx_out;
    int64_t* key_out;  // may be NULL
    float* prio_out;   // may be NULL
    // round 6: a pending priority write-back of <= 64 keys applied INSIDE the launch, before any draw (rlhip_ring_update_sample_
    // gather_prioritized): wave 0 of workgroup 0 runs the one-wavefront update (sumtree_device.h) with write-through stores and
    // raises sync[0]; the drawing wave of every workgroup waits for it and then reads the tree with device-scope loads.
    // upd_n == 0: no update, plain loads (the round-4 / 5 launch, unchanged).
    const int64_t* upd_key;
    const float* upd_prio;
    int upd_n, logP;
    unsigned int* sync;  // [0] "tree updated", [1] departures; both 0 between launches
};
// a tree word for the draw: plain (nothing in this launch writes the tree) or device-scope (after the in-launch update)
__device__ __forceinline__ float tree_ld(const PrioDraw& pd, const float* p) {
    return pd.upd_n > 0 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ float2 tree_ld2(const PrioDraw& pd, const float* p) {  // the child pair (2 node, 2 node + 1): 8-byte aligned
    if (pd.upd_n > 0) {
        const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float2(__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32)));
    }
    return *reinterpret_cast<const float2*>(p);
}
// (Round 5, tried and not kept: an LDS copy of the tree's top 12 levels for batches <= 512, so that thread 0's descent pays L2
// latency for the remaining levels only -- the fused draw + gather of a 32-sample batch went 11.3 -> 10.9 us: the cooperative
// 16 KB load and its barrier cost what twelve L2-resident round trips cost; profiles/r05_summary.md.)
// the bookkeeping behind a descent that ended at heap position `node`: key, priority, logical flat index
__device__ __forceinline__ int64_t prio_draw_finish(const PrioDraw& pd, const RingView& rb, int64_t b, int64_t node) {
    int64_t leaf = node - pd.P;
    if (leaf >= pd.n_leaves) leaf = pd.n_leaves - 1;
    if (pd.key_out) pd.key_out[b] = leaf;
    if (pd.prio_out) pd.prio_out[b] = tree_ld(pd, pd.tree + pd.P + leaf);
    const int64_t pt = leaf / rb.n_env, e = leaf - pt * rb.n_env;
    int64_t li = pt - rb.head_rt;
    if (li < 0) li += rb.capacity;
    const int64_t flat = li * rb.n_env + e;
    pd.idx_out[b] = flat;
    return flat;
}
__device__ __forceinline__ int64_t prio_draw_one(const PrioDraw& pd, const RingView& rb, int64_t b) {
    const u32x4 w = philox4x32_10(pd.seed, (uint32_t)b, 0, pd.draw_ctr, TAG_SAMPLER);
    float v = u01_f32(w.z) * tree_ld(pd, pd.tree + 1);
    int64_t node = 1;
    while (node < pd.P) {  // sumtree_descend (sumtree.hip), restated: never enters a zero-sum subtree
        const float2 c = tree_ld2(pd, pd.tree + 2 * node);
        const bool right = (v > c.x && c.y > 0.0f) || c.x == 0.0f;
        if (right) v -= c.x;
        node = 2 * node + (right ? 1 : 0);
    }
    return prio_draw_finish(pd, rb, b, node);
}
// The same draw by a whole wavefront, FOUR tree levels per memory round trip (round 5: the 20 dependent 8-byte reads of a
// 2^20-leaf descent, ~7 us, were the critical path of a small prioritized batch): lanes 0 .. 14 request the 15 child pairs of the
// depth-4 subtree below the current node at once, then the four decisions are taken from those registers (the pair of relative
// node i on sub-level a sits in lane 2^a - 1 + i; round 6 tried SEVEN levels per trip -- 127 pairs, two loads per lane, three trips
// instead of five: parity-green, batch 32 read 17.8 / 17.1 us against 17.5 / 16.9: no gain, not kept) -- the same comparisons on the same values in the same order as
// prio_draw_one, hence the same leaf.  Called by all 64 lanes of one wave (uniform control flow); lane 0 does the bookkeeping.
__device__ __forceinline__ int64_t prio_draw_wave(const PrioDraw& pd, const RingView& rb, int64_t b) {
    const int lane = (int)threadIdx.x & 63;
    const u32x4 w = philox4x32_10(pd.seed, (uint32_t)b, 0, pd.draw_ctr, TAG_SAMPLER);
    float v = u01_f32(w.z) * tree_ld(pd, pd.tree + 1);
    int64_t node = 1;
    int rem = 0;
    for (int64_t t = 1; t < pd.P; t <<= 1) ++rem;  // levels below the root
    while (rem > 0) {
        const int L = rem < 4 ? rem : 4;
        float cx = 0.0f, cy = 0.0f;
        if (lane < (1 << L) - 1) {
            const int a = 31 - __clz(lane + 1), i = lane + 1 - (1 << a);
            const float2 c = tree_ld2(pd, pd.tree + 2 * ((node << a) + i));
            cx = c.x;
            cy = c.y;
        }
        int i = 0;
        for (int a = 0; a < L; ++a) {
            const int src = (1 << a) - 1 + i;
            const float lx = __shfl(cx, src, 64), ly = __shfl(cy, src, 64);
            const bool right = (v > lx && ly > 0.0f) || lx == 0.0f;
            if (right) v -= lx;
            i = 2 * i + (right ? 1 : 0);
        }
        node = (node << L) + i;
        rem -= L;
    }
    return lane == 0 ? prio_draw_finish(pd, rb, b, node) : 0;
}

// record rings (Float32 observations with OD <= 4 components: the classic-control envs): one LANE per sample, the three
// 16-byte loads of the sample's ONE 64-byte record issued before the first store -- one fabric request per sample (round 4:
// five lines; round 3: eleven; ring_device.h has the measurements).  Stores are coalesced (consecutive lanes = consecutive
// samples).  Nothing is staged in LDS: a lane owns its sample from index to store (the tile-staged generic kernel above
// is the route of the layouts without records).
template <int OD>
__global__ __launch_bounds__(256) void gather_rec_kernel(RingView rb, const int64_t* __restrict__ idx, int64_t batch,
                                                         float* __restrict__ s, int32_t* __restrict__ a, float* __restrict__ r,
                                                         uint8_t* __restrict__ term, float* __restrict__ sn, PrioDraw pd) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const RingRecs rr = {(const uint8_t*)rb.state, rb.capacity, rb.n_env, rb.head_sa};
    const RingTransition t = ring_load_transition(rr, pd.tree ? prio_draw_one(pd, rb, b) : idx[b]);
#pragma unroll
    for (int k = 0; k < OD; ++k) {
        s[k * batch + b] = t.s[k];
        sn[k * batch + b] = t.sn[k];
    }
    a[b] = t.a;
    r[b] = t.r;
    term[b] = (uint8_t)t.t;
}

// large contiguous frames (n_env == 1): one workgroup per sample, 16 B/lane streaming copy.
// Output layout here is sample-major: s[b * frame_bytes ...] (a frame stays contiguous).
__global__ __launch_bounds__(256) void gather_frames_kernel(RingView rb, const int64_t* __restrict__ idx,
                                                            int64_t batch, int64_t frame_bytes,
                                                            uint8_t* __restrict__ s, int32_t* __restrict__ a,
                                                            float* __restrict__ r, uint8_t* __restrict__ term,
                                                            uint8_t* __restrict__ sn, PrioDraw pd) {
    __shared__ int64_t l_off[2];
    int64_t b = blockIdx.x;
    int64_t drawn = 0;
    if (pd.upd_n > 0 && threadIdx.x < 64) {  // (uniform per wave) the pending write-back first: see PrioDraw
        __shared__ SmallUpdateLds l_upd;
        if (blockIdx.x == 0) {
            sumtree_update_small_wave<true>(const_cast<float*>(pd.tree), pd.P, pd.logP, pd.n_leaves, pd.upd_key, pd.upd_prio, pd.upd_n, l_upd);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have reached the device-coherent level
            if (threadIdx.x == 0) __hip_atomic_store(pd.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // workgroups are dispatched in index order, so workgroup 0 is resident before any other can spin here; the bound turns a
        // broken assumption into a loud launch failure instead of a hang (~1 s of polling)
        unsigned int polls = 0;
        while (__hip_atomic_load(pd.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1u << 24)) __builtin_trap();
        }
        asm volatile("" ::: "memory");
    }
    if (pd.tree && threadIdx.x < 64) drawn = prio_draw_wave(pd, rb, b);  // wave 0, four tree levels per round trip
    if (threadIdx.x == 0) {
        int64_t li = pd.tree ? drawn : idx[b];
        int64_t ps = (rb.head_sa + li) % (rb.capacity + 1);
        int64_t pn = (rb.head_sa + li + 1) % (rb.capacity + 1);
        int64_t pt = (rb.head_rt + li) % rb.capacity;
        l_off[0] = ps * frame_bytes;
        l_off[1] = pn * frame_bytes;
        a[b] = rb.action[pt];
        r[b] = rb.reward[pt];
        term[b] = rb.terminal[pt];
    }
    __syncthreads();
    const uint4* src0 = (const uint4*)((const uint8_t*)rb.state + l_off[0]);
    const uint4* src1 = (const uint4*)((const uint8_t*)rb.state + l_off[1]);
    uint4* d0 = (uint4*)(s + b * frame_bytes);
    uint4* d1 = (uint4*)(sn + b * frame_bytes);
    const int64_t n16 = frame_bytes / 16;
    // Sources are read once: non-temporal loads (no allocation in L2 on the way through).  The destination is a pure
    // write stream: ORDINARY 16-byte stores -- the L2 then writes whole lines back; non-temporal stores of the same
    // chunks measured 85 us against 70.5 us per 4096-sample launch (5.4 -> 6.55 TB/s; profiles/r03_store_policy.md).
    constexpr int U = 2;  // chunk pairs in flight per thread and trip
    for (int64_t i0 = threadIdx.x; i0 < n16; i0 += 256 * U) {
        nt_u32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + 256 * u;
            if (i < n16) {
                x[u] = nt_load16(src0 + i);
                y[u] = nt_load16(src1 + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + 256 * u;
            if (i < n16) {
                *reinterpret_cast<nt_u32x4*>(d0 + i) = x[u];
                *reinterpret_cast<nt_u32x4*>(d1 + i) = y[u];
            }
        }
    }
    if (pd.upd_n > 0 && threadIdx.x == 0) {  // re-arm the two words: the workgroup that departs last (every one has passed its wait)
        if (__hip_atomic_fetch_add(pd.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(pd.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pd.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- stack-at-sample gather -----------------------------------------------------------
// The ring stores SINGLE frames; the n_stack-deep observation the reference builds on the way in with StackFrames
// (RLCore/src/utils/stack_frames.jl:11-44: a CircularArrayBuffer of the latest n frames, zero-filled by reset!)
// is assembled on the way out: state stack of transition li = frames li-n+1 .. li, next stack = li-n+2 .. li+1,
// where a frame older than the episode start (a terminal flag between it and the newest frame) or older than the
// ring is all zeros -- exactly what StackFrames holds after reset!.  4x less HBM than storing 4-frame stacks
// (7 GB instead of 28 GB for 2^20 Atari frames); every source frame is read once and written to both stacks.
constexpr int MAX_STACK = 8;

__global__ __launch_bounds__(256) void gather_stacked_kernel(RingView rb, const int64_t* __restrict__ idx,
                                                             int64_t batch, int64_t frame_bytes, int n_stack,
                                                             uint8_t* __restrict__ s, int32_t* __restrict__ a,
                                                             float* __restrict__ r, uint8_t* __restrict__ term,
                                                             uint8_t* __restrict__ sn) {
    __shared__ int64_t l_off[MAX_STACK + 1];
    __shared__ int l_vs[MAX_STACK + 1], l_vn[MAX_STACK + 1];  // validity as state-stack / next-stack member
    __shared__ int l_bound[MAX_STACK + 1];                    // frame j exists and no episode boundary lies before it
    const int64_t b = blockIdx.x;
    const int64_t li = idx[b];
    // one lane per frame fetches its terminal flag (one round trip for all of them), lane 0 chains the verdicts
    if (threadIdx.x <= n_stack) {
        const int j = threadIdx.x;
        const int64_t f = li + 1 - j;  // frame j (j = 0 .. n_stack) is logical state frame li + 1 - j
        int ok = 1;
        if (j >= 1) {  // going one frame further back crosses transition f: stop at an episode boundary
            const bool exists = f >= 0;
            const bool boundary = exists && rb.terminal[(rb.head_rt + f) % rb.capacity] != 0;
            ok = (exists && !boundary) ? 1 : 0;
        }
        l_bound[j] = ok;
        l_off[j] = (f >= 0) ? ((rb.head_sa + f) % (rb.capacity + 1)) * frame_bytes : 0;
    }
    if (threadIdx.x == 64) {  // another wave: the transition's scalars
        const int64_t pt = (rb.head_rt + li) % rb.capacity;
        a[b] = rb.action[pt];
        r[b] = rb.reward[pt];
        term[b] = rb.terminal[pt];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok_n = true, ok_s = true;
        for (int j = 0; j <= n_stack; ++j) {
            const int64_t f = li + 1 - j;
            if (j >= 1) {
                ok_n = ok_n && l_bound[j];
                if (j >= 2) ok_s = ok_s && l_bound[j];
            }
            l_vn[j] = (j < n_stack) && ok_n && f >= 0;              // member k = j of the next-state stack
            l_vs[j] = (j >= 1) && (j == 1 || ok_s) && f >= 0;       // member k = j - 1 of the state stack
        }
    }
    __syncthreads();
    const uint32_t n16 = (uint32_t)(frame_bytes / 16);
    const uint32_t total = n16 * (uint32_t)(n_stack + 1);
    const nt_u32x4 zero = {0u, 0u, 0u, 0u};
    // thread t moves the chunks q = t, t + 256, ... of the flat (frame, chunk) space, so every trip but the last is full
    // (a 7056-byte frame is 441 chunks: per-frame trips of 256 threads would leave 28 % of the second one idle).  Each
    // source chunk is read once (non-temporal) and written to both stacks with ORDINARY stores: the outputs are pure
    // write streams, and non-temporal stores measured 81.5 us against 60.4 us per launch (4.6 -> 6.2 TB/s;
    // profiles/r03_store_policy.md).
    for (uint32_t q = threadIdx.x; q < total; q += 256) {
        const uint32_t j = q / n16, i = q - j * n16;
        nt_u32x4 x = zero;
        if (l_vn[j] | l_vs[j]) x = nt_load16((const uint4*)((const uint8_t*)rb.state + l_off[j]) + i);
        // stacks are oldest-first (StackFrames: the newest frame is the last slice)
        if ((int)j < n_stack)
            *reinterpret_cast<nt_u32x4*>((uint4*)(sn + (b * n_stack + (n_stack - 1 - (int)j)) * frame_bytes) + i) = l_vn[j] ? x : zero;
        if (j >= 1)
            *reinterpret_cast<nt_u32x4*>((uint4*)(s + (b * n_stack + (n_stack - (int)j)) * frame_bytes) + i) = l_vs[j] ? x : zero;
    }
}

// frame = max.(screen1, screen2): the 2-frame max-pool of AtariEnv.act! (RLEnvs/src/environments/3rd_party/
// atari.jl:104-107), fused into the push so the pooled frame is written once
__device__ __forceinline__ uint32_t max_u8x4(uint32_t a, uint32_t b) {
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = (a >> (8 * k)) & 0xFFu, y = (b >> (8 * k)) & 0xFFu;
        o |= (x > y ? x : y) << (8 * k);
    }
    return o;
}
__global__ __launch_bounds__(256) void maxpool_u8_kernel(uint4* __restrict__ dst, const uint4* __restrict__ s1,
                                                         const uint4* __restrict__ s2, int64_t n16) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 x = s1[i], y = s2[i];
        dst[i] = make_uint4(max_u8x4(x.x, y.x), max_u8x4(x.y, y.y), max_u8x4(x.z, y.z), max_u8x4(x.w, y.w));
    }
}

// push!(trajectory, (state = max.(screen1, screen2), action, reward, terminal)) in ONE launch (round 4: was push_art + maxpool):
// the two screens are read once (16-byte non-temporal loads), the pooled frame and the three per-env traces written
template <bool NT_ST>
__global__ __launch_bounds__(256) void push_transition_maxpool_kernel(uint4* __restrict__ dst, const uint4* __restrict__ s1,
                                                                      const uint4* __restrict__ s2, int64_t n16,
                                                                      int32_t* __restrict__ a_dst, float* __restrict__ r_dst,
                                                                      uint8_t* __restrict__ t_dst, const int32_t* __restrict__ a,
                                                                      const float* __restrict__ r, const uint8_t* __restrict__ t,
                                                                      int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = g0; i < n16; i += stride) {
        const nt_u32x4 x = nt_load16(s1 + i), y = nt_load16(s2 + i);
        nt_u32x4 o;
        o[0] = max_u8x4(x[0], y[0]);
        o[1] = max_u8x4(x[1], y[1]);
        o[2] = max_u8x4(x[2], y[2]);
        o[3] = max_u8x4(x[3], y[3]);
        if (NT_ST) nt_store16(dst + i, o);
        else dst[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    for (int64_t i = g0; i < n; i += stride) {
        a_dst[i] = a[i];
        r_dst[i] = r[i];
        t_dst[i] = t[i];
    }
}

// ---- n-step transitions (NStepBatchSampler of RLTrajectories 0.4: un-vendored, PARITY UNPINNED; oracle/rlo_buffer.c states the
// published algorithm).  One lane per sample folds the window li .. li + ns - 1 of its env -- ns = n_step unless a terminal flag
// ends it earlier -- into ONE transition {s_li, a_li, R = r_0 + gamma (r_1 + gamma (...)), any(terminal), s_{li + ns}} and writes it as
// a complete 64-byte record of a batch-sized record ring.  The return is discount_rewards_reduced over the window
// (RLCore/src/utils/basic.jl:237-319: gain = r[i] + gamma * gain from the window's end, Float32, no contraction).  Every DQN
// gradient entry point then runs UNCHANGED on the folded ring with gamma^n as its discount: the record IS the learner's input
// format, so n-step costs one small launch (n_step lines read per sample) and no second copy of the tuned kernels.
constexpr int MAX_NSTEP = 32;
__global__ __launch_bounds__(256) void fold_nstep_kernel(RingView rb, int64_t len_rt, const int64_t* __restrict__ idx, int64_t batch,
                                                         int n_step, float gamma, uint8_t* __restrict__ out, int64_t* __restrict__ iota) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const RingRecs rr = {(const uint8_t*)rb.state, rb.capacity, rb.n_env, rb.head_sa};
    const int64_t fj = idx[b];
    const int64_t li = fj / rb.n_env;
    RingTransition first = ring_load_transition(rr, fj);
    float rew[MAX_NSTEP];
    float sn[4] = {first.sn[0], first.sn[1], first.sn[2], first.sn[3]};
    rew[0] = first.r;
    uint32_t term = first.t;
    int ns = 1;
    bool open = true;  // the window still grows (no `break`: the loop unrolls, rew[] stays in registers)
#pragma unroll
    for (int k = 1; k < MAX_NSTEP; ++k) {
        open = open && k < n_step && !term && li + k < len_rt;  // (a window never runs past the newest stored transition)
        if (open) {
            const RingTransition t = ring_load_transition(rr, fj + (int64_t)k * rb.n_env);
            rew[k] = t.r;
            term = t.t;
#pragma unroll
            for (int c = 0; c < 4; ++c) sn[c] = t.sn[c];
            ns = k + 1;
        } else {
            rew[k] = 0.0f;
        }
    }
    float gain = 0.0f;
#pragma unroll
    for (int k = MAX_NSTEP - 1; k >= 0; --k)
        if (k < ns) gain = rew[k] + gamma * gain;
    uint8_t* o = out + b * RING_REC_BYTES;
    *reinterpret_cast<nt_u32x4*>(o) = nt_u32x4{__float_as_uint(first.s[0]), __float_as_uint(first.s[1]), __float_as_uint(first.s[2]),
                                               __float_as_uint(first.s[3])};
    *reinterpret_cast<nt_u32x4*>(o + 16) = nt_u32x4{(uint32_t)first.a, __float_as_uint(gain), term ? 1u : 0u, 0u};
    *reinterpret_cast<nt_u32x4*>(o + 32) = nt_u32x4{__float_as_uint(sn[0]), __float_as_uint(sn[1]), __float_as_uint(sn[2]), __float_as_uint(sn[3])};
    *reinterpret_cast<nt_u32x4*>(o + 48) = nt_u32x4{0u, 0u, 0u, 0u};
    if (iota) iota[b] = b;
}

static RingView view_of(const rlhip_ring* rb) {
    return {rb->capacity, rb->n_env, rb->obs_dim, rb->head_sa, rb->head_rt,
            rb->state,    rb->action, rb->reward, rb->terminal};
}

static int32_t push_state_frame(rlhip_ring* rb, const void* obs, hipStream_t s) {
    int64_t frames = rb->capacity + 1;
    int64_t fbytes = rb->obs_dim * rb->n_env * (int64_t)rb->elem_bytes;
    int64_t phys;
    if (rb->len_sa < frames) {
        phys = (rb->head_sa + rb->len_sa) % frames;
        rb->len_sa += 1;
    } else {
        phys = rb->head_sa;  // overwrite the oldest frame; it becomes the newest
        rb->head_sa = (rb->head_sa + 1) % frames;
    }
    if (rb->layout == RLHIP_RING_RECORDS)
        return push_record(rb->state, phys, phys, (const float*)obs, rb->n_env, rb->obs_dim, nullptr, nullptr, nullptr, s);
    return copy_bytes((uint8_t*)rb->state + phys * fbytes, obs, fbytes, s);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_ring_init(rlhip_ring* rb, int64_t capacity, int64_t n_env, int64_t obs_dim, int32_t elem_bytes,
                        void* state, int32_t* action, float* reward, uint8_t* terminal) {
    RLHIP_REQUIRE(rb != nullptr, "ring is NULL");
    RLHIP_REQUIRE(capacity >= 1 && n_env >= 1 && obs_dim >= 1, "capacity, n_env, obs_dim must be >= 1");
    RLHIP_REQUIRE(elem_bytes == 4 || elem_bytes == 1, "elem_bytes must be 4 (Float32) or 1 (UInt8)");
    RLHIP_REQUIRE(state != nullptr, "trace storage is NULL");
    const bool records = ring_records(obs_dim, elem_bytes);
    if (records) {
        // the action / reward / terminal traces live inside the 64-byte records of `state` (ring_device.h): a host built
        // against the ABI-1 header (separate arrays, transition-major states) must fail here, not read transposed data later
        RLHIP_REQUIRE(!action && !reward && !terminal,
                      "record ring (Float32, obs_dim <= 4): pass NULL for action / reward / terminal and a state buffer of "
                      "rlhip_ring_state_bytes() bytes (ABI 2)");
        RLHIP_REQUIRE(((uintptr_t)state & 63) == 0, "the record buffer must be 64-byte aligned");
    } else {
        RLHIP_REQUIRE(action && reward && terminal, "trace storage is NULL");
    }
    rb->layout = records ? RLHIP_RING_RECORDS : RLHIP_RING_FRAMES;
    rb->capacity = capacity;
    rb->n_env = n_env;
    rb->obs_dim = obs_dim;
    rb->head_sa = rb->len_sa = rb->head_rt = rb->len_rt = 0;
    rb->elem_bytes = elem_bytes;
    rb->state = state;
    rb->action = action;
    rb->reward = reward;
    rb->terminal = terminal;
    return RLHIP_OK;
}

int64_t rlhip_ring_state_bytes(int64_t capacity, int64_t n_env, int64_t obs_dim, int32_t elem_bytes) {
    if (capacity < 1 || n_env < 1 || obs_dim < 1 || (elem_bytes != 4 && elem_bytes != 1)) return 0;
    if (ring_records(obs_dim, elem_bytes)) return (capacity + 1) * n_env * (int64_t)RING_REC_BYTES;
    return (capacity + 1) * n_env * obs_dim * (int64_t)elem_bytes;
}

int32_t rlhip_ring_layout(const rlhip_ring* rb) { return rb ? rb->layout : -1; }

// The push protocol (RLCore/src/policies/agent/agent_base.jl:45-59; length semantics RLCore/test/policies/agent.jl:27-34), checked
// BEFORE any host counter moves so that a rejected call leaves the ring as it was:
//   * a transition completes the open state: the first push must be a state (len_sa == len_rt + 1 holds from then on);
//   * a second state push while one is open would shift `next_state[i] = state[i + 1]` for every later transition and leave a
//     sampleable slot whose (a, r, t, s') were never written.  The reference's EpisodesBuffer pads such a slot and marks it
//     non-sampleable; this ring has no such mask, so the call is rejected: vector envs auto-reset (one PreEpisode push for the
//     whole run), and a single env pushes the post-reset observation as s' of its terminal transition.
#define RLHIP_RING_REQUIRE_OPEN_STATE(rb) \
    RLHIP_REQUIRE((rb)->len_sa == (rb)->len_rt + 1, "push the first state (rlhip_ring_push_state) before the first transition")
#define RLHIP_RING_REQUIRE_NO_OPEN_STATE(rb)                                                                                   \
    RLHIP_REQUIRE((rb)->len_sa == (rb)->len_rt,                                                                                \
                  "a state is already open (len_sa == len_rt + 1): complete it with rlhip_ring_push_transition; this ring has no " \
                  "pad-and-exclude slot for a second PreEpisodeStage push (see include/rlhip.h)")

int32_t rlhip_ring_push_state(rlhip_ring* rb, const void* obs, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb != nullptr && obs != nullptr, "NULL argument");
    RLHIP_RING_REQUIRE_NO_OPEN_STATE(rb);
    return push_state_frame(rb, obs, as_stream(stream));
}

int32_t rlhip_ring_push_transition(rlhip_ring* rb, const void* next_obs, const int32_t* action,
                                   const float* reward, const uint8_t* terminal, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && next_obs && action && reward && terminal, "NULL argument");
    RLHIP_RING_REQUIRE_OPEN_STATE(rb);
    hipStream_t s = as_stream(stream);
    int64_t frames = rb->capacity, n = rb->n_env, phys;
    if (rb->len_rt < frames) {
        phys = (rb->head_rt + rb->len_rt) % frames;
        rb->len_rt += 1;
    } else {
        phys = rb->head_rt;
        rb->head_rt = (rb->head_rt + 1) % frames;
    }
    // slot of the new state frame (same bookkeeping as push_state_frame)
    const int64_t sframes = rb->capacity + 1;
    const int64_t fbytes = rb->obs_dim * rb->n_env * (int64_t)rb->elem_bytes;
    const int64_t sphys = (rb->len_sa < sframes) ? (rb->head_sa + rb->len_sa) % sframes : rb->head_sa;
    uint8_t* sdst = (uint8_t*)rb->state + sphys * fbytes;
    if (rb->layout == RLHIP_RING_RECORDS) {  // completes the previous slot's record, opens the new one: transposing push, one launch
        if (rb->len_sa < sframes) rb->len_sa += 1;
        else rb->head_sa = (rb->head_sa + 1) % sframes;
        return push_record(rb->state, sphys, (sphys + sframes - 1) % sframes, (const float*)next_obs, n, rb->obs_dim, action,
                           reward, terminal, s);
    }
    if ((((uintptr_t)sdst | (uintptr_t)next_obs | (uintptr_t)fbytes) & 15) == 0 && fbytes <= (64ll << 20)) {
        if (rb->len_sa < sframes) rb->len_sa += 1;
        else rb->head_sa = (rb->head_sa + 1) % sframes;
        const int64_t n16 = fbytes / 16;
        hipLaunchKernelGGL(push_transition_kernel, dim3(grid_for(n16 > n ? n16 : n, 256, STREAM_GRID_CAP)), dim3(256), 0, s, (uint4*)sdst,
                           (const uint4*)next_obs, n16, rb->action + phys * n, rb->reward + phys * n,
                           rb->terminal + phys * n, action, reward, terminal, n);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    hipLaunchKernelGGL(push_art_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, rb->action + phys * n,
                       rb->reward + phys * n, rb->terminal + phys * n, action, reward, terminal, n);
    RLHIP_LAUNCH_CHECK();
    return push_state_frame(rb, next_obs, s);
}

int64_t rlhip_ring_length(const rlhip_ring* rb) { return rb ? rb->len_rt : 0; }

int32_t rlhip_ring_sample_indices(const rlhip_ring* rb, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                                  int64_t* idx_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && idx_out && batch >= 0, "bad arguments");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    if (batch == 0) return RLHIP_OK;
    uint64_t total = (uint64_t)rb->len_rt * (uint64_t)rb->n_env;
    hipLaunchKernelGGL(sample_indices_kernel, dim3((int)((batch + 255) / 256)), dim3(256), 0,
                       as_stream(stream), idx_out, batch, total, seed, draw_ctr);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ring_sample_indices_nstep(const rlhip_ring* rb, int64_t batch, int32_t n_step, uint64_t seed, uint32_t draw_ctr,
                                        int64_t* idx_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && idx_out && batch >= 0, "bad arguments");
    RLHIP_REQUIRE(n_step >= 1 && n_step <= MAX_NSTEP, "n_step must be in 1..32");
    RLHIP_REQUIRE(rb->len_rt >= n_step, "the trajectory holds fewer than n_step transitions");
    if (batch == 0) return RLHIP_OK;
    const uint64_t total = (uint64_t)(rb->len_rt - n_step + 1) * (uint64_t)rb->n_env;
    hipLaunchKernelGGL(sample_indices_kernel, dim3((int)((batch + 255) / 256)), dim3(256), 0, as_stream(stream), idx_out, batch,
                       total, seed, draw_ctr);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ring_fold_nstep(const rlhip_ring* rb, const int64_t* idx, int64_t batch, int32_t n_step, float gamma,
                              rlhip_ring* folded, int64_t* iota_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && idx && folded && batch >= 1, "bad arguments");
    RLHIP_REQUIRE(n_step >= 1 && n_step <= MAX_NSTEP, "n_step must be in 1..32");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "n-step folding is defined for record rings (Float32 observations, obs_dim <= 4)");
    RLHIP_REQUIRE(folded->layout == RLHIP_RING_RECORDS && folded->state != nullptr && folded->capacity >= 1 && folded->n_env == batch &&
                      folded->obs_dim == rb->obs_dim,
                  "`folded` must be a record ring initialised with rlhip_ring_init(capacity >= 1, n_env = batch, the source's obs_dim)");
    RLHIP_REQUIRE(folded->state != rb->state, "`folded` must not alias the source ring");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    hipStream_t st = as_stream(stream);
    // (records are obs_dim-agnostic: components >= obs_dim are zero in the source and copied as such)
    hipLaunchKernelGGL(fold_nstep_kernel, dim3((int)((batch + 255) / 256)), dim3(256), 0, st, view_of(rb), rb->len_rt, idx, batch,
                       (int)n_step, gamma, (uint8_t*)folded->state, iota_out);
    RLHIP_LAUNCH_CHECK();
    // slot 0 of `folded` now holds `batch` complete transitions: one stored vec-step of a `batch`-env ring
    folded->head_sa = 0;
    folded->len_sa = 2;
    folded->head_rt = 0;
    folded->len_rt = 1;
    return RLHIP_OK;
}

float rlhip_gamma_pow(float gamma, int32_t n) { return (float)pow((double)gamma, (double)n); }

int32_t rlhip_ring_check_indices(const rlhip_ring* rb, const int64_t* idx, int64_t batch, int64_t* n_bad_host,
                                 int64_t* first_bad_host, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && n_bad_host && batch >= 0 && (idx || batch == 0), "bad arguments");
    *n_bad_host = 0;
    if (first_bad_host) *first_bad_host = -1;
    if (batch == 0) return RLHIP_OK;
    hipStream_t st = as_stream(stream);
    unsigned long long* d = nullptr;
    RLHIP_CHECK_HIP(hipMallocAsync((void**)&d, 2 * sizeof(unsigned long long), st));
    const unsigned long long init[2] = {0ull, ~0ull};
    RLHIP_CHECK_HIP(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, st));
    const int64_t total = rb->len_rt * rb->n_env;
    hipLaunchKernelGGL(check_indices_kernel, dim3((int)((batch + 255) / 256)), dim3(256), 0, st, idx, batch, total, d);
    unsigned long long h[2] = {0ull, ~0ull};
    hipError_t e1 = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);  // a debugging call: it is allowed to wait
    (void)hipFreeAsync(d, st);
    RLHIP_CHECK_HIP(e1);
    RLHIP_CHECK_HIP(e2);
    *n_bad_host = (int64_t)h[0];
    if (first_bad_host && h[0]) *first_bad_host = (int64_t)h[1];
    return RLHIP_OK;
}

int32_t rlhip_ring_bounds_checked_build(void) {
#ifdef RLHIP_BOUNDS_CHECK
    return 1;
#else
    return 0;
#endif
}

int32_t rlhip_ring_gather_is_frame_major(const rlhip_ring* rb) {
    if (!rb) return 0;
    int64_t frame_bytes = rb->obs_dim * (int64_t)rb->elem_bytes;
    return (rb->n_env == 1 && frame_bytes >= 1024 && (frame_bytes % 16 == 0)) ? 1 : 0;
}

static int32_t ring_gather_impl(const rlhip_ring* rb, const int64_t* idx, int64_t batch, void* s, int32_t* a, float* r,
                                uint8_t* term, void* s_next, const PrioDraw& pd, rlhip_stream_t stream);

int32_t rlhip_ring_gather(const rlhip_ring* rb, const int64_t* idx, int64_t batch, void* s, int32_t* a,
                          float* r, uint8_t* term, void* s_next, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && idx && s && a && r && term && s_next && batch >= 0, "bad arguments");
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    return ring_gather_impl(rb, idx, batch, s, a, r, term, s_next, PrioDraw{}, stream);
}

int32_t rlhip_ring_sample_gather_prioritized(const rlhip_ring* rb, const float* tree, int64_t batch, uint64_t seed,
                                             uint32_t draw_ctr, int64_t* idx_out, int64_t* key_out, float* prio_out,
                                             void* s, int32_t* a, float* r, uint8_t* term, void* s_next,
                                             rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && tree && idx_out && s && a && r && term && s_next && batch >= 0, "bad arguments");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    const int64_t n_leaves = rb->capacity * rb->n_env;
    int64_t P = 1;
    while (P < n_leaves) P <<= 1;
    const bool fused = rlhip_ring_gather_is_frame_major(rb) || rb->layout == RLHIP_RING_RECORDS;
    if (!fused) {  // layouts without a fused kernel (u8 / wide small observations): the two launches
        int32_t rc = rlhip_ring_sample_prioritized(rb, tree, batch, seed, draw_ctr, idx_out, key_out, prio_out, stream);
        if (rc) return rc;
        return rlhip_ring_gather(rb, idx_out, batch, s, a, r, term, s_next, stream);
    }
    return ring_gather_impl(rb, idx_out, batch, s, a, r, term, s_next,
                            PrioDraw{tree, P, n_leaves, seed, draw_ctr, idx_out, key_out, prio_out, nullptr, nullptr, 0, 0, nullptr}, stream);
}

extern "C" int32_t rlhip_sumtree_update(float* tree, int64_t n_leaves, const int64_t* leaf, const float* prio, int64_t n,
                                        rlhip_stream_t stream);

int32_t rlhip_ring_update_sample_gather_prioritized(const rlhip_ring* rb, float* tree, const int64_t* upd_key, const float* upd_prio,
                                                    int64_t n_upd, int64_t batch, uint64_t seed, uint32_t draw_ctr, int64_t* idx_out,
                                                    int64_t* key_out, float* prio_out, void* s, int32_t* a, float* r, uint8_t* term,
                                                    void* s_next, uint32_t* sync, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && tree && idx_out && s && a && r && term && s_next && batch >= 0 && n_upd >= 0, "bad arguments");
    RLHIP_REQUIRE(n_upd == 0 || (upd_key && upd_prio), "NULL write-back arrays");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    const int64_t n_leaves = rb->capacity * rb->n_env;
    int64_t P = 1;
    int logP = 0;
    while (P < n_leaves) P <<= 1, ++logP;
    // ONE launch where the write-back fits the one-wavefront form and the gather is the frame-major one; anything else is the
    // two calls it stands for (same results either way)
    const bool one = n_upd >= 1 && n_upd <= SMALL_UPDATE_MAX && batch >= 1 && sync != nullptr && logP <= SMALL_MAXL &&
                     n_leaves < (1ll << 31) && rlhip_ring_gather_is_frame_major(rb);
    if (!one) {
        if (n_upd > 0) {
            int32_t rc = rlhip_sumtree_update(tree, n_leaves, upd_key, upd_prio, n_upd, stream);
            if (rc) return rc;
        }
        return rlhip_ring_sample_gather_prioritized(rb, tree, batch, seed, draw_ctr, idx_out, key_out, prio_out, s, a, r, term, s_next,
                                                    stream);
    }
    return ring_gather_impl(rb, idx_out, batch, s, a, r, term, s_next,
                            PrioDraw{tree, P, n_leaves, seed, draw_ctr, idx_out, key_out, prio_out, upd_key, upd_prio, (int)n_upd, logP,
                                     sync},
                            stream);
}

static int32_t ring_gather_impl(const rlhip_ring* rb, const int64_t* idx, int64_t batch, void* s, int32_t* a, float* r,
                                uint8_t* term, void* s_next, const PrioDraw& pd, rlhip_stream_t stream) {
    if (batch == 0) return RLHIP_OK;
    hipStream_t st = as_stream(stream);
    RingView v = view_of(rb);
    int64_t frame_bytes = rb->obs_dim * (int64_t)rb->elem_bytes;
    bool big = rlhip_ring_gather_is_frame_major(rb) != 0;
    if (big) {
        RLHIP_REQUIRE(((((uintptr_t)rb->state | (uintptr_t)s | (uintptr_t)s_next) & 15) == 0),
                      "frame-major gather needs 16-byte aligned buffers");
        hipLaunchKernelGGL(gather_frames_kernel, dim3((int)batch), dim3(256), 0, st, v, idx, batch, frame_bytes,
                           (uint8_t*)s, a, r, term, (uint8_t*)s_next, pd);
    } else {
        int grid = (int)((batch + GATHER_TILE - 1) / GATHER_TILE);
        const bool lane = rb->layout == RLHIP_RING_RECORDS;
#define RLHIP_GATHER_LANE(OD)                                                                                      \
    hipLaunchKernelGGL((gather_rec_kernel<OD>), dim3((int)((batch + 255) / 256)), dim3(256), 0, st, v, idx, batch, \
                       (float*)s, a, r, term, (float*)s_next, pd)
        if (lane && rb->obs_dim == 4) RLHIP_GATHER_LANE(4);
        else if (lane && rb->obs_dim == 3) RLHIP_GATHER_LANE(3);
        else if (lane && rb->obs_dim == 2) RLHIP_GATHER_LANE(2);
        else if (lane && rb->obs_dim == 1) RLHIP_GATHER_LANE(1);
#undef RLHIP_GATHER_LANE
        else if (rb->elem_bytes == 4)
            hipLaunchKernelGGL((gather_small_kernel<float>), dim3(grid), dim3(256), 0, st, v, idx, batch,
                               (float*)s, a, r, term, (float*)s_next);
        else
            hipLaunchKernelGGL((gather_small_kernel<uint8_t>), dim3(grid), dim3(256), 0, st, v, idx, batch,
                               (uint8_t*)s, a, r, term, (uint8_t*)s_next);
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ring_gather_stacked(const rlhip_ring* rb, const int64_t* idx, int64_t batch, int32_t n_stack, void* s,
                                  int32_t* a, float* r, uint8_t* term, void* s_next, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && idx && s && a && r && term && s_next && batch >= 0, "bad arguments");
    RLHIP_REQUIRE(n_stack >= 1 && n_stack <= MAX_STACK, "n_stack must be in 1..8");
    RLHIP_REQUIRE(rb->n_env == 1 && rb->layout == RLHIP_RING_FRAMES, "stack-at-sample gather is defined for single-env frame rings");
    const int64_t frame_bytes = rb->obs_dim * (int64_t)rb->elem_bytes;
    RLHIP_REQUIRE(frame_bytes % 16 == 0, "frame size must be a multiple of 16 bytes");
    RLHIP_REQUIRE(((((uintptr_t)rb->state | (uintptr_t)s | (uintptr_t)s_next) & 15) == 0), "buffers must be 16-byte aligned");
    if (batch == 0) return RLHIP_OK;
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    hipLaunchKernelGGL(gather_stacked_kernel, dim3((int)batch), dim3(256), 0, as_stream(stream), view_of(rb), idx, batch,
                       frame_bytes, n_stack, (uint8_t*)s, a, r, term, (uint8_t*)s_next);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

static int32_t maxpool_into_next_state_slot(rlhip_ring* rb, const void* s1, const void* s2, hipStream_t st) {
    RLHIP_REQUIRE(rb->elem_bytes == 1, "the max-pool push is defined for UInt8 frames");
    const int64_t frames = rb->capacity + 1;
    const int64_t fbytes = rb->obs_dim * rb->n_env;
    RLHIP_REQUIRE(fbytes % 16 == 0 && ((((uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)rb->state) & 15) == 0),
                  "frames must be 16-byte aligned multiples of 16 bytes");
    int64_t phys;
    if (rb->len_sa < frames) {
        phys = (rb->head_sa + rb->len_sa) % frames;
        rb->len_sa += 1;
    } else {
        phys = rb->head_sa;
        rb->head_sa = (rb->head_sa + 1) % frames;
    }
    const int64_t n16 = fbytes / 16;
    hipLaunchKernelGGL(maxpool_u8_kernel, dim3(grid_for(n16, 256, STREAM_GRID_CAP)), dim3(256), 0, st,
                       (uint4*)((uint8_t*)rb->state + phys * fbytes), (const uint4*)s1, (const uint4*)s2, n16);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ring_push_state_maxpool(rlhip_ring* rb, const void* screen1, const void* screen2, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && screen1 && screen2, "NULL argument");
    RLHIP_RING_REQUIRE_NO_OPEN_STATE(rb);
    return maxpool_into_next_state_slot(rb, screen1, screen2, as_stream(stream));
}

int32_t rlhip_ring_push_transition_maxpool(rlhip_ring* rb, const void* screen1, const void* screen2,
                                           const int32_t* action, const float* reward, const uint8_t* terminal,
                                           rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && screen1 && screen2 && action && reward && terminal, "NULL argument");
    RLHIP_REQUIRE(rb->elem_bytes == 1, "the max-pool push is defined for UInt8 frames");
    hipStream_t s = as_stream(stream);
    const int64_t fbytes = rb->obs_dim * rb->n_env;
    RLHIP_REQUIRE(fbytes % 16 == 0 && ((((uintptr_t)screen1 | (uintptr_t)screen2 | (uintptr_t)rb->state) & 15) == 0),
                  "frames must be 16-byte aligned multiples of 16 bytes");
    RLHIP_RING_REQUIRE_OPEN_STATE(rb);
    int64_t frames = rb->capacity, n = rb->n_env, phys;
    if (rb->len_rt < frames) {
        phys = (rb->head_rt + rb->len_rt) % frames;
        rb->len_rt += 1;
    } else {
        phys = rb->head_rt;
        rb->head_rt = (rb->head_rt + 1) % frames;
    }
    const int64_t sframes = rb->capacity + 1;
    int64_t sphys;  // slot of the new state frame (same bookkeeping as push_state_frame)
    if (rb->len_sa < sframes) {
        sphys = (rb->head_sa + rb->len_sa) % sframes;
        rb->len_sa += 1;
    } else {
        sphys = rb->head_sa;
        rb->head_sa = (rb->head_sa + 1) % sframes;
    }
    const int64_t n16 = fbytes / 16;
    const int grid = grid_for(n16 > n ? n16 : n, 256, STREAM_GRID_CAP);
    uint4* sdst = (uint4*)((uint8_t*)rb->state + sphys * fbytes);
    // non-temporal stores for the write-once ring frame: 57.6 us per 4096 x 28 KB push against 65.8 with ordinary stores
    hipLaunchKernelGGL((push_transition_maxpool_kernel<true>), dim3(grid), dim3(256), 0, s, sdst, (const uint4*)screen1,
                       (const uint4*)screen2, n16, rb->action + phys * n, rb->reward + phys * n, rb->terminal + phys * n,
                       action, reward, terminal, n);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
