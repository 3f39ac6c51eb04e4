// sumtree.hip -- device-resident priority sum-tree for prioritized replay (BASELINE config 5:
// "1M-slot CircularArraySARTTraces ... prioritized sampling gather").
//
// Replaces, for rings that live in HBM:
//   CircularArrayBuffers.SumTree (compat 0.1.12, RLCore/Project.toml:30) -- `setindex!`, `get`, `rand(rng, t, n)`
//   ReinforcementLearningTrajectories 0.4 `CircularPrioritizedTraces` + the prioritized `BatchSampler` method
//   (`inds, priorities = rand(rng, sumtree, batchsize)`; `trajectory[:priority, keys] = p`)
// Both packages are un-vendored: PARITY UNPINNED, the published algorithm is restated.  Two
// deliberate differences, both documented in DESIGN.md:
//   * internal nodes are RE-COMPUTED as left + right from their children (one f32 add, fixed operand order)
//     instead of the reference's running `tree[parent] += change` delta walk: the tree is then a pure
//     function of the leaf values (no floating-point drift, no dependence on the update order), which is what
//     makes a parallel update bit-reproducible;
//   * the descent never enters a zero-sum subtree while the sibling has mass (the reference can return a
//     zero-priority leaf when u = 0 or on a rounding edge).
//
// Layout: implicit heap, float tree[2P], P = next power of two >= n_leaves; node 1 = root, children of i are
// 2i and 2i+1, leaf k lives at P + k (same positions as the reference's `nparents + k`, nparents = P - 1).
// Leaves are addressed by the PHYSICAL transition slot of the ring (slot * n_env + env), so a push overwrites
// the leaf of the transition it overwrites.  1M leaves = 8 MB: L2 / Infinity-Cache resident.
//
// Kernels (all HBM/L2-latency bound, tiny next to the frame gather they feed):
//   sumtree_level_kernel     one grid per level for bulk range fills (>= 2048 nodes on the level)
//   sumtree_range_kernel     one 1024-thread workgroup finishes the remaining levels of a contiguous range
//   sumtree_update_kernel    leaf writes with "last occurrence wins" for duplicate keys (the sequential reference
//                            semantics), then the ancestors: sparse levels split over workgroups by subtree, the
//                            top 13 levels recomputed whole in LDS by the last-arriving workgroup
//   sumtree_sample_kernel    one lane per draw, log2(P) dependent 8-byte loads
#include "common.h"

namespace rlhip {

static inline int64_t pow2_ge(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}
static inline int log2_of(int64_t p) {
    int l = 0;
    while ((1ll << l) < p) ++l;
    return l;
}

__global__ __launch_bounds__(256) void sumtree_fill_leaves_kernel(float* __restrict__ tree, int64_t first,
                                                                  int64_t count, float value) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) tree[first + i] = value;
}

__global__ __launch_bounds__(256) void sumtree_level_kernel(float* __restrict__ tree, int64_t lo, int64_t hi) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t node = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; node <= hi; node += stride) {
        float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
        tree[node] = c.x + c.y;
    }
}

// remaining levels l = l0 .. logP of the contiguous leaf range [a, b] (heap positions), one workgroup
__global__ __launch_bounds__(1024) void sumtree_range_kernel(float* tree, int64_t a, int64_t b, int l0, int logP) {
    for (int l = l0; l <= logP; ++l) {
        int64_t lo = a >> l, hi = b >> l;
        for (int64_t node = lo + threadIdx.x; node <= hi; node += 1024) {
            float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
            tree[node] = c.x + c.y;
        }
        __syncthreads();  // workgroup-scope release/acquire: the next level reads what this one wrote
    }
}

// Priority write-back `t[keys] .= ps`.  The ancestor chain is latency- AND issue-bound: one dependent global round
// trip per level, and a level's random 8-byte accesses of all keys through ONE CU's memory pipeline (measured: 67 us
// for 4096 keys on a 2^20-leaf tree, growing linearly with the key count).  So:
//  * the tree is cut at the first level with <= TOPN nodes.  Keys below different cut nodes never share a node, so
//    gridDim.x workgroups each take the cut nodes c with (c mod gridDim.x) == blockIdx.x and run election + the
//    sparse levels on their own keys with workgroup barriers only (duplicates of a leaf land in one workgroup);
//    membership is re-evaluated from the key array on every pass (coalesced, L2-resident) instead of building lists;
//  * the child pairs of the sparse levels are touched once up front so that the dependent loads hit the L2;
//  * the workgroup that arrives last (counter in the unused heap slot tree[0], re-armed to 0) recomputes the top of
//    the tree WHOLE in LDS: one load of the 2 TOPN cut-level values, 13 LDS levels, stores on the way.
// Parents are always left + right of the stored children: bit-identical to the sequential reference.
__global__ __launch_bounds__(1024) void sumtree_update_kernel(float* tree, int64_t P, int logP, int64_t n_leaves,
                                                              const int64_t* __restrict__ leaf,
                                                              const float* __restrict__ prio, int64_t n) {
    constexpr int TOPN = 4096;
    __shared__ float l_a[2 * TOPN], l_b[TOPN];  // ping-pong: 2 TOPN children -> TOPN nodes -> TOPN / 2 -> ...
    __shared__ int l_last;
    uint32_t* bits = reinterpret_cast<uint32_t*>(tree);
    const int tid = threadIdx.x;
    const int ltop = logP > 12 ? logP - 12 : 1;  // first level (counted from the leaves) with <= TOPN nodes
    const int64_t gmask = (int64_t)gridDim.x - 1;  // gridDim.x is a power of two <= 2 TOPN
    const int64_t mine = blockIdx.x;
    auto key = [&](int64_t i) -> int64_t {  // out-of-range keys are ignored (never written); -1 also for other
        int64_t k = leaf[i];               // workgroups' keys
        if (k < 0 || k >= n_leaves) return -1;
        return ((((P + k) >> (ltop - 1)) & gmask) == mine) ? k : -1;
    };
    // this workgroup's view of the first 2 TOPN keys (-1 = not mine / out of range) is kept in LDS: every pass below
    // walks the whole key array, and re-deriving membership from global memory costs a round trip per pass
    int32_t* l_keys = reinterpret_cast<int32_t*>(l_a);  // l_a is not needed before the top-of-tree phase
    const int64_t KC = n_leaves < (1ll << 31) ? 2 * TOPN : 0;  // (leaf indices fit the int32 cache)
    for (int64_t i = tid; i < n && i < KC; i += 1024) l_keys[i] = (int32_t)key(i);
    __syncthreads();
    auto keyc = [&](int64_t i) -> int64_t { return i < KC ? (int64_t)l_keys[i] : key(i); };
    // duplicate keys: the LAST occurrence wins (sequential `for (k, p) in zip(keys, ps); t[k] = p; end`).
    // The leaf itself carries the election: zero it, atomicMax the 1-based item number into it, then the
    // winner replaces the tag by its priority.
    for (int64_t i = tid; i < n; i += 1024) {
        int64_t k = keyc(i);
        if (k >= 0) bits[P + k] = 0u;
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += 1024) {
        int64_t k = keyc(i);
        if (k >= 0) atomicMax(&bits[P + k], (uint32_t)(i + 1));
    }
    // chunks of MAXR * 1024 items: every item of a chunk reads its verdict before any winner of that chunk
    // overwrites a tag (a winner is the highest-numbered item of its leaf, so no later chunk reads that leaf)
    constexpr int MAXR = 8;
    for (int64_t base = 0; base < n; base += (int64_t)MAXR * 1024) {
        __syncthreads();
        bool own[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            int64_t i = base + tid + (int64_t)r * 1024;
            int64_t k = i < n ? keyc(i) : -1;
            own[r] = k >= 0 && bits[P + k] == (uint32_t)(i + 1);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            int64_t i = base + tid + (int64_t)r * 1024;
            if (own[r]) tree[P + leaf[i]] = prio[i];
        }
    }
    // sparse levels 1 .. ltop - 1
    if (ltop > 2) {
        for (int64_t i = tid; i < n; i += 1024) {
            int64_t k = keyc(i);
            if (k < 0) continue;
            for (int l = 2; l < ltop; ++l) {  // level 1's pairs are the leaves just written
                float2 c = *reinterpret_cast<const float2*>(tree + 2 * ((P + k) >> l));
                asm volatile("" ::"v"(c.x), "v"(c.y));
            }
        }
    }
    __syncthreads();
    for (int l = 1; l < ltop; ++l) {
        for (int64_t i = tid; i < n; i += 1024) {
            int64_t k = keyc(i);
            if (k < 0) continue;
            int64_t node = (P + k) >> l;
            // agent-scope loads: past the L1, which may hold the line from the touch pass above
            float cx = __hip_atomic_load(tree + 2 * node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float cy = __hip_atomic_load(tree + 2 * node + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tree[node] = cx + cy;  // duplicates write the same value
        }
        __syncthreads();
    }
    // arrival: the last workgroup finishes the top of the tree
    if (gridDim.x > 1) {
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned int prev = __hip_atomic_fetch_add(bits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l_last = (prev == gridDim.x - 1) ? 1 : 0;
            if (l_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(bits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tree[0] = 0.0f again
            }
        }
        __syncthreads();
        if (!l_last) return;
    }
    {
        __syncthreads();  // l_keys (aliasing l_a) is dead from here
        // children of level ltop: level ltop - 1, nodes [P >> (ltop - 1), 2 * that)
        const int64_t cfirst = P >> (ltop - 1), cn = P >> (ltop - 1);
        for (int64_t q = tid; q < cn; q += 1024)
            l_a[q] = __hip_atomic_load(tree + cfirst + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        int64_t m = cn >> 1;  // nodes on level ltop
        float* src = l_a;
        float* dst = l_b;
        for (int l = ltop; l <= logP; ++l) {
            for (int64_t q = tid; q < m; q += 1024) {
                const float v = src[2 * q] + src[2 * q + 1];
                dst[q] = v;
                tree[(P >> l) + q] = v;
            }
            __syncthreads();
            float* t = src;
            src = dst;
            dst = t;
            m >>= 1;
        }
    }
}

__device__ __forceinline__ int64_t sumtree_descend(const float* __restrict__ tree, int64_t P, float v) {
    int64_t node = 1;
    while (node < P) {
        float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
        bool right = (v > c.x && c.y > 0.0f) || c.x == 0.0f;
        if (right) v -= c.x;
        node = 2 * node + (right ? 1 : 0);
    }
    return node - P;
}

// PrioritizedDQN write-back: p = (|td| + eps)^alpha; the power is evaluated in Float64 and rounded once
__global__ __launch_bounds__(256) void per_priority_kernel(const float* __restrict__ td, int64_t n, float eps,
                                                           float alpha, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = fabsf(td[i]) + eps;
    out[i] = (alpha == 1.0f) ? x : (float)pow((double)x, (double)alpha);
}

struct RingGeom {
    int64_t capacity, n_env, head_rt;
};

template <bool RING>
__global__ __launch_bounds__(256) void sumtree_sample_kernel(const float* __restrict__ tree, int64_t P,
                                                             int64_t n_leaves, int64_t batch, uint64_t seed,
                                                             uint32_t draw_ctr, RingGeom rg,
                                                             int64_t* __restrict__ idx_out,
                                                             int64_t* __restrict__ key_out,
                                                             float* __restrict__ prio_out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    u32x4 w = philox4x32_10(seed, (uint32_t)b, 0, draw_ctr, TAG_SAMPLER);
    float v = u01_f32(w.z) * tree[1];  // rand(rng, Float32) * t.tree[1]
    int64_t leaf = sumtree_descend(tree, P, v);
    if (leaf >= n_leaves) leaf = n_leaves - 1;
    if (key_out) key_out[b] = leaf;
    if (prio_out) prio_out[b] = tree[P + leaf];
    if (RING) {
        int64_t pt = leaf / rg.n_env, e = leaf - pt * rg.n_env;
        int64_t li = pt - rg.head_rt;
        if (li < 0) li += rg.capacity;
        idx_out[b] = li * rg.n_env + e;  // logical flat index, the convention of rlhip_ring_gather
    }
}

static int32_t fill_range(float* tree, int64_t n_leaves, int64_t start, int64_t count, float value, hipStream_t s) {
    RLHIP_REQUIRE(n_leaves >= 1 && start >= 0 && count >= 0 && start + count <= n_leaves, "leaf range out of bounds");
    RLHIP_REQUIRE(value >= 0.0f, "priorities must be non-negative");
    if (count == 0) return RLHIP_OK;
    RLHIP_REQUIRE(tree != nullptr, "NULL tree");
    const int64_t P = pow2_ge(n_leaves);
    const int logP = log2_of(P);
    const int64_t a = P + start, b = P + start + count - 1;
    hipLaunchKernelGGL(sumtree_fill_leaves_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, tree, a, count, value);
    int l = 1;
    for (; l <= logP; ++l) {
        int64_t lo = a >> l, hi = b >> l;
        if (hi - lo + 1 <= 2048) break;
        hipLaunchKernelGGL(sumtree_level_kernel, dim3(grid_for(hi - lo + 1, 256)), dim3(256), 0, s, tree, lo, hi);
    }
    if (l <= logP) hipLaunchKernelGGL(sumtree_range_kernel, dim3(1), dim3(1024), 0, s, tree, a, b, l, logP);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int64_t rlhip_sumtree_nodes(int64_t n_leaves) { return n_leaves < 1 ? 0 : 2 * pow2_ge(n_leaves); }

int32_t rlhip_sumtree_fill_range(float* tree, int64_t n_leaves, int64_t start, int64_t count, float value,
                                 rlhip_stream_t stream) {
    return fill_range(tree, n_leaves, start, count, value, as_stream(stream));
}

int32_t rlhip_sumtree_update(float* tree, int64_t n_leaves, const int64_t* leaf, const float* prio, int64_t n,
                             rlhip_stream_t stream) {
    RLHIP_REQUIRE(n_leaves >= 1 && n >= 0, "bad sizes");
    if (n == 0) return RLHIP_OK;
    RLHIP_REQUIRE(tree && leaf && prio, "NULL array");
    RLHIP_REQUIRE(n < (1ll << 31), "too many keys in one update");
    const int64_t P = pow2_ge(n_leaves);
    const int logP = log2_of(P);
    // several workgroups only when there are sparse levels to split (P > 8192) and enough keys to pay for it
    const int groups = (logP <= 13 || n < 1024) ? 1 : (n < 16384 ? 64 : 256);
    hipLaunchKernelGGL(sumtree_update_kernel, dim3(groups), dim3(1024), 0, as_stream(stream), tree, P, logP, n_leaves, leaf,
                       prio, n);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_sumtree_sample(const float* tree, int64_t n_leaves, int64_t batch, uint64_t seed, uint32_t draw_ctr,
                             int64_t* leaf_out, float* prio_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(n_leaves >= 1 && batch >= 0, "bad sizes");
    if (batch == 0) return RLHIP_OK;
    RLHIP_REQUIRE(tree && leaf_out, "NULL array");
    RingGeom rg = {0, 1, 0};
    hipLaunchKernelGGL((sumtree_sample_kernel<false>), dim3((int)((batch + 255) / 256)), dim3(256), 0, as_stream(stream),
                       tree, pow2_ge(n_leaves), n_leaves, batch, seed, draw_ctr, rg, (int64_t*)nullptr, leaf_out,
                       prio_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_per_priority_f32(const float* td, int64_t n, float eps, float alpha, float* out,
                               rlhip_stream_t stream) {
    RLHIP_REQUIRE(n >= 0 && eps >= 0.0f && alpha >= 0.0f, "bad arguments");
    if (n == 0) return RLHIP_OK;
    RLHIP_REQUIRE(td && out, "NULL array");
    hipLaunchKernelGGL(per_priority_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), td, n, eps,
                       alpha, out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ring_push_priority(const rlhip_ring* rb, float* tree, float priority, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && tree, "bad arguments");
    RLHIP_REQUIRE(rb->len_rt >= 1, "no transition has been pushed yet");
    int64_t newest = (rb->head_rt + rb->len_rt - 1) % rb->capacity;
    return fill_range(tree, rb->capacity * rb->n_env, newest * rb->n_env, rb->n_env, priority, as_stream(stream));
}

int32_t rlhip_ring_sample_prioritized(const rlhip_ring* rb, const float* tree, int64_t batch, uint64_t seed,
                                      uint32_t draw_ctr, int64_t* idx_out, int64_t* key_out, float* prio_out,
                                      rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && batch >= 0, "bad arguments");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    if (batch == 0) return RLHIP_OK;
    RLHIP_REQUIRE(tree && idx_out, "NULL array");
    const int64_t n_leaves = rb->capacity * rb->n_env;
    RingGeom rg = {rb->capacity, rb->n_env, rb->head_rt};
    hipLaunchKernelGGL((sumtree_sample_kernel<true>), dim3((int)((batch + 255) / 256)), dim3(256), 0, as_stream(stream),
                       tree, pow2_ge(n_leaves), n_leaves, batch, seed, draw_ctr, rg, idx_out, key_out, prio_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
