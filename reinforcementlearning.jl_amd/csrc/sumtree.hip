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
#include "sumtree_device.h"

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

// Priority write-back `t[keys] .= ps`.  Round 1-3: election on the leaf tags with global atomics (four dependent passes),
// then one dependent global round trip per sparse level behind a workgroup barrier: 30 us for 4096 keys on a 2^20-leaf tree.
// Round 4:
//  * the tree is cut at the first level with <= TOPN nodes (ltop).  Keys below different cut nodes never share a node, so
//    gridDim.x workgroups each take the level-(ltop - 1) nodes c with (c mod gridDim.x) == blockIdx.x -- duplicates of a
//    leaf and all keys of a subtree land in ONE workgroup, whose phases need workgroup barriers only;
//  * ELECTION in LDS: a workgroup's keys (item number + leaf) are compacted into an LDS list while the key array streams by
//    once; "the last occurrence of a leaf wins" (the sequential `for (k, p) in zip(keys, ps); t[k] = p; end`) is then a scan
//    of that short list.  A workgroup whose share exceeds LCAP = 1024 items falls back to the election on the leaf tags
//    (zero the tag, atomicMax the item number, the winner replaces the tag by its priority): any key distribution works;
//  * the sparse levels are not walked: every winner's 2^bl-leaf BLOCK (bl = min(ltop - 1, 7): 128 leaves = 512 contiguous
//    bytes) is recomputed WHOLE by one wave -- one load per lane, bl shuffle-add levels (left + right, the order of the
//    sequential reference), bl coalesced stores -- no dependent global round trip at all (trees above 2^20 leaves walk the
//    remaining ltop - 1 - bl levels the old way);
//  * the workgroup that arrives last (counter in the unused heap slot tree[0], re-armed to 0) recomputes the top of the
//    tree WHOLE in LDS: one load of the 2 TOPN cut-level values, 13 LDS levels, stores on the way.
// Parents are always left + right of the stored children: bit-identical to the sequential reference.
__global__ __launch_bounds__(1024) void sumtree_update_kernel(float* tree, int64_t P, int logP, int64_t n_leaves,
                                                              const int64_t* __restrict__ leaf,
                                                              const float* __restrict__ prio, int64_t n) {
    constexpr int TOPN = 4096;
    constexpr int LCAP = 1024;  // local (item, leaf) list (the election scans it once per entry: one entry per thread at most)
    __shared__ float l_a[2 * TOPN], l_b[TOPN];  // ping-pong: 2 TOPN children -> TOPN nodes -> TOPN / 2 -> ...
    __shared__ int l_last, l_cnt, l_nw;
    uint32_t* bits = reinterpret_cast<uint32_t*>(tree);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NT = blockDim.x, NWV = NT >> 6;  // 256 .. 1024 threads (the grid of a 4096-key update is 256 workgroups: wave dispatch is part of its latency)
    const int ltop = logP > 12 ? logP - 12 : 1;  // first level (counted from the leaves) with <= TOPN nodes
    const int sl = ltop - 1;                      // sparse levels 1 .. sl
    const int bl = sl < 7 ? sl : 7;               // levels recomputed block-wise by one wave
    const int64_t gmask = (int64_t)gridDim.x - 1;  // gridDim.x is a power of two <= 2 TOPN
    const int64_t mine = blockIdx.x;
    auto key = [&](int64_t i) -> int64_t {  // out-of-range keys are ignored (never written); -1 also for other
        int64_t k = leaf[i];               // workgroups' keys
        if (k < 0 || k >= n_leaves) return -1;
        return ((((P + k) >> sl) & gmask) == mine) ? k : -1;
    };
    int32_t* l_item = reinterpret_cast<int32_t*>(l_a);          // [LCAP] item numbers of this workgroup's keys
    int32_t* l_leaf = reinterpret_cast<int32_t*>(l_a) + LCAP;   // [LCAP] their leaves
    int32_t* l_win = reinterpret_cast<int32_t*>(l_b);           // [TOPN] leaves of this round's winners (block recompute)
    if (tid == 0) {
        l_cnt = 0;
        l_nw = 0;
    }
    __syncthreads();
    const bool small_idx = n_leaves < (1ll << 31) && n < (1ll << 31);
    // ---- pass 1: this workgroup's share of the key array -> LDS list (order irrelevant: the item number decides) ----
    // (four keys per thread and trip, every load issued before the first use)
    for (int64_t base = 0; base < n; base += 4 * NT) {
        int64_t kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + tid + NT * u;
            kk[u] = i < n ? leaf[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + tid + NT * u, k = kk[u];
            if (k >= 0 && k < n_leaves && (((P + k) >> sl) & gmask) == mine) {
                const int pos = atomicAdd(&l_cnt, 1);
                if (pos < LCAP && small_idx) {
                    l_item[pos] = (int32_t)i;
                    l_leaf[pos] = (int32_t)k;
                }
            }
        }
    }
    __syncthreads();
    const int cnt = l_cnt;
    // one wave recomputes the 2^bl-leaf blocks of up to four leaves: levels 1 .. bl (agent-scope loads: past this CU's L1; the
    // loads of all four blocks are issued before the first add -- one round trip per four blocks)
    auto blocks_recompute = [&](const int32_t* list, int first, int count, bool is_block_number) {
        if (bl < 1) return;
        const int half = 1 << (bl - 1);  // level-1 nodes of a block (<= 64)
        int64_t base[4];
        float cx[4], cy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t kq = u < count ? (int64_t)list[first + u] : 0;
            const int64_t k = is_block_number ? (kq << bl) : kq;
            base[u] = P + ((k >> bl) << bl);  // heap position of the block's first leaf
            cx[u] = cy[u] = 0.0f;
            if (u < count && lane < half) {
                cx[u] = __hip_atomic_load(tree + base[u] + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                cy[u] = __hip_atomic_load(tree + base[u] + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= count) break;
            float v = cx[u] + cy[u];
            if (lane < half) tree[(base[u] >> 1) + lane] = v;
            for (int m = 2; m <= bl; ++m) {
                const int stride = 1 << (m - 1);
                const float right = __shfl_down(v, stride >> 1, 64);
                v = v + right;  // left + right; meaningful in lanes that are multiples of `stride`
                if (lane < half && (lane & (stride - 1)) == 0) tree[(base[u] >> m) + (lane >> (m - 1))] = v;
            }
        }
    };
    auto recompute_list = [&](const int32_t* list, int nw, bool is_block_number) {
        // wave w takes entries 4 (w + NWV j) .. + 3
        for (int e = 4 * wv; e < nw; e += 4 * NWV) blocks_recompute(list, e, min(4, nw - e), is_block_number);
    };
    if (cnt <= LCAP && small_idx) {
        // ---- election in LDS: entry e wins its leaf iff no other entry of the same leaf has a larger item number ----
        for (int e = tid; e < cnt; e += NT) {
            const int32_t k = l_leaf[e], it = l_item[e];
            bool win = true;
            for (int q = 0; q < cnt; ++q) win = win && !(l_leaf[q] == k && l_item[q] > it);
            if (win) {
                tree[P + k] = prio[it];
                l_win[atomicAdd(&l_nw, 1)] = k;  // cnt <= LCAP = TOPN entries
            }
        }
        __syncthreads();  // the leaves are written (workgroup scope; the block loads below bypass the L1)
        recompute_list(l_win, l_nw, false);
    } else {
        // ---- fallback (a share above LCAP items): election on the leaf tags, rounds of MAXR * 1024 items ----
        for (int64_t i = tid; i < n; i += NT) {
            const int64_t k = key(i);
            if (k >= 0) bits[P + k] = 0u;
        }
        __syncthreads();
        for (int64_t i = tid; i < n; i += NT) {
            const int64_t k = key(i);
            if (k >= 0) atomicMax(&bits[P + k], (uint32_t)(i + 1));
        }
        // every item of a round reads its verdict before any winner of that round overwrites a tag (a winner is the
        // highest-numbered item of its leaf, so no later round reads that leaf)
        constexpr int MAXR = 4;
        for (int64_t base = 0; base < n; base += (int64_t)MAXR * NT) {
            __syncthreads();
            if (tid == 0) l_nw = 0;
            bool own[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int64_t i = base + tid + (int64_t)r * NT;
                const int64_t k = i < n ? key(i) : -1;
                own[r] = k >= 0 && bits[P + k] == (uint32_t)(i + 1);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                const int64_t i = base + tid + (int64_t)r * NT;
                if (own[r]) {
                    tree[P + leaf[i]] = prio[i];
                    l_win[atomicAdd(&l_nw, 1)] = (int32_t)(leaf[i] >> bl);  // block number (fits: P >> bl <= 2^31 for P <= 2^38)
                }
            }
            __syncthreads();
            recompute_list(l_win, l_nw, true);
        }
    }
    __syncthreads();
    // sparse levels above the blocks (trees with more than 2^20 leaves): one dependent round trip per level
    for (int l = bl + 1; l <= sl; ++l) {
        for (int64_t i = tid; i < n; i += NT) {
            const int64_t k = key(i);
            if (k < 0) continue;
            const int64_t node = (P + k) >> l;
            const float cx = __hip_atomic_load(tree + 2 * node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float cy = __hip_atomic_load(tree + 2 * node + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        __syncthreads();  // the lists (aliasing l_a / l_b) are dead from here
        // children of level ltop: level ltop - 1, nodes [P >> (ltop - 1), 2 * that)
        const int64_t cfirst = P >> (ltop - 1), cn = P >> (ltop - 1);
        for (int64_t q = tid; q < cn; q += NT)
            l_a[q] = __hip_atomic_load(tree + cfirst + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        int64_t m = cn >> 1;  // nodes on level ltop
        float* src = l_a;
        float* dst = l_b;
        for (int l = ltop; l <= logP; ++l) {
            if (m == 64) {
                // the last seven levels (64 .. 1 nodes) in ONE wave: shuffle-adds instead of six more barrier + LDS rounds
                if (wv == 0) {
                    float v = src[2 * lane] + src[2 * lane + 1];
                    tree[(P >> l) + lane] = v;
                    for (int st = 1, ll = l + 1; st < 64; st <<= 1, ++ll) {
                        const float right = __shfl_down(v, st, 64);
                        v = v + right;  // left + right; meaningful in lanes that are multiples of 2 st
                        if ((lane & (2 * st - 1)) == 0) tree[(P >> ll) + (lane >> (ll - l))] = v;
                    }
                }
                break;
            }
            for (int64_t q = tid; q < m; q += NT) {
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

// Priority write-back of a SMALL batch (n <= 64 keys: the batch-32 latency regime of SURVEY 8(d) config 5; round 5, VERDICT r4
// item 9): ONE wavefront, two memory round trips, no workgroup barrier and no atomics -- the general kernel above spends
// ~10 us on 32 keys (32 workgroups, arrival counter, the top 13 levels recomputed whole by the last arriver).
//   1. lane i loads (key, priority) i;
//   2. bitonic sort by (leaf, item number): the LAST occurrence of a leaf wins (the sequential `for (k, p) in zip(keys, ps)`)
//      = the last entry of each run of equal leaves; the winners are compacted into lanes 0 .. m - 1, sorted by leaf;
//   3. every winner writes its leaf and requests the SIBLING of each of its logP path nodes -- all loads before the first add;
//   4. the paths are walked upwards together: on a level the active lanes hold distinct node numbers in ascending order, so a
//      node's sibling is on another winner's path iff it is the neighbouring active lane; then the two paths merge (parent =
//      left + right from the two lanes' values, the right lane retires and the neighbour links skip it), otherwise the parent
//      is value + loaded sibling in heap order.  Every parent is `left + right` of its children as stored after the update --
//      the same f32 additions the general kernel (and the sequential reference's recomputation) performs, hence the same bits.
__global__ __launch_bounds__(64) void sumtree_update_small_kernel(float* tree, int64_t P, int logP, int64_t n_leaves,
                                                                  const int64_t* __restrict__ leaf, const float* __restrict__ prio,
                                                                  int n) {
    __shared__ SmallUpdateLds sh;
    sumtree_update_small_wave<false>(tree, P, logP, n_leaves, leaf, prio, n, sh);
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

// Importance-sampling weights of prioritized replay (Schaul et al. 2016; the removed Zoo's PrioritizedDQN, from memory --
// PARITY UNPINNED like the rest of the prioritized path):  w = 1 ./ ((priorities .+ 1f-10) .^ beta);  w ./= maximum(w).
// (N and the total priority of the textbook form (N P(i))^-beta cancel in the normalisation by the maximum.)
// One workgroup: the powers in Float64, rounded once; the maximum over the batch behind one barrier.
__global__ __launch_bounds__(1024) void per_is_weights_kernel(const float* __restrict__ prio, int64_t n, float beta,
                                                              float* __restrict__ out) {
    __shared__ float l_m[16];
    float mx = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float w = (float)(1.0 / pow((double)(prio[i] + 1e-10f), (double)beta));
        out[i] = w;
        mx = fmaxf(mx, w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) l_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = l_m[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, l_m[w]);
    for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = out[i] / mx;  // each thread re-reads what it wrote itself
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
    if (n <= SMALL_UPDATE_MAX && n_leaves < (1ll << 31) && logP <= SMALL_MAXL) {
        // one wavefront, two round trips (sumtree_update_small_kernel): 32 keys on a 2^20-leaf tree 10.6 -> ~6 us
        hipLaunchKernelGGL(sumtree_update_small_kernel, dim3(1), dim3(64), 0, as_stream(stream), tree, P, logP, n_leaves, leaf, prio,
                           (int)n);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    // several workgroups only when there are sparse levels to split (P > 8192) and enough keys to pay for it
    // several workgroups whenever there are sparse levels to split (P > 8192): a workgroup's 16 waves recompute its winners'
    // blocks one after the other (a dependent load -> stores chain each), so the share per workgroup should stay near 16
    // (measured for 4096 keys on a 2^20-leaf tree: 16 / 32 / 64 / 128 / 256 workgroups = 44.9 / 30.7 / 22.8 / 17.2 / 17.9 us:
    // the per-workgroup work decides, not the 256 same-address arrival atomics; tools/sumtree_update_time.py)
    const int groups = logP <= 13 ? 1 : (n < 64 ? 32 : (n < 2048 ? 128 : 256));
    // 1024 threads (us per update of 32 / 512 / 4096 / 65536 keys: 1024 threads 10.2 / 12.2 / 18.2 / 63.9, 512: 11.5 / 14.2 /
    // 20.2 / 85.0, 256: 14.3 / 18.3 / 27.6 / 152 -- the top of the tree wants the threads)
    const int nt = 1024;
    hipLaunchKernelGGL(sumtree_update_kernel, dim3(groups), dim3(nt), 0, as_stream(stream), tree, P, logP, n_leaves, leaf,
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

int32_t rlhip_per_is_weights_f32(const float* prio, int64_t n, float beta, float* w_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(n >= 0 && beta >= 0.0f, "bad arguments");
    if (n == 0) return RLHIP_OK;
    RLHIP_REQUIRE(prio && w_out, "NULL array");
    hipLaunchKernelGGL(per_is_weights_kernel, dim3(1), dim3(1024), 0, as_stream(stream), prio, n, beta, w_out);
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
