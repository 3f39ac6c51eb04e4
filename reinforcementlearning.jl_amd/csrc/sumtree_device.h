// sumtree_device.h -- the one-wavefront priority write-back (<= 64 keys) as a device function, shared by its own launch
// (sumtree.hip: rlhip_sumtree_update) and by the prologue of the fused "update -> draw -> gather" launch (ring.hip, round 6).
#pragma once
#include "common.h"

namespace rlhip {

constexpr int SMALL_UPDATE_MAX = 64;
constexpr int SMALL_MAXL = 32;  // levels (n_leaves < 2^31)

// LDS of one call: 64 keys + 64 values + SMALL_MAXL x 64 siblings
struct SmallUpdateLds {
    uint32_t key[64];
    float val[64];
    float sib[SMALL_MAXL][64];
};

// tree stores: plain when the kernel boundary publishes them (COHERENT = false: the stand-alone launch), device-scope write-through
// when workgroups of the SAME launch read the tree afterwards (COHERENT = true: the fused launch; common.h's hand-off contract)
template <bool COHERENT>
__device__ __forceinline__ void tree_store(float* p, float v) {
    if (COHERENT) store_wt(p, v);
    else *p = v;
}

// Called by ALL 64 lanes of ONE wavefront (uniform control flow); the wave's LDS accesses execute in order, so the exchanges through
// `sh` need no workgroup barrier (the caller may be one wave of a larger workgroup).  Steps: see sumtree.hip.
template <bool COHERENT>
__device__ __forceinline__ void sumtree_update_small_wave(float* tree, int64_t P, int logP, int64_t n_leaves,
                                                          const int64_t* __restrict__ leaf, const float* __restrict__ prio, int n,
                                                          SmallUpdateLds& sh) {
    const int lane = (int)threadIdx.x & 63;
    // ---- 1. load; composite sort key = leaf * 64 + item number (invalid / out-of-range keys sort last) ----
    int64_t k = lane < n ? leaf[lane] : -1;
    float pv = lane < n ? prio[lane] : 0.0f;
    const bool ok = k >= 0 && k < n_leaves;
    uint64_t ck = ok ? ((uint64_t)k << 6) | (uint64_t)lane : ~0ull;
    // ---- 2. bitonic sort over the 64 lanes (ascending) ----
#pragma unroll
    for (int sz = 2; sz <= 64; sz <<= 1) {
#pragma unroll
        for (int st = sz >> 1; st > 0; st >>= 1) {
            const uint32_t olo = __shfl_xor((uint32_t)ck, st, 64), ohi = __shfl_xor((uint32_t)(ck >> 32), st, 64);
            const float opv = __shfl_xor(pv, st, 64);
            const uint64_t ock = ((uint64_t)ohi << 32) | olo;
            const bool up = (lane & sz) == 0;           // direction of this lane's sub-sequence
            const bool lower = (lane & st) == 0;        // this lane keeps the smaller (up) / larger (down) element
            const bool take = (ock < ck) == (up == lower);
            if (ock != ck && take) {
                ck = ock;
                pv = opv;
            }
        }
    }
    const bool valid = ck != ~0ull;
    const uint32_t kk = (uint32_t)(ck >> 6);  // the leaf (n_leaves < 2^31)
    const uint32_t knext = __shfl_down(kk, 1, 64);
    const bool vnext = __shfl_down((int)valid, 1, 64) != 0;
    const bool win = valid && (lane == 63 || !vnext || knext != kk);  // last occurrence of its leaf
    const uint64_t wmask = __ballot(win);
    const int m = __popcll(wmask);
    const int rank = __popcll(wmask & ((1ull << lane) - 1ull));
    if (win) {
        sh.key[rank] = kk;
        sh.val[rank] = pv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool active = lane < m;
    uint32_t cur = active ? (uint32_t)P + sh.key[lane] : 0u;  // heap positions fit 32 bits (n_leaves < 2^31)
    float val = active ? sh.val[lane] : 0.0f;
    // ---- 3. the leaf, and the siblings of the whole path (relaxed agent-scope loads: past this CU's L1).  The loads are issued
    // from an unrolled loop (all in flight together) and parked in LDS, [level][lane]: the walk below is a rolled loop ----
    {
        float sib[SMALL_MAXL];
#pragma unroll
        for (int l = 0; l < SMALL_MAXL; ++l) {
            sib[l] = 0.0f;
            if (active && l < logP) sib[l] = __hip_atomic_load(tree + ((cur >> l) ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (active) tree_store<COHERENT>(tree + cur, val);
#pragma unroll
        for (int l = 0; l < SMALL_MAXL; ++l) sh.sib[l][lane] = sib[l];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 4. walk up: neighbour links over the active lanes (prv / nxt; -1 = none) ----
    int prv = (active && lane > 0) ? lane - 1 : -1;
    int nxt = (active && lane + 1 < m) ? lane + 1 : -1;
    for (int l = 0; l < logP; ++l) {
        const int pl = prv >= 0 ? prv : lane, nl = nxt >= 0 ? nxt : lane;
        const uint32_t lcur = __shfl(cur, pl, 64), rcur = __shfl(cur, nl, 64);
        const float lval = __shfl(val, pl, 64), rval = __shfl(val, nl, 64);
        const float sv = sh.sib[l][lane];
        const bool is_right = (cur & 1u) != 0u;
        const bool sib_left = active && is_right && prv >= 0 && lcur == cur - 1u;   // the even sibling is the left neighbour
        const bool sib_right = active && !is_right && nxt >= 0 && rcur == cur + 1u;
        float parent;
        if (is_right) parent = (sib_left ? lval : sv) + val;   // left + right
        else parent = val + (sib_right ? rval : sv);
        const bool dies = sib_left;  // the pair merges into the left (even) lane
        // links: skip a neighbour that retires on this level (two neighbouring lanes never both retire)
        const int pd = __shfl((int)dies, pl, 64), nd = __shfl((int)dies, nl, 64);
        const int pp = __shfl(prv, pl, 64), nn = __shfl(nxt, nl, 64);
        if (active) {
            if (prv >= 0 && pd) prv = pp;
            if (nxt >= 0 && nd) nxt = nn;
            if (dies) active = false;
            else {
                cur >>= 1;
                val = parent;
                tree_store<COHERENT>(tree + cur, parent);
            }
        }
    }
}

}  // namespace rlhip
