// ppo_persist.hip -- optimise!(policy) of the two-layer PPO learner as ONE persistent launch: all n_epochs x
// n_microbatches optimiser steps { gradient of the shuffled micro-batch -> sum over workgroups -> clip_by_global_norm!
// (RLCore/utils/basic.jl:19-29) -> Adam (flux_approximator.jl:46) } without leaving the device (update loop of the removed
// Zoo PPOPolicy: blog a_practical_introduction_to_RL.jl/index.html:15257-15287).
//
// Why: at 4096 envs x T = 32 an optimiser step is 2 tiles of 64 samples per CU -- about 10 us of arithmetic -- and the
// two-launch form (ppo_grad.hip) spends two kernel fill / drain cycles, a 3.4 MB partial-row flush and a grid barrier
// around it: 29 us per step, 16 steps per update.  Here the grid (one workgroup per CU, co-resident) stays, the parameters
// and the Adam state of all ~3.3 k parameters live in registers / LDS of EVERY workgroup, and the two exchanges a step needs
// are data-tagged hand-offs (8-byte {epoch, value} granules written write-through, polled relaxed at agent scope:
// cdna_hip_programming.md Guideline 16 R2) -- no flag, no fence, no atomic counter on the critical path:
//
//   G  gradient of the workgroup's tiles (ppo_grad_tile.h, the same code as ppo_grad_kernel) -> its partial row, published
//      as granules rows[b][p]
//   R  reduce-scatter: workgroup b owns parameter slices {b, b + grid, ...} of 16; 256 threads sweep the slice's column
//      of every row (each granule is its own "ready" flag) and add the rows in the order of reduce_apply_kernel; the 16
//      sums are published as granules gsl[p]
//   A  all-gather: every workgroup sweeps all np granules of gsl, takes the global norm (same Float64 tree as
//      reduce_apply_kernel), clips, and runs Adam on ALL parameters redundantly (4 per thread: same operations in every
//      workgroup, hence the same bits -- nobody has to wait for a parameter broadcast); the new parameters go to LDS
//      (phase 2's per-thread unit weights, and the workgroup's record copy that phase 1a reads: the weights never
//      leave the CU between steps)
//
// The next step's first tile is gathered from the trajectory while R and A wait (sample indices do not depend on the
// parameters).  Results are bit-identical to the two-launch path whenever both run the same grid (same tile -> row
// assignment, same summation orders; tests/test_gpu_persist.py).
//
// Epochs: tag = base + step + 1 with `base` resident in the workspace (advanced by the last workgroup to leave, so a
// captured launch replays correctly); granule memory never needs re-zeroing.  Every spin is bounded: a workgroup that
// gives up raises the abort word, every sweep returns at once, the update is poisoned with NaN and the sticky status word
// reports RLHIP_ETIMEOUT (rlhip_ppo_persist_status).
#include "ppo_grad_tile.h"

#include <stdlib.h>

namespace rlhip {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int PERSIST_MAX_GRID = 256;
constexpr int PERSIST_MAX_NP = 3600;  // two-layer nets with hidden <= 256, ns <= 4, <= 3 actor outputs: np <= 3588
constexpr int SLICE = 16;                         // parameters per reduce-scatter slice
constexpr int RGRP = 16;                          // row groups per slice (= RG of reduce_apply_kernel: same tree)

struct PersistArgs {
    GradArgs g;  // trajectory, policy description, loss constants, seed / device counters (packed / partials unused)
    float* params;
    float* m;
    float* v;
    float* beta_pow;
    float clip_norm, lr, b1, b2, eps;
    uint32_t update_ctr;
    int n_mb, rowlen;
    unsigned long long* rows;  // [grid][rowlen] granules: a partial gradient row + {s_actor, s_critic, s_ent}
    unsigned long long* gsl;   // [np] granules: the reduced gradient
    unsigned int* state;       // [0] epoch base, [1] abort, [2] departures, [3] sticky status
    float* packed;             // the two-launch path's shared record copy (left current at exit)
    float* grad_out;
    float* losses_out;
    unsigned int spin_limit;
    long long* dbg;  // RLHIP_PERSIST_DEBUG: [grid][64 steps][8] s_memtime stamps of thread 0 (tools/persist_timeline.py)
    int test_fault;  // test hook (RLHIP_PERSIST_TEST_FAULT = b + 1): workgroup b withholds its last row -> the abort path
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ void store_granule(gu64* p, unsigned ep, float v) {
    __hip_atomic_store(p, ((unsigned long long)ep << 32) | (unsigned long long)__float_as_uint(v), RLX_AGENT);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;  // buffer descriptor (4 SGPRs)

// A thread collects its granules at byte offsets off0 + k * stride of `buf`, k < cnt (cnt <= N, per lane).  Every pass
// loads all N entries unconditionally, back to back (one memory round trip per pass; an offset beyond the buffer reads
// zeros, entries k >= cnt are ignored): the compiler's s_waitcnt bookkeeping is exact only for unconditional straight-line
// loads -- with per-lane conditional loads it serialised a pass into three round trips, and polling the missing granules
// one at a time cost a round trip each (measured: 15 us outliers).  The wave leaves when all of its lanes are complete.
// Buffer addressing (uniform descriptor + 32-bit offset): one address register per granule instead of two.
// Entries k >= cnt read +0.0f.  Returns false when the update was aborted (by this wave: spin limit; or by anyone: abort
// word).
template <int N>
__device__ __forceinline__ bool sweep_granules(rsrc_t buf, unsigned off0, unsigned stride, int cnt, unsigned ep,
                                               float (&val)[N], gu32* abort_w, unsigned limit) {
    unsigned missing = cnt >= 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u);
#pragma unroll
    for (int k = 0; k < N; ++k) val[k] = 0.0f;
    unsigned spins = 0;
    for (;;) {
        u32x2 x[N];
#pragma unroll
        for (int k = 0; k < N; ++k) x[k] = __builtin_amdgcn_raw_buffer_load_b64(buf, off0 + (unsigned)k * stride, 0, /*sc1*/ 16);
        __builtin_amdgcn_sched_barrier(0);  // all N loads are in flight before the first tag is looked at
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if ((missing & (1u << k)) && x[k].y == ep) {
                val[k] = __uint_as_float(x[k].x);
                missing &= ~(1u << k);
            }
        }
        if (__all(missing == 0u)) return true;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_w, RLX_AGENT) != 0u) return false;
            if (spins > limit) {
                __hip_atomic_store(abort_w, 1u, RLX_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// The step loop is long and every phase addresses memory from the thread index: left alone, the compiler hoists all of
// those address computations out of the loop and keeps ~110 registers of them alive across the tile loop (230 VGPRs
// wanted; at 1024 threads per workgroup the budget is 128).  Each phase therefore starts from a thread index the
// compiler cannot see through, so its addresses are recomputed where they are used (a few VALU instructions per step).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int NS, int ACT, int NO, int NT>
__global__ __launch_bounds__(512 * NT) void ppo_update_persist_kernel(PersistArgs a) {
    constexpr int NTHR = 512 * NT;
    constexpr int PPT = 4096 / NTHR;  // parameters per thread in the Adam phase (np <= 3600)
    constexpr int WPB = NTHR / 64;    // waves per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float l_g[RGRP][SLICE];
    __shared__ double l_sq[64];
    __shared__ float l_loss[4];
    const GradArgs& g = a.g;
    const int h = g.pd.h, nout = g.pd.nout_a, np = g.np;
    const int np_a = (int)g.pd.np_a;
    const TeamIds id = team_ids<NT>(h);
    const TeamLds L = team_lds(smem, id.team);
    // the learner's state, complete in every workgroup, after the teams' tile areas: parameters (flat Flux.destructure
    // layout) | Adam m | Adam v, PERSIST_MAX_NP floats each
    float* l_rec = reinterpret_cast<float*>(smem + (size_t)NT * grad_team_smem_bytes());  // phase 1a's unit records
    float* l_par = reinterpret_cast<float*>(smem + grad_wg_smem_bytes(NT));
    float* l_m = l_par + PERSIST_MAX_NP;
    float* l_v = l_m + PERSIST_MAX_NP;
    const int gtid = threadIdx.x, lane = gtid & 63, wv = __builtin_amdgcn_readfirstlane(gtid >> 6);
    gu32* st = (gu32*)a.state;
    gu32* abort_w = st + 1;
    gu64* rows = (gu64*)a.rows;
    gu64* gsl = (gu64*)a.gsl;
    const int nb = gridDim.x;
    const int64_t rowlen = a.rowlen;
    gu64* myrow = rows + (int64_t)blockIdx.x * rowlen;
    const rsrc_t rows_buf = __builtin_amdgcn_make_buffer_rsrc(a.rows, 0, (int)(8 * rowlen * nb), 0x00020000);
    const rsrc_t gsl_buf = __builtin_amdgcn_make_buffer_rsrc(a.gsl, 0, 8 * 4096, 0x00020000);
    const unsigned base = st[0];

    // the learner's state in LDS, complete in every workgroup; thread gtid works on p = gtid + NTHR k in the Adam phase
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = gtid + NTHR * k;
        if (p < np) {
            l_par[p] = a.params[p];
            l_m[p] = a.m[p];
            l_v[p] = a.v[p];
        }
    }
    float bp1 = a.beta_pow[0], bp2 = a.beta_pow[1];
    for (int i = gtid; i < NW * 32 * RS; i += NTHR) l_rec[i] = 0.0f;  // the padding slots (units beyond h / 8 per wave) stay zero
    const uint32_t upd = g.ctr ? g.ctr[1] : a.update_ctr;
    const int n_mb = a.n_mb;
    const int nsteps = (int)g.n_epochs * n_mb;
    const int tile0 = blockIdx.x * NT + id.team;
    const bool first_loader = id.tid < TILE && tile0 < g.num_tiles;
    PermKeys pk = perm_keys(g.seed, upd * g.n_epochs, g.total);
    TileRegs first;
    if (first_loader) first = fetch_sample<NS>(g, pk, 0u, tile0, id.tid);
    __syncthreads();  // l_par is complete

    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k_)                                              \
    do {                                                       \
        if (a.dbg) ts[k_] = __builtin_amdgcn_s_memtime();      \
    } while (0)
    for (int s = 0; s < nsteps; ++s) {
        STAMP(0);
        const unsigned ep = base + (unsigned)s + 1u;
        const bool last = s + 1 == nsteps;
        const uint32_t pos0 = (uint32_t)(s % n_mb) * g.bm;
        // ---- this step's weights: phase 2's unit in registers, phase 1a's records in the workgroup's LDS copy ----
        UnitW W;
        const int ot0 = opaque(gtid);
        const int uo = ot0 & 255;  // = id.uidx
        {
            const int j = uo < h ? uo : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                W.w1[k] = k < NS ? f2{l_par[j + h * k], l_par[np_a + j + h * k]} : f2{0.0f, 0.0f};
            W.b1 = f2{l_par[h * NS + j], l_par[np_a + h * NS + j]};
            W.w2p = f2{l_par[h * NS + h + 0 + nout * j], l_par[np_a + h * NS + h + j]};
            W.w2a1 = 1 < nout ? l_par[h * NS + h + 1 + nout * j] : 0.0f;
            W.w2a2 = 2 < nout ? l_par[h * NS + h + 2 + nout * j] : 0.0f;
        }
        float b2[4];
#pragma unroll
        for (int o = 0; o < GMAXO; ++o) b2[o] = o < nout ? l_par[h * NS + h + nout * h + o] : 0.0f;
        b2[3] = l_par[np_a + h * NS + h + h];
        if (ot0 < 256 && uo < h) store_record_lds(l_rec, uo, h, W);  // team 0, first half: one thread per hidden unit
        // (complete behind publish_first_tile's barrier; its previous readers left phase 1a three barriers ago)

        UnitG G;
        G.zero();
        HeadG Hd;
        Hd.zero();
        STAMP(1);
        publish_first_tile<NT>(L, id, first_loader, first);
        long long tl[4] = {0, 0, 0, 0};
        grad_tile_loop<NS, ACT, NO, NT>(g, pk, pos0, L, id, l_rec, b2, G, Hd, a.dbg ? tl : nullptr);
        STAMP(2);
        // slot 7: the tile loop's phases, 16 bits each: 1a | 1b | 2 (ticks)
        ts[7] = (tl[1] - tl[0]) | ((tl[2] - tl[1]) << 16) | ((tl[3] - tl[2]) << 32) | ((tl[0] - ts[1]) << 48);
        grad_fold<NT>(smem, L, id, h, G, Hd);

        // ---- G -> R: the partial row, as granules ----
        const int ot1 = opaque(gtid);
        if (ot1 < 512 && !(last && a.test_fault == (int)blockIdx.x + 1)) {  // team 0
            if (ot1 < 256 && ot1 < h) {  // first half: the row's unit gradients
                const int j = ot1;
                gu64* oa_ = myrow;
                gu64* oc_ = myrow + np_a;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    store_granule(oa_ + j + h * k, ep, G.w1[k].x);
                    store_granule(oc_ + j + h * k, ep, G.w1[k].y);
                }
                store_granule(oa_ + h * NS + j, ep, G.b1.x);
                store_granule(oc_ + h * NS + j, ep, G.b1.y);
                store_granule(oa_ + h * NS + h + 0 + nout * j, ep, G.w2p.x);
                if (1 < nout) store_granule(oa_ + h * NS + h + 1 + nout * j, ep, G.w2a1);
                if (2 < nout) store_granule(oa_ + h * NS + h + 2 + nout * j, ep, G.w2a2);
                store_granule(oc_ + h * NS + h + j, ep, G.w2p.y);
            }
            if (ot1 == 0) {
#pragma unroll
                for (int o = 0; o < GMAXO; ++o)
                    if (o < nout) store_granule(myrow + h * NS + h + nout * h + o, ep, Hd.b2a[o]);
                store_granule(myrow + np_a + h * NS + h + h, ep, Hd.b2c);
                store_granule(myrow + np + 0, ep, Hd.s_actor);
                store_granule(myrow + np + 1, ep, Hd.s_critic);
                store_granule(myrow + np + 2, ep, Hd.s_ent);
            }
        }
        __syncthreads();
        STAMP(3);

        // ---- R: reduce-scatter over the rows (summation tree of reduce_apply_kernel: RGRP groups of `per` rows) ----
        const int per = (nb + RGRP - 1) / RGRP;
        const int nsl = (np + SLICE - 1) / SLICE;
        for (int sl = blockIdx.x; sl < nsl; sl += nb) {
            // threads 256..511 (team 0's second half: they published nothing, so no write-through store of their own
            // sits in front of the sweep's loads in the memory pipeline)
            const int ot2 = opaque(gtid) - 256;
            if (ot2 >= 0 && ot2 < RGRP * SLICE) {
                const int pl = ot2 & (SLICE - 1), grp = ot2 / SLICE;
                const int p = sl * SLICE + pl;
                const int b0 = grp * per;
                const int cnt = p < np ? max(0, min(nb, b0 + per) - b0) : 0;  // trailing groups may be empty
                float val[16];
                sweep_granules<16>(rows_buf, 8u * (unsigned)(b0 * (int)rowlen + p), 8u * (unsigned)rowlen, cnt, ep, val,
                                   abort_w, a.spin_limit);
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc += val[q];  // inactive entries are +0.0f, as the padded batches of
                acc += 0.0f;                                 // reduce_apply_kernel (at least one zero follows the rows)
                l_g[grp][pl] = acc;
            }
            __syncthreads();
            if (ot2 >= 0 && ot2 < SLICE) {
                const int p = sl * SLICE + ot2;
                float gsum = 0.0f;
#pragma unroll
                for (int q = 0; q < RGRP; ++q) gsum += l_g[q][ot2];
                if (p < np) store_granule(gsl + p, ep, gsum);
            }
            __syncthreads();
        }

        // ---- A: all-gather of the reduced gradient, global norm, clip, Adam on every parameter ----
        STAMP(4);
        // ---- the next step's first tile: its gather is in flight while the all-gather waits (sample indices do not
        // depend on the parameters; issued here, not before R, whose 16-deep sweep needs the registers) ----
        if (!last) {
            const int mb2 = (s + 1) % n_mb;
            if (mb2 == 0) pk = perm_keys(g.seed, upd * g.n_epochs + (uint32_t)((s + 1) / n_mb), g.total);
            if (first_loader) first = fetch_sample<NS>(g, pk, (uint32_t)mb2 * g.bm, tile0, id.tid);
        }
        float gx[PPT];
        const int ot3 = opaque(gtid);
        sweep_granules<PPT>(gsl_buf, 8u * (unsigned)ot3, 8u * NTHR, ot3 < np ? (np - ot3 + NTHR - 1) / NTHR : 0, ep, gx, abort_w,
                            a.spin_limit);
        STAMP(5);
        {   // Float64 sum of squares per block of 64 parameters, the tree of reduce_apply_kernel (__shfl_down, lane 0);
            // the PPT blocks of a wave are independent chains: no branch between them, so their cross-lane steps overlap
            double sq[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) sq[k] = (double)gx[k] * (double)gx[k];  // lanes beyond np hold +0.0
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) sq[k] += __shfl_down(sq[k], off, 64);
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int c = k * WPB + wv;  // block of 64 parameters this wave holds in pass k
                    if (c < 64) l_sq[c] = sq[k];
                }
            }
        }
        __syncthreads();
        const int nblk = (np + 63) / 64;
        double part = 0.0;  // same summation order in every workgroup (and as reduce_apply_kernel) -> the same norm
        for (int b = lane; b < nblk; b += 64) part += l_sq[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        const float gn = (float)sqrt(part);
        const float scale = (a.clip_norm > 0.0f && a.clip_norm <= gn) ? a.clip_norm / fmaxf(a.clip_norm, gn) : 1.0f;
        const float c1 = 1.0f - bp1, c2 = 1.0f - bp2;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = ot3 + NTHR * k;
            if (p < np) {
                float g1 = gx[k];
                if (scale != 1.0f) g1 *= scale;
                const float mi = a.b1 * l_m[p] + (1.0f - a.b1) * g1;  // Optimisers.Adam, expression order of optim.hip adam1
                const float vi = a.b2 * l_v[p] + (1.0f - a.b2) * (g1 * g1);
                const float d = mi / c1 / (sqrtf(vi / c2) + a.eps) * a.lr;
                l_m[p] = mi;
                l_v[p] = vi;
                l_par[p] = l_par[p] - d;
                if (last && blockIdx.x == 0) a.grad_out[p] = g1;
            }
        }
        bp1 *= a.b1;
        bp2 *= a.b2;
        if (last && blockIdx.x == 0 && a.losses_out != nullptr) {
            // the loss line of the last step, in the order of reduce_apply_kernel (lanes stride the rows, then a tree)
            if (wv >= 1 && wv <= 3) {
                const int c = wv - 1;
                float val[4];
                const int cnt = lane < nb ? (nb - lane + 63) / 64 : 0;
                sweep_granules<4>(rows_buf, 8u * (unsigned)(lane * (int)rowlen + np + c), 8u * 64u * (unsigned)rowlen, cnt, ep, val,
                                  abort_w, a.spin_limit);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < cnt) sum += val[i];
#pragma unroll
                for (int off2 = 32; off2 >= 1; off2 >>= 1) sum += __shfl_down(sum, off2, 64);
                if (lane == 0) l_loss[c] = sum;
            }
            __syncthreads();
            if (gtid == 0) {
                const float actor_loss = -l_loss[0] * g.inv_b;
                const float critic_loss = l_loss[1] * g.inv_b;
                const float ent_loss = l_loss[2] * g.inv_b;
                a.losses_out[0] = g.wa * actor_loss + g.wc * critic_loss - g.we * ent_loss;
                a.losses_out[1] = actor_loss;
                a.losses_out[2] = critic_loss;
                a.losses_out[3] = ent_loss;
            }
        }
        __syncthreads();  // l_par holds the new parameters; l_sq may be rewritten
        STAMP(6);
        if (a.dbg && gtid == 0 && s < 64) {
            long long* d = a.dbg + ((int64_t)blockIdx.x * 64 + s) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = ts[k];
        }
    }
#undef STAMP

    // ---- exit: workgroup 0 leaves the learner's state in global memory (every workgroup holds the same values) ----
    const bool aborted = __hip_atomic_load(abort_w, RLX_AGENT) != 0u;
    if (blockIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = gtid + NTHR * k;
            if (p < np) {
                // a hand-off that never arrived: the step must not be kept silently -- NaN parameters are loud
                a.params[p] = aborted ? __builtin_nanf("") : l_par[p];
                a.m[p] = l_m[p];
                a.v[p] = l_v[p];
            }
        }
        if (gtid == 0) {
            a.beta_pow[0] = bp1;
            a.beta_pow[1] = bp2;
            if (aborted) __hip_atomic_store(st + 3, 1u, RLX_AGENT);
        }
        pack_records(l_par, a.packed, h, NS, nout, g.pd.np_a, gtid, NTHR);
    }
    // departure: the last workgroup out advances the epoch base and re-arms the abort word (every workgroup has read
    // `base` before any workgroup can be the last to leave)
    if (gtid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(st + 2, 1u, RLX_AGENT);
        if (prev == (unsigned)nb - 1u) {
            unsigned nbase = base + (unsigned)nsteps;
            if (nbase >= 0xFFFF0000u) nbase = 0u;  // tags stay non-zero; a tag from 2^32 steps ago was overwritten long since
            __hip_atomic_store(st + 2, 0u, RLX_AGENT);
            __hip_atomic_store(st + 1, 0u, RLX_AGENT);
            __hip_atomic_store(st + 0, nbase, RLX_AGENT);
        }
    }
}

// ---- host ----
constexpr size_t PERSIST_DEBUG_BYTES = (size_t)PERSIST_MAX_GRID * 64 * 8 * 8;  // the LAST bytes of the workspace
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int64_t ppo_persist_bytes(int64_t np, int h) {
    (void)h;
    return (int64_t)(256 + align_up((size_t)PERSIST_MAX_GRID * (size_t)(np + 4) * 8, 256) + align_up(4096 * 8, 256) +
                     PERSIST_DEBUG_BYTES);
}

template <class K>
static int persist_capacity(K kernel, int threads, size_t lds) {
    // every workgroup must be resident at once (they wait for each other): one per CU of the device at hand, if the
    // kernel fits a CU at all.  Cached per device (a process may drive several).
    static int cap[64];
    static bool known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!known[dev]) {
        int per_cu = 0, cus = 0;
        // more than 64 KB of dynamic LDS is an opt-in per kernel and device
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            per_cu = -1;
        }
        if (per_cu < 0) per_cu = 0;
        else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        cap[dev] = per_cu >= 1 ? cus : 0;
        known[dev] = true;
    }
    return cap[dev];
}

template <int NS, int ACT, int NO, int NT>
static int32_t launch_persist(const PersistArgs& a, int grid_wanted, hipStream_t s, bool* launched) {
    auto kernel = &ppo_update_persist_kernel<NS, ACT, NO, NT>;
    const size_t lds = grad_wg_smem_bytes(NT) + 3 * sizeof(float) * PERSIST_MAX_NP;
    int cap = persist_capacity(kernel, 512 * NT, lds);
    if (cap > PERSIST_MAX_GRID) cap = PERSIST_MAX_GRID;
    static int cap_env = -2;
    if (cap_env == -2) {
        const char* e = getenv("RLHIP_PERSIST_MAX_GRID");  // test hook: a smaller device
        cap_env = e ? atoi(e) : -1;
    }
    if (cap_env >= 0 && cap_env < cap) cap = cap_env;
    *launched = false;
    if (cap < 1) return RLHIP_OK;
    const int grid = grid_wanted < cap ? grid_wanted : cap;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512 * NT), lds, s, a);
    RLHIP_LAUNCH_CHECK();
    *launched = true;
    return RLHIP_OK;
}

int32_t ppo_persist_update(const GradLaunch& L0, const rlhip_ppo_cfg* cfg, float* params, float* m, float* v,
                           float* beta_pow, uint32_t update_ctr, void* persist_ws, float* grad_out, float* losses_out,
                           hipStream_t s) {
    static int enabled = -1;
    if (enabled < 0) {
        // opt-in: measured on MI355X at the headline shape it TIES the two-launch path (0.53 vs 0.52 ms per iteration,
        // profiles/r03_persist.md: the two all-to-all hand-offs of a step cost what the two kernel boundaries cost), and
        // a grid of co-resident spinning workgroups must not share the device with another such grid
        const char* e = getenv("RLHIP_PPO_PERSIST");
        enabled = (e && e[0] == '1') ? 1 : 0;
    }
    const int64_t np = L0.np;
    const int64_t nsteps = (int64_t)cfg->n_epochs * cfg->n_microbatches;
    if (!enabled || np > PERSIST_MAX_NP || nsteps < 1 || nsteps > 4096 || L0.nt < 1 || L0.nt > 2) return 1;
    PersistArgs a{};
    a.g = L0.g;
    a.params = params;
    a.m = m;
    a.v = v;
    a.beta_pow = beta_pow;
    a.clip_norm = cfg->max_grad_norm;
    a.lr = cfg->lr;
    a.b1 = cfg->beta1;
    a.b2 = cfg->beta2;
    a.eps = cfg->adam_eps;
    a.update_ctr = update_ctr;
    a.n_mb = cfg->n_microbatches;
    a.rowlen = (int)(np + 4);
    char* w = (char*)persist_ws;
    a.state = (unsigned int*)w;
    w += 256;
    a.rows = (unsigned long long*)w;
    w += align_up((size_t)PERSIST_MAX_GRID * (size_t)(np + 4) * 8, 256);
    a.gsl = (unsigned long long*)w;
    w += align_up(4096 * 8, 256);
    a.dbg = RLHIP_ENV_FLAG("RLHIP_PERSIST_DEBUG") ? (long long*)w : nullptr;
    a.packed = L0.packed;
    a.grad_out = grad_out;
    a.losses_out = losses_out;
    static unsigned int spin_limit = 0;
    if (spin_limit == 0) {
        const char* e = getenv("RLHIP_PERSIST_SPIN_LIMIT");
        spin_limit = e ? (unsigned int)strtoul(e, nullptr, 10) : (1u << 21);  // ~2 s of polling before giving up
        if (spin_limit < 64) spin_limit = 64;
    }
    a.spin_limit = spin_limit;
    static int test_fault = -1;
    if (test_fault < 0) {
        const char* e = getenv("RLHIP_PERSIST_TEST_FAULT");
        test_fault = e ? atoi(e) : 0;
    }
    a.test_fault = test_fault;
    bool launched = false;
    int32_t rc = RLHIP_OK;
#define LAUNCH_P(NS_, ACT_)                                                                        \
    do {                                                                                           \
        if (L0.nt == 2) {                                                                          \
            if (L0.g.pd.nout_a > 2) rc = launch_persist<NS_, ACT_, 3, 2>(a, L0.nb, s, &launched);  \
            else rc = launch_persist<NS_, ACT_, 2, 2>(a, L0.nb, s, &launched);                     \
        } else if (L0.g.pd.nout_a > 2) rc = launch_persist<NS_, ACT_, 3, 1>(a, L0.nb, s, &launched); \
        else rc = launch_persist<NS_, ACT_, 2, 1>(a, L0.nb, s, &launched);                         \
    } while (0)
    const int act = L0.g.pd.act;
    if (L0.ns == 4) { if (act == 0) LAUNCH_P(4, 0); else LAUNCH_P(4, 1); }
    else if (L0.ns == 3) { if (act == 0) LAUNCH_P(3, 0); else LAUNCH_P(3, 1); }
    else { if (act == 0) LAUNCH_P(2, 0); else LAUNCH_P(2, 1); }
#undef LAUNCH_P
    if (rc) return rc;
    return launched ? 0 : 1;
}

}  // namespace rlhip
