// ppo_grad.hip -- PPO clipped-surrogate loss + gradient over one micro-batch, and the fused
// "reduce partial gradients -> clip_by_global_norm! -> Adam" tail of the single-GPU update.
//
// Replaces the body of the removed Zoo `PPOPolicy` update (hyper-parameters blog
// a_practical_introduction_to_RL.jl/index.html:15257-15278): shuffled micro-batch gather, actor/critic
// forward (Flux Dense chains), softmax / ratio / clamp / min, value and entropy terms, Zygote backward,
// then clip_by_global_norm! (RLCore/utils/basic.jl:19-29) and Flux.Optimise.update! with Adam
// (RLCore/policies/learners/flux_approximator.jl:46).
//
// ppo_grad_kernel: teams of 8 waves walk 64-sample tiles (tile code and its phase description: ppo_grad_tile.h) and
// write one partial gradient row per workgroup; reduce_apply_kernel sums the rows in a fixed order, takes the global
// norm behind a grid barrier (or in the last-arriving workgroup), clips, runs Adam and refreshes the unit records.
// (A persistent whole-update kernel built from the same tile was measured in round 3, tied / lost, and left the product in
// round 4: profiles/attic/ppo_persist.hip, profiles/r03_persist.md.)
// Roofline: VALU-f32 bound by construction (K = ns <= 4 and N = nout <= 3 are far below an MFMA tile);
// algorithmic work 6*h*((ns+nout)+(ns+1)) flop per sample.
#include "ppo_grad_tile.h"

#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

extern "C" int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow,
                                       int64_t n, float grad_scale, float clip_norm, float lr, float beta1,
                                       float beta2, float eps, float* gn_out, rlhip_stream_t stream);
extern "C" int64_t rlhip_ppo_nparams(int32_t kind, const rlhip_ppo_cfg* c);

namespace rlhip {

// NT: teams of 8 waves per workgroup.  NT = 2: a 1024-thread workgroup walks TWO 64-sample tiles side by side (same code,
// same barriers, its own LDS carve per team) and folds both into ONE partial row: half the rows to write at the end of
// the launch (512 rows x 13 KB cost 2.7 us of an 18.8 us launch: the launch cannot retire before they are flushed) and
// half the rows for the optimiser tail to read back.  The two-workgroups-per-CU occupancy of NT = 1 is kept (16 waves).
// Tile code: ppo_grad_tile.h.  Epilogue: each workgroup writes one partial gradient (parameter layout); summation across
// workgroups is done in a fixed order by reduce_apply_kernel, so the result is run-to-run deterministic and replicas on
// different GPUs stay bit-identical.
template <int NS, int ACT, int NO, int NT>
__global__ __launch_bounds__(512 * NT) void ppo_grad_kernel(GradArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long t_entry = 0;
    if (g.dbg) t_entry = __builtin_amdgcn_s_memtime();
    const int h = g.pd.h;
    const TeamIds id = team_ids<NT>(h);
    const TeamLds L = team_lds(smem, id.team);
    const int nout = g.pd.nout_a;
    const float* __restrict__ rec = (const float*)__builtin_assume_aligned(g.packed, 64);

    // ---- prologue: the first tile's scattered gather is issued first; the unit records go to LDS ----
    // (the record image is requested first -- its address costs nothing; the gather's addresses are ~150 instructions away)
    float* l_rec = reinterpret_cast<float*>(smem + (size_t)NT * grad_team_smem_bytes());
    float4 recv[(NW * 32 * 4) / (512 * NT)];
    if (rec) stage_records_load<NT>(recv, rec, h);
    else stage_records_from_params<NT>(recv, g.params, h, NS, nout, g.pd.np_a);
    const PermKeys pk = g.ctr ? perm_keys(g.seed, g.epoch_local + g.ctr[1] * g.n_epochs, g.total) : g.pk;
    // the first tiles of the workgroup's teams are gathered by its FIRST waves (wave t for team t): waves start ~0.1 us apart,
    // team 1's own wave 0 is the workgroup's ninth
    const int lw = (int)threadIdx.x >> 6, ltile = blockIdx.x * NT + lw;
    TileRegs first;
    const bool first_loader = lw < NT && ltile < g.num_tiles;
    if (first_loader) first = fetch_sample<NS>(g, pk, g.pos0, ltile, (int)threadIdx.x & 63);
    stage_records_store<NT>(l_rec, recv);  // phase 1a's copy of the records (complete behind the first barrier)
    UnitG G;
    G.zero();
    HeadG Hd;
    Hd.zero();
    float b2[4];  // {b2a0, b2a1, b2a2, b2c}: behind the records of the packed image, or from the parameter vector
    if (rec) {
        const float* tailb = rec + REC * h;
        b2[0] = tailb[0], b2[1] = tailb[1], b2[2] = tailb[2], b2[3] = tailb[3];
    } else {
        const float* b2a = g.params + h * NS + h + nout * h;
        b2[0] = b2a[0];
        b2[1] = (1 < nout) ? b2a[1] : 0.f;
        b2[2] = (2 < nout) ? b2a[2] : 0.f;
        b2[3] = g.params[g.pd.np_a + h * NS + h + h];
    }

    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g.dbg) ts[0] = __builtin_amdgcn_s_memtime();
    {   // publish into the owning team's buffer 0; a team without a tile gets finite operands (publish_first_tile)
        if (first_loader) {
            const TeamLds Lw = team_lds(smem, lw);
            Lw.x[(int)threadIdx.x & 63] = first.x;
            Lw.misc[(int)threadIdx.x & 63] = first.misc;
        }
        if (NT > 1 && id.tid < TILE && blockIdx.x * NT + id.team >= g.num_tiles) {
            L.x[id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            L.misc[id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            L.x[TILE + id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            L.misc[TILE + id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
    }
    grad_tile_loop<NS, ACT, NO, NT>(g, pk, g.pos0, L, id, l_rec, b2, G, Hd, g.dbg ? ts + 1 : nullptr);
    if (g.dbg) ts[5] = __builtin_amdgcn_s_memtime();
    grad_fold<NT>(smem, L, id, h, G, Hd);
    if (g.dbg && threadIdx.x == 0) {
        ts[6] = __builtin_amdgcn_s_memtime();
        long long* d = g.dbg + (int64_t)blockIdx.x * 8;
#pragma unroll
        for (int k = 0; k < 7; ++k) d[k] = ts[k];
        d[7] = t_entry;
    }
    if (g.samples_out && (NT == 1 || id.team != 0)) {
        // first launch of an update call: this workgroup's slice of the sample records, {x0..x3}, {logp, adv, ret, action bits} per
        // trajectory entry f, written by the threads that have nothing left to do (the team that does not write the partial row)
        const uint32_t nthr = NT == 1 ? 512u : 512u * (NT - 1), me = NT == 1 ? (uint32_t)id.tid : (uint32_t)threadIdx.x - 512u;
        const uint32_t per = (g.total + gridDim.x - 1) / gridDim.x, f0 = blockIdx.x * per;
        const uint32_t f1 = f0 + per < g.total ? f0 + per : g.total;
        const uint32_t n = (uint32_t)g.n;
        for (uint32_t f = f0 + me; f < f1; f += nthr) {
            const uint32_t t = f / n, i = f - t * n;
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NS; ++k) xv[k] = g.obs[((int64_t)t * NS + k) * g.n + i];
            const float a = g.pd.cont ? g.action_f[f] : __int_as_float(g.action_i[f]);
            g.samples_out[2 * (int64_t)f] = make_float4(xv[0], xv[1], xv[2], xv[3]);
            g.samples_out[2 * (int64_t)f + 1] = make_float4(g.logp[f], g.adv[f], g.ret[f], a);
        }
    }
    if (id.team != 0) return;

    // ---- epilogue: this workgroup's partial gradient (fixed layout = parameter layout) ----
    // The row is read by the NEXT launch (reduce_apply_kernel), not by this one -- and yet the stores are device-scope write-through
    // (sc1) ones (store_wt, common.h): with plain stores the 256 workgroups leave 3.4 MB of dirty lines in the eight L2s, which the end of the kernel
    // has to write back before the next launch may start.  Same-box A / B of the headline, three alternations, bit-identical
    // (profiles/raw_r05/partial_row_store_policy_ab.txt): 0.3931 / 0.3910 / 0.3908 -> 0.3769 / 0.3806 / 0.3791 ms per step
    // (-3.2 %); non-temporal stores: no gain.
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
    if (id.owner && id.shalf == 0) {
        const int j = id.uidx;
        float* oa_ = out;
        float* oc_ = out + g.pd.np_a;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            store_wt(&oa_[j + h * k], G.w1[k].x);
            store_wt(&oc_[j + h * k], G.w1[k].y);
        }
        store_wt(&oa_[h * NS + j], G.b1.x);
        store_wt(&oc_[h * NS + j], G.b1.y);
        store_wt(&oa_[h * NS + h + 0 + nout * j], G.w2p.x);
        if (1 < nout) store_wt(&oa_[h * NS + h + 1 + nout * j], G.w2a1);
        if (2 < nout) store_wt(&oa_[h * NS + h + 2 + nout * j], G.w2a2);
        store_wt(&oc_[h * NS + h + j], G.w2p.y);
    }
    if (id.w == 0 && id.lane == 0) {
#pragma unroll
        for (int o = 0; o < GMAXO; ++o)
            if (o < nout) store_wt(&out[h * NS + h + nout * h + o], Hd.b2a[o]);
        store_wt(&out[g.pd.np_a + h * NS + h + h], Hd.b2c);
        float* lo_ = g.loss_partials + (int64_t)blockIdx.x * 4;
        store_wt(&lo_[0], Hd.s_actor);
        store_wt(&lo_[1], Hd.s_critic);
        store_wt(&lo_[2], Hd.s_ent);
        store_wt(&lo_[3], 0.f);
    }
}

static size_t grad_smem_bytes(int nt) { return grad_wg_smem_bytes(nt); }

__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ params, float* __restrict__ packed,
                                                          int h, int ns, int nout, int64_t np_a) {
    pack_records(params, packed, h, ns, nout, np_a, threadIdx.x, blockDim.x);
}

// (Round 3 - 4: `pack_update_kernel` wrote one 32-byte sample record per trajectory entry once per update call; since round 5 the
// first gradient launch of the call does it with its idle team -- ppo_grad_kernel, `samples_out`.)

// ------------------------------------------------------------------------- reduce (+ apply) ----
struct ApplyArgs {
    float* params;
    float* m;
    float* v;
    float* beta_pow;
    float clip_norm, lr, b1, b2, eps;
    unsigned int* counter;  // arrival counter (device), zero between launches
    double* sumsq;          // [gridDim] per-block partial sums of squares
    float* packed;          // unit records to refresh after the step
    int h, ns, nout;
    int64_t np_a;
    long long* dbg;  // RLHIP_GRAD_DEBUG: [workgroup][8] s_memtime stamps of thread 0 (tools/grad_timeline.py), else NULL
};

// grad[p] = sum_b partials[b][p] (fixed order); losses folded by block 0.
// APPLY: the last-arriving workgroup (agent-scope release/acquire around a device counter) computes the
// global norm from the per-block partials, clips, and runs Adam on all parameters -- one launch instead
// of reduce + clip_adam.
constexpr int RP = 64;         // parameters per reduce workgroup = the lanes of wave 0, which continues alone after the reduction
                               // (measured per optimiser step of the headline workload: RP 32 -> 34.7 us, 64 -> 31.9 us, 128 -> 32.7 us)
constexpr int RG = 1024 / RP;  // groups of partial rows per workgroup (RP x RG = 1024 threads)
static_assert(RP == 64, "wave 0 holds the reduced values of all RP parameters");
constexpr int APPLY_NONE = 0, APPLY_LAST = 1, APPLY_GRID = 2, APPLY_XCHG = 3;
// peer exchange of the sharded learner fused into the reduce + apply kernel (APPLY_XCHG; protocol: p2p.hip)
struct XchgArgs {
    float* slot[16];          // comm buffer of every rank (two slots of `cap` floats, then two u32 flags)
    unsigned int* flags[16];
    int rank, world, cap;
    unsigned int seq;
    long long timeout_polls;
    int* status;
    float inv_world;
};

template <int APPLY>
__global__ __launch_bounds__(1024) void reduce_apply_kernel(const float* __restrict__ partials,
                                                            const float* __restrict__ loss_partials, int nb,
                                                            int np, float* __restrict__ grad,
                                                            float* __restrict__ losses, float wa, float wc,
                                                            float we, float inv_b, ApplyArgs ap, XchgArgs xa);
template <int APPLY>
static auto reduce_apply_kernel_ptr() { return &reduce_apply_kernel<APPLY>; }

template <int APPLY>
__global__ __launch_bounds__(1024) void reduce_apply_kernel(const float* __restrict__ partials,
                                                            const float* __restrict__ loss_partials, int nb,
                                                            int np, float* __restrict__ grad,
                                                            float* __restrict__ losses, float wa, float wc,
                                                            float we, float inv_b, ApplyArgs ap, XchgArgs xa) {
    __shared__ float l_g[RG][RP];
    __shared__ double l_d[16];
    __shared__ int l_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (APPLY == APPLY_GRID && ap.dbg) ts[0] = __builtin_amdgcn_s_memtime();
    const int pl = threadIdx.x % RP, grp = threadIdx.x / RP;  // RG groups of RP parameters
    const int p = blockIdx.x * RP + pl;
    const int per = (nb + RG - 1) / RG;
    const int b0 = grp * per, b1 = min(nb, b0 + per);
    float acc = 0.f;
    if (p < np) {
        // all loads of a 32-partial batch are issued before the first add (one memory latency per batch)
        for (int bb = b0; bb < b1; bb += 32) {
            float t[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) t[q] = (bb + q < b1) ? partials[(int64_t)(bb + q) * np + p] : 0.f;
#pragma unroll
            for (int q = 0; q < 32; ++q) acc += t[q];
        }
    }
    if (APPLY == APPLY_GRID && ap.dbg) ts[1] = __builtin_amdgcn_s_memtime();
    l_g[grp][pl] = acc;
    __syncthreads();
    if (APPLY == APPLY_GRID && ap.dbg) ts[2] = __builtin_amdgcn_s_memtime();
    float gsum = 0.f;
    if (grp == 0) {
#pragma unroll
        for (int q = 0; q < RG; ++q) gsum += l_g[q][pl];
        if (p < np) grad[p] = gsum;
        else gsum = 0.f;
    }
    if (blockIdx.x == 0 && losses != nullptr && wv == 1) {
        // the loss line: wave 1 of workgroup 0, alone (no workgroup barrier: wave 0 must not arrive late at the norm exchange
        // every other workgroup waits on; same sums in the same order as when three waves shared the columns)
        float col[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = 0.f;
            for (int b = lane; b < nb; b += 64) a += loss_partials[(int64_t)b * 4 + c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            col[c] = a;
        }
        if (lane == 0) {
            const float actor_loss = -col[0] * inv_b;
            const float critic_loss = col[1] * inv_b;
            const float ent_loss = col[2] * inv_b;
            losses[0] = wa * actor_loss + wc * critic_loss - we * ent_loss;
            losses[1] = actor_loss;
            losses[2] = critic_loss;
            losses[3] = ent_loss;
        }
    }
    if (APPLY == APPLY_NONE) return;

    if (APPLY == APPLY_XCHG) {
        // ---- sharded learner: exchange this workgroup's 64 gradient values with the peers, then clip + Adam ----
        // publish -> local grid barrier -> rank flag (system-scope release) -> poll the peers' flags -> sum the world's
        // slices in rank order (x 1 / world) -> sum of squares -> second grid barrier -> norm -> Adam on the own slice.
        // Counters: ap.counter[0] arrive (publish), [2] arrive (norm), [1] depart.  Same protocol as p2p.hip.
        if (wv != 0) return;
        const bool own = p < np;
        const int par = (int)(xa.seq & 1u);
        const float m0 = own ? ap.m[p] : 0.0f, v0 = own ? ap.v[p] : 0.0f, p0 = own ? ap.params[p] : 0.0f;
        const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
        if (own) __builtin_nontemporal_store(gsum, xa.slot[xa.rank] + (int64_t)par * xa.cap + p);
        __threadfence_system();
        int fail = 0;
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ap.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) {  // the rank's flag goes up when all of its workgroups have published
                while (__hip_atomic_load(ap.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
                    __builtin_amdgcn_s_sleep(1);
                __threadfence_system();
                __hip_atomic_store(xa.flags[xa.rank] + par, xa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            for (int q = 0; q < xa.world && !fail; ++q) {
                long long polls = 0;
                while (__hip_atomic_load(xa.flags[q] + par, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != xa.seq) {
                    if (++polls > xa.timeout_polls) {
                        fail = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (fail) __hip_atomic_store(xa.status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        fail = __shfl(fail, 0, 64);
        __threadfence_system();
        float gx = 0.0f;
        if (own && !fail) {
            for (int q = 0; q < xa.world; ++q) gx += __builtin_nontemporal_load(xa.slot[q] + (int64_t)par * xa.cap + p);
            gx *= xa.inv_world;
        }
        // a peer never arrived: the step must not be taken on an unreduced gradient, and must not be skipped silently
        // either (the replicas would diverge for good) -- NaN gradient -> NaN parameters on this rank, loud everywhere
        // after the next exchange; the host sees the status word (rlhip_comm_check)
        if (fail) gx = __builtin_nanf("");
        double sq = (double)gx * (double)gx;
        sq = wave_sum_down_f64_lane0(sq);  // (the __shfl_down tree, in-row steps on DPP: common.h)
        // the rank-local exchange of the sums of squares: like APPLY_GRID, each partial is its own arrival flag -- two 8-byte
        // {epoch, half of the double} granules in their own part of the sumsq area, polled directly by every workgroup.
        // The epoch is a WORKSPACE-resident word (counter[4] = launches of this variant on this workspace so far; advanced by
        // the last workgroup out, like counter[3] of APPLY_GRID) -- round 3 tagged with the communicator's sequence number,
        // which repeats when a communicator is re-created for the same policy or a caller retries a seq0: a poller could then
        // accept the stale granule of an earlier exchange
        typedef unsigned long long u64;
        const unsigned int xepoch = __hip_atomic_load(ap.counter + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (lane == 0) {
            const u64 bits = (u64)__double_as_longlong(sq), ep = (u64)xepoch << 32;
            u64* gr = reinterpret_cast<u64*>(ap.sumsq) + 512 + 2 * blockIdx.x;
            __hip_atomic_store(gr, ep | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gr + 1, ep | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double part = 0.0;
        for (int b = lane; b < (int)gridDim.x; b += 64) {
            const u64* gr = reinterpret_cast<const u64*>(ap.sumsq) + 512 + 2 * b;
            u64 hi, lo;
            for (;;) {
                hi = __hip_atomic_load(gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = __hip_atomic_load(gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(hi >> 32) == xepoch && (unsigned int)(lo >> 32) == xepoch) break;
                __builtin_amdgcn_s_sleep(1);
            }
            part += __longlong_as_double((long long)(((hi & 0xFFFFFFFFull) << 32) | (lo & 0xFFFFFFFFull)));
        }
        part = wave_sum_xor_f64(part);  // (the __shfl_xor butterfly, in-row steps on DPP: common.h)
        const float gn = (float)sqrt(part);
        const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
        if (own) {
            float g1 = gx;
            if (scale != 1.0f) g1 *= scale;
            const float mi = ap.b1 * m0 + (1.0f - ap.b1) * g1;  // Optimisers.Adam, expression order of optim.hip adam1
            const float vi = ap.b2 * v0 + (1.0f - ap.b2) * (g1 * g1);
            const float d = mi / c1 / (sqrtf(vi / c2) + ap.eps) * ap.lr;
            const float pn = p0 - d;
            ap.m[p] = mi;
            ap.v[p] = vi;
            ap.params[p] = pn;
            grad[p] = g1;
            ap.packed[record_slot(p, ap.h, ap.ns, ap.nout, ap.np_a)] = pn;
        }
        if (lane == 0) {  // (no release fence: see APPLY_GRID's departure)
            unsigned int prev = __hip_atomic_fetch_add(ap.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                ap.beta_pow[0] *= ap.b1;
                ap.beta_pow[1] *= ap.b2;
                __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 4, xepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch: xepoch + 1
            }
        }
        return;
    }

    // per-block partial sum of squares (threads 0..RP-1 of wave 0 hold this block's gradient values)
    if (wv == 0) {
        double sq = (grp == 0) ? (double)gsum * (double)gsum : 0.0;
        sq = wave_sum_down_f64_lane0(sq);  // (the __shfl_down tree, in-row steps on DPP: common.h)
        if (lane == 0) {
            if (APPLY == APPLY_GRID) {
                // the partial IS its own arrival flag: two 8-byte {epoch, half of the double} granules, write-through
                // (device-scope relaxed atomic stores); epoch = launches of this variant so far + 1 (device-resident word,
                // advanced by the last workgroup to leave: replay-safe), so a stale granule never matches
                typedef unsigned long long u64;
                const u64 ep = (u64)(__hip_atomic_load(ap.counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u) << 32;
                const u64 bits = (u64)__double_as_longlong(sq);
                u64* gr = reinterpret_cast<u64*>(ap.sumsq) + 2 * blockIdx.x;
                __hip_atomic_store(gr, ep | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gr + 1, ep | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ap.sumsq[blockIdx.x] = sq;
            }
        }
    }
    if (APPLY == APPLY_GRID) {
        // ---- every workgroup applies Adam to ITS OWN 64 parameters once it has every workgroup's sum of squares ----
        // All workgroups are co-resident (the host launches this variant only for gridDim <= grid_apply_max_blocks()), so
        // polling is safe.  Only wave 0 (which holds the 64 reduced gradient values in registers) continues; the other 15
        // waves are done.  The exchange is ONE hop: lane b polls workgroup b's two granules until their epoch is this
        // launch's (round 2: arrival counter -> spin on the counter -> read the partials: three dependent round trips,
        // 3.1 us of a 9.7 us launch by the stamps of tools/grad_timeline.py).
        if (wv != 0) return;
        const bool own = p < np;
        // operands of this lane's parameter do not depend on the exchange: issue the loads first
        const float m0 = own ? ap.m[p] : 0.0f, v0 = own ? ap.v[p] : 0.0f, p0 = own ? ap.params[p] : 0.0f;
        const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
        typedef unsigned long long u64;
        const unsigned int epoch = __hip_atomic_load(ap.counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (ap.dbg) ts[3] = __builtin_amdgcn_s_memtime();
        double part = 0.0;  // same summation order in every workgroup -> the same norm, bit for bit
        for (int b = lane; b < (int)gridDim.x; b += 64) {
            const u64* gr = reinterpret_cast<const u64*>(ap.sumsq) + 2 * b;
            u64 hi, lo;
            for (;;) {
                hi = __hip_atomic_load(gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = __hip_atomic_load(gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(hi >> 32) == epoch && (unsigned int)(lo >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
            }
            part += __longlong_as_double((long long)(((hi & 0xFFFFFFFFull) << 32) | (lo & 0xFFFFFFFFull)));
        }
        if (ap.dbg) ts[4] = __builtin_amdgcn_s_memtime();
        part = wave_sum_xor_f64(part);  // (the __shfl_xor butterfly, in-row steps on DPP: common.h)
        const float gn = (float)sqrt(part);
        const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
        if (own) {
            float g1 = gsum;
            if (scale != 1.0f) g1 *= scale;
            const float mi = ap.b1 * m0 + (1.0f - ap.b1) * g1;  // Optimisers.Adam, expression order of optim.hip adam1
            const float vi = ap.b2 * v0 + (1.0f - ap.b2) * (g1 * g1);
            const float d = mi / c1 / (sqrtf(vi / c2) + ap.eps) * ap.lr;
            const float pn = p0 - d;
            ap.m[p] = mi;
            ap.v[p] = vi;
            ap.params[p] = pn;
            grad[p] = g1;
            ap.packed[record_slot(p, ap.h, ap.ns, ap.nout, ap.np_a)] = pn;
        }
        if (ap.dbg && lane == 0) {
            ts[5] = __builtin_amdgcn_s_memtime();
            long long* d = ap.dbg + (int64_t)blockIdx.x * 8;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = ts[k];
        }
        // departure: the last workgroup out re-arms the counter and advances the running beta powers and the epoch (every
        // workgroup has USED beta_pow and the epoch above, before its departure increment -- a data dependence, no fence:
        // nothing this workgroup stored is read by another workgroup of this launch; the end of the launch publishes it)
        if (lane == 0) {
            unsigned int prev = __hip_atomic_fetch_add(ap.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                ap.beta_pow[0] *= ap.b1;
                ap.beta_pow[1] *= ap.b2;
                __hip_atomic_store(ap.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 3, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch: epoch + 1
            }
        }
        return;
    }
    // publish: plain stores -> barrier -> one lane: agent-scope release, drain, counter increment
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int prev = __hip_atomic_fetch_add(ap.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l_last = (prev == gridDim.x - 1) ? 1 : 0;
        if (l_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!l_last) return;
    // ---- last workgroup: global norm -> clip -> Adam over all parameters ----
    double part = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x)
        part += __hip_atomic_load(ap.sumsq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part = wave_sum_down_f64_lane0(part);
    if (lane == 0) l_d[wv] = part;
    __syncthreads();
    double tot = 0.0;
    const int nwv = blockDim.x >> 6;
    for (int q = 0; q < nwv; ++q) tot += l_d[q];
    const float gn = (float)sqrt(tot);
    const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
    const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
    __syncthreads();  // every thread has read beta_pow
    // four parameters per thread per trip, all 16 loads issued before the first dependent instruction: with
    // np = 3331 the whole Adam step is one memory round trip instead of four
    constexpr int U = 4;
    for (int q0 = threadIdx.x; q0 < np; q0 += (int)blockDim.x * U) {
        float gi[U], mm[U], vv[U], pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * (int)blockDim.x;
            const bool in = q < np;
            gi[u] = in ? __hip_atomic_load(grad + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            mm[u] = in ? ap.m[q] : 0.0f;
            vv[u] = in ? ap.v[q] : 0.0f;
            pp[u] = in ? ap.params[q] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u * (int)blockDim.x;
            if (q >= np) continue;
            float g1 = gi[u];
            if (scale != 1.0f) g1 *= scale;
            // Optimisers.Adam (same expression order as optim.hip adam1)
            const float mi = ap.b1 * mm[u] + (1.0f - ap.b1) * g1;
            const float vi = ap.b2 * vv[u] + (1.0f - ap.b2) * (g1 * g1);
            ap.m[q] = mi;
            ap.v[q] = vi;
            const float d = mi / c1 / (sqrtf(vi / c2) + ap.eps) * ap.lr;
            ap.params[q] = pp[u] - d;
            grad[q] = g1;
        }
    }
    __syncthreads();  // this workgroup's parameter stores are visible to its own later loads
    pack_records(ap.params, ap.packed, ap.h, ap.ns, ap.nout, ap.np_a, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) {
        ap.beta_pow[0] *= ap.b1;
        ap.beta_pow[1] *= ap.b2;
        __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
    }
}

// all workgroups of the grid-barrier variants must be co-resident: the bound comes from the occupancy of the kernel on the
// device at hand (grid_barrier_capacity, common.h), queried once per device
template <int APPLY>
static int grid_apply_max_blocks() {
    static PerDeviceInt cap_cache;
    // never above 256: the granule areas of APPLY_GRID (u64 words 0 .. 511 of sumsq) and APPLY_XCHG (512 .. 1023) hold two
    // words per workgroup and must not overlap, whatever a future device's occupancy says
    const int cap = grid_barrier_capacity_cached(cap_cache, reduce_apply_kernel_ptr<APPLY>(), 1024);
    return cap < 256 ? cap : 256;
}

static int grad_blocks(int num_tiles) { return num_tiles < MAX_GRAD_BLOCKS ? num_tiles : MAX_GRAD_BLOCKS; }

// location of the unit-record copy inside the workspace (the carve of prepare_grad)
static float* workspace_packed(void* workspace, int64_t np) {
    float* partials = (float*)workspace;
    float* loss_partials = partials + (int64_t)MAX_GRAD_BLOCKS * np;
    uintptr_t q = (uintptr_t)(loss_partials + (int64_t)MAX_GRAD_BLOCKS * 4);
    q = (q + 15) & ~(uintptr_t)15;
    double* sumsq = (double*)q;
    unsigned int* counter = (unsigned int*)(sumsq + 4096);
    long long* dbgp = (long long*)(counter + 16);
    uintptr_t pq = (uintptr_t)(dbgp + (int64_t)MAX_GRAD_BLOCKS * 8);
    pq = (pq + 63) & ~(uintptr_t)63;
    return (float*)pq;
}

// ---- sized workspaces (ABI 2) --------------------------------------------------------------
// rlhip_ppo_update_f32 writes 32 bytes of sample records per trajectory entry behind the fixed part of the workspace, and the
// ABI-1 calls carried no size to check that against.  rlhip_ppo_workspace_init registers (pointer -> bytes) in a host-side table;
// every entry point that takes a workspace looks its pointer up and compares with rlhip_ppo_workspace_bytes(kind, cfg, n, T) of
// THIS call: too small, or never registered, is RLHIP_EINVAL instead of a write past the allocation.
static std::mutex g_ws_mutex;
static std::unordered_map<const void*, int64_t>& ws_table() {
    static std::unordered_map<const void*, int64_t> t;
    return t;
}
static int32_t ws_check(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const void* workspace) {
    RLHIP_REQUIRE(cfg && workspace, "NULL argument");
    const int64_t need = rlhip_ppo_workspace_bytes(kind, cfg, n, T);
    RLHIP_REQUIRE(need > 0, "bad configuration");
    int64_t have = -1;
    {
        std::lock_guard<std::mutex> lk(g_ws_mutex);
        auto it = ws_table().find(workspace);
        if (it != ws_table().end()) have = it->second;
    }
    RLHIP_REQUIRE(have >= 0, "this workspace was never registered: call rlhip_ppo_workspace_init(workspace, bytes, stream) once "
                             "after allocating it (ABI 2)");
    if (have < need) {
        set_error("invalid argument: workspace of %lld bytes is too small for n = %lld, T = %lld (needs %lld: size it by "
                  "rlhip_ppo_workspace_bytes for the largest n * T it is used with)", (long long)have, (long long)n, (long long)T,
                  (long long)need);
        return RLHIP_EINVAL;
    }
    return RLHIP_OK;
}

static int32_t prepare_grad(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                            const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace,
                            GradLaunch* out, const uint32_t* ctr = nullptr) {
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
    RLHIP_REQUIRE(traj && params && workspace, "NULL argument");
    if ((rc = ws_check(kind, cfg, n, T, workspace))) return rc;
    RLHIP_REQUIRE(pd.h <= 256 && pd.h % NW == 0, "the fused gradient kernel supports hidden <= 256, multiple of 8");
    RLHIP_REQUIRE(pd.nout_a <= GMAXO, "the fused gradient kernel supports at most 3 actor outputs");
    RLHIP_REQUIRE(n >= 1 && T >= 1 && n * T <= 0x7FFFFFFFll, "n * T out of range");
    RLHIP_REQUIRE(cfg->n_microbatches >= 1 && mb >= 0 && mb < cfg->n_microbatches, "bad micro-batch index");
    int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    int64_t np = pd.np_a + mlp2_nparams(ns, pd.h, 1);
    uint32_t total = (uint32_t)(n * T);
    uint32_t bm = total / (uint32_t)cfg->n_microbatches;
    RLHIP_REQUIRE(bm >= 1, "micro-batch is empty");
    GradArgs& g = out->g;
    g.obs = traj->obs;
    g.logp = traj->logp;
    g.adv = traj->adv;
    g.ret = traj->ret;
    g.action_f = traj->action_f;
    g.action_i = traj->action_i;
    g.params = params;
    g.n = n;
    g.total = total;
    g.bm = bm;
    g.pos0 = (uint32_t)mb * bm;
    g.num_tiles = (int)((bm + TILE - 1) / TILE);
    g.np = (int)np;
    g.pd = pd;
    g.lo = 1.0f - cfg->clip_range;
    g.hi = 1.0f + cfg->clip_range;
    g.wa = cfg->actor_loss_weight;
    g.wc = cfg->critic_loss_weight;
    g.we = cfg->entropy_loss_weight;
    g.inv_b = 1.0f / (float)bm;
    g.min_logp = (float)log(1e-8);
    g.pk = perm_keys(seed, epoch_ctr, total);
    g.ctr = ctr;
    g.seed = seed;
    g.epoch_local = epoch_ctr;  // with ctr: the epoch index inside this update call
    g.n_epochs = (uint32_t)cfg->n_epochs;
    // two tiles side by side per workgroup (one partial row for both) once there are more tiles than CUs can take one
    // 8-wave workgroup each: below that, more (smaller) workgroups fill the chip better
    out->nt = g.num_tiles > 256 ? 2 : 1;
    out->nb = grad_blocks((g.num_tiles + out->nt - 1) / out->nt);
    out->ns = ns;
    out->np = np;
    // workspace carve: partials [MAX][np] | loss_partials [MAX][4] | sumsq (doubles) | counter
    g.partials = (float*)workspace;
    g.loss_partials = g.partials + (int64_t)MAX_GRAD_BLOCKS * np;
    uintptr_t q = (uintptr_t)(g.loss_partials + (int64_t)MAX_GRAD_BLOCKS * 4);
    q = (q + 15) & ~(uintptr_t)15;
    out->sumsq = (double*)q;
    out->counter = (unsigned int*)(out->sumsq + 4096);
    long long* dbgp = (long long*)(out->counter + 16);  // MAX_GRAD_BLOCKS x 8 words
    g.dbg = RLHIP_ENV_FLAG("RLHIP_GRAD_DEBUG") ? dbgp : nullptr;
    g.samples = nullptr;
    g.samples_out = nullptr;
    uintptr_t pq = (uintptr_t)(dbgp + (int64_t)MAX_GRAD_BLOCKS * 8);
    pq = (pq + 63) & ~(uintptr_t)63;
    out->packed = (float*)pq;
    g.packed = out->packed;
    return RLHIP_OK;
}

// Multi-GPU optimiser step after the gradient all-reduce: [grad_scale] -> global norm -> clip -> Adam -> unit-record
// refresh in ONE single-workgroup launch (np <= 16 k), so that the next rlhip_ppo_grad call needs no pack launch.
__global__ __launch_bounds__(1024) void apply_pack_kernel(float* __restrict__ grad, float grad_scale, int np,
                                                          float* __restrict__ gn_out, ApplyArgs ap) {
    __shared__ double l_d[16];
    constexpr int U = 16;
    float gr[U];
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = threadIdx.x + u * 1024;
        gr[u] = q < np ? grad[q] * grad_scale : 0.0f;
        acc += (double)gr[u] * (double)gr[u];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) l_d[wv] = acc;
    const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
    __syncthreads();
    double tot = 0.0;
    for (int q = 0; q < 16; ++q) tot += l_d[q];
    const float gn = (float)sqrt(tot);
    const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = threadIdx.x + u * 1024;
        if (q >= np) continue;
        float g1 = gr[u];
        if (scale != 1.0f) g1 *= scale;
        const float mi = ap.b1 * ap.m[q] + (1.0f - ap.b1) * g1;  // Optimisers.Adam, expression order of optim.hip adam1
        const float vi = ap.b2 * ap.v[q] + (1.0f - ap.b2) * (g1 * g1);
        ap.m[q] = mi;
        ap.v[q] = vi;
        const float d = mi / c1 / (sqrtf(vi / c2) + ap.eps) * ap.lr;
        ap.params[q] = ap.params[q] - d;
        grad[q] = g1;
    }
    __syncthreads();
    pack_records(ap.params, ap.packed, ap.h, ap.ns, ap.nout, ap.np_a, threadIdx.x, blockDim.x);
    if (threadIdx.x == 0) {
        if (gn_out) gn_out[0] = gn;
        ap.beta_pow[0] *= ap.b1;
        ap.beta_pow[1] *= ap.b2;
    }
}

static int32_t launch_grad(const GradLaunch& L, hipStream_t s) {
    // two teams: 2 x 35 KB of tile areas + the 20 KB record copy = 90 KB of dynamic LDS (above the 64 KB default: opt-in per
    // kernel and device)
#define LAUNCH_GK(NS_, ACT_, NO_, NT_)                                                                                 \
    do {                                                                                                              \
        static unsigned long long done_ = 0;                                                                          \
        const size_t smem_ = grad_smem_bytes(NT_);                                                                    \
        int32_t rc_ = allow_big_lds(ppo_grad_kernel<NS_, ACT_, NO_, NT_>, smem_, &done_);                             \
        if (rc_) return rc_;                                                                                          \
        hipLaunchKernelGGL((ppo_grad_kernel<NS_, ACT_, NO_, NT_>), dim3(L.nb), dim3(64 * NW * NT_), smem_, s, L.g);   \
    } while (0)
#define LAUNCH_G(NS_, ACT_)                                                                                            \
    do {                                                                                                              \
        if (L.nt == 2) {                                                                                              \
            if (L.g.pd.nout_a > 2) LAUNCH_GK(NS_, ACT_, 3, 2);                                                        \
            else LAUNCH_GK(NS_, ACT_, 2, 2);                                                                          \
        } else if (L.g.pd.nout_a > 2) LAUNCH_GK(NS_, ACT_, 3, 1);                                                     \
        else LAUNCH_GK(NS_, ACT_, 2, 1);                                                                              \
    } while (0)
    const int a = L.g.pd.act;
    if (L.ns == 4) { if (a == 0) LAUNCH_G(4, 0); else LAUNCH_G(4, 1); }
    else if (L.ns == 3) { if (a == 0) LAUNCH_G(3, 0); else LAUNCH_G(3, 1); }
    else { if (a == 0) LAUNCH_G(2, 0); else LAUNCH_G(2, 1); }
#undef LAUNCH_G
#undef LAUNCH_GK
    return RLHIP_OK;
}

static void launch_pack(const GradLaunch& L, hipStream_t s) {
    hipLaunchKernelGGL(pack_params_kernel, dim3(1), dim3(256), 0, s, L.g.params, L.packed, L.g.pd.h, L.ns,
                       L.g.pd.nout_a, L.g.pd.np_a);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

// bytes of the carve of prepare_grad (256-byte aligned); behind it the sample records of an update call
// (written by the first gradient launch of the call): 32 B per trajectory entry, up to 2^24 entries (512 MB); beyond that the steps gather from the planes
static int64_t sample_record_bytes(int64_t n, int64_t T) {
    const int64_t total = n * T;
    return (total >= 1 && total <= ((int64_t)1 << 24)) ? 32 * total : 0;
}
static int64_t grad_workspace_bytes(int64_t np) {
    const int64_t b = (int64_t)MAX_GRAD_BLOCKS * (np + 4) * (int64_t)sizeof(float) + 16 + 4096 * (int64_t)sizeof(double) + 64 +
                      (int64_t)MAX_GRAD_BLOCKS * 8 * (int64_t)sizeof(long long) + 64 + (16 * 256 + 8) * (int64_t)sizeof(float);
    return (b + 255) / 256 * 256;
}

// where the sample records of an update call live in the learner's workspace (NULL: too many entries, gather from the planes)
static float4* update_samples_ptr(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, void* workspace) {
    const int64_t np_ = rlhip_ppo_nparams(kind, cfg);
    if (sample_record_bytes(n, T) <= 0 || np_ <= 0) return nullptr;
    return (float4*)((char*)workspace + grad_workspace_bytes(np_));
}
int32_t rlhip_ppo_workspace_init(void* workspace, int64_t bytes, rlhip_stream_t stream) {
    RLHIP_REQUIRE(workspace != nullptr && bytes > 0, "bad arguments");
    RLHIP_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)bytes, as_stream(stream)));
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    ws_table()[workspace] = bytes;
    return RLHIP_OK;
}

int32_t rlhip_ppo_workspace_release(void* workspace) {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    ws_table().erase(workspace);
    return RLHIP_OK;
}

int64_t rlhip_ppo_workspace_bytes(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T) {
    if (is_layers3(cfg)) return ppo3_workspace_bytes(kind, cfg, n, T);
    int64_t np = rlhip_ppo_nparams(kind, cfg);
    if (np < 0) return -1;
    return grad_workspace_bytes(np) + sample_record_bytes(n, T);
}


static int32_t grad_entry(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                          const float* params, uint64_t seed, uint32_t epoch_ctr, const uint32_t* ctr, int32_t mb,
                          void* workspace, float* grad_out, float* losses_out, rlhip_stream_t stream,
                          bool do_pack = true) {
    RLHIP_REQUIRE(grad_out != nullptr, "grad_out is NULL");
    if (is_layers3(cfg)) {
        RLHIP_REQUIRE(ctr == nullptr, "layers = 3: the device-counter (graph replay) variant is not built");
        if (int32_t rcw = ws_check(kind, cfg, n, T, workspace)) return rcw;
        return ppo3_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_out, losses_out, true, stream);
    }
    GradLaunch L;
    int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, &L, ctr);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    if (do_pack) launch_pack(L, s);
    if ((rc = launch_grad(L, s))) return rc;
    ApplyArgs ap{};
    hipLaunchKernelGGL((reduce_apply_kernel<APPLY_NONE>), dim3((int)((L.np + RP - 1) / RP)), dim3(1024), 0, s, L.g.partials,
                       L.g.loss_partials, L.nb, (int)L.np, grad_out, losses_out, L.g.wa, L.g.wc, L.g.we, L.g.inv_b,
                       ap, XchgArgs{});
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ppo_grad_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                           const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr,
                           int32_t mb, void* workspace, float* grad_out, float* losses_out,
                           rlhip_stream_t stream) {
    return grad_entry(kind, cfg, n, T, traj, params, seed, epoch_ctr, nullptr, mb, workspace, grad_out, losses_out,
                      stream);
}

int32_t rlhip_ppo_grad_fresh_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                                 const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr,
                                 int32_t mb, void* workspace, float* grad_out, float* losses_out,
                                 rlhip_stream_t stream) {
    return grad_entry(kind, cfg, n, T, traj, params, seed, epoch_ctr, nullptr, mb, workspace, grad_out, losses_out, stream,
                      /*do_pack=*/false);
}

int32_t rlhip_ppo_apply_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, float* params, float* grad,
                            float* m, float* v, float* beta_pow, float grad_scale, void* workspace, float* gn_out,
                            rlhip_stream_t stream) {
    RLHIP_REQUIRE(cfg && params && grad && m && v && beta_pow && workspace, "NULL argument");
    const int64_t np = rlhip_ppo_nparams(kind, cfg);
    RLHIP_REQUIRE(np > 0, "bad configuration");
    if (is_layers3(cfg) || np > 16 * 1024)  // generic path: fused clip + Adam; the next grad call re-packs
        return rlhip_clip_adam_f32(params, grad, m, v, beta_pow, np, grad_scale, cfg->max_grad_norm, cfg->lr, cfg->beta1,
                                   cfg->beta2, cfg->adam_eps, gn_out, stream);
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
    if ((rc = ws_check(kind, cfg, n, T, workspace))) return rc;
    const int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    ApplyArgs ap{params, m, v, beta_pow, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2, cfg->adam_eps,
                 nullptr, nullptr, workspace_packed(workspace, np), pd.h, ns, pd.nout_a, pd.np_a};
    hipLaunchKernelGGL(apply_pack_kernel, dim3(1), dim3(1024), 0, as_stream(stream), grad, grad_scale, (int)np, gn_out,
                       ap);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

extern "C" int32_t rlhip_p2p_allreduce_f32(float* data, int64_t n, int64_t cap, int32_t rank, int32_t world,
                                           void* const* comm_bufs_host, uint32_t seq, int64_t timeout_polls,
                                           int32_t* status_dev, rlhip_stream_t stream);

/* optimise!(policy) of a SHARDED policy in one call: n_epochs x n_microbatches of { gradient of this rank's shard ->
 * one-shot peer-to-peer SUM over the ranks (p2p.hip) -> clip (1 / world scale) + Adam + record refresh }, 4 launches per
 * optimiser step on one stream, no host work in between.  seq0 = the last sequence number this rank used (the call
 * consumes seq0 + 1 .. seq0 + n_epochs * n_microbatches). */
int32_t rlhip_ppo_update_p2p_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                                 const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow,
                                 uint64_t seed, uint32_t update_ctr, void* workspace, float* grad_scratch,
                                 float* losses_out, int32_t rank, int32_t world, void* const* comm_bufs_host,
                                 int64_t comm_cap, uint32_t seq0, int64_t timeout_polls, int32_t* status_dev,
                                 rlhip_stream_t stream) {
    RLHIP_REQUIRE(cfg && params && m && v && beta_pow && grad_scratch && comm_bufs_host && status_dev, "NULL argument");
    RLHIP_REQUIRE(world >= 1, "bad world size");
    const int64_t np = rlhip_ppo_nparams(kind, cfg);
    RLHIP_REQUIRE(np > 0 && np <= comm_cap, "the gradient does not fit the comm buffer");
    uint32_t seq = seq0;
    bool first = true;
    if (!is_layers3(cfg)) {
        // every argument check of the loop below, once, BEFORE the first exchange is enqueued: an invalid call then fails
        // on every rank alike without consuming a sequence number (a failure after k exchanges would leave this
        // rank's sequence behind its peers')
        GradLaunch L;
        int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, update_ctr * (uint32_t)cfg->n_epochs, 0, workspace, &L,
                                  nullptr);
        if (rc) return rc;
        for (int q = 0; q < world; ++q) RLHIP_REQUIRE(comm_bufs_host[q] != nullptr, "peer buffer is NULL");
    }
    const int rblocks = (int)((np + RP - 1) / RP);
    float4* samples = is_layers3(cfg) ? nullptr : update_samples_ptr(kind, cfg, n, T, workspace);
    const bool fused = !is_layers3(cfg) && rblocks <= grid_apply_max_blocks<APPLY_XCHG>() && world <= 16 &&
                       comm_cap <= (1 << 24) && !RLHIP_ENV_FLAG("RLHIP_P2P_UNFUSED");
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        const uint32_t epoch_ctr = update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            if (fused) {
                // grad kernel, then ONE kernel: partial sums -> peer exchange -> clip -> Adam -> record patch
                GradLaunch L;
                int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, &L, nullptr);
                if (rc) return rc;
                hipStream_t s = as_stream(stream);
                L.g.samples = samples;
                if (first) {  // no pack launch: see update_entry
                    L.g.packed = nullptr;
                    L.g.samples = nullptr;
                    L.g.samples_out = samples;
                }
                first = false;
                if ((rc = launch_grad(L, s))) return rc;
                ApplyArgs ap{params, m, v, beta_pow, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2, cfg->adam_eps,
                             L.counter, L.sumsq, L.packed, L.g.pd.h, L.ns, L.g.pd.nout_a, L.g.pd.np_a};
                XchgArgs xa{};
                for (int q = 0; q < world; ++q) {
                    RLHIP_REQUIRE(comm_bufs_host[q] != nullptr, "peer buffer is NULL");
                    xa.slot[q] = (float*)comm_bufs_host[q];
                    xa.flags[q] = (unsigned int*)((float*)comm_bufs_host[q] + 2 * comm_cap);
                }
                xa.rank = rank;
                xa.world = world;
                xa.cap = (int)comm_cap;
                xa.seq = ++seq;
                xa.timeout_polls = timeout_polls;
                xa.status = status_dev;
                xa.inv_world = 1.0f / (float)world;
                hipLaunchKernelGGL((reduce_apply_kernel<APPLY_XCHG>), dim3(rblocks), dim3(1024), 0, s, L.g.partials,
                                   L.g.loss_partials, L.nb, (int)L.np, grad_scratch, losses_out, L.g.wa, L.g.wc, L.g.we,
                                   L.g.inv_b, ap, xa);
                RLHIP_LAUNCH_CHECK();
                continue;
            }
            int32_t rc = grad_entry(kind, cfg, n, T, traj, params, seed, epoch_ctr, nullptr, mb, workspace, grad_scratch,
                                    losses_out, stream, /*do_pack=*/first);
            if (rc) return rc;
            first = false;
            rc = rlhip_p2p_allreduce_f32(grad_scratch, np, comm_cap, rank, world, comm_bufs_host, ++seq, timeout_polls,
                                         status_dev, stream);
            if (rc) return rc;
            rc = rlhip_ppo_apply_f32(kind, cfg, n, T, params, grad_scratch, m, v, beta_pow, 1.0f / (float)world, workspace,
                                     nullptr, stream);
            if (rc) return rc;
        }
    }
    return RLHIP_OK;
}

/* optimise!(policy) of a sharded policy through a communicator (comm.hip): the fused peer-to-peer kernels when every
 * rank validated that path, otherwise gradient -> rlhip_allreduce_grads (ncclAllReduce on this stream) -> apply */
int32_t rlhip_ppo_update_comm_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                                  const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow,
                                  uint64_t seed, uint32_t update_ctr, void* workspace, float* grad_scratch,
                                  float* losses_out, rlhip_comm_t comm, rlhip_stream_t stream) {
    rlhip_comm_desc d;
    int32_t rc = rlhip_comm_info(comm, &d);
    if (rc) return rc;
    RLHIP_REQUIRE(cfg != nullptr, "cfg is NULL");
    if (d.world == 1 && !d.rccl_active)
        return rlhip_ppo_update_f32(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, update_ctr, workspace, grad_scratch,
                                    losses_out, stream);
    const int64_t np = rlhip_ppo_nparams(kind, cfg);
    RLHIP_REQUIRE(np > 0, "bad configuration");
    const uint32_t n_steps = (uint32_t)cfg->n_epochs * (uint32_t)cfg->n_microbatches;
    if (d.p2p_active && np <= d.cap) {
        rc = rlhip_ppo_update_p2p_f32(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, update_ctr, workspace,
                                      grad_scratch, losses_out, d.rank, d.world, d.bufs, d.cap, d.seq, d.timeout_polls,
                                      d.status, stream);
        // the call validates everything before its first exchange; whatever fails later (a launch error) fails with
        // exchanges already enqueued on the peers: keep this rank's sequence in step with theirs either way
        const int32_t rc2 = (rc == RLHIP_EINVAL) ? RLHIP_OK : rlhip_comm_advance_seq(comm, n_steps);
        return rc ? rc : rc2;
    }
    bool first = true;
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        const uint32_t epoch_ctr = update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            rc = grad_entry(kind, cfg, n, T, traj, params, seed, epoch_ctr, nullptr, mb, workspace, grad_scratch, losses_out,
                            stream, /*do_pack=*/first);
            if (rc) return rc;
            first = false;
            rc = rlhip_allreduce_grads(comm, grad_scratch, np, stream);
            if (rc) return rc;
            rc = rlhip_ppo_apply_f32(kind, cfg, n, T, params, grad_scratch, m, v, beta_pow, 1.0f / (float)d.world, workspace,
                                     nullptr, stream);
            if (rc) return rc;
        }
    }
    return RLHIP_OK;
}

int32_t rlhip_ppo_grad_dc_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                              const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_local,
                              const uint32_t* counters, int32_t mb, void* workspace, float* grad_out,
                              float* losses_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(counters != nullptr, "counters is NULL");
    return grad_entry(kind, cfg, n, T, traj, params, seed, epoch_local, counters, mb, workspace, grad_out, losses_out,
                      stream);
}

static int32_t update_entry(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                            const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow,
                            uint64_t seed, uint32_t update_ctr, const uint32_t* ctr, void* workspace,
                            float* grad_scratch, float* losses_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(cfg && params && m && v && beta_pow && grad_scratch, "NULL argument");
    if (is_layers3(cfg)) {
        RLHIP_REQUIRE(ctr == nullptr, "layers = 3: the device-counter (graph replay) variant is not built");
        if (int32_t rcw = ws_check(kind, cfg, n, T, workspace)) return rcw;
        return ppo3_update(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, update_ctr, workspace, grad_scratch,
                           losses_out, stream);
    }
    hipStream_t s = as_stream(stream);
    float4* samples = update_samples_ptr(kind, cfg, n, T, workspace);
    bool first = true;
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        uint32_t epoch_ctr = ctr ? (uint32_t)e : update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            GradLaunch L;
            int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, &L, ctr);
            if (rc) return rc;
            if (samples) L.g.samples = samples;
            if (first) {
                // Round 5: NO pack launch.  The first gradient launch of a call builds its unit records from the parameter
                // vector (the packed image may be stale: the host may have written `params`), gathers its samples from the
                // trajectory planes and -- with the threads of the team that does not write the partial row -- writes the 32-byte
                // sample records the other 15 steps read; the Adam tail of every step patches the unit-record image for the
                // step after it.  Same values on both routes, hence the same bits (tests/test_gpu_learners.py compares update_
                // with the grad_ / apply_ sequence; same-box A / B against the round-4 form with its pack launch: 0.4012 ->
                // 0.3970 ms per iteration, profiles/raw_r05/pack_launch_ab.txt).
                // The arrival counter needs no per-call memset: the workspace is zero-initialised by its owner (ABI contract)
                // and the last-arriving workgroup re-arms it in-kernel.
                L.g.packed = nullptr;
                L.g.samples = nullptr;
                L.g.samples_out = samples;
                first = false;
            }
            if ((rc = launch_grad(L, s))) return rc;
            ApplyArgs ap{params, m, v, beta_pow, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2, cfg->adam_eps,
                         L.counter, L.sumsq, L.packed, L.g.pd.h, L.ns, L.g.pd.nout_a, L.g.pd.np_a,
                         L.g.dbg ? L.g.dbg + 256 * 8 : nullptr};
            const int rblocks = (int)((L.np + RP - 1) / RP);
            if (rblocks <= grid_apply_max_blocks<APPLY_GRID>())
                hipLaunchKernelGGL((reduce_apply_kernel<APPLY_GRID>), dim3(rblocks), dim3(1024), 0, s, L.g.partials,
                                   L.g.loss_partials, L.nb, (int)L.np, grad_scratch, losses_out, L.g.wa, L.g.wc, L.g.we,
                                   L.g.inv_b, ap, XchgArgs{});
            else
                hipLaunchKernelGGL((reduce_apply_kernel<APPLY_LAST>), dim3(rblocks), dim3(1024), 0, s, L.g.partials,
                                   L.g.loss_partials, L.nb, (int)L.np, grad_scratch, losses_out, L.g.wa, L.g.wc, L.g.we,
                                   L.g.inv_b, ap, XchgArgs{});
        }
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_ppo_update_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                             const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow,
                             uint64_t seed, uint32_t update_ctr, void* workspace, float* grad_scratch,
                             float* losses_out, rlhip_stream_t stream) {
    return update_entry(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, update_ctr, nullptr, workspace,
                        grad_scratch, losses_out, stream);
}

int32_t rlhip_ppo_update_dc_f32(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T,
                                const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow,
                                uint64_t seed, const uint32_t* counters, void* workspace, float* grad_scratch,
                                float* losses_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(counters != nullptr, "counters is NULL");
    return update_entry(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, 0, counters, workspace, grad_scratch,
                        losses_out, stream);
}

}  // extern "C"
