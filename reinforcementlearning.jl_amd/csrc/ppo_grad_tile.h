// ppo_grad_tile.h -- the 64-sample tile of the two-layer PPO learner (loss + gradient), shared by the per-micro-batch
// gradient kernel (ppo_grad.hip: ppo_grad_kernel) and the persistent whole-update kernel (ppo_persist.hip).  One source
// for both, so that a partial gradient row is the same sequence of f32 operations -- the same bits -- on either path.
//
// Replaces the body of the removed Zoo `PPOPolicy` update (hyper-parameters blog
// a_practical_introduction_to_RL.jl/index.html:15257-15278): shuffled micro-batch gather, actor / critic forward (Flux
// Dense chains), softmax / ratio / clamp / min, value and entropy terms, Zygote backward.
//
// A team = 8 waves walking 64-sample tiles:
//   phase 0   64 threads fetch the tile's samples f = perm(pos) (keyed bijection, no index array) from the trajectory
//             into registers ONE TILE AHEAD and publish them to a double-buffered LDS tile.
//   phase 1a  lane = sample, wave w walks its eighth of the hidden units; the unit records (below) are wave-uniform and
//             come through the scalar cache (s_load -> SGPR operands of the FMAs: broadcast LDS reads of the records
//             would cost more LDS clocks than the unit costs VALU clocks).
//   phase 1b  wave 0 finishes logits / value, evaluates the loss terms and dL/d(outputs) per sample.
//   phase 2   lane = hidden unit j (weights in registers); the 64 samples stream from LDS as two broadcast b128 reads
//             each; weight gradients accumulate in registers: no atomics, no cross-lane reductions, fixed order.
//
// Packed f32: actor and critic walk the same (sample, unit) pairs with the same operation sequence, so every actor /
// critic pair of FMAs is ONE v_pk_fma_f32 on a register pair (actor in the low half, critic in the high half): the unit
// record interleaves the two nets -- {(w1a0, w1c0) .. (w1a3, w1c3), (b1a, b1c), (w2a0, w2c), w2a1, w2a2, 0, 0}, 16 floats
// -- so that a pair is one aligned SGPR pair (phase 1a) or VGPR pair (phase 2), broadcast operands (x_k, dL) need no
// move (op_sel), and each half is the IEEE operation the scalar code did: the same bits.  Measured on gfx950 at 4 waves
// per SIMD (tools/micro/valu_pk.hip): v_fma_f32 87 TFLOP/s, v_pk_fma_f32 114-129 TFLOP/s -- a packed FMA costs ~1.4
// scalar ones, not 2.
#pragma once
#include "ppo_common.h"

namespace rlhip {

constexpr float LOG2PI_F = 1.8378770664093453f;  // log(2f0 * pi) as Float32 (RLCore/utils/distributions.jl:9)
constexpr int TILE = 64;
constexpr int MAX_GRAD_BLOCKS = 512;
constexpr int NW = 8;     // waves per team (per 64-sample tile)
constexpr int GMAXO = 3;  // actor outputs handled by the fused gradient kernel (na <= 3, or (mu, log sigma))

struct GradArgs {
    const float* obs;
    const float* logp;
    const float* adv;
    const float* ret;
    const float* action_f;
    const int32_t* action_i;
    const float* params;
    const float* packed;   // unit records [h][16] (layout above) | {b2a0, b2a1, b2a2, b2c}
    float* partials;       // [nb][np]
    float* loss_partials;  // [nb][4]
    int64_t n;
    uint32_t total, bm, pos0;
    int num_tiles, np;
    PolicyDesc pd;
    float lo, hi, wa, wc, we, inv_b, min_logp;
    PermKeys pk;           // epoch permutation keys, evaluated on the host (2 Philox blocks) ...
    const uint32_t* ctr;   // ... or, when non-NULL, in the kernel from the device update counter ctr[1]:
    uint64_t seed;         //     epoch = epoch_local + ctr[1] * n_epochs  (HIP-graph replayable)
    uint32_t epoch_local, n_epochs;
};

struct TileRegs {  // one sample's trajectory entries, held in registers one tile ahead
    float4 x;
    float4 misc;  // {logp_old, adv, ret, action (int bits or float)}
};

// pos0: first position of the micro-batch in the epoch's permutation
template <int NS>
__device__ __forceinline__ TileRegs fetch_sample(const GradArgs& g, const PermKeys& pk, uint32_t pos0, int tile, int s) {
    uint32_t q = (uint32_t)tile * TILE + (uint32_t)s;
    bool valid = q < g.bm;
    uint32_t f = permute(pk, pos0 + (valid ? q : 0u));
    uint32_t t = f / (uint32_t)g.n, i = f - t * (uint32_t)g.n;
    TileRegs r;
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NS; ++k) xv[k] = g.obs[((int64_t)t * NS + k) * g.n + i];
    r.x = make_float4(xv[0], xv[1], xv[2], xv[3]);
    float a = g.pd.cont ? g.action_f[f] : __int_as_float(g.action_i[f]);
    r.misc = make_float4(g.logp[f], valid ? g.adv[f] : 0.0f, g.ret[f], a);
    return r;
}

// LDS of one team of 8 waves: x[2][TILE] | misc[2][TILE] | part[8][TILE] | dL[TILE] (float4) | comb[14][256] + 16 scalars (float)
__host__ __device__ constexpr size_t grad_team_smem_bytes() {
    return sizeof(float4) * (size_t)(2 * TILE + 2 * TILE + NW * TILE + TILE) + sizeof(float) * (14 * 256 + 16);
}
constexpr int GRAD_COMB_FLOATS = 14 * 256 + 16;  // the comb area + its 16 scalars, contiguous

struct TeamLds {
    float4* x;     // [2][TILE]
    float4* misc;  // [2][TILE]
    float4* part;  // [8][TILE]  {a0, a1, a2, v} partial sums
    float4* dL;    // [TILE]     {dl0, dv, dl1, dl2}
    float* comb;   // [14][256] second-half accumulators, then 16 scalars of the team's wave 0
};
__device__ __forceinline__ TeamLds team_lds(char* smem, int team) {
    char* tsm = smem + (size_t)team * grad_team_smem_bytes();
    TeamLds L;
    L.x = reinterpret_cast<float4*>(tsm);
    L.misc = L.x + 2 * TILE;
    L.part = L.misc + 2 * TILE;
    L.dL = L.part + NW * TILE;
    L.comb = reinterpret_cast<float*>(L.dL + TILE);
    return L;
}

typedef float f2 __attribute__((ext_vector_type(2)));  // (actor, critic)
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
constexpr int REC = 16;  // floats per unit record

struct UnitW {  // this thread's hidden unit (phase 2), both nets
    f2 w1[4], b1, w2p;  // w2p = (W2a[0, j], W2c[j])
    float w2a1, w2a2;
};
struct UnitG {  // its gradient accumulators, same pairing
    f2 w1[4], b1, w2p;
    float w2a1, w2a2;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 4; ++k) w1[k] = f2{0.f, 0.f};
        b1 = w2p = f2{0.f, 0.f};
        w2a1 = w2a2 = 0.f;
    }
};
// a unit's weights from its record (any pointer type)
template <class P>
__device__ __forceinline__ UnitW unit_from_record(P r) {
    UnitW W;
#pragma unroll
    for (int k = 0; k < 4; ++k) W.w1[k] = f2{r[2 * k], r[2 * k + 1]};
    W.b1 = f2{r[8], r[9]};
    W.w2p = f2{r[10], r[11]};
    W.w2a1 = r[12];
    W.w2a2 = r[13];
    return W;
}
struct HeadG {  // wave 0 of a team, lane = sample: output-bias gradients and loss sums
    float b2a[GMAXO];
    float b2c, s_actor, s_critic, s_ent;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int o = 0; o < GMAXO; ++o) b2a[o] = 0.f;
        b2c = s_actor = s_critic = s_ent = 0.f;
    }
};

// thread roles inside a team of 512 threads
struct TeamIds {
    int team, tid, lane, w, uidx, shalf;
    bool owner;
};
template <int NT>
__device__ __forceinline__ TeamIds team_ids(int h) {
    TeamIds t;
    const int gtid = threadIdx.x;
    t.team = NT > 1 ? __builtin_amdgcn_readfirstlane(gtid >> 9) : 0;
    t.tid = gtid & 511;
    t.lane = t.tid & 63;
    t.w = __builtin_amdgcn_readfirstlane(t.tid >> 6);
    t.uidx = t.tid & 255;  // phase 2: hidden unit of this thread
    t.shalf = t.w >> 2;    // phase 2: which half of the tile's samples (0: 0..31, 1: 32..63)
    t.owner = t.uidx < h;
    return t;
}

// Publish the team's first tile into LDS buffer 0 (threads tid < TILE hold it in `first` when have_first); a team
// without a tile gets finite operands, so that its (all-zero-weight) sums stay exact zeros.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void publish_first_tile(const TeamLds& L, const TeamIds& id, bool have_first,
                                                   const TileRegs& first) {
    if (have_first) {
        L.x[id.tid] = first.x;
        L.misc[id.tid] = first.misc;
    } else if (NT > 1 && id.tid < TILE) {
        L.x[id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        L.misc[id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        L.x[TILE + id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        L.misc[TILE + id.tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
}

// The tile loop of one workgroup (NT teams side by side, same trip count: the barriers are workgroup-wide; a team
// without a tile left computes on stale LDS with all-invalid samples: dL = 0, nothing is accumulated).
// RecP: pointer type of the unit records -- `const float*` to memory the kernel never writes (the compiler then proves
// the loads scalar), or a constant-address-space pointer (ppo_persist.hip).  b2: {b2a0, b2a1, b2a2, b2c}.
// NO = 2: the actor has at most two outputs (two actions, or (mu, log sigma)) -- the third output's FMAs (zero weights,
// zero dL/dout: exact no-ops) are not issued; NO = 3: three actions.
template <int NS, int ACT, int NO, int NT, class RecP>
__device__ __forceinline__ void grad_tile_loop(const GradArgs& g, const PermKeys& pk, uint32_t pos0, const TeamLds& L,
                                               const TeamIds& id, RecP rec, const float (&b2)[4], const UnitW& W,
                                               UnitG& G, HeadG& Hd) {
    const int h = g.pd.h;
    const int hq = h / NW;  // hidden units per wave in phase 1a
    const int lane = id.lane, w = id.w;
    int tile = blockIdx.x * NT + id.team;
    int buf = 0;
    for (int base = blockIdx.x * NT; base < g.num_tiles; base += gridDim.x * NT, tile += gridDim.x * NT) {
        const int next = tile + gridDim.x * NT;
        const float4* cx = L.x + buf * TILE;
        const float4* cm = L.misc + buf * TILE;
        // ---- phase 0 (next tile): wave 1 issues the gather now, publishes it after phase 2 ----
        TileRegs pre;
        const bool prefetcher = (w == 1) && (next < g.num_tiles);
        if (prefetcher) pre = fetch_sample<NS>(g, pk, pos0, next, lane);
        // ---- phase 1a: lane = sample, wave w walks hidden units [w*hq, (w+1)*hq) ----
        {
            const float4 xv = cx[lane];
            f2 accp = {0.f, 0.f};  // (actor output 0, value)
            float acc1 = 0.f, acc2 = 0.f;
#pragma unroll 8
            for (int jj = w * hq; jj < (w + 1) * hq; ++jj) {
                // wave-uniform addresses: scalar loads; the pairs are SGPR-pair operands of the packed FMAs below
                const RecP r = rec + REC * jj;
                f2 z = {r[8], r[9]};
                z = pk_fma(f2{r[0], r[1]}, splat(xv.x), z);
                if (NS > 1) z = pk_fma(f2{r[2], r[3]}, splat(xv.y), z);
                if (NS > 2) z = pk_fma(f2{r[4], r[5]}, splat(xv.z), z);
                if (NS > 3) z = pk_fma(f2{r[6], r[7]}, splat(xv.w), z);
                const f2 hh = {act_fwd_t<ACT>(z.x), act_fwd_t<ACT>(z.y)};
                accp = pk_fma(f2{r[10], r[11]}, hh, accp);
                acc1 = fmaf(r[12], hh.x, acc1);
                if (NO > 2) acc2 = fmaf(r[13], hh.x, acc2);
            }
            L.part[w * TILE + lane] = make_float4(accp.x, acc1, acc2, accp.y);
        }
        __syncthreads();
        // ---- phase 1b: wave 0 finishes the forward, evaluates the loss and dL/d(outputs) ----
        if (w == 0) {
            const int s = lane;
            const bool valid = ((uint32_t)tile * TILE + (uint32_t)s) < g.bm;
            float4 ps = L.part[s];
#pragma unroll
            for (int q = 1; q < NW; ++q) {  // fixed summation order over the NW waves' partial sums
                const float4 pq = L.part[q * TILE + s];
                ps.x += pq.x;
                ps.y += pq.y;
                ps.z += pq.z;
                ps.w += pq.w;
            }
            float oa[GMAXO], dl[GMAXO] = {0.f, 0.f, 0.f};
            oa[0] = ps.x + b2[0];
            oa[1] = ps.y + b2[1];
            oa[2] = ps.z + b2[2];
            const float v = ps.w + b2[3];
            const float4 mi = cm[s];
            const float lp_old = fmaxf(mi.x, g.min_logp);  // clamp!(log_p, log(1e-8), Inf)
            const float A = mi.y;
            float lp_new, ent;
            if (!g.pd.cont) {
                const int na = g.pd.na;
                float mx = oa[0];
                for (int k = 1; k < na; ++k) mx = fmaxf(mx, oa[k]);
                float se = 0.f;
                for (int k = 0; k < na; ++k) se += expf(oa[k] - mx);
                const float lse = logf(se);
                float logp[GMAXO], pr[GMAXO];
                ent = 0.f;
                for (int k = 0; k < na; ++k) {
                    logp[k] = (oa[k] - mx) - lse;
                    pr[k] = expf(logp[k]);
                    ent -= pr[k] * logp[k];
                }
                const int a = __float_as_int(mi.w);
                lp_new = 0.f;
                for (int k = 0; k < na; ++k)
                    if (k == a) lp_new = logp[k];
                const float ratio = expf(lp_new - lp_old);
                const float surr1 = ratio * A;
                const float rc = fminf(fmaxf(ratio, g.lo), g.hi);
                const float surr2 = rc * A;
                const bool inside = ratio >= g.lo && ratio <= g.hi;
                const float dobj = (inside || surr1 < surr2) ? A : 0.f;
                const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
                if (valid) Hd.s_actor += fminf(surr1, surr2);
                for (int k = 0; k < na; ++k) {
                    const float dlp = ((k == a) ? 1.f : 0.f) - pr[k];
                    const float dent = -pr[k] * (logp[k] + ent);
                    dl[k] = dL_dlp * dlp - g.we * g.inv_b * dent;
                }
            } else {
                const float eps = 1.0e-8f;
                const float mu = oa[0], ls = oa[1];
                const float sg = expf(ls);
                const float z = mi.w;
                const float se = sg + eps;
                const float zz = (z - mu) / se;
                lp_new = -(zz * zz + LOG2PI_F) / 2.0f - logf(se);
                ent = ((LOG2PI_F + 1.0f) + ls) / 2.0f;
                const float dmu = (z - mu) / (se * se);
                const float dls = ((z - mu) * (z - mu) / (se * se * se) - 1.0f / se) * sg;
                const float ratio = expf(lp_new - lp_old);
                const float surr1 = ratio * A;
                const float rc = fminf(fmaxf(ratio, g.lo), g.hi);
                const float surr2 = rc * A;
                const bool inside = ratio >= g.lo && ratio <= g.hi;
                const float dobj = (inside || surr1 < surr2) ? A : 0.f;
                const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
                if (valid) Hd.s_actor += fminf(surr1, surr2);
                dl[0] = dL_dlp * dmu;
                dl[1] = dL_dlp * dls - g.we * g.inv_b * 0.5f;
            }
            const float dv = mi.z - v;
            float dvout = -2.0f * g.wc * g.inv_b * dv;
            if (valid) {
                Hd.s_critic += dv * dv;
                Hd.s_ent += ent;
            } else {
                dl[0] = dl[1] = dl[2] = 0.f;
                dvout = 0.f;
            }
            L.dL[s] = make_float4(dl[0], dvout, dl[1], dl[2]);  // (dl0, dv) is the pair phase 2 multiplies by (ha, hc)
            Hd.b2a[0] += dl[0];
            Hd.b2a[1] += dl[1];
            Hd.b2a[2] += dl[2];
            Hd.b2c += dvout;
        }
        __syncthreads();
        // ---- phase 2: lane = hidden unit j; the tile's samples stream from LDS (broadcast reads) ----
        if (id.owner) {
#pragma unroll 4
            for (int s = id.shalf * (TILE / 2); s < (id.shalf + 1) * (TILE / 2); ++s) {
                const float4 xv = cx[s];
                const float4 d = L.dL[s];  // {dl0, dv, dl1, dl2}
                f2 z = W.b1;
                z = pk_fma(W.w1[0], splat(xv.x), z);
                if (NS > 1) z = pk_fma(W.w1[1], splat(xv.y), z);
                if (NS > 2) z = pk_fma(W.w1[2], splat(xv.z), z);
                if (NS > 3) z = pk_fma(W.w1[3], splat(xv.w), z);
                const f2 hh = {act_fwd_t<ACT>(z.x), act_fwd_t<ACT>(z.y)};
                const f2 d0v = {d.x, d.y};
                G.w2p = pk_fma(d0v, hh, G.w2p);
                G.w2a1 = fmaf(d.z, hh.x, G.w2a1);
                if (NO > 2) G.w2a2 = fmaf(d.w, hh.x, G.w2a2);
                f2 dh = d0v * W.w2p;
                dh.x = fmaf(d.z, W.w2a1, dh.x);
                if (NO > 2) dh.x = fmaf(d.w, W.w2a2, dh.x);
                f2 dz;
                if (ACT == 0) {  // relu: dh * [z > 0] as a select (the product differs only in the sign of a zero)
                    dz = f2{z.x > 0.0f ? dh.x : 0.0f, z.y > 0.0f ? dh.y : 0.0f};
                } else {
                    dz = dh * (splat(1.0f) - hh * hh);
                }
                G.b1 += dz;
                G.w1[0] = pk_fma(dz, splat(xv.x), G.w1[0]);
                if (NS > 1) G.w1[1] = pk_fma(dz, splat(xv.y), G.w1[1]);
                if (NS > 2) G.w1[2] = pk_fma(dz, splat(xv.z), G.w1[2]);
                if (NS > 3) G.w1[3] = pk_fma(dz, splat(xv.w), G.w1[3]);
            }
        }
        // publish the prefetched next tile into the other buffer (nobody reads it before the barrier)
        if (prefetcher) {
            L.x[(buf ^ 1) * TILE + lane] = pre.x;
            L.misc[(buf ^ 1) * TILE + lane] = pre.misc;
        }
        __syncthreads();
        buf ^= 1;
    }
}

__device__ __forceinline__ void comb_store(float* c, const UnitG& G) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[(k)*256] = G.w1[k].x;
        c[(4 + k) * 256] = G.w1[k].y;
    }
    c[8 * 256] = G.b1.x;
    c[9 * 256] = G.b1.y;
    c[10 * 256] = G.w2p.x;
    c[11 * 256] = G.w2a1;
    c[12 * 256] = G.w2a2;
    c[13 * 256] = G.w2p.y;
}
__device__ __forceinline__ void comb_add(const float* c, UnitG& G) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        G.w1[k].x += c[(k)*256];
        G.w1[k].y += c[(4 + k) * 256];
    }
    G.b1.x += c[8 * 256];
    G.b1.y += c[9 * 256];
    G.w2p.x += c[10 * 256];
    G.w2a1 += c[11 * 256];
    G.w2a2 += c[12 * 256];
    G.w2p.y += c[13 * 256];
}

// Fold the workgroup's accumulators into ONE partial row, fixed order: waves 4..7 (second half of the samples) hand
// theirs to waves 0..3 through LDS, wave 0 sums its per-sample-lane accumulators over the 64 lanes, then team 1 hands
// its (half-combined) values to team 0.  Afterwards the threads with team == 0, shalf == 0 hold the row's unit
// gradients and every lane of team 0's wave 0 holds the head sums.  Contains workgroup barriers.
template <int NT>
__device__ __forceinline__ void grad_fold(char* smem, const TeamLds& L, const TeamIds& id, UnitG& G, HeadG& Hd) {
    if (id.shalf == 1) comb_store(L.comb + id.uidx, G);
    __syncthreads();
    if (id.shalf == 0) comb_add(L.comb + id.uidx, G);
    float* l_sc = L.comb + 14 * 256;  // [16] scalars of this team's wave 0
    if (id.w == 0) {  // wave 0: reduce the per-sample-lane accumulators over the 64 lanes
#pragma unroll
        for (int o = 0; o < GMAXO; ++o) Hd.b2a[o] = wave_sum_f32(Hd.b2a[o]);
        Hd.b2c = wave_sum_f32(Hd.b2c);
        Hd.s_actor = wave_sum_f32(Hd.s_actor);
        Hd.s_critic = wave_sum_f32(Hd.s_critic);
        Hd.s_ent = wave_sum_f32(Hd.s_ent);
    }
    if (NT > 1) {
        __syncthreads();  // the readers of the comb areas (above) are done
        if (id.team == 1) {
            if (id.shalf == 0) comb_store(L.comb + id.uidx, G);
            if (id.w == 0 && id.lane == 0) {
                l_sc[0] = Hd.b2a[0];
                l_sc[1] = Hd.b2a[1];
                l_sc[2] = Hd.b2a[2];
                l_sc[3] = Hd.b2c;
                l_sc[4] = Hd.s_actor;
                l_sc[5] = Hd.s_critic;
                l_sc[6] = Hd.s_ent;
            }
        }
        __syncthreads();
        if (id.team == 0) {
            const float* c1 = team_lds(smem, 1).comb;
            if (id.shalf == 0) comb_add(c1 + id.uidx, G);
            if (id.w == 0) {
                const float* sc1 = c1 + 14 * 256;
                Hd.b2a[0] += sc1[0];
                Hd.b2a[1] += sc1[1];
                Hd.b2a[2] += sc1[2];
                Hd.b2c += sc1[3];
                Hd.s_actor += sc1[4];
                Hd.s_critic += sc1[5];
                Hd.s_ent += sc1[6];
            }
        }
    }
}

// position of flat parameter q in the unit-record copy (see pack_records)
__device__ __forceinline__ int64_t record_slot(int64_t q, int h, int ns, int nout, int64_t np_a) {
    const int net = q >= np_a ? 1 : 0;
    const int64_t r = q - (net ? np_a : 0);
    const int no = net ? 1 : nout;
    if (r < (int64_t)h * ns) return REC * (r % h) + 2 * (r / h) + net;           // W1[j + h k] -> rec[j][2 k + net]
    if (r < (int64_t)h * ns + h) return REC * (r - (int64_t)h * ns) + 8 + net;   // b1[j]       -> rec[j][8 + net]
    const int64_t w = r - ((int64_t)h * ns + h);
    if (w < (int64_t)no * h) {
        if (net) return REC * w + 11;                                             // W2c[j]      -> rec[j][11]
        const int64_t j = w / no, o = w % no;
        return REC * j + (o == 0 ? 10 : 11 + o);                                  // W2a[o + no j] -> rec[j][10 | 12 | 13]
    }
    return (int64_t)REC * h + (net ? 3 : (w - (int64_t)no * h));                  // output biases -> tail
}

// unit records (layout at the top of this file) + the output biases; any thread count
__device__ __forceinline__ void pack_records(const float* __restrict__ params, float* __restrict__ packed, int h,
                                             int ns, int nout, int64_t np_a, int tid, int nthreads) {
    const float* W1a = params;
    const float* b1a = W1a + h * ns;
    const float* W2a = b1a + h;
    const float* b2a = W2a + nout * h;
    const float* W1c = params + np_a;
    const float* b1c = W1c + h * ns;
    const float* W2c = b1c + h;
    const float* b2c = W2c + h;
    for (int j = tid; j < h; j += nthreads) {
        float rec[REC];
#pragma unroll
        for (int k = 0; k < REC; ++k) rec[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < ns) {
                rec[2 * k] = W1a[j + h * k];
                rec[2 * k + 1] = W1c[j + h * k];
            }
        rec[8] = b1a[j];
        rec[9] = b1c[j];
        rec[10] = W2a[0 + nout * j];
        rec[11] = W2c[j];
        if (nout > 1) rec[12] = W2a[1 + nout * j];
        if (nout > 2) rec[13] = W2a[2 + nout * j];
        float4* dst = reinterpret_cast<float4*>(packed + (int64_t)REC * j);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3]);
    }
    if (tid == 0) {
        float* t = packed + (int64_t)REC * h;
        t[0] = (0 < nout) ? b2a[0] : 0.f;
        t[1] = (1 < nout) ? b2a[1] : 0.f;
        t[2] = (2 < nout) ? b2a[2] : 0.f;
        t[3] = b2c[0];
    }
}

// ---- what ppo_grad.hip and ppo_persist.hip share on the host ----
struct GradLaunch {
    GradArgs g;
    int nb, ns, nt;  // partial rows (= workgroups), observation size, teams (tiles side by side) per workgroup
    int64_t np;
    unsigned int* counter;
    double* sumsq;
    float* packed;
};

// the persistent whole-update kernel (ppo_persist.hip).  persist_bytes: what it adds to the learner's workspace;
// ppo_persist_update: enqueue all n_epochs x n_microbatches optimiser steps as ONE launch, or return +1 when the
// configuration / device does not admit it (the caller then takes the two-launch-per-step path), < 0 on error.
int64_t ppo_persist_bytes(int64_t np, int h);
int32_t ppo_persist_update(const GradLaunch& L0, const rlhip_ppo_cfg* cfg, float* params, float* m, float* v,
                           float* beta_pow, uint32_t update_ctr, void* persist_ws, float* grad_out, float* losses_out,
                           hipStream_t s);

}  // namespace rlhip
