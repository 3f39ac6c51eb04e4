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
//   phase 1a  layer 1 on the f32 MFMA: wave w owns 32 unit slots, D[unit][sample] = b1 + W1 x by two
//             v_mfma_f32_32x32x2_f32 per net and 32-sample half (the bias is the accumulator's initial value, so the
//             result IS the oracle's fmaf chain b1 + w0 x0 + w1 x1 + ..., bit for bit: tools/micro/mfma_f32_l1.hip).  A
//             lane ends up with two samples x 16 units of both nets; its units' bias / head weights come from the
//             workgroup's LDS copy of the unit records.  (Round 2 walked the records through the scalar cache, 8 records
//             per s_waitcnt: 4 exposed L2 round trips per wave made this phase 10 us of an 18 us tile.)
//   phase 1b  wave 0 finishes logits / value, evaluates the loss terms and dL/d(outputs) per sample.
//   phase 2   the transpose of phase 1a on the same operands: D[sample][unit] = b1 + W1 x (A = x, B = W1 rows, C = the
//             lane's bias): lane (r, kb) of wave w = unit slot 32 w + r with the z of 32 of the tile's samples (rows
//             (q & 3) + 8 (q >> 2) + 4 kb of each 32-sample half) in the accumulator registers.  The first half's MFMAs
//             are issued BEFORE the barrier of phase 1b (they need nothing from it and run in its shadow), the second
//             half's while the first half's rows are consumed.  Per row: activation, dW2 / dh / dz / db1 / dW1 on the
//             VALU (x and dL by 16-byte LDS reads, two addresses per instruction), accumulators in registers: 22
//             VALU instructions per (sample, unit) pair of nets instead of 30.  The two lanes of a unit are added once,
//             in grad_fold: no atomics, fixed order.
//
// The unit record interleaves the two nets -- {(w1a0, w1c0) .. (w1a3, w1c3), (b1a, b1c), (w2a0, w2c), w2a1, w2a2, 0, 0},
// 16 floats -- one 64-byte read per unit for phase 2's registers, 8-byte reads of a pair for phase 1a.  Round 3 (first
// half) issued every actor / critic pair as one v_pk_fma_f32 (a packed FMA costs ~1.4 scalar ones on gfx950:
// tools/micro/valu_pk.hip; -2.5 % on the iteration); with the MFMA in phase 1a the kernel falls under the repo's rule "no
// packed f32 VALU beside an MFMA" and the pairs are two scalar FMAs again -- the MFMA phase saves five times what the
// packing did (profiles/r03_tile_mfma.md).
#pragma once
#include "mfma_common.h"
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
    const float4* samples; // sample records {x0..x3}, {logp, adv, ret, action bits} per trajectory entry f = t n + i, or NULL:
                           // written once per update call (by its first gradient launch: samples_out); a shuffled sample is then ONE 32-byte
                           // read instead of eight 4-byte reads from eight planes (eight cache lines)
    float4* samples_out;   // first launch of an update call (round 5): the kernel gathers from the planes (samples == NULL) and,
                           // off its critical path, writes the sample records the next 15 steps read -- no pack launch
    long long* dbg;        // RLHIP_GRAD_DEBUG: [workgroup][8] s_memtime stamps of thread 0 (tools/grad_timeline.py), else NULL
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
    TileRegs r;
    if (g.samples) {
        r.x = g.samples[2 * (int64_t)f];
        r.misc = g.samples[2 * (int64_t)f + 1];
        if (!valid) r.misc.y = 0.0f;
        return r;
    }
    uint32_t t = f / (uint32_t)g.n, i = f - t * (uint32_t)g.n;
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NS; ++k) xv[k] = g.obs[((int64_t)t * NS + k) * g.n + i];
    r.x = make_float4(xv[0], xv[1], xv[2], xv[3]);
    float a = g.pd.cont ? g.action_f[f] : __int_as_float(g.action_i[f]);
    r.misc = make_float4(g.logp[f], valid ? g.adv[f] : 0.0f, g.ret[f], a);
    return r;
}

// LDS of one team of 8 waves: x[2][TILE] | misc[2][TILE] | part[16][TILE] | dL[TILE] (float4) | comb[14][256] + 16 scalars (float)
constexpr int NPART = 2 * NW;  // partial head sums per sample: one per wave and lane half (phase 1a)
__host__ __device__ constexpr size_t grad_team_smem_bytes() {
    return sizeof(float4) * (size_t)(2 * TILE + 2 * TILE + NPART * TILE + TILE) + sizeof(float) * (14 * 256 + 16);
}
// ... and, shared by the workgroup's teams, behind their areas: the unit records as phase 1a reads them, [NW][32] slots of
// RS floats (slot 32 w + m = unit w (h / NW) + m; slots m >= h / NW stay zero: an absent unit adds exact zeros)
constexpr int RS = 20;  // record pitch in LDS (80 B: 16-byte aligned, and 32 consecutive slots spread over the banks)
constexpr size_t GRAD_REC_LDS_BYTES = sizeof(float) * (size_t)(NW * 32 * RS);
__host__ __device__ constexpr size_t grad_wg_smem_bytes(int nt) { return (size_t)nt * grad_team_smem_bytes() + GRAD_REC_LDS_BYTES; }
constexpr int GRAD_COMB_FLOATS = 14 * 256 + 16;  // the comb area + its 16 scalars, contiguous

struct TeamLds {
    float4* x;     // [2][TILE]
    float4* misc;  // [2][TILE]
    float4* part;  // [16][TILE] {a0, a1, a2, v} partial sums
    float4* dL;    // [TILE]     {dl0, dv, dl1, dl2}
    float* comb;   // [14][256] second-half accumulators, then 16 scalars of the team's wave 0
};
__device__ __forceinline__ TeamLds team_lds(char* smem, int team) {
    char* tsm = smem + (size_t)team * grad_team_smem_bytes();
    TeamLds L;
    L.x = reinterpret_cast<float4*>(tsm);
    L.misc = L.x + 2 * TILE;
    L.part = L.misc + 2 * TILE;
    L.dL = L.part + NPART * TILE;
    L.comb = reinterpret_cast<float*>(L.dL + TILE);
    return L;
}

// (actor, critic) pair.  A plain struct on purpose: every operation below is written per component and the file is built
// without the vectorizers (build.py), so that no packed f32 VALU instruction can appear in a kernel that issues MFMAs
// (tests/test_no_packed_f32_beside_mfma.py; round 2 paired these as v_pk_fma_f32 when the tile had no MFMA in it).
struct f2 {
    float x, y;
};
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2 fma2s(f2 a, float b, f2 c) { return f2{fmaf(a.x, b, c.x), fmaf(a.y, b, c.y)}; }
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
        b1 = f2{0.f, 0.f};
        w2p = f2{0.f, 0.f};
        w2a1 = w2a2 = 0.f;
    }
};
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

// The hidden activation of the tile.  tanh: ocml's tanhf restated without its branches -- the same two evaluations
// (odd polynomial below 0.625, 1 - 2 / (exp(2 |x|) + 1) above; constants and operation order read off the library's
// code for gfx950), both computed, one select: the same bits as tanhf (tools/micro/tanh_sel.hip checks all 2^32 inputs'
// worth of a stride), and no control flow between the reads of an MFMA accumulator tuple (with the library's branches the
// register allocator spilled 170 - 190 registers per lane in the tanh instantiations of phase 1a).
__device__ __forceinline__ float tanh_sel(float x) {
    const float y = fabsf(x), y2 = x * x;
    float p = fmaf(y2, __uint_as_float(0xbbbac73du), __uint_as_float(0x3ca908c9u));
    p = fmaf(y2, p, __uint_as_float(0xbd5c1c4eu));
    p = fmaf(y2, p, __uint_as_float(0x3e088382u));
    p = fmaf(y2, p, __uint_as_float(0xbeaaaa99u));
    const float zs = fmaf(y2, y * p, y);
    const float e = expf(y + y);
    const float zb = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f, 1.0f);
    return copysignf(y < 0.625f ? zs : zb, x);
}
// relu as ONE integer max on the bit pattern (negative floats are negative integers; -0 -> +0; a positive-sign NaN stays NaN):
// fmaxf on an MFMA result costs two v_max_f32 (the compiler cannot prove the accumulator free of signalling NaNs and
// canonicalises it first) -- 64 of the ~220 VALU instructions of phase 1a.  Same value as fmaxf(z, 0) for every non-NaN z.
template <int ACT>
__device__ __forceinline__ float tile_act(float z) {
    if (ACT == 0) {
        const int b = __float_as_int(z);
        return __int_as_float(b > 0 ? b : 0);
    }
    return tanh_sel(z);
}

// ---- the workgroup's LDS copy of the unit records for phase 1a: slot 32 w + m <- unit w hq + m (hq = h / NW), zeros beyond hq ----
// from the packed global image (two-launch kernel prologue), every thread of the workgroup, in two halves: request early,
// store late (the gradient kernel puts the first tile's gather in between); no barrier inside
template <int NT>
__device__ __forceinline__ void stage_records_load(float4 (&v)[(NW * 32 * 4) / (512 * NT)], const float* __restrict__ rec, int h) {
    const int hq = h / NW;
#pragma unroll
    for (int i = 0; i < (NW * 32 * 4) / (512 * NT); ++i) {
        const int idx = (int)threadIdx.x + 512 * NT * i, slot = idx >> 2, part = idx & 3;
        const int m = slot & 31, j = (slot >> 5) * hq + m;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < hq) v[i] = *reinterpret_cast<const float4*>(rec + REC * j + 4 * part);
    }
}
// the same registers straight from the parameter vector (packed == NULL: the first launch of an update call, whose unit-record
// image may be stale -- the host may have written `params` since the last optimiser step; every later step reads the image the
// tail of the step before it patched).  Same values as pack_records writes, so the tile computes the same bits.
template <int NT>
__device__ __forceinline__ void stage_records_from_params(float4 (&v)[(NW * 32 * 4) / (512 * NT)], const float* __restrict__ params,
                                                          int h, int ns, int nout, int64_t np_a) {
    const int hq = h / NW;
    const float* W1a = params;
    const float* b1a = W1a + h * ns;
    const float* W2a = b1a + h;
    const float* W1c = params + np_a;
    const float* b1c = W1c + h * ns;
    const float* W2c = b1c + h;
#pragma unroll
    for (int i = 0; i < (NW * 32 * 4) / (512 * NT); ++i) {
        const int idx = (int)threadIdx.x + 512 * NT * i, slot = idx >> 2, part = idx & 3;
        const int m = slot & 31, j = (slot >> 5) * hq + m;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < hq) {
            if (part < 2) {  // {w1a[k], w1c[k], w1a[k + 1], w1c[k + 1]}, k = 2 part
                const int k = 2 * part;
                if (k < ns) r.x = W1a[j + h * k], r.y = W1c[j + h * k];
                if (k + 1 < ns) r.z = W1a[j + h * (k + 1)], r.w = W1c[j + h * (k + 1)];
            } else if (part == 2) {
                r = make_float4(b1a[j], b1c[j], W2a[0 + nout * j], W2c[j]);
            } else {
                if (nout > 1) r.x = W2a[1 + nout * j];
                if (nout > 2) r.y = W2a[2 + nout * j];
            }
        }
        v[i] = r;
    }
}
template <int NT>
__device__ __forceinline__ void stage_records_store(float* l_rec, const float4 (&v)[(NW * 32 * 4) / (512 * NT)]) {
#pragma unroll
    for (int i = 0; i < (NW * 32 * 4) / (512 * NT); ++i) {
        const int idx = (int)threadIdx.x + 512 * NT * i, slot = idx >> 2, part = idx & 3;
        *reinterpret_cast<float4*>(l_rec + slot * RS + 4 * part) = v[i];
    }
}
__device__ __forceinline__ int record_lds_slot(int j, int h) {
    const int hq = h / NW;
    return (j / hq) * 32 + (j % hq);
}
// one unit's record from registers (persistent kernel: the weights never leave the CU); the padding slots are zeroed once
__device__ __forceinline__ void store_record_lds(float* l_rec, int j, int h, const UnitW& W) {
    float4* d = reinterpret_cast<float4*>(l_rec + record_lds_slot(j, h) * RS);
    d[0] = make_float4(W.w1[0].x, W.w1[0].y, W.w1[1].x, W.w1[1].y);
    d[1] = make_float4(W.w1[2].x, W.w1[2].y, W.w1[3].x, W.w1[3].y);
    d[2] = make_float4(W.b1.x, W.b1.y, W.w2p.x, W.w2p.y);
    d[3] = make_float4(W.w2a1, W.w2a2, 0.0f, 0.0f);
}

// The tile loop of one workgroup (NT teams side by side, same trip count: the barriers are workgroup-wide; a team
// without a tile left computes on stale LDS with all-invalid samples: dL = 0, nothing is accumulated).
// l_rec: the workgroup's LDS copy of the unit records (stage_records / store_record_lds below).  b2: {b2a0, b2a1, b2a2, b2c}.
// NO = 2: the actor has at most two outputs (two actions, or (mu, log sigma)) -- the third output's FMAs (zero weights,
// zero dL/dout: exact no-ops) are not issued; NO = 3: three actions.
template <int NS, int ACT, int NO, int NT>
__device__ __forceinline__ void grad_tile_loop(const GradArgs& g, const PermKeys& pk, uint32_t pos0, const TeamLds& L,
                                               const TeamIds& id, const float* l_rec, const float (&b2)[4], UnitG& G,
                                               HeadG& Hd, long long* tl = nullptr) {
    // tl (debug builds of the persistent kernel only): s_memtime at the loop's four barriers of the LAST pass
    const int lane = id.lane, w = id.w;
    const int r = lane & 31, kb = lane >> 5;
    int tile = blockIdx.x * NT + id.team;
    int buf = 0;
    for (int base = blockIdx.x * NT; base < g.num_tiles; base += gridDim.x * NT, tile += gridDim.x * NT) {
        const int next = tile + gridDim.x * NT;
        const float4* cx = L.x + buf * TILE;
        const float4* cm = L.misc + buf * TILE;
        if (tl) tl[0] = __builtin_amdgcn_s_memtime();
        // ---- phase 0 (next tile): wave 1 issues the gather now, publishes it after phase 2 ----
        TileRegs pre;
        const bool prefetcher = (w == 1) && (next < g.num_tiles);
        if (prefetcher) pre = fetch_sample<NS>(g, pk, pos0, next, lane);
        const float* rU = l_rec + (32 * w + r) * RS;  // record of this lane's unit slot (A operand of 1a, B operand of 2)
        // ---- phase 1a: D[unit slot 32 w + m][sample] = b1 + W1 x on the f32 MFMA, both nets, both 32-sample halves.
        //      A: lane (r, kb) = W1[slot 32 w + r][k = kb + 2 ks]; B: x[k][sample 32 rt + r]; C: the bias by register row.
        //      The lane then holds z of samples r and 32 + r for its 16 slots 32 w + 8 g + 4 kb + e: activation and
        //      head-weight FMAs on the VALU, weights by 8-byte LDS reads (the same address in every lane of a half) ----
        {
            const float* rA = rU + 2 * kb;
            const f2 a0 = *reinterpret_cast<const f2*>(rA);      // (W1a, W1c)[slot r][k = kb]
            const f2 a1 = *reinterpret_cast<const f2*>(rA + 4);  //                   [k = kb + 2]
            const float* xf = reinterpret_cast<const float*>(cx);
            float xb[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) xb[rt][ks] = xf[4 * (32 * rt + r) + kb + 2 * ks];
            const float* rq = l_rec + (32 * w + 4 * kb) * RS;  // slot of register row q: + ((q & 3) + 8 (q >> 2)) RS
            // one net at a time (32 accumulator registers live instead of 64: the 1024-thread workgroup has 128 per lane)
            float ac0[2] = {0.f, 0.f}, ac1[2] = {0.f, 0.f}, ac2[2] = {0.f, 0.f}, acv[2] = {0.f, 0.f};
            {  // actor
                f32x16 z[2];
                {
                    f32x16 bq;
#pragma unroll
                    for (int q = 0; q < 16; ++q) bq[q] = rq[((q & 3) + 8 * (q >> 2)) * RS + 8];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) z[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, xb[rt][0], bq, 0, 0, 0);
                }
                if (NS > 2) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) z[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, xb[rt][1], z[rt], 0, 0, 0);
                }
                // head weights of two slots at a time, the next two requested before the current two are used
                float w20[2][2];
                f2 w2x[2][2];
                auto load2 = [&](int g2, float (&a)[2], f2 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * g2 + e;
                        const float* u = rq + ((q & 3) + 8 * (q >> 2)) * RS;
                        a[e] = u[10];                                 // W2a[0, j]
                        b[e] = *reinterpret_cast<const f2*>(u + 12);  // (W2a[1, j], W2a[2, j])
                    }
                };
                load2(0, w20[0], w2x[0]);
#pragma unroll
                for (int g2 = 0; g2 < 8; ++g2) {
                    if (g2 < 7) load2(g2 + 1, w20[(g2 + 1) & 1], w2x[(g2 + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            const float ha = tile_act<ACT>(z[rt][2 * g2 + e]);
                            ac0[rt] = fmaf(w20[g2 & 1][e], ha, ac0[rt]);
                            ac1[rt] = fmaf(w2x[g2 & 1][e].x, ha, ac1[rt]);
                            if (NO > 2) ac2[rt] = fmaf(w2x[g2 & 1][e].y, ha, ac2[rt]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {  // critic.  Its record reads start from an offset the compiler cannot see through, tied to the actor's last
               // sum: otherwise the (b1a, b1c, w2a0, w2c) quadruples are fetched as one 16-byte read per unit for both nets
               // up front and the critic's halves stay live across the actor's pass (118 spilled registers)
                int co = 9;
                asm volatile("" : "+v"(co), "+v"(ac0[1]));
                const float* rqc = rq + co;
                f32x16 z[2];
                {
                    f32x16 bq;
#pragma unroll
                    for (int q = 0; q < 16; ++q) bq[q] = rqc[((q & 3) + 8 * (q >> 2)) * RS];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) z[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, xb[rt][0], bq, 0, 0, 0);
                }
                if (NS > 2) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) z[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, xb[rt][1], z[rt], 0, 0, 0);
                }
                float w2c[2][4];  // W2c[j], four slots at a time, the next four requested before the current four are used
                auto load4c = [&](int g4, float (&a)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = rqc[(e + 8 * g4) * RS + 2];
                };
                load4c(0, w2c[0]);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (g4 < 3) load4c(g4 + 1, w2c[(g4 + 1) & 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) acv[rt] = fmaf(w2c[g4 & 1][e], tile_act<ACT>(z[rt][4 * g4 + e]), acv[rt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                L.part[(2 * w + kb) * TILE + 32 * rt + r] = make_float4(ac0[rt], ac1[rt], ac2[rt], acv[rt]);
        }
        __syncthreads();
        if (tl) tl[1] = __builtin_amdgcn_s_memtime();
        // ---- phase 1b: wave 0 finishes the forward, evaluates the loss and dL/d(outputs) ----
        if (w == 0) {
            const int s = lane;
            const bool valid = ((uint32_t)tile * TILE + (uint32_t)s) < g.bm;
            float4 ps = L.part[s];
#pragma unroll
            for (int q = 1; q < NPART; ++q) {  // fixed summation order over the partial sums (wave, lane half)
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
                // na <= NO (compile time): every loop over the actions is unrolled to NO trips with a uniform predicate, so
                // that logp / pr / dl live in registers (run-time trip counts made them indexed arrays)
                const int na = g.pd.na;
                float mx = oa[0];
#pragma unroll
                for (int k = 1; k < NO; ++k)
                    if (k < na) mx = fmaxf(mx, oa[k]);
                float se = 0.f;
#pragma unroll
                for (int k = 0; k < NO; ++k)
                    if (k < na) se += expf(oa[k] - mx);
                const float lse = logf(se);
                float logp[GMAXO] = {0.f, 0.f, 0.f}, pr[GMAXO] = {0.f, 0.f, 0.f};
                ent = 0.f;
#pragma unroll
                for (int k = 0; k < NO; ++k)
                    if (k < na) {
                        logp[k] = (oa[k] - mx) - lse;
                        pr[k] = expf(logp[k]);
                        ent -= pr[k] * logp[k];
                    }
                const int a = __float_as_int(mi.w);
                lp_new = 0.f;
#pragma unroll
                for (int k = 0; k < NO; ++k)
                    if (k < na && k == a) lp_new = logp[k];
                const float ratio = expf(lp_new - lp_old);
                const float surr1 = ratio * A;
                const float rc = fminf(fmaxf(ratio, g.lo), g.hi);
                const float surr2 = rc * A;
                const bool inside = ratio >= g.lo && ratio <= g.hi;
                const float dobj = (inside || surr1 < surr2) ? A : 0.f;
                const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
                if (valid) Hd.s_actor += fminf(surr1, surr2);
#pragma unroll
                for (int k = 0; k < NO; ++k)
                    if (k < na) {
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
        if (tl) tl[2] = __builtin_amdgcn_s_memtime();
        // publish the prefetched next tile into the other buffer (nobody reads it before the pass's last barrier; the gather
        // had phases 1a and 1b to arrive).  BEFORE phase 2: with a conditional block between phase 2 and the barrier the
        // compiler sinks phase 2's arithmetic below it while the LDS reads stay above the stores: 266 spilled registers.
        if (prefetcher) {
            L.x[(buf ^ 1) * TILE + lane] = pre.x;
            L.misc[(buf ^ 1) * TILE + lane] = pre.misc;
        }
        // ---- phase 2: lane (r, kb) = unit slot 32 w + r, its rows of the tile's samples (see the top of the file) ----
        {
            const f2 w2p = *reinterpret_cast<const f2*>(rU + 10);  // (W2a[0, j], W2c[j])
            const f2 w2x = *reinterpret_cast<const f2*>(rU + 12);  // (W2a[1, j], W2a[2, j])
            // one row: everything that depends on the sample's dL and x
            auto row = [&](float za, float zc, const float4& xv, const float4& d) __attribute__((always_inline)) {
                const f2 z = {za, zc};
                const f2 hh = {tile_act<ACT>(z.x), tile_act<ACT>(z.y)};
                const f2 d0v = {d.x, d.y};  // d = {dl0, dv, dl1, dl2}
                G.w2p = fma2(d0v, hh, G.w2p);
                G.w2a1 = fmaf(d.z, hh.x, G.w2a1);
                if (NO > 2) G.w2a2 = fmaf(d.w, hh.x, G.w2a2);
                f2 dh = {d0v.x * w2p.x, d0v.y * w2p.y};
                dh.x = fmaf(d.z, w2x.x, dh.x);
                if (NO > 2) dh.x = fmaf(d.w, w2x.y, dh.x);
                f2 dz;
                if (ACT == 0) {  // relu: dh * [z > 0] as a select (the product differs only in the sign of a zero)
                    dz = f2{z.x > 0.0f ? dh.x : 0.0f, z.y > 0.0f ? dh.y : 0.0f};
                } else {
                    dz = f2{dh.x * (1.0f - hh.x * hh.x), dh.y * (1.0f - hh.y * hh.y)};
                }
                G.b1.x += dz.x;
                G.b1.y += dz.y;
                G.w1[0] = fma2s(dz, xv.x, G.w1[0]);
                if (NS > 1) G.w1[1] = fma2s(dz, xv.y, G.w1[1]);
                if (NS > 2) G.w1[2] = fma2s(dz, xv.z, G.w1[2]);
                if (NS > 3) G.w1[3] = fma2s(dz, xv.w, G.w1[3]);
            };
            // the operands of the next row are requested before the current row is consumed
            const float4* xrow = cx + 4 * kb;
            const float4* drow = L.dL + 4 * kb;
            auto fetch = [&](int P, float4& xv, float4& dd) __attribute__((always_inline)) {
                const int q = P & 15, o = 32 * (P >> 4) + (q & 3) + 8 * (q >> 2);
                xv = xrow[o];
                dd = drow[o];
            };
            // one 32-sample half at a time (32 accumulator registers); its MFMAs first
            float4 xv[2], dd[2];
            fetch(0, xv[0], dd[0]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f2 a0 = *reinterpret_cast<const f2*>(rU + 2 * kb);
                const f2 a1 = *reinterpret_cast<const f2*>(rU + 4 + 2 * kb);
                const f2 bu = *reinterpret_cast<const f2*>(rU + 8);  // (b1a, b1c) of the lane's unit
                const float* xf = reinterpret_cast<const float*>(cx) + 4 * (32 * rt + r) + kb;
                f32x16 ya, yc;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    ya[q] = bu.x;
                    yc[q] = bu.y;
                }
                ya = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[0], a0.x, ya, 0, 0, 0);
                yc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[0], a0.y, yc, 0, 0, 0);
                if (NS > 2) {
                    ya = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[2], a1.x, ya, 0, 0, 0);
                    yc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[2], a1.y, yc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int P = 16 * rt + q;
                    if (P < 31) fetch(P + 1, xv[(P + 1) & 1], dd[(P + 1) & 1]);
                    row(ya[q], yc[q], xv[P & 1], dd[P & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the accumulators are "used" here: without this the compiler sinks the rows' arithmetic (results needed only
            // after the loop) below the barrier and the conditional blocks that follow it, while the LDS reads stay above
            // them -- every row's operands then live in scratch (266 spilled registers)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(G.w1[k].x), "v"(G.w1[k].y));
            asm volatile("" ::"v"(G.b1.x), "v"(G.b1.y), "v"(G.w2p.x), "v"(G.w2p.y), "v"(G.w2a1), "v"(G.w2a2));
        }
        __syncthreads();
        if (tl) tl[3] = __builtin_amdgcn_s_memtime();
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

// v[lane & 31] + v[32 + (lane & 31)] in every lane (lower half first): v_permlane32_swap exchanges the upper half of one
// register with the lower half of another on the VALU (gfx950); ds_bpermute would take the same 14 values of all 16 waves
// through the LDS crossbar (measured: +0.7 us on the fold)
__device__ __forceinline__ float half_sum(float v) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// Fold the workgroup's accumulators into ONE partial row, fixed order: the two lanes of a unit (phase 2) are added and
// handed through LDS to thread j of team 0 (j = unit), which adds team 0's and then team 1's value; wave 0 of each team
// sums its per-sample-lane accumulators over the 64 lanes, team 1's go to team 0 the same way.  Afterwards the threads
// with team == 0, shalf == 0 hold the row's unit gradients and every lane of team 0's wave 0 holds the head sums.
// Contains one workgroup barrier.
template <int NT>
__device__ __forceinline__ void grad_fold(char* smem, const TeamLds& L, const TeamIds& id, int h, UnitG& G, HeadG& Hd) {
    {   // the two lanes of a unit hold its sums over complementary rows: add them (lane half 0 first), hand the unit's
        // gradients to the thread that writes them out (unit j of the first 256 threads of the team)
        const int r = id.lane & 31, kb = id.lane >> 5, hq = h / NW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            G.w1[k].x = half_sum(G.w1[k].x);
            G.w1[k].y = half_sum(G.w1[k].y);
        }
        G.b1.x = half_sum(G.b1.x);
        G.b1.y = half_sum(G.b1.y);
        G.w2p.x = half_sum(G.w2p.x);
        G.w2p.y = half_sum(G.w2p.y);
        G.w2a1 = half_sum(G.w2a1);
        G.w2a2 = half_sum(G.w2a2);
        if (kb == 0 && r < hq) comb_store(L.comb + id.w * hq + r, G);
    }
    float* l_sc = L.comb + 14 * 256;  // [16] scalars of this team's wave 0
    if (id.w == 0) {  // wave 0: reduce the per-sample-lane accumulators over the 64 lanes
#pragma unroll
        for (int o = 0; o < GMAXO; ++o) Hd.b2a[o] = wave_sum_f32(Hd.b2a[o]);
        Hd.b2c = wave_sum_f32(Hd.b2c);
        Hd.s_actor = wave_sum_f32(Hd.s_actor);
        Hd.s_critic = wave_sum_f32(Hd.s_critic);
        Hd.s_ent = wave_sum_f32(Hd.s_ent);
        if (NT > 1 && id.team == 1 && id.lane == 0) {
            l_sc[0] = Hd.b2a[0];
            l_sc[1] = Hd.b2a[1];
            l_sc[2] = Hd.b2a[2];
            l_sc[3] = Hd.b2c;
            l_sc[4] = Hd.s_actor;
            l_sc[5] = Hd.s_critic;
            l_sc[6] = Hd.s_ent;
        }
    }
    __syncthreads();  // ONE barrier: every team's unit sums and team 1's head sums are in LDS
    if (id.team == 0) {
        if (id.shalf == 0) {
            G.zero();
            comb_add(L.comb + id.uidx, G);  // team 0 first, then team 1: the order of the three-barrier fold it replaces
            if (NT > 1) comb_add(team_lds(smem, 1).comb + id.uidx, G);
        }
        if (NT > 1 && id.w == 0) {
            const float* sc1 = team_lds(smem, 1).comb + 14 * 256;
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

// ---- host-side launch description (ppo_grad.hip) ----
struct GradLaunch {
    GradArgs g;
    int nb, ns, nt;  // partial rows (= workgroups), observation size, teams (tiles side by side) per workgroup
    int64_t np;
    unsigned int* counter;
    double* sumsq;
    float* packed;
};

}  // namespace rlhip
