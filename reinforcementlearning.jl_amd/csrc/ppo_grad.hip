// ppo_grad.hip -- PPO clipped-surrogate loss + gradient over one micro-batch, and the fused
// "reduce partial gradients -> clip_by_global_norm! -> Adam" tail of the single-GPU update.
//
// Replaces the body of the removed Zoo `PPOPolicy` update (hyper-parameters blog
// a_practical_introduction_to_RL.jl/index.html:15257-15278): shuffled micro-batch gather, actor/critic
// forward (Flux Dense chains), softmax / ratio / clamp / min, value and entropy terms, Zygote backward,
// then clip_by_global_norm! (RLCore/utils/basic.jl:19-29) and Flux.Optimise.update! with Adam
// (RLCore/policies/learners/flux_approximator.jl:46).
//
// ppo_grad_kernel (256 threads = 4 waves, persistent over 64-sample tiles):
//   prologue  every hidden unit's weights go to LDS as one 32-byte record per net
//             {W1[j,0..3], b1[j], W2[0..2,j]} -- phase 1 reads them with two broadcast ds_read_b128 per
//             unit per net (conflict-free: all lanes read one address) instead of 13 dependent scalar
//             loads (the first version's bottleneck: 47.8 us/launch, profiles/r01_a_*).
//   phase 0   64 threads fetch the tile's samples f = perm(pos) (keyed bijection, no index array) from
//             the trajectory into registers ONE TILE AHEAD -- the HBM/L2 latency of the scattered gather
//             hides under phases 1-2 of the current tile -- and publish them to a double-buffered LDS tile.
//   phase 1a  lane = sample, wave w walks its quarter of the hidden units; partial output sums -> LDS.
//   phase 1b  wave 0 finishes logits / value, evaluates the loss terms and dL/d(outputs) per sample.
//   phase 2   lane = hidden unit j (weights in registers); the 64 samples stream from LDS as two
//             broadcast b128 reads each; weight gradients accumulate in registers: no atomics, no
//             cross-lane reductions, fixed summation order.
//   epilogue  each workgroup writes one partial gradient (parameter layout); summation across
//             workgroups is done in a fixed order by reduce_apply_kernel / reduce_partials_kernel, so
//             the result is run-to-run deterministic and replicas on different GPUs stay bit-identical.
// Roofline: VALU-f32 bound by construction (K = ns <= 4 and N = nout <= 3 are far below an MFMA tile);
// algorithmic work 6*h*((ns+nout)+(ns+1)) flop per sample.
#include "ppo_common.h"

#include <stdlib.h>

extern "C" int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow,
                                       int64_t n, float grad_scale, float clip_norm, float lr, float beta1,
                                       float beta2, float eps, float* gn_out, rlhip_stream_t stream);
extern "C" int64_t rlhip_ppo_nparams(int32_t kind, const rlhip_ppo_cfg* c);

namespace rlhip {

constexpr float LOG2PI_F = 1.8378770664093453f;  // log(2f0 * pi) as Float32 (RLCore/utils/distributions.jl:9)
constexpr int TILE = 64;
constexpr int MAX_GRAD_BLOCKS = 512;
constexpr int NW = 8;     // waves per workgroup (per 64-sample tile)
constexpr int GMAXO = 3;  // actor outputs handled by the fused gradient kernel (na <= 3, or (mu, log sigma))

struct GradArgs {
    const float* obs;
    const float* logp;
    const float* adv;
    const float* ret;
    const float* action_f;
    const int32_t* action_i;
    const float* params;
    const float* packed;   // unit records: actor [h][8] | critic [h][8] | {b2a0, b2a1, b2a2, b2c}
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
    long long* dbg;  // optional per-block phase timestamps (s_memtime), 8 per block; NULL in production
};

// Phase timestamps are a compile-time option (-DRLHIP_GRAD_TIMING).  They are kept in (scalar) registers and stored at
// the very end: a global store at kernel entry makes every later uniform global load "possibly clobbered", which turns
// the s_load_dwordx8 record fetches of phase 1 into per-lane global_load_dwordx4 (measured: 22 -> 29.6 us per launch,
// all of it in phase 1a).
#ifdef RLHIP_GRAD_TIMING
#define DBG_DECL long long dbg_t[6] = {0, 0, 0, 0, 0, 0}
#define DBG_STAMP(k) dbg_t[(k)] = __builtin_amdgcn_s_memtime()
#define DBG_FLUSH()                                                                          \
    do {                                                                                     \
        if (g.dbg && threadIdx.x == 0) {                                                     \
            _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) g.dbg[(int64_t)blockIdx.x * 8 + k_] = dbg_t[k_]; \
        }                                                                                    \
    } while (0)
#else
#define DBG_DECL \
    do {         \
    } while (0)
#define DBG_STAMP(k) \
    do {             \
    } while (0)
#define DBG_FLUSH() \
    do {            \
    } while (0)
#endif

struct TileRegs {  // one sample's trajectory entries, held in registers one tile ahead
    float4 x;
    float4 misc;  // {logp_old, adv, ret, action (int bits or float)}
};

template <int NS>
__device__ __forceinline__ TileRegs fetch_sample(const GradArgs& g, const PermKeys& pk, int tile, int s) {
    uint32_t q = (uint32_t)tile * TILE + (uint32_t)s;
    bool valid = q < g.bm;
    uint32_t f = permute(pk, g.pos0 + (valid ? q : 0u));
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

// NO = 2: the actor has at most two outputs (two actions, or (mu, log sigma)) -- the third output's FMAs (zero weights,
// zero dL/dout: exact no-ops) are not issued; NO = 3: three actions.
// NT: teams of 8 waves per workgroup.  NT = 2: a 1024-thread workgroup walks TWO 64-sample tiles side by side (same code,
// same barriers, its own LDS carve per team) and folds both into ONE partial row: half the rows to write at the end of
// the launch (512 rows x 13 KB cost 2.7 us of an 18.8 us launch: the launch cannot retire before they are flushed) and
// half the rows for the optimiser tail to read back.  The two-workgroups-per-CU occupancy of NT = 1 is kept (16 waves).
template <int NS, int ACT, int NO, int NT>
__global__ __launch_bounds__(512 * NT) void ppo_grad_kernel(GradArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DBG_DECL;
    DBG_STAMP(0);
    const int h = g.pd.h;
    const int gtid = threadIdx.x;
    const int team = NT > 1 ? __builtin_amdgcn_readfirstlane(gtid >> 9) : 0;
    // LDS carve per team (all 16-byte aligned): x[2][TILE] | misc[2][TILE] | part[8][TILE] | dL[TILE] | comb[14][256]
    char* tsm = smem + (size_t)team * grad_team_smem_bytes();
    float4* l_x = reinterpret_cast<float4*>(tsm);            // [2][TILE]
    float4* l_misc = l_x + 2 * TILE;                         // [2][TILE]
    float4* l_part = l_misc + 2 * TILE;                      // [8][TILE]  {a0, a1, a2, v} partial sums
    float4* l_dL = l_part + NW * TILE;                       // [TILE]     {dl0, dl1, dl2, dv}
    float* l_comb = reinterpret_cast<float*>(l_dL + TILE);   // [14][256] second-half accumulators

    const int tid = gtid & 511;  // thread within the team
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nout = g.pd.nout_a;
    const int hq = h / NW;               // hidden units per wave in phase 1a
    const int uidx = tid & 255;          // phase 2: hidden unit of this thread
    const int shalf = w >> 2;            // phase 2: which half of the tile's samples (0: 0..31, 1: 32..63)
    const float* __restrict__ recA = (const float*)__builtin_assume_aligned(g.packed, 32);
    const float* __restrict__ recC = recA + 8 * h;
    const float* __restrict__ tailb = recC + 8 * h;

    // ---- prologue: the first tile's scattered gather is issued first; this thread's unit (phase 2)
    // comes from its two records ----
    const PermKeys pk = g.ctr ? perm_keys(g.seed, g.epoch_local + g.ctr[1] * g.n_epochs, g.total) : g.pk;
    int tile = blockIdx.x * NT + team;
    TileRegs first;
    const bool first_loader = tid < TILE && tile < g.num_tiles;
    if (first_loader) first = fetch_sample<NS>(g, pk, tile, tid);
    const bool owner = uidx < h;
    const int jown = owner ? uidx : 0;
    const float4 oa0 = *reinterpret_cast<const float4*>(recA + 8 * jown);
    const float4 oa1 = *reinterpret_cast<const float4*>(recA + 8 * jown + 4);
    const float4 oc0 = *reinterpret_cast<const float4*>(recC + 8 * jown);
    const float4 oc1 = *reinterpret_cast<const float4*>(recC + 8 * jown + 4);
    const float rw1a[4] = {oa0.x, oa0.y, oa0.z, oa0.w}, rw1c[4] = {oc0.x, oc0.y, oc0.z, oc0.w};
    const float rw2a[GMAXO] = {oa1.y, oa1.z, oa1.w};
    const float rb1a = oa1.x, rb1c = oc1.x, rw2c = oc1.y;
    float gw1a[4] = {0.f, 0.f, 0.f, 0.f}, gw1c[4] = {0.f, 0.f, 0.f, 0.f}, gw2a[GMAXO] = {0.f, 0.f, 0.f};
    float gb1a = 0.f, gb1c = 0.f, gw2c = 0.f;
    // wave-0 per-sample-lane accumulators: output-bias gradients and loss sums
    float gb2a[GMAXO] = {0.f, 0.f, 0.f};
    float gb2c = 0.f, s_actor = 0.f, s_critic = 0.f, s_ent = 0.f;
    const float b2a0 = tailb[0], b2a1 = tailb[1], b2a2 = tailb[2], b2cv = tailb[3];

    int buf = 0;
    if (first_loader) {
        l_x[tid] = first.x;
        l_misc[tid] = first.misc;
    } else if (NT > 1 && tid < TILE) {  // a team without a tile: finite operands, so that its (all-zero-weight) sums stay exact zeros
        l_x[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l_misc[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l_x[TILE + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        l_misc[TILE + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    DBG_STAMP(1);

    // both teams run the same number of trips (the barriers are workgroup-wide); a team without a tile left computes on
    // stale LDS with all-invalid samples: dL = 0, nothing is accumulated
    for (int base = blockIdx.x * NT; base < g.num_tiles; base += gridDim.x * NT, tile += gridDim.x * NT) {
        const int next = tile + gridDim.x * NT;
        const float4* cx = l_x + buf * TILE;
        const float4* cm = l_misc + buf * TILE;
        // ---- phase 0 (next tile): wave 1 issues the gather now, publishes it after phase 2 ----
        TileRegs pre;
        const bool prefetcher = (w == 1) && (next < g.num_tiles);
        if (prefetcher) pre = fetch_sample<NS>(g, pk, next, lane);
        // ---- phase 1a: lane = sample, wave w walks hidden units [w*hq, (w+1)*hq) ----
        {
            const float4 xv = cx[lane];
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accv = 0.f;
#pragma unroll 8
            for (int jj = w * hq; jj < (w + 1) * hq; ++jj) {
                // wave-uniform addresses: these become s_load_dwordx8 (SGPR operands of the FMAs below)
                const float* pa_ = recA + 8 * jj;
                const float* pc_ = recC + 8 * jj;
                const float4 wa_ = make_float4(pa_[0], pa_[1], pa_[2], pa_[3]);
                const float4 ra_ = make_float4(pa_[4], pa_[5], pa_[6], pa_[7]);
                const float4 wc_ = make_float4(pc_[0], pc_[1], pc_[2], pc_[3]);
                const float4 rc_ = make_float4(pc_[4], pc_[5], pc_[6], pc_[7]);
                float za = ra_.x, zc = rc_.x;
                za = fmaf(wa_.x, xv.x, za);
                zc = fmaf(wc_.x, xv.x, zc);
                if (NS > 1) {
                    za = fmaf(wa_.y, xv.y, za);
                    zc = fmaf(wc_.y, xv.y, zc);
                }
                if (NS > 2) {
                    za = fmaf(wa_.z, xv.z, za);
                    zc = fmaf(wc_.z, xv.z, zc);
                }
                if (NS > 3) {
                    za = fmaf(wa_.w, xv.w, za);
                    zc = fmaf(wc_.w, xv.w, zc);
                }
                const float ha = act_fwd_t<ACT>(za), hc = act_fwd_t<ACT>(zc);
                acc0 = fmaf(ra_.y, ha, acc0);
                acc1 = fmaf(ra_.z, ha, acc1);
                if (NO > 2) acc2 = fmaf(ra_.w, ha, acc2);
                accv = fmaf(rc_.y, hc, accv);
            }
            l_part[w * TILE + lane] = make_float4(acc0, acc1, acc2, accv);
        }
        __syncthreads();
        DBG_STAMP(2);
        // ---- phase 1b: wave 0 finishes the forward, evaluates the loss and dL/d(outputs) ----
        if (w == 0) {
            const int s = lane;
            const bool valid = ((uint32_t)tile * TILE + (uint32_t)s) < g.bm;
            float4 ps = l_part[s];
#pragma unroll
            for (int q = 1; q < NW; ++q) {  // fixed summation order over the NW waves' partial sums
                const float4 pq = l_part[q * TILE + s];
                ps.x += pq.x;
                ps.y += pq.y;
                ps.z += pq.z;
                ps.w += pq.w;
            }
            float oa[GMAXO], dl[GMAXO] = {0.f, 0.f, 0.f};
            oa[0] = ps.x + b2a0;
            oa[1] = ps.y + b2a1;
            oa[2] = ps.z + b2a2;
            const float v = ps.w + b2cv;
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
                if (valid) s_actor += fminf(surr1, surr2);
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
                if (valid) s_actor += fminf(surr1, surr2);
                dl[0] = dL_dlp * dmu;
                dl[1] = dL_dlp * dls - g.we * g.inv_b * 0.5f;
            }
            const float dv = mi.z - v;
            float dvout = -2.0f * g.wc * g.inv_b * dv;
            if (valid) {
                s_critic += dv * dv;
                s_ent += ent;
            } else {
                dl[0] = dl[1] = dl[2] = 0.f;
                dvout = 0.f;
            }
            l_dL[s] = make_float4(dl[0], dl[1], dl[2], dvout);
            gb2a[0] += dl[0];
            gb2a[1] += dl[1];
            gb2a[2] += dl[2];
            gb2c += dvout;
        }
        __syncthreads();
        DBG_STAMP(3);
        // ---- phase 2: lane = hidden unit j; the tile's samples stream from LDS (broadcast reads) ----
        if (owner) {
#pragma unroll 4
            for (int s = shalf * (TILE / 2); s < (shalf + 1) * (TILE / 2); ++s) {
                const float4 xv = cx[s];
                const float4 d = l_dL[s];
                float za = rb1a, zc = rb1c;
                za = fmaf(rw1a[0], xv.x, za);
                zc = fmaf(rw1c[0], xv.x, zc);
                if (NS > 1) {
                    za = fmaf(rw1a[1], xv.y, za);
                    zc = fmaf(rw1c[1], xv.y, zc);
                }
                if (NS > 2) {
                    za = fmaf(rw1a[2], xv.z, za);
                    zc = fmaf(rw1c[2], xv.z, zc);
                }
                if (NS > 3) {
                    za = fmaf(rw1a[3], xv.w, za);
                    zc = fmaf(rw1c[3], xv.w, zc);
                }
                const float ha = act_fwd_t<ACT>(za), hc = act_fwd_t<ACT>(zc);
                gw2a[0] = fmaf(d.x, ha, gw2a[0]);
                gw2a[1] = fmaf(d.y, ha, gw2a[1]);
                if (NO > 2) gw2a[2] = fmaf(d.z, ha, gw2a[2]);
                float dh = d.x * rw2a[0];
                dh = fmaf(d.y, rw2a[1], dh);
                if (NO > 2) dh = fmaf(d.z, rw2a[2], dh);
                // relu: dh * [z > 0] as a select (the product differs only in the sign of a zero)
                const float dza = ACT == 0 ? (za > 0.0f ? dh : 0.0f) : dh * act_bwd_t<ACT>(za, ha);
                gw2c = fmaf(d.w, hc, gw2c);
                const float dhc = d.w * rw2c;
                const float dzc = ACT == 0 ? (zc > 0.0f ? dhc : 0.0f) : dhc * act_bwd_t<ACT>(zc, hc);
                gb1a += dza;
                gb1c += dzc;
                gw1a[0] = fmaf(dza, xv.x, gw1a[0]);
                gw1c[0] = fmaf(dzc, xv.x, gw1c[0]);
                if (NS > 1) {
                    gw1a[1] = fmaf(dza, xv.y, gw1a[1]);
                    gw1c[1] = fmaf(dzc, xv.y, gw1c[1]);
                }
                if (NS > 2) {
                    gw1a[2] = fmaf(dza, xv.z, gw1a[2]);
                    gw1c[2] = fmaf(dzc, xv.z, gw1c[2]);
                }
                if (NS > 3) {
                    gw1a[3] = fmaf(dza, xv.w, gw1a[3]);
                    gw1c[3] = fmaf(dzc, xv.w, gw1c[3]);
                }
            }
        }
        // publish the prefetched next tile into the other buffer (nobody reads it before the barrier)
        if (prefetcher) {
            l_x[(buf ^ 1) * TILE + lane] = pre.x;
            l_misc[(buf ^ 1) * TILE + lane] = pre.misc;
        }
        __syncthreads();
        DBG_STAMP(4);
        buf ^= 1;
    }

    // ---- epilogue: this workgroup's partial gradient (fixed layout = parameter layout) ----
    // waves 4..7 (second half of the samples) hand their accumulators to waves 0..3 through LDS
    if (shalf == 1) {
        float* c = l_comb + uidx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[(k)*256] = gw1a[k];
            c[(4 + k) * 256] = gw1c[k];
        }
        c[8 * 256] = gb1a;
        c[9 * 256] = gb1c;
        c[10 * 256] = gw2a[0];
        c[11 * 256] = gw2a[1];
        c[12 * 256] = gw2a[2];
        c[13 * 256] = gw2c;
    }
    __syncthreads();
    if (shalf == 0) {
        const float* c = l_comb + uidx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gw1a[k] += c[(k)*256];
            gw1c[k] += c[(4 + k) * 256];
        }
        gb1a += c[8 * 256];
        gb1c += c[9 * 256];
        gw2a[0] += c[10 * 256];
        gw2a[1] += c[11 * 256];
        gw2a[2] += c[12 * 256];
        gw2c += c[13 * 256];
    }
    float* l_sc = l_comb + 14 * 256;  // [16] scalars of this team's wave 0
    if (w == 0) {  // wave 0: reduce the per-sample-lane accumulators over the 64 lanes
#pragma unroll
        for (int o = 0; o < GMAXO; ++o) gb2a[o] = wave_sum_f32(gb2a[o]);
        gb2c = wave_sum_f32(gb2c);
        s_actor = wave_sum_f32(s_actor);
        s_critic = wave_sum_f32(s_critic);
        s_ent = wave_sum_f32(s_ent);
    }
    if (NT > 1) {
        // team 1 hands its (already half-combined) accumulators to team 0 through its own comb area, fixed order
        __syncthreads();  // the readers of the comb areas (above) are done
        if (team == 1) {
            if (shalf == 0) {
                float* c = l_comb + uidx;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    c[(k)*256] = gw1a[k];
                    c[(4 + k) * 256] = gw1c[k];
                }
                c[8 * 256] = gb1a;
                c[9 * 256] = gb1c;
                c[10 * 256] = gw2a[0];
                c[11 * 256] = gw2a[1];
                c[12 * 256] = gw2a[2];
                c[13 * 256] = gw2c;
            }
            if (w == 0 && lane == 0) {
                l_sc[0] = gb2a[0];
                l_sc[1] = gb2a[1];
                l_sc[2] = gb2a[2];
                l_sc[3] = gb2c;
                l_sc[4] = s_actor;
                l_sc[5] = s_critic;
                l_sc[6] = s_ent;
            }
        }
        __syncthreads();
        if (team == 1) return;
        const float* c1 = reinterpret_cast<const float*>(smem + grad_team_smem_bytes() + sizeof(float4) * (size_t)(2 * TILE + 2 * TILE + NW * TILE + TILE));
        if (shalf == 0) {
            const float* c = c1 + uidx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gw1a[k] += c[(k)*256];
                gw1c[k] += c[(4 + k) * 256];
            }
            gb1a += c[8 * 256];
            gb1c += c[9 * 256];
            gw2a[0] += c[10 * 256];
            gw2a[1] += c[11 * 256];
            gw2a[2] += c[12 * 256];
            gw2c += c[13 * 256];
        }
        if (w == 0) {
            const float* sc1 = c1 + 14 * 256;
            gb2a[0] += sc1[0];
            gb2a[1] += sc1[1];
            gb2a[2] += sc1[2];
            gb2c += sc1[3];
            s_actor += sc1[4];
            s_critic += sc1[5];
            s_ent += sc1[6];
        }
    }
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
    if (owner && shalf == 0) {
        const int j = uidx;
        float* oa_ = out;
        float* oc_ = out + g.pd.np_a;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            oa_[j + h * k] = gw1a[k];
            oc_[j + h * k] = gw1c[k];
        }
        oa_[h * NS + j] = gb1a;
        oc_[h * NS + j] = gb1c;
#pragma unroll
        for (int o = 0; o < GMAXO; ++o)
            if (o < nout) oa_[h * NS + h + o + nout * j] = gw2a[o];
        oc_[h * NS + h + j] = gw2c;
    }
    if (w == 0 && lane == 0) {
        for (int o = 0; o < nout; ++o) out[h * NS + h + nout * h + o] = gb2a[o];
        out[g.pd.np_a + h * NS + h + h] = gb2c;
        float* lo_ = g.loss_partials + (int64_t)blockIdx.x * 4;
        lo_[0] = s_actor;
        lo_[1] = s_critic;
        lo_[2] = s_ent;
        lo_[3] = 0.f;
    }
    DBG_STAMP(5);
    DBG_FLUSH();
}

static size_t grad_smem_bytes(int h) {
    (void)h;
    return grad_team_smem_bytes();
}

// unit records {W1[j,0..3], b1[j], W2[0..2,j]} per net + the output biases; any thread count
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
    for (int q = tid; q < 2 * h; q += nthreads) {
        const int net = q / h, j = q - net * h;
        float rec[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (net == 0) {
            for (int k = 0; k < ns; ++k) rec[k] = W1a[j + h * k];
            rec[4] = b1a[j];
            for (int o = 0; o < nout; ++o) rec[5 + o] = W2a[o + nout * j];
        } else {
            for (int k = 0; k < ns; ++k) rec[k] = W1c[j + h * k];
            rec[4] = b1c[j];
            rec[5] = W2c[j];
        }
        float4* dst = reinterpret_cast<float4*>(packed + 8 * (int64_t)q);
        dst[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        dst[1] = make_float4(rec[4], rec[5], rec[6], rec[7]);
    }
    if (tid == 0) {
        float* t = packed + 16 * (int64_t)h;
        t[0] = (0 < nout) ? b2a[0] : 0.f;
        t[1] = (1 < nout) ? b2a[1] : 0.f;
        t[2] = (2 < nout) ? b2a[2] : 0.f;
        t[3] = b2c[0];
    }
}

__global__ __launch_bounds__(256) void pack_params_kernel(const float* __restrict__ params, float* __restrict__ packed,
                                                          int h, int ns, int nout, int64_t np_a) {
    pack_records(params, packed, h, ns, nout, np_a, threadIdx.x, blockDim.x);
}

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

// position of flat parameter q in the unit-record copy (see pack_records)
__device__ __forceinline__ int64_t record_slot(int64_t q, int h, int ns, int nout, int64_t np_a) {
    const int net = q >= np_a ? 1 : 0;
    const int64_t r = q - (net ? np_a : 0);
    const int no = net ? 1 : nout;
    const int64_t base = 8 * (int64_t)net * h;
    if (r < (int64_t)h * ns) return base + 8 * (r % h) + (r / h);               // W1[j + h k] -> rec[j][k]
    if (r < (int64_t)h * ns + h) return base + 8 * (r - (int64_t)h * ns) + 4;   // b1[j]       -> rec[j][4]
    const int64_t w = r - ((int64_t)h * ns + h);
    if (w < (int64_t)no * h) return base + 8 * (w / no) + 5 + (w % no);         // W2[o + no j] -> rec[j][5 + o]
    return 16 * (int64_t)h + (net ? 3 : (w - (int64_t)no * h));                 // output biases -> tail
}

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
    __shared__ float l_loss[4];
    __shared__ double l_d[16];
    __shared__ int l_last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
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
    l_g[grp][pl] = acc;
    __syncthreads();
    float gsum = 0.f;
    if (grp == 0) {
#pragma unroll
        for (int q = 0; q < RG; ++q) gsum += l_g[q][pl];
        if (p < np) grad[p] = gsum;
        else gsum = 0.f;
    }
    if (blockIdx.x == 0 && losses != nullptr) {
        if (wv >= 1 && wv <= 3) {
            const int c = wv - 1;
            float a = 0.f;
            for (int b = lane; b < nb; b += 64) a += loss_partials[(int64_t)b * 4 + c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) l_loss[c] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float actor_loss = -l_loss[0] * inv_b;
            const float critic_loss = l_loss[1] * inv_b;
            const float ent_loss = l_loss[2] * inv_b;
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
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (lane == 0) {
            ap.sumsq[blockIdx.x] = sq;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ap.counter + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ap.counter + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
                __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double part = 0.0;
        for (int b = lane; b < (int)gridDim.x; b += 64)
            part += __hip_atomic_load(ap.sumsq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
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
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned int prev = __hip_atomic_fetch_add(ap.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                ap.beta_pow[0] *= ap.b1;
                ap.beta_pow[1] *= ap.b2;
                __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }

    // per-block partial sum of squares (threads 0..RP-1 of wave 0 hold this block's gradient values)
    if (wv == 0) {
        double sq = (grp == 0) ? (double)gsum * (double)gsum : 0.0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (lane == 0) ap.sumsq[blockIdx.x] = sq;
    }
    if (APPLY == APPLY_GRID) {
        // ---- every workgroup applies Adam to ITS OWN 64 parameters after a grid-wide barrier on the norm ----
        // All workgroups are co-resident (the host launches this variant only for gridDim <= grid_apply_max_blocks()), so a spin barrier on an
        // agent-scope counter is safe.  Only wave 0 (which holds the 64 reduced gradient values in registers)
        // continues; the other 15 waves are done.  The serial tail of the last-arriver variant (one workgroup running
        // Adam over all np parameters + re-packing every record) becomes 53 parallel 64-lane updates.
        if (wv != 0) return;
        const bool own = p < np;
        // operands of this lane's parameter do not depend on the barrier: issue the loads first
        const float m0 = own ? ap.m[p] : 0.0f, v0 = own ? ap.v[p] : 0.0f, p0 = own ? ap.params[p] : 0.0f;
        const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ap.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ap.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
                __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double part = 0.0;  // same summation order in every workgroup -> the same norm, bit for bit
        for (int b = lane; b < (int)gridDim.x; b += 64)
            part += __hip_atomic_load(ap.sumsq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
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
        // departure: the last workgroup out re-arms both counters and advances the running beta powers (every
        // workgroup has read beta_pow above, before its departure increment)
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            unsigned int prev = __hip_atomic_fetch_add(ap.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {
                ap.beta_pow[0] *= ap.b1;
                ap.beta_pow[1] *= ap.b2;
                __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ap.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_down(part, off, 64);
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
// device at hand (grid_barrier_capacity, common.h), queried once per process
template <int APPLY>
static int grid_apply_max_blocks() {
    static int cap = -1;
    if (cap < 0) {
        cap = grid_barrier_capacity(reduce_apply_kernel_ptr<APPLY>(), 1024);
        const char* e = getenv("RLHIP_GRID_BARRIER_CAP");  // test hook: force the barrier-free variants (0) or a small device
        if (e) cap = atoi(e);
    }
    return cap;
}

static int grad_blocks(int num_tiles) {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("RLHIP_GRAD_BLOCKS");
        cap = e ? atoi(e) : 512;
        if (cap < 1 || cap > MAX_GRAD_BLOCKS) cap = MAX_GRAD_BLOCKS;
    }
    return num_tiles < cap ? num_tiles : cap;
}

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

struct GradLaunch {
    GradArgs g;
    int nb, ns, nt;  // partial rows (= workgroups), observation size, teams (tiles side by side) per workgroup
    int64_t np;
    unsigned int* counter;
    double* sumsq;
    float* packed;
};

static int32_t prepare_grad(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                            const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace,
                            GradLaunch* out, const uint32_t* ctr = nullptr) {
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
    RLHIP_REQUIRE(traj && params && workspace, "NULL argument");
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
    // 8-wave workgroup each: below that, more (smaller) workgroups fill the chip better.  RLHIP_GRAD_TEAMS=1 / 2 forces.
    static int teams_env = -1;
    if (teams_env < 0) {
        const char* e = getenv("RLHIP_GRAD_TEAMS");
        teams_env = e ? atoi(e) : 0;
    }
    out->nt = teams_env == 1 ? 1 : (teams_env == 2 ? 2 : (g.num_tiles > 256 ? 2 : 1));
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
    static int dbg_on = -1;
    if (dbg_on < 0) dbg_on = getenv("RLHIP_GRAD_DEBUG") ? 1 : 0;
    long long* dbgp = (long long*)(out->counter + 16);
    g.dbg = dbg_on ? dbgp : nullptr;
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

static void launch_grad(const GradLaunch& L, hipStream_t s) {
    size_t smem = grad_smem_bytes(L.g.pd.h);
#define LAUNCH_G(NS_, ACT_)                                                                                            \
    do {                                                                                                              \
        if (L.nt == 2) {                                                                                              \
            if (L.g.pd.nout_a > 2)                                                                                    \
                hipLaunchKernelGGL((ppo_grad_kernel<NS_, ACT_, 3, 2>), dim3(L.nb), dim3(64 * NW * 2), 2 * smem, s, L.g); \
            else                                                                                                      \
                hipLaunchKernelGGL((ppo_grad_kernel<NS_, ACT_, 2, 2>), dim3(L.nb), dim3(64 * NW * 2), 2 * smem, s, L.g); \
        } else if (L.g.pd.nout_a > 2)                                                                                 \
            hipLaunchKernelGGL((ppo_grad_kernel<NS_, ACT_, 3, 1>), dim3(L.nb), dim3(64 * NW), smem, s, L.g);           \
        else                                                                                                          \
            hipLaunchKernelGGL((ppo_grad_kernel<NS_, ACT_, 2, 1>), dim3(L.nb), dim3(64 * NW), smem, s, L.g);           \
    } while (0)
    const int a = L.g.pd.act;
    if (L.ns == 4) { if (a == 0) LAUNCH_G(4, 0); else LAUNCH_G(4, 1); }
    else if (L.ns == 3) { if (a == 0) LAUNCH_G(3, 0); else LAUNCH_G(3, 1); }
    else { if (a == 0) LAUNCH_G(2, 0); else LAUNCH_G(2, 1); }
#undef LAUNCH_G
}

static void launch_pack(const GradLaunch& L, hipStream_t s) {
    hipLaunchKernelGGL(pack_params_kernel, dim3(1), dim3(256), 0, s, L.g.params, L.packed, L.g.pd.h, L.ns,
                       L.g.pd.nout_a, L.g.pd.np_a);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int64_t rlhip_ppo_workspace_bytes(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T) {
    if (is_layers3(cfg)) return ppo3_workspace_bytes(kind, cfg, n, T);
    int64_t np = rlhip_ppo_nparams(kind, cfg);
    if (np < 0) return -1;
    return (int64_t)MAX_GRAD_BLOCKS * (np + 4) * (int64_t)sizeof(float) + 16 + 4096 * (int64_t)sizeof(double) + 64 +
           (int64_t)MAX_GRAD_BLOCKS * 8 * (int64_t)sizeof(long long) + 64 + (16 * 256 + 8) * (int64_t)sizeof(float);
}

static int32_t grad_entry(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                          const float* params, uint64_t seed, uint32_t epoch_ctr, const uint32_t* ctr, int32_t mb,
                          void* workspace, float* grad_out, float* losses_out, rlhip_stream_t stream,
                          bool do_pack = true) {
    RLHIP_REQUIRE(grad_out != nullptr, "grad_out is NULL");
    if (is_layers3(cfg)) {
        RLHIP_REQUIRE(ctr == nullptr, "layers = 3: the device-counter (graph replay) variant is not built");
        return ppo3_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_out, losses_out, true, stream);
    }
    GradLaunch L;
    int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, &L, ctr);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    if (do_pack) launch_pack(L, s);
    launch_grad(L, s);
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
    (void)n;
    (void)T;
    PolicyDesc pd;
    int32_t rc = make_desc(kind, cfg, &pd);
    if (rc) return rc;
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
    const int rblocks = (int)((np + RP - 1) / RP);
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
                if (first) launch_pack(L, s);
                first = false;
                launch_grad(L, s);
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
        if (rc) return rc;
        return rlhip_comm_advance_seq(comm, n_steps);
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
        return ppo3_update(kind, cfg, n, T, traj, params, m, v, beta_pow, seed, update_ctr, workspace, grad_scratch,
                           losses_out, stream);
    }
    hipStream_t s = as_stream(stream);
    bool first = true;
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        uint32_t epoch_ctr = ctr ? (uint32_t)e : update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            GradLaunch L;
            int32_t rc = prepare_grad(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, &L, ctr);
            if (rc) return rc;
            if (first) {  // pack the unit records once per call; the Adam tail refreshes them after every step.
                          // The arrival counter needs no per-call memset: the workspace is zero-initialised
                          // by its owner (ABI contract) and the last-arriving workgroup re-arms it in-kernel.
                launch_pack(L, s);
                first = false;
            }
            launch_grad(L, s);
            ApplyArgs ap{params, m, v, beta_pow, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2, cfg->adam_eps,
                         L.counter, L.sumsq, L.packed, L.g.pd.h, L.ns, L.g.pd.nout_a, L.g.pd.np_a};
            const int rblocks = (int)((L.np + RP - 1) / RP);
            if (rblocks <= grid_apply_max_blocks<APPLY_GRID>() && !RLHIP_ENV_FLAG("RLHIP_APPLY_LAST_ARRIVER"))
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
