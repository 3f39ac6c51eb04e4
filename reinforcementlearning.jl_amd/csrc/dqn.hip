// dqn.hip -- the DQN hot path on the vector env: plan! (Q forward + eps-greedy) and the learner
// update (replay sample -> TD target -> Huber loss -> gradient), each as one launch.
//
// What it replaces in the reference:
//   plan!(QBasedPolicy, env)      RLCore/policies/q_based_policy.jl:30-32 ->
//                                 RLCore/policies/learners/abstract_learner.jl:28-39 (forward = model(state),
//                                 flux_approximator.jl:43) -> plan!(EpsilonGreedyExplorer, values)
//                                 RLCore/policies/explorers/epsilon_greedy_explorer.jl:108-112
//   optimise!(learner, stage, trajectory)   q_based_policy.jl:49 -> `for batch in trajectory`
//                                 (sampler gather, un-vendored RLTrajectories) -> removed Zoo
//                                 BasicDQN/DQN learner: y = r + gamma * (1 - t) * max_a' Qt(s', a'),
//                                 Flux.Losses.huber_loss, Zygote backward (docs/src/rlcore.md:28,
//                                 blog index.md:320-331, index.html:15121-15147)
// The parameter step itself (Adam, target sync) is optim.hip.
//
// dqn_grad_kernel has the same two-phase structure as ppo_grad_kernel (see ppo.hip): the batch
// indices are drawn inline from the SAMPLER Philox stream and the transitions are gathered straight
// from the HBM ring into LDS -- the sampled batch is never materialised in HBM.
#include "mlp_device.h"
#include "ring_device.h"
#include "optim_device.h"
#include "select_device.h"

extern "C" int64_t rlhip_mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out);

namespace rlhip {

// per-workgroup phase stamps (thread 0) -- -DRLHIP_DQN_TIMING only, read with rlhip_debug_dqn_stamps (tools/dqn_timeline.py)
#ifdef RLHIP_DQN_TIMING
__device__ long long g_dqn_stamps[64][16];  // [..][14], [..][15]: s_memtime (shader clock) at stamps 0 and 7 -> the clock the launch ran at
#define DQN_STAMP(k)                                                                   \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (threadIdx.x == 0 && blockIdx.x < 64) {                                     \
            g_dqn_stamps[blockIdx.x][k] = wall_clock64();                              \
            if ((k) == 0) g_dqn_stamps[blockIdx.x][14] = clock64();                    \
            if ((k) == 7) g_dqn_stamps[blockIdx.x][15] = clock64();                    \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    } while (0)
#else
#define DQN_STAMP(k) \
    do {             \
    } while (0)
#endif

constexpr int DTILE = 64;
constexpr int DTHREADS = 1024, DWAVES = DTHREADS / 64;  // dqn_grad_kernel's workgroup
constexpr int DQN_MAX_BLOCKS = 512;
constexpr int DQN_FUSE_MAX_BLOCKS = 32;  // dqn_grad_kernel<..., FUSE>: partial rows the last workgroup folds alone (measured: a
                                         // 2048-sample batch still gains 0.2 us per vec-step, a 4096-sample one loses 2)

struct DqnApply {
    float* p;
    float* m;
    float* v;
    float* beta_pow;
    float* gn_out;
    unsigned int* counter;  // arrival counter in the workspace tail: zero before the first launch, re-armed here
    float grad_scale, clip_norm, lr, b1, b2, eps;
};

struct DqnArgs {
    RingRecs ring;  // record ring (ring_device.h)
    uint64_t total;
    const float* params;
    const float* tparams;
    float* partials;
    float* loss_partials;
    int h, na, act, np, num_tiles;
    int64_t batch;
    float gamma, delta, inv_b;
    uint64_t seed;
    uint32_t draw_ctr;
    const int64_t* idx;  // optional explicit flat logical indices (prioritized sampler); NULL = inline uniform draw
    float* td_out;       // optional |Q(s,a) - y| per sample (priority write-back)
    const float* isw;    // optional importance-sampling weights per sample (prioritized replay, beta > 0): loss = mean(w .* huber)
    // FUSE form only (dqn_grad_kernel<..., true>): the optimise! tail runs in the workgroup that departs last
    DqnApply ap;
    float* grad;
    float* loss;
};

// a partial-row store: plain when a later LAUNCH reads it, device-scope (write-through) when the last workgroup of THIS launch does
template <bool FUSE>
__device__ __forceinline__ void publish(float* p, float v) {
    if constexpr (FUSE) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// optimise! tail inside the gradient launch, run by the 1024 threads of the workgroup that departed last: the arithmetic of
// dqn_reduce_apply_kernel (= dqn_reduce_kernel, then clip_adam_kernel<4> of optim.hip) element for element and IN ITS ORDER --
//   gradient   four groups of partial rows, ascending inside a group, then ((g0 + g1) + g2) + g3;
//   norm       thread t owns elements t + 1024 k, sums their squares in Float64, block_sum (optim_device.h);
//   Adam       elementwise --
// so parameters, moments, gradient and loss are bit-identical to the two- and three-launch forms (tests/test_gpu_learners.py).
// np <= 4096.
__device__ __forceinline__ void dqn_fused_tail(const DqnArgs& g, double* scratch) {
    constexpr int PER_THREAD = 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = gridDim.x, np = g.np;
    const DqnApply& ap = g.ap;
    const int per = (nb + 3) / 4;
    const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
    float pr[PER_THREAD], mr[PER_THREAD], vr[PER_THREAD], gr[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {  // the Adam operands: requested first, they arrive under the partial rows
        const int i = k * 1024 + tid;
        const bool in = i < np;
        pr[k] = in ? ap.p[i] : 0.0f;
        mr[k] = in ? ap.m[i] : 0.0f;
        vr[k] = in ? ap.v[i] : 0.0f;
    }
    float lp = 0.f;
    if (w == 1 && lane < nb) lp = __hip_atomic_load(g.loss_partials + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nb <= 64
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        gr[k] = 0.0f;
        if (1024 * k < np) {  // (uniform)
            // this element's partial rows: up to 4 groups x 4 rows of independent device-scope loads per round trip.  The row
            // predicates are uniform (scalar branches around the loads); a skipped row adds +0.0f, the identity for every value
            // this sum can reach (it starts from +0, so it is never -0)
            const int i = k * 1024 + tid;
            const bool in = i < np;
            const float* col = g.partials + (in ? i : np - 1);
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int r0 = 0; r0 < per; r0 += 4) {
                float v[4][4];
#pragma unroll
                for (int grp = 0; grp < 4; ++grp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int b = grp * per + r0 + r;
                        v[grp][r] = 0.0f;
                        if (r0 + r < per && b < nb)
                            v[grp][r] = __hip_atomic_load(col + (int64_t)b * np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int grp = 0; grp < 4; ++grp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a4[grp] += v[grp][r];
            }
            gr[k] = in ? (((a4[0] + a4[1]) + a4[2]) + a4[3]) * ap.grad_scale : 0.0f;
        }
        acc += (double)gr[k] * (double)gr[k];
    }
    DQN_STAMP(8);
    if (w == 1 && g.loss != nullptr) {  // dqn_reduce_kernel's loss line: lane-strided, then the shuffle tree
        float a = 0.f + lp;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) g.loss[0] = a * g.inv_b;
    }
    {  // block_sum (optim_device.h) of 1024 threads, its tree and order
        const double ws = wave_sum_down_f64_lane0(acc);  // wave_sum's tree, in-row steps on DPP (common.h)
        if (lane == 0) scratch[w] = ws;
        __syncthreads();
        acc = 0.0;
#pragma unroll
        for (int vw = 0; vw < 16; ++vw) acc += scratch[vw];
    }
    const float gn = (float)sqrt(acc);
    DQN_STAMP(9);
    const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = k * 1024 + tid;
        if (i < np) {
            const float gi = (scale == 1.0f) ? gr[k] : gr[k] * scale;
            adam1(pr[k], gi, mr[k], vr[k], ap.lr, ap.b1, ap.b2, ap.eps, c1, c2);
            ap.p[i] = pr[k];
            ap.m[i] = mr[k];
            ap.v[i] = vr[k];
            g.grad[i] = gi;
        }
    }
    if (tid == 0) {
        if (ap.gn_out) ap.gn_out[0] = gn;
        ap.beta_pow[0] *= ap.b1;  // every thread read beta_pow before the block_sum barrier
        ap.beta_pow[1] *= ap.b2;
    }
}

// One workgroup of 1024 threads = one 64-sample tile (a 512-sample batch is 8 workgroups: the kernel is a latency chain, not a
// throughput problem -- every phase is a few hundred dependent instructions between two LDS round trips).  Round 5 rebuilt it
// around the 16-lane DPP row: a sum over sixteen lanes of a row is four VALU instructions (group_sum_dpp), where the round-1..4
// form (256 threads; partial sums per wave in LDS, a 64-lane TD line that added them up, lane groups combined through LDS) paid an
// LDS round trip per partial -- stamped timeline in profiles/r05_dqn_vec_step.md:
//   phase 0   64 lanes draw their sample (BatchSampler Philox draw or explicit index), request its 64-byte record and keep it in
//             registers; meanwhile ALL lanes stage both networks' weights into LDS (one record per hidden unit) -- one overlapped
//             round trip
//   phase 1   row = sample (wave w owns samples 4 w .. 4 w + 3), lane c of the row walks hidden units c, c + 16, ... for Q(s) and
//             Q_target(s'); row sums by DPP; then the TD target, Huber loss and dL/dq of the sample in the same lanes
//   phase 2   row = hidden unit (wave w owns units 4 w + u + 64 p), lane c of the row walks samples c, c + 16, c + 32, c + 48 of the
//             tile; the weight-gradient sums stay per lane over ALL tiles of the workgroup and meet in one DPP row sum at the end
// Gradient partials per workgroup, summed in a fixed order by dqn_reduce[_apply]_kernel / dqn_fused_tail: run-to-run deterministic.
//
// FUSE (round 5): the whole optimise! in ONE launch -- every workgroup publishes its partial row with device-scope stores and
// counts itself out; the workgroup that departs last runs dqn_fused_tail (reduce -> clip -> Adam).  Host: nb <= DQN_FUSE_MAX_BLOCKS.
// UPL: hidden units per lane in phase 2 = ceil(h / 64) rounded up to 2 or 4 (a template parameter: its accumulators are registers).
template <int NS, int ACT, bool FUSE, int UPL>
__global__ __launch_bounds__(DTHREADS) void dqn_grad_kernel(DqnArgs g) {
    // LDS: everything a later phase reads as a unit is ONE 16-byte vector:
    //   l_rec[net][j] = {W1[j, 0..3]}, {b1[j], W2[0..2, j]}, {W2[3, j], -, -, -}   (rows beyond NS / na are zeros)
    //   l_s4 / l_sn4 [sample] = state / next state, l_dL4[sample] = dL/dq
    constexpr int HMAX = 256;
    static_assert(NS <= 4 && MAXO == 4 && DWAVES * 4 == DTILE, "record layout / one row per sample");
    __shared__ float4 l_rec[2][HMAX][3];
    __shared__ float4 l_s4[DTILE], l_sn4[DTILE], l_dL4[DTILE];
    __shared__ float l_r[DTILE];
    __shared__ int32_t l_a[DTILE];
    __shared__ uint8_t l_t[DTILE];
    __shared__ float l_fin[MAXO + 1][DTILE];
    __shared__ double l_scratch[16];
    __shared__ int l_last;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane >> 4, c = lane & 15;
    const int h = g.h, na = g.na;
    const float* b2 = g.params + h * NS + h + na * h;
    const float* tb2 = g.tparams + h * NS + h + na * h;
    const int smp = 4 * w + row;  // phase 1: this row's sample of the tile

    float gw1[UPL][NS], gw2[UPL][MAXO], gb1[UPL];
#pragma unroll
    for (int p = 0; p < UPL; ++p) {
        gb1[p] = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) gw1[p][k] = 0.f;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) gw2[p][o] = 0.f;
    }
    float gb2[MAXO] = {0.f, 0.f, 0.f, 0.f};  // (every lane of a row carries its sample's sums; lane c == 0 hands them over)
    float s_loss = 0.f;
    bool staged = false;

    DQN_STAMP(0);
    for (int tile = blockIdx.x; tile < g.num_tiles; tile += gridDim.x) {
        // ---- phase 0: gather into registers ----
        float bq[MAXO], btq[MAXO];  // the output biases of both nets: requested first, in flight across the staging and the barrier
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            bq[o] = (o < na) ? b2[o] : 0.f;
            btq[o] = (o < na) ? tb2[o] : 0.f;
        }
        float gs[NS], gsn[NS], gr = 0.f;
        int32_t ga = 0;
        uint8_t gt = 0;
        if (tid < DTILE) {
            int64_t b = (int64_t)tile * DTILE + tid;
            bool valid = b < g.batch;
            int64_t fj;
            if (g.idx) {
                fj = g.idx[valid ? b : 0];
            } else {
                u32x4 wd = philox4x32_10(g.seed, (uint32_t)(valid ? b : 0), 0, g.draw_ctr, TAG_SAMPLER);
                uint64_t xr = ((uint64_t)wd.x << 32) | (uint64_t)wd.y;
                fj = (int64_t)__umul64hi(xr, g.total);
            }
            const RingTransition rt = ring_load_transition(g.ring, fj);  // one 64-byte record = one fabric request per sample
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                gs[k] = rt.s[k];
                gsn[k] = rt.sn[k];
            }
            ga = rt.a;
            gr = rt.r;
            gt = (uint8_t)rt.t;
        }
        if (!staged) {  // both networks -> LDS, while the gather is in flight
            staged = true;
            for (int q = tid; q < 2 * h; q += DTHREADS) {
                const int net = q >= h ? 1 : 0, u = q - net * h;
                const float* P = net ? g.tparams : g.params;
                float w1[4] = {0.f, 0.f, 0.f, 0.f}, w2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NS; ++k) w1[k] = P[u + h * k];
#pragma unroll
                for (int o = 0; o < MAXO; ++o)
                    if (o < na) w2[o] = P[h * NS + h + o + na * u];
                l_rec[net][u][0] = make_float4(w1[0], w1[1], w1[2], w1[3]);
                l_rec[net][u][1] = make_float4(P[h * NS + u], w2[0], w2[1], w2[2]);
                l_rec[net][u][2] = make_float4(w2[3], 0.f, 0.f, 0.f);
            }
        }
        if (tid < DTILE) {
            float a4[4] = {0.f, 0.f, 0.f, 0.f}, n4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                a4[k] = gs[k];
                n4[k] = gsn[k];
            }
            l_s4[tid] = make_float4(a4[0], a4[1], a4[2], a4[3]);
            l_sn4[tid] = make_float4(n4[0], n4[1], n4[2], n4[3]);
            l_a[tid] = ga;
            l_r[tid] = gr;
            l_t[tid] = gt;
        }
        __syncthreads();
        DQN_STAMP(1);
        // ---- phase 1: Q(s), Q_target(s') of sample `smp`, then its TD line ----
        {
            const float4 xs = l_s4[smp], xns = l_sn4[smp];
            const float sr = l_r[smp];
            const int a = l_a[smp];
            const float cont = l_t[smp] ? 0.f : 1.f;
            const float x[4] = {xs.x, xs.y, xs.z, xs.w}, xn[4] = {xns.x, xns.y, xns.z, xns.w};
            float acc[MAXO] = {0.f, 0.f, 0.f, 0.f}, acn[MAXO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int jj = c; jj < h; jj += 16) {
                const float4 a0 = l_rec[0][jj][0], a1 = l_rec[0][jj][1], c0 = l_rec[1][jj][0], c1 = l_rec[1][jj][1];
                const float wa[4] = {a0.x, a0.y, a0.z, a0.w}, wc[4] = {c0.x, c0.y, c0.z, c0.w};
                float z = a1.x, zn = c1.x;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    z = fmaf(wa[k], x[k], z);
                    zn = fmaf(wc[k], xn[k], zn);
                }
                const float hv = act_fwd_t<ACT>(z), hn = act_fwd_t<ACT>(zn);
                acc[0] = fmaf(a1.y, hv, acc[0]);
                acn[0] = fmaf(c1.y, hn, acn[0]);
                acc[1] = fmaf(a1.z, hv, acc[1]);  // (rows o >= na are staged as zeros)
                acn[1] = fmaf(c1.z, hn, acn[1]);
                if (na > 2) {  // (uniform)
                    acc[2] = fmaf(a1.w, hv, acc[2]);
                    acn[2] = fmaf(c1.w, hn, acn[2]);
                }
                if (na == 4) {
                    acc[3] = fmaf(l_rec[0][jj][2].x, hv, acc[3]);
                    acn[3] = fmaf(l_rec[1][jj][2].x, hn, acn[3]);
                }
            }
            float q[MAXO], qn[MAXO];
#pragma unroll
            for (int o = 0; o < MAXO; ++o) {
                q[o] = 0.f, qn[o] = 0.f;
                if (o < na) {  // (uniform)
                    q[o] = group_sum_dpp<16>(acc[o]) + bq[o];
                    qn[o] = group_sum_dpp<16>(acn[o]) + btq[o];
                }
            }
            const int64_t b = (int64_t)tile * DTILE + smp;
            const bool valid = b < g.batch;
            float mx = qn[0];
            for (int k = 1; k < na; ++k) mx = fmaxf(mx, qn[k]);
            const float G = sr + g.gamma * cont * mx;
            float qa = 0.f;
            for (int k = 0; k < na; ++k)
                if (k == a) qa = q[k];
            const float d = qa - G;
            const float e = fabsf(d);
            float l = (e < g.delta) ? (e * e) * 0.5f : g.delta * (e - 0.5f * g.delta);
            float gi = (e < g.delta) ? d : (d > 0.f ? g.delta : (d < 0.f ? -g.delta : 0.f));
            gi *= g.inv_b;
            if (valid && g.isw) {  // PrioritizedDQN: the weighted batch loss  mean(w .* huber(td))
                const float wis = g.isw[b];
                gi *= wis;
                l *= wis;
            }
            if (!valid) {
                gi = 0.f;
                l = 0.f;
            } else if (g.td_out && c == 0) {
                g.td_out[b] = e;
            }
            s_loss += l;
            float dl[MAXO];
#pragma unroll
            for (int o = 0; o < MAXO; ++o) {
                dl[o] = (o == a) ? gi : 0.f;
                gb2[o] += dl[o];
            }
            if (c == 0) {
                l_dL4[smp] = make_float4(dl[0], dl[1], dl[2], dl[3]);
#pragma unroll
                for (int o = 0; o < MAXO; ++o) l_fin[o][smp] = gb2[o];  // running sums over this workgroup's tiles: complete after
                l_fin[MAXO][smp] = s_loss;                              // the last one, read by wave 0 behind the tile loop
            }
        }
        __syncthreads();
        DQN_STAMP(3);
        // ---- phase 2: weight gradients of units 4 w + row + 64 p over samples c, c + 16, c + 32, c + 48 ----
        {
#pragma unroll
            for (int p = 0; p < UPL; ++p) {
                const int j = 4 * w + row + 64 * p;
                if (64 * p < h && j < h) {
                    const float4 r0 = l_rec[0][j][0], r1 = l_rec[0][j][1], r2 = l_rec[0][j][2];
                    const float rw1[4] = {r0.x, r0.y, r0.z, r0.w}, rw2[4] = {r1.y, r1.z, r1.w, r2.x};
                    const float rb1 = r1.x;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 xs = l_s4[c + 16 * i], d4 = l_dL4[c + 16 * i];
                        const float x[4] = {xs.x, xs.y, xs.z, xs.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
                        float z = rb1;
#pragma unroll
                        for (int k = 0; k < NS; ++k) z = fmaf(rw1[k], x[k], z);
                        const float hv = act_fwd_t<ACT>(z);
                        float dh = 0.f;
#pragma unroll
                        for (int o = 0; o < MAXO; ++o) {
                            const float d = dd[o];
                            gw2[p][o] = fmaf(d, hv, gw2[p][o]);
                            dh = fmaf(d, rw2[o], dh);
                        }
                        const float dz = dh * act_bwd_t<ACT>(z, hv);
                        gb1[p] += dz;
#pragma unroll
                        for (int k = 0; k < NS; ++k) gw1[p][k] = fmaf(dz, x[k], gw1[p][k]);
                    }
                }
            }
        }
        __syncthreads();  // the next tile's gather rewrites l_s4 / l_dL4
    }
    DQN_STAMP(4);
    // the sixteen sample slices of a unit meet in its row; lane c == 0 publishes
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
#pragma unroll
    for (int p = 0; p < UPL; ++p) {
        const int j = 4 * w + row + 64 * p;
        if (64 * p < h) {  // (uniform: the DPP sums run with whole rows)
#pragma unroll
            for (int k = 0; k < NS; ++k) gw1[p][k] = group_sum_dpp<16>(gw1[p][k]);
            gb1[p] = group_sum_dpp<16>(gb1[p]);
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < na) gw2[p][o] = group_sum_dpp<16>(gw2[p][o]);
            if (c == 0 && j < h) {
#pragma unroll
                for (int k = 0; k < NS; ++k) publish<FUSE>(out + j + h * k, gw1[p][k]);
                publish<FUSE>(out + h * NS + j, gb1[p]);
#pragma unroll
                for (int o = 0; o < MAXO; ++o)
                    if (o < na) publish<FUSE>(out + h * NS + h + o + na * j, gw2[p][o]);
            }
        }
    }
    // db2 and the loss: one slot per sample of the tile -> wave 0 (fixed order)
    if (tid < DTILE) {  // wave 0
        float f[MAXO + 1];
#pragma unroll
        for (int o = 0; o <= MAXO; ++o) f[o] = wave_sum_f32(l_fin[o][tid]);
        if (tid == 0) {
            for (int o = 0; o < na; ++o) publish<FUSE>(out + h * NS + h + na * h + o, f[o]);
            publish<FUSE>(g.loss_partials + blockIdx.x, f[MAXO]);
        }
    }
    DQN_STAMP(5);
    if constexpr (FUSE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's row stores have reached the L2 / fabric
        __syncthreads();
        DQN_STAMP(6);
        if (tid == 0) {
            const unsigned int prev = __hip_atomic_fetch_add(g.ap.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l_last = (prev == gridDim.x - 1) ? 1 : 0;
            if (l_last) __hip_atomic_store(g.ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
        }
        __syncthreads();
        DQN_STAMP(7);
        if (!l_last) return;
        dqn_fused_tail(g, l_scratch);
        DQN_STAMP(11);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DQN_STAMP(12);
    }
}

__global__ __launch_bounds__(256) void dqn_reduce_kernel(const float* __restrict__ partials,
                                                         const float* __restrict__ loss_partials, int nb, int np,
                                                         float* __restrict__ grad, float* __restrict__ loss,
                                                         float inv_b) {
    __shared__ float l_g[4][64];
    int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    int p = blockIdx.x * 64 + lane;
    int per = (nb + 3) / 4;
    int b0 = grp * per, b1 = min(nb, b0 + per);
    float acc = 0.f;
    if (p < np) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) acc += partials[(int64_t)b * np + p];
    }
    l_g[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && p < np) grad[p] = ((l_g[0][lane] + l_g[1][lane]) + l_g[2][lane]) + l_g[3][lane];
    if (blockIdx.x == 0 && loss != nullptr && grp == 1) {
        float a = 0.f;
        for (int b = lane; b < nb; b += 64) a += loss_partials[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) loss[0] = a * inv_b;
    }
}

// optimise! tail in ONE launch: the partial reduction of dqn_reduce_kernel, then -- in the workgroup that arrives
// last -- the body of clip_adam_kernel<4> (optim.hip): same element-to-thread mapping (thread t owns t + 1024 k),
// same Float64 block sum, same Adam expression order, so parameters / moments / gradient are bit-identical to
// "rlhip_dqn_grad_f32 then rlhip_clip_adam_f32".  Host: np <= 4096 only (the <4> regime of rlhip_clip_adam_f32).

__global__ __launch_bounds__(1024) void dqn_reduce_apply_kernel(const float* __restrict__ partials,
                                                                const float* __restrict__ loss_partials, int nb,
                                                                int np, float* __restrict__ grad,
                                                                float* __restrict__ loss, float inv_b, DqnApply ap) {
    __shared__ float l_g[4][64];
    __shared__ double scratch[16];
    __shared__ int l_last;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    if (grp < 4) {  // identical to dqn_reduce_kernel: four groups of partial blocks, then ((g0 + g1) + g2) + g3
        const int p = blockIdx.x * 64 + lane;
        const int per = (nb + 3) / 4;
        const int b0 = grp * per, b1 = min(nb, b0 + per);
        float acc = 0.f;
        if (p < np) {
#pragma unroll 8
            for (int b = b0; b < b1; ++b) acc += partials[(int64_t)b * np + p];
        }
        l_g[grp][lane] = acc;
    }
    __syncthreads();
    if (grp == 0) {
        // read back by the last workgroup through device-scope loads: a device-scope (write-through) store, drained before
        // the arrival count goes up -- instead of release / acquire fences (an L2 write-back per workgroup, an invalidate)
        const int p = blockIdx.x * 64 + lane;
        if (p < np)
            __hip_atomic_store(grad + p, ((l_g[0][lane] + l_g[1][lane]) + l_g[2][lane]) + l_g[3][lane], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (blockIdx.x == 0 && loss != nullptr && grp == 1) {
        float a = 0.f;
        for (int b = lane; b < nb; b += 64) a += loss_partials[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) loss[0] = a * inv_b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // (behind the workgroup barrier above: wave 0's gradient stores have drained)
        unsigned int prev = __hip_atomic_fetch_add(ap.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l_last = (prev == gridDim.x - 1) ? 1 : 0;
        if (l_last) __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
    }
    __syncthreads();
    if (!l_last) return;
    // ---- clip_adam_kernel<4> ----
    constexpr int PER_THREAD = 4;
    float gr[PER_THREAD], pr[PER_THREAD], mr[PER_THREAD], vr[PER_THREAD];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = k * 1024 + (int)threadIdx.x;
        const bool in = i < np;
        const float x = in ? __hip_atomic_load(grad + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * ap.grad_scale : 0.0f;
        gr[k] = x;
        pr[k] = in ? ap.p[i] : 0.0f;  // operands of the Adam phase: in flight during the block sum
        mr[k] = in ? ap.m[i] : 0.0f;
        vr[k] = in ? ap.v[i] : 0.0f;
        acc += (double)x * (double)x;
    }
    const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
    acc = block_sum_f64_dpp(acc, scratch);  // block_sum's tree, in-row steps on DPP (common.h): bit-identical
    const float gn = (float)sqrt(acc);
    const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = k * 1024 + (int)threadIdx.x;
        if (i < np) {
            const float gi = (scale == 1.0f) ? gr[k] : gr[k] * scale;
            float pi = pr[k], mi = mr[k], vi = vr[k];
            adam1(pi, gi, mi, vi, ap.lr, ap.b1, ap.b2, ap.eps, c1, c2);
            ap.p[i] = pi;
            ap.m[i] = mi;
            ap.v[i] = vi;
            grad[i] = gi;
        }
    }
    if (threadIdx.x == 0) {
        if (ap.gn_out) ap.gn_out[0] = gn;
        ap.beta_pow[0] *= ap.b1;  // every thread has read beta_pow before the block_sum barrier
        ap.beta_pow[1] *= ap.b2;
    }
}

struct RegQ {
    const float* q;
    __device__ __forceinline__ float operator()(int k) const { return q[k]; }
};

template <int NS, int H, int L, int ACT>
__global__ __launch_bounds__(256, 1) void dqn_plan_wide_kernel(const float* __restrict__ params, int na,
                                                               const float* __restrict__ obs, int64_t n,
                                                               double eps, uint64_t seed, uint32_t env_id_base,
                                                               uint32_t step, int32_t* __restrict__ actions,
                                                               float* __restrict__ q_out) {
    constexpr int HPL = H / L;
    int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t env = gl / L;
    int sub = (int)(gl % L);
    bool active = env < n;
    if (!active) env = n - 1;
    NetRegs<NS, HPL> Q;
    load_net<NS, HPL>(Q, params, H, na, sub, L);
    float x[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = obs[(int64_t)k * n + env];
    float q[MAXO];
    net_forward<NS, HPL, L, ACT>(Q, x, q);
    int32_t a = eps_greedy_select1(RegQ{q}, NoMask{}, na, eps, false, seed, env_id_base + (uint32_t)env, step);
    if (active && sub == 0) {
        actions[env] = a;
        if (q_out)
            for (int o = 0; o < na; ++o) q_out[(int64_t)o * n + env] = q[o];
    }
}

template <int NS, int ACT>
__global__ __launch_bounds__(256) void dqn_plan_scalar_kernel(const float* __restrict__ params, int h, int na,
                                                              const float* __restrict__ obs, int64_t n,
                                                              double eps, uint64_t seed, uint32_t env_id_base,
                                                              uint32_t step, int32_t* __restrict__ actions,
                                                              float* __restrict__ q_out) {
    int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    float x[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = obs[(int64_t)k * n + env];
    float q[MAXO];
    net_forward_scalar<NS, ACT>(params, h, na, x, q);
    actions[env] = eps_greedy_select1(RegQ{q}, NoMask{}, na, eps, false, seed, env_id_base + (uint32_t)env, step);
    if (q_out)
        for (int o = 0; o < na; ++o) q_out[(int64_t)o * n + env] = q[o];
}

template <int NS>
static int32_t dqn_plan_impl(const float* params, int h, int na, int act, const float* obs, int64_t n, double eps,
                             uint64_t seed, uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                             hipStream_t s) {
    bool wide = (h == 256 || h == 128 || h == 64) && n * 16 <= (int64_t)1 << 22;
#define LAUNCH_QW(H, L)                                                                                            \
    do {                                                                                                           \
        if (act == 0)                                                                                              \
            hipLaunchKernelGGL((dqn_plan_wide_kernel<NS, H, L, 0>), dim3((int)((n * L + 255) / 256)), dim3(256), 0, s, \
                               params, na, obs, n, eps, seed, env_id_base, step, actions, q_out);                  \
        else                                                                                                       \
            hipLaunchKernelGGL((dqn_plan_wide_kernel<NS, H, L, 1>), dim3((int)((n * L + 255) / 256)), dim3(256), 0, s, \
                               params, na, obs, n, eps, seed, env_id_base, step, actions, q_out);                  \
    } while (0)
    if (wide && h == 256) LAUNCH_QW(256, 16);
    else if (wide && h == 128) LAUNCH_QW(128, 8);
    else if (wide && h == 64) LAUNCH_QW(64, 4);
    else if (act == 0)
        hipLaunchKernelGGL((dqn_plan_scalar_kernel<NS, 0>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, params, h,
                           na, obs, n, eps, seed, env_id_base, step, actions, q_out);
    else
        hipLaunchKernelGGL((dqn_plan_scalar_kernel<NS, 1>), dim3((int)((n + 255) / 256)), dim3(256), 0, s, params, h,
                           na, obs, n, eps, seed, env_id_base, step, actions, q_out);
#undef LAUNCH_QW
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

#ifdef RLHIP_DQN_TIMING
int32_t rlhip_debug_dqn_stamps(long long* out1024) {
    RLHIP_CHECK_HIP(hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_dqn_stamps), sizeof(long long) * 64 * 16));
    return RLHIP_OK;
}
#endif

int64_t rlhip_dqn_workspace_bytes(int64_t ns, int64_t h, int64_t na, int64_t batch) {
    (void)batch;
    int64_t np = mlp2_nparams(ns, h, na);
    // partials | loss partials | 64 B of counters for rlhip_dqn_update_f32 (zero before the first use)
    return (int64_t)DQN_MAX_BLOCKS * (np + 1) * (int64_t)sizeof(float) + 64;
}

static int32_t dqn_grad_impl(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                             const float* target_params, int64_t batch, const int64_t* idx, float gamma,
                             float huber_delta, uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out,
                             float* loss_out, float* td_out, rlhip_stream_t stream, DqnApply* apply = nullptr,
                             const float* isw = nullptr) {
    RLHIP_REQUIRE(rb && params && target_params && workspace && grad_out, "NULL argument");
    RLHIP_REQUIRE(rb->elem_bytes == 4, "the DQN learner expects Float32 observations");
    RLHIP_REQUIRE(rb->obs_dim >= 2 && rb->obs_dim <= 4, "fused DQN kernel supports obs_dim 2..4");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "the DQN learner reads a record ring (rlhip_ring_init, ABI 2)");
    RLHIP_REQUIRE(h >= 4 && h <= 256 && h % 4 == 0, "hidden must be a multiple of 4, <= 256");
    RLHIP_REQUIRE(na >= 1 && na <= MAXO, "na must be <= 4");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE(batch >= 1, "empty batch");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    int ns = (int)rb->obs_dim;
    int64_t np = mlp2_nparams(ns, h, na);
    DqnArgs g;
    g.ring = {(const uint8_t*)rb->state, rb->capacity, rb->n_env, rb->head_sa};
    g.total = (uint64_t)rb->len_rt * (uint64_t)rb->n_env;
    g.params = params;
    g.tparams = target_params;
    g.h = (int)h;
    g.na = (int)na;
    g.act = act;
    g.np = (int)np;
    g.num_tiles = (int)((batch + DTILE - 1) / DTILE);
    g.batch = batch;
    g.gamma = gamma;
    g.delta = huber_delta;
    g.inv_b = 1.0f / (float)batch;
    g.seed = seed;
    g.draw_ctr = draw_ctr;
    g.idx = idx;
    g.td_out = td_out;
    g.isw = isw;
    int nb = g.num_tiles < DQN_MAX_BLOCKS ? g.num_tiles : DQN_MAX_BLOCKS;
    g.partials = (float*)workspace;
    g.loss_partials = g.partials + (int64_t)DQN_MAX_BLOCKS * np;
    hipStream_t s = as_stream(stream);
    // one launch for the whole optimise! while a single workgroup can fold the partial rows quickly (<= 32 rows = a 2048-sample
    // batch); beyond that the tail stays its own launch, 64 parameters per workgroup
    const bool fuse = apply != nullptr && nb <= DQN_FUSE_MAX_BLOCKS && !RLHIP_ENV_FLAG("RLHIP_DQN_NO_FUSE");
    if (fuse) {
        g.ap = *apply;
        g.ap.counter = (unsigned int*)(g.loss_partials + DQN_MAX_BLOCKS);
        g.grad = grad_out;
        g.loss = loss_out;
    }
#define LAUNCH_DG4(NS_, ACT_, FUSE_)                                                                                    \
    do {                                                                                                                 \
        if (h <= 128) hipLaunchKernelGGL((dqn_grad_kernel<NS_, ACT_, FUSE_, 2>), dim3(nb), dim3(DTHREADS), 0, s, g);     \
        else hipLaunchKernelGGL((dqn_grad_kernel<NS_, ACT_, FUSE_, 4>), dim3(nb), dim3(DTHREADS), 0, s, g);              \
    } while (0)
#define LAUNCH_DG(NS_, ACT_)                     \
    do {                                         \
        if (fuse) LAUNCH_DG4(NS_, ACT_, true);   \
        else LAUNCH_DG4(NS_, ACT_, false);       \
    } while (0)
    if (ns == 4) { if (act == 0) LAUNCH_DG(4, 0); else LAUNCH_DG(4, 1); }
    else if (ns == 3) { if (act == 0) LAUNCH_DG(3, 0); else LAUNCH_DG(3, 1); }
    else { if (act == 0) LAUNCH_DG(2, 0); else LAUNCH_DG(2, 1); }
#undef LAUNCH_DG
#undef LAUNCH_DG4
    if (fuse) {
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    if (apply) {
        apply->counter = (unsigned int*)(g.loss_partials + DQN_MAX_BLOCKS);
        hipLaunchKernelGGL(dqn_reduce_apply_kernel, dim3((int)((np + 63) / 64)), dim3(1024), 0, s, g.partials,
                           g.loss_partials, nb, (int)np, grad_out, loss_out, g.inv_b, *apply);
    } else {
        hipLaunchKernelGGL(dqn_reduce_kernel, dim3((int)((np + 63) / 64)), dim3(256), 0, s, g.partials, g.loss_partials,
                           nb, (int)np, grad_out, loss_out, g.inv_b);
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dqn_grad_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                           const float* target_params, int64_t batch, float gamma, float huber_delta,
                           uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out,
                           rlhip_stream_t stream) {
    return dqn_grad_impl(rb, h, na, act, params, target_params, batch, nullptr, gamma, huber_delta, seed, draw_ctr,
                         workspace, grad_out, loss_out, nullptr, stream);
}

int32_t rlhip_clip_adam_f32(float*, float*, float*, float*, float*, int64_t, float, float, float, float, float, float,
                            float*, rlhip_stream_t);

/* optimise!(learner, batch) in two launches: gradient partials, then reduce + clip + Adam (bit-identical to
 * rlhip_dqn_grad_f32 followed by rlhip_clip_adam_f32, which is what runs when the network has > 4096 parameters) */
int32_t rlhip_dqn_update_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, float* params,
                             const float* target_params, int64_t batch, float gamma, float huber_delta, uint64_t seed,
                             uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out, float* m, float* v,
                             float* beta_pow, float grad_scale, float max_grad_norm, float lr, float beta1, float beta2,
                             float adam_eps, float* gn_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(rb && m && v && beta_pow, "NULL argument");
    const int64_t np = mlp2_nparams(rb->obs_dim, h, na);
    if (np > 4096) {
        int32_t rc = dqn_grad_impl(rb, h, na, act, params, target_params, batch, nullptr, gamma, huber_delta, seed,
                                   draw_ctr, workspace, grad_out, loss_out, nullptr, stream);
        if (rc) return rc;
        return rlhip_clip_adam_f32(params, grad_out, m, v, beta_pow, np, grad_scale, max_grad_norm, lr, beta1, beta2,
                                   adam_eps, gn_out, stream);
    }
    DqnApply ap{params, m, v, beta_pow, gn_out, nullptr, grad_scale, max_grad_norm, lr, beta1, beta2, adam_eps};
    return dqn_grad_impl(rb, h, na, act, params, target_params, batch, nullptr, gamma, huber_delta, seed, draw_ctr,
                         workspace, grad_out, loss_out, nullptr, stream, &ap);
}

int32_t rlhip_dqn_grad_idx_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                               const float* target_params, int64_t batch, const int64_t* idx, float gamma,
                               float huber_delta, void* workspace, float* grad_out, float* loss_out, float* td_out,
                               rlhip_stream_t stream) {
    RLHIP_REQUIRE(idx != nullptr && rb != nullptr, "idx / ring is NULL");
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    return dqn_grad_impl(rb, h, na, act, params, target_params, batch, idx, gamma, huber_delta, 0, 0, workspace,
                         grad_out, loss_out, td_out, stream);
}

int32_t rlhip_dqn_grad_idx_w_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                                 const float* target_params, int64_t batch, const int64_t* idx, const float* weights,
                                 float gamma, float huber_delta, void* workspace, float* grad_out, float* loss_out,
                                 float* td_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(idx != nullptr && weights != nullptr && rb != nullptr, "idx / weights / ring is NULL");
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    return dqn_grad_impl(rb, h, na, act, params, target_params, batch, idx, gamma, huber_delta, 0, 0, workspace,
                         grad_out, loss_out, td_out, stream, nullptr, weights);
}

int32_t rlhip_dqn_plan_f32(const float* params, int64_t ns, int64_t h, int64_t na, int32_t act, const float* obs,
                           int64_t n, double eps, uint64_t seed, uint32_t env_id_base, uint32_t step,
                           int32_t* actions, float* q_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && obs && actions, "NULL argument");
    RLHIP_REQUIRE(ns >= 2 && ns <= 4, "obs dim must be 2..4");
    RLHIP_REQUIRE(na >= 1 && na <= MAXO, "na must be <= 4");
    RLHIP_REQUIRE(h >= 1, "bad hidden size");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    if (n == 0) return RLHIP_OK;
    hipStream_t s = as_stream(stream);
    if (ns == 4) return dqn_plan_impl<4>(params, (int)h, (int)na, act, obs, n, eps, seed, env_id_base, step, actions, q_out, s);
    if (ns == 3) return dqn_plan_impl<3>(params, (int)h, (int)na, act, obs, n, eps, seed, env_id_base, step, actions, q_out, s);
    return dqn_plan_impl<2>(params, (int)h, (int)na, act, obs, n, eps, seed, env_id_base, step, actions, q_out, s);
}

}  // extern "C"
