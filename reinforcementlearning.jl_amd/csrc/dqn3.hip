// dqn3.hip -- the DQN hot path for the blog's three-layer Q-network, hidden x hidden layer on the MFMA.
//
//   Chain(Dense(ns, 128, act), Dense(128, 128, act), Dense(128, na))
//   (docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15126-15128)
//
// What it replaces: the same reference code as dqn.hip -- forward(learner, x) = model(x)
// (RLCore/src/policies/learners/flux_approximator.jl:43) inside plan!(QBasedPolicy) (q_based_policy.jl:30-32)
// + EpsilonGreedyExplorer (explorers/epsilon_greedy_explorer.jl:108-112), and optimise!(learner, batch) of the
// removed Zoo DQN learner (y = r + gamma (1 - t) max_a' Qt(s', a'), Flux.Losses.huber_loss, Zygote backward).
//
// Precision contract (BASELINE.json: "MFMA used only for the dense ... MLP GEMMs"; f32 master weights):
//   layer 1 (K = ns <= 4) and the head (N = na <= 4) stay f32 on the VALU -- far below an MFMA tile;
//   the 128 x 128 hidden layer runs on v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (RNE), f32 accumulate;
//   backward: dz2 is rounded to bf16 for dW2 = dz2^T h1 and dh1 = dz2 W2 (both MFMA); every bias / first-layer
//   / head gradient is f32.  The oracle (oracle/rlo_mlp3.c) applies the same roundings, so the only difference
//   left is the MFMA's internal summation order (tolerance stated in tests/test_gpu_dqn3.py).
//
// One workgroup (4 waves) owns a tile of 128 samples; wave w owns rows 32w..32w+31 of every row-parallel GEMM.
// All three GEMMs are "A rows x B rows, both reduction-contiguous", so every MFMA fragment is one 16-byte read:
//   forward   Z2[r][j]  = sum_k H1[r][k]  W2[j][k]    A = H1 tile  (LDS [r][k])   B = W2jk (global, L2-hot)
//   backward  dH1[r][k] = sum_j dZ2[r][j] W2[j][k]    A = dZ2 tile (LDS [r][j])   B = W2kj (global)
//   backward  dW2[j][k] = sum_r dZ2[r][j] H1[r][k]    A = H1^T     (LDS [k][r])   B = dZ2^T (LDS [j][r])
// LDS tiles are bf16 with a 272-byte row pitch (68 dwords: conflict-free ds_read_b128 over 16-lane phases).
// The transposed copies are produced where the values are produced (VALU), never by an LDS transpose pass.
// Roofline: MFMA for the three 128^3 GEMMs per tile (3 * 2 * 128^3 = 12.6 MFLOP); at DQN batch sizes (<= 4096
// samples = 32 workgroups) the launch is latency-bound -- measured numbers in profiles/ and DESIGN.md.
#include "mfma_common.h"
#include "mlp_device.h"
#include "ring_device.h"
#include "select_device.h"

namespace rlhip {

constexpr int D3_MAX_BLOCKS = 1024;

// Phase timestamps of block 0 (compile with -DRLHIP_D3_TIMING; read with rlhip_debug_d3_stamps) -- dev only.
#ifdef RLHIP_D3_TIMING
__device__ long long g_d3_stamps[32];
#define D3_STAMP(k)                                                              \
    do {                                                                         \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_d3_stamps[k] = wall_clock64(); \
    } while (0)
#else
#define D3_STAMP(k) \
    do {            \
    } while (0)
#endif

}  // namespace rlhip

#include <type_traits>

#include "act_device.h"
#include "mlp3_device.h"
#include "optim_device.h"

namespace rlhip {

struct RegQ3 {
    const float* q;
    __device__ __forceinline__ float operator()(int k) const { return q[k]; }
};

// ------------------------------------------------------------------------------ forward / plan!
template <int NS, int NA, int ACT>
__global__ __launch_bounds__(256) void mlp3_plan_kernel(const float* __restrict__ params,
                                                        const uint16_t* __restrict__ packed,
                                                        const float* __restrict__ obs, int64_t n, double eps,
                                                        uint64_t seed, uint32_t env_id_base, uint32_t step,
                                                        int32_t* __restrict__ actions, float* __restrict__ q_out) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    constexpr int na = NA;
    float* l_x = reinterpret_cast<float*>(smem3);                       // [4][TR]
    float* l_q = l_x + 4 * TR;                                          // [MAXO][TR]
    float* l_w = l_q + MAXO * TR;                                       // [SMALLW]
    uint16_t* l_A = reinterpret_cast<uint16_t*>(l_w + SMALLW);          // [TR][LDH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every global load of the kernel is issued up front: the hidden layer's B fragments (registers), the
    // observations, the small tensors -- one overlapped round trip instead of three in a row
    bf16x8 bw[H3 / 16][4];
    load_w2_fragments(packed, lane, bw);
    const int64_t e0 = (int64_t)blockIdx.x * TR;
    float xin[NS];
    if (tid < TR) {
        int64_t e = e0 + tid;
        if (e >= n) e = n - 1;
#pragma unroll
        for (int i = 0; i < NS; ++i) xin[i] = obs[(int64_t)i * n + e];
    }
    const Mlp3 m = stage_small_weights(params, NS, na, l_w, tid);
    if (tid < TR) {
#pragma unroll
        for (int i = 0; i < NS; ++i) l_x[i * TR + tid] = xin[i];
    }
    __syncthreads();
    layer1_to_lds<NS, ACT>(m, l_x, l_A, nullptr, tid);
    __syncthreads();
    f32x16 h2[4];
    layer2_regs<ACT>(l_A, bw, m.b2, w, lane, h2);
    head_to_lds<NA>(m, h2, w, lane, l_q);
    __syncthreads();
    if (tid < TR && e0 + tid < n) {
        const int64_t e = e0 + tid;
        float q[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) q[o] = (o < na) ? l_q[o * TR + tid] : 0.0f;
        if (q_out)
            for (int o = 0; o < na; ++o) q_out[(int64_t)o * n + e] = q[o];
        if (actions) {
            actions[e] = eps_greedy_select1(RegQ3{q}, NoMask{}, na, eps, false, seed, env_id_base + (uint32_t)e, step);
        }
    }
}

// The same forward for SMALL batches (n <= 2^15): 32 rows per workgroup, the four waves share the row tile and split the
// 128 hidden units (8 MFMAs per wave instead of 32; the wave's 8 B fragments come straight from the fragment-ordered
// global copy into registers at kernel entry), layer 1 is 16 units per thread, the head is folded from an f32 H2 tile in
// LDS by all 256 threads.  4096 envs are 128 workgroups instead of 32.  H2 is bit-identical to the 128-row kernel (same
// bf16 roundings, same k order); the head sums run in a different fixed order (inside the stated tolerance).
constexpr int P32 = 32;
constexpr int LDH2 = H3 + 4;  // f32 pitch of the H2 tile
//
// TAIL (round 5): NoActTail = plan! alone; ActTail<P> = the lane that selected env e's action goes on with act!(env, a) (auto-reset)
// and push!(trajectory, (state = s', action, reward, terminal)) -- env_act_push1, the body of env_act_push_kernel, same slots:
// bit-identical to the two launches, one launch and one round trip of the action array fewer per vec-step.  `obs` is then
// read (this step's observation) and rewritten (the next one's) by the same lane.
struct NoActTail {};
template <class P>
struct ActTail {
    P p;
    EnvArrays<float> st;
    uint64_t env_seed;
    ActRing rb;
    float* obs_out;
    float* last_obs;
};
template <int NS, int NA, int ACT, class TAIL>
__global__ __launch_bounds__(256) void mlp3_plan32_kernel(const float* __restrict__ params,
                                                          const uint16_t* __restrict__ packed,
                                                          const float* obs, int64_t n, double eps,
                                                          uint64_t seed, uint32_t env_id_base, uint32_t step,
                                                          int32_t* __restrict__ actions, float* __restrict__ q_out, TAIL tail) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    constexpr int na = NA;
    float* l_x = reinterpret_cast<float*>(smem3);               // [4][P32]
    float* l_part = l_x + 4 * P32;                              // [8 parts][4][P32]
    float* l_h2 = l_part + 8 * 4 * P32;                         // [P32][LDH2]
    float* l_w = l_h2 + P32 * LDH2;                             // [SMALLW]
    uint16_t* l_H = reinterpret_cast<uint16_t*>(l_w + SMALLW);  // [P32][LDH] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    bf16x8 bwf[H3 / 16];  // this wave's column tile t = w of every k-step
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks)
        bwf[ks] = *reinterpret_cast<const bf16x8*>(packed + ((ks * 4 + w) * 64 + lane) * 8);
    const int64_t e0 = (int64_t)blockIdx.x * P32;
    float xin[NS];
    if (tid < P32) {
        int64_t e = e0 + tid;
        if (e >= n) e = n - 1;
#pragma unroll
        for (int i = 0; i < NS; ++i) xin[i] = obs[(int64_t)i * n + e];
    }
    const Mlp3 m = stage_small_weights(params, NS, na, l_w, tid);
    if (tid < P32) {
#pragma unroll
        for (int i = 0; i < NS; ++i) l_x[i * P32 + tid] = xin[i];
    }
    __syncthreads();
    const int row1 = tid & 31, u0 = 16 * (tid >> 5), part = tid >> 5;
    {  // layer 1: the fmaf chain of layer1_to_lds, 16 units per thread
        float x[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = l_x[i * P32 + row1];
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
            float hv[8];
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const int u = u0 + 8 * h8 + 4 * q4;
                const float4 b = *reinterpret_cast<const float4*>(m.b1 + u);
                float z[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float4 wv = *reinterpret_cast<const float4*>(m.W1 + u + H3 * i);
                    z[0] = fmaf(wv.x, x[i], z[0]);
                    z[1] = fmaf(wv.y, x[i], z[1]);
                    z[2] = fmaf(wv.z, x[i], z[2]);
                    z[3] = fmaf(wv.w, x[i], z[3]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) hv[4 * q4 + c] = act_fwd_t<ACT>(z[c]);
            }
            *reinterpret_cast<uint4*>(l_H + row1 * LDH + u0 + 8 * h8) = pack8_bf16(hv);
        }
    }
    __syncthreads();
    {  // layer 2: this wave's 32 output columns
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        const uint16_t* ap = l_H + r * LDH + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < H3 / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwf[ks], acc, 0, 0, 0);
        }
        const int col = 32 * w + r;
        const float bv = m.b2[col];
#pragma unroll
        for (int q = 0; q < 16; ++q) l_h2[mfma_row(q, kb) * LDH2 + col] = act_fwd_t<ACT>(acc[q] + bv);
    }
    __syncthreads();
    {  // head: thread (row1, part) folds 16 columns of its H2 row for every output
        float pq[NA];
#pragma unroll
        for (int o = 0; o < NA; ++o) pq[o] = 0.0f;
        const float* hp = l_h2 + row1 * LDH2 + 16 * part;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = *reinterpret_cast<const float4*>(hp + 4 * c4);
            const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 16 * part + 4 * c4 + c;
#pragma unroll
                for (int o = 0; o < NA; ++o) pq[o] = fmaf(m.W3[o + NA * j], hv[c], pq[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < NA; ++o) l_part[(part * 4 + o) * P32 + row1] = pq[o];
    }
    __syncthreads();
    if (tid < P32 && e0 + tid < n) {
        const int64_t e = e0 + tid;
        float q[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            float acc = 0.0f;
            if (o < na) {
                acc = l_part[o * P32 + tid];
#pragma unroll
                for (int pp = 1; pp < 8; ++pp) acc += l_part[(pp * 4 + o) * P32 + tid];
                acc += m.b3[o];
            }
            q[o] = acc;
        }
        if (q_out)
            for (int o = 0; o < na; ++o) q_out[(int64_t)o * n + e] = q[o];
        if (actions) {
            const int32_t a = eps_greedy_select1(RegQ3{q}, NoMask{}, na, eps, false, seed, env_id_base + (uint32_t)e, step);
            actions[e] = a;
            if constexpr (!std::is_same<TAIL, NoActTail>::value)
                env_act_push1(tail.p, tail.st, n, e, a, tail.env_seed, env_id_base, tail.rb, tail.obs_out, tail.last_obs);
        }
    }
}

constexpr size_t PLAN32_LDS = (4 * P32 + 8 * 4 * P32 + P32 * LDH2 + SMALLW) * sizeof(float) + P32 * LDH * sizeof(uint16_t);

// ---------------------------------------------------------------------------------- gradient
struct Dqn3Args {
    RingRecs ring;  // record ring (ring_device.h)
    uint64_t total;
    const int64_t* idx;  // optional explicit flat logical indices (prioritized sampler); NULL = inline uniform draw
    const float* params;
    const float* tparams;
    const uint16_t* packed;   // online net: W2jk | W2kj
    const uint16_t* tpacked;  // target net
    float* partials;          // [nb][np]
    float* loss_partials;     // [nb]
    float* td_out;            // optional |Q(s,a) - y| per sample (priority write-back), may be NULL
    const float* isw;         // optional importance-sampling weights per sample (prioritized replay): loss = mean(w .* huber)
    int na, np;
    int num_tiles;            // dqn3_grad32_kernel: 32-sample tiles of the batch (a workgroup walks several)
    int64_t batch;
    float gamma, delta, inv_b;
    uint64_t seed;
    uint32_t draw_ctr;
    int wt_rows;              // dqn3_grad32_kernel: partial rows stored write-through (many rows) or plainly (few): common.h store_row
};

template <int NS, int NA, int ACT>
__global__ __launch_bounds__(256) void dqn3_grad_kernel(Dqn3Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    float* l_x = reinterpret_cast<float*>(smem3);       // [4][TR]
    float* l_xn = l_x + 4 * TR;                         // [4][TR]
    float* l_q = l_xn + 4 * TR;                         // [MAXO][TR]
    float* l_qn = l_q + MAXO * TR;                      // [MAXO][TR]
    float* l_dq = l_qn + MAXO * TR;                     // [MAXO][TR]
    float* l_r = l_dq + MAXO * TR;                      // [TR]
    float* l_small = l_r + TR;                          // [2][8]
    int32_t* l_a = reinterpret_cast<int32_t*>(l_small + 16);  // [TR]
    int32_t* l_t = l_a + TR;                                  // [TR]
    float* l_red = reinterpret_cast<float*>(l_t + TR);        // [4][5][H3]
    float* l_w = l_red + 4 * 5 * H3;                          // [2][SMALLW] online / target small weights
    uint16_t* l_A = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLW);    // H1 [r][k], later dZ2 [r][j]
    uint16_t* l_B = l_A + TILE_ELEMS;                                 // H1^T [k][r]
    uint16_t* l_C = l_B + TILE_ELEMS;                                 // dZ2^T [j][r]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int na = NA;
    const Mlp3 m = stage_small_weights(g.params, NS, na, l_w, tid);
    const Mlp3 mt = stage_small_weights(g.tparams, NS, na, l_w + SMALLW, tid);
    const int tile = blockIdx.x;
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
    const int ob3 = H3 * NS + H3 + H3 * H3 + H3 + na * H3;

    D3_STAMP(0);
    // the target network's hidden-layer B fragments: requested now, consumed after the gather and layer 1
    bf16x8 bw[H3 / 16][4];
    load_w2_fragments(g.tpacked, lane, bw);
    // ---- sample + gather the tile's transitions straight from the HBM ring ----
    if (tid < TR) {
        int64_t b = (int64_t)tile * TR + tid;
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
            l_x[k * TR + tid] = rt.s[k];
            l_xn[k * TR + tid] = rt.sn[k];
        }
        l_a[tid] = rt.a;
        l_r[tid] = rt.r;
        l_t[tid] = rt.t;
    }
    __syncthreads();

    D3_STAMP(1);
    // ---- target network on s' ----
    f32x16 h2[4];
    layer1_to_lds<NS, ACT>(mt, l_xn, l_A, nullptr, tid);
    __syncthreads();
    D3_STAMP(2);
    layer2_regs<ACT>(l_A, bw, mt.b2, w, lane, h2);
    load_w2_fragments(g.packed, lane, bw);  // the online network's, in flight during the head and its layer 1
    D3_STAMP(3);
    head_to_lds<NA>(mt, h2, w, lane, l_qn);
    D3_STAMP(4);
    __syncthreads();  // all waves are done reading l_A

    // ---- online network on s (h2 stays in registers for the backward pass) ----
    layer1_to_lds<NS, ACT>(m, l_x, l_A, l_B, tid);
    __syncthreads();
    D3_STAMP(5);
    layer2_regs<ACT>(l_A, bw, m.b2, w, lane, h2);
    D3_STAMP(6);
    head_to_lds<NA>(m, h2, w, lane, l_q);
    D3_STAMP(7);
    __syncthreads();

    // ---- TD target, Huber loss, dL/dq per sample ----
    if (tid < TR) {
        const int s = tid;
        const int64_t b = (int64_t)tile * TR + s;
        const bool valid = b < g.batch;
        float mx = l_qn[s];
        for (int k = 1; k < na; ++k) mx = fmaxf(mx, l_qn[k * TR + s]);
        float cont = l_t[s] ? 0.f : 1.f;
        float G = l_r[s] + g.gamma * cont * mx;
        int a = l_a[s];
        float qa = 0.f;
        for (int k = 0; k < na; ++k)
            if (k == a) qa = l_q[k * TR + s];
        float d = qa - G;
        float e = fabsf(d);
        float l = (e < g.delta) ? (e * e) * 0.5f : g.delta * (e - 0.5f * g.delta);
        float gi = (e < g.delta) ? d : (d > 0.f ? g.delta : (d < 0.f ? -g.delta : 0.f));
        gi *= g.inv_b;
        if (valid && g.isw) {
            const float wis = g.isw[b];
            gi *= wis;
            l *= wis;
        }
        if (!valid) {
            gi = 0.f;
            l = 0.f;
        }
        if (valid && g.td_out) g.td_out[b] = e;
        float red[MAXO + 1];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            float dl = (o == a) ? gi : 0.f;
            l_dq[o * TR + s] = dl;
            red[o] = dl;
        }
        red[MAXO] = l;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int o = 0; o <= MAXO; ++o) red[o] += __shfl_down(red[o], off, 64);
        if (lane == 0)
#pragma unroll
            for (int o = 0; o <= MAXO; ++o) l_small[w * 8 + o] = red[o];
    }
    __syncthreads();
    if (tid == 0) {
        for (int o = 0; o < na; ++o) store_wt(&out[ob3 + o], l_small[o] + l_small[8 + o]);
        store_wt(&g.loss_partials[blockIdx.x], l_small[MAXO] + l_small[8 + MAXO]);
    }

    D3_STAMP(8);
    mlp3_backward_tile<NS, NA, ACT>(m, g.packed + H3 * H3, h2, l_x, l_dq, l_red, l_A, l_B, l_C, out, tid);
    D3_STAMP(11);
}

// ---- the learner step for SMALL batches: 32 samples per workgroup ------------------------------------------------------
// A 512-sample batch is 4 workgroups of the 128-row kernel above -- 1.5 % of the CUs, 22 us of latency.  Here the four
// waves of a workgroup share one 32-sample tile and split the 128 hidden units / columns everywhere: a wave runs 8 MFMAs
// per GEMM (target forward, online forward, dH1 = dZ2 W2, dW2^T = H1^T dZ2) instead of 32, column sums of the backward
// pass never cross a wave (its 32 columns are its own), and the heads are folded from an f32 H2 tile by all 256
// threads.  Same bf16 roundings and k order as the 128-row kernel inside every GEMM; the sums over samples and over
// hidden units run in a different fixed order, and there are four times as many partial rows for d3_reduce / d3_apply.
constexpr int G32 = 32;
constexpr int D3_GRAD32_BLOCKS = 512;  // persistent workgroups of the 32-sample kernel = rows of partial gradients
constexpr int LDT = G32 + 8;  // bf16 pitch of the transposed tiles [k][sample]
// OCC = workgroups per CU asked of the compiler: 1 (280 VGPRs, no spills: the latency-bound small batches) or 2 (256 VGPRs,
// ~30 spilled: two workgroups interleave their phases, which wins from 65536 samples up)
// per-phase wall-clock stamps of the LAST tile of workgroup 0 (steady state when it has several), thread 0 -- -DRLHIP_D3_TIMING only
// (tools/d3g32_timeline.py)
#ifdef RLHIP_D3_TIMING
#define G32_STAMP(k)                                                                                             \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tile + (int)gridDim.x >= g.num_tiles) g_d3_stamps[12 + (k)] = wall_clock64(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#else
#define G32_STAMP(k) \
    do {             \
    } while (0)
#endif
template <int NS, int NA, int ACT, int OCC>
__global__ __launch_bounds__(256, OCC) void dqn3_grad32_kernel(Dqn3Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    constexpr int na = NA;
    float* l_x = reinterpret_cast<float*>(smem3);                     // [4][G32]
    float* l_xn = l_x + 4 * G32;                                      // [4][G32]
    float* l_dq = l_xn + 4 * G32;                                     // [MAXO][G32]
    float* l_r = l_dq + MAXO * G32;                                   // [G32]
    float* l_part = l_r + G32;                                        // [8][4][G32] online head partials
    float* l_partn = l_part + 8 * 4 * G32;                            // [8][4][G32] target head partials
    int32_t* l_a = reinterpret_cast<int32_t*>(l_partn + 8 * 4 * G32); // [G32]
    int32_t* l_t = l_a + G32;                                         // [G32]
    float* l_h2 = reinterpret_cast<float*>(l_t + G32);                // [G32][LDH2] f32 H2 of the current net
    float* l_w = l_h2 + G32 * LDH2;                                   // [2][SMALLW]
    uint16_t* l_H = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLW);    // H1 [row][k]
    uint16_t* l_HT = l_H + G32 * LDH;                                 // online H1^T [k][row]
    uint16_t* l_Z = l_HT + H3 * LDT;                                  // dZ2 [row][j]
    uint16_t* l_ZT = l_Z + G32 * LDH;                                 // dZ2^T [j][row]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col = 32 * w + r;  // the hidden unit / column this lane owns in every D-layout phase
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
    const int oW1 = 0, ob1 = H3 * NS, oW2 = ob1 + H3, ob2 = oW2 + H3 * H3, oW3 = ob2 + H3, ob3 = oW3 + na * H3;

    const Mlp3 m = stage_small_weights(g.params, NS, na, l_w, tid);
    const Mlp3 mt = stage_small_weights(g.tparams, NS, na, l_w + SMALLW, tid);
    const int row1 = tid & 31, u0 = 16 * (tid >> 5), part = tid >> 5;
    // gradient accumulators of this workgroup over ALL its tiles (tile, tile + gridDim.x, ...): every sum below is per
    // lane already (a lane owns its column in each D-layout phase), so a persistent workgroup costs no extra traffic and
    // writes ONE partial row however large the batch is
    float acc_b2 = 0.0f, acc_w3[MAXO] = {0.f, 0.f, 0.f, 0.f}, acc_b1 = 0.0f, acc_w1[NS], acc_b3[MAXO + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NS; ++i) acc_w1[i] = 0.0f;
    f32x16 dw[4];
    zero_acc(dw);
    bf16x8 bwf[H3 / 16];  // B fragments of this wave's column tile
    for (int tile = blockIdx.x; tile < g.num_tiles; tile += gridDim.x) {
    G32_STAMP(0);
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks)  // target net first
        bwf[ks] = *reinterpret_cast<const bf16x8*>(g.tpacked + ((ks * 4 + w) * 64 + lane) * 8);
    // ---- sample + gather ----
    float gs[NS], gsn[NS], gr = 0.f;
    int32_t ga = 0, gt = 0;
    if (tid < G32) {
        int64_t b = (int64_t)tile * G32 + tid;
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
        gt = (int32_t)rt.t;
    }
    if (tid < G32) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            l_x[k * G32 + tid] = gs[k];
            l_xn[k * G32 + tid] = gsn[k];
        }
        l_a[tid] = ga;
        l_r[tid] = gr;
        l_t[tid] = gt;
    }
    G32_STAMP(1);
    __syncthreads();
    G32_STAMP(2);

    // layer 1 of one net for (row1, units u0 .. u0 + 15): bf16 into l_H [row][k] and, for the online net, l_HT [k][row]
    auto layer1 = [&](const Mlp3& mm, const float* lx, bool transposed_too) {
        float x[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = lx[i * G32 + row1];
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
            float hv[8];
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const int u = u0 + 8 * h8 + 4 * q4;
                const float4 b = *reinterpret_cast<const float4*>(mm.b1 + u);
                float z[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float4 wv = *reinterpret_cast<const float4*>(mm.W1 + u + H3 * i);
                    z[0] = fmaf(wv.x, x[i], z[0]);
                    z[1] = fmaf(wv.y, x[i], z[1]);
                    z[2] = fmaf(wv.z, x[i], z[2]);
                    z[3] = fmaf(wv.w, x[i], z[3]);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) hv[4 * q4 + c] = act_fwd_t<ACT>(z[c]);
            }
            *reinterpret_cast<uint4*>(l_H + row1 * LDH + u0 + 8 * h8) = pack8_bf16(hv);
            if (transposed_too) {
#pragma unroll
                for (int c = 0; c < 8; ++c) l_HT[(u0 + 8 * h8 + c) * LDT + row1] = f32_to_bf16_rne(hv[c]);
            }
        }
    };
    // layer 2 for this wave's 32 columns: activated H2 in the D layout (+ the f32 tile for the head)
    auto layer2w = [&](const float* b2, f32x16& h2) {
#pragma unroll
        for (int q = 0; q < 16; ++q) h2[q] = 0.0f;
        const uint16_t* ap = l_H + r * LDH + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < H3 / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
            h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwf[ks], h2, 0, 0, 0);
        }
        const float bv = b2[col];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            h2[q] = act_fwd_t<ACT>(h2[q] + bv);
            l_h2[mfma_row(q, kb) * LDH2 + col] = h2[q];
        }
    };
    // head partials of thread (row1, part) over 16 columns of its H2 row
    auto head = [&](const Mlp3& mm, float* lp) {
        float pq[NA];
#pragma unroll
        for (int o = 0; o < NA; ++o) pq[o] = 0.0f;
        const float* hp = l_h2 + row1 * LDH2 + 16 * part;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = *reinterpret_cast<const float4*>(hp + 4 * c4);
            const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 16 * part + 4 * c4 + c;
#pragma unroll
                for (int o = 0; o < NA; ++o) pq[o] = fmaf(mm.W3[o + NA * j], hv[c], pq[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < NA; ++o) lp[(part * 4 + o) * G32 + row1] = pq[o];
    };

    // ---- target network on s' ----
    f32x16 h2;
    layer1(mt, l_xn, false);
    G32_STAMP(3);
    __syncthreads();
    G32_STAMP(4);
    layer2w(mt.b2, h2);
    G32_STAMP(5);
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks)  // the online network's fragments: in flight during the head and its layer 1
        bwf[ks] = *reinterpret_cast<const bf16x8*>(g.packed + ((ks * 4 + w) * 64 + lane) * 8);
    __syncthreads();
    G32_STAMP(6);
    head(mt, l_partn);
    G32_STAMP(7);
    // ---- online network on s (its H2 stays in registers for the backward pass) ----
    layer1(m, l_x, true);  // l_H was last read before the barrier above
    G32_STAMP(8);
    __syncthreads();       // also: every thread is done reading the target's H2 tile
    G32_STAMP(9);
    layer2w(m.b2, h2);
    G32_STAMP(10);
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks)  // W2kj fragments (ks over j, this wave's k tile) for dH1
        bwf[ks] = *reinterpret_cast<const bf16x8*>(g.packed + H3 * H3 + ((ks * 4 + w) * 64 + lane) * 8);
    __syncthreads();
    G32_STAMP(11);
    head(m, l_part);
    G32_STAMP(12);
    __syncthreads();
    G32_STAMP(13);
    // ---- TD target, Huber loss, dL/dq per sample ----
    if (tid < G32) {
        const int s = tid;
        const int64_t b = (int64_t)tile * G32 + s;
        const bool valid = b < g.batch;
        float q[MAXO], qn[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            q[o] = 0.f, qn[o] = 0.f;
            if (o < na) {
                float a1 = l_part[o * G32 + s], a2 = l_partn[o * G32 + s];
#pragma unroll
                for (int pp = 1; pp < 8; ++pp) {
                    a1 += l_part[(pp * 4 + o) * G32 + s];
                    a2 += l_partn[(pp * 4 + o) * G32 + s];
                }
                q[o] = a1 + m.b3[o];
                qn[o] = a2 + mt.b3[o];
            }
        }
        float mx = qn[0];
        for (int k = 1; k < na; ++k) mx = fmaxf(mx, qn[k]);
        float cont = l_t[s] ? 0.f : 1.f;
        float G = l_r[s] + g.gamma * cont * mx;
        int a = l_a[s];
        float qa = 0.f;
        for (int k = 0; k < na; ++k)
            if (k == a) qa = q[k];
        float d = qa - G;
        float e = fabsf(d);
        float l = (e < g.delta) ? (e * e) * 0.5f : g.delta * (e - 0.5f * g.delta);
        float gi = (e < g.delta) ? d : (d > 0.f ? g.delta : (d < 0.f ? -g.delta : 0.f));
        gi *= g.inv_b;
        if (valid && g.isw) {
            const float wis = g.isw[b];
            gi *= wis;
            l *= wis;
        }
        if (!valid) {
            gi = 0.f;
            l = 0.f;
        }
        if (valid && g.td_out) g.td_out[b] = e;
        // db3 and the loss stay per lane (= per row slot of the tile) over ALL tiles of this workgroup and meet once behind the
        // tile loop: the per-tile form cost 25 ds_bpermute round trips in the one wave every other wave waits for (0.7 of the
        // tile's 6.6 us, profiles/r05_tile_budget.md section 2)
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            float dl = (o == a) ? gi : 0.f;
            l_dq[o * G32 + s] = dl;
            acc_b3[o] += dl;
        }
        acc_b3[MAXO] += l;
    }
    G32_STAMP(14);
    __syncthreads();
    G32_STAMP(15);
    // ---- head backward in the D layout (this wave's 32 columns): dW3, db2, dZ2 -> bf16 tiles [row][j] and [j][row] ----
    {
        float w3[MAXO], accw[MAXO], accb = 0.0f;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            w3[o] = (o < na) ? m.W3[o + na * col] : 0.0f;
            accw[o] = 0.0f;
        }
        uint16_t pk[4];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = mfma_row(q, kb);
            const float hv = h2[q];
            float dh = 0.0f;
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < na) {
                    const float dqv = l_dq[o * G32 + row];
                    accw[o] = fmaf(dqv, hv, accw[o]);
                    dh = fmaf(dqv, w3[o], dh);
                }
            const float dz = dh * act_bwd_t<ACT>(hv, hv);  // relu: h2 > 0 <=> z2 > 0
            accb += dz;
            const uint16_t dzb = f32_to_bf16_rne(dz);
            l_Z[row * LDH + col] = dzb;
            pk[q & 3] = dzb;
            if ((q & 3) == 3) {
                uint2 v2;
                v2.x = (uint32_t)pk[0] | ((uint32_t)pk[1] << 16);
                v2.y = (uint32_t)pk[2] | ((uint32_t)pk[3] << 16);
                *reinterpret_cast<uint2*>(l_ZT + col * LDT + 8 * (q >> 2) + 4 * kb) = v2;  // rows 8 (q >> 2) + 4 kb .. + 3
            }
        }
        accb += __shfl_xor(accb, 32, 64);  // the other 16 rows
#pragma unroll
        for (int o = 0; o < MAXO; ++o) accw[o] += __shfl_xor(accw[o], 32, 64);
        acc_b2 += accb;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) acc_w3[o] += accw[o];
    }
    G32_STAMP(16);
    __syncthreads();
    G32_STAMP(17);
    // ---- dH1 = dZ2 W2 (MFMA, this wave's 32 hidden units k), dz1 = dH1 act'(z1), dW1 / db1 ----
    {
        f32x16 dh1;
#pragma unroll
        for (int q = 0; q < 16; ++q) dh1[q] = 0.0f;
        const uint16_t* ap = l_Z + r * LDH + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < H3 / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
            dh1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwf[ks], dh1, 0, 0, 0);
        }
        float w1[NS], acc1[NS], accb = 0.0f;
        const float bb = m.b1[col];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            w1[i] = m.W1[col + H3 * i];
            acc1[i] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = mfma_row(q, kb);
            float x[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = l_x[i * G32 + row];
            float z = bb;
#pragma unroll
            for (int i = 0; i < NS; ++i) z = fmaf(w1[i], x[i], z);
            const float hv = act_fwd_t<ACT>(z);
            const float dz = dh1[q] * act_bwd_t<ACT>(z, hv);
            accb += dz;
#pragma unroll
            for (int i = 0; i < NS; ++i) acc1[i] = fmaf(dz, x[i], acc1[i]);
        }
        accb += __shfl_xor(accb, 32, 64);
#pragma unroll
        for (int i = 0; i < NS; ++i) acc1[i] += __shfl_xor(acc1[i], 32, 64);
        acc_b1 += accb;
#pragma unroll
        for (int i = 0; i < NS; ++i) acc_w1[i] += acc1[i];
    }
    G32_STAMP(18);
    // ---- dW2^T[k][j] = sum_s H1[s][k] dZ2[s][j] (MFMA, K = the 32 samples); stored as Flux W2[j + H3 k] ----
    {
        const uint16_t* ap = l_HT + (32 * w + r) * LDT + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < G32 / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(l_ZT + (32 * t + r) * LDT + 16 * ks + 8 * kb);
                dw[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, dw[t], 0, 0, 0);
            }
        }
    }
    G32_STAMP(19);
    __syncthreads();  // the next tile's gather rewrites l_x / l_a / ... that the phases above read
    }  // tiles
    // ---- this workgroup's partial row ----
    if (tid < G32) {  // lanes 0..31 of wave 0: the row slots' db3 / loss sums (fixed tree)
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
            for (int o = 0; o <= MAXO; ++o) acc_b3[o] += __shfl_down(acc_b3[o], off, 64);
    }
    const bool wt = g.wt_rows != 0;
    if (tid == 0) {
        for (int o = 0; o < na; ++o) store_row(&out[ob3 + o], acc_b3[o], wt);
        store_row(&g.loss_partials[blockIdx.x], acc_b3[MAXO], wt);
    }
    if (kb == 0) {
        store_row(&out[ob2 + col], acc_b2, wt);
        for (int o = 0; o < na; ++o) store_row(&out[oW3 + o + na * col], acc_w3[o], wt);
        store_row(&out[ob1 + col], acc_b1, wt);
#pragma unroll
        for (int i = 0; i < NS; ++i) store_row(&out[oW1 + col + H3 * i], acc_w1[i], wt);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) store_row(&out[oW2 + (r + 32 * t) + H3 * (32 * w + mfma_row(q, kb))], dw[t][q], wt);
}

constexpr size_t GRAD32_LDS = (4 * G32 * 2 + MAXO * G32 + G32 + 2 * 8 * 4 * G32 + 2 * G32 + G32 * LDH2 + 2 * SMALLW) *
                                  sizeof(float) +
                              (2 * G32 * LDH + 2 * H3 * LDT) * sizeof(uint16_t);

// bf16 copies of W2 in MFMA B-fragment order, one per operand orientation (16-byte units, see gemm_slab):
//   packed[0 .. H*H)      "W2jk": fragment (ks, t), lane l holds W2[j = 32 t + (l & 31)][k = 16 ks + 8 (l >> 5) + u]
//   packed[H*H .. 2 H*H)  "W2kj": fragment (ks, t), lane l holds W2[j = 16 ks + 8 (l >> 5) + u][k = 32 t + (l & 31)]
// with W2[j][k] = Flux W2[j + H k].
__global__ __launch_bounds__(256) void mlp3_pack_kernel(const float* __restrict__ params, int ns,
                                                        uint16_t* __restrict__ packed) {
    const float* W2 = params + H3 * ns + H3;
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= H3 * H3) return;
    int u = q & 7, l = (q >> 3) & 63, f = q >> 9;
    int t = f & 3, ks = f >> 2;
    int col = 32 * t + (l & 31), kk = 16 * ks + 8 * (l >> 5) + u;
    packed[q] = f32_to_bf16_rne(W2[col + H3 * kk]);            // B(col = j, kk = k) = W2[j][k]
    packed[H3 * H3 + q] = f32_to_bf16_rne(W2[kk + H3 * col]);  // B(col = k, kk = j) = W2[j][k]
}

// glorot_uniform stand-in (same convention as mlp2_init_kernel): tensor ids net_id * 4 + {0: W1, 1: W2, 2: W3}
__global__ __launch_bounds__(256) void mlp3_init_kernel(float* __restrict__ p, int ns, int h, int na, uint64_t seed,
                                                        uint32_t net_id) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t o1 = (int64_t)h * ns, o2 = o1 + h, o3 = o2 + (int64_t)h * h, o4 = o3 + h, o5 = o4 + (int64_t)na * h;
    if (q >= o5 + na) return;
    float val = 0.0f;  // Flux Dense bias default: zeros
    int64_t off = -1;
    uint32_t tid = 0;
    float fan = 1.0f;
    if (q < o1) { off = q; tid = 0; fan = (float)(ns + h); }
    else if (q >= o2 && q < o3) { off = q - o2; tid = 1; fan = (float)(h + h); }
    else if (q >= o4 && q < o5) { off = q - o4; tid = 2; fan = (float)(h + na); }
    if (off >= 0) {
        u32x4 w = philox4x32_10(seed, (uint32_t)off, 0, net_id * 4u + tid, TAG_INIT);
        val = (2.0f * u01_f32(w.x) - 1.0f) * sqrtf(6.0f / fan);
    }
    p[q] = val;
}

// partial gradients [nb][np] -> grad[np]: 64 parameters per workgroup, the block range split over the 4 waves,
// fixed summation order (ascending blocks within a wave, then waves 0..3)
__global__ __launch_bounds__(256) void d3_reduce_kernel(const float* __restrict__ partials,
                                                        const float* __restrict__ loss_partials, int nb, int np,
                                                        float* __restrict__ grad, float* __restrict__ loss,
                                                        float inv_b) {
    __shared__ float l_g[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const int per = (nb + 3) / 4;
    const int b0 = grp * per, b1 = min(nb, b0 + per);
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

// optimise! tail of the 3-layer learner in ONE launch: d3_reduce_kernel + sumsq_scaled_partial_kernel +
// clip_adam_grid_kernel (optim.hip, the > 12 k parameter regime of rlhip_clip_adam_f32) + mlp3_pack_kernel.
// One lane per parameter, ceil(np / 256) workgroups (69 for the 17 410-parameter CartPole net: all co-resident, so
// a spin barrier on an agent-scope counter is safe).  Every step reproduces the arithmetic of the kernel it
// replaces -- the block reduction order, the Float64 per-workgroup sums of squares and their fold, the Adam expression
// -- so the result is bit-identical to the four launches; the bf16 fragment copies of W2 are refreshed in place.
struct D3Apply {
    float* p;
    float* m;
    float* v;
    float* beta_pow;
    float* gn_out;
    uint16_t* packed;
    double* sumsq;           // [gridDim.x] in the workspace tail
    unsigned int* counter;   // [0] arrive, [1] depart: zero before the first launch, re-armed here
    float grad_scale, clip_norm, lr, b1, b2, eps;
    int ns;
    // PPO actor / critic (ppo3.hip): two nets in one flat vector, [actor (np_a) | critic]; np_a = 0: one net (DQN).
    // packed then holds [actor W2jk | W2kj | critic W2jk | W2kj]; the loss line is the four PPO numbers.
    int np_a;
    float wa, wc, we;
};

__global__ __launch_bounds__(256) void d3_apply_kernel(const float* __restrict__ partials,
                                                       const float* __restrict__ loss_partials, int nb, int np,
                                                       float* __restrict__ grad, float* __restrict__ loss, float inv_b,
                                                       D3Apply ap) {
    __shared__ double scratch[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x * 256 + tid;
    const bool own = i < np;
    // the norm exchange: up to 128 workgroups publish their sum of squares as two 8-byte {epoch, half of the double} granules
    // that the others poll directly (one hop; epoch = launches so far + 1, device-resident, advanced by the last workgroup
    // out) -- larger grids keep the arrival counter (the tail holds 256 doubles)
    const bool gran = 2 * gridDim.x <= 256;
    unsigned int epoch = 0;
    if (gran) epoch = __hip_atomic_load(ap.counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // operands that do not depend on the barrier
    const float p0 = own ? ap.p[i] : 0.f, m0 = own ? ap.m[i] : 0.f, v0 = own ? ap.v[i] : 0.f;
    const float c1 = 1.0f - ap.beta_pow[0], c2 = 1.0f - ap.beta_pow[1];
    // d3_reduce_kernel: block range split in four, ascending inside a part, then ((a0 + a1) + a2) + a3
    float g = 0.f;
    if (own) {
        const int per = (nb + 3) / 4;
        float a[4];
        if (nb <= 16) {
            // all loads issued before the first add: one round trip (x + 0.0f is exact, so the padded slots change no bit)
            float t[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b = q * per + u;
                    t[q][u] = (u < per && b < nb) ? partials[(int64_t)b * np + i] : 0.0f;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += t[q][u];
                a[q] = acc;
            }
        } else {
            // 64 loads in flight per trip (32 rows of two of the four groups): with up to 512 partial rows coming from
            // other XCDs' L2 slices every dependent round trip costs ~2 us, and the adds must stay sequential per group --
            // the same ascending order as d3_reduce_kernel / ppo3_reduce_kernel (x + 0.0f is exact: padded slots change no bit)
#pragma unroll
            for (int qp = 0; qp < 4; qp += 2) {
                const int b0a = qp * per, b1a = min(nb, b0a + per);
                const int b0b = (qp + 1) * per, b1b = min(nb, b0b + per);
                float acca = 0.f, accb = 0.f;
                for (int off = 0; off < per; off += 32) {
                    float ta[32], tb[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        ta[u] = (b0a + off + u < b1a) ? partials[(int64_t)(b0a + off + u) * np + i] : 0.0f;
                        tb[u] = (b0b + off + u < b1b) ? partials[(int64_t)(b0b + off + u) * np + i] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        acca += ta[u];
                        accb += tb[u];
                    }
                }
                a[qp] = acca;
                a[qp + 1] = accb;
            }
        }
        g = ((a[0] + a[1]) + a[2]) + a[3];
    }
    if (blockIdx.x == 0 && loss != nullptr && ap.np_a == 0 && wv == 1) {
        float a = 0.f;
        for (int b = lane; b < nb; b += 64) a += loss_partials[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) loss[0] = a * inv_b;
    }
    if (blockIdx.x == 0 && loss != nullptr && ap.np_a > 0 && wv == 1) {
        // ppo3_reduce_kernel's loss line: wave 1 alone, no workgroup barrier (workgroup 0 must not arrive late at the norm
        // exchange every other workgroup waits on); the same sums in the same order as with one wave per column
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
            loss[0] = ap.wa * actor_loss + ap.wc * critic_loss - ap.we * ent_loss;
            loss[1] = actor_loss;
            loss[2] = critic_loss;
            loss[3] = ent_loss;
        }
    }
    // sumsq_scaled_partial_kernel (one element per lane at this size)
    const float x = own ? g * ap.grad_scale : 0.0f;
    double acc = own ? (double)x * (double)x : 0.0;
    acc = block_sum_f64_dpp(acc, scratch);
    typedef unsigned long long u64;
    double tot = 0.0;
    if (gran) {
        if (tid == 0) {
            const u64 bits = (u64)__double_as_longlong(acc), ep = (u64)epoch << 32;
            u64* gr = reinterpret_cast<u64*>(ap.sumsq) + 2 * blockIdx.x;
            __hip_atomic_store(gr, ep | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gr + 1, ep | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // clip_adam_grid_kernel's sum over the partials, thread q takes workgroup q's
        for (int q = tid; q < (int)gridDim.x; q += 256) {
            const u64* gr = reinterpret_cast<const u64*>(ap.sumsq) + 2 * q;
            u64 hi, lo;
            for (;;) {
                hi = __hip_atomic_load(gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = __hip_atomic_load(gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(hi >> 32) == epoch && (unsigned int)(lo >> 32) == epoch) break;
                __builtin_amdgcn_s_sleep(1);
            }
            tot += __longlong_as_double((long long)(((hi & 0xFFFFFFFFull) << 32) | (lo & 0xFFFFFFFFull)));
        }
    } else {
        if (tid == 0) {
            // the only data other workgroups read is this partial: a device-scope (write-through) store, drained before the
            // arrival count goes up, read back by device-scope loads -- no release / acquire fences
            __hip_atomic_store(ap.sumsq + blockIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(ap.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ap.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
                __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        // clip_adam_grid_kernel
        for (int q = tid; q < (int)gridDim.x; q += 256)
            tot += __hip_atomic_load(ap.sumsq + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tot = block_sum_f64_dpp(tot, scratch);
    const float gn = (float)sqrt(tot);
    const float scale = (ap.clip_norm > 0.0f && ap.clip_norm <= gn) ? ap.clip_norm / fmaxf(ap.clip_norm, gn) : 1.0f;
    if (own) {
        float gi = g * ap.grad_scale;
        if (scale != 1.0f) gi *= scale;
        float pi = p0, mi = m0, vi = v0;
        adam1(pi, gi, mi, vi, ap.lr, ap.b1, ap.b2, ap.eps, c1, c2);
        ap.p[i] = pi;
        ap.m[i] = mi;
        ap.v[i] = vi;
        grad[i] = gi;
        // mlp3_pack_kernel, parameter-centric: W2[j + H3 k] goes to one slot of each fragment orientation
        const bool second = ap.np_a > 0 && i >= ap.np_a;  // the critic of a PPO pair
        const int e = (second ? i - ap.np_a : i) - (H3 * ap.ns + H3);
        if (e >= 0 && e < H3 * H3) {
            const int j = e & (H3 - 1), k = e >> 7;
            const uint16_t hb = f32_to_bf16_rne(pi);
            const int q1 = (((k >> 4) * 4 + (j >> 5)) << 9) | (((j & 31) + 32 * ((k >> 3) & 1)) << 3) | (k & 7);
            const int q2 = (((j >> 4) * 4 + (k >> 5)) << 9) | (((k & 31) + 32 * ((j >> 3) & 1)) << 3) | (j & 7);
            uint16_t* pk = ap.packed + (second ? 2 * H3 * H3 : 0);
            pk[q1] = hb;
            pk[H3 * H3 + q2] = hb;
        }
    }
    __syncthreads();  // every lane of this workgroup has read beta_pow
    if (tid == 0) {
        if (blockIdx.x == 0 && ap.gn_out) ap.gn_out[0] = gn;
        unsigned int prev = __hip_atomic_fetch_add(ap.counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {  // last one out: nobody reads beta_pow or the arrival counter any more
            ap.beta_pow[0] *= ap.b1;
            ap.beta_pow[1] *= ap.b2;
            __hip_atomic_store(ap.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ap.counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gran) __hip_atomic_store(ap.counter + 3, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// optimiser tail of the 3-layer PPO learner (ppo3.hip): partial rows -> gradient, PPO loss line, Float64 norm, clip,
// Adam, bf16 re-pack of both nets' W2 in ONE launch behind a grid barrier -- the arithmetic of ppo3_reduce_kernel,
// sumsq_scaled_partial_kernel, clip_adam_grid_kernel and ppo3_pack_kernel, which it replaces (their sequence stays the
// fallback when the grid does not fit the device: grid_barrier_capacity).  tail: 256 doubles + 16 counters, zero before
// the first call.  Returns 1 when the caller has to run the unfused sequence.
int32_t ppo3_apply_fused(const float* partials, const float* loss_partials, int nb, int np, int np_a, int ns, float* grad,
                         float* losses, float inv_b, float wa, float wc, float we, float* params, float* m, float* v,
                         float* beta_pow, uint16_t* packed, void* tail, float clip_norm, float lr, float b1, float b2,
                         float eps, hipStream_t s) {
    static PerDeviceInt cap_cache;
    const int cap = grid_barrier_capacity_cached(cap_cache, d3_apply_kernel, 256);
    const int grid = (np + 255) / 256;
    if (grid > cap || grid > 256) return 1;
    D3Apply ap{params, m, v, beta_pow, nullptr, packed, (double*)tail, (unsigned int*)((double*)tail + 256), 1.0f,
               clip_norm, lr, b1, b2, eps, ns, np_a, wa, wc, we};
    hipLaunchKernelGGL(d3_apply_kernel, dim3(grid), dim3(256), 0, s, partials, loss_partials, nb, np, grad, losses, inv_b,
                       ap);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// hidden = 256: the streaming kernels of ppo3w.hip (same loss, same precision contract, one operand register-resident per GEMM)
constexpr int HWIDE = 256;
int64_t dqn3w_workspace_bytes(int64_t ns, int64_t na, int64_t batch);
int32_t dqn3w_pack(const float* params, int64_t ns, uint16_t* packed, rlhip_stream_t stream);
int32_t dqn3w_plan(const float* params, const uint16_t* packed, int64_t ns, int64_t na, int32_t act, const float* obs, int64_t n,
                   double eps, uint64_t seed, uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                   rlhip_stream_t stream);
int32_t dqn3w_grad_entry(const rlhip_ring* rb, int64_t na, int32_t act, const float* params, const uint16_t* packed,
                         const float* target_params, const uint16_t* target_packed, int64_t batch, const int64_t* idx,
                         float gamma, float huber_delta, uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out,
                         float* loss_out, float* td_out, rlhip_stream_t stream, float* apply_p, uint16_t* apply_packed,
                         float* m, float* v, float* beta_pow, float* gn_out, float grad_scale, float clip_norm, float lr,
                         float b1, float b2, float eps, const float* isw);

constexpr size_t PLAN_LDS = (4 * TR + MAXO * TR + SMALLW) * sizeof(float) + TILE_ELEMS * sizeof(uint16_t);
constexpr size_t GRAD_LDS = (8 * TR + 3 * MAXO * TR + TR + 16 + 2 * TR + 4 * 5 * H3 + 2 * SMALLW) * sizeof(float) +
                            3 * TILE_ELEMS * sizeof(uint16_t);

template <class P, int NA>
static int32_t dqn3_act_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, const float* params,
                             const uint16_t* packed, int act, double eps, uint64_t explorer_seed, uint32_t step,
                             uint64_t env_seed, uint32_t env_id_base, ActRing rb, int32_t* actions, float* q_out, float* obs,
                             float* last_obs, hipStream_t s) {
    typename P::cfg_t c2 = *cfg;
    c2.continuous = 0;
    ActTail<P> tail{P::make(c2), EnvArrays<float>::from(*st), env_seed, rb, obs, last_obs};
    const dim3 grid32((unsigned)((n + P32 - 1) / P32));
    if (act == 0)
        hipLaunchKernelGGL((mlp3_plan32_kernel<P::ODIM, NA, 0, ActTail<P>>), grid32, dim3(256), PLAN32_LDS, s, params, packed, obs,
                           n, eps, explorer_seed, env_id_base, step, actions, q_out, tail);
    else
        hipLaunchKernelGGL((mlp3_plan32_kernel<P::ODIM, NA, 1, ActTail<P>>), grid32, dim3(256), PLAN32_LDS, s, params, packed, obs,
                           n, eps, explorer_seed, env_id_base, step, actions, q_out, tail);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

template <typename K>
static int32_t allow_lds(K kernel, size_t bytes, unsigned long long* done) { return allow_big_lds(kernel, bytes, done); }

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int64_t rlhip_mlp3_nparams(int64_t ns, int64_t h, int64_t na) { return mlp3_nparams(ns, h, na); }

#ifdef RLHIP_D3_TIMING
int32_t rlhip_debug_d3_stamps(long long* out32) {
    RLHIP_CHECK_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_d3_stamps), sizeof(long long) * 32));
    return RLHIP_OK;
}
#endif

int64_t rlhip_mlp3_packed_elems(int64_t h) { return 2 * h * h; }

int32_t rlhip_mlp3_init_f32(float* params, int64_t ns, int64_t h, int64_t na, uint64_t seed, uint32_t net_id,
                            rlhip_stream_t stream) {
    RLHIP_REQUIRE(params != nullptr && ns >= 1 && h >= 1 && na >= 1, "bad arguments");
    int64_t np = mlp3_nparams(ns, h, na);
    hipLaunchKernelGGL(mlp3_init_kernel, dim3((int)((np + 255) / 256)), dim3(256), 0, as_stream(stream), params, (int)ns,
                       (int)h, (int)na, seed, net_id);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_mlp3_pack_bf16(const float* params, int64_t ns, int64_t h, int64_t na, uint16_t* packed,
                             rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && packed, "NULL argument");
    RLHIP_REQUIRE(h == H3 || h == HWIDE, "the MFMA Q-network path is built for hidden = 128 or 256");
    RLHIP_REQUIRE(ns >= 2 && ns <= 4 && na >= 1 && na <= MAXO, "obs dim must be 2..4, na <= 4");
    if (h == HWIDE) return dqn3w_pack(params, ns, packed, stream);
    hipLaunchKernelGGL(mlp3_pack_kernel, dim3(H3 * H3 / 256), dim3(256), 0, as_stream(stream), params, (int)ns, packed);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dqn3_plan_f32(const float* params, const uint16_t* packed, int64_t ns, int64_t h, int64_t na,
                            int32_t act, const float* obs, int64_t n, double eps, uint64_t seed,
                            uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                            rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && packed && obs && (actions || q_out), "NULL argument");
    RLHIP_REQUIRE(h == H3 || h == HWIDE, "the MFMA Q-network path is built for hidden = 128 or 256");
    RLHIP_REQUIRE((ns == 4 && na == 2) || (ns == 2 && na == 3) || (ns == 3 && na == 3),
                  "(obs dim, actions) must be (4, 2) CartPole, (2, 3) MountainCar or (3, 3) Pendulum");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE((((uintptr_t)packed) & 15) == 0, "packed weights must be 16-byte aligned");
    if (n == 0) return RLHIP_OK;
    if (h == HWIDE) return dqn3w_plan(params, packed, ns, na, act, obs, n, eps, seed, env_id_base, step, actions, q_out, stream);
    hipStream_t s = as_stream(stream);
    dim3 grid((unsigned)((n + TR - 1) / TR));
    const bool small = n <= (1 << 15);
    const dim3 grid32((unsigned)((n + P32 - 1) / P32));
#define LAUNCH_P(NS_, NA_, ACT_)                                                                               \
    do {                                                                                                       \
        if (small) {                                                                                           \
            hipLaunchKernelGGL((mlp3_plan32_kernel<NS_, NA_, ACT_, NoActTail>), grid32, dim3(256), PLAN32_LDS, s,  \
                               params, packed, obs, n, eps, seed, env_id_base, step, actions, q_out, NoActTail{}); \
            break;                                                                                             \
        }                                                                                                      \
        static unsigned long long done_ = 0;                                                                             \
        int32_t rc_ = allow_lds(mlp3_plan_kernel<NS_, NA_, ACT_>, PLAN_LDS, &done_);                           \
        if (rc_) return rc_;                                                                                   \
        hipLaunchKernelGGL((mlp3_plan_kernel<NS_, NA_, ACT_>), grid, dim3(256), PLAN_LDS, s, params, packed,   \
                           obs, n, eps, seed, env_id_base, step, actions, q_out);                              \
    } while (0)
    // (obs dim, actions) of the three classic-control envs: CartPole (4, 2), MountainCar (2, 3), Pendulum (3, 3)
    if (ns == 4 && na == 2) { if (act == 0) LAUNCH_P(4, 2, 0); else LAUNCH_P(4, 2, 1); }
    else if (ns == 2 && na == 3) { if (act == 0) LAUNCH_P(2, 3, 0); else LAUNCH_P(2, 3, 1); }
    else { if (act == 0) LAUNCH_P(3, 3, 0); else LAUNCH_P(3, 3, 1); }
#undef LAUNCH_P
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dqn3_act_supported(int32_t kind, int64_t n, int64_t h, int64_t na) {
    const int64_t want = kind == 0 ? 2 : 3;  // CartPole (4, 2), Pendulum (3, 3), MountainCar (2, 3): the shapes the plan kernels are built for
    return (kind >= 0 && kind <= 2 && h == H3 && na == want && n >= 1 && n <= (1 << 15)) ? 1 : 0;
}

/* plan! + act! + push! of one vec-step of the 3-layer Q-network in ONE launch (mlp3_plan32_kernel<..., ActTail>): what
 * rlhip_dqn3_plan_f32 followed by rlhip_env_act_push_f32 computes, bit for bit */
int32_t rlhip_dqn3_act_f32(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, const float* params,
                           const uint16_t* packed, int64_t h, int64_t na, int32_t act, double eps, uint64_t explorer_seed,
                           uint32_t explorer_step, uint64_t env_seed, uint32_t env_id_base, rlhip_ring* rb, int32_t* actions,
                           float* q_out, float* obs, float* last_obs, rlhip_stream_t stream) {
    RLHIP_REQUIRE(env_cfg && st && params && packed && rb && actions && obs, "NULL argument");
    RLHIP_REQUIRE(rlhip_dqn3_act_supported(kind, n, h, na), "unsupported (kind, n, hidden, actions) for the fused 3-layer act kernel");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE((((uintptr_t)packed) & 15) == 0, "packed weights must be 16-byte aligned");
    RLHIP_REQUIRE(st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    RLHIP_REQUIRE(rb->elem_bytes == 4 && rb->n_env == n && rb->obs_dim == (kind == 0 ? 4 : (kind == 1 ? 3 : 2)),
                  "ring geometry does not match the env");
    RLHIP_REQUIRE(rb->len_sa >= 1, "push the first state before the first transition");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "the fused act + push kernels write a record ring (rlhip_ring_init, ABI 2)");
    const ActRing ar = claim_slots(rb);
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return dqn3_act_impl<CartPoleParams<float>, 2>((const rlhip_cartpole_cfg*)env_cfg, st, n, params, packed, act, eps,
                                                       explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out, obs,
                                                       last_obs, s);
    if (kind == 1)
        return dqn3_act_impl<PendulumParams<float>, 3>((const rlhip_pendulum_cfg*)env_cfg, st, n, params, packed, act, eps,
                                                       explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out, obs,
                                                       last_obs, s);
    return dqn3_act_impl<MountainCarParams<float>, 3>((const rlhip_mountaincar_cfg*)env_cfg, st, n, params, packed, act, eps,
                                                      explorer_seed, explorer_step, env_seed, env_id_base, ar, actions, q_out, obs,
                                                      last_obs, s);
}

int64_t rlhip_dqn3_workspace_bytes(int64_t ns, int64_t h, int64_t na, int64_t batch) {
    if (h == HWIDE) return dqn3w_workspace_bytes(ns, na, batch);
    // rows of partials: the larger of what the 32-sample kernel (persistent, <= D3_GRAD32_BLOCKS) and the 128-row kernel use
    int64_t nb = (batch + G32 - 1) / G32;
    if (nb > D3_GRAD32_BLOCKS) nb = D3_GRAD32_BLOCKS;
    if ((batch + TR - 1) / TR > nb) nb = (batch + TR - 1) / TR;
    if (nb < 1) nb = 1;
    // partials | loss partials | (8-byte aligned) 256 Float64 sums of squares + 64 B of counters for
    // rlhip_dqn3_update_f32 (the tail must be zero before the first use)
    int64_t base = nb * (mlp3_nparams(ns, h, na) + 1) * (int64_t)sizeof(float);
    base = (base + 7) & ~(int64_t)7;
    return base + 256 * (int64_t)sizeof(double) + 64;
}

static int32_t dqn3_grad_impl(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                              const uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                              int64_t batch, const int64_t* idx, float gamma, float huber_delta, uint64_t seed,
                              uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out, float* td_out,
                              rlhip_stream_t stream, D3Apply* apply, const float* isw = nullptr) {
    RLHIP_REQUIRE(rb && params && packed && target_params && target_packed && workspace && grad_out, "NULL argument");
    RLHIP_REQUIRE(rb->elem_bytes == 4, "the DQN learner expects Float32 observations");
    RLHIP_REQUIRE(rb->obs_dim >= 2 && rb->obs_dim <= 4, "fused DQN kernel supports obs_dim 2..4");
    RLHIP_REQUIRE(rb->layout == RLHIP_RING_RECORDS, "the DQN learner reads a record ring (rlhip_ring_init, ABI 2)");
    RLHIP_REQUIRE(h == H3 || h == HWIDE, "the MFMA Q-network path is built for hidden = 128 or 256");
    RLHIP_REQUIRE((rb->obs_dim == 4 && na == 2) || (rb->obs_dim == 2 && na == 3) || (rb->obs_dim == 3 && na == 3),
                  "(obs dim, actions) must be (4, 2) CartPole, (2, 3) MountainCar or (3, 3) Pendulum");
    RLHIP_REQUIRE(act == 0 || act == 1, "act must be 0 (relu) or 1 (tanh)");
    RLHIP_REQUIRE(batch >= 1, "empty batch");
    RLHIP_REQUIRE(rb->len_rt >= 1, "cannot sample from an empty trajectory");
    RLHIP_REQUIRE(((((uintptr_t)packed) | ((uintptr_t)target_packed)) & 15) == 0, "packed weights must be 16-byte aligned");
    if (h == HWIDE) {
        if (apply)
            return dqn3w_grad_entry(rb, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta,
                                    seed, draw_ctr, workspace, grad_out, loss_out, td_out, stream, apply->p, apply->packed,
                                    apply->m, apply->v, apply->beta_pow, apply->gn_out, apply->grad_scale, apply->clip_norm,
                                    apply->lr, apply->b1, apply->b2, apply->eps, isw);
        return dqn3w_grad_entry(rb, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta, seed,
                                draw_ctr, workspace, grad_out, loss_out, td_out, stream, nullptr, nullptr, nullptr, nullptr,
                                nullptr, nullptr, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, isw);
    }
    RLHIP_REQUIRE(batch <= (int64_t)D3_MAX_BLOCKS * TR, "batch too large for one launch");
    const int ns = (int)rb->obs_dim;
    const int64_t np = mlp3_nparams(ns, h, na);
    // measured (gradient + reduce, us; 32-sample persistent / 128-row): batch 512: 14.2 / 24.0, 4096: 19.0 / 25.3,
    // 8192: 27.0 / 27.2, 16384: 39.7 / 32.1, 32768: 52.3 / 37.0, 65536: 76.2 / 69.3, 131072: 121.7 / 131.9 -- the
    // 32-sample kernel (280 VGPRs persistent: one workgroup per CU) wins where latency or the partial-row volume decide;
    // with two workgroups per CU (256 VGPRs, some spills) it is 66.2 us at 65536 and 106.4 us at 131072 (161 TFLOP/s)
    const bool small = batch <= 8192 || batch >= 65536;
    const int64_t tiles32 = (batch + G32 - 1) / G32;
    const int nb = small ? (int)(tiles32 < D3_GRAD32_BLOCKS ? tiles32 : D3_GRAD32_BLOCKS) : (int)((batch + TR - 1) / TR);
    Dqn3Args g;
    g.ring = {(const uint8_t*)rb->state, rb->capacity, rb->n_env, rb->head_sa};
    g.total = (uint64_t)rb->len_rt * (uint64_t)rb->n_env;
    g.idx = idx;
    g.params = params;
    g.tparams = target_params;
    g.packed = packed;
    g.tpacked = target_packed;
    g.partials = (float*)workspace;
    g.loss_partials = g.partials + (int64_t)nb * np;
    g.td_out = td_out;
    g.isw = isw;
    g.na = (int)na;
    g.np = (int)np;
    g.num_tiles = (int)tiles32;
    g.batch = batch;
    g.gamma = gamma;
    g.delta = huber_delta;
    g.inv_b = 1.0f / (float)batch;
    g.seed = seed;
    g.draw_ctr = draw_ctr;
    g.wt_rows = (int64_t)nb * np * (int64_t)sizeof(float) >= ((int64_t)4 << 20) ? 1 : 0;  // (measured: 1 MB plain, 9 MB write-through)
    hipStream_t s = as_stream(stream);
#define LAUNCH_G(NS_, NA_, ACT_)                                                                        \
    do {                                                                                                \
        if (small) {                                                                                    \
            if (batch >= 65536) {                                                                       \
                static unsigned long long done32b_ = 0;                                                           \
                int32_t rc_ = allow_lds(dqn3_grad32_kernel<NS_, NA_, ACT_, 2>, GRAD32_LDS, &done32b_);   \
                if (rc_) return rc_;                                                                    \
                hipLaunchKernelGGL((dqn3_grad32_kernel<NS_, NA_, ACT_, 2>), dim3(nb), dim3(256), GRAD32_LDS, s, g); \
                break;                                                                                  \
            }                                                                                           \
            static unsigned long long done32_ = 0;                                                                \
            int32_t rc_ = allow_lds(dqn3_grad32_kernel<NS_, NA_, ACT_, 1>, GRAD32_LDS, &done32_);       \
            if (rc_) return rc_;                                                                        \
            hipLaunchKernelGGL((dqn3_grad32_kernel<NS_, NA_, ACT_, 1>), dim3(nb), dim3(256), GRAD32_LDS, s, g); \
            break;                                                                                      \
        }                                                                                               \
        static unsigned long long done_ = 0;                                                                      \
        int32_t rc_ = allow_lds(dqn3_grad_kernel<NS_, NA_, ACT_>, GRAD_LDS, &done_);                    \
        if (rc_) return rc_;                                                                            \
        hipLaunchKernelGGL((dqn3_grad_kernel<NS_, NA_, ACT_>), dim3(nb), dim3(256), GRAD_LDS, s, g);    \
    } while (0)
    if (ns == 4 && na == 2) { if (act == 0) LAUNCH_G(4, 2, 0); else LAUNCH_G(4, 2, 1); }
    else if (ns == 2 && na == 3) { if (act == 0) LAUNCH_G(2, 3, 0); else LAUNCH_G(2, 3, 1); }
    else { if (act == 0) LAUNCH_G(3, 3, 0); else LAUNCH_G(3, 3, 1); }
#undef LAUNCH_G
    if (apply) {
        int64_t base = ((int64_t)nb * (np + 1) * (int64_t)sizeof(float) + 7) & ~(int64_t)7;
        apply->sumsq = (double*)((char*)workspace + base);
        apply->counter = (unsigned int*)((char*)workspace + base + 256 * sizeof(double));
        apply->ns = ns;
        // the kernel spins on a grid-wide counter: every workgroup must be resident at once (69 for np = 17 410) --
        // checked against the occupancy of this kernel on this device (grid_barrier_capacity, common.h); beyond it the
        // tail runs as the launches it fuses (bit-identical): reduce, clip + Adam, bf16 re-pack
        static PerDeviceInt cap_cache;
        const int cap = grid_barrier_capacity_cached(cap_cache, d3_apply_kernel, 256);
        if ((np + 255) / 256 <= cap) {
            hipLaunchKernelGGL(d3_apply_kernel, dim3((int)((np + 255) / 256)), dim3(256), 0, s, g.partials, g.loss_partials,
                               nb, (int)np, grad_out, loss_out, g.inv_b, *apply);
        } else {
            hipLaunchKernelGGL(d3_reduce_kernel, dim3((int)((np + 63) / 64)), dim3(256), 0, s, g.partials, g.loss_partials,
                               nb, (int)np, grad_out, loss_out, g.inv_b);
            RLHIP_LAUNCH_CHECK();
            int32_t rc2 = rlhip_clip_adam_f32(apply->p, grad_out, apply->m, apply->v, apply->beta_pow, np, apply->grad_scale,
                                              apply->clip_norm, apply->lr, apply->b1, apply->b2, apply->eps, apply->gn_out,
                                              stream);
            if (rc2) return rc2;
            rc2 = rlhip_mlp3_pack_bf16(apply->p, ns, h, na, apply->packed, stream);
            if (rc2) return rc2;
        }
    } else {
        hipLaunchKernelGGL(d3_reduce_kernel, dim3((int)((np + 63) / 64)), dim3(256), 0, s, g.partials, g.loss_partials,
                           nb, (int)np, grad_out, loss_out, g.inv_b);
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dqn3_grad_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                            const uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                            int64_t batch, const int64_t* idx, float gamma, float huber_delta, uint64_t seed,
                            uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out, float* td_out,
                            rlhip_stream_t stream) {
    if (idx != nullptr && rb != nullptr) RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    return dqn3_grad_impl(rb, h, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta,
                          seed, draw_ctr, workspace, grad_out, loss_out, td_out, stream, nullptr);
}

/* rlhip_dqn3_grad_f32 with importance-sampling weights: loss = mean(weights .* huber(td)) (prioritized replay, beta > 0) */
int32_t rlhip_dqn3_grad_w_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, const float* params,
                              const uint16_t* packed, const float* target_params, const uint16_t* target_packed,
                              int64_t batch, const int64_t* idx, const float* weights, float gamma, float huber_delta,
                              void* workspace, float* grad_out, float* loss_out, float* td_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(idx != nullptr && weights != nullptr && rb != nullptr, "idx / weights / ring is NULL");
    RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream);
    return dqn3_grad_impl(rb, h, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta, 0, 0,
                          workspace, grad_out, loss_out, td_out, stream, nullptr, weights);
}

/* optimise!(learner, batch) of the 3-layer learner in two launches: gradient, then reduce + clip + Adam + bf16
 * re-pack (bit-identical to rlhip_dqn3_grad_f32, rlhip_clip_adam_f32, rlhip_mlp3_pack_bf16 in sequence) */
int32_t rlhip_dqn3_update_f32(const rlhip_ring* rb, int64_t h, int64_t na, int32_t act, float* params, uint16_t* packed,
                              const float* target_params, const uint16_t* target_packed, int64_t batch, float gamma,
                              float huber_delta, uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out,
                              float* loss_out, float* m, float* v, float* beta_pow, float grad_scale,
                              float max_grad_norm, float lr, float beta1, float beta2, float adam_eps, float* gn_out,
                              rlhip_stream_t stream) {
    RLHIP_REQUIRE(m && v && beta_pow, "NULL argument");
    D3Apply ap{params, m, v, beta_pow, gn_out, packed, nullptr, nullptr, grad_scale, max_grad_norm, lr, beta1, beta2,
               adam_eps, 0, 0, 0.f, 0.f, 0.f};
    return dqn3_grad_impl(rb, h, na, act, params, packed, target_params, target_packed, batch, nullptr, gamma,
                          huber_delta, seed, draw_ctr, workspace, grad_out, loss_out, nullptr, stream, &ap);
}

}  // extern "C"
