// ppo3.hip -- PPO rollout + clipped-surrogate gradient for THREE-layer actor / critic networks
//     actor  ns -> 128 -> 128 -> nout_a      critic  ns -> 128 -> 128 -> 1
// with the hidden x hidden layers on v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate, f32 master
// weights): BASELINE.json configs[2] "PendulumEnv + PPOPolicy ..., actor/critic MLP in bf16 MFMA".  Selected by
// rlhip_ppo_cfg.layers = 3 behind the unchanged rlhip_ppo_* entry points; the two-layer path (ppo.hip /
// ppo_grad.hip) has no GEMM-shaped layer and stays on the VALU.
//
// Replaces the same reference code as ppo.hip / ppo_grad.hip (the removed Zoo PPOPolicy on the blog's
// MultiThreadEnv loop; hyper-parameters index.html:15257-15278) with the blog-style deeper
// networks (index.html:15126-15128 uses the same Chain(Dense, Dense, Dense) shape for DQN).
//
// ppo3_rollout_kernel  one workgroup = 128 env instances for ALL T vec-steps of the update period (instances are
//     independent while the weights are frozen, so there is no inter-workgroup communication and the whole
//     rollout is one launch).  Both W2 matrices are converted to bf16 MFMA fragments in LDS once per launch;
//     per step: layer 1 (VALU) -> LDS tile -> layer 2 (MFMA) -> head (VALU + DPP) for the actor, then the critic,
//     then lane-per-env sampling (select_device.h / ppo_sample_device.h), env step with auto-reset
//     (env_device.h) and the trajectory writes.
// ppo3_grad_kernel     one workgroup = one 128-sample tile of the shuffled micro-batch (keyed bijection, common.h):
//     actor forward -> PPO loss terms and dL/d(head outputs) per sample -> mlp3_backward_tile; critic likewise.
//     Per-workgroup partial gradients, summed in a fixed order by ppo3_reduce_kernel (deterministic).
// Precision contract and tolerances: as dqn3.hip (the oracle applies the same bf16 roundings: oracle/rlo_learn.c
// with cfg.layers = 3).
#include "env_device.h"
#include "ppo_common.h"
#include "ppo_sample_device.h"
#include "mlp3_device.h"

extern "C" int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow, int64_t n,
                                       float grad_scale, float clip_norm, float lr, float beta1, float beta2,
                                       float eps, float* gn_out, rlhip_stream_t stream);

namespace rlhip {

int32_t ppo3_apply_fused(const float* partials, const float* loss_partials, int nb, int np, int np_a, int ns, float* grad,
                         float* losses, float inv_b, float wa, float wc, float we, float* params, float* m, float* v,
                         float* beta_pow, uint16_t* packed, void* tail, float clip_norm, float lr, float b1, float b2,
                         float eps, hipStream_t s);  // dqn3.hip

// hidden = 256: ppo3w.hip (three streaming kernels per net instead of one tile kernel; same contract)
int64_t ppo3w_nparams(int ns, int nout_a);
int64_t ppo3w_workspace_bytes(int ns, int nout_a, const rlhip_ppo_cfg* c, int64_t n, int64_t T);
int32_t ppo3w_rollout(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                      const PolicyDesc& pd, const float* params, uint64_t seed, uint32_t env_id_base, uint32_t vec_step0,
                      const rlhip_ppo_traj* traj, rlhip_stream_t stream);
int32_t ppo3w_grad(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                   const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb,
                   void* workspace, float* grad_out, float* losses_out, rlhip_stream_t stream);
int32_t ppo3w_update(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                     const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow, uint64_t seed,
                     uint32_t update_ctr, void* workspace, float* grad_scratch, float* losses_out, rlhip_stream_t stream);
constexpr int HWIDE = 256;

constexpr int P3_MAX_BLOCKS = 2048;
constexpr int64_t P3_TAIL_BYTES = 256 * 8 + 64 + 64;  // fused optimiser tail: Float64 partial norms + counters (zero-initialised)

__host__ __device__ __forceinline__ int64_t mlp3_np(int64_t ns, int64_t nout) {
    return H3 * ns + H3 + (int64_t)H3 * H3 + H3 + nout * H3 + nout;
}

// f32 W2 (Flux order W2[j + H k]) -> bf16 MFMA B fragments "W2jk" (see mlp3_pack_kernel in dqn3.hip) in LDS
__device__ __forceinline__ void stage_w2_fragments(const float* __restrict__ W2, uint16_t* l_frag, int tid) {
    for (int q8 = tid; q8 < H3 * H3 / 8; q8 += 256) {  // one 16-byte fragment slot per iteration
        const int l = q8 & 63, f = q8 >> 6;
        const int t = f & 3, ks = f >> 2;
        const int j = 32 * t + (l & 31), k0 = 16 * ks + 8 * (l >> 5);
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = W2[j + H3 * (k0 + u)];
        *reinterpret_cast<uint4*>(l_frag + 8 * q8) = pack8_bf16(v);
    }
}

// ------------------------------------------------------------------------------------ rollout
// The random part of the action sampling (policy_noise: Philox + Float64 log / sqrt / sin / cos) depends on (env, step) only:
// the rollout kernels evaluate it NCH steps ahead with ALL 256 threads (the env step itself occupies 128 / 32 of them)
// into a double-buffered LDS table [chunk parity][step in chunk][env][MAXO]; policy_select reads it back.  Bit-identical
// to policy_sample (same operations on the same operands).
constexpr int NCH3 = 2;   // 256 threads / 128 envs
constexpr size_t ROLL3_NOISE_OFF = (((4 * TR + 2 * MAXO * TR + 2 * SMALLW) * sizeof(float) +
                                     (2 * H3 * H3 + TILE_ELEMS) * sizeof(uint16_t)) + 15) & ~(size_t)15;

template <class P, int NOUT_A, int ACT>
__global__ __launch_bounds__(256) void ppo3_rollout_kernel(P p, EnvArrays<float> st, int64_t n, int T, int cont, int na,
                                                           const float* __restrict__ params, int64_t np_a,
                                                           uint64_t seed, uint32_t env_id_base, uint32_t vec_step0,
                                                           TrajPtrs tr, float gamma, float lambda) {
    constexpr int NS = P::ODIM;
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    float* l_x = reinterpret_cast<float*>(smem3);                   // [4][TR]
    float* l_q = l_x + 4 * TR;                                      // [MAXO][TR] actor head outputs
    float* l_v = l_q + MAXO * TR;                                   // [MAXO][TR] critic head output (row 0)
    float* l_w = l_v + MAXO * TR;                                   // [2][SMALLW]
    uint16_t* l_fa = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLW);  // actor W2 fragments  [H3 * H3]
    uint16_t* l_fc = l_fa + H3 * H3;                                // critic W2 fragments
    uint16_t* l_A = l_fc + H3 * H3;                                 // H1 tile [TR][LDH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Mlp3 ma = stage_small_weights(params, NS, NOUT_A, l_w, tid);
    const Mlp3 mc = stage_small_weights(params + np_a, NS, 1, l_w + SMALLW, tid);
    stage_w2_fragments(params + H3 * NS + H3, l_fa, tid);
    stage_w2_fragments(params + np_a + H3 * NS + H3, l_fc, tid);

    const int64_t env = (int64_t)blockIdx.x * TR + tid;
    const bool active = tid < TR && env < n;
    const int64_t envc = env < n ? env : n - 1;
    const uint32_t id = env_id_base + (uint32_t)envc;
    LaneState<float> e;
    float last_r = 0.0f;
    bool last_d = false;
    if (tid < TR) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][envc];
        e.t = st.t[envc];
        e.episode = st.episode[envc];
    }
    double* l_noise = reinterpret_cast<double*>(smem3 + ROLL3_NOISE_OFF);  // [2][NCH3][TR][MAXO]
    for (int t = 0; t <= T; ++t) {
        if ((t & (NCH3 - 1)) == 0) {
            const int i = tid >> 7, er = tid & (TR - 1);
            if (t + i < T) {
                const int64_t en = (int64_t)blockIdx.x * TR + er;
                double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
                policy_noise(cont, na, seed, env_id_base + (uint32_t)(en < n ? en : n - 1), vec_step0 + (uint32_t)(t + i), nz);
                double* dst = l_noise + ((size_t)((((t / NCH3) & 1) * NCH3 + i) * TR + er)) * MAXO;
#pragma unroll
                for (int k = 0; k < MAXO; ++k) dst[k] = nz[k];
            }
        }
        if (tid < TR) {
            float x[4];
            env_obs1(p, e, x);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                l_x[k * TR + tid] = x[k];
                if (active) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
            }
        }
        __syncthreads();
        f32x16 h2[4];
        if (t < T) {  // actor
            layer1_to_lds<NS, ACT>(ma, l_x, l_A, nullptr, tid);
            __syncthreads();
            layer2<ACT>(l_A, l_fa, ma.b2, w, lane, h2);
            head_to_lds<NOUT_A>(ma, h2, w, lane, l_q);
            __syncthreads();  // every wave is done with the actor's H1 tile
        }
        layer1_to_lds<NS, ACT>(mc, l_x, l_A, nullptr, tid);
        __syncthreads();
        layer2<ACT>(l_A, l_fc, mc.b2, w, lane, h2);
        head_to_lds<1>(mc, h2, w, lane, l_v);
        __syncthreads();
        if (tid < TR) {
            const float v = l_v[tid];
            if (active) tr.value[(int64_t)t * n + env] = v;
            if (t < T) {
                float oa[MAXO];
#pragma unroll
                for (int o = 0; o < MAXO; ++o) oa[o] = (o < NOUT_A) ? l_q[o * TR + tid] : 0.0f;
                int32_t ai;
                float af, lp;
                policy_select(cont, na, oa, l_noise + ((size_t)((((t / NCH3) & 1) * NCH3 + (t & (NCH3 - 1))) * TR + tid)) * MAXO,
                              ai, af, lp);
                env_step1(p, e, ai, af, last_r, last_d);
                if (last_d) env_reset1(p, e, seed, id);
                if (active) {
                    tr.logp[(int64_t)t * n + env] = lp;
                    if (cont) tr.action_f[(int64_t)t * n + env] = af;
                    else tr.action_i[(int64_t)t * n + env] = ai;
                    tr.reward[(int64_t)t * n + env] = last_r;
                    tr.terminal[(int64_t)t * n + env] = (uint8_t)last_d;
                }
            }
        }
        __syncthreads();  // l_x / l_q / l_v are rewritten by the next step
    }
    if (active && T > 0 && tr.adv && tr.ret)  // GAE + returns fused into the rollout launch (gae_device.h)
        gae_scan_lane(tr.adv, tr.ret, tr.reward, tr.value, tr.terminal, n, T, env, gamma, lambda);
    if (active) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
        st.t[env] = e.t;
        st.episode[env] = e.episode;
        if (T > 0) {
            st.reward[env] = last_r;
            st.done[env] = (uint8_t)last_d;
        }
    }
}

// ---- the same rollout for SMALL env counts: 32 env instances per workgroup ------------------------------------------------
// With 128-env tiles 4096 envs are 32 workgroups -- an eighth of the CUs, each walking a long per-step chain (measured
// 8.5 us per step).  Here the four waves of a workgroup SHARE one 32-row tile and split the 128 hidden units: a wave
// multiplies the 32 x 128 H1 tile by its own 32-column block of W2 (8 MFMAs per net and step instead of 32), layer 1 is
// 16 units per thread, and the heads are finished from an f32 H2 tile in LDS by all 256 threads (32 rows x 8 column
// parts).  H1 / W2 are rounded to bf16 and accumulated in f32 exactly as in the 128-row kernel (same k order), so H2 is
// bit-identical; the head sums run in a different (fixed) order -- inside the tolerance the oracle comparison states.
constexpr int R32 = 32;
constexpr int LDH2 = H3 + 4;  // f32 pitch of the H2 tile
constexpr int NCH32 = 8;  // 256 threads / 32 envs: noise of 8 steps per evaluation
constexpr size_t ROLL32_NOISE_OFF = (((4 * R32 + 8 * 4 * R32 + 2 * R32 * LDH2 + 2 * SMALLW) * sizeof(float) +
                                      (2 * H3 * H3 + 2 * R32 * LDH) * sizeof(uint16_t)) + 15) & ~(size_t)15;
template <class P, int NOUT_A, int ACT>
__global__ __launch_bounds__(256) void ppo3_rollout32_kernel(P p, EnvArrays<float> st, int64_t n, int T, int cont, int na,
                                                             const float* __restrict__ params, int64_t np_a,
                                                             uint64_t seed, uint32_t env_id_base, uint32_t vec_step0,
                                                             TrajPtrs tr, float gamma, float lambda) {
    constexpr int NS = P::ODIM;
    constexpr int NO = NOUT_A + 1;  // head outputs per env: the actor's, then the value
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    float* l_x = reinterpret_cast<float*>(smem3);                    // [4][R32]
    float* l_part = l_x + 4 * R32;                                   // [8 parts][4][R32]
    float* l_h2a = l_part + 8 * 4 * R32;                             // [R32][LDH2] actor H2 (f32)
    float* l_h2c = l_h2a + R32 * LDH2;                               // critic H2
    float* l_w = l_h2c + R32 * LDH2;                                 // [2][SMALLW]
    uint16_t* l_fa = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLW);  // actor W2 fragments [H3 * H3]
    uint16_t* l_fc = l_fa + H3 * H3;                                 // critic W2 fragments
    uint16_t* l_Ha = l_fc + H3 * H3;                                 // actor H1 tile [R32][LDH] (bf16)
    uint16_t* l_Hc = l_Ha + R32 * LDH;                               // critic H1 tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const Mlp3 ma = stage_small_weights(params, NS, NOUT_A, l_w, tid);
    const Mlp3 mc = stage_small_weights(params + np_a, NS, 1, l_w + SMALLW, tid);
    stage_w2_fragments(params + H3 * NS + H3, l_fa, tid);
    stage_w2_fragments(params + np_a + H3 * NS + H3, l_fc, tid);

    const int64_t env = (int64_t)blockIdx.x * R32 + tid;
    const bool active = tid < R32 && env < n;
    const int64_t envc = env < n ? env : n - 1;
    const uint32_t id = env_id_base + (uint32_t)envc;
    LaneState<float> e;
    float last_r = 0.0f;
    bool last_d = false;
    if (tid < R32) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][envc];
        e.t = st.t[envc];
        e.episode = st.episode[envc];
    }
    const int row1 = tid & 31, u0 = 16 * (tid >> 5);  // layer 1: this thread's row and its 16 hidden units
    const int part = tid >> 5;                        // heads: this thread's 16-column part of the H2 row `row1`
    double* l_noise = reinterpret_cast<double*>(smem3 + ROLL32_NOISE_OFF);  // [2][NCH32][R32][MAXO]
    __syncthreads();
    for (int t = 0; t <= T; ++t) {
        if ((t & (NCH32 - 1)) == 0) {
            const int i = tid >> 5, er = tid & (R32 - 1);
            if (t + i < T) {
                const int64_t en = (int64_t)blockIdx.x * R32 + er;
                double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
                policy_noise(cont, na, seed, env_id_base + (uint32_t)(en < n ? en : n - 1), vec_step0 + (uint32_t)(t + i), nz);
                double* dst = l_noise + ((size_t)((((t / NCH32) & 1) * NCH32 + i) * R32 + er)) * MAXO;
#pragma unroll
                for (int k = 0; k < MAXO; ++k) dst[k] = nz[k];
            }
        }
        if (tid < R32) {
            float x[4];
            env_obs1(p, e, x);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                l_x[k * R32 + tid] = x[k];
                if (active) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
            }
        }
        __syncthreads();
        // ---- layer 1 of both nets: h1 = act(b1 + W1 x), the fmaf chain of layer1_to_lds, 16 units per thread ----
        {
            float x[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = l_x[i * R32 + row1];
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                if (net == 0 && t == T) continue;  // the last pass only needs V(s_T)
                const Mlp3& m = net ? mc : ma;
                uint16_t* dst = (net ? l_Hc : l_Ha) + row1 * LDH + u0;
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
                    *reinterpret_cast<uint4*>(dst + 8 * h8) = pack8_bf16(hv);
                }
            }
        }
        __syncthreads();
        // ---- layer 2: this wave's 32 output columns of both nets (MFMA), bias + activation, f32 tile to LDS ----
        {
            f32x16 aa, ac;
#pragma unroll
            for (int q = 0; q < 16; ++q) aa[q] = 0.0f, ac[q] = 0.0f;
            const uint16_t* apa = l_Ha + r * LDH + 8 * kb;
            const uint16_t* apc = l_Hc + r * LDH + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < H3 / 16; ++ks) {
                const int fo = ((ks * 4 + w) * 64 + lane) * 8;
                if (t < T) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(apa + 16 * ks);
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(l_fa + fo);
                    aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, aa, 0, 0, 0);
                }
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(apc + 16 * ks);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(l_fc + fo);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, ac, 0, 0, 0);
            }
            const int col = 32 * w + r;
            const float ba = ma.b2[col], bc = mc.b2[col];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = mfma_row(q, kb);
                if (t < T) l_h2a[row * LDH2 + col] = act_fwd_t<ACT>(aa[q] + ba);
                l_h2c[row * LDH2 + col] = act_fwd_t<ACT>(ac[q] + bc);
            }
        }
        __syncthreads();
        // ---- heads: thread (row1, part) folds 16 columns of its H2 row for every output ----
        {
            float pa[NOUT_A], pc = 0.0f;
#pragma unroll
            for (int o = 0; o < NOUT_A; ++o) pa[o] = 0.0f;
            const float* ha = l_h2a + row1 * LDH2 + 16 * part;
            const float* hc = l_h2c + row1 * LDH2 + 16 * part;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 vc = *reinterpret_cast<const float4*>(hc + 4 * c4);
                const float hcv[4] = {vc.x, vc.y, vc.z, vc.w};
                float hav[4] = {0.f, 0.f, 0.f, 0.f};
                if (t < T) {
                    const float4 va = *reinterpret_cast<const float4*>(ha + 4 * c4);
                    hav[0] = va.x, hav[1] = va.y, hav[2] = va.z, hav[3] = va.w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 16 * part + 4 * c4 + c;
                    pc = fmaf(mc.W3[j], hcv[c], pc);
#pragma unroll
                    for (int o = 0; o < NOUT_A; ++o) pa[o] = fmaf(ma.W3[o + NOUT_A * j], hav[c], pa[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < NOUT_A; ++o) l_part[(part * 4 + o) * R32 + row1] = pa[o];
            l_part[(part * 4 + NOUT_A) * R32 + row1] = pc;
        }
        __syncthreads();
        if (tid < R32) {
            float out[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                float acc = l_part[o * R32 + tid];
#pragma unroll
                for (int pp = 1; pp < 8; ++pp) acc += l_part[(pp * 4 + o) * R32 + tid];
                out[o] = acc + (o < NOUT_A ? ma.b3[o] : mc.b3[0]);
            }
            const float v = out[NOUT_A];
            if (active) tr.value[(int64_t)t * n + env] = v;
            if (t < T) {
                float oa[MAXO];
#pragma unroll
                for (int o = 0; o < MAXO; ++o) oa[o] = (o < NOUT_A) ? out[o] : 0.0f;
                int32_t ai;
                float af, lp;
                policy_select(cont, na, oa, l_noise + ((size_t)((((t / NCH32) & 1) * NCH32 + (t & (NCH32 - 1))) * R32 + tid)) * MAXO,
                              ai, af, lp);
                env_step1(p, e, ai, af, last_r, last_d);
                if (last_d) env_reset1(p, e, seed, id);
                if (active) {
                    tr.logp[(int64_t)t * n + env] = lp;
                    if (cont) tr.action_f[(int64_t)t * n + env] = af;
                    else tr.action_i[(int64_t)t * n + env] = ai;
                    tr.reward[(int64_t)t * n + env] = last_r;
                    tr.terminal[(int64_t)t * n + env] = (uint8_t)last_d;
                }
            }
        }
        // no barrier here: the next pass rewrites l_x (last read before the second barrier of this pass) from the same
        // 32 lanes in program order; every other buffer is rewritten only after the next pass's barriers
    }
    if (active && T > 0 && tr.adv && tr.ret)  // GAE + returns fused into the rollout launch (gae_device.h)
        gae_scan_lane(tr.adv, tr.ret, tr.reward, tr.value, tr.terminal, n, T, env, gamma, lambda);
    if (active) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) st.s[k][env] = e.s[k];
        st.t[env] = e.t;
        st.episode[env] = e.episode;
        if (T > 0) {
            st.reward[env] = last_r;
            st.done[env] = (uint8_t)last_d;
        }
    }
}

constexpr size_t ROLL32_LDS = ROLL32_NOISE_OFF + 2 * NCH32 * R32 * MAXO * sizeof(double);
constexpr size_t ROLL3_LDS = ROLL3_NOISE_OFF + 2 * NCH3 * TR * MAXO * sizeof(double);

// ----------------------------------------------------------------------------------- gradient
struct P3Args {
    const float* obs;
    const float* logp;
    const float* adv;
    const float* ret;
    const float* action_f;
    const int32_t* action_i;
    const float* params;
    const uint16_t* packed;  // actor W2jk | W2kj | critic W2jk | W2kj   (fragment order)
    float* partials;         // [nb][np]
    float* loss_partials;    // [nb][4] {sum min(surr1, surr2), sum (ret - v)^2, sum entropy, -}
    int64_t n, np_a;
    uint32_t total, bm, pos0;
    int np, na;
    float lo, hi, wa, wc, we, inv_b, min_logp;
    PermKeys pk;
};

template <int NS, int NOUT_A, int ACT, int CONT>
__global__ __launch_bounds__(256) void ppo3_grad_kernel(P3Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    float* l_x = reinterpret_cast<float*>(smem3);              // [4][TR]
    float* l_q = l_x + 4 * TR;                                 // [MAXO][TR] head outputs of the current net
    float* l_dq = l_q + MAXO * TR;                             // [MAXO][TR] dL/d(head outputs)
    float* l_lp = l_dq + MAXO * TR;                            // [TR] old log-prob
    float* l_adv = l_lp + TR;                                  // [TR]
    float* l_ret = l_adv + TR;                                 // [TR]
    float* l_act = l_ret + TR;                                 // [TR] action (int bits or float)
    float* l_small = l_act + TR;                               // [2][8]
    float* l_red = l_small + 16;                               // [4][5][H3]
    float* l_w = l_red + 4 * 5 * H3;                           // [2][SMALLW]
    uint16_t* l_A = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLW);
    uint16_t* l_B = l_A + TILE_ELEMS;
    uint16_t* l_C = l_B + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Mlp3 ma = stage_small_weights(g.params, NS, NOUT_A, l_w, tid);
    const Mlp3 mc = stage_small_weights(g.params + g.np_a, NS, 1, l_w + SMALLW, tid);
    const int tile = blockIdx.x;
    float* out = g.partials + (int64_t)blockIdx.x * g.np;
    const int ob3a = H3 * NS + H3 + H3 * H3 + H3 + NOUT_A * H3;
    const int ob3c = H3 * NS + H3 + H3 * H3 + H3 + H3;

    // the actor's hidden-layer B fragments: requested now, consumed after the gather and layer 1
    bf16x8 bw[H3 / 16][4];
    load_w2_fragments(g.packed, lane, bw);
    // ---- gather the tile's samples f = perm(pos) from the trajectory ----
    if (tid < TR) {
        const uint32_t q = (uint32_t)tile * TR + (uint32_t)tid;
        const bool valid = q < g.bm;
        const uint32_t f = permute(g.pk, g.pos0 + (valid ? q : 0u));
        const uint32_t t = f / (uint32_t)g.n, i = f - t * (uint32_t)g.n;
#pragma unroll
        for (int k = 0; k < NS; ++k) l_x[k * TR + tid] = g.obs[((int64_t)t * NS + k) * g.n + i];
        l_lp[tid] = g.logp[f];
        l_adv[tid] = valid ? g.adv[f] : 0.0f;
        l_ret[tid] = g.ret[f];
        l_act[tid] = CONT ? g.action_f[f] : __int_as_float(g.action_i[f]);
    }
    __syncthreads();

    // ================================== actor ==================================
    f32x16 h2[4];
    layer1_to_lds<NS, ACT>(ma, l_x, l_A, l_B, tid);
    __syncthreads();
    layer2_regs<ACT>(l_A, bw, ma.b2, w, lane, h2);
    head_to_lds<NOUT_A>(ma, h2, w, lane, l_q);
    __syncthreads();
    if (tid < TR) {  // PPO clipped surrogate + entropy: loss terms and dL/d(actor outputs), as ppo_grad.hip phase 1b
        const int s = tid;
        const bool valid = ((uint32_t)tile * TR + (uint32_t)s) < g.bm;
        float oa[MAXO], dl[MAXO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < MAXO; ++o) oa[o] = (o < NOUT_A) ? l_q[o * TR + s] : 0.0f;
        const float lp_old = fmaxf(l_lp[s], g.min_logp);  // clamp!(log_p, log(1e-8), Inf)
        const float A = l_adv[s];
        float ent, surr_min;
        if (!CONT) {
            const int na = g.na;
            float mx = oa[0];
            for (int k = 1; k < na; ++k) mx = fmaxf(mx, oa[k]);
            float se = 0.f;
            for (int k = 0; k < na; ++k) se += expf(oa[k] - mx);
            const float lse = logf(se);
            float logp[MAXO], pr[MAXO];
            ent = 0.f;
            for (int k = 0; k < na; ++k) {
                logp[k] = (oa[k] - mx) - lse;
                pr[k] = expf(logp[k]);
                ent -= pr[k] * logp[k];
            }
            const int a = __float_as_int(l_act[s]);
            float lp_new = 0.f;
            for (int k = 0; k < na; ++k)
                if (k == a) lp_new = logp[k];
            const float ratio = expf(lp_new - lp_old);
            const float surr1 = ratio * A;
            const float surr2 = fminf(fmaxf(ratio, g.lo), g.hi) * A;
            const bool inside = ratio >= g.lo && ratio <= g.hi;
            const float dobj = (inside || surr1 < surr2) ? A : 0.f;
            const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
            surr_min = fminf(surr1, surr2);
            for (int k = 0; k < na; ++k) {
                const float dlp = ((k == a) ? 1.f : 0.f) - pr[k];
                const float dent = -pr[k] * (logp[k] + ent);
                dl[k] = dL_dlp * dlp - g.we * g.inv_b * dent;
            }
        } else {
            const float eps = 1.0e-8f;
            const float mu = oa[0], ls = oa[1];
            const float sg = expf(ls);
            const float z = l_act[s];
            const float se = sg + eps;
            const float zz = (z - mu) / se;
            const float lp_new = -(zz * zz + LOG2PI_F) / 2.0f - logf(se);
            ent = ((LOG2PI_F + 1.0f) + ls) / 2.0f;
            const float dmu = (z - mu) / (se * se);
            const float dls = ((z - mu) * (z - mu) / (se * se * se) - 1.0f / se) * sg;
            const float ratio = expf(lp_new - lp_old);
            const float surr1 = ratio * A;
            const float surr2 = fminf(fmaxf(ratio, g.lo), g.hi) * A;
            const bool inside = ratio >= g.lo && ratio <= g.hi;
            const float dobj = (inside || surr1 < surr2) ? A : 0.f;
            const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
            surr_min = fminf(surr1, surr2);
            dl[0] = dL_dlp * dmu;
            dl[1] = dL_dlp * dls - g.we * g.inv_b * 0.5f;
        }
        if (!valid) {
            dl[0] = dl[1] = dl[2] = dl[3] = 0.f;
            surr_min = 0.f;
            ent = 0.f;
        }
        float red[MAXO + 2];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            l_dq[o * TR + s] = dl[o];
            red[o] = dl[o];
        }
        red[MAXO] = surr_min;
        red[MAXO + 1] = ent;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int o = 0; o < MAXO + 2; ++o) red[o] += __shfl_down(red[o], off, 64);
        if (lane == 0)
#pragma unroll
            for (int o = 0; o < MAXO + 2; ++o) l_small[w * 8 + o] = red[o];
    }
    __syncthreads();
    if (tid == 0) {
        for (int o = 0; o < NOUT_A; ++o) store_wt(&out[ob3a + o], l_small[o] + l_small[8 + o]);
        store_wt(&g.loss_partials[(int64_t)blockIdx.x * 4 + 0], l_small[MAXO] + l_small[8 + MAXO]);
        store_wt(&g.loss_partials[(int64_t)blockIdx.x * 4 + 2], l_small[MAXO + 1] + l_small[8 + MAXO + 1]);
    }
    mlp3_backward_tile<NS, NOUT_A, ACT>(ma, g.packed + H3 * H3, h2, l_x, l_dq, l_red, l_A, l_B, l_C, out, tid);
    __syncthreads();  // the actor's tiles, l_red and l_small are free again

    // ================================== critic ==================================
    const uint16_t* pkc = g.packed + 2 * H3 * H3;
    layer1_to_lds<NS, ACT>(mc, l_x, l_A, l_B, tid);
    __syncthreads();
    layer2<ACT>(l_A, pkc, mc.b2, w, lane, h2);
    head_to_lds<1>(mc, h2, w, lane, l_q);
    __syncthreads();
    if (tid < TR) {
        const int s = tid;
        const bool valid = ((uint32_t)tile * TR + (uint32_t)s) < g.bm;
        const float dv = l_ret[s] - l_q[s];
        float dvout = -2.0f * g.wc * g.inv_b * dv;
        float sq = dv * dv;
        if (!valid) {
            dvout = 0.f;
            sq = 0.f;
        }
        l_dq[s] = dvout;
        float r0 = dvout, r1 = sq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            r0 += __shfl_down(r0, off, 64);
            r1 += __shfl_down(r1, off, 64);
        }
        if (lane == 0) {
            l_small[w * 8] = r0;
            l_small[w * 8 + 1] = r1;
        }
    }
    __syncthreads();
    float* outc = out + g.np_a;
    if (tid == 0) {
        outc[ob3c] = l_small[0] + l_small[8];
        store_wt(&g.loss_partials[(int64_t)blockIdx.x * 4 + 1], l_small[1] + l_small[9]);
    }
    mlp3_backward_tile<NS, 1, ACT>(mc, pkc + H3 * H3, h2, l_x, l_dq, l_red, l_A, l_B, l_C, outc, tid);
}

constexpr size_t GRAD3_LDS = (4 * TR + 2 * MAXO * TR + 4 * TR + 16 + 4 * 5 * H3 + 2 * SMALLW) * sizeof(float) +
                             3 * TILE_ELEMS * sizeof(uint16_t);

}  // namespace rlhip
#include "ppo3t_kernel.h"
namespace rlhip {
static bool g_ppo3_force128 = false;  // test hook (rlhip_debug_ppo3_force128): the round-1 tile for A / B comparisons

// both nets' W2 -> bf16 MFMA fragments: [actor W2jk | actor W2kj | critic W2jk | critic W2kj]
__global__ __launch_bounds__(256) void ppo3_pack_kernel(const float* __restrict__ params, int ns, int64_t np_a,
                                                        uint16_t* __restrict__ packed) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 2 * H3 * H3) return;
    const int net = q / (H3 * H3);
    q -= net * H3 * H3;
    const float* W2 = params + (net ? np_a : 0) + H3 * ns + H3;
    uint16_t* pk = packed + (int64_t)net * 2 * H3 * H3;
    int u = q & 7, l = (q >> 3) & 63, f = q >> 9;
    int t = f & 3, ks = f >> 2;
    int col = 32 * t + (l & 31), kk = 16 * ks + 8 * (l >> 5) + u;
    pk[q] = f32_to_bf16_rne(W2[col + H3 * kk]);
    pk[H3 * H3 + q] = f32_to_bf16_rne(W2[kk + H3 * col]);
}

// partial gradients [nb][np] -> grad[np] (64 parameters per workgroup, block range split over 4 waves, fixed order)
// and the four PPO loss numbers
__global__ __launch_bounds__(256) void ppo3_reduce_kernel(const float* __restrict__ partials,
                                                          const float* __restrict__ loss_partials, int nb, int np,
                                                          float* __restrict__ grad, float* __restrict__ losses, float wa,
                                                          float wc, float we, float inv_b) {
    __shared__ float l_g[4][64];
    __shared__ float l_loss[4];
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
    if (blockIdx.x == 0 && losses != nullptr) {
        if (grp < 3) {
            float a = 0.f;
            for (int b = lane; b < nb; b += 64) a += loss_partials[(int64_t)b * 4 + grp];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) l_loss[grp] = a;
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
}

template <typename K>
static int32_t allow_lds3(K kernel, size_t bytes, unsigned long long* done) { return allow_big_lds(kernel, bytes, done); }

static int32_t check3(int32_t kind, const rlhip_ppo_cfg* c, PolicyDesc* pd) {
    int32_t rc = make_desc(kind, c, pd);
    if (rc) return rc;
    RLHIP_REQUIRE(c->hidden == H3 || c->hidden == HWIDE, "layers = 3 (MFMA actor / critic) is built for hidden = 128 or 256");
    RLHIP_REQUIRE(pd->nout_a == 2, "layers = 3 supports CartPole (discrete, 2 actions) and Pendulum (continuous)");
    return RLHIP_OK;
}

template <class P>
static int32_t rollout3_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                             const PolicyDesc& pd, const float* params, uint64_t seed, uint32_t env_id_base,
                             uint32_t vec_step0, const rlhip_ppo_traj* traj, hipStream_t s) {
    RLHIP_REQUIRE(st && st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    typename P::cfg_t c2 = *cfg;
    c2.continuous = pd.cont;
    P p = P::make(c2);
    EnvArrays<float> a = EnvArrays<float>::from(*st);
    TrajPtrs tr = TrajPtrs::from(*traj);
    // up to 2^15 envs the 32-env workgroups fill more of the chip and have the shorter per-step chain
    const bool small = n <= (1 << 15);
    dim3 grid((unsigned)(small ? (n + R32 - 1) / R32 : (n + TR - 1) / TR));
#define LAUNCH_R3(ACT_)                                                                                           \
    do {                                                                                                          \
        if (small) {                                                                                              \
            static unsigned long long done32_ = 0;                                                                          \
            int32_t rc_ = allow_lds3(ppo3_rollout32_kernel<P, 2, ACT_>, ROLL32_LDS, &done32_);                    \
            if (rc_) return rc_;                                                                                  \
            hipLaunchKernelGGL((ppo3_rollout32_kernel<P, 2, ACT_>), grid, dim3(256), ROLL32_LDS, s, p, a, n, (int)T, \
                               pd.cont, pd.na, params, pd.np_a, seed, env_id_base, vec_step0, tr, pd.gamma, pd.lambda); \
            break;                                                                                                \
        }                                                                                                         \
        static unsigned long long done_ = 0;                                                                                \
        int32_t rc_ = allow_lds3(ppo3_rollout_kernel<P, 2, ACT_>, ROLL3_LDS, &done_);                             \
        if (rc_) return rc_;                                                                                      \
        hipLaunchKernelGGL((ppo3_rollout_kernel<P, 2, ACT_>), grid, dim3(256), ROLL3_LDS, s, p, a, n, (int)T, pd.cont, \
                           pd.na, params, pd.np_a, seed, env_id_base, vec_step0, tr, pd.gamma, pd.lambda);      \
    } while (0)
    if (pd.act == 0) LAUNCH_R3(0);
    else LAUNCH_R3(1);
#undef LAUNCH_R3
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// ---- host entry points used by the rlhip_ppo_* functions when cfg->layers == 3 ----
int64_t ppo3_nparams(int32_t kind, const rlhip_ppo_cfg* c) {
    PolicyDesc pd;
    if (check3(kind, c, &pd)) return -1;
    const int ns = kind == 0 ? 4 : (kind == 1 ? 3 : 2);
    if (c->hidden == HWIDE) return ppo3w_nparams(ns, pd.nout_a);
    return mlp3_np(ns, pd.nout_a) + mlp3_np(ns, 1);
}

static int64_t ppo3_nb(const rlhip_ppo_cfg* c, int64_t n, int64_t T) {
    const int64_t bm = (n * T) / (c->n_microbatches > 0 ? c->n_microbatches : 1);
    return (bm + TR - 1) / TR;
}

int64_t ppo3_workspace_bytes(int32_t kind, const rlhip_ppo_cfg* c, int64_t n, int64_t T) {
    const int64_t np = ppo3_nparams(kind, c);
    if (np < 0) return -1;
    if (c->hidden == HWIDE) {
        PolicyDesc pd;
        if (check3(kind, c, &pd)) return -1;
        return ppo3w_workspace_bytes(kind == 0 ? 4 : (kind == 1 ? 3 : 2), pd.nout_a, c, n, T);
    }
    const int64_t nb = ppo3_nb(c, n, T);
    return 4 * H3 * H3 * (int64_t)sizeof(uint16_t) + nb * (np + 4) * (int64_t)sizeof(float) + 256 + P3_TAIL_BYTES;
}

int32_t ppo3_rollout(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                     const rlhip_ppo_cfg* cfg, const float* params, uint64_t seed, uint32_t env_id_base,
                     uint32_t vec_step0, const rlhip_ppo_traj* traj, rlhip_stream_t stream) {
    PolicyDesc pd;
    int32_t rc = check3(kind, cfg, &pd);
    if (rc) return rc;
    const int ns = kind == 0 ? 4 : 3;
    RLHIP_REQUIRE(kind == 0 || kind == 1, "layers = 3 supports CartPole and Pendulum");
    pd.np_a = mlp3_np(ns, pd.nout_a);
    RLHIP_REQUIRE(env_cfg && st && params && traj, "NULL argument");
    RLHIP_REQUIRE(n >= 1 && n <= 0x7FFFFFFFll && T >= 0 && T <= 0x7FFFFFFFll, "bad n / T");
    RLHIP_REQUIRE(traj->obs && traj->logp && traj->value && traj->reward && traj->terminal, "trajectory array is NULL");
    RLHIP_REQUIRE(pd.cont ? (traj->action_f != nullptr) : (traj->action_i != nullptr), "action trace is NULL");
    if (cfg->hidden == HWIDE)
        return ppo3w_rollout(kind, env_cfg, st, n, T, pd, params, seed, env_id_base, vec_step0, traj, stream);
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return rollout3_impl<CartPoleParams<float>>((const rlhip_cartpole_cfg*)env_cfg, st, n, T, pd, params, seed,
                                                    env_id_base, vec_step0, traj, s);
    return rollout3_impl<PendulumParams<float>>((const rlhip_pendulum_cfg*)env_cfg, st, n, T, pd, params, seed,
                                                env_id_base, vec_step0, traj, s);
}

// one micro-batch: pack -> grad -> reduce; grad_out = summed gradient, losses_out (4) optional
struct P3Tail {  // optimise! state for the fused tail (ppo3_update); *fused_out tells the caller whether it ran
    float *params, *m, *v, *beta_pow;
    bool* fused_out;
};

static int32_t ppo3_grad_impl(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                              const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace,
                              float* grad_out, float* losses_out, bool do_pack, rlhip_stream_t stream, const P3Tail* tail);

int32_t ppo3_grad(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                  const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace, float* grad_out,
                  float* losses_out, bool do_pack, rlhip_stream_t stream) {
    return ppo3_grad_impl(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_out, losses_out, do_pack, stream,
                          nullptr);
}

static int32_t ppo3_grad_impl(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                              const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb, void* workspace,
                              float* grad_out, float* losses_out, bool do_pack, rlhip_stream_t stream, const P3Tail* tail) {
    PolicyDesc pd;
    int32_t rc = check3(kind, cfg, &pd);
    if (rc) return rc;
    RLHIP_REQUIRE(kind == 0 || kind == 1, "layers = 3 supports CartPole and Pendulum");
    RLHIP_REQUIRE(traj && params && workspace && grad_out, "NULL argument");
    RLHIP_REQUIRE(cfg->n_microbatches >= 1 && mb >= 0 && mb < cfg->n_microbatches, "micro-batch index out of range");
    if (cfg->hidden == HWIDE)  // (no fused tail: *tail->fused_out stays false and the caller runs clip + Adam)
        return ppo3w_grad(kind, cfg, pd, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_out, losses_out, stream);
    const int ns = kind == 0 ? 4 : 3;
    const int64_t total = n * T;
    RLHIP_REQUIRE(total >= 1 && total <= 0x7FFFFFFFll, "n * T out of range");
    const int64_t bm = total / cfg->n_microbatches;
    RLHIP_REQUIRE(bm >= 1, "empty micro-batch");
    const int64_t nb = (bm + TR - 1) / TR;
    RLHIP_REQUIRE(nb <= P3_MAX_BLOCKS, "micro-batch too large for one launch (layers = 3: <= 262144 samples)");
    hipStream_t s = as_stream(stream);
    P3Args g;
    g.obs = traj->obs;
    g.logp = traj->logp;
    g.adv = traj->adv;
    g.ret = traj->ret;
    g.action_f = traj->action_f;
    g.action_i = traj->action_i;
    RLHIP_REQUIRE(g.obs && g.logp && g.adv && g.ret && (pd.cont ? (const void*)g.action_f : (const void*)g.action_i),
                  "trajectory array is NULL");
    g.params = params;
    g.np_a = mlp3_np(ns, pd.nout_a);
    g.np = (int)(g.np_a + mlp3_np(ns, 1));
    uint16_t* packed = (uint16_t*)workspace;
    g.packed = packed;
    g.partials = (float*)((char*)workspace + 4 * H3 * H3 * sizeof(uint16_t));
    g.loss_partials = g.partials + nb * (int64_t)g.np;
    g.n = n;
    g.total = (uint32_t)total;
    g.bm = (uint32_t)bm;
    g.pos0 = (uint32_t)(mb * bm);
    g.na = pd.na;
    g.lo = 1.0f - cfg->clip_range;
    g.hi = 1.0f + cfg->clip_range;
    g.wa = cfg->actor_loss_weight;
    g.wc = cfg->critic_loss_weight;
    g.we = cfg->entropy_loss_weight;
    g.inv_b = 1.0f / (float)bm;
    g.min_logp = (float)::log(1e-8);
    g.pk = perm_keys(seed, epoch_ctr, (uint32_t)total);
    if (do_pack)
        hipLaunchKernelGGL(ppo3_pack_kernel, dim3(2 * H3 * H3 / 256), dim3(256), 0, s, params, ns, g.np_a, packed);
    // relu: the register-chained tile (ppo3t_kernel.h), persistent workgroups, one partial row per workgroup and net.
    // tanh: the round-1 kernel (one 128-row tile per workgroup) -- the chained tile needs 64 more live registers for
    // act'(h1) there, spills ~300 dwords per lane and measured 693 us against 326 us (profiles/r02_ppo3_gradT.md).
    // rlhip_debug_ppo3_force128 (test hook) keeps the round-1 kernel for relu too: A / B comparisons in tests/.
    const bool chained = pd.act == 0 && !g_ppo3_force128;
    const int ntiles = (int)nb;
    static int t3_wg_cap = -1;
    if (t3_wg_cap < 0) {
        const char* e = getenv("RLHIP_PPO3_WGS");  // test hook: many tiles per persistent workgroup at small sizes
        t3_wg_cap = e ? atoi(e) : 128;
        if (t3_wg_cap < 1 || t3_wg_cap > 1024) t3_wg_cap = 128;
    }
    const int nwg = ntiles < t3_wg_cap ? ntiles : t3_wg_cap;  // per net
    const int nrows = chained ? nwg : (int)nb;
    g.loss_partials = g.partials + (int64_t)nrows * g.np;
#define LAUNCH_G3T(NS_, CONT_)                                                                            \
    do {                                                                                                  \
        static unsigned long long donet_ = 0;                                                                       \
        int32_t rc_ = allow_lds3(ppo3_gradT_kernel<NS_, 0, CONT_>, GRADT_LDS, &donet_);                   \
        if (rc_) return rc_;                                                                              \
        hipLaunchKernelGGL((ppo3_gradT_kernel<NS_, 0, CONT_>), dim3(2 * nwg), dim3(256), GRADT_LDS, s, g, nwg, ntiles); \
    } while (0)
#define LAUNCH_G3(NS_, ACT_, CONT_)                                                                       \
    do {                                                                                                  \
        static unsigned long long done_ = 0;                                                                        \
        int32_t rc_ = allow_lds3(ppo3_grad_kernel<NS_, 2, ACT_, CONT_>, GRAD3_LDS, &done_);               \
        if (rc_) return rc_;                                                                              \
        hipLaunchKernelGGL((ppo3_grad_kernel<NS_, 2, ACT_, CONT_>), dim3((int)nb), dim3(256), GRAD3_LDS, s, g); \
    } while (0)
    if (kind == 0) {
        RLHIP_REQUIRE(!pd.cont, "layers = 3: CartPole uses the categorical head");
        if (chained) LAUNCH_G3T(4, 0);
        else if (pd.act == 0) LAUNCH_G3(4, 0, 0);
        else LAUNCH_G3(4, 1, 0);
    } else {
        RLHIP_REQUIRE(pd.cont, "layers = 3: Pendulum uses the Gaussian head");
        if (chained) LAUNCH_G3T(3, 1);
        else if (pd.act == 0) LAUNCH_G3(3, 0, 1);
        else LAUNCH_G3(3, 1, 1);
    }
#undef LAUNCH_G3
#undef LAUNCH_G3T
    if (tail) {
        // reduce + PPO loss line + Float64 norm + clip + Adam + bf16 re-pack of both W2 in one launch (dqn3.hip)
        char* tp = (char*)workspace + ppo3_workspace_bytes(kind, cfg, n, T) - P3_TAIL_BYTES;
        tp = (char*)(((uintptr_t)tp + 63) & ~(uintptr_t)63);
        const int32_t rcf = ppo3_apply_fused(g.partials, g.loss_partials, nrows, g.np, (int)g.np_a, ns, grad_out, losses_out,
                                             g.inv_b, g.wa, g.wc, g.we, tail->params, tail->m, tail->v, tail->beta_pow,
                                             packed, tp, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2, cfg->adam_eps,
                                             s);
        if (rcf < 0) return rcf;
        *tail->fused_out = rcf == 0;
        if (rcf == 0) return RLHIP_OK;
    }
    hipLaunchKernelGGL(ppo3_reduce_kernel, dim3((g.np + 63) / 64), dim3(256), 0, s, g.partials, g.loss_partials, nrows,
                       g.np, grad_out, losses_out, g.wa, g.wc, g.we, g.inv_b);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

/* test hook, not part of include/rlhip.h: 1 = the round-1 128-row tile for every PPO layers = 3 gradient */
extern "C" int32_t rlhip_debug_ppo3_force128(int32_t on) {
    g_ppo3_force128 = on != 0;
    return RLHIP_OK;
}

#ifdef RLHIP_T3_TIMING
extern "C" int32_t rlhip_debug_t3_stamps(long long* out_host) {
    RLHIP_CHECK_HIP(hipDeviceSynchronize());
    RLHIP_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_t3_stamps), 16 * sizeof(long long)));
    return RLHIP_OK;
}
#endif

int32_t ppo3_update(int32_t kind, const rlhip_ppo_cfg* cfg, int64_t n, int64_t T, const rlhip_ppo_traj* traj,
                    float* params, float* m, float* v, float* beta_pow, uint64_t seed, uint32_t update_ctr,
                    void* workspace, float* grad_scratch, float* losses_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(cfg && params && m && v && beta_pow && grad_scratch, "NULL argument");
    const int64_t np = ppo3_nparams(kind, cfg);
    RLHIP_REQUIRE(np > 0, "bad configuration");
    if (cfg->hidden == HWIDE) {
        PolicyDesc pd;
        int32_t rc = check3(kind, cfg, &pd);
        if (rc) return rc;
        RLHIP_REQUIRE(kind == 0 || kind == 1, "layers = 3 supports CartPole and Pendulum");
        RLHIP_REQUIRE(traj && workspace && cfg->n_microbatches >= 1, "NULL argument");
        return ppo3w_update(kind, cfg, pd, n, T, traj, params, m, v, beta_pow, seed, update_ctr, workspace, grad_scratch,
                            losses_out, stream);
    }
    bool packed_fresh = false;  // the previous optimiser step's fused tail left the bf16 W2 images up to date
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        const uint32_t epoch_ctr = update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            bool fused = false;
            const P3Tail tail{params, m, v, beta_pow, &fused};
            int32_t rc = ppo3_grad_impl(kind, cfg, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_scratch,
                                        losses_out, /*do_pack=*/!packed_fresh, stream, &tail);
            if (rc) return rc;
            packed_fresh = fused;
            if (!fused) {
                rc = rlhip_clip_adam_f32(params, grad_scratch, m, v, beta_pow, np, 1.0f, cfg->max_grad_norm, cfg->lr,
                                         cfg->beta1, cfg->beta2, cfg->adam_eps, nullptr, stream);
                if (rc) return rc;
            }
        }
    }
    return RLHIP_OK;
}

}  // namespace rlhip
