// ppo3w.hip -- the three-layer networks at hidden width 256 (layers = 3, hidden = 256), for both learners:
//     PPO   actor  ns -> 256 -> 256 -> nout_a,  critic  ns -> 256 -> 256 -> 1      (rlhip_ppo_*: ppo3.hip dispatches here)
//     DQN   Q-network  ns -> 256 -> 256 -> na  (online + target)                   (rlhip_dqn3_*: dqn3.hip dispatches here)
// Same reference code, precision contract and oracle as ppo3.hip / dqn3.hip (bf16 operands / f32 accumulate on
// v_mfma_f32_32x32x16_bf16 for the hidden x hidden layer and its two backward GEMMs, f32 master weights, everything else
// f32; oracle/rlo_learn.c, oracle/rlo_mlp3.c with h = 256), behind the unchanged entry points.
//
// Why a separate design.  At 128 the learner tiles keep both bf16 images of W2 (64 KB) in LDS and the whole forward -> loss
// -> backward chain of a sample tile inside one workgroup.  At 256 one net's two images are 256 KB: they do not fit the
// 160 KB of LDS, and the dW2 accumulators alone (256 x 256 f32) are 128 registers per lane of an 8-wave workgroup.  So the
// width-256 learner is THREE streaming kernels per net, each with exactly one operand resident in registers for the
// lifetime of a persistent workgroup (the scheme of dense_persist_kernel, dense_mfma.hip), 8 waves per workgroup, wave w
// owning the 32 output columns [32 w, 32 w + 32) of its GEMM, 64-sample tiles, ONE workgroup per CU:
//   ppo3w_fwd_kernel  W2 fragments resident (64 VGPRs).  layer 1 (VALU, the observation in registers: lane = sample) ->
//                     H1 tile in LDS -> MFMA -> H2 in registers -> head through a wave-private LDS transposition of the
//                     wave's own 64 x 32 block -> loss line per sample on wave 0 (modes: PPO actor, PPO critic, DQN target
//                     network = forward only, DQN online network) -> dZ2 in the MFMA D layout; db2 / dW3 / db3 and the loss
//                     sums stay in registers across tiles.  dZ2 leaves as bf16 ONCE since round 6, in MFMA fragment order (the
//                     D registers as they are: 512 B contiguous per store instruction, no transposition), one buffer per net, kept
//                     until the dW2 launch.  (Rounds 2 - 5 wrote a row-major image for the backward kernel as well -- half of
//                     the kernel's store bytes: the tile's stores are a serialised ~2000-cycle resource of the CU, and one image
//                     less took 7 us off each forward launch; RLHIP_W3_DZ_ONCE above.)
//   ppo3w_bwd_kernel  W2^T fragments resident.  dZ2 fragment tile -> LDS, copied lane-linear (double-buffered, register-staged two
//                     passes ahead); the A operand (lane = sample row) is read out of it TRANSPOSED: two ds_read_b64_tr_b16 per
//                     fragment -> MFMA -> dH1 in registers; z1 is recomputed from the observation (ns <= 4 FMAs per element:
//                     cheaper than 2 bytes of HBM), dz1, db1 / dW1 in registers across tiles; one barrier per pass.
//   ppo3w_dw2_kernel  the f32 accumulator resident: a workgroup owns one half of the k range (128 x 256 outputs = 64
//                     registers per lane) for a strided set of sample tiles; A = H1^T recomputed into LDS in [k][sample]
//                     order, B = the dZ2 fragments as the forward kernel stored them (16-byte loads, two tiles in flight); the two
//                     halves of a sample range run on ONE XCD so that its L2 serves the second read.
// Tile inputs come from a once-per-step gather in sample order (ppo3w_gather*_kernel / dqn3w_gather_kernel) as coalesced
// wave loads at addresses that depend on the tile index only, issued unconditionally by every wave (see load_x).
// Partial gradients are rows (one per persistent workgroup) summed in a fixed order (ppo3w_reduce_kernel, or the two-launch
// tail ppo3w_reduce_sumsq_kernel + ppo3w_adam_pack_kernel): deterministic, no atomics, no grid barrier.
// Measurements, counters and the negative results: profiles/r02_ppo3w.md; DESIGN.md section 5.
// ppo3w_rollout_kernel  the rollout of ppo3.hip's 32-env workgroups at width 256: 8 waves, BOTH nets' W2 fragments in
//                     registers (2 x 64 VGPRs, converted from the f32 master weights at kernel entry) for all T vec-steps,
//                     f32 H2 tiles in LDS for the heads.      dqn3w_plan_kernel  plan! of the Q-network (forward + eps-greedy).
#include "env_device.h"
#include "ppo_common.h"
#include "ring_device.h"
#include "ppo_sample_device.h"
#include "mlp3_device.h"
#include "optim_device.h"
#include <type_traits>
#include <mutex>
#include <dirent.h>
#include <unistd.h>
#include <time.h>
#include <ctype.h>
#include <string.h>

extern "C" int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow, int64_t n,
                                       float grad_scale, float clip_norm, float lr, float beta1, float beta2,
                                       float eps, float* gn_out, rlhip_stream_t stream);

namespace rlhip {

constexpr int HW = 256;        // hidden width
constexpr int WV = HW / 32;    // 8 waves; wave w owns output columns [32 w, 32 w + 32)
constexpr int NTW = 64 * WV;   // 512 threads
constexpr int RW = 64;         // samples per tile (two 32-row MFMA tiles)
constexpr int PW = HW + 8;     // bf16 pitch of a [sample][feature] tile: 528 B
constexpr int KSW = HW / 16;   // MFMA k-steps across the hidden width
constexpr int PT = RW + 8;     // bf16 pitch of a [feature][sample] tile: 144 B
constexpr int SMALLWW = HW * 4 + HW + HW + MAXO * HW + MAXO + 4;  // floats: W1 | b1 | b2 | W3 | b3 of one net (ns <= 4)

__host__ __device__ __forceinline__ int64_t mlp3w_np(int64_t ns, int64_t nout) {
    return HW * ns + HW + (int64_t)HW * HW + HW + nout * HW + nout;
}
// "small" parameter space of one net = everything except W2, in parameter order: W1 | b1 | b2 | W3 | b3
__host__ __device__ __forceinline__ int mlp3w_ns_small(int ns, int nout) { return HW * ns + 2 * HW + nout * HW + nout; }

// per-phase cycle stamps of one steady-state tile (workgroup 0, thread 0, its second / third tile): -DRLHIP_W3_TIMING.
// PROPORTIONS ONLY: the stamps change the register allocation (the backward kernel spilled 584 bytes per lane in one timing
// build and ran 4x slower than the shipped one) -- kernel times come from rocprofv3 on the normal build.
// RLHIP_W3_DZ_ONCE (round 6, VERDICT r5 item 2 "dZ2 written once"): the forward kernel writes dZ2 in ONE layout (rounds 2 - 5: two, rows
// for the backward kernel and MFMA B-fragment order for the dW2 kernel; -128 MB written and -128 MB read per optimiser step of a PPO pair,
// half of the forward kernel's store bytes -- and the tile's stores are what that kernel's passes wait for).  Which one:
//   2 (shipped)  the FRAGMENT image: the forward kernel's D registers leave as they are (no transposition through the wave's private LDS
//                block), the dW2 kernel loads its B operand as in rounds 2 - 5, and the BACKWARD kernel -- whose A operand is the transposed
//                view, lane = sample row -- copies the tile lane-linear into LDS and reads it with ds_read_b64_tr_b16 (two per fragment)
//   1            the ROW image: forward transposes, backward reads rows as in rounds 2 - 5, the dW2 kernel stages the rows in LDS and
//                gathers its fragments with the transposing reads (first form of the round; 9 - 12 us per optimiser step slower than 2)
//   0            both images (rounds 2 - 5), kept for A / B
// RLHIP_W3_DZF_PAD (mode 2): four 16-byte slots of padding behind every 32 of the backward kernel's LDS copy make the transposing reads
// bank-conflict free: the kernel -3 us per launch (30.5 -> 27.5), 5 % fewer cycles per optimiser step.  On boxes without an active clock limiter
// that is -9 us per step (188.5 -> 179.3); on the others the firmware then runs the whole step at a LOWER clock (2.11 - 2.21 vs 2.29 - 2.35 GHz at
// 1.05 vs 1.16 kW under a 1.4 kW cap) and the step is 2 - 3 % SLOWER.  Both kernels are always built; the macro / the environment variable of
// the same name select 0 = unpadded, 1 = padded, 2 ("auto", the default) = by the chip's clock (W3Pad below); profiles/r06_ppo3w.md section 5.
#ifndef RLHIP_W3_DZ_ONCE
#define RLHIP_W3_DZ_ONCE 2
#endif
#ifndef RLHIP_W3_DZF_PAD
#define RLHIP_W3_DZF_PAD 2
#endif
#ifndef RLHIP_W3_FWD_T
#define RLHIP_W3_FWD_T 1  // the forward-only modes of ppo3w_fwd_kernel compute layer 2 transposed: the head in-lane, no LDS transposition (0: as the others)
#endif
#ifndef RLHIP_W3_TIMING_NET
#define RLHIP_W3_TIMING_NET 0  // which mode of ppo3w_fwd_kernel a -DRLHIP_W3_TIMING build stamps (0 PPO actor ... 2 DQN target network, forward only)
#endif
#ifdef RLHIP_W3_TIMING
__device__ long long g_w3_stamps[3][16];
#define W3_STAMP(kern, k)                                                                                  \
    do {                                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (int)(W3_STRIDE)) g_w3_stamps[kern][(k)] = clock64(); \
    } while (0)
#define W3_MARK(kern, k, on)                                                                 \
    do {                                                                                     \
        if ((on) && blockIdx.x == 0 && threadIdx.x == 0) g_w3_stamps[kern][(k)] = clock64(); \
    } while (0)
#else
#define W3_STAMP(kern, k) \
    do {                  \
    } while (0)
#define W3_MARK(kern, k, on) \
    do {                     \
    } while (0)
#endif

typedef short tr_v4s __attribute__((ext_vector_type(4)));  // ds_read_b64_tr_b16 result: four 16-bit elements

struct Mlp3W {
    const float *W1, *b1, *b2, *W3, *b3;
};

// W1 | b1 and b2 | W3 | b3 (the two parameter ranges around W2) -> LDS.  Caller barriers before use.
__device__ __forceinline__ Mlp3W stage_small_w(const float* __restrict__ p, int ns, int nout, float* l_w, int tid) {
    const int n1 = HW * ns + HW;
    const int n2 = HW + nout * HW + nout;
    const float* p2 = p + n1 + HW * HW;
    for (int i = tid; i < n1; i += NTW) l_w[i] = p[i];
    for (int i = tid; i < n2; i += NTW) l_w[n1 + i] = p2[i];
    Mlp3W v;
    v.W1 = l_w;
    v.b1 = l_w + HW * ns;
    v.b2 = l_w + n1;
    v.W3 = v.b2 + HW;
    v.b3 = v.W3 + nout * HW;
    return v;
}

struct P3WArgs {
    const float* obs;
    const float* logp;
    const float* adv;
    const float* ret;
    const float* action_f;
    const int32_t* action_i;
    const float* params;     // [actor | critic]
    const uint16_t* packed;  // actor W2jk | actor W2kj | critic W2jk | critic W2kj   (MFMA B-fragment order, HW * HW each)
    const float* tparams;    // DQN: the target network's parameters (forward mode 2)
    const uint16_t* tpacked;  // DQN: its W2 fragments (W2jk | W2kj)
    float* xg2;              // DQN: [NS][npad] the next observations s'
    float* td_out;           // DQN: optional |Q(s, a) - y| per sample (priority write-back)
    const float* isw;        // DQN: optional importance-sampling weights per sample (prioritized replay): mean(w .* huber)
    float gamma, delta;      // DQN: discount, Huber threshold
    const float* rec;        // [n T][8] {x0..x3, old log-prob, advantage, return, action}: ppo3w_update's record copy, or NULL
    float* xg;               // [NS][npad] the micro-batch's observations in sample order (ppo3w_gather_kernel), npad = ntiles RW
    float* sg;               // [4][npad]  old log-prob | advantage (0 on padding) | return | action (float or int bits)
    uint16_t* dz_rows;       // [ntiles * RW][HW] bf16
    uint16_t* dz_frag;       // [nets][ntiles][RW / 16][WV][64 lanes][8] bf16 (PPO: one buffer per net, frag_stride apart)
    int64_t frag_stride;     // elements between the nets' fragment buffers (0: one net)
    float* partS;            // [rows][npS]: partial gradients of the small tensors, [actor small | critic small]
    float* partW;            // [rows][2][HW * HW]: partial dW2 (Flux order W2[j + HW k])
    float* loss_partials;    // [rows][4] {sum min(surr1, surr2), sum (ret - v)^2, sum entropy, -}
    int64_t n, np_a;
    uint32_t total, bm, pos0;
    int ntiles, npS, nS_a, na, npad, wnets;  // wnets: nets per partial dW2 row (2: actor | critic; 1: the DQN learner)
    float lo, hi, wa, wc, we, inv_b, min_logp;
    PermKeys pk;
};

// both nets' W2 -> bf16 MFMA B fragments.  Fragment (ks, t) = 64 lanes x 8 elements, lane l, element u:
//   W2jk (forward, H2 = H1 W2^T):   B[k = 16 ks + 8 (l >> 5) + u][col j = 32 t + (l & 31)] = W2[j + HW k]
//   W2kj (backward, dH1 = dZ2 W2):  B[j = 16 ks + 8 (l >> 5) + u][col k = 32 t + (l & 31)] = W2[j + HW k]
__global__ __launch_bounds__(256) void ppo3w_pack_kernel(const float* __restrict__ params, int ns, int64_t np_a,
                                                         uint16_t* __restrict__ packed) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 2 * HW * HW) return;
    const int net = q / (HW * HW);
    q -= net * HW * HW;
    const float* W2 = params + (net ? np_a : 0) + HW * ns + HW;
    uint16_t* pk = packed + (int64_t)net * 2 * HW * HW;
    const int u = q & 7, l = (q >> 3) & 63, f = q >> 9;
    const int t = f % WV, ks = f / WV;
    const int col = 32 * t + (l & 31), kk = 16 * ks + 8 * (l >> 5) + u;
    pk[q] = f32_to_bf16_rne(W2[col + HW * kk]);
    pk[HW * HW + q] = f32_to_bf16_rne(W2[kk + HW * col]);
}

__device__ __forceinline__ void load_frags_w(const uint16_t* __restrict__ wf, int w, int lane, bf16x8 (&bw)[KSW]) {
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) bw[ks] = *reinterpret_cast<const bf16x8*>(wf + ((int64_t)(ks * WV + w) * 64 + lane) * 8);
}

// the same fragments converted on the fly from the f32 master weights (rollout: once per launch, no packed image needed):
// lane l of wave w, k-step ks: W2[j = 32 w + (l & 31)][k = 16 ks + 8 (l >> 5) + u], u = 0..7 (128 B coalesced per load)
__device__ __forceinline__ void load_frags_f32(const float* __restrict__ W2, int w, int lane, bf16x8 (&bw)[KSW]) {
    const float* src = W2 + 32 * w + (lane & 31) + HW * 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[HW * (16 * ks + u)];
        const uint4 pk = pack8_bf16(v);
        bw[ks] = __builtin_bit_cast(bf16x8, pk);
    }
}

// The shuffled micro-batch is gathered ONCE per optimiser step into sample-ordered arrays (6 tile kernels read it): the
// keyed permutation and the dependent random loads (~4000 cycles, measured 19 % of a forward tile and 56 - 65 % of the
// backward / dW2 tiles when done in place) leave the tile loops, whose inputs become coalesced 256-byte wave loads at
// addresses that depend on the tile index only.
template <int NS, int CONT>
__global__ __launch_bounds__(256) void ppo3w_gather_kernel(P3WArgs g) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (uint32_t)g.npad) return;
    const bool valid = q < g.bm;
    const uint32_t f = permute(g.pk, g.pos0 + (valid ? q : 0u));
    const uint32_t t = f / (uint32_t)g.n, i = f - t * (uint32_t)g.n;
#pragma unroll
    for (int k = 0; k < NS; ++k) g.xg[(int64_t)k * g.npad + q] = g.obs[((int64_t)t * NS + k) * g.n + i];
    g.sg[q] = g.logp[f];
    g.sg[(int64_t)g.npad + q] = valid ? g.adv[f] : 0.0f;
    g.sg[2 * (int64_t)g.npad + q] = g.ret[f];
    g.sg[3 * (int64_t)g.npad + q] = CONT ? g.action_f[f] : __int_as_float(g.action_i[f]);
}

// optimise! walks the trajectory 16 times (4 epochs x 4 micro-batches) in shuffled order: seven 4-byte reads from seven arrays
// per sample = seven 64-byte sectors, 58 MB of sector traffic for 3.7 MB of payload, 13 us per optimiser step.  ppo3w_update
// therefore copies the trajectory ONCE per update into 32-byte records (coalesced, 31 MB, ~8 us); the shuffled gather then
// touches one sector per sample.
template <int NS, int CONT>
__global__ __launch_bounds__(256) void ppo3w_build_rec_kernel(P3WArgs g, float* __restrict__ rec) {
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;
    if (f >= g.total) return;
    const uint32_t t = f / (uint32_t)g.n, i = f - t * (uint32_t)g.n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b;
    a.x = g.obs[((int64_t)t * NS + 0) * g.n + i];
    a.y = g.obs[((int64_t)t * NS + 1) * g.n + i];
    if (NS > 2) a.z = g.obs[((int64_t)t * NS + 2) * g.n + i];
    if (NS > 3) a.w = g.obs[((int64_t)t * NS + 3) * g.n + i];
    b.x = g.logp[f];
    b.y = g.adv[f];
    b.z = g.ret[f];
    b.w = CONT ? g.action_f[f] : __int_as_float(g.action_i[f]);
    reinterpret_cast<float4*>(rec)[2 * (int64_t)f] = a;
    reinterpret_cast<float4*>(rec)[2 * (int64_t)f + 1] = b;
}

template <int NS>
__global__ __launch_bounds__(256) void ppo3w_gather_rec_kernel(P3WArgs g) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (uint32_t)g.npad) return;
    const bool valid = q < g.bm;
    const uint32_t f = permute(g.pk, g.pos0 + (valid ? q : 0u));
    const float4 a = reinterpret_cast<const float4*>(g.rec)[2 * (int64_t)f];
    const float4 b = reinterpret_cast<const float4*>(g.rec)[2 * (int64_t)f + 1];
    const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < NS; ++k) g.xg[(int64_t)k * g.npad + q] = x[k];
    g.sg[q] = b.x;
    g.sg[(int64_t)g.npad + q] = valid ? b.y : 0.0f;
    g.sg[2 * (int64_t)g.npad + q] = b.z;
    g.sg[3 * (int64_t)g.npad + q] = b.w;
}

// lane = sample of the tile.  Every wave issues these loads (uniform instruction streams keep the compiler's vmcnt
// bookkeeping exact across the pass loop; a wave-conditional load made it fall back to vmcnt(0), which exposed the latency
// of the youngest prefetch instead of the oldest)
template <int NS>
__device__ __forceinline__ void load_x_from(const float* __restrict__ xg, int npad, int tile, int lane, float (&x)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = xg[(int64_t)k * npad + (int64_t)tile * RW + lane];
}
// the same from the time-major trajectory traces obs (T + 1, NS, n): sample q = t n + env (value pass, NET == 4)
template <int NS>
__device__ __forceinline__ void load_x_traj(const float* __restrict__ obs, int64_t n, uint32_t total, int tile, int lane,
                                            float (&x)[NS]) {
    uint32_t q = (uint32_t)tile * RW + (uint32_t)lane;
    q = q < total ? q : total - 1;  // padding rows of the last tile re-read the last sample (their output is not stored)
    const uint32_t t = q / (uint32_t)n, e = q - t * (uint32_t)n;
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k] = obs[((int64_t)t * NS + k) * n + e];
}
// The observations of a tile as the B operand of layer 1 on v_mfma_f32_32x32x2_f32 (D[unit][sample] = b1 + W1 x):
// lane (r = lane & 31, kb = lane >> 5) holds x[component kb + 2 ks] of sample 32 rt + r in xm[rt][ks] -- NS loads per lane
// like the row-per-lane layout, straight from the component planes.  A component index >= NS (odd NS) is clamped to a real
// component; its A operand (the W1 column) is zero.
template <int NS>
__device__ __forceinline__ void load_xm_from(const float* __restrict__ xg, int npad, int tile, int lane,
                                             float (&xm)[2][(NS + 1) / 2]) {
    const int r = lane & 31, kb = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < (NS + 1) / 2; ++ks) {
            const int kk = min(2 * ks + kb, NS - 1);
            xm[rt][ks] = xg[(int64_t)kk * npad + (int64_t)tile * RW + 32 * rt + r];
        }
}
template <int NS>
__device__ __forceinline__ void load_xm_traj(const float* __restrict__ obs, int64_t n, uint32_t total, int tile, int lane,
                                             float (&xm)[2][(NS + 1) / 2]) {
    const int r = lane & 31, kb = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        uint32_t q = (uint32_t)tile * RW + (uint32_t)(32 * rt + r);
        q = q < total ? q : total - 1;
        const uint32_t t = q / (uint32_t)n, e = q - t * (uint32_t)n;
#pragma unroll
        for (int ks = 0; ks < (NS + 1) / 2; ++ks) {
            const int kk = min(2 * ks + kb, NS - 1);
            xm[rt][ks] = obs[((int64_t)t * NS + kk) * n + e];
        }
    }
}
template <int NS>
__device__ __forceinline__ void load_x(const P3WArgs& g, int tile, int lane, float (&x)[NS]) {
    load_x_from<NS>(g.xg, g.npad, tile, lane, x);
}
// this wave's private copy [NS][RW] of a tile's observations
template <int NS>
__device__ __forceinline__ void store_x(float* l_xw, int lane, const float (&x)[NS]) {
#pragma unroll
    for (int k = 0; k < NS; ++k) l_xw[k * RW + lane] = x[k];
}
// N floats global -> LDS by the whole workgroup, every load issued before the first store
template <int N>
__device__ __forceinline__ void copy_to_lds(const float* __restrict__ src, float* dst, int tid) {
    constexpr int IT = (N + NTW - 1) / NTW;
    float v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = (tid + NTW * i < N) ? src[tid + NTW * i] : 0.0f;
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (tid + NTW * i < N) dst[tid + NTW * i] = v[i];
}
template <int NS, int NOUT>
__device__ __forceinline__ Mlp3W stage_small_w2(const float* __restrict__ p, float* l_w, int tid) {
    constexpr int n1 = HW * NS + HW, n2 = HW + NOUT * HW + NOUT;
    copy_to_lds<n1>(p, l_w, tid);
    copy_to_lds<n2>(p + n1 + HW * HW, l_w + n1, tid);
    Mlp3W v;
    v.W1 = l_w;
    v.b1 = l_w + HW * NS;
    v.b2 = l_w + n1;
    v.W3 = v.b2 + HW;
    v.b3 = v.W3 + NOUT * HW;
    return v;
}

// LDS exchange inside ONE wave (its private block): LDS instructions of a wave execute in order, so no workgroup barrier
// is needed -- only the compiler has to keep the accesses in program order (the block is viewed as f32 and as bf16)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------ forward + loss + dZ2
constexpr int TPW = 36;  // f32 pitch of a wave's private 64 x 32 transposition block (144 B rows)
[[maybe_unused]] constexpr int ZPW = 40;  // bf16 pitch of the same block when it holds the wave's dZ2 columns (80 B rows)
constexpr size_t FWDW_LDS = (MAXO * RW + WV * MAXO * RW + SMALLWW + WV * RW * TPW) * sizeof(float) +
                            (size_t)RW * PW * sizeof(uint16_t);

template <int NS, int NOUT, int ACT, int CONT, int NET>
__global__ __launch_bounds__(NTW, 2) void ppo3w_fwd_kernel(P3WArgs g) {
    // NET: 0 PPO actor, 1 PPO critic (second net of the pair), 2 DQN target network (forward only: TD target y -> sg[0]),
    // 3 DQN online network (Huber loss on Q(s, a) - y), 4 PPO critic forward only over the trajectory's observations
    // (the rollout's batched value pass: V(s) -> sg[q], sg = the value trace; W2 fragments converted from the f32 master
    // weights, no packed image needed).  Per-sample inputs sg[0..3]: PPO {old log-prob, advantage, return,
    // action}; DQN {y, reward, terminal (0 / 1), action bits}
    W3_MARK(0, 8, NET == RLHIP_W3_TIMING_NET);
    extern __shared__ __attribute__((aligned(16))) char smw[];
    float* l_dq = reinterpret_cast<float*>(smw);  // [MAXO][RW] dL/d(head outputs)
    float* l_part = l_dq + MAXO * RW;            // [WV][MAXO][RW] head partial sums per wave
    float* l_w = l_part + WV * MAXO * RW;        // [SMALLWW]
    float* l_t = l_w + SMALLWW;                  // [WV][RW][TPW] wave-private transposition blocks
    uint16_t* l_H = reinterpret_cast<uint16_t*>(l_t + WV * RW * TPW);  // [RW][PW] H1 rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col = 32 * w + r;
    constexpr bool FWD_T = RLHIP_W3_FWD_T && (NET == 2 || NET == 4);  // forward-only modes compute layer 2 transposed (see the MFMA loop)
    float* l_tw = l_t + w * RW * TPW;
#if RLHIP_W3_DZ_ONCE != 2
    uint16_t* l_zw = reinterpret_cast<uint16_t*>(l_tw);  // the wave's dZ2 columns as bf16 rows (the row image's copy-out)
#endif
    const float* pnet = NET == 2 ? g.tparams : g.params + ((NET == 1 || NET == 4) ? g.np_a : 0);
    const float* xsrc = NET == 2 ? g.xg2 : g.xg;
    // every global load of the prologue is issued before the first wait: fragments, tile 0's inputs, the small tensors
    bf16x8 bw[KSW];
    if (NET == 4) load_frags_f32(pnet + HW * NS + HW, w, lane, bw);
    else load_frags_w(NET == 2 ? g.tpacked : g.packed + (NET == 1 ? 2 * HW * HW : 0), w, lane, bw);
    const int stride = gridDim.x, last = g.ntiles - 1;
    // lane = sample row: the observation feeds layer 1 straight from registers (row1 == lane); log-prob / advantage / action
    // or the return feed the loss line on wave 0.  Each is re-requested for the NEXT tile right after its last use, i.e.
    // most of a pass (~9000 cycles) ahead of its next use.
    constexpr int KS1 = (NS + 1) / 2;
    float xr[2][KS1], sr0, sr1 = 0.0f, sr2 = 0.0f;
    if (NET == 4) load_xm_traj<NS>(g.obs, g.n, g.bm, blockIdx.x, lane, xr);
    else load_xm_from<NS>(xsrc, g.npad, blockIdx.x, lane, xr);
    auto load_s = [&](int tile_) __attribute__((always_inline)) {
        const int64_t q0 = (int64_t)tile_ * RW + lane;
        if (NET == 4) {
        } else if (NET == 0) {
            sr0 = g.sg[q0];
            sr1 = g.sg[(int64_t)g.npad + q0];
            sr2 = g.sg[3 * (int64_t)g.npad + q0];
        } else if (NET == 1) {
            sr0 = g.sg[2 * (int64_t)g.npad + q0];
        } else if (NET == 2) {
            sr0 = g.sg[(int64_t)g.npad + q0];
            sr1 = g.sg[2 * (int64_t)g.npad + q0];
        } else {
            sr0 = g.sg[q0];
            sr2 = g.sg[3 * (int64_t)g.npad + q0];
        }
    };
    load_s(blockIdx.x);
    const Mlp3W m = stage_small_w2<NS, NOUT>(pnet, l_w, tid);
    __syncthreads();
    W3_MARK(0, 9, NET == RLHIP_W3_TIMING_NET);
    const float b2v = m.b2[col];
    float w3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) w3[o] = m.W3[o + NOUT * col];
    float a_db2 = 0.0f, a_dw3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a_dw3[o] = 0.0f;
    float s_red[NOUT + 2];  // wave 0, one sample row per lane: sum dq[o] (= db3), loss terms
#pragma unroll
    for (int o = 0; o < NOUT + 2; ++o) s_red[o] = 0.0f;
    // layer 1 on the f32 MFMA, D[unit][sample]: A = this wave's 32 rows of W1 (lane r = unit 32 w + r, component kb + 2 ks),
    // C = b1 by register row -- both loop-invariant
    const int u0 = 32 * w;
    float w1a[KS1], b1q[16];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int kk = 2 * ks + kb;
        const float v = m.W1[u0 + r + HW * min(kk, NS - 1)];
        w1a[ks] = kk < NS ? v : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) b1q[q] = m.b1[u0 + mfma_row(q, kb)];

#define W3_STRIDE (NET == RLHIP_W3_TIMING_NET ? gridDim.x : 0x7fffffff)
    int it = 0;
    for (int tile = blockIdx.x; tile < g.ntiles; tile += stride, ++it) {
        W3_STAMP(0, 0);
        const int tnext = min(tile + stride, last);  // (clamped: the tail re-reads the last tile instead of branching)
        W3_STAMP(0, 1);
        // ---- layer 1: h1 = act(b1 + W1 x), bf16 rows.  v_mfma_f32_32x32x2_f32 with the bias as the accumulator's initial
        //      value is the fmaf chain of mlp2 / the oracle bit for bit (tools/micro/mfma_f32_l1.hip); lane (r, kb) ends up
        //      with sample 32 rt + r's units u0 + 8 g + 4 kb + {0..3}: four 8-byte stores per 32 samples ----
        {
            float x[2][KS1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) x[rt][ks] = xr[rt][ks];
            if (NET == 4) load_xm_traj<NS>(g.obs, g.n, g.bm, tnext, lane, xr);
            else load_xm_from<NS>(xsrc, g.npad, tnext, lane, xr);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16 z;
#pragma unroll
                for (int q = 0; q < 16; ++q) z[q] = b1q[q];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) z = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[ks], x[rt][ks], z, 0, 0, 0);
                uint16_t* dst = l_H + (32 * rt + r) * PW + u0 + 4 * kb;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    uint2 o;
                    o.x = pack2_bf16(act_fwd_t<ACT>(z[4 * g4 + 0]), act_fwd_t<ACT>(z[4 * g4 + 1]));
                    o.y = pack2_bf16(act_fwd_t<ACT>(z[4 * g4 + 2]), act_fwd_t<ACT>(z[4 * g4 + 3]));
                    *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
                }
            }
        }
        __syncthreads();  // A: the H1 tile is complete
        W3_STAMP(0, 2);
        // ---- layer 2 on the MFMA: this wave's 32 columns for the tile's 64 rows; bias + activation ----
        f32x16 h2[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) h2[rt][q] = 0.0f;
        {
            const uint16_t* ap = l_H + r * PW + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 32 * rt * PW + 16 * ks);
                    // FWD_T: the operands swapped -- the same products summed over k in the same order, D = H2^T (lane = sample row, registers
                    // = this wave's columns 8 (q >> 2) + 4 kb + (q & 3)): the register image of the B operand IS the A operand's and vice versa
                    if (FWD_T) h2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ks], a, h2[rt], 0, 0, 0);
                    else h2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], h2[rt], 0, 0, 0);
                }
        }
        if (FWD_T) {
            // ---- forward-only modes (DQN target network, the rollout's batched value pass): with lane = sample the head is IN-LANE -- 16 FMAs per
            //      output and row tile against this lane's 16 columns of W3, one add across the two k halves -- instead of the wave's 64 x 32
            //      block of H2 going through its private LDS block (the largest phase of the tile: profiles/r06_ppo3w.md section 4) ----
            float pa[2][NOUT];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) pa[rt][o] = 0.0f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c0 = 32 * w + 8 * g4 + 4 * kb;  // this lane's columns c0 .. c0 + 3 of register group g4
                const float4 bb = *reinterpret_cast<const float4*>(m.b2 + c0);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                float wv[4 * NOUT];
#pragma unroll
                for (int i = 0; i < 4 * NOUT; ++i) wv[i] = m.W3[NOUT * c0 + i];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const float hv = act_fwd_t<ACT>(h2[rt][4 * g4 + i] + bv[i]);
#pragma unroll
                        for (int o = 0; o < NOUT; ++o) pa[rt][o] = fmaf(wv[NOUT * i + o], hv, pa[rt][o]);
                    }
            }
            W3_STAMP(0, 3);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const uint32_t u = __float_as_uint(pa[rt][o]);
                    const auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // t[0] + t[1]: the kb = 0 half first, in every lane
                    const float sum = __uint_as_float(t[0]) + __uint_as_float(t[1]);
                    if (kb == 0) l_part[(w * MAXO + o) * RW + 32 * rt + r] = sum;
                }
        } else {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) h2[rt][q] = act_fwd_t<ACT>(h2[rt][q] + b2v);
        W3_STAMP(0, 3);
        // ---- head: this wave's share of sum_j W3[o, j] h2[j] per row.  The wave's 64 x 32 block of H2 goes through its
        //      private LDS block (D layout in, one row per lane out): 32 FMAs per output and lane, no cross-lane sums
        //      (the DPP row reductions this replaces were 35 % of the tile) ----
        wave_lds_fence();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) l_tw[(32 * rt + mfma_row(q, kb)) * TPW + r] = h2[rt][q];
        wave_lds_fence();
        {
            float pa[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) pa[o] = 0.0f;
            const float* hrow = l_tw + lane * TPW;
            const float* w3p = m.W3 + NOUT * 32 * w;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(hrow + 4 * c4);
                const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) pa[o] = fmaf(w3p[o + NOUT * (4 * c4 + e)], hv[e], pa[o]);
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) l_part[(w * MAXO + o) * RW + lane] = pa[o];
        }
        }
        wave_lds_fence();
        __syncthreads();  // C: every wave's partial sums
        W3_STAMP(0, 4);
        // ---- the loss line of each sample and dL/d(head outputs): wave 0, one row per lane ----
        if (tid < RW) {
            const int s = tid;
            const bool valid = ((uint32_t)tile * RW + (uint32_t)s) < g.bm;
            float oa[MAXO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float acc = l_part[o * RW + s];
#pragma unroll
                for (int ww = 1; ww < WV; ++ww) acc += l_part[(ww * MAXO + o) * RW + s];
                oa[o] = acc + m.b3[o];
            }
            if (NET == 0) {
                float dl[MAXO] = {0.f, 0.f, 0.f, 0.f};
                const float lp_old = fmaxf(sr0, g.min_logp);  // clamp!(log_p, log(1e-8), Inf)
                const float A = sr1;
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
                    const int a = __float_as_int(sr2);
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
                    const float mu = oa[0], lsg = oa[1];
                    const float sg = expf(lsg);
                    const float z = sr2;
                    const float se = sg + eps;
                    const float zz = (z - mu) / se;
                    const float lp_new = -(zz * zz + LOG2PI_F) / 2.0f - logf(se);
                    ent = ((LOG2PI_F + 1.0f) + lsg) / 2.0f;
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
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    l_dq[o * RW + s] = dl[o];
                    s_red[o] += dl[o];
                }
                s_red[NOUT] += surr_min;
                s_red[NOUT + 1] += ent;
            } else if (NET == 1) {
                const float dv = sr0 - oa[0];
                float dvout = -2.0f * g.wc * g.inv_b * dv;
                float sq = dv * dv;
                if (!valid) {
                    dvout = 0.f;
                    sq = 0.f;
                }
                l_dq[s] = dvout;
                s_red[0] += dvout;
                s_red[NOUT] += sq;
            } else if (NET == 4) {  // V(s) of sample q = t n + env: the value trace is (T + 1, n) contiguous
                if (valid) g.sg[(int64_t)tile * RW + s] = oa[0];
            } else if (NET == 2) {  // y = r + gamma (1 - terminal) max_a' Qt(s', a')
                float mx = oa[0];
#pragma unroll
                for (int k = 1; k < NOUT; ++k) mx = fmaxf(mx, oa[k]);
                const float cont = sr1 != 0.0f ? 0.f : 1.f;
                g.sg[(int64_t)tile * RW + s] = sr0 + g.gamma * cont * mx;
            } else {  // Huber(delta) on Q(s, a) - y, mean over the batch
                const int a = __float_as_int(sr2);
                float qa = 0.f;
#pragma unroll
                for (int k = 0; k < NOUT; ++k)
                    if (k == a) qa = oa[k];
                const float d = qa - sr0;
                const float e = fabsf(d);
                float l = (e < g.delta) ? (e * e) * 0.5f : g.delta * (e - 0.5f * g.delta);
                float gi = (e < g.delta) ? d : (d > 0.f ? g.delta : (d < 0.f ? -g.delta : 0.f));
                gi *= g.inv_b;
                if (valid && g.isw) {
                    const float wis = g.isw[(int64_t)tile * RW + s];
                    gi *= wis;
                    l *= wis;
                }
                if (!valid) {
                    gi = 0.f;
                    l = 0.f;
                }
                if (valid && g.td_out) g.td_out[(int64_t)tile * RW + s] = e;
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const float dl = (o == a) ? gi : 0.f;
                    l_dq[o * RW + s] = dl;
                    s_red[o] += dl;
                }
                s_red[NOUT] += l;
            }
        }
        load_s(tnext);  // the next tile's loss inputs (every wave: uniform streams; wave 0 uses them)
        if (NET == 2 || NET == 4) {
            W3_STAMP(0, 5);
            continue;
        }  // forward only: the other waves are already in the next pass's layer 1
        __syncthreads();  // D: dL/d(head outputs) of the tile
        W3_STAMP(0, 5);
        // ---- head backward in the MFMA D layout: dW3, dh2 -> dz2 (f32) -> db2; dz2 -> bf16: fragments straight to global,
        //      rows through the wave's private block (64 rows x 64 B of this wave's columns, no workgroup barrier) ----
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * rt + mfma_row(q, kb);
                float dqv[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) dqv[o] = l_dq[o * RW + row];
                const float hv = h2[rt][q];
                float dh = 0.0f;
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    a_dw3[o] = fmaf(dqv[o], hv, a_dw3[o]);
                    dh = fmaf(dqv[o], w3[o], dh);
                }
                const float dz = dh * act_bwd_t<ACT>(hv, hv);  // relu: h2 > 0 <=> z2 > 0
                a_db2 += dz;
                h2[rt][q] = dz;  // the register is free: keep dz for the packed stores below
#if RLHIP_W3_DZ_ONCE != 2
                l_zw[row * ZPW + r] = f32_to_bf16_rne(dz);
#endif
            }
            // fragment order: samples 32 rt + 8 gq + 4 kb + {0..3} of column `col` = bytes 8 kb .. 8 kb + 7 of slot
            // (k-step 2 rt + (gq >> 1), column tile w, lane 32 (gq & 1) + r)
#if RLHIP_W3_DZ_ONCE != 1
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 v2;
                v2.x = pack2_bf16(h2[rt][4 * gq + 0], h2[rt][4 * gq + 1]);
                v2.y = pack2_bf16(h2[rt][4 * gq + 2], h2[rt][4 * gq + 3]);
                const int64_t slot = (((int64_t)tile * (RW / 16) + 2 * rt + (gq >> 1)) * WV + w) * 64 + 32 * (gq & 1) + r;
                *reinterpret_cast<uint2*>(g.dz_frag + (NET == 1 ? g.frag_stride : 0) + slot * 8 + 4 * kb) = v2;
            }
#endif
        }
        W3_STAMP(0, 6);
#if RLHIP_W3_DZ_ONCE != 2
        wave_lds_fence();
        {
            uint16_t* dst = g.dz_rows + (int64_t)tile * RW * HW + 32 * w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + 64 * i, row = c >> 2, cc = c & 3;
                *reinterpret_cast<uint4*>(dst + row * HW + 8 * cc) = *reinterpret_cast<const uint4*>(l_zw + row * ZPW + 8 * cc);
            }
        }
        wave_lds_fence();
#endif
        W3_STAMP(0, 7);
        // no barrier: the next pass writes l_H (last read before barrier C) and this wave's private block in program order
    }
#undef W3_STRIDE
    W3_MARK(0, 10, NET == RLHIP_W3_TIMING_NET);
    if (NET == 2 || NET == 4) return;
    // ---- this workgroup's partial row: b2, W3, b3 and the loss sums ----
    const int sb2 = HW * NS + HW, sW3 = sb2 + HW, sb3 = sW3 + NOUT * HW;
    float* rowS = g.partS + (int64_t)blockIdx.x * g.npS + (NET == 1 ? g.nS_a : 0);
    a_db2 += __shfl_xor(a_db2, 32, 64);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a_dw3[o] += __shfl_xor(a_dw3[o], 32, 64);
    if (kb == 0) {
        rowS[sb2 + col] = a_db2;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) rowS[sW3 + o + NOUT * col] = a_dw3[o];
    }
    if (w == 0) {
#pragma unroll
        for (int o = 0; o < NOUT + 2; ++o) s_red[o] = wave_sum_f32(s_red[o]);
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) rowS[sb3 + o] = s_red[o];
            float* lp = g.loss_partials + (int64_t)blockIdx.x * 4;
            if (NET == 0) {
                lp[0] = s_red[NOUT];
                lp[2] = s_red[NOUT + 1];
            } else if (NET == 1) {
                lp[1] = s_red[NOUT];
            } else {
                lp[0] = s_red[NOUT];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dH1 -> dW1 / db1
// PAD (template parameter of the backward kernel, chosen per launch: w3_dzf_pad()): the padded LDS copy of the fragment tile, see RLHIP_W3_DZF_PAD
template <bool PAD>
struct Dzf {
#if RLHIP_W3_DZ_ONCE == 2
    static constexpr int G = PAD ? 72 : 64, H = PAD ? 36 : 32;  // 16-byte slots per 64-slot group / per 32-slot half of the LDS copy
    static constexpr int TILE = (RW / 16) * WV * G * 8;         // elements of one copy (32 KB; 36 KB padded)
#else
    static constexpr int TILE = RW * PW;
#endif
    static constexpr size_t LDS = (2 * WV * 4 * RW + HW * 4 + HW) * sizeof(float) + (size_t)2 * TILE * sizeof(uint16_t);
};

// this thread's four 16-byte chunks of a 64 x 256 bf16 tile: chunk c = tid + 512 i -> row c >> 5, column 8 (c & 31)
__device__ __forceinline__ void load_dz_tile(const uint16_t* __restrict__ dz_rows, int tile, int tid, nt_u32x4 (&d)[4]) {
    const uint16_t* src = dz_rows + (int64_t)tile * RW * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + NTW * i, row = c >> 5, cc = c & 31;
        d[i] = *reinterpret_cast<const nt_u32x4*>(src + row * HW + 8 * cc);
    }
}
__device__ __forceinline__ void store_dz_tile(uint16_t* lH, int tid, const nt_u32x4 (&d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + NTW * i, row = c >> 5, cc = c & 31;
        *reinterpret_cast<nt_u32x4*>(lH + row * PW + 8 * cc) = d[i];
    }
}

#if RLHIP_W3_DZ_ONCE == 2
// RLHIP_W3_DZ_ONCE == 2: the ONE image of dZ2 is the fragment image [tile][k-step s >> 4][column tile j >> 5][slot 32 ((s >> 3) & 1) +
// (j & 31)][u = s & 7] (what the forward kernel's D registers store without any transposition, and what the dW2 kernel loads as its B
// operand).  This kernel needs the transposed view -- A operand of dH1 = dZ2 W2: lane = sample row, 8 consecutive columns j -- and takes
// it from a lane-linear LDS copy of the tile with two transposing reads per fragment: in 16-lane group G (rows 16 (G & 1) + 0 .. 15 of
// the 32-row half rt, k half kb = G >> 1) lane 4 i + q addresses samples 4 q .. 4 q + 3 of column 16 ks + 8 kb + 4 h + i and lane g
// receives its own sample's four columns (tools/micro/tr16_probe.hip).
__device__ __forceinline__ void load_dzf_tile(const uint16_t* __restrict__ dz_frag, int tile, int tid, nt_u32x4 (&d)[4]) {
    const uint16_t* src = dz_frag + (int64_t)tile * RW * HW;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const nt_u32x4*>(src + 8 * (tid + NTW * i));
}
// LDS copy of the tile: the 16-byte slots in image order.  With RLHIP_W3_DZF_PAD, FOUR slots of padding behind every 32 (slot c -> 72 (c >> 6)
// + 36 ((c >> 5) & 1) + (c & 31)): the 16 segments a 16-lane group addresses in one transposing read -- columns i = 0 .. 3 (16 bytes apart),
// sample quads q = 0 .. 3 (8 bytes apart for q & 1, the other 32-slot half for q >> 1) -- then fall on 16 different bank pairs (without the
// padding the two halves are 512 bytes apart: the same banks, a two-way conflict)
template <bool PAD>
__device__ __forceinline__ void store_dzf_tile(uint16_t* lF, int tid, const nt_u32x4 (&d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + NTW * i;
        *reinterpret_cast<nt_u32x4*>(lF + 8 * (Dzf<PAD>::G * (c >> 6) + Dzf<PAD>::H * ((c >> 5) & 1) + (c & 31))) = d[i];
    }
}
template <bool PAD>
__device__ __forceinline__ int dzf_lane_base(int lane) {  // element offset of this lane's segment for (rt, ks, h) = (0, 0, 0)
    const int G = lane >> 4, g = lane & 15, i = g >> 2, q = g & 3;
    return (((G & 1) * WV * Dzf<PAD>::G) + Dzf<PAD>::H * (q >> 1) + 8 * (G >> 1) + i) * 8 + 4 * (q & 1);
}
template <bool PAD>
__device__ __forceinline__ bf16x8 dzf_a_frag(const uint16_t* lF, int base, int rt, int ks) {
    typedef __attribute__((address_space(3))) tr_v4s* lds_v4s_ptr;
    const uint16_t* src = lF + base + (2 * rt * WV * Dzf<PAD>::G + (ks >> 1) * Dzf<PAD>::G + 16 * (ks & 1)) * 8;
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(src));
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(src + 4 * 8));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    const nt_u32x4 u = {l2.x, l2.y, h2.x, h2.y};
    return __builtin_bit_cast(bf16x8, u);
}
#define W3_LOAD_DZ(tile_, d_) load_dzf_tile(g.dz_frag + (net ? g.frag_stride : 0), tile_, tid, d_)
#define W3_STORE_DZ(l_, d_) store_dzf_tile<PAD>(l_, tid, d_)
#else
#define W3_LOAD_DZ(tile_, d_) load_dz_tile(g.dz_rows, tile_, tid, d_)
#define W3_STORE_DZ(l_, d_) store_dz_tile(l_, tid, d_)
#endif

template <int NS, int ACT, bool PAD>
__global__ __launch_bounds__(NTW, 2) void ppo3w_bwd_kernel(P3WArgs g, int net) {
    constexpr int BWD_TILE_ELEMS = Dzf<PAD>::TILE;
    extern __shared__ __attribute__((aligned(16))) char smw[];
    float* l_x = reinterpret_cast<float*>(smw);  // [2][WV][4][RW]: every wave keeps its own copy of the tile's observations
    float* l_w = l_x + 2 * WV * 4 * RW;          // W1 | b1
    uint16_t* l_H = reinterpret_cast<uint16_t*>(l_w + HW * 4 + HW);  // [2][RW][PW] dZ2 rows, double-buffered
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int k = 32 * w + r;  // this lane's hidden unit of layer 1
    W3_MARK(1, 8, net == 0);
    const float* pnet = g.params + (net ? g.np_a : 0);
    bf16x8 bw[KSW];
    load_frags_w(g.packed + (net ? 2 * HW * HW : 0) + HW * HW, w, lane, bw);
    const int stride = gridDim.x, last = g.ntiles - 1;
    float* l_xw = l_x + w * 4 * RW;
    // Two register sets keep the tiles of the next two passes in flight (a tile's loads take ~4000 cycles, a pass ~3000):
    // pass `it` lands set it & 1 (tile + 1) in the other LDS buffer and re-issues it with tile + 3.  (Three sets / three
    // passes in flight were measured too: 29.8 vs 28.9 us -- with two sets the kernel is no longer waiting for memory; per
    // pass and SIMD it issues 2 x 354 VALU instructions (4 cycles each) beside 2048 cycles of MFMA.)
    // (tile indices are clamped to the last tile instead of branching: unconditional loads keep the vmcnt bookkeeping exact)
    nt_u32x4 dzs[2][4];
    float xr[2][NS];
    {
        const int t0 = blockIdx.x;
        nt_u32x4 d0[4];
        float x0[NS];
        // issue order = the steady state of the pass loop (older: everything the prologue itself consumes; then set 0, then
        // set 1), so that the loop is entered with exactly the outstanding loads its back edge carries
        W3_LOAD_DZ(t0, d0);
        load_x<NS>(g, t0, lane, x0);
        constexpr int NWL = (HW * NS + HW + NTW - 1) / NTW;
        float wv[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wv[i] = (tid + NTW * i < HW * NS + HW) ? pnet[tid + NTW * i] : 0.0f;
        W3_LOAD_DZ(min(t0 + stride, last), dzs[0]);
        load_x<NS>(g, min(t0 + stride, last), lane, xr[0]);
        W3_LOAD_DZ(min(t0 + 2 * stride, last), dzs[1]);
        load_x<NS>(g, min(t0 + 2 * stride, last), lane, xr[1]);
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + NTW * i < HW * NS + HW) l_w[tid + NTW * i] = wv[i];
        W3_STORE_DZ(l_H, d0);
        store_x<NS>(l_xw, lane, x0);
    }
    __syncthreads();
    W3_MARK(1, 9, net == 0);
    float w1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) w1[i] = l_w[k + HW * i];
    const float b1v = l_w[HW * NS + k];
    float a_db1 = 0.0f, a_dw1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) a_dw1[i] = 0.0f;

#define W3_STRIDE (net == 0 ? 2 * gridDim.x : 0x7fffffff)
    int tile = blockIdx.x;
    auto pass = [&](auto PC) __attribute__((always_inline)) {
        constexpr int p = decltype(PC)::value;
        W3_STAMP(1, 0);
        const float* lx = l_xw + p * WV * 4 * RW;
        const uint16_t* lH = l_H + p * BWD_TILE_ELEMS;
        {
            int t3 = tile + 3 * stride;  // past the end: re-read this workgroup's OWN first tile (never one tile for all)
            if (t3 > last) t3 = blockIdx.x;
            W3_STORE_DZ(l_H + (p ^ 1) * BWD_TILE_ELEMS, dzs[p]);
            store_x<NS>(l_xw + (p ^ 1) * WV * 4 * RW, lane, xr[p]);
            W3_LOAD_DZ(t3, dzs[p]);
            load_x<NS>(g, t3, lane, xr[p]);
        }
        W3_STAMP(1, 1);
        f32x16 dh[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int q = 0; q < 16; ++q) dh[rt][q] = 0.0f;
        {
#if RLHIP_W3_DZ_ONCE == 2
            const int abase = dzf_lane_base<PAD>(lane);
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                if ((ks & 3) == 0 && ks) __builtin_amdgcn_sched_barrier(0);  // (as in the dW2 kernel: bounds how far the LDS reads are hoisted)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const bf16x8 a = dzf_a_frag<PAD>(lH, abase, rt, ks);
                    dh[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], dh[rt], 0, 0, 0);
                }
            }
#else
            const uint16_t* ap = lH + r * PW + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 32 * rt * PW + 16 * ks);
                    dh[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], dh[rt], 0, 0, 0);
                }
#endif
        }
        W3_STAMP(1, 2);
        // z1 = b1 + W1 x of the tile in the layout of dH1 (lane = hidden unit, registers = sample rows) on the f32 MFMA:
        // v_mfma_f32_32x32x2_f32 with the bias as the accumulator's initial value IS the oracle's fmaf chain, bit for bit
        // (tools/micro/mfma_f32_l1.hip: 0 of 1.2 M values differ), and takes the 3 - 4 FMAs per element off the VALU, which
        // is what this kernel is bound by (2 x 354 VALU instructions per pass beside 2048 cycles of MFMA)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 z1;
#pragma unroll
            for (int q = 0; q < 16; ++q) z1[q] = b1v;
#pragma unroll
            for (int ks = 0; ks < (NS + 1) / 2; ++ks) {
                const int kk = 2 * ks + kb;  // A[m = sample row][k = kk], B[k = kk][n = this lane's unit]
                const float av = lx[(kk < NS ? kk : NS - 1) * RW + 32 * rt + r];
                const float a = kk < NS ? av : 0.0f;
                float b = 0.0f;
#pragma unroll
                for (int i = 0; i < NS; ++i) b = (i == kk) ? w1[i] : b;
                z1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * rt + mfma_row(q, kb);
                float x[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) x[i] = lx[i * RW + row];
                const float z = z1[q];
                // tanh: 1 - h1^2 = 4 t / (t + 1)^2 with t = exp(2 |z|), branch-free on the hardware exponential.  The library
                // tanhf (96 inlined copies of a branchy routine in this kernel) left 446 registers per lane in scratch and
                // the launch at 266 us against 29 us for relu; the factor agrees with 1 - tanhf(z)^2 to ~1e-6 relative
                // (the bar on this gradient is 5e-4 of max |g|); the FORWARD passes keep tanhf (bf16 rounding parity)
                float dact;
                if (ACT == 0) {
                    dact = z > 0.0f ? 1.0f : 0.0f;
                } else {
                    const float t = __expf(2.0f * fminf(fabsf(z), 40.0f));
                    const float u = __builtin_amdgcn_rcpf(t + 1.0f);  // v_rcp_f32: 1 ulp
                    dact = 4.0f * t * u * u;
                }
                const float dz = dh[rt][q] * dact;
                a_db1 += dz;
#pragma unroll
                for (int i = 0; i < NS; ++i) a_dw1[i] = fmaf(dz, x[i], a_dw1[i]);
                // keep the unrolled rows in program order (otherwise every LDS read of the 32 rows is hoisted to the top
                // and the live x values spill)
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        W3_STAMP(1, 3);
        __syncthreads();  // the one barrier of a pass: buffers p are free, buffers p ^ 1 are complete
        W3_STAMP(1, 4);
    };
    // pairs of passes without an exit in between (a straight-line loop body lets the compiler count the outstanding loads
    // exactly: with an exit between the two passes it waited for the YOUNGER register set in every second pass)
    const int npass = (g.ntiles - (int)blockIdx.x + stride - 1) / stride;
    for (int it2 = 0; it2 + 1 < npass; it2 += 2) {
        pass(std::integral_constant<int, 0>{});
        tile += stride;
        pass(std::integral_constant<int, 1>{});
        tile += stride;
    }
    if (npass & 1) pass(std::integral_constant<int, 0>{});
#undef W3_STRIDE
    W3_MARK(1, 10, net == 0);
    const int nS_net = net ? g.nS_a : 0;
    float* rowS = g.partS + (int64_t)blockIdx.x * g.npS + nS_net;
    a_db1 += __shfl_xor(a_db1, 32, 64);
#pragma unroll
    for (int i = 0; i < NS; ++i) a_dw1[i] += __shfl_xor(a_dw1[i], 32, 64);
    if (kb == 0) {
        rowS[HW * NS + k] = a_db1;
#pragma unroll
        for (int i = 0; i < NS; ++i) rowS[k + HW * i] = a_dw1[i];
    }
}

// ------------------------------------------------------------------------------------------------ dW2 = H1^T dZ2
constexpr size_t DW2W_LDS = (2 * WV * 4 * RW + HW * 4 + HW) * sizeof(float) + (size_t)2 * (HW / 2) * PT * sizeof(uint16_t)
#if RLHIP_W3_DZ_ONCE == 1
                            + (size_t)2 * RW * PW * sizeof(uint16_t)  // the tile's dZ2 ROWS, double-buffered (as ppo3w_bwd_kernel stages them)
#endif
    ;

__device__ __forceinline__ void load_dz_frags(const uint16_t* __restrict__ dz_frag, int tile, int w, int lane,
                                              bf16x8 (&b)[RW / 16]) {
#pragma unroll
    for (int ks = 0; ks < RW / 16; ++ks)
        b[ks] = *reinterpret_cast<const bf16x8*>(dz_frag + ((((int64_t)tile * (RW / 16) + ks) * WV + w) * 64 + lane) * 8);
}

// RLHIP_W3_DZ_ONCE: the same fragments out of the ROW image dz[tile * RW + sample][HW].  The tile's rows travel global -> registers
// (16-byte coalesced loads two passes ahead: load_dz_tile) -> LDS [RW][PW] in front of the pass's one barrier (store_dz_tile), and a
// lane gathers the fragment of k-step ks -- samples 16 ks + 8 kb + 0 .. 7 of column 32 w + r -- out of it.
// (First form tried: eight 2-byte reads per fragment straight from global memory, no LDS: ppo3w_dw2_kernel 43.8 -> 57.4 us; tools/contacts_r06/r6_m.sh.)
// Two transposing LDS reads per fragment (ds_read_b64_tr_b16; semantics pinned by tools/micro/tr16_probe.hip: within a 16-lane group,
// lane g passes the address of M[R0 + (g >> 2)][C0 + 4 (g & 3)] and receives M[R0 .. R0 + 3][C0 + g]): group G = lane >> 4 covers columns
// 32 w + 16 (G & 1) + 0 .. 15 of the k half kb = G >> 1, rows 16 ks + 8 kb + {0 .. 3 | 4 .. 7}.
// (Second form tried: eight 2-byte LDS reads + four packs per fragment: 42.1 -> 53.0 us.)
__device__ __forceinline__ bf16x8 gather_dz_col(const uint16_t* lD, int ks, int w, int lane) {
    const int G = lane >> 4, g = lane & 15;
    const uint16_t* src = lD + (16 * ks + 8 * (G >> 1) + (g >> 2)) * PW + 32 * w + 16 * (G & 1) + 4 * (g & 3);
    typedef __attribute__((address_space(3))) tr_v4s* lds_v4s_ptr;
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(src));
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(src + 4 * PW));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);  // (whole registers: no 16-bit shuffling)
    const nt_u32x4 u = {l2.x, l2.y, h2.x, h2.y};
    return __builtin_bit_cast(bf16x8, u);
}

template <int NS, int ACT>
// net < 0: one launch for BOTH nets of a PPO pair, net = blockIdx.y (half as many sample ranges per net, hence half as many
// partial rows for the reduction: 64 instead of 128 MB per optimiser step, and twice as many passes per prologue)
__global__ __launch_bounds__(NTW, 2) void ppo3w_dw2_kernel(P3WArgs g, int net_arg, int nsr) {
    const int net = net_arg < 0 ? (int)blockIdx.y : net_arg;
    const uint16_t* const dzf = g.dz_frag + (net ? g.frag_stride : 0);
    extern __shared__ __attribute__((aligned(16))) char smw[];
    float* l_x = reinterpret_cast<float*>(smw);  // [2][WV][4][RW]: wave-private copies of the tile's observations
    float* l_w = l_x + 2 * WV * 4 * RW;          // W1 | b1
    uint16_t* l_T = reinterpret_cast<uint16_t*>(l_w + HW * 4 + HW);  // [2][HW / 2][PT]: H1^T of this k half
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    // The two k halves of a sample range read the SAME dZ2 fragments.  Workgroups are dealt to the 8 XCDs round-robin
    // (workgroup b -> XCD b % 8, each XCD with its own L2), so the pair is placed on ONE XCD, back to back in its dispatch
    // order: the second read of every fragment can hit that XCD's L2 instead of crossing the fabric twice (measured: within
    // noise, 28.0 -> 27.5 us -- the kernel is bound by its instruction streams, not by the fabric).
    int kh, sr;
    if ((nsr & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        kh = j & 1;
        sr = (j >> 1) * 8 + xcd;
    } else {
        kh = blockIdx.x & 1;
        sr = blockIdx.x >> 1;
    }
    W3_MARK(2, 8, net == 0);
    const float* pnet = g.params + (net ? g.np_a : 0);
    float* l_xw = l_x + w * 4 * RW;
    f32x16 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[kt][q] = 0.0f;
    // this wave's column tile of dZ2 for a tile's four k-steps, two sets: set it & 1 feeds pass `it` and is re-issued with
    // the tile of pass it + 2 right after its MFMAs (two passes in flight); the gather alternates between waves 7 and 6
    // observations: set it & 1 holds tile + 1 at the top of pass `it`, lands in the other LDS copy, is re-issued with tile + 3
    // (indices clamped to the last tile instead of branching: unconditional loads keep the vmcnt bookkeeping exact)
#if RLHIP_W3_DZ_ONCE == 1
    nt_u32x4 dzs[2][4];  // this thread's four 16-byte chunks of a tile's rows, two tiles in flight
    uint16_t* l_dz = l_T + 2 * (HW / 2) * PT;  // [2][RW][PW]
#else
    bf16x8 bq[2][RW / 16];
#endif
    float xr[2][NS];
    const int last = g.ntiles - 1;
    {
        float x0[NS];
        // issue order = the steady state of the pass loop: what the prologue consumes first, then x set 0, fragment set 0
        // (tile 0; re-issued at the END of a pass), x set 1, fragment set 1
        load_x<NS>(g, min(sr, last), lane, x0);
        constexpr int NWL = (HW * NS + HW + NTW - 1) / NTW;
        float wv[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) wv[i] = (tid + NTW * i < HW * NS + HW) ? pnet[tid + NTW * i] : 0.0f;
        load_x<NS>(g, min(sr + nsr, last), lane, xr[0]);
#if RLHIP_W3_DZ_ONCE == 1
        load_dz_tile(dzf, min(sr, last), tid, dzs[0]);
        load_x<NS>(g, min(sr + 2 * nsr, last), lane, xr[1]);
        load_dz_tile(dzf, min(sr + nsr, last), tid, dzs[1]);
#else
        load_dz_frags(dzf, min(sr, last), w, lane, bq[0]);
        load_x<NS>(g, min(sr + 2 * nsr, last), lane, xr[1]);
        load_dz_frags(dzf, min(sr + nsr, last), w, lane, bq[1]);
#endif
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + NTW * i < HW * NS + HW) l_w[tid + NTW * i] = wv[i];
        store_x<NS>(l_xw, lane, x0);
    }
    __syncthreads();
    W3_MARK(2, 9, net == 0);

    // layer 1 of a pass: wave w owns units 32 (w & 3) .. + 31 of this k half for samples 32 (w >> 2) .. + 31.  B operand = this
    // lane's unit's W1 row (component kb + 2 ks; zero beyond NS), C = its bias: loop-invariant registers
    const int kt1 = w & 3, rt1 = w >> 2;
    float w1k[(NS + 1) / 2];
    const int k1 = (HW / 2) * kh + 32 * kt1 + r;
#pragma unroll
    for (int ks = 0; ks < (NS + 1) / 2; ++ks) {
        const float v = l_w[k1 + HW * min(2 * ks + kb, NS - 1)];
        w1k[ks] = 2 * ks + kb < NS ? v : 0.0f;
    }
    const float b1k = l_w[HW * NS + k1];

#define W3_STRIDE (net == 0 ? 2 * nsr : 0x7fffffff)
    int tile = sr;
    auto pass = [&](auto PC) __attribute__((always_inline)) {
        constexpr int p = decltype(PC)::value;
        W3_STAMP(2, 0);
        const float* lx = l_xw + p * WV * 4 * RW;
        uint16_t* lT = l_T + p * (HW / 2) * PT;
        store_x<NS>(l_xw + (p ^ 1) * WV * 4 * RW, lane, xr[p]);
        load_x<NS>(g, min(tile + 3 * nsr, last), lane, xr[p]);
#if RLHIP_W3_DZ_ONCE == 1
        uint16_t* lD = l_dz + p * RW * PW;
        store_dz_tile(lD, tid, dzs[p]);  // this pass's rows (requested two passes ago); read behind the pass's barrier
        load_dz_tile(dzf, min(tile + 2 * nsr, last), tid, dzs[p]);
#endif
        wave_lds_fence();
        W3_STAMP(2, 1);
        // ---- layer 1 in [k][sample] order for the 128 hidden units of this half, on the f32 MFMA (bit-identical to the
        //      fmaf chain, see ppo3w_bwd_kernel): one 32-sample x 32-unit block per wave, D[sample][unit] with lane = unit,
        //      so a lane's four consecutive sample rows of a register group are 8 contiguous bytes of H1^T ----
        {
            f32x16 z;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = b1k;
#pragma unroll
            for (int ks = 0; ks < (NS + 1) / 2; ++ks) {
                const float av = lx[min(2 * ks + kb, NS - 1) * RW + 32 * rt1 + r];
                z = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w1k[ks], z, 0, 0, 0);
            }
            uint16_t* dst = lT + (32 * kt1 + r) * PT + 32 * rt1 + 4 * kb;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 o;
                o.x = pack2_bf16(act_fwd_t<ACT>(z[4 * g4 + 0]), act_fwd_t<ACT>(z[4 * g4 + 1]));
                o.y = pack2_bf16(act_fwd_t<ACT>(z[4 * g4 + 2]), act_fwd_t<ACT>(z[4 * g4 + 3]));
                *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
            }
        }
        __syncthreads();  // the one barrier of a pass
        W3_STAMP(2, 2);
#if RLHIP_W3_DZ_ONCE == 1
        bf16x8 bnx = gather_dz_col(lD, 0, w, lane);
#endif
#pragma unroll
        for (int ks = 0; ks < RW / 16; ++ks) {
#if RLHIP_W3_DZ_ONCE == 1
            // the fragment of k-step ks + 1 is requested in front of the MFMAs of k-step ks; the scheduling barrier keeps the scheduler from
            // hoisting all 24 LDS reads of a pass to its top (256 registers + spills without it)
            const bf16x8 bfr = bnx;
            if (ks + 1 < RW / 16) bnx = gather_dz_col(lD, ks + 1, w, lane);
            if (ks == 2) __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(lT + (32 * kt + r) * PT + 16 * ks + 8 * kb);
#if RLHIP_W3_DZ_ONCE == 1
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr, acc[kt], 0, 0, 0);
#else
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[p][ks], acc[kt], 0, 0, 0);
#endif
            }
        }
        W3_STAMP(2, 3);
#if RLHIP_W3_DZ_ONCE != 1
        load_dz_frags(dzf, min(tile + 2 * nsr, last), w, lane, bq[p]);
#endif
        W3_STAMP(2, 4);
    };
    const int npass = (g.ntiles - sr + nsr - 1) / nsr;
    for (int it2 = 0; it2 + 1 < npass; it2 += 2) {
        pass(std::integral_constant<int, 0>{});
        tile += nsr;
        pass(std::integral_constant<int, 1>{});
        tile += nsr;
    }
    if (npass & 1) pass(std::integral_constant<int, 0>{});
#undef W3_STRIDE
    W3_MARK(2, 10, net == 0);
    // D[row = k (local)][col = j]: dW2[j + HW k]
    float* out = g.partW + ((int64_t)sr * g.wnets + net) * HW * HW;
    const int j = 32 * w + r;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) store_wt(&out[j + HW * ((HW / 2) * kh + 32 * kt + mfma_row(q, kb))], acc[kt][q]);
    W3_MARK(2, 11, net == 0);
}

// ------------------------------------------------------------------------------------------------ partial rows -> gradient
__global__ __launch_bounds__(256) void ppo3w_reduce_kernel(const float* __restrict__ partS, const float* __restrict__ partW,
                                                           const float* __restrict__ loss_partials, int nrowsS, int nrowsW,
                                                           int npS, int nS_a, int np, int np_a, int ns,
                                                           float* __restrict__ grad, float* __restrict__ losses, float wa,
                                                           float wc, float we, float inv_b, int wnets) {
    __shared__ float l_g[4][64];
    __shared__ float l_loss[4];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (p < np) {
        const int net = p >= np_a ? 1 : 0;
        const int q = p - net * np_a;
        const int nA = HW * ns + HW;
        const float* src;
        int64_t stride;
        int nrows;
        if (q < nA) {
            src = partS + (net ? nS_a : 0) + q;
            stride = npS;
            nrows = nrowsS;
        } else if (q < nA + HW * HW) {
            src = partW + (int64_t)net * HW * HW + (q - nA);
            stride = (int64_t)wnets * HW * HW;
            nrows = nrowsW;
        } else {
            src = partS + (net ? nS_a : 0) + (q - HW * HW);
            stride = npS;
            nrows = nrowsS;
        }
        const int per = (nrows + 3) / 4;
        const int b0 = grp * per, b1 = min(nrows, b0 + per);
#pragma unroll 8
        for (int b = b0; b < b1; ++b) acc += src[(int64_t)b * stride];
    }
    l_g[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && p < np) grad[p] = ((l_g[0][lane] + l_g[1][lane]) + l_g[2][lane]) + l_g[3][lane];
    if (blockIdx.x == 0 && losses != nullptr) {
        if (grp < 3) {
            float a = 0.f;
            for (int b = lane; b < nrowsS; b += 64) a += loss_partials[(int64_t)b * 4 + grp];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) l_loss[grp] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0 && wa < 0.0f) {  // DQN (marked by a negative actor weight): the mean Huber loss, one number
            losses[0] = l_loss[0] * inv_b;
        } else if (threadIdx.x == 0) {
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

// ------------------------------------------------------------------------------------------------ optimiser tail
// optimise! after the six tile kernels, in TWO launches with no grid barrier between or inside them (the kernel boundary is
// the barrier), bit-identical to ppo3w_reduce_kernel + rlhip_clip_adam_f32 + ppo3w_pack_kernel (four launches):
//   ppo3w_reduce_sumsq_kernel  256 workgroups, element i = 256 b + t + 65536 k (the element-to-thread map of
//       sumsq_scaled_partial_kernel): partial rows -> gradient in the order of ppo3w_reduce_kernel (four row groups,
//       ascending inside a group, ((a0 + a1) + a2) + a3), 64 loads in flight per trip; the Float64 sum of squares of the
//       workgroup's elements in the same order as the stand-alone kernel; the PPO loss line
//   ppo3w_adam_pack_kernel     clip_adam_grid_kernel (norm from the 256 partials, clip, Adam, beta powers by the last workgroup
//       out) + the bf16 re-pack of both nets' W2 in both fragment orientations (parameter-centric)
constexpr int W3T_BLOCKS = 256;

// one row group (quarter) of an element's partial rows, ascending, 32 loads in flight per trip
__device__ __forceinline__ float sum_rows_quarter(const float* __restrict__ src, int64_t stride, int nrows, int qtr) {
    const int per = (nrows + 3) / 4;
    const int b0 = qtr * per, b1 = min(nrows, b0 + per);
    float acc = 0.f;
    for (int off = b0; off < b1; off += 32) {
        float t[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) t[u] = (off + u < b1) ? src[(int64_t)(off + u) * stride] : 0.0f;
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += t[u];  // x + 0.0f is exact: the padded slots change no bit
    }
    return acc;
}

// 1024 threads: thread (t = tid & 255, qtr = tid >> 8) sums row group `qtr` of element 256 b + t + 65536 k; threads of group 0
// (waves 0..3, i.e. exactly the 256-thread workgroup of sumsq_scaled_partial_kernel) combine and carry the squares.  The row
// sums of a thread's (up to W3T_EPT) elements are independent: they are all issued before the one exchange through LDS.
constexpr int W3T_EPT = 3;
__global__ __launch_bounds__(1024) void ppo3w_reduce_sumsq_kernel(const float* __restrict__ partS, const float* __restrict__ partW,
                                                                  const float* __restrict__ loss_partials, int nrowsS,
                                                                  int nrowsW, int npS, int nS_a, int np, int np_a, int ns,
                                                                  float* __restrict__ grad, float* __restrict__ losses,
                                                                  float wa, float wc, float we, float inv_b, float grad_scale,
                                                                  double* __restrict__ sumsq, int wnets) {
    __shared__ double scratch[16];
    __shared__ float l_q[W3T_EPT][3][256];
    __shared__ float l_loss[4];
    const int tid = threadIdx.x, t = tid & 255, qtr = tid >> 8, lane = tid & 63, wv = tid >> 6;
    double acc = 0.0;
    const int64_t stride_t = (int64_t)gridDim.x * 256;
    const int nk = (int)((np + stride_t - 1) / stride_t);
    for (int k0 = 0; k0 < nk; k0 += W3T_EPT) {
        float part[W3T_EPT];
#pragma unroll
        for (int u = 0; u < W3T_EPT; ++u) {
            const int64_t i = (int64_t)blockIdx.x * 256 + t + (k0 + u) * stride_t;
            part[u] = 0.f;
            if (k0 + u < nk && i < np) {
                const int p = (int)i;
                const int net = p >= np_a ? 1 : 0;
                const int q = p - net * np_a;
                const int nA = HW * ns + HW;
                if (q < nA) part[u] = sum_rows_quarter(partS + (net ? nS_a : 0) + q, npS, nrowsS, qtr);
                else if (q < nA + HW * HW)
                    part[u] = sum_rows_quarter(partW + (int64_t)net * HW * HW + (q - nA), (int64_t)wnets * HW * HW, nrowsW, qtr);
                else part[u] = sum_rows_quarter(partS + (net ? nS_a : 0) + (q - HW * HW), npS, nrowsS, qtr);
            }
        }
        if (qtr > 0) {
#pragma unroll
            for (int u = 0; u < W3T_EPT; ++u) l_q[u][qtr - 1][t] = part[u];
        }
        __syncthreads();
        if (qtr == 0) {
#pragma unroll
            for (int u = 0; u < W3T_EPT; ++u) {  // ascending k: the accumulation order of the stand-alone kernel
                const int64_t i = (int64_t)blockIdx.x * 256 + t + (k0 + u) * stride_t;
                if (k0 + u < nk && i < np) {
                    const float gi = ((part[u] + l_q[u][0][t]) + l_q[u][1][t]) + l_q[u][2][t];
                    grad[i] = gi;
                    const float x = gi * grad_scale;
                    acc += (double)x * (double)x;
                }
            }
        }
        __syncthreads();
    }
    // block_sum of a 256-thread workgroup, on waves 0..3
    acc = wave_sum_down_f64_lane0(acc);  // (wave_sum's tree; only lane 0 is read)
    if (qtr == 0 && lane == 0) scratch[wv] = acc;
    __syncthreads();
    if (tid == 0) sumsq[blockIdx.x] = ((0.0 + scratch[0]) + scratch[1] + scratch[2]) + scratch[3];
    if (blockIdx.x == 0 && losses != nullptr) {
        if (wv < 3) {
            float a = 0.f;
            for (int b = lane; b < nrowsS; b += 64) a += loss_partials[(int64_t)b * 4 + wv];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) l_loss[wv] = a;
        }
        __syncthreads();
        if (tid == 0 && wa < 0.0f) {  // DQN (marked by a negative actor weight): the mean Huber loss, one number
            losses[0] = l_loss[0] * inv_b;
        } else if (tid == 0) {
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

__global__ __launch_bounds__(256) void ppo3w_adam_pack_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, float* __restrict__ beta_pow, int np,
                                                              int np_a, int ns, float grad_scale, float clip_norm, float lr,
                                                              float b1, float b2, float eps, const double* __restrict__ sumsq,
                                                              int npart, unsigned int* __restrict__ departed,
                                                              uint16_t* __restrict__ packed, float* __restrict__ gn_out) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npart; i += blockDim.x) acc += sumsq[i];
    acc = block_sum_f64_dpp(acc, scratch);
    const float gn = (float)sqrt(acc);
    const float scale = (clip_norm > 0.0f && clip_norm <= gn) ? clip_norm / fmaxf(clip_norm, gn) : 1.0f;
    const float c1 = 1.0f - load_once(beta_pow), c2 = 1.0f - load_once(beta_pow + 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < np; i0 += W3T_EPT * stride) {
        float gv[W3T_EPT], pv[W3T_EPT], mv[W3T_EPT], vv[W3T_EPT];
#pragma unroll
        for (int u = 0; u < W3T_EPT; ++u) {  // every load of the trip before the first dependent instruction
            const int64_t i = i0 + u * stride;
            const bool own = i < np;
            gv[u] = own ? g[i] : 0.f;
            pv[u] = own ? p[i] : 0.f;
            mv[u] = own ? m[i] : 0.f;
            vv[u] = own ? v[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < W3T_EPT; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= np) continue;
            float gi = gv[u] * grad_scale;
            if (scale != 1.0f) gi *= scale;
            float pi = pv[u], mi = mv[u], vi = vv[u];
            adam1(pi, gi, mi, vi, lr, b1, b2, eps, c1, c2);
            p[i] = pi;
            m[i] = mi;
            v[i] = vi;
            g[i] = gi;
            // ppo3w_pack_kernel, parameter-centric: W2[j + HW k] goes to one slot of each fragment orientation
            const int net = i >= np_a ? 1 : 0;
            const int e = (int)i - net * np_a - (HW * ns + HW);
            if (e >= 0 && e < HW * HW) {
                const int j = e & (HW - 1), k = e / HW;
                const uint16_t hb = f32_to_bf16_rne(pi);
                const int q1 = ((((k >> 4) * WV + (j >> 5)) * 64) + ((j & 31) + 32 * ((k >> 3) & 1))) * 8 + (k & 7);
                const int q2 = ((((j >> 4) * WV + (k >> 5)) * 64) + ((k & 31) + 32 * ((j >> 3) & 1))) * 8 + (j & 7);
                uint16_t* pk = packed + (int64_t)net * 2 * HW * HW;
                pk[q1] = hb;
                pk[HW * HW + q2] = hb;
            }
        }
    }
    __syncthreads();  // every thread of this workgroup has read beta_pow
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0 && gn_out) gn_out[0] = gn;
        // (no release fence: nothing this workgroup stored is read by another workgroup of the launch)
        depart_barrier();
        const unsigned int prev = __hip_atomic_fetch_add(departed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {  // last one out: nobody reads beta_pow any more
            beta_pow[0] *= b1;
            beta_pow[1] *= b2;
            __hip_atomic_store(departed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ rollout
// 32 env instances per workgroup for all T vec-steps (ppo3_rollout32_kernel at width 256): 8 waves share the 32-row tile,
// wave w multiplies it by its 32-column block of W2 -- both nets' blocks live in registers for the whole launch.
constexpr int R32W = 32;
constexpr int LDH2W = HW + 4;   // f32 pitch of the H2 tiles
constexpr int NCHW = 8;         // sampling noise of 8 steps per evaluation (threads 0..255: 8 steps x 32 envs)
constexpr int NPARTW = NTW / R32W;  // 16 column parts per H2 row in the head phase
constexpr size_t ROLLW_NOISE_OFF = (((4 * R32W + NPARTW * 4 * R32W + 2 * R32W * LDH2W + 2 * SMALLWW) * sizeof(float) +
                                     (2 * R32W * PW) * sizeof(uint16_t)) + 15) & ~(size_t)15;
constexpr size_t ROLLW_LDS = ROLLW_NOISE_OFF + 2 * NCHW * R32W * MAXO * sizeof(double);

// CRIT = false: the actor alone (one set of W2 fragments resident: no register spills, half the layer-1 / MFMA / head work
// on the per-step critical path); the values V(s_0 .. s_T) then come from ONE batched pass over the trajectory's
// observations (ppo3w_fwd_kernel<.., 4>) and the GAE scan from its own launch -- V is needed by nobody during the rollout.
template <class P, int NOUT_A, int ACT, bool CRIT>
__global__ __launch_bounds__(NTW) void ppo3w_rollout_kernel(P p, EnvArrays<float> st, int64_t n, int T, int cont, int na,
                                                            const float* __restrict__ params, int64_t np_a, uint64_t seed,
                                                            uint32_t env_id_base, uint32_t vec_step0, TrajPtrs tr, float gamma,
                                                            float lambda) {
    constexpr int NS = P::ODIM;
    constexpr int NO = NOUT_A + 1;  // head outputs per env: the actor's, then the value
    extern __shared__ __attribute__((aligned(16))) char smw[];
    float* l_x = reinterpret_cast<float*>(smw);   // [4][R32W]
    float* l_part = l_x + 4 * R32W;               // [NPARTW][4][R32W]
    float* l_h2a = l_part + NPARTW * 4 * R32W;    // [R32W][LDH2W] actor H2 (f32)
    float* l_h2c = l_h2a + R32W * LDH2W;          // critic H2
    float* l_w = l_h2c + R32W * LDH2W;            // [2][SMALLWW]
    uint16_t* l_Ha = reinterpret_cast<uint16_t*>(l_w + 2 * SMALLWW);  // actor H1 tile [R32W][PW] (bf16)
    uint16_t* l_Hc = l_Ha + R32W * PW;            // critic H1 tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const Mlp3W ma = stage_small_w(params, NS, NOUT_A, l_w, tid);
    const Mlp3W mc = stage_small_w(params + np_a, NS, 1, l_w + SMALLWW, tid);
    bf16x8 bwa[KSW], bwc[CRIT ? KSW : 1];
    load_frags_f32(params + HW * NS + HW, w, lane, bwa);
    if constexpr (CRIT) load_frags_f32(params + np_a + HW * NS + HW, w, lane, bwc);

    const int64_t env = (int64_t)blockIdx.x * R32W + tid;
    const bool active = tid < R32W && env < n;
    const int64_t envc = env < n ? env : n - 1;
    const uint32_t id = env_id_base + (uint32_t)envc;
    LaneState<float> e;
    float last_r = 0.0f;
    bool last_d = false;
    if (tid < R32W) {
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][envc];
        e.t = st.t[envc];
        e.episode = st.episode[envc];
    }
    const int row1 = tid & 31, u0 = 16 * (tid >> 5);  // layer 1: this thread's row and its 16 hidden units
    const int part = tid >> 5;                        // heads: this thread's 16-column part of the H2 row `row1`
    double* l_noise = reinterpret_cast<double*>(smw + ROLLW_NOISE_OFF);  // [2][NCHW][R32W][MAXO]
    __syncthreads();
    const int colw = 32 * w + r;
    const float b2a = ma.b2[colw], b2c = mc.b2[colw];
    for (int t = 0; t <= T; ++t) {
        if ((t & (NCHW - 1)) == 0 && tid < NCHW * R32W) {
            const int i = tid >> 5, er = tid & (R32W - 1);
            if (t + i < T) {
                const int64_t en = (int64_t)blockIdx.x * R32W + er;
                double nz[MAXO] = {0.0, 0.0, 0.0, 0.0};
                policy_noise(cont, na, seed, env_id_base + (uint32_t)(en < n ? en : n - 1), vec_step0 + (uint32_t)(t + i), nz);
                double* dst = l_noise + ((size_t)((((t / NCHW) & 1) * NCHW + i) * R32W + er)) * MAXO;
#pragma unroll
                for (int k = 0; k < MAXO; ++k) dst[k] = nz[k];
            }
        }
        if (tid < R32W) {
            float x[4];
            env_obs1(p, e, x);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                l_x[k * R32W + tid] = x[k];
                if (active) tr.obs[((int64_t)t * NS + k) * n + env] = x[k];
            }
        }
        if (!CRIT && t == T) break;  // the last pass only records s_T (its value comes from the batched pass)
        __syncthreads();
        // ---- layer 1 of both nets, 16 units per thread ----
        {
            float x[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = l_x[i * R32W + row1];
#pragma unroll
            for (int net = 0; net < (CRIT ? 2 : 1); ++net) {
                if (net == 0 && t == T) continue;  // the last pass only needs V(s_T)
                const Mlp3W& m = net ? mc : ma;
                uint16_t* dst = (net ? l_Hc : l_Ha) + row1 * PW + u0;
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
                            const float4 wv = *reinterpret_cast<const float4*>(m.W1 + u + HW * i);
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
        // ---- layer 2: this wave's 32 output columns of both nets (MFMA), bias + activation, f32 tiles to LDS ----
        {
            f32x16 aa, ac;
#pragma unroll
            for (int q = 0; q < 16; ++q) aa[q] = 0.0f, ac[q] = 0.0f;
            const uint16_t* apa = l_Ha + r * PW + 8 * kb;
            const uint16_t* apc = l_Hc + r * PW + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                if (t < T) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(apa + 16 * ks);
                    aa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwa[ks], aa, 0, 0, 0);
                }
                if constexpr (CRIT) {
                    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(apc + 16 * ks);
                    ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bwc[ks], ac, 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = mfma_row(q, kb);
                if (t < T) l_h2a[row * LDH2W + colw] = act_fwd_t<ACT>(aa[q] + b2a);
                if constexpr (CRIT) l_h2c[row * LDH2W + colw] = act_fwd_t<ACT>(ac[q] + b2c);
            }
        }
        __syncthreads();
        // ---- heads: thread (row1, part) folds 16 columns of its H2 row for every output ----
        {
            float pa[NOUT_A], pc = 0.0f;
#pragma unroll
            for (int o = 0; o < NOUT_A; ++o) pa[o] = 0.0f;
            const float* ha = l_h2a + row1 * LDH2W + 16 * part;
            const float* hc = l_h2c + row1 * LDH2W + 16 * part;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                float hcv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (CRIT) {
                    const float4 vc = *reinterpret_cast<const float4*>(hc + 4 * c4);
                    hcv[0] = vc.x, hcv[1] = vc.y, hcv[2] = vc.z, hcv[3] = vc.w;
                }
                float hav[4] = {0.f, 0.f, 0.f, 0.f};
                if (t < T) {
                    const float4 va = *reinterpret_cast<const float4*>(ha + 4 * c4);
                    hav[0] = va.x, hav[1] = va.y, hav[2] = va.z, hav[3] = va.w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = 16 * part + 4 * c4 + c;
                    if constexpr (CRIT) pc = fmaf(mc.W3[j], hcv[c], pc);
#pragma unroll
                    for (int o = 0; o < NOUT_A; ++o) pa[o] = fmaf(ma.W3[o + NOUT_A * j], hav[c], pa[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < NOUT_A; ++o) l_part[(part * 4 + o) * R32W + row1] = pa[o];
            if constexpr (CRIT) l_part[(part * 4 + NOUT_A) * R32W + row1] = pc;
        }
        __syncthreads();
        if (tid < R32W) {
            float out[NO];
#pragma unroll
            for (int o = 0; o < (CRIT ? NO : NOUT_A); ++o) {
                float acc = l_part[o * R32W + tid];
#pragma unroll
                for (int pp = 1; pp < NPARTW; ++pp) acc += l_part[(pp * 4 + o) * R32W + tid];
                out[o] = acc + (o < NOUT_A ? ma.b3[o] : mc.b3[0]);
            }
            if constexpr (CRIT) {
                const float v = out[NOUT_A];
                if (active) tr.value[(int64_t)t * n + env] = v;
            }
            if (t < T) {
                float oa[MAXO];
#pragma unroll
                for (int o = 0; o < MAXO; ++o) oa[o] = (o < NOUT_A) ? out[o] : 0.0f;
                int32_t ai;
                float af, lp;
                policy_select(cont, na, oa, l_noise + ((size_t)((((t / NCHW) & 1) * NCHW + (t & (NCHW - 1))) * R32W + tid)) * MAXO,
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
        // no barrier here (as ppo3_rollout32_kernel): the next pass rewrites l_x from the same 32 lanes in program order and
        // every other buffer only after the next pass's barriers
    }
    if (CRIT && active && T > 0 && tr.adv && tr.ret)  // GAE + returns fused into the rollout launch (gae_device.h)
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

// ------------------------------------------------------------------------------------------------ host side
constexpr int P3W_ROWS_S = 512;  // upper bound of the persistent workgroups (= partial rows) of the forward / backward kernels
constexpr int P3W_ROWS_W = 128;  // sample ranges (= partial rows) of the dW2 kernel; its grid is twice that (two k halves)
constexpr int64_t P3W_MAX_TILES = 1 << 20;

template <typename K>
static int32_t allow_lds_w(K kernel, size_t bytes, unsigned long long* done) { return allow_big_lds(kernel, bytes, done); }

static int p3w_rows_w() { return P3W_ROWS_W; }

struct P3WLayout {
    int64_t ntiles, off_xg, off_sg, off_rows, off_frag, off_partS, off_partW, off_loss, off_tail, off_rec, bytes;
    int npS, nS_a;
};

static P3WLayout p3w_layout(int ns, int nout_a, const rlhip_ppo_cfg* c, int64_t n, int64_t T) {
    P3WLayout L;
    const int64_t bm = (n * T) / (c->n_microbatches > 0 ? c->n_microbatches : 1);
    L.ntiles = (bm + RW - 1) / RW;
    L.nS_a = mlp3w_ns_small(ns, nout_a);
    L.npS = L.nS_a + mlp3w_ns_small(ns, 1);
    int64_t o = 4 * (int64_t)HW * HW * sizeof(uint16_t);
    L.off_xg = o;
    o += 4 * L.ntiles * RW * (int64_t)sizeof(float);
    L.off_sg = o;
    o += 4 * L.ntiles * RW * (int64_t)sizeof(float);
    L.off_rows = o;
    o += L.ntiles * RW * HW * (int64_t)sizeof(uint16_t);
    L.off_frag = o;
    o += 2 * L.ntiles * RW * HW * (int64_t)sizeof(uint16_t);  // one fragment buffer per net (the dW2 launch reads both)
    L.off_partS = o;
    o += (int64_t)P3W_ROWS_S * L.npS * (int64_t)sizeof(float);
    L.off_partW = o;
    o += (L.ntiles < 256 ? L.ntiles : 256) * 2 * (int64_t)HW * HW * (int64_t)sizeof(float);  // <= 256 dW2 sample ranges
    L.off_loss = o;
    o += (int64_t)P3W_ROWS_S * 4 * (int64_t)sizeof(float);
    o = (o + 63) & ~(int64_t)63;
    L.off_tail = o;  // W3T_BLOCKS Float64 partial sums of squares + the departure counter (zero-initialised workspace: ABI)
    o += W3T_BLOCKS * (int64_t)sizeof(double) + 64;
    L.off_rec = o;  // ppo3w_update's 32-byte records of the whole trajectory
    o += n * T * 8 * (int64_t)sizeof(float);
    L.bytes = o + 256;
    return L;
}

// ---- which LDS copy the backward kernel uses (RLHIP_W3_DZF_PAD above): 0 / 1 forced, 2 = chosen by the chip's clock ----------------------------
// The two kernels are bit-identical; which one is faster is a property of the BOX (profiles/r06_ppo3w.md section 5): the padded copy needs 5 % fewer
// cycles per optimiser step, and on boxes whose firmware limiter is active it is given a 6 - 8 % lower clock for it.  Measured model (nine
// contacts): unpadded runs at >= 0.985 of the top clock exactly on the boxes where padded holds >= 0.94 of it, and padded wins iff its clock stays
// above 0.946 of the unpadded one.  Mode 2 follows that model with the device's own hwmon reading: every 512 backward launches (~50 ms of a
// training loop) the host reads freq1_input (one sysfs read, ~10 us, no GPU work); six consecutive dense readings >= 0.985 top -> padded; four
// consecutive dense readings < 0.94 top while padded -> back, and padded is not tried again for 2 s x 2^k.  Readings across a gap (> 0.25 s
// since the last one: rollouts, host stalls) and the first eight readings behind a gap (the ramp from the sleep clock) are ignored.  The start is
// optimistic (padded) when the sensor is there; no sensor -> unpadded.
struct W3Pad {
    int mode = -1;     // -1 not initialised; 0 / 1 forced; 2 automatic
    int variant = 0;   // what the next launch uses
    char path[320] = {0};
    double top_mhz = 0.0, last_mhz = 0.0, last_t = 0.0;
    unsigned long long launches = 0, next_check = 512, probation_until = 0;
    int hi_run = 0, lo_run = 0, dense_run = 0, settle = 0, backoff = 0, switches = 0;
};
static W3Pad g_w3pad;
static std::mutex g_w3pad_mu;

static double w3pad_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void w3pad_find_sensor(W3Pad& P) {
    int dev = 0, khz = 0;
    char bus[64] = {0};
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) return;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev) == hipSuccess) P.top_mhz = khz * 1e-3;
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    char dir[160];
    snprintf(dir, sizeof(dir), "/sys/bus/pci/devices/%s/hwmon", bus);
    DIR* d = opendir(dir);
    if (d == nullptr) return;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
        char f[320];
        snprintf(f, sizeof(f), "%s/%s/freq1_input", dir, e->d_name);
        if (access(f, R_OK) == 0) {
            snprintf(P.path, sizeof(P.path), "%s", f);
            break;
        }
    }
    closedir(d);
    if (P.top_mhz < 500.0) P.path[0] = 0;  // no usable top clock: stay with the default kernel
}
static double w3pad_read_mhz(const W3Pad& P) {
    FILE* f = fopen(P.path, "r");
    if (f == nullptr) return 0.0;
    long long hz = 0;
    const int n = fscanf(f, "%lld", &hz);
    fclose(f);
    return n == 1 ? (double)hz * 1e-6 : 0.0;
}
static void w3pad_init(W3Pad& P) {
    const char* e = getenv("RLHIP_W3_DZF_PAD");
    int m = RLHIP_W3_DZF_PAD;
    if (e != nullptr && e[0] != 0) m = (e[0] == 'a' || e[0] == '2') ? 2 : (e[0] != '0' ? 1 : 0);
    P.mode = m;
    P.variant = m == 1 ? 1 : 0;
    if (m == 2) {
        w3pad_find_sensor(P);
        P.variant = P.path[0] != 0;  // optimistic start: padded until the clock says otherwise (a burst is 5 % faster with it where the chip has
                                     // headroom and 1 % slower where it has not; sustained, the readings decide within ~0.6 s)
    }
}
// one call per backward launch
static bool w3_dzf_pad() {
    if (RLHIP_W3_DZ_ONCE != 2) return false;
    std::lock_guard<std::mutex> lk(g_w3pad_mu);
    W3Pad& P = g_w3pad;
    if (P.mode < 0) w3pad_init(P);
    if (P.mode != 2 || P.path[0] == 0) return P.variant != 0;
    if (++P.launches < P.next_check) return P.variant != 0;
    P.next_check = P.launches + 512;
    const double t = w3pad_now(), mhz = w3pad_read_mhz(P);
    const bool dense = P.last_t > 0.0 && t - P.last_t < 0.25;
    P.last_t = t;
    P.last_mhz = mhz;
    if (!dense || mhz <= 0.0) {
        P.hi_run = P.lo_run = P.dense_run = 0;
        return P.variant != 0;
    }
    if (++P.dense_run < 8) return P.variant != 0;  // the first ~0.4 s of load after a gap are the ramp from the sleep clock, not the limiter
    if (P.settle > 0) {  // the firmware takes a few tenths of a second to answer a change of the kernel mix
        --P.settle;
        return P.variant != 0;
    }
    if (P.variant == 0) {
        P.hi_run = mhz >= 0.985 * P.top_mhz ? P.hi_run + 1 : 0;
        if (P.hi_run >= 6 && P.launches >= P.probation_until) {
            P.variant = 1, P.hi_run = 0, P.lo_run = 0, P.settle = 8, ++P.switches;
        }
    } else {
        P.lo_run = mhz < 0.94 * P.top_mhz ? P.lo_run + 1 : 0;
        if (P.lo_run >= 4) {
            P.variant = 0, P.hi_run = 0, P.lo_run = 0, P.settle = 8, ++P.switches;
            P.probation_until = P.launches + (20480ull << (P.backoff < 8 ? P.backoff : 8));  // ~2 s of launches, doubling
            ++P.backoff;
        }
    }
    return P.variant != 0;
}
// not part of the ABI: on = 0 / 1 force a kernel, 2 automatic, < 0 query; returns the kernel the next launch uses (0 / 1).  `info` (may be null)
// receives {mode, variant, last reading in MHz, top clock in MHz, switches so far, sensor found}
extern "C" int32_t rlhip_debug_w3_dzf_pad_info(int32_t on, double* info) {
    std::lock_guard<std::mutex> lk(g_w3pad_mu);
    W3Pad& P = g_w3pad;
    if (P.mode < 0) w3pad_init(P);
    if (on >= 0) {
        P.mode = on > 1 ? 2 : on;
        if (P.mode != 2) P.variant = P.mode;
        else if (P.path[0] == 0 && P.top_mhz == 0.0) w3pad_find_sensor(P);
        P.hi_run = P.lo_run = P.dense_run = P.settle = 0, P.last_t = 0.0;
    }
    if (info != nullptr) {
        info[0] = P.mode, info[1] = P.variant, info[2] = P.last_mhz, info[3] = P.top_mhz, info[4] = P.switches, info[5] = P.path[0] != 0;
    }
    return RLHIP_W3_DZ_ONCE == 2 ? P.variant : 0;
}
extern "C" int32_t rlhip_debug_w3_dzf_pad(int32_t on) {
    const int32_t prev = rlhip_debug_w3_dzf_pad_info(-1, nullptr);
    if (on >= 0) rlhip_debug_w3_dzf_pad_info(on, nullptr);
    return prev;
}
#define W3_LAUNCH_BWD(NS_, ACT_, net_)                                                                                               \
    do {                                                                                                                             \
        static unsigned long long dp0_ = 0, dp1_ = 0;                                                                                \
        int32_t rcb_;                                                                                                                \
        if (w3_dzf_pad()) {                                                                                                          \
            if ((rcb_ = allow_lds_w(ppo3w_bwd_kernel<NS_, ACT_, true>, Dzf<true>::LDS, &dp1_))) return rcb_;                         \
            hipLaunchKernelGGL((ppo3w_bwd_kernel<NS_, ACT_, true>), dim3(nrowsS), dim3(NTW), Dzf<true>::LDS, s, g, net_);            \
        } else {                                                                                                                     \
            if ((rcb_ = allow_lds_w(ppo3w_bwd_kernel<NS_, ACT_, false>, Dzf<false>::LDS, &dp0_))) return rcb_;                       \
            hipLaunchKernelGGL((ppo3w_bwd_kernel<NS_, ACT_, false>), dim3(nrowsS), dim3(NTW), Dzf<false>::LDS, s, g, net_);          \
        }                                                                                                                            \
    } while (0)

#ifdef RLHIP_W3_TIMING
extern "C" int32_t rlhip_debug_w3_stamps(long long* out_host) {
    RLHIP_CHECK_HIP(hipDeviceSynchronize());
    RLHIP_CHECK_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_w3_stamps), 3 * 16 * sizeof(long long)));
    return RLHIP_OK;
}
#endif

int64_t ppo3w_nparams(int ns, int nout_a) { return mlp3w_np(ns, nout_a) + mlp3w_np(ns, 1); }

int64_t ppo3w_workspace_bytes(int ns, int nout_a, const rlhip_ppo_cfg* c, int64_t n, int64_t T) {
    return p3w_layout(ns, nout_a, c, n, T).bytes;
}

static int32_t ppo3w_pack(const float* params, int ns, int64_t np_a, uint16_t* packed, hipStream_t s) {
    hipLaunchKernelGGL(ppo3w_pack_kernel, dim3(2 * HW * HW / 256), dim3(256), 0, s, params, ns, np_a, packed);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

template <class P>
static int32_t rollout3w_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                              const PolicyDesc& pd, const float* params, uint64_t seed, uint32_t env_id_base, uint32_t vec_step0, const rlhip_ppo_traj* traj, hipStream_t s) {
    RLHIP_REQUIRE(st && st->episode, "this entry point needs the separate episode[] array (packed step / episode words are an rlhip_env_step / rlhip_env_reset mode)");
    typename P::cfg_t c2 = *cfg;
    c2.continuous = pd.cont;
    P p = P::make(c2);
    EnvArrays<float> a = EnvArrays<float>::from(*st);
    TrajPtrs tr = TrajPtrs::from(*traj);
    dim3 grid((unsigned)((n + R32W - 1) / R32W));
    constexpr int NS = P::ODIM;
    // 1. the actor's rollout: T vec-steps in one launch (obs, action, log-prob, reward, terminal traces)
    // 2. V(s_0 .. s_T): one batched critic pass over the (T + 1) n recorded observations (MFMA forward, persistent
    //    workgroups, one per CU)
    // 3. the GAE + returns scan (the kernel of rlhip_ppo_gae_f32)
    P3WArgs g{};
    g.obs = traj->obs;
    g.params = params;
    g.np_a = pd.np_a;
    g.n = n;
    g.sg = traj->value;
    const int64_t total = (T + 1) * n;
    RLHIP_REQUIRE(traj->value != nullptr && traj->obs != nullptr, "trajectory array is NULL");
    RLHIP_REQUIRE(total <= 0x7FFFFFFFll, "(T + 1) n out of range");
    g.bm = (uint32_t)total;
    g.ntiles = (int)((total + RW - 1) / RW);
    g.npad = g.ntiles * RW;
    const int n_cu = device_cu_count() < P3W_ROWS_S ? device_cu_count() : P3W_ROWS_S;
    const int nwg = g.ntiles < n_cu ? g.ntiles : n_cu;
#define LAUNCH_RW(ACT_)                                                                                            \
    do {                                                                                                           \
        static unsigned long long done_ = 0, donev_ = 0;                                                           \
        int32_t rc_ = allow_lds_w(ppo3w_rollout_kernel<P, 2, ACT_, false>, ROLLW_LDS, &done_);                     \
        if (rc_) return rc_;                                                                                       \
        if ((rc_ = allow_lds_w(ppo3w_fwd_kernel<NS, 1, ACT_, 0, 4>, FWDW_LDS, &donev_))) return rc_;               \
        hipLaunchKernelGGL((ppo3w_rollout_kernel<P, 2, ACT_, false>), grid, dim3(NTW), ROLLW_LDS, s, p, a, n, (int)T, pd.cont, \
                           pd.na, params, pd.np_a, seed, env_id_base, vec_step0, tr, pd.gamma, pd.lambda);         \
        hipLaunchKernelGGL((ppo3w_fwd_kernel<NS, 1, ACT_, 0, 4>), dim3(nwg), dim3(NTW), FWDW_LDS, s, g);           \
    } while (0)
    if (pd.act == 0) LAUNCH_RW(0);
    else LAUNCH_RW(1);
#undef LAUNCH_RW
    RLHIP_LAUNCH_CHECK();
    if (T > 0 && traj->adv && traj->ret)
        return rlhip_gae_returns_f32(traj->adv, traj->ret, traj->reward, traj->value, traj->terminal, n, T, pd.gamma,
                                     pd.lambda, (rlhip_stream_t)s);
    return RLHIP_OK;
}

int32_t ppo3w_rollout(int32_t kind, const void* env_cfg, const rlhip_env_state* st, int64_t n, int64_t T,
                      const PolicyDesc& pd_in, const float* params, uint64_t seed, uint32_t env_id_base, uint32_t vec_step0,
                      const rlhip_ppo_traj* traj, rlhip_stream_t stream) {
    PolicyDesc pd = pd_in;
    const int ns = kind == 0 ? 4 : 3;
    pd.np_a = mlp3w_np(ns, pd.nout_a);
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return rollout3w_impl<CartPoleParams<float>>((const rlhip_cartpole_cfg*)env_cfg, st, n, T, pd, params, seed,
                                                     env_id_base, vec_step0, traj, s);
    return rollout3w_impl<PendulumParams<float>>((const rlhip_pendulum_cfg*)env_cfg, st, n, T, pd, params, seed,
                                                 env_id_base, vec_step0, traj, s);
}

// optimise! state for the two-launch tail (ppo3w_update); NULL: plain reduce into grad_out (the caller applies)
struct P3WTail {
    float *params, *m, *v, *beta_pow;
    bool rec_ready;  // the workspace holds the record copy of THIS trajectory (ppo3w_update built it)
};

static int32_t ppo3w_grad_impl(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                               const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb,
                               void* workspace, float* grad_out, float* losses_out, bool do_pack, const P3WTail* tail,
                               rlhip_stream_t stream);

int32_t ppo3w_grad(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                   const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb,
                   void* workspace, float* grad_out, float* losses_out, rlhip_stream_t stream) {
    return ppo3w_grad_impl(kind, cfg, pd, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_out, losses_out, true, nullptr,
                           stream);
}

static int32_t ppo3w_grad_impl(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                               const rlhip_ppo_traj* traj, const float* params, uint64_t seed, uint32_t epoch_ctr, int32_t mb,
                               void* workspace, float* grad_out, float* losses_out, bool do_pack, const P3WTail* tail,
                               rlhip_stream_t stream) {
    const int ns = kind == 0 ? 4 : 3;
    const int64_t total = n * T;
    RLHIP_REQUIRE(total >= 1 && total <= 0x7FFFFFFFll, "n * T out of range");
    const int64_t bm = total / cfg->n_microbatches;
    RLHIP_REQUIRE(bm >= 1, "empty micro-batch");
    const P3WLayout L = p3w_layout(ns, pd.nout_a, cfg, n, T);
    RLHIP_REQUIRE(L.ntiles <= P3W_MAX_TILES, "micro-batch too large for one launch");
    RLHIP_REQUIRE(cfg->actor_loss_weight >= 0.0f, "actor_loss_weight must be >= 0 (a negative value marks the DQN loss line)");
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    P3WArgs g;
    g.obs = traj->obs;
    g.logp = traj->logp;
    g.adv = traj->adv;
    g.ret = traj->ret;
    g.action_f = traj->action_f;
    g.action_i = traj->action_i;
    RLHIP_REQUIRE(g.obs && g.logp && g.adv && g.ret && (pd.cont ? (const void*)g.action_f : (const void*)g.action_i),
                  "trajectory array is NULL");
    g.params = params;
    uint16_t* packed = (uint16_t*)ws;
    g.packed = packed;
    g.xg = (float*)(ws + L.off_xg);
    g.sg = (float*)(ws + L.off_sg);
    g.npad = (int)(L.ntiles * RW);
    g.wnets = 2;
    g.tparams = nullptr;
    g.tpacked = nullptr;
    g.xg2 = nullptr;
    g.td_out = nullptr;
    g.gamma = 0.0f;
    g.delta = 0.0f;
    g.rec = (tail != nullptr && tail->rec_ready) ? (const float*)(ws + L.off_rec) : nullptr;
    g.dz_rows = (uint16_t*)(ws + L.off_rows);
    g.dz_frag = (uint16_t*)(ws + L.off_frag);
    g.frag_stride = L.ntiles * RW * HW;
    g.partS = (float*)(ws + L.off_partS);
    g.partW = (float*)(ws + L.off_partW);
    g.loss_partials = (float*)(ws + L.off_loss);
    g.n = n;
    g.np_a = mlp3w_np(ns, pd.nout_a);
    const int np = (int)(g.np_a + mlp3w_np(ns, 1));
    g.total = (uint32_t)total;
    g.bm = (uint32_t)bm;
    g.pos0 = (uint32_t)(mb * bm);
    g.ntiles = (int)L.ntiles;
    g.npS = L.npS;
    g.nS_a = L.nS_a;
    g.na = pd.na;
    g.lo = 1.0f - cfg->clip_range;
    g.hi = 1.0f + cfg->clip_range;
    g.wa = cfg->actor_loss_weight;
    g.wc = cfg->critic_loss_weight;
    g.we = cfg->entropy_loss_weight;
    g.inv_b = 1.0f / (float)bm;
    g.min_logp = (float)::log(1e-8);
    g.pk = perm_keys(seed, epoch_ctr, (uint32_t)total);
    if (do_pack) {
        int32_t rc = ppo3w_pack(params, ns, g.np_a, packed, s);
        if (rc) return rc;
    }
    // one persistent workgroup per CU (the kernels hold 160 - 220 registers per lane: 2 waves per SIMD = one 8-wave workgroup):
    // a second round of workgroups would pay the ~6000-cycle prologue (weights, fragments, first tile) twice
    const int n_cu = device_cu_count() < P3W_ROWS_S ? device_cu_count() : P3W_ROWS_S;
    const int nrowsS = (int)(L.ntiles < n_cu ? L.ntiles : n_cu);
    {
        const int gb = (g.npad + 255) / 256;
        if (g.rec != nullptr) {
            if (kind == 0) hipLaunchKernelGGL((ppo3w_gather_rec_kernel<4>), dim3(gb), dim3(256), 0, s, g);
            else hipLaunchKernelGGL((ppo3w_gather_rec_kernel<3>), dim3(gb), dim3(256), 0, s, g);
        } else {
            if (kind == 0) hipLaunchKernelGGL((ppo3w_gather_kernel<4, 0>), dim3(gb), dim3(256), 0, s, g);
            else hipLaunchKernelGGL((ppo3w_gather_kernel<3, 1>), dim3(gb), dim3(256), 0, s, g);
        }
    }
    // the dW2 launch covers both nets (blockIdx.y): 2 k halves x nsr sample ranges x 2 nets = one workgroup per CU at nsr = 64
    const int rows_w = (p3w_rows_w() + 1) / 2;
    const int nsr = (int)(L.ntiles < rows_w ? L.ntiles : rows_w);
#define LAUNCH_GW(NS_, ACT_, CONT_)                                                                                   \
    do {                                                                                                              \
        static unsigned long long d0_ = 0, d1_ = 0, d3_ = 0;                                                       \
        int32_t rc_;                                                                                                  \
        if ((rc_ = allow_lds_w(ppo3w_fwd_kernel<NS_, 2, ACT_, CONT_, 0>, FWDW_LDS, &d0_))) return rc_;                \
        if ((rc_ = allow_lds_w(ppo3w_fwd_kernel<NS_, 1, ACT_, CONT_, 1>, FWDW_LDS, &d1_))) return rc_;                \
        if ((rc_ = allow_lds_w(ppo3w_dw2_kernel<NS_, ACT_>, DW2W_LDS, &d3_))) return rc_;                             \
        if (RLHIP_W3_DZ_ONCE == 1) g.dz_rows = g.dz_frag; /* one row image per net, kept for the dW2 launch */              \
        hipLaunchKernelGGL((ppo3w_fwd_kernel<NS_, 2, ACT_, CONT_, 0>), dim3(nrowsS), dim3(NTW), FWDW_LDS, s, g);      \
        W3_LAUNCH_BWD(NS_, ACT_, 0);                                                                                  \
        if (RLHIP_W3_DZ_ONCE == 1) g.dz_rows = g.dz_frag + g.frag_stride;                                                  \
        hipLaunchKernelGGL((ppo3w_fwd_kernel<NS_, 1, ACT_, CONT_, 1>), dim3(nrowsS), dim3(NTW), FWDW_LDS, s, g);      \
        W3_LAUNCH_BWD(NS_, ACT_, 1);                                                                                  \
        hipLaunchKernelGGL((ppo3w_dw2_kernel<NS_, ACT_>), dim3(2 * nsr, 2), dim3(NTW), DW2W_LDS, s, g, -1, nsr);      \
    } while (0)
    if (kind == 0) {
        RLHIP_REQUIRE(!pd.cont, "layers = 3: CartPole uses the categorical head");
        if (pd.act == 0) LAUNCH_GW(4, 0, 0);
        else LAUNCH_GW(4, 1, 0);
    } else {
        RLHIP_REQUIRE(pd.cont, "layers = 3: Pendulum uses the Gaussian head");
        if (pd.act == 0) LAUNCH_GW(3, 0, 1);
        else LAUNCH_GW(3, 1, 1);
    }
#undef LAUNCH_GW
    if (tail != nullptr) {
        double* sumsq = (double*)(ws + L.off_tail);
        unsigned int* departed = (unsigned int*)(sumsq + W3T_BLOCKS);
        const int nbt = (int)((np + 255) / 256 < W3T_BLOCKS ? (np + 255) / 256 : W3T_BLOCKS);  // grid_for(np, 256, 256)
        hipLaunchKernelGGL(ppo3w_reduce_sumsq_kernel, dim3(nbt), dim3(1024), 0, s, g.partS, g.partW, g.loss_partials, nrowsS, nsr,
                           g.npS, g.nS_a, np, (int)g.np_a, ns, grad_out, losses_out, g.wa, g.wc, g.we, g.inv_b, 1.0f, sumsq, 2);
        hipLaunchKernelGGL(ppo3w_adam_pack_kernel, dim3(nbt), dim3(256), 0, s, tail->params, grad_out, tail->m, tail->v,
                           tail->beta_pow, np, (int)g.np_a, ns, 1.0f, cfg->max_grad_norm, cfg->lr, cfg->beta1, cfg->beta2,
                           cfg->adam_eps, (const double*)sumsq, nbt, departed, packed, (float*)nullptr);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    hipLaunchKernelGGL(ppo3w_reduce_kernel, dim3((np + 63) / 64), dim3(256), 0, s, g.partS, g.partW, g.loss_partials, nrowsS,
                       nsr, g.npS, g.nS_a, np, (int)g.np_a, ns, grad_out, losses_out, g.wa, g.wc, g.we, g.inv_b, 2);
    RLHIP_LAUNCH_CHECK();
    if (tail != nullptr)
        return rlhip_clip_adam_f32(tail->params, grad_out, tail->m, tail->v, tail->beta_pow, np, 1.0f, cfg->max_grad_norm, cfg->lr,
                                   cfg->beta1, cfg->beta2, cfg->adam_eps, nullptr, stream);
    return RLHIP_OK;
}

int32_t ppo3w_update(int32_t kind, const rlhip_ppo_cfg* cfg, const PolicyDesc& pd, int64_t n, int64_t T,
                     const rlhip_ppo_traj* traj, float* params, float* m, float* v, float* beta_pow, uint64_t seed,
                     uint32_t update_ctr, void* workspace, float* grad_scratch, float* losses_out, rlhip_stream_t stream) {
    const int ns = kind == 0 ? 4 : 3;
    const int64_t np = ppo3w_nparams(ns, pd.nout_a);
    (void)np;
    bool rec_ready = false;
    {
        RLHIP_REQUIRE(traj->obs && traj->logp && traj->adv && traj->ret && (pd.cont ? (const void*)traj->action_f : (const void*)traj->action_i),
                      "trajectory array is NULL");
        const int64_t total = n * T;
        RLHIP_REQUIRE(total >= 1 && total <= 0x7FFFFFFFll, "n * T out of range");
        const P3WLayout L = p3w_layout(ns, pd.nout_a, cfg, n, T);
        P3WArgs g{};
        g.obs = traj->obs;
        g.logp = traj->logp;
        g.adv = traj->adv;
        g.ret = traj->ret;
        g.action_f = traj->action_f;
        g.action_i = traj->action_i;
        g.n = n;
        g.total = (uint32_t)total;
        float* rec = (float*)((char*)workspace + L.off_rec);
        const int gb = (int)((total + 255) / 256);
        hipStream_t s = as_stream(stream);
        if (kind == 0) hipLaunchKernelGGL((ppo3w_build_rec_kernel<4, 0>), dim3(gb), dim3(256), 0, s, g, rec);
        else hipLaunchKernelGGL((ppo3w_build_rec_kernel<3, 1>), dim3(gb), dim3(256), 0, s, g, rec);
        rec_ready = true;
    }
    const P3WTail tail{params, m, v, beta_pow, rec_ready};
    const bool fused = true;
    bool packed_fresh = false;  // the previous optimiser step's tail left the bf16 images of both W2 up to date
    for (int32_t e = 0; e < cfg->n_epochs; ++e) {
        const uint32_t epoch_ctr = update_ctr * (uint32_t)cfg->n_epochs + (uint32_t)e;
        for (int32_t mb = 0; mb < cfg->n_microbatches; ++mb) {
            int32_t rc = ppo3w_grad_impl(kind, cfg, pd, n, T, traj, params, seed, epoch_ctr, mb, workspace, grad_scratch,
                                         losses_out, /*do_pack=*/!packed_fresh, &tail, stream);
            if (rc) return rc;
            packed_fresh = fused;
        }
    }
    return RLHIP_OK;
}

// ================================================================================================ DQN learner at hidden = 256
// The same three streaming kernels serve the 3-layer Q-network ns -> 256 -> 256 -> na of the QBasedPolicy / DQN learner
// (dqn3.hip dispatches here on h == 256; reference code replaced, loss and precision contract: see dqn3.hip):
//   dqn3w_gather_kernel   the batch drawn (or taken from `idx`) and gathered once from the HBM ring: s, s', a, r, terminal
//   ppo3w_fwd_kernel<..., 2>  target network on s' (forward only): y = r + gamma (1 - terminal) max_a' Qt(s', a')
//   ppo3w_fwd_kernel<..., 3>  online network on s: Huber(delta) on Q(s, a) - y, dZ2
//   ppo3w_bwd_kernel / ppo3w_dw2_kernel, then the reduce (or the two-launch tail with clip + Adam + bf16 re-pack)
// and dqn3w_plan_kernel is plan!: forward + eps-greedy selection for n env instances.
struct D3WRing {
    RingRecs ring;  // record ring (ring_device.h)
    uint64_t total;
    const int64_t* idx;
    uint64_t seed;
    uint32_t draw_ctr;
};

template <int NS>
__global__ __launch_bounds__(256) void dqn3w_gather_kernel(D3WRing rb, P3WArgs g) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= (uint32_t)g.npad) return;
    const bool valid = q < g.bm;
    const int64_t b = valid ? (int64_t)q : 0;
    int64_t fj;
    if (rb.idx) {
        fj = rb.idx[b];
    } else {  // the uniform BatchSampler draw of dqn3.hip: SAMPLER stream, multiply-high
        const u32x4 wd = philox4x32_10(rb.seed, (uint32_t)b, 0, rb.draw_ctr, TAG_SAMPLER);
        const uint64_t xr = ((uint64_t)wd.x << 32) | (uint64_t)wd.y;
        fj = (int64_t)__umul64hi(xr, rb.total);
    }
    const RingTransition rt = ring_load_transition(rb.ring, fj);  // one 64-byte record = one fabric request per sample
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        g.xg[(int64_t)k * g.npad + q] = rt.s[k];
        g.xg2[(int64_t)k * g.npad + q] = rt.sn[k];
    }
    g.sg[(int64_t)g.npad + q] = rt.r;
    g.sg[2 * (int64_t)g.npad + q] = rt.t ? 1.0f : 0.0f;
    g.sg[3 * (int64_t)g.npad + q] = __int_as_float(rt.a);
}

// one net's W2 -> bf16 MFMA B fragments, both orientations (the layout of ppo3w_pack_kernel)
__global__ __launch_bounds__(256) void mlp3w_pack_kernel(const float* __restrict__ params, int ns, uint16_t* __restrict__ pk) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= HW * HW) return;
    const float* W2 = params + HW * ns + HW;
    const int u = q & 7, l = (q >> 3) & 63, f = q >> 9;
    const int t = f % WV, ks = f / WV;
    const int col = 32 * t + (l & 31), kk = 16 * ks + 8 * (l >> 5) + u;
    pk[q] = f32_to_bf16_rne(W2[col + HW * kk]);
    pk[HW * HW + q] = f32_to_bf16_rne(W2[kk + HW * col]);
}

struct RegQW {
    const float* q;
    __device__ __forceinline__ float operator()(int k) const { return q[k]; }
};

constexpr size_t PLANW_LDS = (WV * MAXO * RW + SMALLWW + WV * RW * TPW) * sizeof(float) + (size_t)RW * PW * sizeof(uint16_t);

// plan!(policy, env) of the Q-network: forward of 64-env tiles + eps-greedy on the EXPLORE stream; persistent workgroups (the
// 128 KB of W2 fragments are fetched once per workgroup, not once per tile: at 2^20 envs that is 64 MB instead of 2 GB of L2 traffic)
template <int NS, int NA, int ACT>
__global__ __launch_bounds__(NTW) void dqn3w_plan_kernel(const float* __restrict__ params, const uint16_t* __restrict__ packed,
                                                         const float* __restrict__ obs, int64_t n, double eps, uint64_t seed,
                                                         uint32_t env_id_base, uint32_t step, int32_t* __restrict__ actions,
                                                         float* __restrict__ q_out) {
    extern __shared__ __attribute__((aligned(16))) char smw[];
    float* l_part = reinterpret_cast<float*>(smw);  // [WV][MAXO][RW]
    float* l_w = l_part + WV * MAXO * RW;           // [SMALLWW]
    float* l_t = l_w + SMALLWW;                     // [WV][RW][TPW]
    uint16_t* l_H = reinterpret_cast<uint16_t*>(l_t + WV * RW * TPW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col = 32 * w + r;
    float* l_tw = l_t + w * RW * TPW;
    bf16x8 bw[KSW];
    load_frags_w(packed, w, lane, bw);
    const Mlp3W m = stage_small_w2<NS, NA>(params, l_w, tid);
    const int64_t ntiles = (n + RW - 1) / RW;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t e0 = tile * RW;
    float x[NS];
    {
        int64_t e = e0 + lane;
        if (e >= n) e = n - 1;
#pragma unroll
        for (int i = 0; i < NS; ++i) x[i] = obs[(int64_t)i * n + e];
    }
    __syncthreads();  // small weights staged (first pass); the previous tile's l_H / l_part are free
    {
        uint16_t* dst = l_H + lane * PW + 32 * w;
#pragma unroll
        for (int h8 = 0; h8 < 4; ++h8) {
            float hv[8];
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const int u = 32 * w + 8 * h8 + 4 * q4;
                const float4 b = *reinterpret_cast<const float4*>(m.b1 + u);
                float z[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float4 wv = *reinterpret_cast<const float4*>(m.W1 + u + HW * i);
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
    __syncthreads();
    f32x16 h2[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 16; ++q) h2[rt][q] = 0.0f;
    {
        const uint16_t* ap = l_H + r * PW + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 32 * rt * PW + 16 * ks);
                h2[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], h2[rt], 0, 0, 0);
            }
    }
    const float b2v = m.b2[col];
    wave_lds_fence();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int q = 0; q < 16; ++q) l_tw[(32 * rt + mfma_row(q, kb)) * TPW + r] = act_fwd_t<ACT>(h2[rt][q] + b2v);
    wave_lds_fence();
    {
        float pa[NA];
#pragma unroll
        for (int o = 0; o < NA; ++o) pa[o] = 0.0f;
        const float* hrow = l_tw + lane * TPW;
        const float* w3p = m.W3 + NA * 32 * w;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 v = *reinterpret_cast<const float4*>(hrow + 4 * c4);
            const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 0; o < NA; ++o) pa[o] = fmaf(w3p[o + NA * (4 * c4 + e)], hv[e], pa[o]);
        }
#pragma unroll
        for (int o = 0; o < NA; ++o) l_part[(w * MAXO + o) * RW + lane] = pa[o];
    }
    __syncthreads();
    if (tid < RW && e0 + tid < n) {
        const int64_t e = e0 + tid;
        float q[MAXO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < NA; ++o) {
            float acc = l_part[o * RW + tid];
#pragma unroll
            for (int ww = 1; ww < WV; ++ww) acc += l_part[(ww * MAXO + o) * RW + tid];
            q[o] = acc + m.b3[o];
        }
        if (q_out)
            for (int o = 0; o < NA; ++o) q_out[(int64_t)o * n + e] = q[o];
        if (actions) actions[e] = eps_greedy_select1(RegQW{q}, NoMask{}, NA, eps, false, seed, env_id_base + (uint32_t)e, step);
    }
    }  // tile loop (the barrier at its top separates this pass's l_part reads from the next pass's writes)
}

struct D3WLayout {
    int64_t ntiles, npad, off_xg, off_xg2, off_sg, off_rows, off_frag, off_partS, off_partW, off_loss, off_tail, bytes;
    int nS;
};

static D3WLayout d3w_layout(int ns, int na, int64_t batch) {
    D3WLayout L;
    L.ntiles = (batch + RW - 1) / RW;
    if (L.ntiles < 1) L.ntiles = 1;
    L.npad = L.ntiles * RW;
    L.nS = mlp3w_ns_small(ns, na);
    int64_t o = 0;
    L.off_xg = o;
    o += 4 * L.npad * (int64_t)sizeof(float);
    L.off_xg2 = o;
    o += 4 * L.npad * (int64_t)sizeof(float);
    L.off_sg = o;
    o += 4 * L.npad * (int64_t)sizeof(float);
    L.off_rows = o;
    o += L.npad * HW * (int64_t)sizeof(uint16_t);
    L.off_frag = o;
    o += L.npad * HW * (int64_t)sizeof(uint16_t);
    L.off_partS = o;
    o += (L.ntiles < P3W_ROWS_S ? L.ntiles : P3W_ROWS_S) * (int64_t)L.nS * (int64_t)sizeof(float);
    L.off_partW = o;
    o += (L.ntiles < 256 ? L.ntiles : 256) * (int64_t)HW * HW * (int64_t)sizeof(float);
    L.off_loss = o;
    o += (int64_t)P3W_ROWS_S * 4 * (int64_t)sizeof(float);
    o = (o + 63) & ~(int64_t)63;
    L.off_tail = o;  // zero-initialised by the host (as the 128-wide workspace): Float64 partial sums + departure counter
    o += W3T_BLOCKS * (int64_t)sizeof(double) + 64;
    L.bytes = o + 256;
    return L;
}

int64_t dqn3w_workspace_bytes(int64_t ns, int64_t na, int64_t batch) { return d3w_layout((int)ns, (int)na, batch).bytes; }

int32_t dqn3w_pack(const float* params, int64_t ns, uint16_t* packed, rlhip_stream_t stream) {
    hipLaunchKernelGGL(mlp3w_pack_kernel, dim3(HW * HW / 256), dim3(256), 0, as_stream(stream), params, (int)ns, packed);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t dqn3w_plan(const float* params, const uint16_t* packed, int64_t ns, int64_t na, int32_t act, const float* obs, int64_t n,
                   double eps, uint64_t seed, uint32_t env_id_base, uint32_t step, int32_t* actions, float* q_out,
                   rlhip_stream_t stream) {
    hipStream_t s = as_stream(stream);
    const int64_t pt = (n + RW - 1) / RW;
    const dim3 grid((unsigned)(pt < 512 ? pt : 512));
#define LAUNCH_PW(NS_, NA_, ACT_)                                                                                       \
    do {                                                                                                                \
        static unsigned long long done_ = 0;                                                                                      \
        int32_t rc_ = allow_lds_w(dqn3w_plan_kernel<NS_, NA_, ACT_>, PLANW_LDS, &done_);                                \
        if (rc_) return rc_;                                                                                            \
        hipLaunchKernelGGL((dqn3w_plan_kernel<NS_, NA_, ACT_>), grid, dim3(NTW), PLANW_LDS, s, params, packed, obs, n, eps, \
                           seed, env_id_base, step, actions, q_out);                                                    \
    } while (0)
    if (ns == 4 && na == 2) { if (act == 0) LAUNCH_PW(4, 2, 0); else LAUNCH_PW(4, 2, 1); }
    else if (ns == 2 && na == 3) { if (act == 0) LAUNCH_PW(2, 3, 0); else LAUNCH_PW(2, 3, 1); }
    else { if (act == 0) LAUNCH_PW(3, 3, 0); else LAUNCH_PW(3, 3, 1); }
#undef LAUNCH_PW
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// optimise! state of rlhip_dqn3_update_f32 (NULL: gradient only)
struct D3WApply {
    float *p, *m, *v, *beta_pow, *gn_out;
    uint16_t* packed;
    float grad_scale, clip_norm, lr, b1, b2, eps;
};

int32_t dqn3w_grad(const rlhip_ring* rb, int64_t na, int32_t act, const float* params, const uint16_t* packed,
                   const float* target_params, const uint16_t* target_packed, int64_t batch, const int64_t* idx, float gamma,
                   float huber_delta, uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out, float* loss_out,
                   float* td_out, rlhip_stream_t stream, const D3WApply* apply, const float* isw = nullptr) {
    const int ns = (int)rb->obs_dim;
    RLHIP_REQUIRE(batch <= (int64_t)P3W_MAX_TILES * RW, "batch too large for one launch");
    const D3WLayout L = d3w_layout(ns, (int)na, batch);
    hipStream_t s = as_stream(stream);
    char* ws = (char*)workspace;
    D3WRing r;
    r.ring = {(const uint8_t*)rb->state, rb->capacity, rb->n_env, rb->head_sa};
    r.total = (uint64_t)rb->len_rt * (uint64_t)rb->n_env;
    r.idx = idx;
    r.seed = seed;
    r.draw_ctr = draw_ctr;
    P3WArgs g{};
    g.params = params;
    g.packed = packed;
    g.tparams = target_params;
    g.tpacked = target_packed;
    g.xg = (float*)(ws + L.off_xg);
    g.xg2 = (float*)(ws + L.off_xg2);
    g.sg = (float*)(ws + L.off_sg);
    g.dz_rows = (uint16_t*)(ws + L.off_rows);
    g.dz_frag = (uint16_t*)(ws + L.off_frag);
    g.partS = (float*)(ws + L.off_partS);
    g.partW = (float*)(ws + L.off_partW);
    g.loss_partials = (float*)(ws + L.off_loss);
    g.td_out = td_out;
    g.isw = isw;
    g.frag_stride = 0;
    const int np = (int)mlp3w_np(ns, na);
    g.np_a = np;  // a single net: every parameter index belongs to "net 0"
    g.bm = (uint32_t)batch;
    g.ntiles = (int)L.ntiles;
    g.npad = (int)L.npad;
    g.npS = L.nS;
    g.nS_a = L.nS;
    g.na = (int)na;
    g.wnets = 1;
    g.wa = -1.0f;  // marks the DQN loss line for the reduce kernels
    g.inv_b = 1.0f / (float)batch;
    g.gamma = gamma;
    g.delta = huber_delta;
    const int n_cu = device_cu_count() < P3W_ROWS_S ? device_cu_count() : P3W_ROWS_S;
    const int nrowsS = (int)(L.ntiles < n_cu ? L.ntiles : n_cu);
    const int nsr = (int)(L.ntiles < p3w_rows_w() ? L.ntiles : p3w_rows_w());
    const int gb = (g.npad + 255) / 256;
#define LAUNCH_DW(NS_, NA_, ACT_)                                                                                      \
    do {                                                                                                               \
        static unsigned long long d0_ = 0, d1_ = 0, d3_ = 0;                                                        \
        int32_t rc_;                                                                                                   \
        if ((rc_ = allow_lds_w(ppo3w_fwd_kernel<NS_, NA_, ACT_, 0, 2>, FWDW_LDS, &d0_))) return rc_;                   \
        if ((rc_ = allow_lds_w(ppo3w_fwd_kernel<NS_, NA_, ACT_, 0, 3>, FWDW_LDS, &d1_))) return rc_;                   \
        if ((rc_ = allow_lds_w(ppo3w_dw2_kernel<NS_, ACT_>, DW2W_LDS, &d3_))) return rc_;                              \
        if (RLHIP_W3_DZ_ONCE == 1) g.dz_rows = g.dz_frag; /* the one dZ2 image (rows), read by bwd AND dw2 */                \
        hipLaunchKernelGGL((dqn3w_gather_kernel<NS_>), dim3(gb), dim3(256), 0, s, r, g);                               \
        hipLaunchKernelGGL((ppo3w_fwd_kernel<NS_, NA_, ACT_, 0, 2>), dim3(nrowsS), dim3(NTW), FWDW_LDS, s, g);         \
        hipLaunchKernelGGL((ppo3w_fwd_kernel<NS_, NA_, ACT_, 0, 3>), dim3(nrowsS), dim3(NTW), FWDW_LDS, s, g);         \
        W3_LAUNCH_BWD(NS_, ACT_, 0);                                                                                   \
        hipLaunchKernelGGL((ppo3w_dw2_kernel<NS_, ACT_>), dim3(2 * nsr), dim3(NTW), DW2W_LDS, s, g, 0, nsr);           \
    } while (0)
    if (ns == 4 && na == 2) { if (act == 0) LAUNCH_DW(4, 2, 0); else LAUNCH_DW(4, 2, 1); }
    else if (ns == 2 && na == 3) { if (act == 0) LAUNCH_DW(2, 3, 0); else LAUNCH_DW(2, 3, 1); }
    else { if (act == 0) LAUNCH_DW(3, 3, 0); else LAUNCH_DW(3, 3, 1); }
#undef LAUNCH_DW
    if (apply != nullptr) {
        double* sumsq = (double*)(ws + L.off_tail);
        unsigned int* departed = (unsigned int*)(sumsq + W3T_BLOCKS);
        const int nbt = (int)((np + 255) / 256 < W3T_BLOCKS ? (np + 255) / 256 : W3T_BLOCKS);
        hipLaunchKernelGGL(ppo3w_reduce_sumsq_kernel, dim3(nbt), dim3(1024), 0, s, g.partS, g.partW, g.loss_partials, nrowsS, nsr,
                           g.npS, g.nS_a, np, np, ns, grad_out, loss_out, g.wa, 0.0f, 0.0f, g.inv_b, apply->grad_scale, sumsq, 1);
        hipLaunchKernelGGL(ppo3w_adam_pack_kernel, dim3(nbt), dim3(256), 0, s, apply->p, grad_out, apply->m, apply->v,
                           apply->beta_pow, np, np, ns, apply->grad_scale, apply->clip_norm, apply->lr, apply->b1, apply->b2,
                           apply->eps, (const double*)sumsq, nbt, departed, apply->packed, apply->gn_out);
    } else {
        hipLaunchKernelGGL(ppo3w_reduce_kernel, dim3((np + 63) / 64), dim3(256), 0, s, g.partS, g.partW, g.loss_partials, nrowsS,
                           nsr, g.npS, g.nS_a, np, np, ns, grad_out, loss_out, g.wa, 0.0f, 0.0f, g.inv_b, 1);
    }
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

// flat-argument entry for dqn3.hip (apply_p == NULL: gradient only; otherwise the two-launch optimiser tail follows)
int32_t dqn3w_grad_entry(const rlhip_ring* rb, int64_t na, int32_t act, const float* params, const uint16_t* packed,
                         const float* target_params, const uint16_t* target_packed, int64_t batch, const int64_t* idx,
                         float gamma, float huber_delta, uint64_t seed, uint32_t draw_ctr, void* workspace, float* grad_out,
                         float* loss_out, float* td_out, rlhip_stream_t stream, float* apply_p, uint16_t* apply_packed,
                         float* m, float* v, float* beta_pow, float* gn_out, float grad_scale, float clip_norm, float lr,
                         float b1, float b2, float eps, const float* isw) {
    if (apply_p == nullptr)
        return dqn3w_grad(rb, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta, seed,
                          draw_ctr, workspace, grad_out, loss_out, td_out, stream, nullptr, isw);
    const D3WApply ap{apply_p, m, v, beta_pow, gn_out, apply_packed, grad_scale, clip_norm, lr, b1, b2, eps};
    return dqn3w_grad(rb, na, act, params, packed, target_params, target_packed, batch, idx, gamma, huber_delta, seed, draw_ctr,
                      workspace, grad_out, loss_out, td_out, stream, &ap, isw);
}

}  // namespace rlhip
