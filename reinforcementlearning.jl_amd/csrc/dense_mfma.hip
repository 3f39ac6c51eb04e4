// dense_mfma.hip -- bf16 MFMA path for the GEMM-shaped layers of the actor / critic / Q MLPs:
//     Y = act(X * W + b),  X (B x K) activations, W (K x N) = a Flux Dense weight, f32 accumulate.
//
// Replaces `FluxApproximator.forward` = `A.model(x)` (RLCore/policies/learners/flux_approximator.jl:43) ->
// Flux `Dense` (NNlib dense + bias + activation) for hidden x hidden layers, e.g. the blog's DQN net
// 4 -> 128 -> 128 -> 2 (a_practical_introduction_to_RL.jl/index.html:15126-15128) and BASELINE config 3's
// "actor/critic MLP in bf16 MFMA".  The ns -> h first layers (K <= 4) and h -> nout heads (N <= 3) stay on
// the VALU (mlp_device.h): they are far below an MFMA tile.
//
// Kernel: v_mfma_f32_32x32x16_bf16 (gfx950), one wave = 32 batch rows x 128 output features (4
// accumulators of 32x32 f32, 64 acc registers), 4 waves per workgroup = 128 rows.  Operand layout is
// chosen so that every MFMA fragment is ONE 16-byte global load per lane, no LDS transposition:
//   A fragment (32 x 16): lane l holds X[row = l & 31][k0 + 8 * (l >> 5) .. +7]   -> X row-major bf16
//   B fragment (16 x 32): lane l holds W[k0 + 8 * (l >> 5) .. +7][col = l & 31]   -> W stored TRANSPOSED
//                                                                                    (Wt[n][k], k contiguous)
//   C/D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)           (MI355X guide section 3)
// The 4 waves of a workgroup read identical Wt fragments in lock-step (L1 hits); X is streamed once per
// 128-column block.  Epilogue fuses bias + activation + the bf16 / f32 store.
// Roofline: MFMA (dense bf16 peak ~2.5 PFLOP/s); 2*B*K*N flop per launch.  This first version has no LDS
// staging / software pipelining -- measured numbers in profiles/ and DESIGN.md; it is the parity-checked
// starting point for the tuned kernel, not the end state.
#include "mfma_common.h"

namespace rlhip {

constexpr int NT = 4;  // 32-column blocks per wave (128 output features)

template <int ACT, bool OUT_BF16>
__global__ __launch_bounds__(256) void dense_mfma_kernel(const uint16_t* __restrict__ X,
                                                         const uint16_t* __restrict__ Wt,
                                                         const float* __restrict__ bias, int64_t B, int K, int N,
                                                         void* __restrict__ Yv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int col0 = blockIdx.y * (32 * NT);
    if (row0 >= B) return;
    const int r = lane & 31, kb = lane >> 5;
    const uint16_t* xa = X + (row0 + r) * (int64_t)K + 8 * kb;
    const uint16_t* wb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wb[t] = Wt + (int64_t)(col0 + 32 * t + r) * K + 8 * kb;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(xa + k0);
        bf16x8 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const bf16x8*>(wb[t] + k0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[t], acc[t], 0, 0, 0);
    }
    // epilogue: bias + activation, store.  Lane holds column (col0 + 32 t + (lane & 31)), 16 rows.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = col0 + 32 * t + r;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t row = row0 + (q & 3) + 8 * (q >> 2) + 4 * kb;
            float y = acc[t][q] + bv;
            if (ACT == 0) y = fmaxf(y, 0.0f);
            else if (ACT == 1) y = tanhf(y);
            if (OUT_BF16) reinterpret_cast<uint16_t*>(Yv)[row * N + col] = f32_to_bf16_rne(y);
            else reinterpret_cast<float*>(Yv)[row * N + col] = y;
        }
    }
}

// ---- tiled kernel: X tile staged through LDS, weights in MFMA fragment order, coalesced bf16 output ----------
// One workgroup = 128 batch rows x NB = 32 * NTT output columns (NB = N when N <= 256, so X is read from HBM
// exactly once).  Phase 1: the 128 x K bf16 tile (contiguous in memory: rows are contiguous) is copied to LDS
// with 16 B/lane coalesced loads, row pitch K + 8 elements (odd multiple of 16 B -> conflict-free fragment
// reads).  Phase 2: each wave multiplies its 32 rows by the whole column block; B fragments come from the
// fragment-ordered weight copy (one wave load = 1 KB contiguous, shared by every workgroup -> L1/L2 hits).
// Phase 3: bias + activation; bf16 results go back through the same LDS region and leave as 16 B/lane row-major
// stores (f32 results are stored straight from the accumulator layout: 128 B contiguous per row).
// At the actor/critic shapes (K, N <= 512) the layer is HBM-bound: 2 (K + N) bytes per row against 2 K N flop.
template <int NTT, int ACT, bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void dense_tiled_kernel(const uint16_t* __restrict__ X,
                                                             const uint16_t* __restrict__ Wfrag,
                                                             const float* __restrict__ bias, int64_t B, int K, int N,
                                                             void* __restrict__ Yv) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    uint16_t* tile = reinterpret_cast<uint16_t*>(dsm);
    constexpr int NB = 32 * NTT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int col0 = blockIdx.y * NB;
    const int pitch = K + 8;
    // phase 1: global -> LDS (16-byte chunks, fully coalesced)
    {
        const int k8 = K >> 3;
        const uint4* src = reinterpret_cast<const uint4*>(X + row0 * K);
        const int chunks = 128 * k8;
        for (int c = tid; c < chunks; c += 256) {
            const int row = c / k8, kc = c - row * k8;
            *reinterpret_cast<uint4*>(tile + row * pitch + 8 * kc) = src[c];
        }
    }
    __syncthreads();
    // phase 2: MFMA
    f32x16 acc[NTT];
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
    {
        const uint16_t* ap = tile + (32 * w + r) * pitch + 8 * kb;
        const int nt_all = N >> 5;
        const uint16_t* bp = Wfrag + ((int64_t)(col0 >> 5) * 64 + lane) * 8;
        const int ksteps = K >> 4;
        for (int ks = 0; ks < ksteps; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
            const uint16_t* bk = bp + (int64_t)ks * nt_all * 512;
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(bk + t * 512);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // phase 3: epilogue
    if (OUT_BF16) {
        __syncthreads();  // every wave is done reading the X tile
        const int opitch = NB + 8;
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int col = 32 * t + r;
            const float bv = bias ? bias[col0 + col] : 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float y = acc[t][q] + bv;
                if (ACT == 0) y = fmaxf(y, 0.0f);
                else if (ACT == 1) y = tanhf(y);
                tile[(32 * w + mfma_row(q, kb)) * opitch + col] = f32_to_bf16_rne(y);
            }
        }
        __syncthreads();
        constexpr int c8 = NB >> 3;
        uint16_t* Y = reinterpret_cast<uint16_t*>(Yv);
        for (int c = tid; c < 128 * c8; c += 256) {
            const int row = c / c8, cc = c - row * c8;
            *reinterpret_cast<uint4*>(Y + (row0 + row) * N + col0 + 8 * cc) =
                *reinterpret_cast<const uint4*>(tile + row * opitch + 8 * cc);
        }
    } else {
        float* Y = reinterpret_cast<float*>(Yv);
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int col = col0 + 32 * t + r;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float y = acc[t][q] + bv;
                if (ACT == 0) y = fmaxf(y, 0.0f);
                else if (ACT == 1) y = tanhf(y);
                Y[(row0 + 32 * w + mfma_row(q, kb)) * N + col] = y;
            }
        }
    }
}

// ---- persistent kernel for the 256-wide hidden layer: weights in REGISTERS, X tiles double-buffered in LDS -----------
// The tiled kernel above re-reads the B fragments from L2 in every wave of every workgroup: 4 waves x 128 KB per
// 128-row tile against 64 KB of X -- the L2 -> CU fragment traffic (512 MB per 131072-row launch), not HBM (134 MB), was
// what bounded it (72 us, 1.9 TB/s).  Here a wave owns 64 output columns for ALL 128 rows of a tile, so its B
// fragments are 2 x K/16 registers-resident 16-byte vectors (128 VGPRs at K = 256), loaded once per workgroup
// lifetime; one workgroup per CU walks row tiles blockIdx.x, blockIdx.x + gridDim.x, ...  While the MFMAs of tile i
// run, the global loads of tile i + 1 are in flight (registers -> the other LDS buffer after the epilogue); the
// bf16 result tile leaves through the consumed X buffer as 16 B/lane row-major stores.  LDS per workgroup: 2 x 128 x
// (K + 8) x 2 B = 132 KB of the CU's 160 KB.  A fragments are read 4x (once per wave) from LDS: 256 KB per tile,
// ~0.9 us at 128 B/clk -- below the tile's HBM time (128 KB at 1/256 of ~5 TB/s = 6.5 us).
// RT = 2 (64-row tiles, 66 KB of LDS, <= 256 VGPRs) puts two workgroups on a CU: while one waits for its tile or drains
// its result the other multiplies -- 35.8 -> 25.2 us at batch 131072 (683 TFLOP/s, 5.3 TB/s).
// RT row tiles of 32 rows per workgroup tile: 4 (1 workgroup / CU) or 2 (2 / CU); NC column tiles of 32 per wave: the
// workgroup covers N = 128 NC output columns starting at 128 NC blockIdx.y of the layer's `ntot` (wider layers are split
// over blockIdx.y and re-read X once per column block; K = 512 keeps 256 VGPRs of fragments, hence one workgroup per CU)
template <int KS, int ACT, int RT, int NC>
__global__ __launch_bounds__(256, ((RT == 4 || KS > 16) ? 1 : 2)) void dense_persist_kernel(const uint16_t* __restrict__ X,
                                                               const uint16_t* __restrict__ Wfrag,
                                                               const float* __restrict__ bias, int64_t ntiles,
                                                               int ntot, uint16_t* __restrict__ Y) {
    constexpr int K = 16 * KS, N = 128 * NC, PITCH = K + 8, OPITCH = N + 8;
    constexpr int ROWS = 32 * RT;
    constexpr int BUF = ROWS * (PITCH > OPITCH ? PITCH : OPITCH);
    constexpr int K8 = K / 8;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    uint16_t* const buf0 = reinterpret_cast<uint16_t*>(dsm);
    uint16_t* const buf1 = buf0 + BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col0 = blockIdx.y * N;
    // this wave's 64 columns of W: fragments (ks, tg = 2 w + c), one 16-byte load each, kept for the whole launch
    bf16x8 bw[KS][NC];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int c = 0; c < NC; ++c)
            bw[ks][c] = *reinterpret_cast<const bf16x8*>(Wfrag + ((int64_t)(ks * (ntot / 32) + (col0 >> 5) + NC * w + c) * 64 + lane) * 8);
    float bv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) bv[c] = bias ? bias[col0 + 32 * NC * w + 32 * c + r] : 0.0f;
    constexpr int NPRE = KS * RT / 4;  // ROWS x K bf16 = ROWS K8 16-byte chunks = NPRE per thread
    nt_u32x4 pre[NPRE];
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    {
        const nt_u32x4* src = reinterpret_cast<const nt_u32x4*>(X + tile * ROWS * K);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) pre[i] = nt_load16(src + tid + 256 * i);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int c = tid + 256 * i, row = c / K8, kc = c - row * K8;
            *reinterpret_cast<nt_u32x4*>(buf0 + row * PITCH + 8 * kc) = pre[i];
        }
    }
    __syncthreads();
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        uint16_t* const cur = (it & 1) ? buf1 : buf0;
        uint16_t* const nxt = (it & 1) ? buf0 : buf1;
        const int64_t next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) {  // in flight during the MFMA phase
            const nt_u32x4* src = reinterpret_cast<const nt_u32x4*>(X + next * ROWS * K);
#pragma unroll
            for (int i = 0; i < NPRE; ++i) pre[i] = nt_load16(src + tid + 256 * i);
        }
        f32x16 acc[RT][NC];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[rt][c][q] = 0.0f;
        const uint16_t* ap = cur + r * PITCH + 8 * kb;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 32 * rt * PITCH + 16 * ks);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks][c], acc[rt][c], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave is done reading the X tile: it becomes the output tile
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float y = acc[rt][c][q] + bv[c];
                    if (ACT == 0) y = fmaxf(y, 0.0f);
                    else if (ACT == 1) y = tanhf(y);
                    cur[(32 * rt + mfma_row(q, kb)) * OPITCH + 32 * NC * w + 32 * c + r] = f32_to_bf16_rne(y);
                }
        __syncthreads();
        {
            uint16_t* dst = Y + tile * ROWS * (int64_t)ntot + col0;
#pragma unroll
            for (int i = 0; i < 2 * RT * NC; ++i) {  // ROWS x N bf16 = ROWS N / 8 chunks
                const int c = tid + 256 * i, row = c / (N / 8), cc = c - row * (N / 8);
                nt_store16(dst + (int64_t)row * ntot + 8 * cc, *reinterpret_cast<const nt_u32x4*>(cur + row * OPITCH + 8 * cc));
            }
        }
        if (has_next) {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int c = tid + 256 * i, row = c / K8, kc = c - row * K8;
                *reinterpret_cast<nt_u32x4*>(nxt + row * PITCH + 8 * kc) = pre[i];
            }
        }
        __syncthreads();
    }
}

// Wt[n][k] (row-major bf16, k contiguous) -> MFMA B-fragment order: fragment (ks, tg) = 64 lanes x 8 elements,
// lane l holds Wt[n = 32 tg + (l & 31)][k = 16 ks + 8 (l >> 5) + u]
__global__ __launch_bounds__(256) void frag_weight_kernel(const uint16_t* __restrict__ wt, int K, int N,
                                                          uint16_t* __restrict__ wf) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)K * N) return;
    int u = (int)(q & 7), l = (int)((q >> 3) & 63);
    int64_t f = q >> 9;
    int nt_all = N >> 5;
    int tg = (int)(f % nt_all), ks = (int)(f / nt_all);
    int n = 32 * tg + (l & 31), k = 16 * ks + 8 * (l >> 5) + u;
    wf[q] = wt[(int64_t)n * K + k];
}

// f32 SoA activations (K x B, batch contiguous) -> bf16 rows (B x Kpad), zero padded to Kpad
__global__ __launch_bounds__(256) void soa_to_bf16_rows_kernel(const float* __restrict__ x, int64_t B, int K,
                                                               int Kpad, uint16_t* __restrict__ out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int k = 0; k < Kpad; ++k) out[b * Kpad + k] = (k < K) ? f32_to_bf16_rne(x[(int64_t)k * B + b]) : (uint16_t)0;
}

// Flux Dense weight (N out x K in, column-major: w[o + N * i]) -> Wt[n][k] bf16 (k contiguous), zero padded
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ w, int K, int N, int Kpad,
                                                               int Npad, uint16_t* __restrict__ wt) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)Npad * Kpad) return;
    int n = (int)(q / Kpad), k = (int)(q % Kpad);
    wt[q] = (n < N && k < K) ? f32_to_bf16_rne(w[n + (int64_t)N * k]) : (uint16_t)0;
}

__global__ __launch_bounds__(256) void bf16_rows_to_f32_soa_kernel(const uint16_t* __restrict__ y, int64_t B, int N,
                                                                   int ld, float* __restrict__ out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int n = 0; n < N; ++n) out[(int64_t)n * B + b] = bf16_to_f32(y[b * ld + n]);
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_dense_bf16_forward(const uint16_t* x_rows, const uint16_t* wt, const float* bias, int32_t act,
                                 int64_t batch, int32_t k, int32_t n, void* y_rows, int32_t y_is_bf16,
                                 rlhip_stream_t stream) {
    RLHIP_REQUIRE(x_rows && wt && y_rows, "NULL array");
    RLHIP_REQUIRE(batch >= 32 && batch % 128 == 0, "batch must be a multiple of 128 (pad the batch)");
    RLHIP_REQUIRE(k >= 16 && k % 16 == 0, "K must be a multiple of 16 (zero-pad)");
    RLHIP_REQUIRE(n >= 128 && n % 128 == 0, "N must be a multiple of 128 (zero-pad)");
    RLHIP_REQUIRE(act >= 0 && act <= 2, "act: 0 relu, 1 tanh, 2 identity");
    RLHIP_REQUIRE((((uintptr_t)x_rows | (uintptr_t)wt) & 15) == 0, "operands must be 16-byte aligned");
    dim3 grid((unsigned)(batch / 128), (unsigned)(n / 128));
    hipStream_t s = as_stream(stream);
#define LAUNCH_D(A_, O_) \
    hipLaunchKernelGGL((dense_mfma_kernel<A_, O_>), grid, dim3(256), 0, s, x_rows, wt, bias, batch, k, n, y_rows)
    if (y_is_bf16) {
        if (act == 0) LAUNCH_D(0, true);
        else if (act == 1) LAUNCH_D(1, true);
        else LAUNCH_D(2, true);
    } else {
        if (act == 0) LAUNCH_D(0, false);
        else if (act == 1) LAUNCH_D(1, false);
        else LAUNCH_D(2, false);
    }
#undef LAUNCH_D
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dense_frag_weight_bf16(const uint16_t* wt, int32_t k, int32_t n, uint16_t* w_frag,
                                     rlhip_stream_t stream) {
    RLHIP_REQUIRE(wt && w_frag, "NULL array");
    RLHIP_REQUIRE(k >= 16 && k % 16 == 0 && n >= 32 && n % 32 == 0, "K must be a multiple of 16, N of 32 (zero-pad)");
    int64_t total = (int64_t)k * n;
    hipLaunchKernelGGL(frag_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), wt, k,
                       n, w_frag);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dense_bf16_forward_tiled(const uint16_t* x_rows, const uint16_t* w_frag, const float* bias, int32_t act,
                                       int64_t batch, int32_t k, int32_t n, void* y_rows, int32_t y_is_bf16,
                                       rlhip_stream_t stream) {
    RLHIP_REQUIRE(x_rows && w_frag && y_rows, "NULL array");
    RLHIP_REQUIRE(batch >= 128 && batch % 128 == 0, "batch must be a multiple of 128 (pad the batch)");
    RLHIP_REQUIRE(k >= 16 && k % 16 == 0 && k <= 512, "K must be a multiple of 16, <= 512 (zero-pad)");
    RLHIP_REQUIRE(n >= 128 && n % 128 == 0, "N must be a multiple of 128 (zero-pad)");
    RLHIP_REQUIRE(act >= 0 && act <= 2, "act: 0 relu, 1 tanh, 2 identity");
    RLHIP_REQUIRE((((uintptr_t)x_rows | (uintptr_t)w_frag | (uintptr_t)y_rows) & 15) == 0,
                  "operands must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    if ((n % 256 == 0 || n == 128) && n <= 1024 && y_is_bf16 && (k == 512 || k == 256 || k == 128) &&
        !RLHIP_ENV_FLAG("RLHIP_DENSE_NO_PERSIST")) {
        // the 256-wide hidden layer: weights in registers, one workgroup per CU walking the row tiles
        // 64-row tiles, two workgroups per CU (they interleave their load / MFMA / store phases) unless asked otherwise
        const bool rows64 = k == 512 || (!RLHIP_ENV_FLAG("RLHIP_DENSE_ROWS128") && batch % 64 == 0);
        const int rows = rows64 ? 64 : 128;
        const int64_t ntiles = batch / rows;
        const int nw = n == 128 ? 128 : 256;  // columns per workgroup
        const int ncb = n / nw;               // column blocks (blockIdx.y)
        const int64_t cap = ((rows64 && k <= 256) ? 512 : 256) / ncb;
        const dim3 grid((unsigned)(ntiles < cap ? ntiles : (cap < 1 ? 1 : cap)), (unsigned)ncb);
        const size_t lds = (size_t)2 * rows * (size_t)((k > nw ? k : nw) + 8) * sizeof(uint16_t);
#define LAUNCH_P(KS_, A_, RT_, NC_)                                                                                     \
    do {                                                                                                             \
        static bool set_ = false;                                                                                    \
        if (!set_) {                                                                                                 \
            RLHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_persist_kernel<KS_, A_, RT_, NC_>),   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
            set_ = true;                                                                                             \
        }                                                                                                            \
        hipLaunchKernelGGL((dense_persist_kernel<KS_, A_, RT_, NC_>), grid, dim3(256), lds, s, x_rows, w_frag, bias,       \
                           ntiles, n, (uint16_t*)y_rows);                                                               \
    } while (0)
#define LAUNCH_PR(KS_, A_)                                \
    do {                                                  \
        if (n != 128) {                                   \
            if (rows64) LAUNCH_P(KS_, A_, 2, 2);          \
            else LAUNCH_P(KS_, A_, 4, 2);                 \
        } else {                                          \
            if (rows64) LAUNCH_P(KS_, A_, 2, 1);          \
            else LAUNCH_P(KS_, A_, 4, 1);                 \
        }                                                 \
    } while (0)
#define LAUNCH_PA(KS_)                        \
    do {                                      \
        if (act == 0) LAUNCH_PR(KS_, 0);      \
        else if (act == 1) LAUNCH_PR(KS_, 1); \
        else LAUNCH_PR(KS_, 2);               \
    } while (0)
        if (k == 512) {  // 256 VGPRs of fragments: 64-row tiles only, one workgroup per CU
            if (n != 128) {
                if (act == 0) LAUNCH_P(32, 0, 2, 2);
                else if (act == 1) LAUNCH_P(32, 1, 2, 2);
                else LAUNCH_P(32, 2, 2, 2);
            } else {
                if (act == 0) LAUNCH_P(32, 0, 2, 1);
                else if (act == 1) LAUNCH_P(32, 1, 2, 1);
                else LAUNCH_P(32, 2, 2, 1);
            }
        } else if (k == 256) LAUNCH_PA(16);
        else LAUNCH_PA(8);
#undef LAUNCH_PA
#undef LAUNCH_PR
#undef LAUNCH_P
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    const bool wide = (n % 256 == 0);  // 256-column blocks: X is read once when N = 256
    const int nb = wide ? 256 : 128;
    size_t lds = (size_t)128 * (size_t)((k > nb ? k : nb) + 8) * sizeof(uint16_t);
    dim3 grid((unsigned)(batch / 128), (unsigned)(n / nb));
#define LAUNCH_T(NTT_, A_, O_)                                                                                   \
    do {                                                                                                         \
        static size_t allowed_ = 0;                                                                              \
        if (lds > allowed_) {                                                                                    \
            RLHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_tiled_kernel<NTT_, A_, O_>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
            allowed_ = lds;                                                                                      \
        }                                                                                                        \
        hipLaunchKernelGGL((dense_tiled_kernel<NTT_, A_, O_>), grid, dim3(256), lds, s, x_rows, w_frag, bias,    \
                           batch, k, n, y_rows);                                                                 \
    } while (0)
#define LAUNCH_TA(NTT_, O_)              \
    do {                                 \
        if (act == 0) LAUNCH_T(NTT_, 0, O_); \
        else if (act == 1) LAUNCH_T(NTT_, 1, O_); \
        else LAUNCH_T(NTT_, 2, O_);      \
    } while (0)
    if (wide) { if (y_is_bf16) LAUNCH_TA(8, true); else LAUNCH_TA(8, false); }
    else { if (y_is_bf16) LAUNCH_TA(4, true); else LAUNCH_TA(4, false); }
#undef LAUNCH_TA
#undef LAUNCH_T
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_soa_f32_to_bf16_rows(const float* x_soa, int64_t batch, int32_t k, int32_t k_pad, uint16_t* out_rows,
                                   rlhip_stream_t stream) {
    RLHIP_REQUIRE(x_soa && out_rows && batch >= 0 && k >= 1 && k_pad >= k, "bad arguments");
    if (batch == 0) return RLHIP_OK;
    hipLaunchKernelGGL(soa_to_bf16_rows_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x_soa, batch, k, k_pad, out_rows);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_bf16_rows_to_soa_f32(const uint16_t* y_rows, int64_t batch, int32_t n, int32_t ld, float* out_soa,
                                   rlhip_stream_t stream) {
    RLHIP_REQUIRE(y_rows && out_soa && batch >= 0 && n >= 1 && ld >= n, "bad arguments");
    if (batch == 0) return RLHIP_OK;
    hipLaunchKernelGGL(bf16_rows_to_f32_soa_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0,
                       as_stream(stream), y_rows, batch, n, ld, out_soa);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_dense_pack_weight_bf16(const float* w_flux, int32_t k, int32_t n, int32_t k_pad, int32_t n_pad,
                                     uint16_t* wt, rlhip_stream_t stream) {
    RLHIP_REQUIRE(w_flux && wt && k >= 1 && n >= 1 && k_pad >= k && n_pad >= n, "bad arguments");
    int64_t total = (int64_t)n_pad * k_pad;
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w_flux, k, n, k_pad, n_pad, wt);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // extern "C"
