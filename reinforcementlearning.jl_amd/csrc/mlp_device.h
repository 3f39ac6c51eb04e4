// mlp_device.h -- per-lane pieces of Chain(Dense(ns, h, act), Dense(h, nout)) (device inline).
//
// The actor / critic / Q networks of the named configs are one-hidden-layer MLPs with ns <= 4 inputs
// and <= 4 outputs (blog a_practical_introduction_to_RL.jl/index.html:15257-15287; BASELINE.md
// configs 2-4).  K = ns <= 4 and N = nout <= 4 are far below an MFMA tile (K = 16 / N = 32), so these
// layers run on the VALU with the weights held in registers ("wide" variant: L lanes cooperate on
// one env, each owning HPL = h / L hidden units) or in SGPRs via scalar loads ("scalar" variant: one
// lane per env, every lane of the wave walks the same hidden unit, so W is wave-uniform).
// MFMA is reserved for layers with a real GEMM shape (hidden x hidden).
//
// Flat parameter layout = Flux.destructure order: W1 (h x ns col-major) | b1 (h) | W2 (nout x h
// col-major) | b2 (nout).
#pragma once
#include "common.h"

namespace rlhip {

constexpr int MAXO = 4;  // max outputs handled by the fused kernels

__host__ __device__ __forceinline__ int64_t mlp2_nparams(int64_t n_in, int64_t h, int64_t n_out) {
    return h * n_in + h + n_out * h + n_out;
}

__device__ __forceinline__ float act_fwd(int act, float z) { return act == 0 ? fmaxf(z, 0.0f) : tanhf(z); }
__device__ __forceinline__ float act_bwd(int act, float z, float hv) {
    return act == 0 ? (z > 0.0f ? 1.0f : 0.0f) : (1.0f - hv * hv);
}
// compile-time activation (hot loops): a run-time `act` puts a branch and a tanh body inside every
// unrolled hidden-unit iteration and blocks software pipelining (measured: 280 cycles per unit).
template <int ACT>
__device__ __forceinline__ float act_fwd_t(float z) {
    return ACT == 0 ? fmaxf(z, 0.0f) : tanhf(z);
}
template <int ACT>
__device__ __forceinline__ float act_bwd_t(float z, float hv) {
    return ACT == 0 ? (z > 0.0f ? 1.0f : 0.0f) : (1.0f - hv * hv);
}

// ---- wide variant: weights of this lane's HPL hidden units in registers -------------------------
template <int NS, int HPL>
struct NetRegs {
    float w1[HPL][NS];
    float b1[HPL];
    float w2[HPL][MAXO];
    float b2[MAXO];
};

// lane `sub` of an L-lane group owns hidden units j = sub + L * m
template <int NS, int HPL>
__device__ __forceinline__ void load_net(NetRegs<NS, HPL>& r, const float* __restrict__ p, int h, int nout,
                                         int sub, int L) {
    const float* W1 = p;
    const float* b1 = W1 + h * NS;
    const float* W2 = b1 + h;
    const float* b2 = W2 + nout * h;
#pragma unroll
    for (int m = 0; m < HPL; ++m) {
        int j = sub + L * m;
#pragma unroll
        for (int k = 0; k < NS; ++k) r.w1[m][k] = W1[j + h * k];
        r.b1[m] = b1[j];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) r.w2[m][o] = (o < nout) ? W2[o + nout * j] : 0.0f;
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o) r.b2[o] = (o < nout) ? b2[o] : 0.0f;
}

// out[o] = b2[o] + sum_j W2[o,j] * act(b1[j] + sum_k W1[j,k] x[k]); partial sums per lane (ascending m),
// then an xor-butterfly over the L lanes of the group (every lane ends with the identical total).
// NO: outputs actually evaluated (the net has <= NO outputs; the rest of out[] is b2 = 0 either way)
template <int NS, int HPL, int L, int ACT, int NO = MAXO>
__device__ __forceinline__ void net_forward(const NetRegs<NS, HPL>& r, const float x[NS], float out[MAXO]) {
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int m = 0; m < HPL; ++m) {
        float z = r.b1[m];
#pragma unroll
        for (int k = 0; k < NS; ++k) z = fmaf(r.w1[m][k], x[k], z);
        float hv = act_fwd_t<ACT>(z);
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = fmaf(r.w2[m][o], hv, acc[o]);
    }
    // the L = 4 / 8 / 16 lanes of a group sit inside one DPP row: row-local adds, no LDS crossbar
    static_assert(L == 4 || L == 8 || L == 16 || L == 32, "lane groups: a DPP row, or two rows joined by a swizzle");
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = group_sum_dpp<L>(acc[o]);
#pragma unroll
    for (int o = 0; o < MAXO; ++o) out[o] = acc[o] + r.b2[o];
}

// ---- scalar variant: one lane per sample, hidden units walked in order (wave-uniform weights) ----
// Same accumulation order as the CPU oracle: out = b2; for j ascending: out = fma(W2[:,j], h_j, out).
template <int NS, int ACT>
__device__ __forceinline__ void net_forward_scalar(const float* __restrict__ p, int h, int nout,
                                                   const float x[NS], float out[MAXO]) {
    const float* W1 = p;
    const float* b1 = W1 + h * NS;
    const float* W2 = b1 + h;
    const float* b2 = W2 + nout * h;
#pragma unroll
    for (int o = 0; o < MAXO; ++o) out[o] = (o < nout) ? b2[o] : 0.0f;
    for (int j = 0; j < h; ++j) {
        float z = b1[j];
#pragma unroll
        for (int k = 0; k < NS; ++k) z = fmaf(W1[j + h * k], x[k], z);
        float hv = act_fwd_t<ACT>(z);
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
            if (o < nout) out[o] = fmaf(W2[o + nout * j], hv, out[o]);
    }
}

}  // namespace rlhip
