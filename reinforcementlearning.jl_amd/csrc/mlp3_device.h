// mlp3_device.h -- device building blocks of the three-layer network ns -> 128 -> 128 -> nout whose hidden x hidden
// layer runs on v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate, f32 master weights): tile-wise forward
// (layer 1 on the VALU into bf16 LDS tiles, layer 2 on the MFMA, head on the VALU with DPP reductions) and the
// tile-wise backward (dW3/db3/db2 on the VALU, dH1 = dZ2 W2 and dW2 = dZ2^T H1 on the MFMA, dW1/db1 on the VALU).
// Shared by the DQN path (dqn3.hip) and the PPO actor-critic path (ppo3.hip).  See dqn3.hip for the design notes.
#pragma once
#include "mfma_common.h"
#include "mlp_device.h"

namespace rlhip {

constexpr int H3 = 128;   // hidden width of the MFMA path
constexpr int TR = 128;   // samples per tile (4 waves x 32 rows)
constexpr int LDH = 136;  // LDS row pitch in bf16 elements (272 B)
constexpr int TILE_ELEMS = TR * LDH;

#ifndef D3_STAMP
#define D3_STAMP(k) \
    do {            \
    } while (0)
#endif

__host__ __device__ __forceinline__ int64_t mlp3_nparams(int64_t ns, int64_t h, int64_t na) {
    return h * ns + h + h * h + h + na * h + na;
}

struct Mlp3 {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
};
__host__ __device__ __forceinline__ Mlp3 mlp3_view(const float* p, int ns, int na) {
    Mlp3 v;
    v.W1 = p;
    v.b1 = v.W1 + H3 * ns;
    v.W2 = v.b1 + H3;
    v.b2 = v.W2 + H3 * H3;
    v.W3 = v.b2 + H3;
    v.b3 = v.W3 + na * H3;
    return v;
}

// The small f32 tensors of one net (everything except W2) staged in LDS: W1 | b1 | b2 | W3 | b3, i.e. the two
// contiguous parameter ranges around W2.  Global-memory round trips (~2 us each on a cold L2) were on the critical
// path of every phase of the single-tile latency; LDS reads are ~30x closer.
constexpr int SMALLW = H3 * 4 + H3 + H3 + MAXO * H3 + MAXO + 4;  // floats reserved per net (NS <= 4, na <= MAXO)

__device__ __forceinline__ Mlp3 stage_small_weights(const float* __restrict__ p, int ns, int na, float* l_w, int tid) {
    const int n1 = H3 * ns + H3;          // W1 | b1
    const int n2 = H3 + na * H3 + na;     // b2 | W3 | b3
    const float* p2 = p + n1 + H3 * H3;
    for (int i = tid; i < n1; i += 256) l_w[i] = p[i];
    for (int i = tid; i < n2; i += 256) l_w[n1 + i] = p2[i];
    Mlp3 v;
    v.W1 = l_w;
    v.b1 = l_w + H3 * ns;
    v.W2 = nullptr;
    v.b2 = l_w + n1;
    v.W3 = v.b2 + H3;
    v.b3 = v.W3 + na * H3;
    return v;
}

// acc[t] += A(32 x 128) * B(128 x 128)^T ; A element (row, kk) at A[row * lda + kk]; the wave's 32 A rows start
// at A.  B either row-major with pitch ldb (element (col, kk) at B[col * ldb + kk], LDS tiles) or, BFRAG = true,
// pre-packed in MFMA fragment order (global weights): fragment f = (k0 / 16) * 4 + t is 64 lanes x 16 bytes, so one
// wave load is 1 KB contiguous instead of 64 cache lines.
template <bool BFRAG>
__device__ __forceinline__ void gemm_slab(const uint16_t* A, int lda, const uint16_t* B, int ldb, f32x16 (&acc)[4],
                                          int lane) {
    const int r = lane & 31, kb = lane >> 5;
    const uint16_t* ap = A + r * lda + 8 * kb;
    const uint16_t* bp = BFRAG ? (B + lane * 8) : (B + r * ldb + 8 * kb);
#pragma unroll
    for (int k0 = 0; k0 < H3; k0 += 16) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + k0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 b = BFRAG ? *reinterpret_cast<const bf16x8*>(bp + ((k0 / 16) * 4 + t) * 512)
                                   : *reinterpret_cast<const bf16x8*>(bp + (32 * t) * ldb + k0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
    }
}

// (dpp_add / reduce16_dpp / swap16_add: common.h)
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
}

__device__ __forceinline__ uint4 pack8_bf16(const float (&v)[8]) {
    uint4 o;
    o.x = pack2_bf16(v[0], v[1]);
    o.y = pack2_bf16(v[2], v[3]);
    o.z = pack2_bf16(v[4], v[5]);
    o.w = pack2_bf16(v[6], v[7]);
    return o;
}

// h1 = act(b1 + W1 x) for the whole tile, written as bf16 in [row][k] layout (dst_rk) and, when dst_kr is
// non-NULL, also in [k][row] layout.  lx: f32 [NS][TR] in LDS.  Same fmaf chain as mlp2 / the oracle.
template <int NS, int ACT>
__device__ __forceinline__ void layer1_to_lds(const Mlp3& m, const float* lx, uint16_t* dst_rk, uint16_t* dst_kr,
                                              int tid) {
    {
        const int g = tid & 15, r0 = tid >> 4;  // this thread: units 8g..8g+7, rows r0 + 16 it
        float w1[8][NS], bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bb[u] = m.b1[8 * g + u];
#pragma unroll
            for (int i = 0; i < NS; ++i) w1[u][i] = m.W1[8 * g + u + H3 * i];
        }
#pragma unroll
        for (int it = 0; it < TR / 16; ++it) {
            const int row = r0 + 16 * it;
            float x[NS], hv[8];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = lx[i * TR + row];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float z = bb[u];
#pragma unroll
                for (int i = 0; i < NS; ++i) z = fmaf(w1[u][i], x[i], z);
                hv[u] = act_fwd_t<ACT>(z);
            }
            *reinterpret_cast<uint4*>(dst_rk + row * LDH + 8 * g) = pack8_bf16(hv);
        }
    }
    if (dst_kr) {
        const int rg = tid & 15, k0 = tid >> 4;  // this thread: rows 8rg..8rg+7, units k0 + 16 it
        float x[8][NS];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NS; ++i) x[u][i] = lx[i * TR + 8 * rg + u];
#pragma unroll
        for (int it = 0; it < H3 / 16; ++it) {
            const int k = k0 + 16 * it;
            float w1[NS], hv[8];
            const float bb = m.b1[k];
#pragma unroll
            for (int i = 0; i < NS; ++i) w1[i] = m.W1[k + H3 * i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float z = bb;
#pragma unroll
                for (int i = 0; i < NS; ++i) z = fmaf(w1[i], x[u][i], z);
                hv[u] = act_fwd_t<ACT>(z);
            }
            *reinterpret_cast<uint4*>(dst_kr + k * LDH + 8 * rg) = pack8_bf16(hv);
        }
    }
}

// hidden layer 2 (MFMA) + bias + activation for this wave's 32 rows; h2[t][q] in the MFMA D layout
template <int ACT>
__device__ __forceinline__ void layer2(const uint16_t* l_h1rk, const uint16_t* w2jk, const float* b2, int w, int lane,
                                       f32x16 (&h2)[4]) {
    zero_acc(h2);
    gemm_slab<true>(l_h1rk + 32 * w * LDH, LDH, w2jk, H3, h2, lane);
    const int r = lane & 31;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bv = b2[r + 32 * t];
#pragma unroll
        for (int q = 0; q < 16; ++q) h2[t][q] = act_fwd_t<ACT>(h2[t][q] + bv);
    }
}

// the same layer with the 32 B fragments already in registers (fetched at kernel entry, so that their L2 round trip
// overlaps the first layer instead of following the barrier after it): identical MFMA order, identical bits
__device__ __forceinline__ void load_w2_fragments(const uint16_t* __restrict__ w2jk, int lane, bf16x8 (&bw)[H3 / 16][4]) {
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) bw[ks][t] = *reinterpret_cast<const bf16x8*>(w2jk + lane * 8 + (ks * 4 + t) * 512);
}
template <int ACT>
__device__ __forceinline__ void layer2_regs(const uint16_t* l_h1rk, const bf16x8 (&bw)[H3 / 16][4], const float* b2, int w,
                                            int lane, f32x16 (&h2)[4]) {
    zero_acc(h2);
    const int r = lane & 31, kb = lane >> 5;
    const uint16_t* ap = l_h1rk + 32 * w * LDH + r * LDH + 8 * kb;
#pragma unroll
    for (int ks = 0; ks < H3 / 16; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
#pragma unroll
        for (int t = 0; t < 4; ++t) h2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks][t], h2[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bv = b2[r + 32 * t];
#pragma unroll
        for (int q = 0; q < 16; ++q) h2[t][q] = act_fwd_t<ACT>(h2[t][q] + bv);
    }
}

// head: q[o] = b3[o] + sum_j W3[o, j] h2[j] for the wave's 32 rows -> l_q[o][row]   (f32 [MAXO][TR])
template <int NA>
__device__ __forceinline__ void head_to_lds(const Mlp3& m, const f32x16 (&h2)[4], int w, int lane, float* l_q) {
    constexpr int na = NA;
    const int r = lane & 31, kb = lane >> 5;
    float w3[MAXO][4];
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
#pragma unroll
        for (int t = 0; t < 4; ++t) w3[o][t] = (o < na) ? m.W3[o + na * (r + 32 * t)] : 0.0f;
    float b3[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) b3[o] = (o < na) ? m.b3[o] : 0.0f;
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += 4) {
        float p[4][MAXO];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int o = 0; o < MAXO; ++o) {
                p[qq][o] = 0.0f;
                if (o < na) {
                    float a = w3[o][0] * h2[0][q0 + qq];
#pragma unroll
                    for (int t = 1; t < 4; ++t) a = fmaf(w3[o][t], h2[t][q0 + qq], a);
                    p[qq][o] = reduce16_dpp(a);
                }
            }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < na) p[qq][o] = swap16_add(p[qq][o]);
        if (r == 0) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int row = 32 * w + mfma_row(q0 + qq, kb);
#pragma unroll
                for (int o = 0; o < MAXO; ++o)
                    if (o < na) l_q[o * TR + row] = p[qq][o] + b3[o];
            }
        }
    }
}

// column sums held per lane (col = r + 32 t) for NV quantities -> l_red[w][v][col], then summed over the 4 waves
// by threads 0..127 in a fixed order.
template <int NV>
__device__ __forceinline__ void reduce_cols_to_lds(float (&acc)[NV][4], float* l_red, int w, int lane) {
    const int r = lane & 31, kb = lane >> 5;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s = acc[v][t] + __shfl_xor(acc[v][t], 32, 64);
            if (kb == 0) l_red[(w * NV + v) * H3 + r + 32 * t] = s;
        }
}
template <int NV>
__device__ __forceinline__ float sum_waves(const float* l_red, int v, int c) {
    return ((l_red[(0 * NV + v) * H3 + c] + l_red[(1 * NV + v) * H3 + c]) + l_red[(2 * NV + v) * H3 + c]) +
           l_red[(3 * NV + v) * H3 + c];
}

// Backward pass of one 128-sample tile through one network, given dL/d(outputs) per sample in l_dq ([MAXO][TR], f32)
// and the hidden activations h2 of THIS wave's 32 rows still in registers (MFMA D layout).  Writes this workgroup's
// partial gradient of every tensor except b3 (the caller sums l_dq) to `out` (parameter layout of the net).
// LDS: l_A (free; receives dZ2 [r][j]), l_B = H1^T [k][r] written by layer1_to_lds, l_C (free; receives dZ2^T [j][r]),
// l_red [4][5][H3].  Contains workgroup barriers: every thread of the workgroup must call it.
template <int NS, int NA, int ACT>
__device__ __forceinline__ void mlp3_backward_tile(const Mlp3& m, const uint16_t* __restrict__ w2kj,
                                                   const f32x16 (&h2)[4], const float* l_x, const float* l_dq,
                                                   float* l_red, uint16_t* l_A, const uint16_t* l_B, uint16_t* l_C,
                                                   float* __restrict__ out, int tid) {
    constexpr int na = NA;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int oW1 = 0, ob1 = H3 * NS, oW2 = ob1 + H3, ob2 = oW2 + H3 * H3, oW3 = ob2 + H3;
    // ---- head backward in the MFMA D layout: dW3, dh2 -> dz2 (f32), db2; dz2 -> bf16 tiles [r][j] and [j][r] ----
    {
        float w3[MAXO][4];
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
#pragma unroll
            for (int t = 0; t < 4; ++t) w3[o][t] = (o < na) ? m.W3[o + na * (r + 32 * t)] : 0.0f;
        float acc[MAXO + 1][4];  // [0] = db2, [1 + o] = dW3[o]
#pragma unroll
        for (int v = 0; v <= MAXO; ++v)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[v][t] = 0.0f;
        uint16_t pk[4][4];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 32 * w + mfma_row(q, kb);
            float dqv[MAXO];
#pragma unroll
            for (int o = 0; o < MAXO; ++o) dqv[o] = (o < na) ? l_dq[o * TR + row] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float hv = h2[t][q];
                float dh = 0.0f;
#pragma unroll
                for (int o = 0; o < MAXO; ++o)
                    if (o < na) {
                        acc[1 + o][t] = fmaf(dqv[o], hv, acc[1 + o][t]);
                        dh = fmaf(dqv[o], w3[o][t], dh);
                    }
                const float dz = dh * act_bwd_t<ACT>(hv, hv);  // relu: h2 > 0 <=> z2 > 0
                acc[0][t] += dz;
                const uint16_t dzb = f32_to_bf16_rne(dz);
                l_A[row * LDH + r + 32 * t] = dzb;
                pk[t][q & 3] = dzb;
                if ((q & 3) == 3) {
                    uint2 v2;
                    v2.x = (uint32_t)pk[t][0] | ((uint32_t)pk[t][1] << 16);
                    v2.y = (uint32_t)pk[t][2] | ((uint32_t)pk[t][3] << 16);  // (already bf16 bits)
                    *reinterpret_cast<uint2*>(l_C + (r + 32 * t) * LDH + 32 * w + 8 * (q >> 2) + 4 * kb) = v2;
                }
            }
            // keep the fully unrolled rows in program order: without this the scheduler hoists every LDS read
            // of the 16 rows to the top, runs out of VGPRs and spills into AGPRs next to the live accumulators
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        reduce_cols_to_lds<MAXO + 1>(acc, l_red, w, lane);
    }
    __syncthreads();
    if (tid < H3) {
        store_wt(&out[ob2 + tid], sum_waves<MAXO + 1>(l_red, 0, tid));
        for (int o = 0; o < na; ++o) store_wt(&out[oW3 + o + na * tid], sum_waves<MAXO + 1>(l_red, 1 + o, tid));
    }
    __syncthreads();  // l_red is reused below

    // ---- dH1 = dZ2 * W2 (MFMA), dz1 = dH1 * act'(z1), dW1 / db1 (f32) ----
    {
        f32x16 dh1[4];
        zero_acc(dh1);
        gemm_slab<true>(l_A + 32 * w * LDH, LDH, w2kj, H3, dh1, lane);
        float w1[4][NS], bb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bb[t] = m.b1[r + 32 * t];
#pragma unroll
            for (int i = 0; i < NS; ++i) w1[t][i] = m.W1[r + 32 * t + H3 * i];
        }
        float acc[NS + 1][4];  // [0] = db1, [1 + i] = dW1[:, i]
#pragma unroll
        for (int v = 0; v <= NS; ++v)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[v][t] = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 32 * w + mfma_row(q, kb);
            float x[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = l_x[i * TR + row];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float z = bb[t];
#pragma unroll
                for (int i = 0; i < NS; ++i) z = fmaf(w1[t][i], x[i], z);
                const float hv = act_fwd_t<ACT>(z);
                const float dz = dh1[t][q] * act_bwd_t<ACT>(z, hv);
                acc[0][t] += dz;
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[1 + i][t] = fmaf(dz, x[i], acc[1 + i][t]);
            }
            if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        reduce_cols_to_lds<NS + 1>(acc, l_red, w, lane);
    }
    __syncthreads();
    if (tid < H3) {
        store_wt(&out[ob1 + tid], sum_waves<NS + 1>(l_red, 0, tid));
#pragma unroll
        for (int i = 0; i < NS; ++i) store_wt(&out[oW1 + tid + H3 * i], sum_waves<NS + 1>(l_red, 1 + i, tid));
    }

    // ---- dW2^T[k][j] = sum_r H1[r][k] dZ2[r][j] (MFMA); stored as Flux W2[j + h k] ----
    {
        f32x16 dw[4];
        zero_acc(dw);
        gemm_slab<false>(l_B + 32 * w * LDH, LDH, l_C, LDH, dw, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) store_wt(&out[oW2 + (r + 32 * t) + H3 * (32 * w + mfma_row(q, kb))], dw[t][q]);
    }
}

}  // namespace rlhip
