// ppo3p_kernel.h -- the register-chained learner tile of ppo3t_kernel.h with TWO waves per SIMD.
//
// ppo3_gradT_kernel is VALU-bound with one wave per SIMD (512 registers per lane): 37 % of its wave time is spent parked at
// s_waitcnt / s_barrier and every MFMA issue slot is exposed, because no second wave is there to issue meanwhile
// (profiles/r02_ppo3_gradT.md).  Here a wave stays under 256 registers and eight waves share a CU:
//   * 6 PRODUCER waves run the tile of 32 samples each (layer 1, layer 2 in both operand roles, head, loss, dZ2, dH1,
//     dW1 / db1 / dW3 / db2 per lane), with the two operand roles of layer 2 one after the other and two unit tiles of the
//     "units on lanes" forms at a time: at most 64 accumulator registers are live;
//   * 2 CONSUMER waves own dW2 (64 rows u x 128 columns = 128 accumulator registers each) for the samples of all six producers.  Their
//     B operands (dZ2a) come through 8 KB LDS slabs the producers write in fragment order; their A operands (H1Y = layer 1
//     with the units on the lanes) they RECOMPUTE from the 32 observations per producer (16 bytes each, in LDS anyway) with
//     the same fmaf chain -- bit-identical to the producers' bf16 h1, ~0.9 k VALU instructions per round on the two SIMDs
//     that host only one producer, and 48 KB of LDS instead of 96 KB, which is what lets BOTH weight images live in LDS
//     (W2^T fragments from L2 behind the scattered sample gather, in-order vector-memory returns: 150 - 250 us per launch);
//   * the producers need act'(z1) on the "units on lanes" form: the MFMA transposition of their bf16 h1 (ppo3t_kernel.h),
//     two MFMAs per unit tile right where the dH1 epilogue wants it, sign test on the f32 result -- no LDS round trip.
// LDS: small tensors 13 KB | W2 image 32 KB | W2^T image 32 KB | six 8 KB slabs = 125 KB, one 8-wave workgroup per CU.
// Two barriers per round of 192 samples: A = slabs and observations complete (consumers start), B = consumers done.
// relu only (as the chained tile); numerics identical to ppo3t_kernel.h up to summation order.
//
// Built with -fno-slp-vectorize (build.py): SLP-packed v_pk_fma_f32 pairs cost registers (even-aligned pairs -> scratch
// at the 256-register budget) and issue slots beside MFMAs, and one op_sel form of it (high half of an LDS-loaded pair as
// the multiplier of both lanes) produced run-to-run different dW1 sums in this kernel at two waves per SIMD; the
// determinism test (tests/test_gpu_ppo3.py) pins all three learner tiles bit-identical run to run.
#pragma once

namespace rlhip {

// per-phase cycle stamps of one steady-state round (workgroup 0 of the actor; waves 0, 2 and 6), -DRLHIP_P3P_TIMING
#ifdef RLHIP_P3P_TIMING
__device__ long long g_p3p_stamps[3][16];
#define P3P_STAMP(k)                                                                                           \
    do {                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!CRITIC && wg == 0 && lane == 0 && round == wg + nwg && (w == 0 || w == 2 || w == 6))             \
            g_p3p_stamps[w == 0 ? 0 : (w == 2 ? 1 : 2)][(k)] = __builtin_readcyclecounter();                 \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    } while (0)
#else
#define P3P_STAMP(k) \
    do {             \
    } while (0)
#endif

constexpr int P3P_PROD = 6;   // producer waves per workgroup
constexpr int P3P_CONS = 2;   // consumer waves (each owns 64 columns of dW2)
constexpr int P3P_ROUND = 32 * P3P_PROD;  // samples per workgroup round
constexpr int P3P_SMALL = (H3 * 4 + H3 + H3 + MAXO * H3) * 4 + 2 * P3P_PROD * 512 + 2048;  // w1r, b1, b2, w3 | xs | dl | identity
constexpr size_t GRADP_LDS = P3P_SMALL + 2 * T3_FRAG * sizeof(uint16_t) + (size_t)P3P_PROD * 8192;

__device__ __forceinline__ void zero_acc2(f32x16 (&a)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) a[t][q] = 0.0f;
}

template <int NS, int NOUT, int CONT, int CRITIC>
__device__ __forceinline__ void ppo3P_body(const P3Args& g, int wg, int nwg, int nrounds, char* smem) {
    // LDS map (bytes).  Small tensors first: every constant offset from a lane-dependent base then fits the 16-bit offset
    // field of the ds instructions.
    constexpr int O_B1 = H3 * 16, O_B2 = O_B1 + H3 * 4, O_W3 = O_B2 + H3 * 4, O_XS = O_W3 + MAXO * H3 * 4,
                  O_DL = O_XS + P3P_PROD * 512, O_ID = O_DL + P3P_PROD * 512, O_F = O_ID + 2048, O_G = O_F + T3_FRAG * 2,
                  O_X = O_G + T3_FRAG * 2;
    static_assert(O_F == P3P_SMALL, "LDS map");
    float4* l_w1r = reinterpret_cast<float4*>(smem);  // [H3] {W1[u, 0..3]}
    float* l_b1 = reinterpret_cast<float*>(smem + O_B1);
    float* l_b2 = reinterpret_cast<float*>(smem + O_B2);
    float* l_w3 = reinterpret_cast<float*>(smem + O_W3);  // [MAXO][H3]
    uint16_t* l_id = reinterpret_cast<uint16_t*>(smem + O_ID);  // block-identity fragments [2][64][8]
    uint16_t* l_F = reinterpret_cast<uint16_t*>(smem + O_F);    // W2 fragments (lane = j, k = u in pi order) | W2^T fragments
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, kb = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ p = g.params + (CRITIC ? g.np_a : 0);
    const float* W1 = p;
    const float* b1 = W1 + H3 * NS;
    const float* b2 = b1 + H3 + H3 * H3;
    const float* W3 = b2 + H3;
    const float* b3 = W3 + NOUT * H3;
    const uint16_t* __restrict__ pk = g.packed + (CRITIC ? 2 * H3 * H3 : 0);

    // ---- stage: both pi-ordered W2 images (ppo3t_kernel.h: stage_pi_images, here with 512 threads) and the small tensors ----
    {
        constexpr int IT = 2 * T3_FRAG / 8 / 512;  // 8 slots of 16 bytes per thread
        uint2 a0[IT], a1[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int q8 = tid + 512 * it;
            const int l = q8 & 63, f = q8 >> 6;
            a0[it] = *reinterpret_cast<const uint2*>(pk + ((size_t)(f * 64 + (l & 31)) * 8 + 4 * (l >> 5)));
            a1[it] = *reinterpret_cast<const uint2*>(pk + ((size_t)(f * 64 + (l & 31) + 32) * 8 + 4 * (l >> 5)));
        }
#pragma unroll
        for (int it = 0; it < IT; ++it)
            *reinterpret_cast<uint4*>(l_F + 8 * (size_t)(tid + 512 * it)) = make_uint4(a0[it].x, a0[it].y, a1[it].x, a1[it].y);
    }
    if (tid >= 256 && tid < 256 + 128) {  // block-identity B fragments of the MFMA transposition (ppo3t_kernel.h)
        const int pp = (tid - 256) >> 6, l = tid & 63, cc = l & 31, kk = l >> 5;
        float one8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) one8[i] = (16 * pp + 4 * kk + (i & 3) + 8 * (i >> 2) == cc) ? 1.0f : 0.0f;
        reinterpret_cast<bf16x8*>(l_id)[pp * 64 + l] = as_frag(one8);
    }
    if (tid < H3) {
        const int u = tid;
        l_w1r[u] = make_float4(W1[u], NS > 1 ? W1[u + H3] : 0.f, NS > 2 ? W1[u + 2 * H3] : 0.f, NS > 3 ? W1[u + 3 * H3] : 0.f);
        l_b1[u] = b1[u];
        l_b2[u] = b2[u];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) l_w3[o * H3 + u] = (o < NOUT) ? W3[o + NOUT * u] : 0.f;
    }
    __syncthreads();

    float* out = g.partials + (int64_t)wg * g.np + (CRITIC ? g.np_a : 0);
    const int ob1 = H3 * NS, oW2 = ob1 + H3, ob2 = oW2 + H3 * H3, oW3 = ob2 + H3, ob3 = oW3 + NOUT * H3;
    // LDS addressing: one register per lane-dependent part, made opaque so that every access is  base + 16-bit constant
    // (left to itself the compiler fuses  lane part * scale + constant  into one v_mad per distinct constant and keeps
    // ~30 loop-invariant address registers alive across the round loop)
    auto opq = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };
    char* const p_kb64 = smem + opq(kb * 64);   // float4 rows 4 kb + k: l_w1r, and (+ O_XS + 512 v) the observations of producer v
    char* const p_l16 = smem + opq(lane * 16);  // fragment slot of this lane: F, G, identity, slabs
#define P3P_FF(f) (*reinterpret_cast<const bf16x8*>(p_l16 + O_F + 1024 * (f)))
#define P3P_GG(f) (*reinterpret_cast<const bf16x8*>(p_l16 + O_G + 1024 * (f)))

    if (w >= P3P_PROD) {
        // ============ consumer: dW2[64 cc .. 64 cc + 63, :] += H1Y^T dZ2a over the samples of all producers ============
        // (rows u, all 128 columns j: a consumer recomputes only ITS two unit tiles of H1Y)
        const int cc = w - P3P_PROD;
        float w1u[2][NS], b1u[2];  // layer 1 of this lane's units u = 32 (2 cc + tl) + c
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const float4 wr = l_w1r[32 * (2 * cc + tl) + c];
            const float wv[4] = {wr.x, wr.y, wr.z, wr.w};
#pragma unroll
            for (int i = 0; i < NS; ++i) w1u[tl][i] = wv[i];
            b1u[tl] = l_b1[32 * (2 * cc + tl) + c];
        }
        f32x16 acc[2][4];  // [unit tile tl][column tile jt]
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) zero_acc(acc[tl]);
        for (int round = wg; round < nrounds; round += nwg) {
            P3P_STAMP(0);
            __syncthreads();  // A: every producer's observations and dZ2a fragments of this round are in LDS
            P3P_STAMP(1);
#pragma unroll
            for (int v = 0; v < P3P_PROD; ++v)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 bfr[4];
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)  // dZ2a (lane = j): slot (column tile, s) of producer v's slab
                        bfr[jt] = *reinterpret_cast<const bf16x8*>(p_l16 + O_X + 8192 * v + 1024 * (jt * 2 + s));
                    // H1Y (lane = u): element i is sample row(8 s + i, kb) of producer v -- the layer-1 chain of the oracle
                    float h8[2][8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 xq = *reinterpret_cast<const float4*>(p_kb64 + O_XS + 512 * v + 16 * t3_row(8 * s + i, 0));
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl) {
                            float z = b1u[tl];
                            z = fmaf(w1u[tl][0], xq.x, z);
                            if (NS > 1) z = fmaf(w1u[tl][NS > 1 ? 1 : 0], xq.y, z);
                            if (NS > 2) z = fmaf(w1u[tl][NS > 2 ? 2 : 0], xq.z, z);
                            if (NS > 3) z = fmaf(w1u[tl][NS > 3 ? 3 : 0], xq.w, z);
                            h8[tl][i] = fmaxf(z, 0.0f);
                        }
                    }
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) {
                        const bf16x8 afr = as_frag(h8[tl]);
#pragma unroll
                        for (int jt = 0; jt < 4; ++jt)
                            acc[tl][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr[jt], acc[tl][jt], 0, 0, 0);
                    }
                }
            P3P_STAMP(2);
            __syncthreads();  // B: the slabs and the observations may be rewritten
            P3P_STAMP(3);
        }
        // D[u][j]: lanes = j (32 jt + c), regs = u (32 (2 cc + tl) + row(q, kb)); Flux layout W2[j + H u]
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    out[oW2 + (32 * jt + c) + H3 * (32 * (2 * cc + tl) + t3_row(q, kb))] = acc[tl][jt][q];
        __syncthreads();  // the final reduction of the producers' per-lane sums (below) has one barrier
        return;
    }

    // ======================================================= producer =======================================================
    float b3v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) b3v[o] = b3[o];
    float gW3[NOUT][4], gb2[4], gW1[NS][4], gb1[4], gb3[NOUT], sA = 0.f, sE = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        gb2[t] = 0.f;
        gb1[t] = 0.f;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) gW3[o][t] = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) gW1[i][t] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) gb3[o] = 0.f;
    char* const p_kb16 = smem + opq(kb * 16);              // floats 4 kb + k of l_b1 / l_b2 / l_w3
    char* const p_c4 = smem + opq(c * 4);                  // float c of l_b2 / l_w3
    char* const p_xw = smem + opq(c * 16 + w * 512);       // float4 row c of this wave's xs / dl
    char* const p_xr = smem + opq(kb * 64 + w * 512);      // float4 rows 4 kb + k of this wave's xs / dl
    char* const p_slab = smem + opq(lane * 16 + w * 8192);  // fragment slot of this lane in this wave's slab
#define P3P_W1R(k) (*reinterpret_cast<const float4*>(p_kb64 + 16 * (k)))
#define P3P_F4(off, k) (*reinterpret_cast<const float4*>(p_kb16 + (off) + 4 * (k)))
#define P3P_XS(row0) (*reinterpret_cast<const float4*>(p_xr + O_XS + 16 * (row0)))
#define P3P_DL(row0) (*reinterpret_cast<const float4*>(p_xr + O_DL + 16 * (row0)))
#define P3P_XD(slot) (*reinterpret_cast<bf16x8*>(p_slab + O_X + 1024 * (slot)))
    struct Samp {
        float x[4];
        float lp, adv, ret, act;
        bool valid;
    };
    auto fetch = [&](int round) {
        Samp sm;
        const uint32_t qs = (uint32_t)round * (uint32_t)P3P_ROUND + 32u * (uint32_t)w + (uint32_t)c;
        sm.valid = round < nrounds && qs < g.bm;
        const uint32_t f = permute(g.pk, g.pos0 + (sm.valid ? qs : 0u));
        const uint32_t tt = f / (uint32_t)g.n, ii = f - tt * (uint32_t)g.n;
#pragma unroll
        for (int k = 0; k < 4; ++k) sm.x[k] = (k < NS) ? g.obs[((int64_t)tt * NS + k) * g.n + ii] : 0.f;
        sm.lp = sm.adv = sm.ret = sm.act = 0.f;
        if (!CRITIC) {
            sm.lp = g.logp[f];
            sm.adv = sm.valid ? g.adv[f] : 0.0f;
            sm.act = CONT ? g.action_f[f] : __int_as_float(g.action_i[f]);
        } else {
            sm.ret = g.ret[f];
        }
        return sm;
    };
    struct CG {
        float4 b;
        float4 w[NOUT];
    };
    auto ldg = [&](int gi) {
        CG r;
        const int j0 = 32 * (gi >> 2) + 8 * (gi & 3);  // + 4 kb (in the base)
        r.b = P3P_F4(O_B2, j0);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) r.w[o] = P3P_F4(O_W3, o * H3 + j0);
        return r;
    };

    Samp nxt = fetch(wg);
    for (int round = wg; round < nrounds; round += nwg) {
        P3P_STAMP(0);
        const Samp cur = nxt;
        // the next round's scattered gather, one round ahead (anywhere else in the round the divergent cycle-walk of the
        // keyed permutation splits a scheduling region and the register allocation falls into scratch: 160 - 790 dwords)
        nxt = fetch(round + nwg);
        const bool valid = cur.valid;
        const float x[4] = {cur.x[0], cur.x[1], cur.x[2], cur.x[3]};
        if (kb == 0) *reinterpret_cast<float4*>(p_xw + O_XS) = make_float4(x[0], x[1], x[2], x[3]);

        // ---- (1) H1X (lanes = samples, regs = units) fused with H2b = mfma(A = W2 frag, B = H1X): lanes = samples ----
        bf16x8 hx[8];
        f32x16 acc[4];
        zero_acc(acc);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int t = ks >> 1, a = ks & 1;
            float h8[8];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int u0 = 32 * t + 8 * (2 * a + gg);  // + 4 kb (in the bases)
                const float4 bb = P3P_F4(O_B1, u0);
                const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 wr = P3P_W1R(u0 + i);
                    float z = bbv[i];
                    z = fmaf(wr.x, x[0], z);
                    if (NS > 1) z = fmaf(wr.y, x[1], z);
                    if (NS > 2) z = fmaf(wr.z, x[2], z);
                    if (NS > 3) z = fmaf(wr.w, x[3], z);
                    h8[4 * gg + i] = fmaxf(z, 0.0f);
                }
            }
            hx[ks] = as_frag(h8);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P3P_FF(ks * 4 + tt), hx[ks], acc[tt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // keeps the layer-1 operand loads of later k-steps out of this one's registers
        }
        __builtin_amdgcn_sched_barrier(0);
        P3P_STAMP(2);

        // ---- (2) form b: head, loss, dL/dout per lane; dZ2b (A operand of dH1a) ----
        float outv[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) outv[o] = 0.f;
        {
            CG cg = ldg(0), cn = cg;
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) {
                if (gi + 1 < 16) cn = ldg(gi + 1);
                const float bbv[4] = {cg.b.x, cg.b.y, cg.b.z, cg.b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = fmaxf(acc[gi >> 2][4 * (gi & 3) + i] + bbv[i], 0.0f);
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        const float w3v = i == 0 ? cg.w[o].x : (i == 1 ? cg.w[o].y : (i == 2 ? cg.w[o].z : cg.w[o].w));
                        outv[o] = fmaf(w3v, hv, outv[o]);
                    }
                }
                cg = cn;
            }
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) outv[o] = (outv[o] + __shfl_xor(outv[o], 32, 64)) + b3v[o];
        float dl[NOUT];
        {
            float l0, l1;
            ppo3_sample_loss<NOUT, CONT, CRITIC>(g, outv, cur.lp, cur.adv, cur.ret, cur.act, valid, dl, l0, l1);
            if (kb == 0) {
                sA += l0;
                sE += l1;
            }
        }
        if (kb == 0) {
            *reinterpret_cast<float4*>(p_xw + O_DL) = make_float4(dl[0], NOUT > 1 ? dl[NOUT > 1 ? 1 : 0] : 0.f, NOUT > 2 ? dl[NOUT > 2 ? 2 : 0] : 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) gb3[o] += dl[o];
        }
        // the second pass over b2 / W3 re-reads them from LDS: without the compiler barrier the 16 groups of the first pass
        // (192 registers) are kept alive for it -- through scratch
        asm volatile("" ::: "memory");
        bf16x8 dzb[8];
        {
            CG cg = ldg(0), cn = cg;
            float d8[8];
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) {
                if (gi + 1 < 16) cn = ldg(gi + 1);
                const float bbv[4] = {cg.b.x, cg.b.y, cg.b.z, cg.b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z2 = acc[gi >> 2][4 * (gi & 3) + i] + bbv[i];
                    float dh = 0.0f;
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        const float w3v = i == 0 ? cg.w[o].x : (i == 1 ? cg.w[o].y : (i == 2 ? cg.w[o].z : cg.w[o].w));
                        dh = fmaf(dl[o], w3v, dh);
                    }
                    d8[4 * (gi & 1) + i] = z2 > 0.0f ? dh : 0.0f;
                }
                if (gi & 1) dzb[gi >> 1] = as_frag(d8);
                cg = cn;
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        P3P_STAMP(3);
        // ---- (3) H2a = mfma(A = H1X, B = W2 frag): lanes = units, regs = samples; dW3 / db2 per lane, dZ2a into the slab.
        // Two unit tiles at a time: 32 accumulator registers live instead of 64 ----
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            f32x16 ac2[2];
            zero_acc2(ac2);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
                    ac2[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[ks], P3P_FF(ks * 4 + 2 * th + tl), ac2[tl], 0, 0, 0);
            float b2u[2], w3u[NOUT][2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                b2u[tl] = *reinterpret_cast<const float*>(p_c4 + O_B2 + 128 * (2 * th + tl));
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
                    w3u[o][tl] = *reinterpret_cast<const float*>(p_c4 + O_W3 + 4 * (o * H3 + 32 * (2 * th + tl)));
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t pkd[2][4];  // dZ2a of this half, bf16 pairs, packed as they are produced
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    float dzp[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 8 * s2 + 2 * i2 + e;
                        const float4 dq = P3P_DL(t3_row(q, 0));
                        const float dqv[3] = {dq.x, dq.y, dq.z};
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl) {
                            const int t = 2 * th + tl;
                            const float z2 = ac2[tl][q] + b2u[tl];
                            const float hv = fmaxf(z2, 0.0f);
                            float dh = 0.0f;
#pragma unroll
                            for (int o = 0; o < NOUT; ++o) {
                                gW3[o][t] = fmaf(dqv[o], hv, gW3[o][t]);
                                dh = fmaf(dqv[o], w3u[o][tl], dh);
                            }
                            const float dz = z2 > 0.0f ? dh : 0.0f;
                            gb2[t] += dz;
                            dzp[e][tl] = dz;
                        }
                    }
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl) pkd[tl][i2] = pack2_bf16(dzp[0][tl], dzp[1][tl]);
                }
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
                    P3P_XD((2 * th + tl) * 2 + s2) =
                        __builtin_bit_cast(bf16x8, make_uint4(pkd[tl][0], pkd[tl][1], pkd[tl][2], pkd[tl][3]));
            }
        }
        // W2^T fragments come from the packed image in global memory (L2): a ring of 16 fragments, the first half of the
        // image requested before the barrier, fragment f + 16 requested when the MFMA of fragment f has been issued
        // (an L2 round trip under load is several hundred cycles: with one k-step of look-ahead the 64 loads of a round
        //  ran almost serialised and the kernel took 250 us)
        P3P_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // A: the slabs and observations of this round are complete -- the consumers start on dW2
        P3P_STAMP(6);

        // ---- (4) dH1a = mfma(A = dZ2b, B = W2^T frag): lanes = units u, regs = samples; dW1 / db1 per lane.  Two unit tiles at
        // a time; act'(z1) = [h1 > 0] from the MFMA transposition of this wave's bf16 h1 (exact), tile by tile ----
        {
            const bf16x8 idf[2] = {*reinterpret_cast<const bf16x8*>(p_l16 + O_ID),
                                   *reinterpret_cast<const bf16x8*>(p_l16 + O_ID + 1024)};
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                f32x16 ac2[2];
                zero_acc2(ac2);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
                        ac2[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzb[ks], P3P_GG(ks * 4 + 2 * th + tl), ac2[tl], 0, 0, 0);
                    if (ks & 1) __builtin_amdgcn_sched_barrier(0);  // at most four weight fragments in flight
                }
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const int tu = 2 * th + tl;
                    f32x16 hyD;
#pragma unroll
                    for (int q = 0; q < 16; ++q) hyD[q] = 0.0f;
                    hyD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[2 * tu], idf[0], hyD, 0, 0, 0);
                    hyD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[2 * tu + 1], idf[1], hyD, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float dz = hyD[q] > 0.0f ? ac2[tl][q] : 0.0f;
                        const float4 xq = P3P_XS(t3_row(q, 0));
                        const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
                        gb1[tu] += dz;
#pragma unroll
                        for (int k = 0; k < NS; ++k) gW1[k][tu] = fmaf(dz, xv[k], gW1[k][tu]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        P3P_STAMP(8);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // B: the consumers have read the slabs and the observations; xs / dl of this round are dead
        P3P_STAMP(9);
    }

    // ---- this workgroup's partial gradient: per-lane sums of the two halves of a wave and of the six producers ----
    constexpr int NV = 2 + NOUT + NS;  // db2, db1, dW3[o], dW1[i]
    float* l_red = reinterpret_cast<float*>(l_F);  // [producer][NV][H3] (the weight images are dead: their last readers are behind barrier B)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float vals[NV];
        vals[0] = gb2[t];
        vals[1] = gb1[t];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) vals[2 + o] = gW3[o][t];
#pragma unroll
        for (int i = 0; i < NS; ++i) vals[2 + NOUT + i] = gW1[i][t];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float sum = vals[v] + __shfl_xor(vals[v], 32, 64);
            if (kb == 0) l_red[(w * NV + v) * H3 + 32 * t + c] = sum;
        }
    }
    float small[NOUT + 2];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) small[o] = wave_sum_f32(gb3[o]);
    small[NOUT] = wave_sum_f32(sA);
    small[NOUT + 1] = wave_sum_f32(sE);
    float* l_small = l_red + P3P_PROD * NV * H3;  // [producer][8]
    if (lane == 0)
#pragma unroll
        for (int o = 0; o < NOUT + 2; ++o) l_small[w * 8 + o] = small[o];
    __syncthreads();
    if (tid < H3) {
        const int u = tid;
        auto sumP = [&](int v) {
            float a = l_red[(0 * NV + v) * H3 + u];
#pragma unroll
            for (int pw = 1; pw < P3P_PROD; ++pw) a += l_red[(pw * NV + v) * H3 + u];
            return a;
        };
        out[ob2 + u] = sumP(0);
        out[ob1 + u] = sumP(1);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[oW3 + o + NOUT * u] = sumP(2 + o);
#pragma unroll
        for (int i = 0; i < NS; ++i) out[u + H3 * i] = sumP(2 + NOUT + i);
    }
    if (tid == 0) {
        auto sP = [&](int o) {
            float a = l_small[o];
#pragma unroll
            for (int pw = 1; pw < P3P_PROD; ++pw) a += l_small[pw * 8 + o];
            return a;
        };
#pragma unroll
        for (int o = 0; o < NOUT; ++o) out[ob3 + o] = sP(o);
        float* lo = g.loss_partials + (int64_t)wg * 4;
        if (!CRITIC) {
            lo[0] = sP(NOUT);
            lo[2] = sP(NOUT + 1);
        } else {
            lo[1] = sP(NOUT);
        }
    }
}

template <int NS, int CONT>
__global__ __launch_bounds__(512, 2) void ppo3_gradP_kernel(P3Args g, int nwg, int nrounds) {
    extern __shared__ __attribute__((aligned(16))) char smemP[];
    const int net = blockIdx.x & 1, wg = blockIdx.x >> 1;
    if (net == 0) ppo3P_body<NS, 2, CONT, 0>(g, wg, nwg, nrounds, smemP);
    else ppo3P_body<NS, 1, 0, 1>(g, wg, nwg, nrounds, smemP);
}

#undef P3P_W1R
#undef P3P_F4
#undef P3P_XS
#undef P3P_DL
#undef P3P_FF
#undef P3P_GG
#undef P3P_XD

}  // namespace rlhip
