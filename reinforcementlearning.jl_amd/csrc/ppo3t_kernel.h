// ppo3t_kernel.h -- the register-chained ("transposed") learner tile of the three-layer PPO actor / critic
//     ns -> 128 -> 128 -> nout, hidden x hidden on v_mfma_f32_32x32x16_bf16, included by ppo3.hip.
//
// Why a second formulation.  The round-1 tile (ppo3_grad_kernel, mlp3_device.h) gives every wave 32 sample ROWS and
// moves activations between layers through LDS tiles in two layouts; it sits at 12 workgroup barriers per tile with
// 62 % of the wave time parked (profiles/r01_pmc_gather_dqn3.md), writes a full 138 KB partial gradient per 128-sample
// tile (141 MB per 131072-sample micro-batch, read back by the reduction) and reaches 5 % of the bf16 MFMA peak.
//
// Here every MFMA operand that is an ACTIVATION comes straight out of the accumulator registers of the MFMA (or the VALU
// code) that produced it -- no LDS tile, no layout conversion, no barrier -- because the three register images of
// v_mfma_f32_32x32x16 are compatible up to a fixed permutation:
//     A operand   lane (row  = l & 31, kb = l >> 5) holds 8 consecutive k
//     B operand   lane (col  = l & 31, kb = l >> 5) holds 8 consecutive k
//     D result    lane (col  = l & 31, kb = l >> 5) holds 16 rows  q -> (q & 3) + 8 (q >> 2) + 4 kb
// A D register file, converted to bf16 eight values at a time (q = 0..7 / 8..15), IS an A or a B operand of the next
// MFMA whose reduction index runs over D's rows in the order pi(16 s + 8 kb + i) = 16 s + 4 kb + (i & 3) + 8 (i >> 2);
// the WEIGHT fragments it meets are stored in that order once (staged into LDS at kernel entry from the packed images).
// Using a register file as A puts the samples on the result's ROWS, using it as B puts them on the COLUMNS, so each
// quantity is produced directly in the form its consumer needs:
//
//   H1X  = act(W1 x + b1)          lanes = samples, regs = units   (VALU, f32 fmaf chain of the oracle)      -> bf16
//   H2b  = mfma(A = W2 frag, B = H1X)   lanes = samples, regs = units    head, loss, dL/dout per LANE, no reductions
//   H2a  = mfma(A = H1X, B = W2 frag)   lanes = units,   regs = samples  dW3 / db2 accumulate per LANE
//          (the two MFMAs share one 16-byte LDS read of the weight fragment)
//   dZ2b -> A operand of  dH1a = mfma(A = dZ2b, B = W2^T frag)   lanes = units, regs = samples: dW1 / db1 per LANE
//   H1Y  = the same layer 1 with lanes = units, regs = samples   (VALU)  -> A operand of dW2
//   dW2 += mfma(A = H1Y, B = dZ2a)   reduction over the samples; result lanes = j, regs = u
//
// The one exchange between waves: dW2 (128 x 128 per net) would be 256 accumulator registers per wave, so the four
// waves of a workgroup split its COLUMNS -- wave w accumulates dW2[:, 32 w .. 32 w + 31] over the samples of all four
// waves, whose H1Y / dZ2a fragments it reads from an LDS slab the owners wrote in fragment order (lane-linear 16-byte
// accesses, conflict-free): two barriers per 128-sample tile instead of twelve.  Gradient accumulators live in
// registers across all tiles of a persistent workgroup; one partial row per workgroup (<= 128 per net instead of
// 1024 per micro-batch).  Actor and critic are independent given (obs, adv, ret): even workgroups take the actor,
// odd ones the critic, so each holds one net's 64 KB of weight fragments.
//
// Numerics: identical roundings to mlp3_device.h / oracle/rlo_mlp3.c (layer 1 in f32 with the same fmaf chain, h1 and
// dz2 rounded to bf16 RNE, f32 accumulation); only the summation ORDER inside the MFMAs and over the samples differs.
#pragma once

namespace rlhip {

// per-phase cycle stamps of one steady-state tile (workgroup 0, wave 0), compile-time option -DRLHIP_T3_TIMING
#ifdef RLHIP_T3_TIMING
__device__ long long g_t3_stamps[16];
#define T3_STAMP(k)                                                                                  \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (!CRITIC && wg == 0 && tid == 0 && tile == wg + 2 * nwg) g_t3_stamps[(k)] = clock64();    \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#else
#define T3_STAMP(k) \
    do {            \
    } while (0)
#endif

constexpr int T3_FRAG = H3 * H3;  // bf16 elements of one 128 x 128 fragment image (32 KB)
constexpr size_t GRADT_LDS = 4 * T3_FRAG * sizeof(uint16_t) + (H3 * 4 + H3 + H3 + MAXO * H3) * sizeof(float) +
                             2 * 4 * 32 * sizeof(float4);

__device__ __forceinline__ bf16x8 as_frag(const float (&v)[8]) { return __builtin_bit_cast(bf16x8, pack8_bf16(v)); }
__device__ __forceinline__ int t3_row(int q, int kb) { return (q & 3) + 8 * (q >> 2) + 4 * kb; }

// natural fragment images (lane = column, 8 consecutive k: ppo3_pack_kernel; W2jk then W2kj, 2 x 32 KB contiguous) -> the
// pi-ordered images in LDS.  All 32 8-byte loads of a thread are issued before the first LDS store: one L2 round trip
// for the 64 KB instead of sixteen (the per-launch fixed cost of this kernel was ~20 us, half of a 2-tile launch).
__device__ __forceinline__ void stage_pi_images(const uint16_t* __restrict__ nat, uint16_t* l_img, int tid) {
    constexpr int IT = 2 * T3_FRAG / 8 / 256;  // 16 slots of 16 bytes per thread over both images
    uint2 a0[IT], a1[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int q8 = tid + 256 * it;
        const int l = q8 & 63, f = q8 >> 6;
        const int cc = l & 31, kk = l >> 5;
        a0[it] = *reinterpret_cast<const uint2*>(nat + ((size_t)(f * 64 + cc) * 8 + 4 * kk));
        a1[it] = *reinterpret_cast<const uint2*>(nat + ((size_t)(f * 64 + cc + 32) * 8 + 4 * kk));
    }
#pragma unroll
    for (int it = 0; it < IT; ++it)
        *reinterpret_cast<uint4*>(l_img + 8 * (size_t)(tid + 256 * it)) = make_uint4(a0[it].x, a0[it].y, a1[it].x, a1[it].y);
}

// Per-sample loss terms and dL/d(head outputs) of one net (the per-sample block of ppo3_grad_kernel): PPO clipped surrogate
// + entropy for the actor (categorical or Gaussian head), squared value error for the critic.  acc0 receives
// min(surr1, surr2) (actor) or (ret - v)^2 (critic), acc1 the entropy (actor); zero for samples beyond the micro-batch.
template <int NOUT, int CONT, int CRITIC>
__device__ __forceinline__ void ppo3_sample_loss(const P3Args& g, const float (&outv)[NOUT], float m_lp, float m_adv,
                                                 float m_ret, float m_act, bool valid, float (&dl)[NOUT], float& acc0,
                                                 float& acc1) {
    acc0 = 0.f;
    acc1 = 0.f;
    if (!CRITIC) {
        const float lp_old = fmaxf(m_lp, g.min_logp);  // clamp!(log_p, log(1e-8), Inf)
        const float A = m_adv;
        float ent, surr_min;
        if (!CONT) {
            float mx = outv[0];
#pragma unroll
            for (int k = 1; k < NOUT; ++k) mx = fmaxf(mx, outv[k]);
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) se += expf(outv[k] - mx);
            const float lse = logf(se);
            float logp[NOUT], pr[NOUT];
            ent = 0.f;
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                logp[k] = (outv[k] - mx) - lse;
                pr[k] = expf(logp[k]);
                ent -= pr[k] * logp[k];
            }
            const int a = __float_as_int(m_act);
            float lp_new = 0.f;
#pragma unroll
            for (int k = 0; k < NOUT; ++k)
                if (k == a) lp_new = logp[k];
            const float ratio = expf(lp_new - lp_old);
            const float surr1 = ratio * A;
            const float surr2 = fminf(fmaxf(ratio, g.lo), g.hi) * A;
            const bool inside = ratio >= g.lo && ratio <= g.hi;
            const float dobj = (inside || surr1 < surr2) ? A : 0.f;
            const float dL_dlp = -g.wa * g.inv_b * dobj * ratio;
            surr_min = fminf(surr1, surr2);
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                const float dlp = ((k == a) ? 1.f : 0.f) - pr[k];
                const float dent = -pr[k] * (logp[k] + ent);
                dl[k] = dL_dlp * dlp - g.we * g.inv_b * dent;
            }
        } else {
            const float eps = 1.0e-8f;
            const float mu = outv[0], ls = outv[NOUT > 1 ? 1 : 0];
            const float sg = expf(ls);
            const float z = m_act;
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
            if (NOUT > 1) dl[NOUT > 1 ? 1 : 0] = dL_dlp * dls - g.we * g.inv_b * 0.5f;
        }
        if (!valid) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dl[o] = 0.f;
            surr_min = 0.f;
            ent = 0.f;
        }
        acc0 = surr_min;
        acc1 = ent;
    } else {
        const float dv = m_ret - outv[0];
        float dvout = -2.0f * g.wc * g.inv_b * dv;
        float sq = dv * dv;
        if (!valid) {
            dvout = 0.f;
            sq = 0.f;
        }
        dl[0] = dvout;
        acc0 = sq;
    }
}

template <int NS, int NOUT, int ACT, int CONT, int CRITIC>
__device__ __forceinline__ void ppo3T_body(const P3Args& g, int wg, int nwg, int ntiles, char* smem) {
    uint16_t* l_F = reinterpret_cast<uint16_t*>(smem);  // W2 fragments   (lane = j, k = u in pi order)
    uint16_t* l_G = l_F + T3_FRAG;                      // W2^T fragments (lane = u, k = j in pi order)
    uint16_t* l_XH = l_G + T3_FRAG;                     // exchange: H1Y fragments  [wave][t][s][lane][8]
    uint16_t* l_XD = l_XH + T3_FRAG;                    // exchange: dZ2a fragments [wave][t][s][lane][8]
    float4* l_w1r = reinterpret_cast<float4*>(l_XD + T3_FRAG);  // [H3] {W1[u, 0..3]}
    float* l_b1 = reinterpret_cast<float*>(l_w1r + H3);
    float* l_b2 = l_b1 + H3;
    float* l_w3 = l_b2 + H3;  // [MAXO][H3]
    float4* l_xs = reinterpret_cast<float4*>(l_w3 + MAXO * H3);  // [4 waves][32] observations
    float4* l_dl = l_xs + 4 * 32;                                // [4 waves][32] dL/d(outputs)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, kb = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ p = g.params + (CRITIC ? g.np_a : 0);
    const float* W1 = p;
    const float* b1 = W1 + H3 * NS;
    const float* b2 = b1 + H3 + H3 * H3;
    const float* W3 = b2 + H3;
    const float* b3 = W3 + NOUT * H3;
    const uint16_t* __restrict__ pk = g.packed + (CRITIC ? 2 * H3 * H3 : 0);

    // ---- stage the weights ----
    stage_pi_images(pk, l_F, tid);  // l_G follows l_F, W2kj follows W2jk: one pass over both
    if (tid < H3) {
        const int u = tid;
        l_w1r[u] = make_float4(W1[u], NS > 1 ? W1[u + H3] : 0.f, NS > 2 ? W1[u + 2 * H3] : 0.f, NS > 3 ? W1[u + 3 * H3] : 0.f);
        l_b1[u] = b1[u];
        l_b2[u] = b2[u];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) l_w3[o * H3 + u] = (o < NOUT) ? W3[o + NOUT * u] : 0.f;
    }
    float b3v[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) b3v[o] = b3[o];
    // Block-identity B fragments: D = mfma(A = X, B = I) re-emits a register file X (lanes = samples, regs = units) with
    // the roles swapped (lanes = units, regs = samples) -- a transposition on the idle matrix pipe, exact for bf16 data.
    // Tile t of the result only meets the k-steps 2 t and 2 t + 1; idf[p] is the fragment of parity p = ks & 1:
    // element i of lane (c, kb) is 1 where pi(16 p + 8 kb + i) = 16 p + 4 kb + (i & 3) + 8 (i >> 2) equals the column c.
    bf16x8 idf[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        float one8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) one8[i] = (16 * pp + 4 * kb + (i & 3) + 8 * (i >> 2) == c) ? 1.0f : 0.0f;
        idf[pp] = as_frag(one8);
    }
    __syncthreads();

    // ---- accumulators that live across all tiles of this workgroup ----
    f32x16 accW2[4];
    zero_acc(accW2);
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
    float4* xs = l_xs + 32 * w;
    float4* dls = l_dl + 32 * w;
    const bf16x8* Ff = reinterpret_cast<const bf16x8*>(l_F) + lane;  // fragment (ks, t) at Ff[(ks * 4 + t) * 64]
    const bf16x8* Gf = reinterpret_cast<const bf16x8*>(l_G) + lane;
    bf16x8* XHw = reinterpret_cast<bf16x8*>(l_XH) + (size_t)w * 8 * 64 + lane;  // own slab: [t][s] at XHw[(t * 2 + s) * 64]
    bf16x8* XDw = reinterpret_cast<bf16x8*>(l_XD) + (size_t)w * 8 * 64 + lane;

    // the scattered gather of a tile's samples (keyed permutation -> trajectory) is issued ONE TILE AHEAD: with a single
    // wave per SIMD its ~2 us of HBM / L2 latency would otherwise be exposed at the top of every tile
    struct Samp {
        float x[4];
        float lp, adv, ret, act;
        bool valid;
    };
    auto fetch = [&](int tile) {
        Samp sm;
        const uint32_t qs = (uint32_t)tile * 128u + 32u * (uint32_t)w + (uint32_t)c;
        sm.valid = tile < ntiles && qs < g.bm;
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
    Samp nxt = fetch(wg);
#ifdef RLHIP_T3_TIMING
    if (!CRITIC && wg == 0 && tid == 0) g_t3_stamps[12] = clock64();
#endif
    for (int tile = wg; tile < ntiles; tile += nwg) {
        // ---- this lane's sample r = c (both halves of the wave hold the same 32 samples) ----
        const Samp cur = nxt;
        nxt = fetch(tile + nwg);
        const bool valid = cur.valid;
        const float x[4] = {cur.x[0], cur.x[1], cur.x[2], cur.x[3]};
        const float m_lp = cur.lp, m_adv = cur.adv, m_ret = cur.ret, m_act = cur.act;
        T3_STAMP(0);
        if (kb == 0) xs[c] = make_float4(x[0], x[1], x[2], x[3]);

        // ---- (1) H1X fused with layer 2 in both forms.  H1X: lanes = samples, regs (t, q) = unit 32 t + row(q, kb); its bf16
        // operand of k-step ks = 2 t + a covers q = 8 a .. 8 a + 7.  Step ks of the MFMA loop needs only hx[ks], so the VALU
        // evaluation of step ks + 1 sits in the same scheduling region as the 8 MFMAs of step ks: a single wave per SIMD
        // has no other wave to fill the matrix pipe's 32-cycle issue slots ----
        f32x16 aa[4], ab[4];
        zero_acc(aa);
        zero_acc(ab);
        bf16x8 hx[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int t = ks >> 1, a = ks & 1;
            float h8[8];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int u0 = 32 * t + 8 * (2 * a + gg) + 4 * kb;
                const float4 bb = *reinterpret_cast<const float4*>(l_b1 + u0);
                const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 wr = l_w1r[u0 + i];
                    float z = bbv[i];
                    z = fmaf(wr.x, x[0], z);
                    if (NS > 1) z = fmaf(wr.y, x[1], z);
                    if (NS > 2) z = fmaf(wr.z, x[2], z);
                    if (NS > 3) z = fmaf(wr.w, x[3], z);
                    h8[4 * gg + i] = act_fwd_t<ACT>(z);
                }
            }
            hx[ks] = as_frag(h8);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const bf16x8 fr = Ff[(ks * 4 + tt) * 64];
                aa[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[ks], fr, aa[tt], 0, 0, 0);  // [sample][unit]: lanes = j
                ab[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr, hx[ks], ab[tt], 0, 0, 0);  // [unit][sample]: lanes = r
            }
        }
        if (ACT == 0) {
            // H1Y (lanes = units, regs = samples), the A operand of dW2, as the transposition of H1X: 8 MFMAs instead of
            // a second VALU evaluation of layer 1; one result tile at a time, straight into this wave's exchange slab
            // (its readers of the previous tile are behind the closing barrier)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16 hyD;
#pragma unroll
                for (int q = 0; q < 16; ++q) hyD[q] = 0.0f;
                hyD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[2 * t], idf[0], hyD, 0, 0, 0);
                hyD = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hx[2 * t + 1], idf[1], hyD, 0, 0, 0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float h8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) h8[i] = hyD[8 * s2 + i];
                    XHw[(t * 2 + s2) * 64] = as_frag(h8);
                }
            }
        }
        T3_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);

        // ---- (2) form b (lanes = samples): bias + activation, head, loss, dL/d(outputs) -- everything per lane ----
        float outv[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) outv[o] = 0.f;
        // b2 / W3 of the lane's 64 units stream from LDS in 16 groups of 4 units; with one wave per SIMD nothing hides an
        // LDS round trip but the wave's own instructions, so group g + 1 is requested before group g is used and scheduling
        // barriers pin that order (the compiler's own placement was load -> wait -> use, 16 exposed latencies per pass)
        struct CG {
            float4 b;
            float4 w[NOUT];
        };
        auto ldg = [&](int gi) {
            CG r;
            const int j0 = 32 * (gi >> 2) + 8 * (gi & 3) + 4 * kb;
            r.b = *reinterpret_cast<const float4*>(l_b2 + j0);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) r.w[o] = *reinterpret_cast<const float4*>(l_w3 + o * H3 + j0);
            return r;
        };
        {
            CG cur = ldg(0), nxt = cur;
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) {
                if (gi + 1 < 16) nxt = ldg(gi + 1);
                const float bbv[4] = {cur.b.x, cur.b.y, cur.b.z, cur.b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = act_fwd_t<ACT>(ab[gi >> 2][4 * (gi & 3) + i] + bbv[i]);
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        const float w3v = i == 0 ? cur.w[o].x : (i == 1 ? cur.w[o].y : (i == 2 ? cur.w[o].z : cur.w[o].w));
                        outv[o] = fmaf(w3v, hv, outv[o]);
                    }
                }
                cur = nxt;
            }
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) outv[o] = (outv[o] + __shfl_xor(outv[o], 32, 64)) + b3v[o];
        float dl[NOUT];
        {
            float l0, l1;
            ppo3_sample_loss<NOUT, CONT, CRITIC>(g, outv, m_lp, m_adv, m_ret, m_act, valid, dl, l0, l1);
            if (kb == 0) {
                sA += l0;
                sE += l1;
            }
        }
        if (kb == 0) {
            dls[c] = make_float4(dl[0], NOUT > 1 ? dl[NOUT > 1 ? 1 : 0] : 0.f, NOUT > 2 ? dl[NOUT > 2 ? 2 : 0] : 0.f, 0.f);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) gb3[o] += dl[o];
        }
        T3_STAMP(2);
        // dZ2b = (W3^T dl) * act'(z2): the A operand of dH1a.  The accumulators stay untouched in their registers: the
        // activation pattern is re-derived from z2 = acc + b2 (relu: one add, one compare) instead of keeping 64 h values
        bf16x8 dzb[8];
        {
            CG cur = ldg(0), nxt = cur;
            float d8[8];
#pragma unroll
            for (int gi = 0; gi < 16; ++gi) {
                if (gi + 1 < 16) nxt = ldg(gi + 1);
                const float bbv[4] = {cur.b.x, cur.b.y, cur.b.z, cur.b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float z2 = ab[gi >> 2][4 * (gi & 3) + i] + bbv[i];
                    float dh = 0.0f;
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        const float w3v = i == 0 ? cur.w[o].x : (i == 1 ? cur.w[o].y : (i == 2 ? cur.w[o].z : cur.w[o].w));
                        dh = fmaf(dl[o], w3v, dh);
                    }
                    if (ACT == 0) {
                        d8[4 * (gi & 1) + i] = z2 > 0.0f ? dh : 0.0f;  // = dh * act'(z2) for relu
                    } else {
                        const float hv = act_fwd_t<ACT>(z2);
                        d8[4 * (gi & 1) + i] = dh * act_bwd_t<ACT>(hv, hv);
                    }
                }
                if (gi & 1) dzb[gi >> 1] = as_frag(d8);  // k-step 2 t + a covers the groups g4 = 2 a, 2 a + 1
                cur = nxt;
            }
        }
        T3_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);

        // ---- (3) ONE scheduling region: the 32 MFMAs of dH1a = dZ2b W2 (lanes = units u, regs = samples) and, independent
        // of them, the VALU work of form a (lanes = units j = 32 t + c, regs = samples): h2a, dW3 / db2 per lane, dZ2a
        // (f32 for db2, bf16 into the slab as the B operand of dW2) ----
        f32x16 dh1[4];
        zero_acc(dh1);
        {
            float b2u[4], w3u[NOUT][4];  // this lane's units j = 32 t + c
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                b2u[t] = l_b2[32 * t + c];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) w3u[o][t] = l_w3[o * H3 + 32 * t + c];
            }
            // hand-pipelined: row q of form a (~40 VALU instructions) runs in the shadow of the two MFMAs 2 q, 2 q + 1;
            // their weight fragments and the row's dL/dout were requested one row earlier
            bf16x8 gcur[2] = {Gf[0], Gf[64]}, gnxt[2] = {gcur[0], gcur[1]};
            float4 dqc = dls[t3_row(0, kb)], dqn = dqc;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q + 1 < 16) {
                    gnxt[0] = Gf[(2 * q + 2) * 64];
                    gnxt[1] = Gf[(2 * q + 3) * 64];
                    dqn = dls[t3_row(q + 1, kb)];
                }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int m = 2 * q + mm;  // fragment (ks, tu) = (m >> 2, m & 3)
                    dh1[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzb[m >> 2], gcur[mm], dh1[m & 3], 0, 0, 0);
                }
                const float dqv[3] = {dqc.x, dqc.y, dqc.z};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float z2 = aa[t][q] + b2u[t];
                    const float hv = act_fwd_t<ACT>(z2);
                    float dh = 0.0f;
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) {
                        gW3[o][t] = fmaf(dqv[o], hv, gW3[o][t]);
                        dh = fmaf(dqv[o], w3u[o][t], dh);
                    }
                    const float dz = (ACT == 0) ? (z2 > 0.0f ? dh : 0.0f) : dh * act_bwd_t<ACT>(hv, hv);
                    gb2[t] += dz;
                    aa[t][q] = dz;  // the accumulator slot is free now: dz2a in place
                }
                gcur[0] = gnxt[0];
                gcur[1] = gnxt[1];
                dqc = dqn;
            }
            // dZ2a fragments straight into this wave's exchange slab (the B operands of dW2)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float d8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) d8[i] = aa[t][8 * s2 + i];
                    XDw[(t * 2 + s2) * 64] = as_frag(d8);
                }
        }
        float dact[ACT == 0 ? 1 : 64];  // tanh: 1 - h1^2 per element (relu: the sign of the bf16 h1 in the slab)
        if (ACT != 0) {
            // H1Y by a second VALU evaluation of layer 1 (lanes = units, regs = samples): the f32 h1 is needed for act'
            float w1u[4][NS], b1u[4];  // this lane's units u = 32 t + c
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 wr = l_w1r[32 * t + c];
                const float wv[4] = {wr.x, wr.y, wr.z, wr.w};
#pragma unroll
                for (int i = 0; i < NS; ++i) w1u[t][i] = wv[i];
                b1u[t] = l_b1[32 * t + c];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float h8[4][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = 8 * s + i;
                    const float4 xq = xs[t3_row(q, kb)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float z = b1u[t];
                        z = fmaf(w1u[t][0], xq.x, z);
                        if (NS > 1) z = fmaf(w1u[t][NS > 1 ? 1 : 0], xq.y, z);
                        if (NS > 2) z = fmaf(w1u[t][NS > 2 ? 2 : 0], xq.z, z);
                        if (NS > 3) z = fmaf(w1u[t][NS > 3 ? 3 : 0], xq.w, z);
                        const float hv = act_fwd_t<ACT>(z);
                        h8[t][i] = hv;
                        dact[ACT == 0 ? 0 : (t * 16 + q)] = 1.0f - hv * hv;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) XHw[(t * 2 + s) * 64] = as_frag(h8[t]);
            }
        }
        T3_STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // every wave's H1Y / dZ2a fragments of this tile are in the slabs
        T3_STAMP(5);

        // ---- (4) ONE scheduling region: the 32 MFMAs of dW2 (wave w owns the columns j = 32 w .. 32 w + 31 for the samples
        // of all four waves) and the VALU epilogue of dH1a: dz1 = dH1a * act'(z1), dW1 / db1 per lane ----
        {
            const bf16x8* XH = reinterpret_cast<const bf16x8*>(l_XH) + lane;
            const bf16x8* XD = reinterpret_cast<const bf16x8*>(l_XD) + lane;
            // MFMA m = 0 .. 31 <-> (source wave v, k-step s, row tile tu) = (m >> 3, (m >> 2) & 1, m & 3); in its shadow the
            // two elements (tu' = m >> 3, q = 2 (m & 7), q + 1) of the dH1a epilogue.  Operands of step m + 1 are requested
            // at the top of step m.
            bf16x8 acur = XH[0], anxt = acur, bcur = XD[((0 * 4 + w) * 2 + 0) * 64], bnxt = bcur;
            float4 x0c = xs[t3_row(0, kb)], x1c = xs[t3_row(1, kb)], x0n = x0c, x1n = x1c;
            uint4 hbc = make_uint4(0u, 0u, 0u, 0u), hbn = hbc;
            if (ACT == 0) hbc = *reinterpret_cast<const uint4*>(&XHw[0]);
            hbn = hbc;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                if (m + 1 < 32) {
                    const int m1 = m + 1, v1 = m1 >> 3, s1 = (m1 >> 2) & 1, tu1 = m1 & 3;
                    anxt = XH[((v1 * 4 + tu1) * 2 + s1) * 64];
                    if ((m1 & 3) == 0) {
                        bnxt = XD[((v1 * 4 + w) * 2 + s1) * 64];
                        if (ACT == 0) hbn = *reinterpret_cast<const uint4*>(&XHw[(v1 * 2 + s1) * 64]);
                    }
                    x0n = xs[t3_row(2 * (m1 & 7), kb)];
                    x1n = xs[t3_row(2 * (m1 & 7) + 1, kb)];
                }
                accW2[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur, bcur, accW2[m & 3], 0, 0, 0);
                {
                    const int tu = m >> 3;
                    const uint32_t hw = (m & 3) == 0 ? hbc.x : ((m & 3) == 1 ? hbc.y : ((m & 3) == 2 ? hbc.z : hbc.w));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int q = 2 * (m & 7) + e;
                        float dz;
                        if (ACT == 0) {
                            // relu: act'(z1) = [h1 > 0], read off the bf16 h1 fragments this wave put into its slab
                            // (rounding to bf16 keeps the sign and cannot reach zero from a normal f32)
                            const uint32_t bits = e ? (hw >> 16) : (hw & 0xFFFFu);
                            dz = bits != 0u ? dh1[tu][q] : 0.0f;
                        } else {
                            dz = dh1[tu][q] * dact[ACT == 0 ? 0 : (tu * 16 + q)];
                        }
                        const float4 xq = e ? x1c : x0c;
                        const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
                        gb1[tu] += dz;
#pragma unroll
                        for (int k = 0; k < NS; ++k) gW1[k][tu] = fmaf(dz, xv[k], gW1[k][tu]);
                    }
                }
                acur = anxt;
                bcur = bnxt;
                hbc = hbn;
                x0c = x0n;
                x1c = x1n;
            }
        }
        T3_STAMP(10);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // every wave has read the exchange slabs (and xs / dl of this tile) before the next tile's writes
        T3_STAMP(11);
    }

#ifdef RLHIP_T3_TIMING
    if (!CRITIC && wg == 0 && tid == 0) g_t3_stamps[13] = clock64();
#endif
    // ---- write this workgroup's partial gradient (parameter layout of the net) ----
    float* out = g.partials + (int64_t)wg * g.np + (CRITIC ? g.np_a : 0);
    const int ob1 = H3 * NS, oW2 = ob1 + H3, ob2 = oW2 + H3 * H3, oW3 = ob2 + H3, ob3 = oW3 + NOUT * H3;
    // dW2: D[u][j], lanes = j (32 w + c), regs = u (32 tu + row(q, kb)); Flux layout W2[j + H u]
#pragma unroll
    for (int tu = 0; tu < 4; ++tu)
#pragma unroll
        for (int q = 0; q < 16; ++q) store_wt(&out[oW2 + (32 * w + c) + H3 * (32 * tu + t3_row(q, kb))], accW2[tu][q]);
    // per-lane sums: the two halves of a wave hold different samples, the four waves as well -> LDS, fixed order
    constexpr int NV = 2 + NOUT + NS;  // db2, db1, dW3[o], dW1[i]
    float* l_red = reinterpret_cast<float*>(l_XH);  // [4 waves][NV][H3]  (the exchange slabs are free now)
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
    float* l_small = l_red + 4 * NV * H3;  // [4][8]
    if (lane == 0)
#pragma unroll
        for (int o = 0; o < NOUT + 2; ++o) l_small[w * 8 + o] = small[o];
    __syncthreads();
    if (tid < H3) {
        const int u = tid;
        auto sum4 = [&](int v) {
            return ((l_red[(0 * NV + v) * H3 + u] + l_red[(1 * NV + v) * H3 + u]) + l_red[(2 * NV + v) * H3 + u]) +
                   l_red[(3 * NV + v) * H3 + u];
        };
        store_wt(&out[ob2 + u], sum4(0));
        store_wt(&out[ob1 + u], sum4(1));
#pragma unroll
        for (int o = 0; o < NOUT; ++o) store_wt(&out[oW3 + o + NOUT * u], sum4(2 + o));
#pragma unroll
        for (int i = 0; i < NS; ++i) store_wt(&out[u + H3 * i], sum4(2 + NOUT + i));
    }
    if (tid == 0) {
        auto s4 = [&](int o) { return ((l_small[o] + l_small[8 + o]) + l_small[16 + o]) + l_small[24 + o]; };
#pragma unroll
        for (int o = 0; o < NOUT; ++o) store_wt(&out[ob3 + o], s4(o));
        float* lo = g.loss_partials + (int64_t)wg * 4;
        if (!CRITIC) {
            store_wt(&lo[0], s4(NOUT));
            store_wt(&lo[2], s4(NOUT + 1));
        } else {
            store_wt(&lo[1], s4(NOUT));
        }
    }
}

template <int NS, int ACT, int CONT>
__global__ __launch_bounds__(256, 1) void ppo3_gradT_kernel(P3Args g, int nwg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smemT[];
    const int net = blockIdx.x & 1, wg = blockIdx.x >> 1;
    if (net == 0) ppo3T_body<NS, 2, ACT, CONT, 0>(g, wg, nwg, ntiles, smemT);
    else ppo3T_body<NS, 1, ACT, 0, 1>(g, wg, nwg, ntiles, smemT);
}

}  // namespace rlhip
