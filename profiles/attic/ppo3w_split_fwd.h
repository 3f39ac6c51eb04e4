// profiles/attic/ppo3w_split_fwd.h -- ARCHIVED (round 4, not part of the product): the "split forward" of the 256-wide PPO
// learner.  Built to test the hypothesis that ppo3w_fwd_kernel (one 8-wave workgroup per CU, 194 - 212 VGPRs, 124 KB of LDS) is
// latency-bound and wants more waves per SIMD: the forward was split in two kernels that both fit TWO workgroups per CU
// (128 VGPRs, <= 79 KB of LDS; hipOccupancyMaxActiveBlocksPerMultiprocessor = 2), the second recomputing layer 1 and the
// MFMA.  Parity: green on the first run (tests/test_gpu_ppo3w.py + tests/test_gpu_bf16_tight.py, 43 tests).  Speed: it LOSES --
//   one-kernel forward (shipped)   actor 46 us + critic 40 us = 86 us per optimiser step, step 211 us
//   split forward                  A 43.0 / 26.7 us + B 35.5 / 33.1 us = 138 us, step 270 us (first version with 83 - 131
//                                  spilled registers in B: 255 us of forward, step 384 us)
// PMC (per launch): VALU instructions per wave and tile 309 (A) + 536 (B) = 845 against 651; MFMA busy cycles doubled;
// SQ_WAIT_ANY 68 % (A actor: every wave waits for wave 0's loss line at the next tile's barrier) / 42 % (B); doubling the
// resident workgroups bought ~1.2 x per-CU throughput per kernel, not 2 x.  profiles/r04_ppo3w.md has the table and the reading.
// The text below is the kernel pair as measured (it compiled inside csrc/ppo3w.hip next to ppo3w_fwd_kernel).
// ------------------------------------------------------------------------------------------------ split forward (round 4)
// ppo3w_fwd_kernel is bound by latency, not by its instruction streams: one 8-wave workgroup per CU (194 - 212 VGPRs, 124 KB
// of LDS) = 2 waves per SIMD, ~650 VALU instructions + 32 MFMAs per wave and tile against ~12 600 cycles per tile (the VALU
// issue floor of two such waves is ~3 900 cycles, the matrix pipe's 2 048: profiles/r04_ppo3w.md).  What the kernel needs is
// MORE WAVES PER SIMD, which its register and LDS footprints forbid: the W2 fragments (64), the accumulators of both 32-row
// halves (32) that must survive the loss line's barriers, and the 74 KB transposition block.  So the PPO learner's forward
// is split in two kernels that both fit TWO workgroups per CU (<= 128 VGPRs, <= 80 KB of LDS):
//   ppo3w_fwdA_kernel   layer 1 -> H1 tile -> per 32-row half: MFMA -> H2 -> head partial sums through HALF a transposition
//                       block -> loss line (wave 0) -> dL/d(head outputs) to GLOBAL memory (NOUT floats per sample), the loss
//                       sums and db3; nothing of H2 survives the half's head sums: 16 accumulators
//   ppo3w_fwdB_kernel   layer 1 and the MFMA AGAIN (same instructions on the same operands: the same H2 bit for bit) ->
//                       dZ2 = (W3^T dL/dout) .* act'(H2) per half, with dL/dout of the tile read back from global memory ->
//                       db2 / dW3 in registers across tiles -> bf16 rows + fragments as before; no loss line, one transposition
//                       half-block of bf16 (20 KB)
// The recomputation costs ~150 VALU instructions and 32 MFMAs per wave and tile; it buys four waves per SIMD.
// Head sums: a lane adds its 16-column half of the wave's 32 columns, the two halves are joined by one lane-crossing add (the
// one-kernel version walked all 32 columns in one lane) -- a different f32 summation order of the same terms.
constexpr int TPH = 36;  // f32 pitch of a wave's private 32 x 32 transposition half-block
constexpr int ZPH = 40;  // bf16 pitch of the same half-block when it holds the wave's dZ2 columns
template <int NOUT>
constexpr size_t fwda_lds() {
    return (size_t)RW * PW * sizeof(uint16_t) + ((size_t)WV * 32 * TPH + WV * NOUT * RW + HW + NOUT * HW + MAXO + HW) * sizeof(float);
}
template <int NOUT>
constexpr size_t fwdb_lds() {
    return (size_t)RW * PW * sizeof(uint16_t) + (size_t)WV * 32 * ZPH * sizeof(uint16_t) + ((size_t)2 * NOUT * RW + HW * 4 + HW) * sizeof(float);
}

template <int NS, int NOUT, int ACT, int CONT, int NET>
__global__ __launch_bounds__(NTW, 4) void ppo3w_fwdA_kernel(P3WArgs g) {
    static_assert(NET == 0 || NET == 1, "PPO actor / critic");
    extern __shared__ __attribute__((aligned(16))) char smw[];
    uint16_t* l_H = reinterpret_cast<uint16_t*>(smw);                 // [RW][PW] H1 rows
    float* l_t = reinterpret_cast<float*>(l_H + RW * PW);            // [WV][32][TPH] transposition half-blocks
    float* l_part = l_t + WV * 32 * TPH;                              // [WV][NOUT][RW] head partial sums per wave
    float* l_w2 = l_part + WV * NOUT * RW;                            // b2 [HW] | W3 [NOUT * HW] | b3 [MAXO] | b1 [HW]
    float* l_w1 = l_t;                                                // prologue only: W1 | b1 (dead before the first half-block write)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col = 32 * w + r;
    float* l_tw = l_t + w * 32 * TPH;
    const float* pnet = g.params + (NET == 1 ? g.np_a : 0);
    bf16x8 bw[KSW];
    load_frags_w(g.packed + (NET == 1 ? 2 * HW * HW : 0), w, lane, bw);
    const int stride = gridDim.x, last = g.ntiles - 1;
    constexpr int KS1 = (NS + 1) / 2;
    float xr[2][KS1], sr0, sr1 = 0.0f, sr2 = 0.0f;
    load_xm_from<NS>(g.xg, g.npad, blockIdx.x, lane, xr);
    auto load_s = [&](int tile_) __attribute__((always_inline)) {
        const int64_t q0 = (int64_t)tile_ * RW + lane;
        if (NET == 0) {
            sr0 = g.sg[q0];
            sr1 = g.sg[(int64_t)g.npad + q0];
            sr2 = g.sg[3 * (int64_t)g.npad + q0];
        } else {
            sr0 = g.sg[2 * (int64_t)g.npad + q0];
        }
    };
    load_s(blockIdx.x);
    constexpr int n1 = HW * NS + HW, n2 = HW + NOUT * HW + NOUT;
    copy_to_lds<n1>(pnet, l_w1, tid);
    copy_to_lds<n2>(pnet + n1 + HW * HW, l_w2, tid);
    __syncthreads();
    const float b2v = l_w2[col];
    const float* W3l = l_w2 + HW;
    const float* b3l = W3l + NOUT * HW;
    float s_red[NOUT + 2];
#pragma unroll
    for (int o = 0; o < NOUT + 2; ++o) s_red[o] = 0.0f;
    const int u0 = 32 * w;
    float w1a[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int kk = 2 * ks + kb;
        const float v = l_w1[u0 + r + HW * min(kk, NS - 1)];
        w1a[ks] = kk < NS ? v : 0.0f;
    }
    // b1 (the layer-1 accumulator's initial value) stays in LDS, in its own slot: four 16-byte reads per 32-row half instead
    // of 16 resident registers (the 128-register budget of four waves per SIMD)
    float* l_b1 = l_w2 + HW + NOUT * HW + MAXO;
    if (tid < HW) l_b1[tid] = l_w1[HW * NS + tid];
    const float* b1l = l_b1 + u0 + 4 * kb;
    __syncthreads();  // every wave has its layer-1 operands: l_w1 (= the half-blocks) may be overwritten
    float* dqg = g.dq + (NET == 1 ? (int64_t)MAXO * g.npad : 0);
    for (int tile = blockIdx.x; tile < g.ntiles; tile += stride) {
        const int tnext = min(tile + stride, last);
        {
            float x[2][KS1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) x[rt][ks] = xr[rt][ks];
            load_xm_from<NS>(g.xg, g.npad, tnext, lane, xr);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16 z;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bq = *reinterpret_cast<const float4*>(b1l + 8 * g4);
                    z[4 * g4 + 0] = bq.x;
                    z[4 * g4 + 1] = bq.y;
                    z[4 * g4 + 2] = bq.z;
                    z[4 * g4 + 3] = bq.w;
                }
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
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 h2;
#pragma unroll
            for (int q = 0; q < 16; ++q) h2[q] = 0.0f;
            const uint16_t* ap = l_H + (32 * rt + r) * PW + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
                h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], h2, 0, 0, 0);
            }
            // this wave's 32 x 32 block of H2 (rows of this half x its columns) through its private half-block: D layout in,
            // lane (r, kb) out = row r, columns 16 kb .. 16 kb + 15
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) l_tw[mfma_row(q, kb) * TPH + r] = act_fwd_t<ACT>(h2[q] + b2v);
            wave_lds_fence();
            float pa[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) pa[o] = 0.0f;
            const float* hrow = l_tw + r * TPH + 16 * kb;
            const float* w3p = W3l + NOUT * (32 * w + 16 * kb);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(hrow + 4 * c4);
                const float hv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) pa[o] = fmaf(w3p[o + NOUT * (4 * c4 + e)], hv[e], pa[o]);
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float other = __shfl_xor(pa[o], 32, 64);
                const float tot = kb == 0 ? pa[o] + other : other + pa[o];  // columns 0..15 + columns 16..31, on both lanes
                if (kb == 0) l_part[(w * NOUT + o) * RW + 32 * rt + r] = tot;
            }
        }
        __syncthreads();  // C: every wave's partial sums (and: every wave is done reading l_H)
        if (tid < RW) {
            const int sidx = tid;
            const bool valid = ((uint32_t)tile * RW + (uint32_t)sidx) < g.bm;
            float oa[MAXO] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float acc = l_part[o * RW + sidx];
#pragma unroll
                for (int ww = 1; ww < WV; ++ww) acc += l_part[(ww * NOUT + o) * RW + sidx];
                oa[o] = acc + b3l[o];
            }
            float dl[MAXO] = {0.f, 0.f, 0.f, 0.f};
            if (NET == 0) {
                const float lp_old = fmaxf(sr0, g.min_logp);
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
                for (int o = 0; o < NOUT; ++o) s_red[o] += dl[o];
                s_red[NOUT] += surr_min;
                s_red[NOUT + 1] += ent;
            } else {
                const float dv = sr0 - oa[0];
                float dvout = -2.0f * g.wc * g.inv_b * dv;
                float sq = dv * dv;
                if (!valid) {
                    dvout = 0.f;
                    sq = 0.f;
                }
                dl[0] = dvout;
                s_red[0] += dvout;
                s_red[NOUT] += sq;
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dqg[(int64_t)o * g.npad + (int64_t)tile * RW + sidx] = dl[o];
        }
        load_s(tnext);
        // no barrier here: the next pass writes l_H (last read before barrier C), the half-blocks (wave-private) and l_part
        // (after the next barrier A, which wave 0 reaches only after this loss line)
    }
    if (w == 0) {
        const int sb2 = HW * NS + HW, sW3 = sb2 + HW, sb3 = sW3 + NOUT * HW;
        float* rowS = g.partS + (int64_t)blockIdx.x * g.npS + (NET == 1 ? g.nS_a : 0);
#pragma unroll
        for (int o = 0; o < NOUT + 2; ++o) s_red[o] = wave_sum_f32(s_red[o]);
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) rowS[sb3 + o] = s_red[o];
            float* lp = g.loss_partials + (int64_t)blockIdx.x * 4;
            if (NET == 0) {
                lp[0] = s_red[NOUT];
                lp[2] = s_red[NOUT + 1];
            } else {
                lp[1] = s_red[NOUT];
            }
        }
    }
}

template <int NS, int NOUT, int ACT, int NET>
__global__ __launch_bounds__(NTW, 4) void ppo3w_fwdB_kernel(P3WArgs g) {
    static_assert(NET == 0 || NET == 1, "PPO actor / critic");
    extern __shared__ __attribute__((aligned(16))) char smw[];
    uint16_t* l_H = reinterpret_cast<uint16_t*>(smw);                 // [RW][PW] H1 rows
    uint16_t* l_z = l_H + RW * PW;                                    // [WV][32][ZPH] this half's dZ2 columns of each wave
    float* l_dq2 = reinterpret_cast<float*>(l_z + WV * 32 * ZPH);     // [2][NOUT][RW] dL/d(head outputs) of the tile, double-buffered
    float* l_w1 = l_dq2 + 2 * NOUT * RW;                              // W1 | b1 (prologue), then unused
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kb = lane >> 5;
    const int col = 32 * w + r;
    uint16_t* l_zw = l_z + w * 32 * ZPH;
    const float* pnet = g.params + (NET == 1 ? g.np_a : 0);
    bf16x8 bw[KSW];
    load_frags_w(g.packed + (NET == 1 ? 2 * HW * HW : 0), w, lane, bw);
    const int stride = gridDim.x, last = g.ntiles - 1;
    constexpr int KS1 = (NS + 1) / 2;
    const float* dqg = g.dq + (NET == 1 ? (int64_t)MAXO * g.npad : 0);
    float xr[2][KS1], dqr[NOUT];
    load_xm_from<NS>(g.xg, g.npad, blockIdx.x, lane, xr);
    auto load_dq = [&](int tile_) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) dqr[o] = dqg[(int64_t)o * g.npad + (int64_t)tile_ * RW + lane];
    };
    load_dq(blockIdx.x);
    constexpr int n1 = HW * NS + HW;
    copy_to_lds<n1>(pnet, l_w1, tid);
    const float* p2 = pnet + n1 + HW * HW;  // b2 | W3 | b3
    const float b2v = p2[col];
    float w3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) w3[o] = p2[HW + o + NOUT * col];
    __syncthreads();
    const int u0 = 32 * w;
    float w1a[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int kk = 2 * ks + kb;
        const float v = l_w1[u0 + r + HW * min(kk, NS - 1)];
        w1a[ks] = kk < NS ? v : 0.0f;
    }
    // b1 as the layer-1 accumulator's initial value: re-read from LDS per 32-row half (four 16-byte reads) instead of 16
    // resident registers -- the 128-register budget of four waves per SIMD
    const float* b1l = l_w1 + HW * NS + u0 + 4 * kb;
    float a_db2 = 0.0f, a_dw3[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a_dw3[o] = 0.0f;
    int it = 0;
    for (int tile = blockIdx.x; tile < g.ntiles; tile += stride, ++it) {
        const int tnext = min(tile + stride, last);
        // wave 0 fills buffer it & 1 before barrier A of this pass; its last readers (pass it - 2) are behind barrier A of pass
        // it - 1, which wave 0 has passed as well: no third barrier
        float* l_dq = l_dq2 + (it & 1) * NOUT * RW;
        {
            float x[2][KS1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) x[rt][ks] = xr[rt][ks];
            load_xm_from<NS>(g.xg, g.npad, tnext, lane, xr);
            if (w == 0) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) l_dq[o * RW + lane] = dqr[o];
            }
            load_dq(tnext);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16 z;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bq = *reinterpret_cast<const float4*>(b1l + 8 * g4);
                    z[4 * g4 + 0] = bq.x;
                    z[4 * g4 + 1] = bq.y;
                    z[4 * g4 + 2] = bq.z;
                    z[4 * g4 + 3] = bq.w;
                }
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
        __syncthreads();  // A: the H1 tile and dL/dout of the tile are in LDS
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 h2;
#pragma unroll
            for (int q = 0; q < 16; ++q) h2[q] = 0.0f;
            const uint16_t* ap = l_H + (32 * rt + r) * PW + 8 * kb;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
                h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw[ks], h2, 0, 0, 0);
            }
            if (rt == 1) __syncthreads();  // E: every wave is done reading l_H (the next pass's layer 1 may overwrite it)
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = 32 * rt + mfma_row(q, kb);
                float dqv[NOUT];
#pragma unroll
                for (int o = 0; o < NOUT; ++o) dqv[o] = l_dq[o * RW + row];
                const float hv = act_fwd_t<ACT>(h2[q] + b2v);
                float dh = 0.0f;
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    a_dw3[o] = fmaf(dqv[o], hv, a_dw3[o]);
                    dh = fmaf(dqv[o], w3[o], dh);
                }
                const float dz = dh * act_bwd_t<ACT>(hv, hv);  // relu: h2 > 0 <=> z2 > 0
                a_db2 += dz;
                h2[q] = dz;
                l_zw[mfma_row(q, kb) * ZPH + r] = f32_to_bf16_rne(dz);
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // rows in program order: no 32 hoisted LDS reads
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 v2;
                v2.x = pack2_bf16(h2[4 * gq + 0], h2[4 * gq + 1]);
                v2.y = pack2_bf16(h2[4 * gq + 2], h2[4 * gq + 3]);
                const int64_t slot = (((int64_t)tile * (RW / 16) + 2 * rt + (gq >> 1)) * WV + w) * 64 + 32 * (gq & 1) + r;
                *reinterpret_cast<uint2*>(g.dz_frag + (NET == 1 ? g.frag_stride : 0) + slot * 8 + 4 * kb) = v2;
            }
            wave_lds_fence();
            {
                uint16_t* dst = g.dz_rows + ((int64_t)tile * RW + 32 * rt) * HW + 32 * w;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = lane + 64 * i, row = c >> 2, cc = c & 3;
                    *reinterpret_cast<uint4*>(dst + row * HW + 8 * cc) = *reinterpret_cast<const uint4*>(l_zw + row * ZPH + 8 * cc);
                }
            }
        }
    }
    const int sb2 = HW * NS + HW, sW3 = sb2 + HW;
    float* rowS = g.partS + (int64_t)blockIdx.x * g.npS + (NET == 1 ? g.nS_a : 0);
    a_db2 += __shfl_xor(a_db2, 32, 64);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a_dw3[o] += __shfl_xor(a_dw3[o], 32, 64);
    if (kb == 0) {
        rowS[sb2 + col] = a_db2;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) rowS[sW3 + o + NOUT * col] = a_dw3[o];
    }
}

