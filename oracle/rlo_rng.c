/* rlo_rng.c -- Philox4x32-10 counter-based RNG specification shared (by specification, not by
 * code) with the HIP kernels.  TEST INFRASTRUCTURE ONLY (see rl_oracle.h).
 *
 * Why Philox and not Julia's streams: the reference draws from `rng::AbstractRNG` fields
 * (RLEnvs/CartPoleEnv.jl:54,99,101; RLCore/policies/explorers/epsilon_greedy_explorer.jl:44,105-111),
 * i.e. Xoshiro256++/MersenneTwister/StableRNG sequential streams that cannot be reproduced without
 * Julia and cannot be advanced by 4096 lanes in parallel.  Parity for random draws is therefore
 * defined against this specification ("parity unpinned by the reference").
 */
#include "rl_oracle.h"
#include <math.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c[0];
    uint64_t p1 = (uint64_t)PHILOX_M1 * c[2];
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c[1] ^ k[0];
    uint32_t n2 = hi0 ^ c[3] ^ k[1];
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
}

void rlo_philox4x32_10(uint64_t seed, uint32_t idx, uint32_t blk, uint32_t t, uint32_t tag,
                       uint32_t out[4]) {
    uint32_t c[4] = {idx, blk, t, tag};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += PHILOX_W0;
        k[1] += PHILOX_W1;
    }
    out[0] = c[0];
    out[1] = c[1];
    out[2] = c[2];
    out[3] = c[3];
}

float rlo_u01_f32(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

double rlo_u01_f64(uint32_t w_hi, uint32_t w_lo) {
    uint64_t x = ((uint64_t)w_hi << 32) | (uint64_t)w_lo;
    return (double)(x >> 11) * 0x1p-53;
}

uint32_t rlo_randint(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * (uint64_t)n) >> 32); }

static inline uint32_t feistel_f(uint32_t r, uint32_t k) {
    uint32_t f = r * 0x9E3779B1u + k;
    f ^= f >> 15;
    f *= 0x85EBCA77u;
    f ^= f >> 13;
    f *= 0xC2B2AE3Du;
    f ^= f >> 16;
    return f;
}

uint32_t rlo_permute(uint64_t seed, uint32_t epoch, uint32_t n, uint32_t i) {
    if (n <= 1) return 0;
    /* half-width h: smallest h with 4^h >= n */
    uint32_t h = 1;
    while (h < 16 && (1ull << (2 * h)) < (uint64_t)n) ++h;
    uint32_t mask = (h == 16) ? 0xFFFFFFFFu >> 16 : ((1u << h) - 1u);
    uint32_t k[8];
    rlo_philox4x32_10(seed, 0, 0, epoch, RLO_TAG_SHUFFLE, k);
    rlo_philox4x32_10(seed, 0, 1, epoch, RLO_TAG_SHUFFLE, k + 4);
    uint32_t x = i;
    do {
        uint32_t L = (x >> h) & mask, R = x & mask;
        for (int r = 0; r < 6; ++r) {
            uint32_t nl = R;
            uint32_t nr = L ^ (feistel_f(R, k[r]) & mask);
            L = nl;
            R = nr;
        }
        x = (L << h) | R;
    } while (x >= n);
    return x;
}

void rlo_fill_uniform_f32(float* out, int64_t n, uint64_t seed, uint32_t t, uint32_t tag) {
    for (int64_t i = 0; i < n; i += 4) {
        uint32_t w[4];
        rlo_philox4x32_10(seed, (uint32_t)(i / 4), 0, t, tag, w);
        for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = rlo_u01_f32(w[k]);
    }
}

void rlo_normal_pair_f32(uint32_t w0, uint32_t w1, float* z0, float* z1) {
    /* evaluated in Float64 and rounded once, so CPU libm and GPU ocml agree bit for bit */
    double u1 = (double)((w0 >> 8) + 1u) * 0x1p-24; /* (0, 1] */
    double u2 = (double)(w1 >> 8) * 0x1p-24;        /* [0, 1) */
    double r = sqrt(-2.0 * log(u1));
    double a = 6.283185307179586 * u2;
    *z0 = (float)(r * cos(a));
    *z1 = (float)(r * sin(a));
}
