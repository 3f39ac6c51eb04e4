// adam_stream.hip -- where does the time of the streaming Adam launch go?  (Adam at 2^26 parameters sits at
// 0.69 of the HBM peak while Polyak -- 3 streams instead of 7 -- holds 0.87; PMC traffic ratio 1.0000, so it is not re-reads.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/adam_stream.hip -o tools/micro/adam_stream.bin
//   tools/micro/adam_stream.bin [log2 n = 26]
//
// Variants of csrc/optim.hip's adam_vec4_kernel on the bench's layout (p, g, m, v carved from one allocation, 4352-byte stagger),
// each timed with HIP events over 20 launches after 10 warm-up launches:
//   K0  the shipped kernel body (adam1: three IEEE divisions + IEEE sqrt per element), NT loads + NT stores, one 16-byte chunk per lane
//   K1  same seven streams, NO math (p += g, m += g, v += g): the memory pattern alone
//   K2  K0 with the two divisions by the launch-uniform c1, c2 as Markstein quotients (x * r, one FMA remainder, one FMA correction,
//       r = RN(1 / c) once per thread; IEEE path for zero / tiny / non-finite numerators): BIT-IDENTICAL to K0 (checked here on every element)
//   K3  approximate math (v_rcp / v_rsq, no fix-ups): NOT bit-identical -- the floor of what cheaper arithmetic could buy
//   K4  K0 with ordinary (temporal) loads
//   K5  K0, two ADJACENT 16-byte chunks per lane (32 bytes per lane and stream, 8 KB contiguous per workgroup and stream)
//   K6  K1 with two chunks per lane, `stride` apart (more bytes in flight per wave)
//   K7  K2 + the third division and the square root restated on v_rcp / v_rsq with FMA corrections, IEEE fallback outside the safe
//       range: bit-identical where checked (every element of the run)
// Prints us per launch, GB/s on the 28 algorithmic bytes per parameter, the fraction of 8 TB/s, and for K2 / K7 the number of
// elements (of 3 x n) that differ from K0's result.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union V4 {
    u32x4 u;
    float f[4];
};
__device__ __forceinline__ u32x4 ntl(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ void nts(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

// ---- the element updates
__device__ __forceinline__ void adam_ieee(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float c1, float c2) {
    float mi = b1 * m + (1.0f - b1) * g;
    float vi = b2 * v + (1.0f - b2) * (g * g);
    m = mi;
    v = vi;
    float d = mi / c1 / (sqrtf(vi / c2) + eps) * lr;
    p = p - d;
}
__device__ __forceinline__ void adam_nomath(float& p, float g, float& m, float& v) {
    p += g;
    m += g;
    v += g;
}
// x / c for a launch-uniform c with r = RN(1 / c): q = RN(x r); e = x - q c (exact: one FMA); RN(q + e r) = RN(x / c) (Markstein 1990,
// Theorem: valid for every x whose quotient and remainder stay in the normal range, for every c whose significand is not all ones --
// the caller takes the IEEE route for such a c; numerators outside [2^-40, 2^40) incl. 0 / inf / nan go through the IEEE division).
__device__ __forceinline__ bool safe_num(float x) {  // |x| in [2^-40, 2^40): quotients of two such numbers stay 2^+-80 -- inside the range
    const uint32_t e = (__float_as_uint(x) >> 23) & 0xffu;  // in which v_div_scale_f32 does not rescale and nothing over- / underflows
    return e >= 87u && e < 167u;
}
__device__ __forceinline__ float div_uniform(float x, float c, float r) {
    if (!safe_num(x)) return x / c;
    const float q = x * r;
    const float e = fmaf(-q, c, x);
    return fmaf(e, r, q);
}
__device__ __forceinline__ void adam_markstein(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float c1, float c2,
                                               float r1, float r2) {
    float mi = b1 * m + (1.0f - b1) * g;
    float vi = b2 * v + (1.0f - b2) * (g * g);
    m = mi;
    v = vi;
    float d = div_uniform(mi, c1, r1) / (sqrtf(div_uniform(vi, c2, r2)) + eps) * lr;
    p = p - d;
}
__device__ __forceinline__ void adam_approx(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float r1, float r2) {
    float mi = b1 * m + (1.0f - b1) * g;
    float vi = b2 * v + (1.0f - b2) * (g * g);
    m = mi;
    v = vi;
    float s = vi * r2;
    float d = mi * r1 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(s) + eps) * lr;
    p = p - d;
}
// correctly rounded sqrt of a normal-range x from v_rsq: y ~ 1/sqrt(x); s = x y; two FMA corrections (the classical scheme: h = y / 2;
// e = x - s s; s += e h), then one more remainder step decides the last bit.  Outside [2^-40, 2^40): sqrtf.
__device__ __forceinline__ float sqrt_fast(float x) {
    const uint32_t ex = (__float_as_uint(x) >> 23);  // sign + exponent: negative / nan / inf / tiny all fail the range test
    if (ex < 87u || ex >= 167u) return sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    float e = fmaf(-s, s, x);
    s = fmaf(e, h, s);
    e = fmaf(-s, s, x);
    return fmaf(e, h, s);
}
// a / b for a per-element b > 0 in the normal range: r0 = v_rcp(b) (1 ulp), one Newton step, then Markstein's quotient correction twice
__device__ __forceinline__ float div_fast(float a, float b) {
    if (!safe_num(a) || !safe_num(b)) return a / b;
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    float e = fmaf(-q, b, a);
    q = fmaf(e, r, q);
    e = fmaf(-q, b, a);
    return fmaf(e, r, q);
}
__device__ __forceinline__ void adam_fast(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float c1, float c2,
                                          float r1, float r2) {
    float mi = b1 * m + (1.0f - b1) * g;
    float vi = b2 * v + (1.0f - b2) * (g * g);
    m = mi;
    v = vi;
    float d = div_fast(div_uniform(mi, c1, r1), sqrt_fast(div_uniform(vi, c2, r2)) + eps) * lr;
    p = p - d;
}

template <int K>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              const float* __restrict__ beta_pow, int64_t n4, float lr, float b1, float b2, float eps) {
    const float c1 = 1.0f - beta_pow[0], c2 = 1.0f - beta_pow[1];
    const float r1 = 1.0f / c1, r2 = 1.0f / c2;
    const bool odd_c = ((__float_as_uint(c1) & 0x7fffffu) == 0x7fffffu) || ((__float_as_uint(c2) & 0x7fffffu) == 0x7fffffu);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int U = (K == 5 || K == 6) ? 2 : 1;
    V4 pi[U], gi[U], mi[U], vi[U];
    int64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        idx[u] = K == 5 ? 2 * t + u : t + u * stride;
        if (idx[u] < n4) {
            if (K == 4) {
                pi[u].u = *reinterpret_cast<const u32x4*>(p + 4 * idx[u]);
                gi[u].u = *reinterpret_cast<const u32x4*>(g + 4 * idx[u]);
                mi[u].u = *reinterpret_cast<const u32x4*>(m + 4 * idx[u]);
                vi[u].u = *reinterpret_cast<const u32x4*>(v + 4 * idx[u]);
            } else {
                pi[u].u = ntl(p + 4 * idx[u]);
                gi[u].u = ntl(g + 4 * idx[u]);
                mi[u].u = ntl(m + 4 * idx[u]);
                vi[u].u = ntl(v + 4 * idx[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (idx[u] < n4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (K == 1 || K == 6) adam_nomath(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k]);
                else if (K == 2) {
                    if (odd_c) adam_ieee(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2);
                    else adam_markstein(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2, r1, r2);
                } else if (K == 7) {
                    if (odd_c) adam_ieee(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2);
                    else adam_fast(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2, r1, r2);
                } else if (K == 3) adam_approx(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, r1, r2);
                else adam_ieee(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2);
            }
            nts(p + 4 * idx[u], pi[u].u);
            nts(m + 4 * idx[u], mi[u].u);
            nts(v + 4 * idx[u], vi[u].u);
        }
    }
}

// pseudo-random finite floats (the sweeps otherwise stream the zeros of hipMemset: constant data)
__global__ void fill_random(uint32_t* p, size_t n32) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + 12345u;
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        p[i] = (h & 0x807fffffu) | 0x3f000000u;  // +-[0.5, 1)
    }
}

// every x of a 2^32 sweep: Markstein quotient by c against the IEEE division
__global__ void sweep_div(float c, unsigned long long* bad) {
    const float r = 1.0f / c;
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = x / c, b = div_uniform(x, c, r);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) ++local;
    }
    if (local) atomicAdd(bad, local);
}
__global__ void sweep_sqrt(unsigned long long* bad) {
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = sqrtf(x), b = sqrt_fast(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) ++local;
    }
    if (local) atomicAdd(bad, local);
}
// a / b over 2^32 numerators for a handful of denominators of the shape sqrt(v) + eps
__global__ void sweep_divfast(float b, unsigned long long* bad) {
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = x / b, f = div_fast(x, b);
        if (__float_as_uint(a) != __float_as_uint(f) && !(a != a && f != f)) ++local;
    }
    if (local) atomicAdd(bad, local);
}

// ---- the memory pattern alone, by stream mix: R read-only + W in-place (read, then written) + X write-only arrays of n floats, no
// arithmetic beyond one add per value; NT loads / NT stores, one 16-byte chunk per lane and array, one trip per lane (the shipped shape)
struct Ptrs {
    float* a[12];
};
template <int R, int W, int X, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void mix_k(Ptrs ps, int64_t n4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n4) return;
    V4 in[R + W > 0 ? R + W : 1];
#pragma unroll
    for (int k = 0; k < R + W; ++k) in[k].u = NTL ? ntl(ps.a[k] + 4 * t) : *reinterpret_cast<const u32x4*>(ps.a[k] + 4 * t);
    V4 acc;
    acc.u = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc.f[c] += in[k].f[c];
#pragma unroll
    for (int k = 0; k < W + X; ++k) {
        V4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o.f[c] = (k < W ? in[R + k].f[c] : 1.0f) + acc.f[c];
        if (NTS) nts(ps.a[R + k] + 4 * t, o.u);
        else *reinterpret_cast<u32x4*>(ps.a[R + k] + 4 * t) = o.u;
    }
}
template <int R, int W, int X, bool NTL = true, bool NTS = true>
static void time_mix(uint8_t* buf, size_t pitch, int64_t n, const char* what) {
    Ptrs ps;
    for (int k = 0; k < 12; ++k) ps.a[k] = (float*)(buf + pitch * k);
    const int64_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_k<R, W, X, NTL, NTS>), dim3(grid), dim3(256), 0, 0, ps, n4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mix_k<R, W, X, NTL, NTS>), dim3(grid), dim3(256), 0, 0, ps, n4);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 20 * 1e3, gb = 4.0 * n * (R + 2 * W + X) / 1e9;
    printf("mix R=%d W=%d X=%d %-34s pitch-2^k %7zu B  %8.2f us  %7.1f GB/s  frac %.4f\n", R, W, X, what, pitch - (size_t)n * 4, us, gb / (us * 1e-6),
           gb / (us * 1e-6) / 8000.0);
}

// ---- the same, with U chunk rows per wave that are CONTIGUOUS per array (a wave covers U KB of every array), and optionally with
// the W in-place arrays merged into ONE array of 1 KB rows [row r of array 0][row r of array 1]... (BLOCKED: the optimiser state as
// row-blocked SoA -- a wave's m and v rows are neighbours in memory, the W arrays form a single stream)
template <int R, int W, int U, bool BLOCKED>
__global__ __launch_bounds__(256) void mix2_k(Ptrs ps, int64_t nrows) {  // nrows = n / 256: 1 KB rows per array
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    V4 in[U][R + W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = wave * U + u;
        if (row < nrows) {
#pragma unroll
            for (int k = 0; k < R; ++k) in[u][k].u = ntl(ps.a[k] + row * 256 + 4 * lane);
#pragma unroll
            for (int k = 0; k < W; ++k)
                in[u][R + k].u = BLOCKED ? ntl(ps.a[R] + (row * W + k) * 256 + 4 * lane) : ntl(ps.a[R + k] + row * 256 + 4 * lane);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t row = wave * U + u;
        if (row < nrows) {
            V4 acc;
            acc.u = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < R; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc.f[c] += in[u][k].f[c];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                V4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o.f[c] = in[u][R + k].f[c] + acc.f[c];
                if (BLOCKED) nts(ps.a[R] + (row * W + k) * 256 + 4 * lane, o.u);
                else nts(ps.a[R + k] + row * 256 + 4 * lane, o.u);
            }
        }
    }
}
template <int R, int W, int U, bool BLOCKED>
static void time_mix2(uint8_t* buf, size_t pitch, int64_t n, const char* what) {
    Ptrs ps;
    for (int k = 0; k < 12; ++k) ps.a[k] = (float*)(buf + pitch * k);  // BLOCKED: a[R] spans W pitches
    const int64_t nrows = n / 256, waves = (nrows + U - 1) / U;
    const int grid = (int)((waves + 3) / 4);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix2_k<R, W, U, BLOCKED>), dim3(grid), dim3(256), 0, 0, ps, nrows);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mix2_k<R, W, U, BLOCKED>), dim3(grid), dim3(256), 0, 0, ps, nrows);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / 20 * 1e3, gb = 4.0 * n * (R + 2 * W) / 1e9;
    printf("mix2 R=%d W=%d U=%d %-8s %-40s %8.2f us  %7.1f GB/s  frac %.4f\n", R, W, U, BLOCKED ? "blocked" : "separate", what, us, gb / (us * 1e-6),
           gb / (us * 1e-6) / 8000.0);
}

template <int K>
static float time_k(float* p, float* g, float* m, float* v, float* bp, int64_t n, int iters) {
    const int64_t n4 = n / 4;
    constexpr int U = (K == 5 || K == 6) ? 2 : 1;
    const int grid = (int)((n4 / U + 255) / 256);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((adam_k<K>), dim3(grid), dim3(256), 0, 0, p, g, m, v, bp, n4, 1e-3f, 0.9f, 0.999f, 1e-8f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((adam_k<K>), dim3(grid), dim3(256), 0, 0, p, g, m, v, bp, n4, 1e-3f, 0.9f, 0.999f, 1e-8f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 26;
    const int64_t n = 1ll << logn;
    const size_t stag = 4352, bytes = (size_t)n * 4, pitch = (bytes + stag + 255) / 256 * 256;
    uint8_t* buf;
    CK(hipMalloc((void**)&buf, pitch * 8 + 256));
    float* arr[8];
    for (int k = 0; k < 8; ++k) arr[k] = (float*)(buf + pitch * k);
    float *p = arr[0], *g = arr[1], *m = arr[2], *v = arr[3], *p2 = arr[4], *m2 = arr[5], *v2 = arr[6], *g2 = arr[7];
    float* bp;
    CK(hipMalloc((void**)&bp, 8));
    // inputs: p, g, m ~ N(0,1)-ish from an LCG on the host would take too long at 2^26: fill on the device from a hash
    std::vector<float> h((size_t)1 << 20);
    uint32_t s = 12345u;
    for (auto& x : h) {
        s = s * 1664525u + 1013904223u;
        x = ((int32_t)s) * (1.0f / 2147483648.0f) * 1.7f;
    }
    for (float* a : {p, g, m}) for (size_t o = 0; o < (size_t)n; o += h.size()) CK(hipMemcpy(a + o, h.data(), std::min(h.size(), (size_t)n - o) * 4, hipMemcpyHostToDevice));
    for (auto& x : h) x = fabsf(x) * 0.5f + 0.01f;
    for (size_t o = 0; o < (size_t)n; o += h.size()) CK(hipMemcpy(v + o, h.data(), std::min(h.size(), (size_t)n - o) * 4, hipMemcpyHostToDevice));
    const double gb = 28.0 * n / 1e9;
    auto report = [&](const char* name, float us) { printf("%-44s %9.2f us  %8.1f GB/s  frac %.4f\n", name, us, gb / (us * 1e-6), gb / (us * 1e-6) / 8000.0); };
    // --- bit-identity of K2 / K7 with K0 on the run's own data, for several Adam step counts (c1, c2)
    for (int t : {1, 2, 7, 100, 5000}) {
        float hb[2] = {powf(0.9f, (float)t), powf(0.999f, (float)t)};
        CK(hipMemcpy(bp, hb, 8, hipMemcpyHostToDevice));
        for (int kk : {2, 7}) {
            CK(hipMemcpy(p2, p, bytes, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(m2, m, bytes, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(v2, v, bytes, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(g2, p, bytes, hipMemcpyDeviceToDevice));  // g2 = second copy of p for the reference run
            const int grid = (int)((n / 4 + 255) / 256);
            if (kk == 2) hipLaunchKernelGGL((adam_k<2>), dim3(grid), dim3(256), 0, 0, p2, g, m2, v2, bp, n / 4, 1e-3f, 0.9f, 0.999f, 1e-8f);
            else hipLaunchKernelGGL((adam_k<7>), dim3(grid), dim3(256), 0, 0, p2, g, m2, v2, bp, n / 4, 1e-3f, 0.9f, 0.999f, 1e-8f);
            // reference into (g2 as p, and fresh copies of m, v in arr... reuse: run K0 on copies held in host-side chunks would be slow;
            // instead run K0 in place on a THIRD set: p <- g2 path needs m, v copies -> compare p only plus m2 / v2 against a K0 run on p, m, v copies
            std::vector<float> a((size_t)1 << 22), b((size_t)1 << 22);
            // K0 on the first 2^22 elements of fresh copies (enough: the data repeats with period 2^20)
            float *pr, *mr, *vr;
            CK(hipMalloc((void**)&pr, (size_t)4 << 22));
            CK(hipMalloc((void**)&mr, (size_t)4 << 22));
            CK(hipMalloc((void**)&vr, (size_t)4 << 22));
            CK(hipMemcpy(pr, p, (size_t)4 << 22, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(mr, m, (size_t)4 << 22, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(vr, v, (size_t)4 << 22, hipMemcpyDeviceToDevice));
            hipLaunchKernelGGL((adam_k<0>), dim3((1 << 20) / 256), dim3(256), 0, 0, pr, g, mr, vr, bp, (int64_t)1 << 20, 1e-3f, 0.9f, 0.999f, 1e-8f);
            CK(hipDeviceSynchronize());
            long long diff = 0;
            float* pairs[3][2] = {{p2, pr}, {m2, mr}, {v2, vr}};
            for (auto& pr2 : pairs) {
                CK(hipMemcpy(a.data(), pr2[0], (size_t)4 << 22, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), pr2[1], (size_t)4 << 22, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < a.size(); ++i) diff += memcmp(&a[i], &b[i], 4) != 0;
            }
            printf("K%d vs K0 at Adam step %d (c1 = %.9g, c2 = %.9g): %lld of %d values differ\n", kk, t, 1.0f - hb[0], 1.0f - hb[1], diff, 3 << 22);
            CK(hipFree(pr));
            CK(hipFree(mr));
            CK(hipFree(vr));
        }
    }
    // --- exhaustive sweeps of the restated operations
    unsigned long long* bad;
    CK(hipMalloc((void**)&bad, 8));
    for (float c : {1.0f - 0.9f, 1.0f - 0.999f, 1.0f - 0.81f, 0.0019990206f, 0.6513215f, 0.99999994f, 0.39346933f}) {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(sweep_div, dim3(4096), dim3(256), 0, 0, c, bad);
        unsigned long long hbad;
        CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
        printf("div_uniform(x, c = %.9g) over all 2^32 x: %llu differ from x / c\n", c, hbad);
    }
    {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(sweep_sqrt, dim3(4096), dim3(256), 0, 0, bad);
        unsigned long long hbad;
        CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
        printf("sqrt_fast(x) over all 2^32 x: %llu differ from sqrtf(x)\n", hbad);
    }
    for (float b : {1e-8f, 0.1f + 1e-8f, 0.7071068f, 1.0f, 3.1415927f, 1.9999999f, 1e-4f, 123.456f}) {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(sweep_divfast, dim3(4096), dim3(256), 0, 0, b, bad);
        unsigned long long hbad;
        CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
        printf("div_fast(x, b = %.9g) over all 2^32 x: %llu differ from x / b\n", b, hbad);
    }
    // --- timings
    float hb[2] = {0.9f, 0.999f};
    CK(hipMemcpy(bp, hb, 8, hipMemcpyHostToDevice));
    printf("n = 2^%d parameters, %.1f MB per launch\n", logn, gb * 1e3);
    for (int rep = 0; rep < 2; ++rep) {
        report("K0 shipped (IEEE div x3 + sqrt)", time_k<0>(p, g, m, v, bp, n, 20));
        report("K1 no math, 7 streams", time_k<1>(p, g, m, v, bp, n, 20));
        report("K2 Markstein division by c1, c2", time_k<2>(p, g, m, v, bp, n, 20));
        report("K3 approximate math (floor)", time_k<3>(p, g, m, v, bp, n, 20));
        report("K4 K0 with temporal loads", time_k<4>(p, g, m, v, bp, n, 20));
        report("K5 K0, 32 contiguous bytes per lane", time_k<5>(p, g, m, v, bp, n, 20));
        report("K6 no math, 2 chunks per lane", time_k<6>(p, g, m, v, bp, n, 20));
        report("K7 all three divisions + sqrt restated", time_k<7>(p, g, m, v, bp, n, 20));
    }
    // --- stream-mix and placement sweeps (no arithmetic)
    CK(hipFree(buf));
    for (size_t stag2 : {(size_t)0, (size_t)256, (size_t)4352, (size_t)8448, (size_t)69888, (size_t)1052928, (size_t)(4352 + (2u << 20))}) {
        const size_t pitch2 = (bytes + stag2 + 255) / 256 * 256;
        uint8_t* b2;
        CK(hipMalloc((void**)&b2, pitch2 * 12 + 256));
        if (getenv("ADAM_STREAM_ZEROS")) CK(hipMemset(b2, 0, pitch2 * 12));
        else hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, (uint32_t*)b2, pitch2 * 12 / 4);
        time_mix<1, 3, 0>(b2, pitch2, n, "(Adam: g | p m v)");
        if (stag2 == 4352) {
            time_mix2<1, 3, 1, false>(b2, pitch2, n, "Adam, 1 KB per wave and array (= shipped)");
            time_mix2<1, 3, 2, false>(b2, pitch2, n, "Adam, 2 KB contiguous per wave and array");
            time_mix2<1, 3, 4, false>(b2, pitch2, n, "Adam, 4 KB contiguous per wave and array");
            time_mix2<1, 3, 1, true>(b2, pitch2, n, "Adam, p m v merged in 1 KB rows");
            time_mix2<1, 3, 2, true>(b2, pitch2, n, "Adam, p m v merged, 2 rows per wave");
            time_mix2<2, 2, 1, true>(b2, pitch2, n, "g, p read-only-slot | m v merged (2 KB)");
            time_mix2<2, 2, 2, true>(b2, pitch2, n, "... 2 rows per wave");
            time_mix2<1, 1, 1, false>(b2, pitch2, n, "Polyak shape");
            time_mix2<1, 1, 2, false>(b2, pitch2, n, "Polyak shape, 2 KB per wave");
            time_mix<1, 1, 0>(b2, pitch2, n, "(Polyak: src | dst)");
            time_mix<1, 1, 0, true, false>(b2, pitch2, n, "(Polyak, ordinary stores = shipped)");
            time_mix<4, 0, 3>(b2, pitch2, n, "(Adam out of place)");
            time_mix<7, 0, 0>(b2, pitch2, n, "(7 read-only)");
            time_mix<0, 0, 7>(b2, pitch2, n, "(7 write-only)");
            time_mix<0, 3, 0>(b2, pitch2, n, "(3 in place)");
            time_mix<0, 7, 0>(b2, pitch2, n, "(7 in place)");
            time_mix<1, 5, 2>(b2, pitch2, n, "(env-step like)");
            time_mix<2, 2, 0>(b2, pitch2, n, "(2 + 2)");
            time_mix<3, 1, 0>(b2, pitch2, n, "(3 + 1)");
            time_mix<1, 3, 0, true, false>(b2, pitch2, n, "(Adam, ordinary stores)");
            time_mix<1, 3, 0, false, true>(b2, pitch2, n, "(Adam, ordinary loads)");
        }
        CK(hipFree(b2));
    }
    // separately allocated arrays (what a host without the carve gets)
    {
        Ptrs ps;
        for (int k = 0; k < 4; ++k) CK(hipMalloc((void**)&ps.a[k], bytes));
        uint8_t* base = (uint8_t*)ps.a[0];
        printf("separate hipMalloc calls: array starts at +%lld, +%lld, +%lld bytes from the first\n", (long long)((uint8_t*)ps.a[1] - base),
               (long long)((uint8_t*)ps.a[2] - base), (long long)((uint8_t*)ps.a[3] - base));
        const int64_t n4 = n / 4;
        const int grid = (int)((n4 + 255) / 256);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_k<1, 3, 0, true, true>), dim3(grid), dim3(256), 0, 0, ps, n4);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((mix_k<1, 3, 0, true, true>), dim3(grid), dim3(256), 0, 0, ps, n4);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mix R=1 W=3 separately allocated arrays: %8.2f us  frac %.4f\n", ms / 20 * 1e3, 28.0 * n / 1e9 / (ms / 20 * 1e-3) / 8000.0);
    }
    return 0;
}
