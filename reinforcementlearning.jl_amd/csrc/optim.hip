// optim.hip -- flat-buffer parameter-update and loss primitives.
//
// Replaces:
//   TargetNetwork sync (Polyak / hard copy)   RLCore/policies/learners/target_network.jl:76-85
//   clip_by_global_norm! / global_norm        RLCore/utils/basic.jl:19-29
//   Flux.Optimise.update!(opt_state, model, grad) with Optimisers.Adam   flux_approximator.jl:46
//   normlogpdf / diagnormlogpdf               RLCore/utils/distributions.jl:18-21, :31-34
//   Flux.Losses.huber_loss, DQN TD target     (removed Zoo learner)
//
// All HBM-bound element-wise / reduction kernels: Adam 28 B/param, Polyak 12 B/param, clip 4-12 B/param.
// The learners' parameter vectors are tiny (3 k .. 1 M floats), so the hot variant is
// clip_adam_kernel: ONE single-workgroup launch doing [scale] -> global norm -> clip -> Adam with the
// gradient held in registers between the norm and the update (one read of g instead of three).
#include "optim_device.h"
#include <mutex>

namespace rlhip {

constexpr float LOG2PI_F = 1.8378770664093453f;  // log(2f0 * pi) rounded to Float32 (distributions.jl:9)

__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                     int64_t n, float rho) {
    float om = 1.0f - rho;  // (1 - rho)  target_network.jl:81
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = rho * dst[i] + om * src[i];
}

// stage 1 of the general (large n) global norm: per-block partial sums of squares (Float64 partials)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            double* __restrict__ partials) {
    __shared__ double scratch[16];
    double acc = 0.0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x = g[i];
        acc += (double)x * (double)x;
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// stage 2: gn = sqrt(sum); scale = clip_norm <= gn ? clip_norm / max(clip_norm, gn) : 1   (:23-26)
__global__ __launch_bounds__(256) void norm_finalize_kernel(const double* __restrict__ partials, int np,
                                                            float clip_norm, float* __restrict__ gn_out,
                                                            float* __restrict__ scale_out) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partials[i];
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        float gn = (float)sqrt(acc);
        gn_out[0] = gn;
        scale_out[0] = (clip_norm <= gn) ? clip_norm / fmaxf(clip_norm, gn) : 1.0f;
    }
}

__global__ __launch_bounds__(256) void scale_const_kernel(float* __restrict__ g, int64_t n, float s) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= s;
}

__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ g, int64_t n,
                                                       const float* __restrict__ scale) {
    float s = scale[0];
    if (s == 1.0f) return;  // not clipped: the reference leaves g untouched
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= s;
}

// `bt = bt .* b` of Optimisers.Adam folded into the Adam launch (round 5: was a second, one-thread launch -- 4.7 us for 8 bytes
// of work): every workgroup read beta_pow before its first update; the one that LEAVES last advances the running
// powers.  Two-level departure count: 65536 workgroups incrementing ONE counter serialise at ~12 ns per atomic (measured: Adam
// at 2^26 parameters 307 -> 771 us), so workgroup b counts into class b mod 64 (one 128-byte line each), and only the last of
// a class counts into the top word.  One counter block per stream (launches of one stream never overlap); every counter is
// re-armed by the workgroup that completes it.  No fence: nothing this workgroup stored is read by another one of the launch.
constexpr int DEPART_CLASSES = 64;
constexpr int DEPART_LINE_WORDS = 32;                                          // 128 bytes between two counters
constexpr int DEPART_BLOCK_WORDS = (DEPART_CLASSES + 1) * DEPART_LINE_WORDS;   // 64 class counters + the top word
__device__ __forceinline__ void advance_beta_pow_last_out(float* beta_pow, float b1, float b2, unsigned int* departed) {
    if (!departed) return;  // two-launch form: beta_pow_advance_kernel follows
    __syncthreads();  // every thread of this workgroup has read beta_pow
    if (threadIdx.x != 0) return;
    depart_barrier();
    const unsigned int grid = gridDim.x, cls = blockIdx.x % DEPART_CLASSES;
    const unsigned int in_class = (grid - cls + DEPART_CLASSES - 1) / DEPART_CLASSES;  // workgroups b < grid with b mod 64 == cls
    unsigned int* c = departed + cls * DEPART_LINE_WORDS;
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_class - 1) return;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int classes = grid < DEPART_CLASSES ? grid : DEPART_CLASSES;
    unsigned int* top = departed + DEPART_CLASSES * DEPART_LINE_WORDS;
    if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != classes - 1) return;
    beta_pow[0] *= b1;  // bt = bt .* b
    beta_pow[1] *= b2;
    __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ beta_pow, int64_t n, float lr,
                                                   float b1, float b2, float eps, unsigned int* __restrict__ departed) {
    float c1 = 1.0f - load_once(beta_pow), c2 = 1.0f - load_once(beta_pow + 1);
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam1(pi, g[i], mi, vi, lr, b1, b2, eps, c1, c2);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
    advance_beta_pow_last_out(beta_pow, b1, b2, departed);
}

// Streaming variants for parameter vectors that leave the caches (n >= STREAM_MIN_N floats, 16-byte aligned): four
// parameters per lane, 16-byte non-temporal LOADS (every byte is read once), element-wise arithmetic identical to the
// scalar kernels (same adam1 per element: bit-identical results).  NT_ST: non-temporal stores as well (chosen per kernel
// by A / B, see the launches).  Tail elements (n % 4) by the first lanes, scalar.
constexpr int64_t STREAM_MIN_N = 1 << 16;
#ifndef RLHIP_ADAM_CHUNKS
#define RLHIP_ADAM_CHUNKS 1
#endif
constexpr int ADAM_CHUNKS = RLHIP_ADAM_CHUNKS;
#ifndef RLHIP_POLYAK_ROWS
#define RLHIP_POLYAK_ROWS 1
#endif  // 16-byte chunks per thread of adam_vec4_kernel (A / B: tools/adam_grid_ab.py)
union f32x4_bits {
    nt_u32x4 u;
    float f[4];
};
template <bool NT_ST>
__device__ __forceinline__ void st16(void* p, const f32x4_bits& x) {
    if (NT_ST) nt_store16(p, x.u);
    else *reinterpret_cast<nt_u32x4*>(p) = x.u;
}

template <bool NT_ST, int U>
__global__ __launch_bounds__(256) void adam_vec4_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ beta_pow, int64_t n, float lr,
                                                        float b1, float b2, float eps, unsigned int* __restrict__ departed) {
    const float c1 = 1.0f - load_once(beta_pow), c2 = 1.0f - load_once(beta_pow + 1);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    // U 16-byte chunks per lane and iteration, `stride` apart (each chunk row stays coalesced): 4 U loads in flight per lane
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += (int64_t)U * stride) {
        f32x4_bits pi[U], gi[U], mi[U], vi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                pi[u].u = nt_load16(p + 4 * i);
                gi[u].u = nt_load16(g + 4 * i);
                mi[u].u = nt_load16(m + 4 * i);
                vi[u].u = nt_load16(v + 4 * i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) adam1(pi[u].f[k], gi[u].f[k], mi[u].f[k], vi[u].f[k], lr, b1, b2, eps, c1, c2);
                st16<NT_ST>(p + 4 * i, pi[u]);
                st16<NT_ST>(m + 4 * i, mi[u]);
                st16<NT_ST>(v + 4 * i, vi[u]);
            }
        }
    }
    const int64_t t = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float pi = p[t], mi = m[t], vi = v[t];
        adam1(pi, g[t], mi, vi, lr, b1, b2, eps, c1, c2);
        p[t] = pi;
        m[t] = mi;
        v[t] = vi;
    }
    advance_beta_pow_last_out(beta_pow, b1, b2, departed);
}

// ROWS 1 KB rows (64 lanes x 16 bytes) per wave, CONTIGUOUS in memory, every load of the wave issued before the first use.  ROWS = 1
// is shipped: two rows per wave measured the same with ordinary stores and 7 % slower with non-temporal ones (see the launch).
template <bool NT_ST, int ROWS>
__global__ __launch_bounds__(256) void polyak_vec4_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                          int64_t n, float rho) {
    const float om = 1.0f - rho;
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x * ROWS;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int64_t i0 = wave0 * (64 * ROWS) + lane; i0 < n4; i0 += stride) {
        f32x4_bits d[ROWS], x[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 + 64 * r;
            if (i < n4) {
                d[r].u = nt_load16(dst + 4 * i);
                x[r].u = nt_load16(src + 4 * i);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t i = i0 + 64 * r;
            if (i < n4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[r].f[k] = rho * d[r].f[k] + om * x[r].f[k];
                st16<NT_ST>(dst + 4 * i, d[r]);
            }
        }
    }
    const int64_t t = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = rho * dst[t] + om * src[t];
}

static bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0);
}

__global__ void beta_pow_advance_kernel(float* beta_pow, float b1, float b2) {
    beta_pow[0] *= b1;  // bt = bt .* b
    beta_pow[1] *= b2;
}

// Single-workgroup fused [grad_scale] -> global norm -> clip -> Adam.  Each thread keeps up to
// PER_THREAD gradient values in registers between the norm and the update.
template <int PER_THREAD>
__global__ __launch_bounds__(1024) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ beta_pow, int64_t n,
                                                         float grad_scale, float clip_norm, float lr,
                                                         float b1, float b2, float eps,
                                                         float* __restrict__ gn_out) {
    __shared__ double scratch[16];
    float gr[PER_THREAD];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        int64_t i = (int64_t)k * blockDim.x + threadIdx.x;
        float x = (i < n) ? g[i] * grad_scale : 0.0f;
        gr[k] = x;
        acc += (double)x * (double)x;
    }
    float c1 = 1.0f - beta_pow[0], c2 = 1.0f - beta_pow[1];
    acc = block_sum(acc, scratch);
    float gn = (float)sqrt(acc);
    float scale = (clip_norm > 0.0f && clip_norm <= gn) ? clip_norm / fmaxf(clip_norm, gn) : 1.0f;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        int64_t i = (int64_t)k * blockDim.x + threadIdx.x;
        if (i < n) {
            float gi = (scale == 1.0f) ? gr[k] : gr[k] * scale;
            float pi = p[i], mi = m[i], vi = v[i];
            adam1(pi, gi, mi, vi, lr, b1, b2, eps, c1, c2);
            p[i] = pi;
            m[i] = mi;
            v[i] = vi;
            g[i] = gi;  // the clipped gradient stays observable, like gs[p] .*= ... in the reference
        }
    }
    if (threadIdx.x == 0) {
        if (gn_out) gn_out[0] = gn;
        beta_pow[0] *= b1;  // every thread has read beta_pow before the block_sum barrier
        beta_pow[1] *= b2;
    }
}

// Same fused step for mid-size vectors (4 k .. 64 k parameters, e.g. the 17 410-parameter 3-layer Q-network):
// 16 B/lane accesses and ALL loads of the Adam phase issued before the first dependent use.  The scalar variant
// above walks 64 predicated iterations of {load p, m, v -> ~30 dependent instructions -> store}: one memory round
// trip per iteration on a single workgroup (measured 19.6 us for 17 k parameters; this one: see profiles/).
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ a, int64_t i, int64_t n) {
    if (i + 3 < n) return *reinterpret_cast<const float4*>(a + i);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) r.x = a[i];
    if (i + 1 < n) r.y = a[i + 1];
    if (i + 2 < n) r.z = a[i + 2];
    return r;
}
__device__ __forceinline__ void store4_guard(float* __restrict__ a, int64_t i, int64_t n, float4 x) {
    if (i + 3 < n) {
        *reinterpret_cast<float4*>(a + i) = x;
        return;
    }
    if (i < n) a[i] = x.x;
    if (i + 1 < n) a[i + 1] = x.y;
    if (i + 2 < n) a[i + 2] = x.z;
}

template <int IT>
__global__ __launch_bounds__(1024) void clip_adam_vec_kernel(float* __restrict__ p, float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             float* __restrict__ beta_pow, int64_t n,
                                                             float grad_scale, float clip_norm, float lr, float b1,
                                                             float b2, float eps, float* __restrict__ gn_out) {
    __shared__ double scratch[16];
    float4 gr[IT];  // the gradient stays in registers between the norm and the update (4 * IT <= 64 VGPRs)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < IT; ++k) gr[k] = load4_guard(g, ((int64_t)k * 1024 + threadIdx.x) * 4, n);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        gr[k].x *= grad_scale;
        gr[k].y *= grad_scale;
        gr[k].z *= grad_scale;
        gr[k].w *= grad_scale;
        // element order inside a thread: ascending index, as in the scalar kernel
        acc += (double)gr[k].x * (double)gr[k].x;
        acc += (double)gr[k].y * (double)gr[k].y;
        acc += (double)gr[k].z * (double)gr[k].z;
        acc += (double)gr[k].w * (double)gr[k].w;
    }
    const float c1 = 1.0f - beta_pow[0], c2 = 1.0f - beta_pow[1];
    acc = block_sum(acc, scratch);
    const float gn = (float)sqrt(acc);
    const float scale = (clip_norm > 0.0f && clip_norm <= gn) ? clip_norm / fmaxf(clip_norm, gn) : 1.0f;
    // Adam phase two vectors at a time: 6 independent 16-byte loads in flight per thread, 24 VGPRs of operands
    // (1024 threads per workgroup leave 128 VGPRs per lane: holding p, m, v of every iteration spills)
#pragma unroll
    for (int k0 = 0; k0 < IT; k0 += 2) {
        float4 pv[2], mv[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = ((int64_t)(k0 + u) * 1024 + threadIdx.x) * 4;
            pv[u] = load4_guard(p, i, n);
            mv[u] = load4_guard(m, i, n);
            vv[u] = load4_guard(v, i, n);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = ((int64_t)(k0 + u) * 1024 + threadIdx.x) * 4;
            if (i >= n) continue;
            float4 gi = gr[k0 + u];
            if (scale != 1.0f) {
                gi.x *= scale;
                gi.y *= scale;
                gi.z *= scale;
                gi.w *= scale;
            }
            adam1(pv[u].x, gi.x, mv[u].x, vv[u].x, lr, b1, b2, eps, c1, c2);
            adam1(pv[u].y, gi.y, mv[u].y, vv[u].y, lr, b1, b2, eps, c1, c2);
            adam1(pv[u].z, gi.z, mv[u].z, vv[u].z, lr, b1, b2, eps, c1, c2);
            adam1(pv[u].w, gi.w, mv[u].w, vv[u].w, lr, b1, b2, eps, c1, c2);
            store4_guard(p, i, n, pv[u]);
            store4_guard(m, i, n, mv[u]);
            store4_guard(v, i, n, vv[u]);
            store4_guard(g, i, n, gi);
        }
    }
    if (threadIdx.x == 0) {
        if (gn_out) gn_out[0] = gn;
        beta_pow[0] *= b1;
        beta_pow[1] *= b2;
    }
}

__global__ __launch_bounds__(256) void normlogpdf_kernel(const float* __restrict__ mu,
                                                         const float* __restrict__ sigma,
                                                         const float* __restrict__ x, float* __restrict__ out,
                                                         int64_t n) {
    const float eps = 1.0e-8f;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float se = sigma[i] + eps;
        float z = (x[i] - mu[i]) / se;                        // :19
        out[i] = -(z * z + LOG2PI_F) / 2.0f - logf(se);       // :20
    }
}

__global__ __launch_bounds__(256) void diagnormlogpdf_kernel(const float* __restrict__ mu,
                                                             const float* __restrict__ sigma,
                                                             const float* __restrict__ x, int64_t d,
                                                             int64_t n, float* __restrict__ out) {
    const float eps = 1.0e-8f;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float prod = 1.0f, sum = 0.0f;
    for (int64_t k = 0; k < d; ++k) {
        float s = sigma[i * d + k] + eps;
        float vv = s * s;                                      // :32
        float dx = x[i * d + k] - mu[i * d + k];
        prod *= vv;
        sum += (dx * dx) / vv;
    }
    out[i] = -0.5f * (logf(prod) + sum + (float)d * LOG2PI_F);  // :33
}

// huber: per-block partial sums (Float64) + optional dL/dq
__global__ __launch_bounds__(256) void huber_partial_kernel(const float* __restrict__ q,
                                                            const float* __restrict__ target, int64_t n,
                                                            float delta, float* __restrict__ dq,
                                                            double* __restrict__ partials) {
    __shared__ double scratch[16];
    double acc = 0.0;
    float inv_n = 1.0f / (float)n;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float d = q[i] - target[i];
        float e = fabsf(d);
        float l = (e < delta) ? (e * e) * 0.5f : delta * (e - 0.5f * delta);
        acc += (double)l;
        if (dq) {
            float gi = (e < delta) ? d : (d > 0.0f ? delta : (d < 0.0f ? -delta : 0.0f));
            dq[i] = gi / (float)n;
        }
    }
    (void)inv_n;
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void mean_finalize_kernel(const double* __restrict__ partials, int np,
                                                            int64_t n, float* __restrict__ out) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partials[i];
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[0] = (float)(acc / (double)n);
}

__global__ __launch_bounds__(256) void td_target_kernel(const float* __restrict__ qt, int64_t na, int64_t n,
                                                        int64_t ks, int64_t is, const float* __restrict__ r,
                                                        const uint8_t* __restrict__ term, float gamma,
                                                        float* __restrict__ target) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* qp = qt + i * is;
    float mx = qp[0];
    for (int64_t k = 1; k < na; ++k) {
        float x = qp[k * ks];
        if (x > mx) mx = x;
    }
    float cont = term[i] ? 0.0f : 1.0f;
    target[i] = r[i] + gamma * cont * mx;
}

// Small persistent scratch for the multi-launch reductions (per stream use is serialized by the
// stream itself; one scratch per device is enough because every user enqueues on a single stream
// per device in this library's usage; callers that need concurrency pass disjoint streams at
// their own risk -- documented in DESIGN.md).
// Two-launch fused step for vectors beyond the single-workgroup range (> 32 k parameters):
//   sumsq_scaled_partial_kernel  per-workgroup Float64 partial sums of (grad_scale * g)^2
//   clip_adam_grid_kernel        every workgroup re-derives the norm from the <= 256 partials (same order -> same
//                                value everywhere), clips and applies Adam to its slice; the workgroup that leaves
//                                last (agent-scope counter) advances the running beta powers and re-arms the counter
__global__ __launch_bounds__(256) void sumsq_scaled_partial_kernel(const float* __restrict__ g, int64_t n,
                                                                   float grad_scale, double* __restrict__ partials) {
    __shared__ double scratch[16];
    double acc = 0.0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x = g[i] * grad_scale;
        acc += (double)x * (double)x;
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void clip_adam_grid_kernel(float* __restrict__ p, float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             float* __restrict__ beta_pow, int64_t n, float grad_scale,
                                                             float clip_norm, float lr, float b1, float b2, float eps,
                                                             const double* __restrict__ partials, int npart,
                                                             unsigned int* __restrict__ departed,
                                                             float* __restrict__ gn_out) {
    __shared__ double scratch[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npart; i += blockDim.x) acc += partials[i];
    acc = block_sum(acc, scratch);
    const float gn = (float)sqrt(acc);
    const float scale = (clip_norm > 0.0f && clip_norm <= gn) ? clip_norm / fmaxf(clip_norm, gn) : 1.0f;
    const float c1 = 1.0f - load_once(beta_pow), c2 = 1.0f - load_once(beta_pow + 1);
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gi = g[i] * grad_scale;
        if (scale != 1.0f) gi *= scale;
        float pi = p[i], mi = m[i], vi = v[i];
        adam1(pi, gi, mi, vi, lr, b1, b2, eps, c1, c2);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        g[i] = gi;
    }
    __syncthreads();  // every thread of this workgroup has read beta_pow
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0 && gn_out) gn_out[0] = gn;
        // (no release fence: nothing this workgroup stored is read by another workgroup of the launch, and an L2 write-back
        // on every workgroup's tail costs ~1 us per launch -- measured on reduce_apply_kernel, profiles/r03_tile_mfma.md)
        if (departed) {  // (nullptr: no departure counter for this stream -- beta_pow_advance_kernel follows as its own launch)
            depart_barrier();
            unsigned int prev = __hip_atomic_fetch_add(departed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == gridDim.x - 1) {  // last one out: nobody reads beta_pow any more
                beta_pow[0] *= b1;
                beta_pow[1] *= b2;
                __hip_atomic_store(departed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

constexpr int DEPART_SLOTS = 64;
constexpr int PARTIALS_PER_SLOT = 1024;
struct Scratch {
    // one block of 1024 doubles per stream slot + a last, SHARED block for the streams beyond the slots (one buffer per
    // device made two streams running the clipped update at once race on the norm partials)
    double* partials = nullptr;
    float* scalars = nullptr;    // 4 floats per slot (+ the shared block)
    unsigned int* counter = nullptr;  // departure counters (zero between launches), one block per STREAM slot: launches of one
                                      // stream never overlap, launches of two streams must not share a counter
    hipStream_t stream_of[DEPART_SLOTS] = {};
    int n_streams = 0;
    int device = -1;
};
static Scratch g_scratch[16];
static std::mutex g_scratch_mutex;

// the slot of `stream` on this device, -1 when more than DEPART_SLOTS distinct streams have asked.  A slot is never recycled (a
// kernel of its stream may still be in flight); the callers of a slot-less stream take their counter-free, two-launch routes and
// the shared scratch block -- slower, still correct for one slot-less stream at a time.
static int stream_slot(Scratch& s, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_scratch_mutex);
    for (int i = 0; i < s.n_streams; ++i)
        if (s.stream_of[i] == stream) return i;
    if (s.n_streams == DEPART_SLOTS) return -1;
    s.stream_of[s.n_streams] = stream;
    return s.n_streams++;
}
static unsigned int* slot_counter(Scratch& s, int slot) { return slot < 0 ? nullptr : s.counter + (size_t)DEPART_BLOCK_WORDS * slot; }
static double* slot_partials(Scratch& s, int slot) { return s.partials + (size_t)PARTIALS_PER_SLOT * (slot < 0 ? DEPART_SLOTS : slot); }
static float* slot_scalars(Scratch& s, int slot) { return s.scalars + 4 * (size_t)(slot < 0 ? DEPART_SLOTS : slot); }

static int32_t get_scratch(Scratch** out) {
    int dev = 0;
    RLHIP_CHECK_HIP(hipGetDevice(&dev));
    RLHIP_REQUIRE(dev >= 0 && dev < 16, "device index out of range");
    Scratch& s = g_scratch[dev];
    if (s.device != dev) {
        RLHIP_CHECK_HIP(hipMalloc((void**)&s.partials, (size_t)PARTIALS_PER_SLOT * (DEPART_SLOTS + 1) * sizeof(double)));
        RLHIP_CHECK_HIP(hipMalloc((void**)&s.scalars, 4 * (DEPART_SLOTS + 1) * sizeof(float)));
        RLHIP_CHECK_HIP(hipMalloc((void**)&s.counter, sizeof(unsigned int) * DEPART_BLOCK_WORDS * DEPART_SLOTS));
        RLHIP_CHECK_HIP(hipMemset(s.counter, 0, sizeof(unsigned int) * DEPART_BLOCK_WORDS * DEPART_SLOTS));
        s.device = dev;
    }
    *out = &s;
    return RLHIP_OK;
}

// one launch: the update of every parameter AND `bt = bt .* b` (the last workgroup out advances the running powers)
static int32_t adam_launch(float* params, const float* grad, float* m, float* v, float* beta_pow, int64_t n, float lr, float beta1,
                           float beta2, float eps, unsigned int* departed, hipStream_t s) {
    if (n >= STREAM_MIN_N && aligned16(params, grad, m, v)) {
        // one trip per thread (no persistent grid-stride loop below 2^30 parameters): at 2^26 parameters Adam takes 307 us with
        // 65536 workgroups against 371 - 429 us with 1024 - 16384 looping ones (tools/adam_grid_ab.py); non-temporal stores.
        // U = 16-byte chunks per thread and array, `stride` apart; the grid is sized so that all U really are in flight (ADVICE
        // r4: round 4 launched one thread per chunk with U = 2, which left the second chunk of every thread out of range, so its
        // "two chunks" A / B compared nothing).  Re-measured with the grid halved (tools/r5_b.sh, same box): U = 2 is 1 - 2 %
        // faster at both sizes (21.2 vs 21.5 us, 333 - 338 vs 341 us) -- inside the run-to-run spread; U stays 1
        constexpr int U = ADAM_CHUNKS;
        const int grid = grid_for((n / 4 + U - 1) / U, 256, 1 << 20);
        hipLaunchKernelGGL((adam_vec4_kernel<true, U>), dim3(grid), dim3(256), 0, s, params, grad, m, v, beta_pow, n, lr, beta1,
                           beta2, eps, departed);
    } else
        hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, params, grad, m, v, beta_pow, n, lr, beta1,
                           beta2, eps, departed);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_polyak_f32(float* dst, const float* src, int64_t n, float rho, rlhip_stream_t stream) {
    RLHIP_REQUIRE(dst != nullptr && src != nullptr && n >= 0, "bad arguments");
    RLHIP_REQUIRE(rho >= 0.0f && rho <= 1.0f, "rho must be in [0,1] (AssertionError in the reference, target_network.jl:50)");
    if (n == 0) return RLHIP_OK;
    if (n >= STREAM_MIN_N && aligned16(dst, src)) {
        // one 16-byte chunk per thread (no persistent grid-stride loop below 2^30 parameters): at 2^26 parameters Adam takes
        // 307 us with 65536 workgroups against 371 - 429 us with 1024 - 16384 looping ones, Polyak 118 against 120 - 126
        // (tools/adam_grid_ab.py); the loops in the kernels only serve vectors beyond the grid cap
        // one 1 KB row per wave and array, ordinary stores.  Round 6 A / B on one box, 2^26 parameters (tools/polyak_ab.py, three
        // alternations): one row + ordinary stores 115.9 - 116.1 us, two contiguous rows + ordinary 116.0 - 116.3, one row + NT
        // 116.0 - 116.1, two rows + NT 124.5 - 125.1.  (A stand-alone twin of this launch read 101 us with two rows + NT stores --
        // on the all-zero data of a hipMemset; on random floats the same twin reads 116.5 / 120.1 us: profiles/r06_adam.md section 5.)
        constexpr int ROWS = RLHIP_POLYAK_ROWS;
        const int grid = grid_for((n / 4 + ROWS - 1) / ROWS, 256, 1 << 20);
        hipLaunchKernelGGL((polyak_vec4_kernel<false, ROWS>), dim3(grid), dim3(256), 0, as_stream(stream), dst, src, n, rho);
    } else
        hipLaunchKernelGGL(polyak_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), dst, src, n, rho);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_clip_by_global_norm_f32(float* grad, int64_t n, float clip_norm, float* gn_out,
                                      rlhip_stream_t stream) {
    RLHIP_REQUIRE(grad != nullptr && gn_out != nullptr && n >= 0, "bad arguments");
    Scratch* sc;
    int32_t rc = get_scratch(&sc);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    const int slot = stream_slot(*sc, s);
    double* partials = slot_partials(*sc, slot);
    float* scalars = slot_scalars(*sc, slot);
    int nb = grid_for(n > 0 ? n : 1, 256, 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, grad, n, partials);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, s, partials, nb, clip_norm, gn_out, scalars);
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n > 0 ? n : 1, 256)), dim3(256), 0, s, grad, n, scalars);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_adam_f32(float* params, const float* grad, float* m, float* v, float* beta_pow, int64_t n,
                       float lr, float beta1, float beta2, float eps, rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && grad && m && v && beta_pow && n >= 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    Scratch* sc;
    int32_t rc = get_scratch(&sc);
    if (rc) return rc;
    // `bt = bt .* b` inside the update launch (last workgroup out) up to 2^23 parameters, as its own one-thread launch beyond:
    // same-box A / B (tools/r5_c.sh): 2^22 parameters 21.2 us folded against 22.05 us in two launches; 2^26 parameters 342.5
    // against 339.5 -- 65536 departure atomics cost more than the second launch saves, even counted in two levels (with ONE
    // counter: 771 us)
    unsigned int* dep = n <= ((int64_t)1 << 23) ? slot_counter(*sc, stream_slot(*sc, s)) : nullptr;
    if (n == 0 || !dep) {  // nothing to fold the advance into / two-launch form: the update, then `bt = bt .* b` on its own
        if (n > 0 && (rc = adam_launch(params, grad, m, v, beta_pow, n, lr, beta1, beta2, eps, nullptr, s))) return rc;
        hipLaunchKernelGGL(beta_pow_advance_kernel, dim3(1), dim3(1), 0, s, beta_pow, beta1, beta2);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    return adam_launch(params, grad, m, v, beta_pow, n, lr, beta1, beta2, eps, dep, s);
}

int32_t rlhip_clip_adam_f32(float* params, float* grad, float* m, float* v, float* beta_pow, int64_t n,
                            float grad_scale, float clip_norm, float lr, float beta1, float beta2,
                            float eps, float* gn_out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(params && grad && m && v && beta_pow && n >= 0, "bad arguments");
    hipStream_t s = as_stream(stream);
    int64_t per = (n + 1023) / 1024;
    if (per > 12) {
        // beyond ~12 k parameters two grid-wide launches (partial norm, then clip + Adam per slice) beat one workgroup
        // (measured: 17 k parameters 13.8 us single workgroup vs ~9 us; 64 k: 45 vs 10.5 us)
        Scratch* sc;
        int32_t rc = get_scratch(&sc);
        if (rc) return rc;
        const int nb = grid_for(n, 256, 256);
        const int slot = stream_slot(*sc, s);  // resolved BEFORE anything is enqueued
        double* partials = slot_partials(*sc, slot);
        unsigned int* dep = slot_counter(*sc, slot);
        hipLaunchKernelGGL(sumsq_scaled_partial_kernel, dim3(nb), dim3(256), 0, s, grad, n, grad_scale, partials);
        hipLaunchKernelGGL(clip_adam_grid_kernel, dim3(nb), dim3(256), 0, s, params, grad, m, v, beta_pow, n, grad_scale,
                           clip_norm, lr, beta1, beta2, eps, partials, nb, dep, gn_out);
        if (!dep)  // a 65th stream: no departure counter -- `bt = bt .* b` as its own launch behind the update
            hipLaunchKernelGGL(beta_pow_advance_kernel, dim3(1), dim3(1), 0, s, beta_pow, beta1, beta2);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
    const bool aligned = ((((uintptr_t)params | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#define LAUNCH_CAV(IT)                                                                                     \
    hipLaunchKernelGGL((clip_adam_vec_kernel<IT>), dim3(1), dim3(1024), 0, s, params, grad, m, v, beta_pow, n, \
                       grad_scale, clip_norm, lr, beta1, beta2, eps, gn_out)
    if (aligned && per > 4 && per <= 32) {  // 4 k .. 32 k parameters: 16 B/lane, loads hoisted, no spills
        if (per <= 8) LAUNCH_CAV(2);
        else if (per <= 16) LAUNCH_CAV(4);
        else LAUNCH_CAV(8);
        RLHIP_LAUNCH_CHECK();
        return RLHIP_OK;
    }
#define LAUNCH_CA(PT)                                                                                  \
    hipLaunchKernelGGL((clip_adam_kernel<PT>), dim3(1), dim3(1024), 0, s, params, grad, m, v, beta_pow, n, \
                       grad_scale, clip_norm, lr, beta1, beta2, eps, gn_out)
    if (per <= 4) LAUNCH_CA(4);
    else if (per <= 16) LAUNCH_CA(16);
    else LAUNCH_CA(64);
#undef LAUNCH_CA
#undef LAUNCH_CAV
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_normlogpdf_f32(const float* mu, const float* sigma, const float* x, float* out, int64_t n,
                             rlhip_stream_t stream) {
    RLHIP_REQUIRE(mu && sigma && x && out && n >= 0, "bad arguments");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(normlogpdf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), mu, sigma,
                       x, out, n);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_diagnormlogpdf_f32(const float* mu, const float* sigma, const float* x, int64_t d, int64_t n,
                                 float* out, rlhip_stream_t stream) {
    RLHIP_REQUIRE(mu && sigma && x && out && n >= 0 && d >= 1, "bad arguments");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(diagnormlogpdf_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       mu, sigma, x, d, n, out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_huber_f32(const float* q, const float* target, int64_t n, float delta, float* loss_out,
                        float* dq, rlhip_stream_t stream) {
    RLHIP_REQUIRE(q && target && loss_out && n >= 1, "bad arguments");
    Scratch* sc;
    int32_t rc = get_scratch(&sc);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    int nb = grid_for(n, 256, 1024);
    double* partials = slot_partials(*sc, stream_slot(*sc, s));
    hipLaunchKernelGGL(huber_partial_kernel, dim3(nb), dim3(256), 0, s, q, target, n, delta, dq, partials);
    hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(256), 0, s, partials, nb, n, loss_out);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

int32_t rlhip_td_target_f32(const float* qt_next, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride,
                            const float* reward, const uint8_t* terminal, float gamma, float* target,
                            rlhip_stream_t stream) {
    RLHIP_REQUIRE(qt_next && reward && terminal && target && na >= 1 && n >= 0, "bad arguments");
    if (n == 0) return RLHIP_OK;
    hipLaunchKernelGGL(td_target_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), qt_next,
                       na, n, k_stride, i_stride, reward, terminal, gamma, target);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

/* the n-step form (R = r + gamma^n (1 - t) max Qt(s')): `reward` holds the n-step returns of the sampled windows
 * (rlhip_ring_fold_nstep), `terminal` any(terminal[window]), qt_next the target values of s_{i + n}; gamma^n = rlhip_gamma_pow */
int32_t rlhip_td_target_n_f32(const float* qt_next, int64_t na, int64_t n, int64_t k_stride, int64_t i_stride,
                              const float* reward, const uint8_t* terminal, float gamma, int32_t n_step, float* target,
                              rlhip_stream_t stream) {
    RLHIP_REQUIRE(n_step >= 1, "n_step must be >= 1");
    return rlhip_td_target_f32(qt_next, na, n, k_stride, i_stride, reward, terminal, rlhip_gamma_pow(gamma, n_step), target, stream);
}

}  // extern "C"
