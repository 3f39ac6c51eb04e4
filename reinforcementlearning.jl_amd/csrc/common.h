// common.h -- shared host/device helpers of librlhip.so (gfx950 only).
//
// Error model of the C ABI: every entry point returns int32 (0 = RLHIP_OK, <0 = error) and stores
// a message readable through rlhip_last_error() (thread-local).  No exceptions cross the boundary.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rlhip.h"

namespace rlhip {

void set_error(const char* fmt, ...);

#define RLHIP_CHECK_HIP(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ::rlhip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,               \
                               hipGetErrorString(_e));                                     \
            return RLHIP_EHIP;                                                             \
        }                                                                                  \
    } while (0)

#define RLHIP_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            ::rlhip::set_error("%s:%d: invalid argument: %s (%s)", __FILE__, __LINE__, msg, \
                               #cond);                                                     \
            return RLHIP_EINVAL;                                                           \
        }                                                                                  \
    } while (0)

#define RLHIP_LAUNCH_CHECK() RLHIP_CHECK_HIP(hipGetLastError())

#ifdef RLHIP_BOUNDS_CHECK
// the debug build (RLHIP_EXTRA_FLAGS=-DRLHIP_BOUNDS_CHECK python reinforcementlearning.jl_amd/build.py --force): every gather
// that takes caller-supplied indices validates them first (one more launch and a stream synchronisation per call)
#define RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream)                                                               \
    do {                                                                                                                 \
        int64_t nbad_ = 0, first_ = -1;                                                                                  \
        int32_t rc_ = rlhip_ring_check_indices(rb, idx, batch, &nbad_, &first_, stream);                                 \
        if (rc_) return rc_;                                                                                             \
        if (nbad_) {                                                                                                     \
            ::rlhip::set_error("%s:%d: %lld of %lld gather indices are outside [0, %lld) (first at position %lld)",      \
                               __FILE__, __LINE__, (long long)nbad_, (long long)(batch),                                 \
                               (long long)((rb)->len_rt * (rb)->n_env), (long long)first_);                              \
            return RLHIP_EINVAL;                                                                                         \
        }                                                                                                                \
    } while (0)
#else
#define RLHIP_CHECK_GATHER_INDICES(rb, idx, batch, stream) \
    do {                                                   \
    } while (0)
#endif


static inline hipStream_t as_stream(rlhip_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Grid sizing for streaming kernels: enough workgroups to fill 256 CUs x 8 blocks, grid-stride the
// rest (cdna_hip_programming.md Guideline 11).
static inline int grid_for(int64_t n, int block, int max_blocks = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// Kernels that synchronise their whole grid through a spin barrier (reduce_apply_kernel<APPLY_GRID / APPLY_XCHG>,
// d3_apply_kernel) are only launched when every workgroup can be RESIDENT at once with room to spare: the occupancy API's
// answer for this kernel on this device, minus one workgroup per CU (on gfx950 / ROCm 7.2 the API can be one block per CU
// high: MI355X_MICROARCH.md, residency), times the CU count of the device actually present (not an assumed 256), and
// the grid may use at most HALF of that (other streams, other ranks sharing the device, a profiler's kernels).  Work that
// other streams have already placed on the CUs drains -- it does not wait for us -- so residency is delayed, never
// denied; a grid beyond the capacity would deadlock, and the callers then take their barrier-free variant.
template <class K>
static inline int grid_barrier_capacity(K kernel, int block_threads, size_t dynamic_lds = 0) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dynamic_lds) != hipSuccess) return 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (per_cu > 1) per_cu -= 1;
    return (per_cu * cus) / 2;
}

// Per-DEVICE facts are cached per device: a process may drive several (hipSetDevice), and the LDS opt-in, the CU count and
// a grid barrier's capacity are properties of the device a launch goes to (process-wide statics made a second
// device miss the opt-in and inherit the first one's barrier capacity -- too high a capacity deadlocks a spin barrier).
constexpr int RLHIP_MAX_DEVICES = 64;
static inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % RLHIP_MAX_DEVICES;
}
struct PerDeviceInt {  // value per device, -1 = not yet known
    int v[RLHIP_MAX_DEVICES];
    PerDeviceInt() {
        for (int i = 0; i < RLHIP_MAX_DEVICES; ++i) v[i] = -1;
    }
};
// grid-barrier capacity of `kernel` on the current device (RLHIP_GRID_BARRIER_CAP: test hook, forces the barrier-free
// variants (0) or a small device)
template <class K>
static inline int grid_barrier_capacity_cached(PerDeviceInt& cache, K kernel, int block_threads, size_t dynamic_lds = 0) {
    int& c = cache.v[current_device_slot()];
    if (c < 0) {
        c = grid_barrier_capacity(kernel, block_threads, dynamic_lds);
        const char* e = getenv("RLHIP_GRID_BARRIER_CAP");
        if (e) c = atoi(e);
    }
    return c;
}
static inline int device_cu_count() {
    static PerDeviceInt cache;
    int& c = cache.v[current_device_slot()];
    if (c < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 1;
        c = n < 1 ? 1 : n;
    }
    return c;
}
// gfx950 has 160 KB of LDS per workgroup; anything above 64 KB of dynamic LDS must be opted into per kernel AND device.
// done_mask: one static word per call site (= per kernel instantiation), bit = device.
template <class K>
static inline int32_t allow_big_lds(K kernel, size_t bytes, unsigned long long* done_mask) {
    const unsigned long long bit = 1ull << current_device_slot();
    if (*done_mask & bit) return RLHIP_OK;
    RLHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bytes));
    *done_mask |= bit;
    return RLHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10, ctr = {idx, blk, t, tag}, key = {seed_lo, seed_hi}.  Same specification as the
// oracle (oracle/rlo_rng.c) -- shared by specification, not by code; checked by tests/.
// ---------------------------------------------------------------------------------------------
enum : uint32_t {
    TAG_RESET = 0,
    TAG_EXPLORE = 1,
    TAG_GUMBEL = 2,
    TAG_NORMAL = 3,
    TAG_SAMPLER = 4,
    TAG_SHUFFLE = 5,
    TAG_INIT = 6,
    TAG_SYNTH = 7,
    TAG_ENVNOISE = 8
};

// development switches are read from the environment once per process
// ---- the memory-system contract of the intra-device hand-offs -------------------------------------------------
// Several kernels hand data between workgroups of ONE device with relaxed agent-scope atomics instead of release / acquire
// fences: a partial value IS its own arrival flag (8-byte {epoch, payload} granules stored and polled with relaxed agent-scope
// atomics), or plain stores are drained with `s_waitcnt vmcnt(0)` before a relaxed agent-scope arrival increment and read by
// the last arriver with relaxed agent-scope loads (reduce_apply_kernel, dqn_reduce_apply_kernel, d3_apply_kernel,
// clip_adam_grid_kernel, ppo3w_adam_pack_kernel, sumtree_update_kernel keeps its fences).  That is a data race in the HIP /
// LLVM memory model; it is correct on gfx942 / gfx950 because (i) vector-memory STORES are counted by vmcnt there (no separate
// store counter as on gfx10+), (ii) agent-scope atomics and sc1 accesses are performed at the device-coherent L2, past the
// per-CU L1, and (iii) a wave's plain stores are written through to that L2 in program order once vmcnt reaches 0.
// The library is built for gfx950 only (build.py); this check turns any other target into a compile error instead of a
// latent race.  Everything that crosses DEVICES (p2p.hip, the APPLY_XCHG exchange, comm.hip) keeps system-scope release /
// acquire fences and flags -- DESIGN.md section 6 has the table.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "rlhip: the relaxed intra-device hand-off protocols are only valid on gfx942 / gfx950 (see the comment above)"
#endif

#define RLHIP_ENV_FLAG(name)                          \
    ([]() -> bool {                                   \
        static const bool v_ = []() {                 \
            const char* e_ = getenv(name);            \
            return e_ != nullptr && e_[0] != '\0' && !(e_[0] == '0' && e_[1] == '\0'); \
        }();                                          \
        return v_;                                    \
    }())

// Cross-lane sums without the LDS crossbar where the hardware allows it: DPP adds inside a 16-lane row (quad xor 1,
// quad xor 2, half-row mirror, row mirror), ds_swizzle for lane ^ 16, one ds_bpermute for lane ^ 32.  Every lane ends
// with the total; fixed order.  (A __shfl_xor butterfly is one ds_bpermute round trip per step: measured 5x slower.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int L>
__device__ __forceinline__ float group_sum_dpp(float v) {  // sum over aligned groups of L = 4, 8, 16 or 32 lanes
    v = dpp_add<0xB1>(v);                  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                  // quad_perm [2,3,0,1]
    if (L >= 8) v = dpp_add<0x141>(v);     // row_half_mirror
    if (L >= 16) v = dpp_add<0x140>(v);    // row_mirror
    if (L >= 32) v = v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // lane ^ 16
    return v;
}
__device__ __forceinline__ float reduce16_dpp(float v) { return group_sum_dpp<16>(v); }
__device__ __forceinline__ float swap16_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // lane ^ 16
}
__device__ __forceinline__ float wave_sum_f32(float v) {  // all 64 lanes
    v = swap16_add(reduce16_dpp(v));
    return v + __shfl_xor(v, 32, 64);
}

// A word that ANOTHER workgroup of the same launch overwrites once every workgroup has counted its departure (the running powers
// of Adam: read by all, advanced by the last one out).  The read-before-count order must hold in the instruction stream too: a
// plain load of a __restrict__ pointer may legally be re-issued by the compiler BEHIND the relaxed departure atomic.
// An atomic load is performed exactly once, where it stands; `depart_barrier()` keeps every earlier access in front of the count.
__device__ __forceinline__ float load_once(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void depart_barrier() { asm volatile("" ::: "memory"); }

// A store of a result that the NEXT launch reads (partial gradient rows, above all): device-scope write-through (sc1) instead of a
// plain store.  With plain stores a kernel leaves its output as dirty lines in the eight L2s, and the end of the kernel has to write
// them back before the next launch may start: measured on the headline's gradient kernel (3.4 MB of partial rows per launch), the
// same bits in -3.2 % of the step (profiles/raw_r05/partial_row_store_policy_ab.txt; non-temporal stores: no gain).
__device__ __forceinline__ void store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... where the rows are MANY: with 16 rows (1 MB, the 3-layer DQN learner at batch 512) the plain form was 0.6 us per vec-step faster, with
// 128 rows (9 MB, batch 4096) the write-through form 1.0 us (profiles/raw_r05/dqn_partial_row_store_policy_ab.txt).  wt: uniform.
__device__ __forceinline__ void store_row(float* p, float v, bool wt) {
    if (wt) store_wt(p, v);
    else *p = v;
}

// Float64 cross-lane sums with the SAME addition trees as the __shfl loops they replace (bit-identical results), where the four
// steps inside a 16-lane row are DPP moves of the two halves of the double instead of ds_bpermute round trips (each shuffled double
// is two ds_bpermute_b32 with a dependent wait: ~1 us per six-step chain in the optimiser tails, profiles/r05_dqn_vec_step.md).
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_mov_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
// v += __shfl_down(v, o) for o = 32, 16, 8, 4, 2, 1: the total is valid in LANE 0 only (the lanes an out-of-row shift leaves alone
// hold partial garbage, as the upper lanes of the shuffle loop do)
__device__ __forceinline__ double wave_sum_down_f64_lane0(double v) {
    v += __shfl_down(v, 32, 64);
    v += __shfl_down(v, 16, 64);
    v += dpp_mov_f64<0x108, 0xF>(v, v);  // row_shl:8  lane i <- lane i + 8
    v += dpp_mov_f64<0x104, 0xF>(v, v);  // row_shl:4
    v += dpp_mov_f64<0x102, 0xF>(v, v);  // row_shl:2
    v += dpp_mov_f64<0x101, 0xF>(v, v);  // row_shl:1
    return v;
}
// v += __shfl_xor(v, o) for o = 32, 16, 8, 4, 2, 1: every lane ends with the total
__device__ __forceinline__ double wave_sum_xor_f64(double v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += dpp_mov_f64<0x128, 0xF>(v, v);  // row_ror:8  lane i <- lane (i + 8) % 16 = i ^ 8
    {
        double t = dpp_mov_f64<0x104, 0x5>(v, v);  // banks 0, 2 (lanes 0-3, 8-11 of a row) <- lane i + 4
        t = dpp_mov_f64<0x114, 0xA>(t, v);         // banks 1, 3 <- lane i - 4 (row_shr:4): t[i] = v[i ^ 4]
        v += t;
    }
    v += dpp_mov_f64<0x4E, 0xF>(v, v);  // quad_perm [2,3,0,1] = i ^ 2
    v += dpp_mov_f64<0xB1, 0xF>(v, v);  // quad_perm [1,0,3,2] = i ^ 1
    return v;
}

// block_sum of optim_device.h (wave sums by the __shfl_down tree, the wave totals added in ascending order; result in every
// thread) with the wave sums on wave_sum_down_f64_lane0: the same additions in the same order, bit-identical.  scratch: >= 16 doubles.
__device__ __forceinline__ double block_sum_f64_dpp(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum_down_f64_lane0(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += scratch[w];
    __syncthreads();
    return t;
}

// 16-byte non-temporal accesses for pure streaming kernels (every byte touched once, working set >> caches)
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ nt_u32x4 nt_load16(const void* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
}
__device__ __forceinline__ void nt_store16(void* p, nt_u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<nt_u32x4*>(p));
}

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint64_t seed, uint32_t idx, uint32_t blk,
                                                        uint32_t t, uint32_t tag) {
    uint32_t c0 = idx, c1 = blk, c2 = t, c3 = tag;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0;
        c1 = (uint32_t)p1;
        c2 = n2;
        c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ float u01_f32(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }
__host__ __device__ __forceinline__ double u01_f64(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (double)(x >> 11) * 0x1p-53;
}
__host__ __device__ __forceinline__ uint32_t randint32(uint32_t w, uint32_t n) {
    return (uint32_t)(((uint64_t)w * (uint64_t)n) >> 32);
}

// Keyed bijection on [0, n): 6-round Feistel over the enclosing power of 4 + cycle walking
// (the `shuffle(rng, 1:n)` stand-in; every index appears exactly once per epoch).
struct PermKeys {
    uint32_t k[8];
    uint32_t h, mask, n;
};

__host__ __device__ __forceinline__ PermKeys perm_keys(uint64_t seed, uint32_t epoch, uint32_t n) {
    PermKeys pk;
    u32x4 a = philox4x32_10(seed, 0, 0, epoch, TAG_SHUFFLE);
    u32x4 b = philox4x32_10(seed, 0, 1, epoch, TAG_SHUFFLE);
    pk.k[0] = a.x; pk.k[1] = a.y; pk.k[2] = a.z; pk.k[3] = a.w;
    pk.k[4] = b.x; pk.k[5] = b.y; pk.k[6] = b.z; pk.k[7] = b.w;
    uint32_t h = 1;
    while (h < 16 && (1ull << (2 * h)) < (uint64_t)n) ++h;
    pk.h = h;
    pk.mask = (1u << h) - 1u;
    pk.n = n;
    return pk;
}

__host__ __device__ __forceinline__ uint32_t feistel_f(uint32_t r, uint32_t k) {
    uint32_t f = r * 0x9E3779B1u + k;
    f ^= f >> 15;
    f *= 0x85EBCA77u;
    f ^= f >> 13;
    f *= 0xC2B2AE3Du;
    f ^= f >> 16;
    return f;
}

__host__ __device__ __forceinline__ uint32_t permute(const PermKeys& pk, uint32_t i) {
    if (pk.n <= 1) return 0;
    uint32_t x = i;
    do {
        uint32_t L = (x >> pk.h) & pk.mask, R = x & pk.mask;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            uint32_t nl = R;
            uint32_t nr = L ^ (feistel_f(R, pk.k[r]) & pk.mask);
            L = nl;
            R = nr;
        }
        x = (L << pk.h) | R;
    } while (x >= pk.n);
    return x;
}

}  // namespace rlhip
