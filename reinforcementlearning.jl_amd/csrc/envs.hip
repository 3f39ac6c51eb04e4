// envs.hip -- vectorised classic-control env kernels (reset!, act!, state) behind the C ABI.
//
// Replaces, for n env instances at once (one wavefront lane per env, SoA arrays in HBM):
//   reset!            RLEnvs/CartPoleEnv.jl:98-104, PendulumEnv.jl:84-92, MountainCarEnv.jl:99-105
//   act! / _step!     CartPoleEnv.jl:106-140,      PendulumEnv.jl:94-122, MountainCarEnv.jl:107-135
//   reward / is_terminated / state    CartPoleEnv.jl:84-86, PendulumEnv.jl:70,80-82, MountainCarEnv.jl:95-97
// and the thread-per-env loop of the historical MultiThreadEnv (blog index.md:347-376).
//
// Roofline: HBM-bound streaming kernel.  Algorithmic bytes per env-step: CartPole
// 24 read + 25 written = 49 B, Pendulum 45 B (with obs), MountainCar 33 B.  Each lane moves 16 B per
// array per access (EPL = 4 Float32 / 2 Float64 envs per lane -> global_load/store_dwordx4), the
// `episode` counters are touched only on the (rare) reset branch, and the grid is sized to cover the
// envs exactly (>> 256 workgroups at the sizes where HBM matters).
#include "env_device.h"

namespace rlhip {

template <typename T, int N>
struct alignas(sizeof(T) * N) VecN {
    T v[N];
};

// Every state array is read once and written once per step.  When the working set is far larger than the caches
// (NT = true, chosen by the host for n >= 2^20) the accesses are non-temporal -- no allocation in L2 on the way through:
// 182 -> 153 us per launch at 2^24 envs (4.5 -> 5.4 TB/s).  Small vector envs keep ordinary accesses: the next kernel
// of the loop (plan!, push!) reads what this one wrote straight from L2.
template <typename T, int N>
struct EvT {
    typedef T type __attribute__((ext_vector_type(N)));
};
template <typename T, int N, bool NT>
__device__ __forceinline__ VecN<T, N> ldv(const T* p, int64_t i) {
    if constexpr (!NT) {
        return *reinterpret_cast<const VecN<T, N>*>(p + i);
    } else {
    VecN<T, N> r;
    if constexpr (N == 1) {
        r.v[0] = __builtin_nontemporal_load(p + i);
    } else {
        typedef typename EvT<T, N>::type ev;
        ev x = __builtin_nontemporal_load(reinterpret_cast<const ev*>(p + i));
#pragma unroll
        for (int k = 0; k < N; ++k) r.v[k] = x[k];
    }
    return r;
    }
}
template <typename T, int N, bool NT>
__device__ __forceinline__ void stv(T* p, int64_t i, const VecN<T, N>& x) {
    if constexpr (!NT) {
        *reinterpret_cast<VecN<T, N>*>(p + i) = x;
    } else if constexpr (N == 1) {
        __builtin_nontemporal_store(x.v[0], p + i);
    } else {
        typedef typename EvT<T, N>::type ev;
        ev y;
#pragma unroll
        for (int k = 0; k < N; ++k) y[k] = x.v[k];
        __builtin_nontemporal_store(y, reinterpret_cast<ev*>(p + i));
    }
}

// PK (packed episode counters, rlhip_env_state.episode == NULL): the reset counter of an env lives in the bits of its
// step-counter word that max_steps leaves free -- t[i] = step | episode << tbits -- so an auto-reset touches no array the
// kernel does not stream anyway.  With a separate episode[] every reset is a scattered 4-byte read-modify-write = one
// 64-byte sector in and out: at the ~4.8 % of CartPole envs that terminate per step under a random policy more than
// half of all sectors of episode[] are hit, +9 % HBM traffic over the 49 algorithmic bytes (profiles/r01_pmc_env_step.md).
// NTW: the store policy of the arrays that are WRITTEN ONLY (reward, done, the observation planes).  The in-place arrays
// (state, step counter) keep non-temporal stores when NT (ordinary stores measured slower there, round 3); a pure write
// stream is better left to the L2 (GAE, gathers: round 3; this kernel: round 4, numbers in step_impl).
template <class P, typename T, int EPL, bool NT, bool PK, bool NTW = NT>
__global__ __launch_bounds__(256) void env_step_kernel(P p, EnvArrays<T> st, int64_t n,
                                                       const void* __restrict__ actions,
                                                       int auto_reset, uint64_t seed,
                                                       uint32_t env_id_base, T* __restrict__ last_obs,
                                                       T* __restrict__ obs_out, int tbits) {
    int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * EPL;
    if (base >= n) return;
    VecN<T, EPL> s[P::SDIM];
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) s[k] = ldv<T, EPL, NT>(st.s[k], base);
    VecN<int32_t, EPL> tv = ldv<int32_t, EPL, NT>(st.t, base);
    VecN<int32_t, EPL> ai;
    VecN<T, EPL> af;
    if (p.continuous) {
        af = ldv<T, EPL, NT>((const T*)actions, base);
#pragma unroll
        for (int j = 0; j < EPL; ++j) ai.v[j] = 0;
    } else {
        ai = ldv<int32_t, EPL, NT>((const int32_t*)actions, base);
#pragma unroll
        for (int j = 0; j < EPL; ++j) af.v[j] = (T)0;
    }
    VecN<T, EPL> rew;
    VecN<uint8_t, EPL> dn;
    VecN<T, EPL> lo[P::ODIM], oo[P::ODIM];
    uint32_t pend = 0;  // envs of this lane that terminated and restart right away
    uint32_t epv[EPL];  // their episode counters: requested as soon as `done` is known, so that the scattered load
                        // is in flight under the physics of the following envs instead of stalling the reset
                        // (PK: unpacked from the step-counter word, no load at all)
    const uint32_t tmask = PK ? ((1u << tbits) - 1u) : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < EPL; ++j) epv[j] = PK ? ((uint32_t)tv.v[j] >> tbits) : 0u;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        LaneState<T> e;
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) e.s[k] = s[k].v[j];
        e.t = (int32_t)((uint32_t)tv.v[j] & tmask);
        e.episode = epv[j];
        if constexpr (EnvTraits<P>::STEP_NOISE) {
            if (p.noise > (T)0) {  // act! draws from the env's rng: uniform keyed by (env, t, episode)
                if constexpr (!PK) e.episode = st.episode[base + j];
                af.v[j] = acrobot_noise_u(e, seed, env_id_base + (uint32_t)(base + j));
            }
        }
        T r;
        bool d;
        env_step1(p, e, ai.v[j], af.v[j], r, d);
        rew.v[j] = r;
        dn.v[j] = (uint8_t)d;
        if (last_obs) {
            T o[6];
            env_obs1(p, e, o);
#pragma unroll
            for (int k = 0; k < P::ODIM; ++k) lo[k].v[j] = o[k];
        }
        if (d && auto_reset) {
            pend |= 1u << j;
            if constexpr (EPL > 1 && !PK) epv[j] = st.episode[base + j];
        }
#pragma unroll
        for (int k = 0; k < P::SDIM; ++k) s[k].v[j] = e.s[k];
        tv.v[j] = e.t;
    }
    // MultiThreadEnv protocol: a terminated env starts a fresh episode right away; the reward / terminal of the
    // finished step stay visible in the reward / done arrays.  The reset (a Philox block per env) is the most
    // expensive part of the kernel and a branch per j would be taken by almost every wave (any of its 64 lanes);
    // instead every lane works off its own pending envs, one per trip: the trip count is the largest number of
    // terminated envs in one lane (1-2) rather than EPL.
    if constexpr (EPL == 1) {
        if (pend) {
            LaneState<T> e;
            if constexpr (PK) e.episode = epv[0];
            else e.episode = st.episode[base];
            env_reset1(p, e, seed, env_id_base + (uint32_t)base);
            if constexpr (PK) epv[0] = e.episode;
            else st.episode[base] = e.episode;
#pragma unroll
            for (int k = 0; k < P::SDIM; ++k) s[k].v[0] = e.s[k];
            tv.v[0] = e.t;
        }
    } else {
        while (__builtin_amdgcn_ballot_w64(pend != 0) != 0) {
            if (pend) {
                const int j = __builtin_ctz(pend);
                pend &= pend - 1;
                LaneState<T> e;
                uint32_t ep = epv[0];
#pragma unroll
                for (int jj = 1; jj < EPL; ++jj) ep = (jj == j) ? epv[jj] : ep;
                e.episode = ep;
                env_reset1(p, e, seed, env_id_base + (uint32_t)(base + j));
                if constexpr (!PK) st.episode[base + j] = e.episode;
#pragma unroll
                for (int jj = 0; jj < EPL; ++jj) {
#pragma unroll
                    for (int k = 0; k < P::SDIM; ++k) s[k].v[jj] = (jj == j) ? e.s[k] : s[k].v[jj];
                    tv.v[jj] = (jj == j) ? e.t : tv.v[jj];
                    if constexpr (PK) epv[jj] = (jj == j) ? e.episode : epv[jj];
                }
            }
        }
    }
    if (obs_out) {
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            LaneState<T> e;
#pragma unroll
            for (int k = 0; k < P::SDIM; ++k) e.s[k] = s[k].v[j];
            T o[6];
            env_obs1(p, e, o);
#pragma unroll
            for (int k = 0; k < P::ODIM; ++k) oo[k].v[j] = o[k];
        }
    }
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) stv<T, EPL, NT>(st.s[k], base, s[k]);
    if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < EPL; ++j)  // the counter SATURATES at its field's maximum (it never wraps into the step bits)
            tv.v[j] = (int32_t)((uint32_t)tv.v[j] | (min(epv[j], 0xFFFFFFFFu >> tbits) << tbits));
    }
    stv<int32_t, EPL, NT>(st.t, base, tv);
    stv<T, EPL, NTW>(st.reward, base, rew);
    stv<uint8_t, EPL, NTW>(st.done, base, dn);
    if (last_obs) {
#pragma unroll
        for (int k = 0; k < P::ODIM; ++k) stv<T, EPL, NTW>(last_obs + (int64_t)k * n, base, lo[k]);
    }
    if (obs_out) {
#pragma unroll
        for (int k = 0; k < P::ODIM; ++k) stv<T, EPL, NTW>(obs_out + (int64_t)k * n, base, oo[k]);
    }
}

template <class P, typename T>
__global__ __launch_bounds__(256) void env_reset_kernel(P p, EnvArrays<T> st, int64_t n, uint64_t seed,
                                                        uint32_t env_id_base,
                                                        const uint8_t* __restrict__ mask, int tbits) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    LaneState<T> e;
    const bool packed = st.episode == nullptr;
    e.episode = packed ? ((uint32_t)st.t[i] >> tbits) : st.episode[i];
    env_reset1(p, e, seed, env_id_base + (uint32_t)i);
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) st.s[k][i] = e.s[k];
    if (packed) {
        st.t[i] = (int32_t)(min(e.episode, 0xFFFFFFFFu >> tbits) << tbits);
    } else {
        st.t[i] = 0;
        st.episode[i] = e.episode;
    }
    st.done[i] = 0;          // reset!: done = false
    st.reward[i] = (T)EnvTraits<P>::RESET_REWARD;
}

template <class P, typename T>
__global__ __launch_bounds__(256) void env_obs_kernel(P p, EnvArrays<T> st, int64_t n,
                                                      T* __restrict__ obs) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LaneState<T> e;
#pragma unroll
    for (int k = 0; k < P::SDIM; ++k) e.s[k] = st.s[k][i];
    T o[6];
    env_obs1(p, e, o);
#pragma unroll
    for (int k = 0; k < P::ODIM; ++k) obs[(int64_t)k * n + i] = o[k];
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// bits of the step-counter word that hold the step count in packed mode: t can reach max_steps + 1 (CartPole's strict `>`)
static int packed_tbits(int64_t max_steps) {
    int b = 1;
    while (b < 31 && ((int64_t)1 << b) <= max_steps + 1) ++b;
    return b;
}

template <class P, typename T>
static int32_t step_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n,
                         const void* actions, int32_t auto_reset, uint64_t seed, uint32_t env_id_base,
                         void* last_obs, void* obs_out, hipStream_t stream) {
    P p = P::make(*cfg);
    EnvArrays<T> a = EnvArrays<T>::from(*st);
    constexpr int EPL = 16 / sizeof(T);
    bool vec = (n % EPL == 0) && aligned16(actions) && aligned16(st->t) && aligned16(st->reward);
    for (int k = 0; k < P::SDIM; ++k) vec = vec && aligned16(st->s[k]);
    // done is u8: an EPL-byte vector store needs EPL-byte alignment only
    vec = vec && (((uintptr_t)st->done % EPL) == 0);
    if (last_obs) vec = vec && aligned16(last_obs);
    if (obs_out) vec = vec && aligned16(obs_out);
    const bool streaming = n >= ((int64_t)1 << 20);  // state arrays beyond the L2: non-temporal accesses
    const bool packed = st->episode == nullptr;
    const int tbits = packed_tbits(p.max_steps);
    RLHIP_REQUIRE(!packed || tbits <= 20, "packed episode counters need max_steps < 2^20 - 1");
#define STEP_LAUNCH(EPL_, NT_, PK_, GRID_)                                                                               \
    hipLaunchKernelGGL((env_step_kernel<P, T, EPL_, NT_, PK_>), dim3(GRID_), dim3(256), 0, stream, p, a, n, actions,      \
                       auto_reset, seed, env_id_base, (T*)last_obs, (T*)obs_out, tbits)
    if (vec) {
        int64_t lanes = n / EPL;
        int grid = (int)((lanes + 255) / 256);
        if (streaming) {
            // in-place arrays non-temporal, write-only arrays ORDINARY stores: same-box A / B at 2^24 envs (round 4,
            // tools/envstep_wo_ab.py): CartPole 139.5 -> 128.6 - 130.4 us (0.735 -> 0.79 - 0.80 of 8 TB/s), MountainCar
            // 91.6 -> 84.3 us (0.755 -> 0.82); Pendulum + observation planes unchanged (156 us: its Float64 trig binds it)
            if (packed)
                hipLaunchKernelGGL((env_step_kernel<P, T, EPL, true, true, false>), dim3(grid), dim3(256), 0, stream, p, a, n,
                                   actions, auto_reset, seed, env_id_base, (T*)last_obs, (T*)obs_out, tbits);
            else
                hipLaunchKernelGGL((env_step_kernel<P, T, EPL, true, false, false>), dim3(grid), dim3(256), 0, stream, p, a, n,
                                   actions, auto_reset, seed, env_id_base, (T*)last_obs, (T*)obs_out, tbits);
        } else {
            if (packed) STEP_LAUNCH(EPL, false, true, grid);
            else STEP_LAUNCH(EPL, false, false, grid);
        }
    } else {
        int grid = (int)((n + 255) / 256);
        if (packed) STEP_LAUNCH(1, false, true, grid);
        else STEP_LAUNCH(1, false, false, grid);
    }
#undef STEP_LAUNCH
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

template <class P, typename T>
static int32_t reset_impl(const typename P::cfg_t* cfg, const rlhip_env_state* st, int64_t n,
                          uint64_t seed, uint32_t env_id_base, const uint8_t* mask,
                          hipStream_t stream) {
    P p = P::make(*cfg);
    EnvArrays<T> a = EnvArrays<T>::from(*st);
    const int tbits = packed_tbits(p.max_steps);
    RLHIP_REQUIRE(st->episode != nullptr || tbits <= 20, "packed episode counters need max_steps < 2^20 - 1");
    hipLaunchKernelGGL((env_reset_kernel<P, T>), dim3((int)((n + 255) / 256)), dim3(256), 0, stream, p,
                       a, n, seed, env_id_base, mask, tbits);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

template <class P, typename T>
static int32_t obs_impl(const rlhip_env_state* st, int64_t n, void* obs, hipStream_t stream) {
    P p{};
    EnvArrays<T> a = EnvArrays<T>::from(*st);
    hipLaunchKernelGGL((env_obs_kernel<P, T>), dim3((int)((n + 255) / 256)), dim3(256), 0, stream, p, a,
                       n, (T*)obs);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}

static int32_t check_state(int32_t kind, const rlhip_env_state* st, int64_t n, bool allow_packed = false) {
    RLHIP_REQUIRE(kind >= 0 && kind <= 3, "kind must be 0 (cartpole), 1 (pendulum), 2 (mountaincar) or 3 (acrobot)");
    RLHIP_REQUIRE(st != nullptr, "env state is NULL");
    RLHIP_REQUIRE(n >= 0 && n <= 0xFFFFFFFFll, "n out of range");
    int sd = (kind == 0 || kind == 3) ? 4 : 2;
    for (int k = 0; k < sd; ++k) RLHIP_REQUIRE(st->s[k] != nullptr, "state array is NULL");
    RLHIP_REQUIRE(st->t && st->done && st->reward && (st->episode || allow_packed), "state array is NULL");
    return RLHIP_OK;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" {

int32_t rlhip_cartpole_default(rlhip_cartpole_cfg* c) {
    RLHIP_REQUIRE(c != nullptr, "cfg is NULL");
    // RLEnvs/CartPoleEnv.jl:22-32
    c->gravity = 9.8;
    c->masscart = 1.0;
    c->masspole = 0.1;
    c->halflength = 0.5;
    c->forcemag = 10.0;
    c->dt = 0.02;
    c->thetathreshold_deg = 12.0;
    c->xthreshold = 2.4;
    c->max_steps = 200;
    c->continuous = 0;
    return RLHIP_OK;
}

int32_t rlhip_pendulum_default(rlhip_pendulum_cfg* c) {
    RLHIP_REQUIRE(c != nullptr, "cfg is NULL");
    // RLEnvs/PendulumEnv.jl:41-53
    c->max_speed = 8;
    c->max_torque = 2;
    c->g = 10;
    c->m = 1;
    c->l = 1;
    c->dt = 0.05;
    c->max_steps = 200;
    c->continuous = 1;
    c->n_actions = 3;
    return RLHIP_OK;
}

int32_t rlhip_mountaincar_default(rlhip_mountaincar_cfg* c, int32_t continuous) {
    RLHIP_REQUIRE(c != nullptr, "cfg is NULL");
    // RLEnvs/MountainCarEnv.jl:19-29; continuous overrides :74
    c->min_pos = -1.2;
    c->max_pos = 0.6;
    c->max_speed = 0.07;
    c->goal_pos = continuous ? 0.45 : 0.5;
    c->goal_velocity = 0.0;
    c->power = continuous ? 0.0015 : 0.001;
    c->gravity = 0.0025;
    c->max_steps = 200;
    c->continuous = continuous ? 1 : 0;
    return RLHIP_OK;
}

int32_t rlhip_acrobot_default(rlhip_acrobot_cfg* c) {
    RLHIP_REQUIRE(c != nullptr, "cfg is NULL");
    // RLEnvs/src/environments/3rd_party/AcrobotEnv.jl:22-40
    c->link_length_a = 1.0;
    c->link_length_b = 1.0;
    c->link_mass_a = 1.0;
    c->link_mass_b = 1.0;
    c->link_com_pos_a = 0.5;
    c->link_com_pos_b = 0.5;
    c->link_moi = 1.0;
    c->max_torque_noise = 0.0;
    c->max_vel_a = 4 * RLHIP_PI;
    c->max_vel_b = 9 * RLHIP_PI;
    c->g = 9.8;
    c->dt = 0.2;
    c->max_steps = 200;
    c->nips = 0;
    return RLHIP_OK;
}

/* resets an instance can count in packed mode before its counter saturates (then every further reset of that instance
 * re-draws the same initial state): 2^(32 - tbits) - 1.  A host that steps that long must use the episode[] array. */
int64_t rlhip_env_packed_episode_capacity(int64_t max_steps) {
    if (max_steps < 1) return -1;
    const int tbits = packed_tbits(max_steps);
    if (tbits > 20) return -1;
    return ((int64_t)1 << (32 - tbits)) - 1;
}

int32_t rlhip_env_obs_dim(int32_t kind) { return kind == 0 ? 4 : (kind == 1 ? 3 : (kind == 2 ? 2 : 6)); }
int32_t rlhip_env_state_dim(int32_t kind) { return (kind == 0 || kind == 3) ? 4 : 2; }

int32_t rlhip_env_reset(int32_t kind, int32_t is_f64, const void* cfg, const rlhip_env_state* st,
                        int64_t n, uint64_t seed, uint32_t env_id_base, const uint8_t* mask,
                        rlhip_stream_t stream) {
    int32_t rc = check_state(kind, st, n, /*allow_packed=*/true);
    if (rc) return rc;
    RLHIP_REQUIRE(cfg != nullptr, "cfg is NULL");
    if (n == 0) return RLHIP_OK;
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return is_f64 ? reset_impl<CartPoleParams<double>, double>((const rlhip_cartpole_cfg*)cfg, st, n, seed, env_id_base, mask, s)
                      : reset_impl<CartPoleParams<float>, float>((const rlhip_cartpole_cfg*)cfg, st, n, seed, env_id_base, mask, s);
    if (kind == 1)
        return is_f64 ? reset_impl<PendulumParams<double>, double>((const rlhip_pendulum_cfg*)cfg, st, n, seed, env_id_base, mask, s)
                      : reset_impl<PendulumParams<float>, float>((const rlhip_pendulum_cfg*)cfg, st, n, seed, env_id_base, mask, s);
    if (kind == 3)
        return is_f64 ? reset_impl<AcrobotParams<double>, double>((const rlhip_acrobot_cfg*)cfg, st, n, seed, env_id_base, mask, s)
                      : reset_impl<AcrobotParams<float>, float>((const rlhip_acrobot_cfg*)cfg, st, n, seed, env_id_base, mask, s);
    return is_f64 ? reset_impl<MountainCarParams<double>, double>((const rlhip_mountaincar_cfg*)cfg, st, n, seed, env_id_base, mask, s)
                  : reset_impl<MountainCarParams<float>, float>((const rlhip_mountaincar_cfg*)cfg, st, n, seed, env_id_base, mask, s);
}

int32_t rlhip_env_step(int32_t kind, int32_t is_f64, const void* cfg, const rlhip_env_state* st,
                       int64_t n, const void* actions, int32_t auto_reset, uint64_t seed,
                       uint32_t env_id_base, void* last_obs, void* obs_out, rlhip_stream_t stream) {
    int32_t rc = check_state(kind, st, n, /*allow_packed=*/true);
    if (rc) return rc;
    RLHIP_REQUIRE(cfg != nullptr, "cfg is NULL");
    RLHIP_REQUIRE(actions != nullptr, "actions is NULL");
    if (n == 0) return RLHIP_OK;
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return is_f64 ? step_impl<CartPoleParams<double>, double>((const rlhip_cartpole_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s)
                      : step_impl<CartPoleParams<float>, float>((const rlhip_cartpole_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s);
    if (kind == 1)
        return is_f64 ? step_impl<PendulumParams<double>, double>((const rlhip_pendulum_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s)
                      : step_impl<PendulumParams<float>, float>((const rlhip_pendulum_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s);
    if (kind == 3)
        return is_f64 ? step_impl<AcrobotParams<double>, double>((const rlhip_acrobot_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s)
                      : step_impl<AcrobotParams<float>, float>((const rlhip_acrobot_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s);
    return is_f64 ? step_impl<MountainCarParams<double>, double>((const rlhip_mountaincar_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s)
                  : step_impl<MountainCarParams<float>, float>((const rlhip_mountaincar_cfg*)cfg, st, n, actions, auto_reset, seed, env_id_base, last_obs, obs_out, s);
}

int32_t rlhip_env_obs(int32_t kind, int32_t is_f64, const rlhip_env_state* st, int64_t n, void* obs,
                      rlhip_stream_t stream) {
    int32_t rc = check_state(kind, st, n, /*allow_packed=*/true);
    if (rc) return rc;
    RLHIP_REQUIRE(obs != nullptr, "obs is NULL");
    if (n == 0) return RLHIP_OK;
    hipStream_t s = as_stream(stream);
    if (kind == 0)
        return is_f64 ? obs_impl<CartPoleParams<double>, double>(st, n, obs, s) : obs_impl<CartPoleParams<float>, float>(st, n, obs, s);
    if (kind == 1)
        return is_f64 ? obs_impl<PendulumParams<double>, double>(st, n, obs, s) : obs_impl<PendulumParams<float>, float>(st, n, obs, s);
    if (kind == 3)
        return is_f64 ? obs_impl<AcrobotParams<double>, double>(st, n, obs, s) : obs_impl<AcrobotParams<float>, float>(st, n, obs, s);
    return is_f64 ? obs_impl<MountainCarParams<double>, double>(st, n, obs, s) : obs_impl<MountainCarParams<float>, float>(st, n, obs, s);
}

}  // extern "C"
