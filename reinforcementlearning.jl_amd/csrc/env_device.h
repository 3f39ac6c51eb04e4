// env_device.h -- per-lane physics of the classic-control envs + Acrobot (device inline functions).
//
// One wavefront lane owns one env instance; state lives in registers between load and store.  Used
// by the stand-alone env kernels (envs.hip) and by the fused rollout kernel (ppo.hip), so both paths
// produce bit-identical trajectories.
//
// The element type T is Float32 or Float64 like `CartPoleEnv(; T = ...)`.  With T = Float32 the
// reference silently promotes a few sub-expressions to Float64 (Float64 literals `4 / 3`, `0.1`,
// `0.001`, `0.2`, `0.6`, `2 * pi`).  Those promotions are reproduced here --
// the file is compiled with -ffp-contract=off so no a*b+c is fused behind our back.
//
// Trigonometry: the reference's sin/cos(::Float32) (Julia Base, double-precision kernel, rounded
// once) is correctly rounded in all but ~1e-9 of inputs; evaluating in Float64 and rounding to
// Float32 reproduces that, where the GPU's native sinf/cosf (1-2 ulp) would not.  The open-loop
// pole amplifies a 1-ulp difference by ~1e6 over an episode (Appendix A.7), so this matters.
#pragma once
#include "common.h"

namespace rlhip {

#ifndef RLHIP_PI
#define RLHIP_PI 3.14159265358979323846
#endif

// Float64 sin / cos on [-pi/4, pi/4] without range reduction: the fdlibm kernel polynomials
// (k_sin.c / k_cos.c coefficient sets, |error| < 2^-58), ~18 Float64 FMAs instead of the ~150
// instructions of the general ocml sincos.  CartPole's theta lives in +-0.42 rad (2 x the 12 degree
// threshold), so the env-step kernel always takes this path; it was VALU-bound on the general routine
// (profiles/r01_final_bench_stats.md: 156-223 us per 2^24-env launch against a 130 us HBM floor).
// The coefficients come from constant memory through scalar loads: as SGPR pairs they are operands of v_fma_f64,
// whereas 64-bit literals cost two v_mov_b32 per FMA (measured: 24 of the ~160 VALU instructions of a CartPole step).
__constant__ double SINCOS_SMALL_COEF[12] = {
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,  -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01, -1.13596475577881948265e-11, 2.08757232129817482790e-09,
    -2.75573143513906633035e-07, 2.48015872894767294178e-05, -1.38888888888741095749e-03, 4.16666666666666019037e-02};

// a * b + c with the addend in an SGPR pair (VOP3 form).  The compiler selects v_fmac_f64 for a Horner step and copies
// the coefficient into the accumulator first (two v_mov_b32 per step); the three-operand form needs neither.
__device__ __forceinline__ double fma_sgpr_addend(double a, double b, double c_uniform) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}

__device__ __forceinline__ void sincos_small_f64(double x, double* s, double* c) {
    const double* __restrict__ k = SINCOS_SMALL_COEF;
    const double z = x * x;
    double ps = fma(z, k[0], k[1]);
    ps = fma_sgpr_addend(z, ps, k[2]);
    ps = fma_sgpr_addend(z, ps, k[3]);
    ps = fma_sgpr_addend(z, ps, k[4]);
    ps = fma_sgpr_addend(z, ps, k[5]);
    *s = fma(x * z, ps, x);
    double pc = fma(z, k[6], k[7]);
    pc = fma_sgpr_addend(z, pc, k[8]);
    pc = fma_sgpr_addend(z, pc, k[9]);
    pc = fma_sgpr_addend(z, pc, k[10]);
    pc = fma_sgpr_addend(z, pc, k[11]);
    *c = fma(z * z, pc, fma(z, -0.5, 1.0));
}

// sin / cos of a Float32 ARGUMENT, evaluated in Float64 and rounded once by the caller -- what Julia's own sin(::Float32) does
// (a double-precision kernel behind a Float32 interface) and what the oracle restates as (float) sin((double) x).
//   |x| <= pi/4      : the kernel polynomials directly (CartPole's theta)
//   |x| <= 2^16      : x = n pi/2 + r + t with n = rint(x 2/pi); r0 = x - n PIO2_HI is EXACT in one fma (x has 24 significant
//                      bits, n < 2^17, the difference is below 1 with its last bit at 2^-52 or above); the second term of pi/2 and
//                      the rounding of its product are carried as a tail t (|t| < 1e-11) and enter to first order:
//                      sin(r + t) = sin r + t cos r, cos(r + t) = cos r - t sin r.  ~45 Float64 instructions against the ~130 of
//                      ocml's general sincos (Payne-Hanek branch and all); Pendulum's theta and MountainCar's 3 x live here.
//   otherwise        : ocml.
// tools/micro/trig_f32arg.hip runs EVERY Float32 with |x| <= 2^16 (2.4e9 values) on the GPU: the Float32 roundings of sin and
// cos equal the host libm's -- the oracle's -- for all of them, so the two sides agree by enumeration there, not by accuracy class.
constexpr double TRIG_PIO2_HI = 1.57079632679489655800e+00;  // RN(pi / 2)
constexpr double TRIG_PIO2_LO = 6.12323399573676603587e-17;  // RN(pi / 2 - PIO2_HI)
constexpr double TRIG_2_OVER_PI = 6.36619772367581382433e-01;
constexpr float TRIG_MEDIUM_MAX = 65536.0f;

__device__ __forceinline__ void sincos_medium_f64(double x, double* s, double* c) {
    const double n = ::rint(x * TRIG_2_OVER_PI);
    const double r0 = ::fma(-n, TRIG_PIO2_HI, x);  // exact
    const double w = n * TRIG_PIO2_LO;
    const double wl = ::fma(n, TRIG_PIO2_LO, -w);  // the product's rounding error, exact
    const double r = r0 - w;
    const double t = ((r0 - r) - w) - wl;
    double sr, cr;
    sincos_small_f64(r, &sr, &cr);
    const double st = ::fma(t, cr, sr), ct = ::fma(-t, sr, cr);
    const int q = (int)n & 3;  // two's complement: the quadrant of negative n as well
    const double a = (q & 1) ? ct : st, b = (q & 1) ? st : ct;
    *s = (q & 2) ? -a : a;                // q: 0 s, 1 c, 2 -s, 3 -c
    *c = ((q + 1) & 2) ? -b : b;          // q: 0 c, 1 -s, 2 -c, 3 s
}

__device__ __forceinline__ void sincos_f32arg(float x, double* s, double* c) {
    const float ax = fabsf(x);
    if (ax <= 0.78539816f) sincos_small_f64((double)x, s, c);  // (sin(-0) comes out as +0: the one sign libm gives differently)
    else if (ax <= TRIG_MEDIUM_MAX) sincos_medium_f64((double)x, s, c);
    else ::sincos((double)x, s, c);
}

template <typename T>
struct Trig;
template <>
struct Trig<float> {
    static __device__ __forceinline__ float sin_(float x) {
        double ds, dc;
        sincos_f32arg(x, &ds, &dc);
        return (float)ds;
    }
    static __device__ __forceinline__ float cos_(float x) {
        double ds, dc;
        sincos_f32arg(x, &ds, &dc);
        return (float)dc;
    }
    static __device__ __forceinline__ void sincos_(float x, float* s, float* c) {
        double ds, dc;
        sincos_f32arg(x, &ds, &dc);
        *s = (float)ds;
        *c = (float)dc;
    }
};
template <>
struct Trig<double> {
    static __device__ __forceinline__ double sin_(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos_(double x) { return ::cos(x); }
    static __device__ __forceinline__ void sincos_(double x, double* s, double* c) { ::sincos(x, s, c); }
};

// x / c for a divisor c that is a parameter of the env (known on the host): with rc = RN(1 / c), q0 = RN(x rc),
// r = x - q0 c (exact: one FMA) and q = RN(q0 + r rc), q is the correctly rounded quotient (Markstein's theorem; it
// is also the last step of the hardware division sequence) -- 3 instructions instead of the ~10 (Float32) / ~35
// (Float64) of a general division.  The env-step kernel issues ~1000 VALU instructions per wave against ~130 us of HBM
// time at 2^24 envs: instruction issue, not bandwidth, was what kept it at 150 us.
__device__ __forceinline__ float div_const(float x, float c, float rc) {
    const float q0 = x * rc;
    const float r = fmaf(-q0, c, x);
    return fmaf(r, rc, q0);
}
__device__ __forceinline__ double div_const(double x, double c, double rc) {
    const double q0 = x * rc;
    const double r = fma(-q0, c, x);
    return fma(r, rc, q0);
}

template <typename T>
__device__ __forceinline__ T clampT(T x, T lo, T hi) {
    return (x > hi) ? hi : ((x < lo) ? lo : x);  // Base.clamp
}

// per-lane uniform draws of reset!: rand(rng, T) stand-ins from the RESET stream
template <typename T>
struct ResetDraw;
template <>
struct ResetDraw<float> {
    // up to 4 uniforms from one Philox block
    static __device__ __forceinline__ void draw4(uint64_t seed, uint32_t id, uint32_t ep, float u[4]) {
        u32x4 w = philox4x32_10(seed, id, 0, ep, TAG_RESET);
        u[0] = u01_f32(w.x);
        u[1] = u01_f32(w.y);
        u[2] = u01_f32(w.z);
        u[3] = u01_f32(w.w);
    }
    static __device__ __forceinline__ void draw2(uint64_t seed, uint32_t id, uint32_t ep, float u[2]) {
        u32x4 w = philox4x32_10(seed, id, 0, ep, TAG_RESET);
        u[0] = u01_f32(w.x);
        u[1] = u01_f32(w.y);
    }
};
template <>
struct ResetDraw<double> {
    static __device__ __forceinline__ void draw4(uint64_t seed, uint32_t id, uint32_t ep, double u[4]) {
        u32x4 a = philox4x32_10(seed, id, 0, ep, TAG_RESET);
        u32x4 b = philox4x32_10(seed, id, 1, ep, TAG_RESET);
        u[0] = u01_f64(a.x, a.y);
        u[1] = u01_f64(a.z, a.w);
        u[2] = u01_f64(b.x, b.y);
        u[3] = u01_f64(b.z, b.w);
    }
    static __device__ __forceinline__ void draw2(uint64_t seed, uint32_t id, uint32_t ep, double u[2]) {
        u32x4 a = philox4x32_10(seed, id, 0, ep, TAG_RESET);
        u[0] = u01_f64(a.x, a.y);
        u[1] = u01_f64(a.z, a.w);
    }
};

// --------------------------------------------------------------------------------- CartPole --
template <typename T>
struct CartPoleParams {  // CartPoleEnvParams{T}  RLEnvs/CartPoleEnv.jl:3-15
    T gravity, masscart, masspole, totalmass, halflength, polemasslength, forcemag, dt,
        thetathreshold, xthreshold;
    T rtotalmass;        // RN(1 / totalmass) in T, for div_const
    double rtotalmass_d;  // RN(1 / Float64(totalmass))
    int32_t max_steps;
    int32_t continuous;
    static constexpr int SDIM = 4;
    static constexpr int ODIM = 4;
    static constexpr int KIND = RLHIP_ENV_CARTPOLE;
    typedef rlhip_cartpole_cfg cfg_t;

    // CartPoleEnvParams{T}(; kwargs...)  :22-46 -- derive in Float64, then convert to T
    static CartPoleParams make(const rlhip_cartpole_cfg& c) {
        CartPoleParams p;
        p.gravity = (T)c.gravity;
        p.masscart = (T)c.masscart;
        p.masspole = (T)c.masspole;
        p.totalmass = (T)(c.masscart + c.masspole);
        p.rtotalmass = (T)1 / p.totalmass;
        p.rtotalmass_d = 1.0 / (double)p.totalmass;
        p.halflength = (T)c.halflength;
        p.polemasslength = (T)(c.masspole * c.halflength);
        p.forcemag = (T)c.forcemag;
        p.dt = (T)c.dt;
        p.thetathreshold = (T)(c.thetathreshold_deg * RLHIP_PI / 180);
        p.xthreshold = (T)c.xthreshold;
        p.max_steps = (int32_t)c.max_steps;
        p.continuous = c.continuous;
        return p;
    }
    bool is_continuous() const { return continuous != 0; }
};

template <typename T>
struct LaneState {
    T s[4];
    int32_t t;
    uint32_t episode;
};

// reset!  :98-104   state = T(0.1) * rand(rng, T, 4) .- T(0.05); t = 0
template <typename T>
__device__ __forceinline__ void env_reset1(const CartPoleParams<T>&, LaneState<T>& e, uint64_t seed,
                                           uint32_t id) {
    T u[4];
    ResetDraw<T>::draw4(seed, id, e.episode, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.s[k] = (T)0.1 * u[k] - (T)0.05;
    e.t = 0;
    e.episode += 1;
}

// act! + _step!  :106-140, reward :84.  action: raw pointer element already converted by the caller:
//   discrete -> a in {0,1} (Julia 1,2);  continuous -> a of type T
template <typename T>
__device__ __forceinline__ void env_step1(const CartPoleParams<T>& p, LaneState<T>& e, int32_t ai, T af,
                                          T& reward, bool& done) {
    T a = p.continuous ? af : ((ai == 1) ? (T)1 : (T)-1);  // :115  a == 2 ? 1 : -1
    e.t += 1;                                              // :119
    T force = a * p.forcemag;                              // :120
    T x = e.s[0], xdot = e.s[1], theta = e.s[2], thetadot = e.s[3];  // :121 (pre-step values)
    T sintheta, costheta;
    Trig<T>::sincos_(theta, &sintheta, &costheta);         // :122-123
    T tmp = div_const(force + p.polemasslength * (thetadot * thetadot) * sintheta, p.totalmass, p.rtotalmass);  // :124
    // :125-129  the literal 4 / 3 is Float64: denominator, thetaacc and xacc are Float64
    T num = p.gravity * sintheta - costheta * tmp;
    T frac = div_const(p.masspole * (costheta * costheta), p.totalmass, p.rtotalmass);
    double den = (double)p.halflength * (4.0 / 3.0 - (double)frac);
    double thetaacc = (double)num / den;
    double xacc = (double)tmp - div_const((double)p.polemasslength * thetaacc * (double)costheta, (double)p.totalmass,
                                          p.rtotalmass_d);  // :130
    e.s[0] = x + p.dt * xdot;                                 // :131
    e.s[1] = (T)((double)xdot + (double)p.dt * xacc);         // :132
    e.s[2] = theta + p.dt * thetadot;                         // :133
    e.s[3] = (T)((double)thetadot + (double)p.dt * thetaacc); // :134
    T ax = e.s[0] < (T)0 ? -e.s[0] : e.s[0];
    T ath = e.s[2] < (T)0 ? -e.s[2] : e.s[2];
    done = ax > p.xthreshold || ath > p.thetathreshold || e.t > p.max_steps;  // :135-138
    reward = done ? (T)0 : (T)1;                                             // :84
}

template <typename T>
__device__ __forceinline__ void env_obs1(const CartPoleParams<T>&, const LaneState<T>& e, T o[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = e.s[k];  // :86
}

// --------------------------------------------------------------------------------- Pendulum --
template <typename T>
struct PendulumParams {  // PendulumEnvParams{T}  RLEnvs/PendulumEnv.jl:3-11
    T max_speed, max_torque, g, m, l, dt;
    int32_t max_steps;
    int32_t continuous, n_actions;
    static constexpr int SDIM = 2;
    static constexpr int ODIM = 3;
    static constexpr int KIND = RLHIP_ENV_PENDULUM;
    typedef rlhip_pendulum_cfg cfg_t;
    static PendulumParams make(const rlhip_pendulum_cfg& c) {
        PendulumParams p;
        p.max_speed = (T)c.max_speed;
        p.max_torque = (T)c.max_torque;
        p.g = (T)c.g;
        p.m = (T)c.m;
        p.l = (T)c.l;
        p.dt = (T)c.dt;
        p.max_steps = (int32_t)c.max_steps;
        p.continuous = c.continuous;
        p.n_actions = c.n_actions;
        return p;
    }
    bool is_continuous() const { return continuous != 0; }
};

// Julia mod(x::Float64, y::Float64): floored modulo on top of the exact rem (= fmod)
__device__ __forceinline__ double jl_mod(double x, double y) {
    double r = ::fmod(x, y);
    if (r == 0) return ::copysign(r, y);
    if ((r > 0) != (y > 0)) return r + y;
    return r;
}

// the same for a Float32-VALUED x with |x| <= 2^16 and y = 2 pi (angle_normalize of a Float32 Pendulum): fmod(x, y) = x - q y
// with q = trunc(x / y) is exactly representable, and one fma delivers it exactly once q is right (x has 24 significant
// bits, q < 2^14, |x - q y| < 8 with its last bit at 2^-50 or above); q from a multiplication by RN(1 / y) can be off by one
// next to an integer quotient, which the sign / range of the remainder shows.  8 instructions instead of ocml's fmod loop;
// tools/micro/trig_f32arg.hip compares it with the host's fmod-based jl_mod for every Float32 in the range.
constexpr double TWO_PI_D = 2.0 * RLHIP_PI;
__device__ __forceinline__ double jl_mod_2pi_f32arg(float xf) {
    const double x = (double)xf;
    if (!(fabsf(xf) <= TRIG_MEDIUM_MAX)) return jl_mod(x, TWO_PI_D);
    double q = ::trunc(x * (1.0 / TWO_PI_D));
    double r = ::fma(-q, TWO_PI_D, x);
    if (x >= 0.0 ? r < 0.0 : r > 0.0) {  // q one too large in magnitude
        q -= (x >= 0.0 ? 1.0 : -1.0);
        r = ::fma(-q, TWO_PI_D, x);
    } else if (::fabs(r) >= TWO_PI_D) {  // q one too small in magnitude
        q += (x >= 0.0 ? 1.0 : -1.0);
        r = ::fma(-q, TWO_PI_D, x);
    }
    if (r == 0.0) return ::copysign(r, TWO_PI_D);
    if (r < 0.0) return r + TWO_PI_D;
    return r;
}

template <typename T>
struct AngleMod;
template <>
struct AngleMod<float> {
    static __device__ __forceinline__ double mod_2pi(float x) { return jl_mod_2pi_f32arg(x); }
};
template <>
struct AngleMod<double> {
    static __device__ __forceinline__ double mod_2pi(double x) { return jl_mod(x, 2.0 * RLHIP_PI); }
};

// reset!  :84-92
template <typename T>
__device__ __forceinline__ void env_reset1(const PendulumParams<T>&, LaneState<T>& e, uint64_t seed,
                                           uint32_t id) {
    T u[2];
    ResetDraw<T>::draw2(seed, id, e.episode, u);
    e.s[0] = (T)((2.0 * RLHIP_PI) * (double)(u[0] - (T)1));  // :85  2 * pi * (rand(T) - 1), Float64 product
    e.s[1] = (T)2 * (u[1] - (T)1);                           // :86
    e.t = 0;
    e.episode += 1;
}

// act! + torque + _step!  :94-122
template <typename T>
__device__ __forceinline__ void env_step1(const PendulumParams<T>& p, LaneState<T>& e, int32_t ai, T af,
                                          T& reward, bool& done) {
    T a;
    if (p.continuous) {
        a = af;  // :122
    } else {
        // :120-121  (4 / (n - 1)) * (a - (n - 1) / 2 - 1) in Float64 with the 1-based a; env.action::T rounds
        double a1 = (double)(ai + 1);
        double nm1 = (double)(p.n_actions - 1);
        a = (T)((4.0 / nm1) * (a1 - nm1 / 2.0 - 1.0));
    }
    e.t += 1;                                     // :101
    T th = e.s[0], thdot = e.s[1];                // :102
    a = clampT(a, -p.max_torque, p.max_torque);   // :103
    // :104  costs = angle_normalize(th)^2 + 0.1 * thdot^2 + 0.001 * a^2 (Float64);
    //       angle_normalize(x) = mod(x + pi, 2 * pi) - pi  (:71): x + pi is T, 2 * pi is Float64
    T thpi = th + (T)RLHIP_PI;
    double an = AngleMod<T>::mod_2pi(thpi) - RLHIP_PI;
    double costs = an * an + 0.1 * (double)(thdot * thdot) + 0.001 * (double)(a * a);
    // :105-110  pure T; sin(th + pi) literally
    T newthdot = thdot + ((T)-3 * p.g / ((T)2 * p.l) * Trig<T>::sin_(thpi) +
                          (T)3 * a / (p.m * (p.l * p.l))) *
                             p.dt;
    th = th + newthdot * p.dt;                                 // :111
    newthdot = clampT(newthdot, -p.max_speed, p.max_speed);    // :112
    e.s[0] = th;
    e.s[1] = newthdot;
    done = e.t >= p.max_steps;  // :115
    reward = (T)(-costs);       // :116
}

template <typename T>
__device__ __forceinline__ void env_obs1(const PendulumParams<T>&, const LaneState<T>& e, T o[4]) {
    Trig<T>::sincos_(e.s[0], &o[0], &o[1]);  // :70  [sin(th), cos(th), thdot]
    o[2] = e.s[1];
}

// ------------------------------------------------------------------------------ MountainCar --
template <typename T>
struct MountainCarParams {  // MountainCarEnvParams{T}  RLEnvs/MountainCarEnv.jl:3-12
    T min_pos, max_pos, max_speed, goal_pos, goal_velocity, power, gravity;
    int32_t max_steps;
    int32_t continuous;
    static constexpr int SDIM = 2;
    static constexpr int ODIM = 2;
    static constexpr int KIND = RLHIP_ENV_MOUNTAINCAR;
    typedef rlhip_mountaincar_cfg cfg_t;
    static MountainCarParams make(const rlhip_mountaincar_cfg& c) {
        MountainCarParams p;
        p.min_pos = (T)c.min_pos;
        p.max_pos = (T)c.max_pos;
        p.max_speed = (T)c.max_speed;
        p.goal_pos = (T)c.goal_pos;
        p.goal_velocity = (T)c.goal_velocity;
        p.power = (T)c.power;
        p.gravity = (T)c.gravity;
        p.max_steps = (int32_t)c.max_steps;
        p.continuous = c.continuous;
        return p;
    }
    bool is_continuous() const { return continuous != 0; }
};

// reset!  :99-105   x = 0.2 * rand(T) - 0.6 (Float64 arithmetic), v = 0
template <typename T>
__device__ __forceinline__ void env_reset1(const MountainCarParams<T>&, LaneState<T>& e, uint64_t seed,
                                           uint32_t id) {
    T u[2];
    ResetDraw<T>::draw2(seed, id, e.episode, u);
    e.s[0] = (T)(0.2 * (double)u[0] - 0.6);
    e.s[1] = (T)0;
    e.t = 0;
    e.episode += 1;
}

// act! + _step!  :107-135, reward :95
template <typename T>
__device__ __forceinline__ void env_step1(const MountainCarParams<T>& p, LaneState<T>& e, int32_t ai,
                                          T af, T& reward, bool& done) {
    T force = p.continuous ? af : (T)(ai - 1);  // :117  a - 2 with the 1-based a
    e.t += 1;                                   // :120
    T x = e.s[0], v = e.s[1];
    v = v + (force * p.power + Trig<T>::cos_((T)3 * x) * (-p.gravity));  // :122
    v = clampT(v, -p.max_speed, p.max_speed);                            // :123
    x = x + v;                                                           // :124
    x = clampT(x, p.min_pos, p.max_pos);                                 // :125
    if (x == p.min_pos && v < (T)0) v = (T)0;                            // :126-128
    done = (x >= p.goal_pos && v >= p.goal_velocity) || e.t >= p.max_steps;  // :129-131
    e.s[0] = x;
    e.s[1] = v;
    reward = done ? (T)0 : (T)-1;  // :95
}

template <typename T>
__device__ __forceinline__ void env_obs1(const MountainCarParams<T>&, const LaneState<T>& e, T o[4]) {
    o[0] = e.s[0];
    o[1] = e.s[1];
}

// ---------------------------------------------------------------------------------- Acrobot --
// RLEnvs/src/environments/3rd_party/AcrobotEnv.jl.  PARITY UNPINNED for this env (include/rlhip.h): the reference
// integrates act! with the un-vendored, adaptive OrdinaryDiffEq.solve(ode, RK4()) (:128-129); here one classic RK4
// step of length dt over the reference's own dsdt (:147-199), in Float64, the state stored as T.  A heavier per-lane
// integrator than the other three: 16 Float64 trig evaluations per env-step.
template <typename T>
struct AcrobotParams {
    double m1, m2, l1, lc1, lc2, I1, I2, g, dt;
    T max_vel_a, max_vel_b, noise;
    int32_t max_steps;
    int32_t nips;
    int32_t continuous;  // always 0 (Base.OneTo(3)); present for the generic kernels
    static constexpr int SDIM = 4;
    static constexpr int ODIM = 6;
    static constexpr int KIND = RLHIP_ENV_ACROBOT;
    typedef rlhip_acrobot_cfg cfg_t;
    static AcrobotParams make(const rlhip_acrobot_cfg& c) {
        AcrobotParams p;  // AcrobotEnvParams{T} stores every field as T (:42-56); dsdt reads them back (:149-156)
        p.m1 = (double)(T)c.link_mass_a;
        p.m2 = (double)(T)c.link_mass_b;
        p.l1 = (double)(T)c.link_length_a;
        p.lc1 = (double)(T)c.link_com_pos_a;
        p.lc2 = (double)(T)c.link_com_pos_b;
        p.I1 = (double)(T)c.link_moi;
        p.I2 = (double)(T)c.link_moi;
        p.g = (double)(T)c.g;
        p.dt = (double)(T)c.dt;
        p.max_vel_a = (T)c.max_vel_a;
        p.max_vel_b = (T)c.max_vel_b;
        p.noise = (T)c.max_torque_noise;
        p.max_steps = (int32_t)c.max_steps;
        p.nips = c.nips;
        p.continuous = 0;
        return p;
    }
    bool is_continuous() const { return false; }
};

// reset!  :94-101  state = T(0.1) * rand(rng, T, 4) .- T(0.05)  (reward = -1: EnvTraits below)
template <typename T>
__device__ __forceinline__ void env_reset1(const AcrobotParams<T>&, LaneState<T>& e, uint64_t seed, uint32_t id) {
    T u[4];
    ResetDraw<T>::draw4(seed, id, e.episode, u);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.s[k] = (T)0.1 * u[k] - (T)0.05;
    e.t = 0;
    e.episode += 1;
}

// uniform of the torque noise of the coming act! (:110-113): Philox(seed, env, t + 1, episode, ENVNOISE)
template <typename T>
__device__ __forceinline__ T acrobot_noise_u(const LaneState<T>& e, uint64_t seed, uint32_t id) {
    u32x4 w = philox4x32_10(seed, id, (uint32_t)(e.t + 1), e.episode, TAG_ENVNOISE);
    if constexpr (sizeof(T) == 8) return (T)u01_f64(w.x, w.y);
    else return (T)u01_f32(w.x);
}

// dsdt  :147-199
template <typename T>
__device__ __forceinline__ void acrobot_dsdt(const AcrobotParams<T>& p, const double s[4], double a, double du[4]) {
    const double m1 = p.m1, m2 = p.m2, l1 = p.l1, lc1 = p.lc1, lc2 = p.lc2, I1 = p.I1, I2 = p.I2, g = p.g;
    const double theta1 = s[0], theta2 = s[1], dtheta1 = s[2], dtheta2 = s[3];
    double s2, c2;
    ::sincos(theta2, &s2, &c2);
    const double d1 = ((m1 * (lc1 * lc1) + m2 * ((l1 * l1 + lc2 * lc2) + ((2 * l1) * lc2) * c2)) + I1) + I2;  // :171
    const double d2 = m2 * (lc2 * lc2 + (l1 * lc2) * c2) + I2;                                               // :172
    const double phi2 = ((m2 * lc2) * g) * ::cos((theta1 + theta2) - RLHIP_PI / 2.0);                        // :173
    double phi1 = (((((-m2) * l1) * lc2) * (dtheta2 * dtheta2)) * s2 -
                   (((((2 * m2) * l1) * lc2) * dtheta2) * dtheta1) * s2) +
                  ((m1 * lc1 + m2 * l1) * g) * ::cos(theta1 - RLHIP_PI / 2);                                 // :174-179
    phi1 = phi1 + phi2;
    double ddtheta1 = 0.0, ddtheta2;
    if (p.nips) {
        ddtheta2 = ((a + (d2 / d1) * phi1) - phi2) / ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1);  // :183
    } else {
        ddtheta2 = (((a + (d2 / d1) * phi1) - (((m2 * l1) * lc2) * (dtheta1 * dtheta1)) * s2) - phi2) /
                   ((m2 * (lc2 * lc2) + I2) - (d2 * d2) / d1);  // :187-190
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;                // :191
    }
    du[0] = dtheta1;
    du[1] = dtheta2;
    du[2] = ddtheta1;
    du[3] = ddtheta2;
}

__device__ __forceinline__ double acrobot_wrap(double x, double m, double M) {  // :204-222
    const double diff = M - m;
    while (x > M) x = x - diff;
    while (x < m) x = x + diff;
    return x;
}

// act!  :104-145.  ai: 0-based action (torque ai - 1); af: the noise uniform (only read when p.noise > 0)
template <typename T>
__device__ __forceinline__ void env_step1(const AcrobotParams<T>& p, LaneState<T>& e, int32_t ai, T af, T& reward,
                                          bool& done) {
    e.t += 1;               // :106
    T torque = (T)(ai - 1);  // :107
    if (p.noise > (T)0) torque = (torque + (T)(2.0 * (double)p.noise) * af) - p.noise;  // :110-113
    double y[4], k1[4], k2[4], k3[4], k4[4], yt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = (double)e.s[k];
    const double a = (double)torque, h = p.dt, h2 = h / 2.0;
    acrobot_dsdt(p, y, a, k1);
#pragma unroll
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h2 * k1[k];
    acrobot_dsdt(p, yt, a, k2);
#pragma unroll
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h2 * k2[k];
    acrobot_dsdt(p, yt, a, k3);
#pragma unroll
    for (int k = 0; k < 4; ++k) yt[k] = y[k] + h * k3[k];
    acrobot_dsdt(p, yt, a, k4);
    double ns[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ns[k] = y[k] + (h / 6.0) * (((k1[k] + 2 * k2[k]) + 2 * k3[k]) + k4[k]);
    ns[0] = acrobot_wrap(ns[0], -RLHIP_PI, RLHIP_PI);  // :135-136
    ns[1] = acrobot_wrap(ns[1], -RLHIP_PI, RLHIP_PI);
    const double va = (double)p.max_vel_a, vb = (double)p.max_vel_b;
    ns[2] = ::fmin(::fmax(ns[2], -va), va);  // :137-138
    ns[3] = ::fmin(::fmax(ns[3], -vb), vb);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.s[k] = (T)ns[k];
    const bool succeeded = (-::cos((double)e.s[0]) - ::cos((double)e.s[1] + (double)e.s[0])) > 1.0;  // :141
    done = succeeded || e.t > p.max_steps;                                                         // :142
    reward = succeeded ? (T)0 : (T)-1;                                                             // :143
}

template <typename T>
__device__ __forceinline__ void env_obs1(const AcrobotParams<T>&, const LaneState<T>& e, T o[6]) {
    Trig<T>::sincos_(e.s[0], &o[1], &o[0]);  // :73  [cos(s1), sin(s1), cos(s2), sin(s2), s3, s4]
    Trig<T>::sincos_(e.s[1], &o[3], &o[2]);
    o[4] = e.s[2];
    o[5] = e.s[3];
}

// per-env traits the generic kernels need: reward(env) right after reset!, and whether act! draws from the RNG
template <class P>
struct EnvTraits {
    static constexpr bool STEP_NOISE = false;
    static constexpr int RESET_REWARD = 0;
};
template <typename T>
struct EnvTraits<AcrobotParams<T>> {
    static constexpr bool STEP_NOISE = true;
    static constexpr int RESET_REWARD = -1;  // AcrobotEnv.jl:99
};

// device-pointer view of rlhip_env_state
template <typename T>
struct EnvArrays {
    T* s[4];
    int32_t* t;
    uint8_t* done;
    T* reward;
    uint32_t* episode;
    static EnvArrays from(const rlhip_env_state& st) {
        EnvArrays a;
        for (int k = 0; k < 4; ++k) a.s[k] = (T*)st.s[k];
        a.t = st.t;
        a.done = st.done;
        a.reward = (T*)st.reward;
        a.episode = st.episode;
        return a;
    }
};

}  // namespace rlhip
