"""Env oracle: the reference's own env tests are interface-only (RLBase.test_interfaces! / test_runnable!,
RLBase/src/base.jl:77-227; RLEnvs/test/environments/examples/{cart_pole,pendulum,mountain_car}.jl) -- restated
here for the oracle, plus hand-derived single-step values and BASELINE config 1 (RandomPolicy x CartPoleEnv,
1000 steps) as the N = 1 plumbing check."""
import math

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("kind,continuous", [("cartpole", False), ("cartpole", True), ("pendulum", True),
                                             ("pendulum", False), ("mountaincar", False), ("mountaincar", True)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_interface_contract(kind, continuous, dtype):
    n = 16
    a = oracle.VecEnv(kind, n, seed=3, dtype=dtype, continuous=continuous)
    b = oracle.VecEnv(kind, n, seed=3, dtype=dtype, continuous=continuous)  # copy determinism under equal seeds
    rng = np.random.default_rng(0)
    c = a.cfg
    for step in range(1000):
        o = a.obs()
        if kind == "cartpole":
            th = c.thetathreshold_deg * math.pi / 180
            assert (np.abs(o[0]) <= 2 * c.xthreshold).all() and (np.abs(o[2]) <= 2 * th).all()
        elif kind == "pendulum":
            assert (np.abs(o[0]) <= 1).all() and (np.abs(o[1]) <= 1).all() and (np.abs(o[2]) <= c.max_speed).all()
        else:
            assert (o[0] >= dtype(c.min_pos)).all() and (o[0] <= dtype(c.max_pos)).all()
            assert (np.abs(o[1]) <= dtype(c.max_speed)).all()
        if continuous:
            lim = 2.0 if kind == "pendulum" else 1.0
            act = rng.uniform(-lim, lim, n).astype(dtype)
        else:
            act = rng.integers(0, 2 if kind == "cartpole" else 3, n).astype(np.int32)
        a.step(act)
        b.step(act)
        assert all(np.array_equal(x, y) for x, y in zip(a.s, b.s))
        assert np.array_equal(a.done, b.done) and np.array_equal(a.reward, b.reward)
    assert a.episode.min() >= 1 + 1000 // 202  # auto-reset happened


def test_cartpole_first_step_by_hand():
    """One explicit-Euler step evaluated by hand in Float64 (CartPoleEnv.jl:118-140)."""
    env = oracle.VecEnv("cartpole", 1, seed=0, dtype=np.float64, auto_reset=False)
    x, xd, th, thd = 0.01, -0.02, 0.03, 0.04
    env.set_state([[x], [xd], [th], [thd]], [0])
    env.step(np.array([1], np.int32))  # Julia action 2 -> force +10
    force, g, mp, M, l, pml, dt = 10.0, 9.8, 0.1, 1.1, 0.5, 0.05, 0.02
    tmp = (force + pml * thd ** 2 * math.sin(th)) / M
    thacc = (g * math.sin(th) - math.cos(th) * tmp) / (l * (4 / 3 - mp * math.cos(th) ** 2 / M))
    xacc = tmp - pml * thacc * math.cos(th) / M
    exp = [x + dt * xd, xd + dt * xacc, th + dt * thd, thd + dt * thacc]
    got = [env.s[k][0] for k in range(4)]
    np.testing.assert_allclose(got, exp, rtol=1e-14)
    assert env.reward[0] == 1.0 and env.done[0] == 0 and env.t[0] == 1


def test_cartpole_float32_promotion_quirk():
    """With T = Float32 thetaacc / xacc are Float64 (the `4 / 3` literal): the velocity updates equal the
    Float64 formula evaluated on the Float32 inputs and rounded once -- not an all-Float32 evaluation."""
    env = oracle.VecEnv("cartpole", 1, seed=0, dtype=np.float32, auto_reset=False)
    s = np.float32([0.013, -0.021, 0.034, 0.047])
    env.set_state([[v] for v in s], [0])
    env.step(np.array([0], np.int32))
    f = np.float32
    force = f(-10.0)
    c, si = f(math.cos(float(s[2]))), f(math.sin(float(s[2])))  # correctly rounded Float32 trig
    tmp = f(f(force + f(f(f(0.05) * f(s[3] * s[3])) * si)) / f(1.1))
    num = f(f(f(9.8) * si) - f(c * tmp))
    frac = f(f(f(0.1) * f(c * c)) / f(1.1))
    thacc = float(num) / (0.5 * (4.0 / 3.0 - float(frac)))
    xacc = float(tmp) - float(f(0.05)) * thacc * float(c) / float(f(1.1))
    exp1 = f(float(s[1]) + float(f(0.02)) * xacc)
    exp3 = f(float(s[3]) + float(f(0.02)) * thacc)
    assert abs(env.s[1][0] - exp1) <= abs(np.spacing(exp1)) and abs(env.s[3][0] - exp3) <= abs(np.spacing(exp3))
    assert env.s[0][0] == f(s[0] + f(f(0.02) * s[1])) and env.s[2][0] == f(s[2] + f(f(0.02) * s[3]))


def _cartpole_step_restated(s, a, dtype):
    """CartPoleEnv.jl:118-140 over arrays, independent of oracle/rlo_envs_impl.h: numpy scalars carry Julia's
    promotion rules (T op T -> T; T op Float64 -> Float64 through the `4 / 3` literal)."""
    f = dtype
    x, xd, th, thd = (v.astype(f) for v in s)
    force = np.where(a == 1, f(10.0), f(-10.0)).astype(f)
    c, si = np.cos(th.astype(np.float64)).astype(f), np.sin(th.astype(np.float64)).astype(f)
    tmp = ((force + f(0.1 * 0.5) * thd ** 2 * si) / f(1.1)).astype(f)
    num = (f(9.8) * si - c * tmp).astype(f)
    frac = (f(0.1) * c ** 2 / f(1.1)).astype(f)
    thacc = num.astype(np.float64) / (np.float64(f(0.5)) * (4 / 3 - frac.astype(np.float64)))
    xacc = tmp.astype(np.float64) - np.float64(f(0.05)) * thacc * c.astype(np.float64) / np.float64(f(1.1))
    dt = f(0.02)
    return [(x + dt * xd).astype(f), (xd.astype(np.float64) + np.float64(dt) * xacc).astype(f), (th + dt * thd).astype(f),
            (thd.astype(np.float64) + np.float64(dt) * thacc).astype(f)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cartpole_reference_test_config_wide_theta(dtype):
    """`CartPoleEnv(; T = Float32, thetathreshold = 90.0)` -- the configuration the reference's own test runs
    (RLEnvs/test/environments/examples/cart_pole.jl:7): states over |theta| in (pi/4, pi), threshold at pi/2."""
    n = 4000
    rng = np.random.default_rng(0)
    env = oracle.VecEnv("cartpole", n, seed=3, dtype=dtype, auto_reset=False, thetathreshold=90.0)
    th = rng.uniform(math.pi / 4, math.pi, n) * rng.choice([-1.0, 1.0], n)
    s = [rng.uniform(-2.3, 2.3, n), rng.uniform(-3, 3, n), th, rng.uniform(-6, 6, n)]
    s = [v.astype(dtype) for v in s]
    a = rng.integers(0, 2, n).astype(np.int32)
    env.set_state(s, np.zeros(n, np.int32))
    env.step(a)
    exp = _cartpole_step_restated(s, a, dtype)
    # positions are one exact T operation each; velocities go through sinf / cosf (<= 1 ulp from numpy's rounding
    # of the Float64 value)
    assert np.array_equal(env.s[0], exp[0]) and np.array_equal(env.s[2], exp[2])
    tol = dict(rtol=2e-6, atol=1e-6) if dtype == np.float32 else dict(rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(env.s[1], exp[1], **tol)
    np.testing.assert_allclose(env.s[3], exp[3], **tol)
    thr = dtype(90.0 * math.pi / 180) if dtype == np.float64 else np.float32(90.0 * math.pi / 180)
    done = (np.abs(exp[0]) > dtype(2.4)) | (np.abs(exp[2]) > thr)
    assert np.array_equal(env.done.astype(bool), done)
    assert np.array_equal(env.reward, np.where(done, 0, 1).astype(dtype))
    assert 0.2 < done.mean() < 0.9  # both sides of the 90 degree threshold are populated


def test_pendulum_and_mountaincar_rules():
    p = oracle.VecEnv("pendulum", 1, seed=0, dtype=np.float64, auto_reset=False, continuous=True)
    th, thd, a = 0.5, -0.3, 5.0  # action beyond max_torque is clamped inside _step!
    p.set_state([[th], [thd]], [0])
    p.step(np.array([a]))
    ac = 2.0
    cost = ((th + math.pi) % (2 * math.pi) - math.pi) ** 2 + 0.1 * thd ** 2 + 0.001 * ac ** 2
    nthd = thd + (-3 * 10 / 2 * math.sin(th + math.pi) + 3 * ac) * 0.05
    np.testing.assert_allclose([p.s[0][0], p.s[1][0], p.reward[0]], [th + nthd * 0.05, nthd, -cost], rtol=1e-13)
    d = oracle.VecEnv("pendulum", 1, seed=0, dtype=np.float64, continuous=False)  # discrete torques {-2, 0, 2}
    for a0, tq in ((0, -2.0), (1, 0.0), (2, 2.0)):
        d.set_state([[0.1], [0.0]], [0])
        d.step(np.array([a0], np.int32))
        assert d.s[1][0] == pytest.approx((-15 * math.sin(0.1 + math.pi) + 3 * tq) * 0.05, rel=1e-12)
    m = oracle.VecEnv("mountaincar", 1, seed=0, dtype=np.float64, auto_reset=False)
    m.set_state([[-1.2], [-0.01]], [0])
    m.step(np.array([0], np.int32))  # pushes left at the left wall: x clamps to min_pos and v is zeroed
    assert m.s[0][0] == -1.2 and m.s[1][0] == 0.0 and m.reward[0] == -1.0
    m.set_state([[0.49], [0.07]], [0])
    m.step(np.array([2], np.int32))
    assert m.done[0] == 1 and m.reward[0] == 0.0  # reaching the goal terminates with reward 0


def test_reset_ranges_and_stream_disjointness():
    n = 4000
    c = oracle.VecEnv("cartpole", n, seed=9)
    assert all(np.abs(c.s[k]).max() <= 0.05 for k in range(4)) and np.abs(c.s[0]).max() > 0.04
    p = oracle.VecEnv("pendulum", n, seed=9)
    assert p.s[0].min() >= -2 * math.pi - 1e-6 and p.s[0].max() < 0 and p.s[1].min() >= -2 and p.s[1].max() < 0
    m = oracle.VecEnv("mountaincar", n, seed=9)
    assert m.s[0].min() >= -0.6 - 1e-7 and m.s[0].max() <= -0.4 and (m.s[1] == 0).all()
    # shards: the same global env ids give the same states whatever the shard layout; different ids differ
    whole = oracle.VecEnv("cartpole", 8, seed=5, env_id_base=0)
    lo, hi = oracle.VecEnv("cartpole", 4, seed=5, env_id_base=0), oracle.VecEnv("cartpole", 4, seed=5, env_id_base=4)
    for k in range(4):
        assert np.array_equal(whole.s[k], np.concatenate([lo.s[k], hi.s[k]]))
    assert not np.array_equal(lo.s[0], hi.s[0])


def test_config1_random_policy_cartpole_1000_steps():
    """BASELINE configs[0]: RandomPolicy on CartPoleEnv, StopAfterNSteps(1_000), single env (README example).
    The N = 1 vector env with auto-reset must reproduce the scalar run loop's episode structure."""
    seed = 123
    vec = oracle.VecEnv("cartpole", 1, seed=seed, dtype=np.float64)
    scalar = oracle.VecEnv("cartpole", 1, seed=seed, dtype=np.float64, auto_reset=False)
    lens_vec, lens_scalar, cur_v, cur_s = [], [], 0, 0
    total_reward = 0.0
    for step in range(1, 1001):
        w = oracle.philox(seed, 0, 0, step, oracle.TAG["EXPLORE"])
        a = np.array([oracle.lib().rlo_randint(w[2], 2)], np.int32)  # rand(rng, 1:2)
        vec.step(a)
        cur_v += 1
        total_reward += vec.reward[0]
        if vec.done[0]:
            lens_vec.append(cur_v)
            cur_v = 0
        # scalar protocol (core/run.jl:44-72): reset!(env) at the top of the next episode
        if scalar.done[0]:
            scalar.reset()
        scalar.step(a)
        cur_s += 1
        if scalar.done[0]:
            lens_scalar.append(cur_s)
            cur_s = 0
    assert lens_vec == lens_scalar and len(lens_vec) > 20
    assert total_reward == 1000 - len(lens_vec)  # every terminal step pays 0
    assert 10 < np.mean(lens_vec) < 40  # random policy on CartPole
