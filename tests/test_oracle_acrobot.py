"""AcrobotEnv oracle.  The reference's own test is interface-only
(RLEnvs/test/environments/3rd_party/acrobot_env.jl: test_interfaces! + test_runnable!) and its integrator is the
un-vendored adaptive OrdinaryDiffEq RK4, so parity is UNPINNED for this env; what is pinned here: the interface
contract, the dynamics against an independent restatement of the equations, and the size of the gap between the
single classic RK4 step and a converged solution of the same ODE (what an adaptive driver approaches)."""
import math

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_interface_contract(dtype):
    n = 32
    a = oracle.VecEnv("acrobot", n, seed=5, dtype=dtype)
    b = oracle.VecEnv("acrobot", n, seed=5, dtype=dtype)
    assert a.odim == 6 and a.sdim == 4
    assert (a.reward == -1).all() and not a.done.any()          # reset!: reward = -1, done = false  (:94-101)
    assert all((np.abs(s) <= 0.05 + 1e-7).all() for s in a.s)   # state = 0.1 * rand(4) - 0.05
    rng = np.random.default_rng(0)
    c = a.cfg
    finished = 0
    for step in range(450):
        o = a.obs()
        assert (np.abs(o[:4]) <= 1).all()                        # state_space  :77-86
        assert (np.abs(o[4]) <= dtype(c.max_vel_a)).all() and (np.abs(o[5]) <= dtype(c.max_vel_b)).all()
        np.testing.assert_allclose(o[0] ** 2 + o[1] ** 2, 1, atol=1e-6)
        t_before = a.t.copy()
        act = rng.integers(0, 3, n).astype(np.int32)
        a.step(act)
        b.step(act)
        assert all(np.array_equal(x, y) for x, y in zip(a.s, b.s)) and np.array_equal(a.reward, b.reward)
        assert np.isin(a.reward, [-1, 0]).all()
        assert ((a.reward == 0) <= (a.done == 1)).all()          # success terminates
        timeout = (t_before + 1 > c.max_steps)                   # done = succeeded || t > max_steps  (:142)
        assert (a.done[timeout] == 1).all()
        finished += int(a.done.sum())
    assert finished >= 2 * n and a.episode.min() >= 3            # auto-reset: 201-step episodes at most


def _dsdt(y, a, book=True):
    """the gym / book equations written independently of the oracle (sin form of the gravity terms)"""
    m1 = m2 = l1 = 1.0
    lc1 = lc2 = 0.5
    I1 = I2 = 1.0
    g = 9.8
    t1, t2, d1_, d2_ = y
    d1 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * math.cos(t2)) + I1 + I2
    d2 = m2 * (lc2 ** 2 + l1 * lc2 * math.cos(t2)) + I2
    phi2 = m2 * lc2 * g * math.sin(t1 + t2)
    phi1 = -m2 * l1 * lc2 * d2_ ** 2 * math.sin(t2) - 2 * m2 * l1 * lc2 * d2_ * d1_ * math.sin(t2) \
        + (m1 * lc1 + m2 * l1) * g * math.sin(t1) + phi2
    if book:
        dd2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * d1_ ** 2 * math.sin(t2) - phi2) / (m2 * lc2 ** 2 + I2 - d2 ** 2 / d1)
        dd1 = -(d2 * dd2 + phi1) / d1
    else:
        dd2 = (a + d2 / d1 * phi1 - phi2) / (m2 * lc2 ** 2 + I2 - d2 ** 2 / d1)
        dd1 = 0.0   # the reference's "nips" branch leaves ddtheta1 = 0.0  (:163,183)
    return np.array([d1_, d2_, dd1, dd2])


def _rk4(y, a, h, book=True):
    k1 = _dsdt(y, a, book)
    k2 = _dsdt(y + h / 2 * k1, a, book)
    k3 = _dsdt(y + h / 2 * k2, a, book)
    k4 = _dsdt(y + h * k3, a, book)
    return y + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def _wrap(x):
    while x > math.pi:
        x -= 2 * math.pi
    while x < -math.pi:
        x += 2 * math.pi
    return x


@pytest.mark.parametrize("book", [True, False])
def test_one_step_matches_an_independent_restatement(book):
    n = 64
    env = oracle.VecEnv("acrobot", n, seed=1, dtype=np.float64, auto_reset=False, nips=0 if book else 1)
    rng = np.random.default_rng(3)
    s0 = np.stack([rng.uniform(-math.pi, math.pi, n), rng.uniform(-math.pi, math.pi, n),
                   rng.uniform(-4 * math.pi, 4 * math.pi, n), rng.uniform(-9 * math.pi, 9 * math.pi, n)])
    env.set_state(s0)
    act = rng.integers(0, 3, n).astype(np.int32)
    env.step(act)
    for i in range(n):
        y = _rk4(s0[:, i].copy(), float(act[i] - 1), 0.2, book)
        y[0], y[1] = _wrap(y[0]), _wrap(y[1])
        y[2] = min(max(y[2], -4 * math.pi), 4 * math.pi)
        y[3] = min(max(y[3], -9 * math.pi), 9 * math.pi)
        got = np.array([env.s[k][i] for k in range(4)])
        np.testing.assert_allclose(got, y, rtol=0, atol=2e-12)
        ok = -math.cos(y[0]) - math.cos(y[1] + y[0]) > 1.0
        assert env.done[i] == int(ok) and env.reward[i] == (0.0 if ok else -1.0)


def test_gap_to_a_converged_solution_of_the_same_ode():
    """what 'parity unpinned' costs: one RK4 step of 0.2 s against a converged integration (what the reference's
    adaptive driver approaches at its reltol = 1e-3) from the states a random policy visits"""
    from scipy.integrate import solve_ivp

    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(40):
        y0 = np.array([rng.uniform(-math.pi, math.pi), rng.uniform(-math.pi, math.pi), rng.uniform(-3, 3),
                       rng.uniform(-5, 5)])
        a = float(rng.integers(0, 3) - 1)
        ref = solve_ivp(lambda t, y: _dsdt(y, a), (0.0, 0.2), y0, rtol=1e-11, atol=1e-12).y[:, -1]
        worst = max(worst, float(np.abs(_rk4(y0, a, 0.2) - ref).max()))
    assert worst < 0.1, worst   # measured 0.028 (on velocities of a few rad/s): the same order as reltol = 1e-3


def test_torque_noise_is_a_per_step_per_env_draw():
    n = 8
    quiet = oracle.VecEnv("acrobot", n, seed=2, dtype=np.float64)
    noisy = oracle.VecEnv("acrobot", n, seed=2, dtype=np.float64, max_torque_noise=0.5)
    again = oracle.VecEnv("acrobot", n, seed=2, dtype=np.float64, max_torque_noise=0.5)
    act = np.ones(n, np.int32)
    for _ in range(5):
        for e in (quiet, noisy, again):
            e.step(act)
    assert all(np.array_equal(x, y) for x, y in zip(noisy.s, again.s))
    assert not np.array_equal(noisy.s[3], quiet.s[3])
    same_start = oracle.VecEnv("acrobot", n, seed=2, dtype=np.float64, max_torque_noise=0.5, auto_reset=False)
    same_start.set_state([np.zeros(n)] * 4)
    same_start.step(act)
    assert len(np.unique(same_start.s[3])) == n      # one draw per env
