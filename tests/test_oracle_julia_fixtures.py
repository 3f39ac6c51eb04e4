"""E1-E3 physics against the REAL reference environments -- once somebody with Julia has generated the fixture.

tests/golden/gen_julia_fixtures.jl runs CartPoleEnv / PendulumEnv / MountainCarEnv of the reference (T = Float32 / Float64,
default and test-suite configurations, discrete and continuous actions) on teacher-forced states and writes
tests/golden/julia_env_steps.json.  This image has no `julia` binary, so the fixture is not committed and the test SKIPS;
with the file present it turns the oracle's env rows ("parity unpinned") green or red:

  Float64 cases  bit for bit (both sides IEEE double arithmetic in the source's operation order; sin / cos of glibc and of
                 Julia's openlibm-derived kernels are both correctly rounded to < 1 ulp -- a last-bit difference there would
                 show up here and is then to be judged, not hidden),
  Float32 cases  next state within 1 ulp per component (Julia evaluates sin / cos of a Float32 in double precision and rounds
                 once; the oracle calls sinf / cosf), reward / terminal flag / step counter exactly.

Also checked here WITHOUT Julia: the generator script names exactly the fields this loader reads, so the two cannot drift.
"""
import json
import os
import re

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "julia_env_steps.json")
GENERATOR = os.path.join(HERE, "golden", "gen_julia_fixtures.jl")
STEP_FIELDS = ("env", "T", "cfg", "state_bits", "t", "action", "action_bits", "next_state_bits", "reward_bits", "terminated",
               "t_out", "obs_bits")
RESET_FIELDS = ("env", "T", "uniforms", "state_bits", "t")


def test_generator_and_loader_agree_on_the_fixture_schema():
    src = open(GENERATOR).read()
    for f in STEP_FIELDS + RESET_FIELDS:
        assert f'"{f}"' in src.replace('\\"', '"') or f'"""{f}' in src, f
    for name in ("CartPoleEnv(; T = T", "PendulumEnv(; T = T", "MountainCarEnv(; T = T", "act!(env, a)", "reset!(env)"):
        assert name in src
    assert re.search(r'open\(joinpath\(@__DIR__, "julia_env_steps.json"\)', src)


def _from_bits(bits, T):
    u = np.array(bits, dtype=np.uint32 if T == "Float32" else np.uint64)
    return u.view(np.float32 if T == "Float32" else np.float64)


def _ulps(a, b):
    ia = a.view(np.int32 if a.dtype == np.float32 else np.int64).astype(np.int64)
    ib = b.view(np.int32 if b.dtype == np.float32 else np.int64).astype(np.int64)
    ia = np.where(ia < 0, np.iinfo(np.int64).min - ia if a.dtype == np.float64 else -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, np.iinfo(np.int64).min - ib if b.dtype == np.float64 else -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/julia_env_steps.json absent: generate it with "
                    "`julia tests/golden/gen_julia_fixtures.jl` (needs ReinforcementLearningEnvironments; no julia in this image)")
def test_oracle_env_steps_match_the_real_reference_environments():
    fx = json.load(open(FIXTURE))
    assert fx["generator"] == "tests/golden/gen_julia_fixtures.jl" and len(fx["steps"]) >= 1000
    worst = {}
    for c in fx["steps"]:
        T, kind = c["T"], c["env"]
        dt = np.float32 if T == "Float32" else np.float64
        cfg = dict(c["cfg"])
        s = _from_bits(c["state_bits"], T)
        env = oracle.VecEnv(kind, 1, seed=0, dtype=dt, auto_reset=False, **cfg)
        env.set_state([np.array([v], dt) for v in s], t=c["t"])
        env.done[:] = 0
        continuous = bool(cfg.get("continuous", False)) and kind != "cartpole"
        a = _from_bits([c["action_bits"]], T) if continuous else np.array([c["action"] - 1], np.int32)  # Julia is 1-based
        env.step(a)
        got = np.array([env.s[k][0] for k in range(env.sdim)], dt)
        want = _from_bits(c["next_state_bits"], T)
        u = int(_ulps(got, want).max())
        worst[(kind, T)] = max(worst.get((kind, T), 0), u)
        assert u <= (0 if T == "Float64" else 1), (c, got, want)
        assert bool(env.done[0]) == c["terminated"], c
        assert int(env.t[0]) == c["t_out"], c
        assert np.array_equal(np.array([env.reward[0]], dt).view(np.uint32 if T == "Float32" else np.uint64),
                              np.array([c["reward_bits"]], np.uint32 if T == "Float32" else np.uint64)), c
        ob = env.obs()[:, 0]
        assert int(_ulps(ob.astype(dt), _from_bits(c["obs_bits"], T)).max()) <= (0 if T == "Float64" else 1), c
    print("worst ulp distance per (env, T):", worst)


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/julia_env_steps.json absent")
def test_reset_formulas_match_the_real_reference_environments():
    """reset!(env) as a function of the uniforms it drew (the streams themselves are replaced by Philox: DESIGN.md section 4):
    CartPole 0.1 u - 0.05 (x 4), Pendulum theta = 2 pi (u - 1), thetadot = 2 (u - 1), MountainCar x = 0.2 u - 0.6, v = 0"""
    fx = json.load(open(FIXTURE))
    for c in fx["resets"]:
        T = np.float32 if c["T"] == "Float32" else np.float64
        u = np.array(c["uniforms"], np.float64).astype(T)
        want = _from_bits(c["state_bits"], c["T"])
        if c["env"] == "cartpole":
            got = T(0.1) * u[:4] - T(0.05)
        elif c["env"] == "pendulum":
            got = np.array([T(2 * np.pi) * (u[0] - T(1)), T(2) * (u[1] - T(1))], T)
        else:
            got = np.array([T(0.2) * u[0] - T(0.6), T(0)], T)
        assert int(_ulps(got.astype(T), want).max()) <= (0 if c["T"] == "Float64" else 1), (c, got)
        assert c["t"] == 0
