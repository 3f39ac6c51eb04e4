"""GPU parity tests proper: every HIP kernel, called through the C ABI (rlhip._lib -> librlhip.so),
against the CPU oracle on the same seeded inputs.  Bit-exact for integer / index / flag outputs and
for Float arithmetic that has a fixed operation order; stated tolerances elsewhere."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402  (the checker)

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def rl():
    import rlhip
    from rlhip import _lib

    n = _lib.i32(0)
    _lib.call("rlhip_device_count", _lib.C.byref(n))
    assert n.value >= 1
    return rlhip


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def host(t):
    return t.detach().cpu().numpy()


def load(name):
    with open(os.path.join(G, name + ".json")) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------ RNG
def test_philox_streams_bit_exact(rl):
    from rlhip import ops

    for n in (1, 3, 4, 1000, 4099):
        g = host(ops.fill_uniform(n, seed=0x123456789ABCDEF, t=7, tag=oracle.TAG["SYNTH"]))
        o = oracle.fill_uniform_f32(n, 0x123456789ABCDEF, 7, oracle.TAG["SYNTH"])
        assert np.array_equal(g, o)


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 4096 * 32, 100003])
def test_permutation_is_bijection_and_matches_oracle(rl, n):
    from rlhip import ops

    p = host(ops.permutation(n, seed=99, epoch=3)).astype(np.int64)
    assert np.array_equal(np.sort(p), np.arange(n))
    m = min(n, 2000)
    o = np.array([oracle.permute(99, 3, n, i) for i in range(m)])
    assert np.array_equal(p[:m], o)
    if n > 64:
        p2 = host(ops.permutation(n, seed=99, epoch=4)).astype(np.int64)
        assert not np.array_equal(p, p2)


# ----------------------------------------------------------------------------------------- envs
ENV_CASES = [
    ("cartpole", False, torch.float32), ("cartpole", True, torch.float32), ("cartpole", False, torch.float64),
    ("pendulum", True, torch.float32), ("pendulum", False, torch.float32), ("pendulum", True, torch.float64),
    ("mountaincar", False, torch.float32), ("mountaincar", True, torch.float32),
    ("mountaincar", False, torch.float64),
    ("acrobot", False, torch.float32), ("acrobot", False, torch.float64),
]


def _rand_actions(kind, continuous, n, rng, dtype):
    if continuous:
        lim = 2.0 if kind == "pendulum" else 1.0
        return (rng.uniform(-lim * 1.2 if kind == "pendulum" else -lim, lim * 1.2 if kind == "pendulum" else lim,
                            n)).astype(dtype)
    na = 2 if kind == "cartpole" else 3
    return rng.integers(0, na, n).astype(np.int32)


@pytest.mark.parametrize("kind,continuous,T", ENV_CASES)
def test_env_reset_matches_oracle_bit_exact(rl, kind, continuous, T):
    n = 1000  # not a multiple of 4 -> also exercises the scalar tail path of the step kernel later
    npdt = np.float32 if T == torch.float32 else np.float64
    env = rl.HipVecEnv(kind, n, T=T, continuous=continuous, seed=123, env_id_base=77)
    ref = oracle.VecEnv(kind, n, seed=123, env_id_base=77, dtype=npdt, continuous=continuous)
    for k in range(env.sdim):
        assert np.array_equal(host(env.raw_state()[k]), ref.s[k]), f"state component {k}"
    assert np.array_equal(host(env._episode), ref.episode.view(np.int32))
    # state(env) observation
    np.testing.assert_allclose(host(env.state()), ref.obs(), rtol=0, atol=1e-7 if npdt == np.float32 else 1e-15)
    # masked reset (reset only "terminated" ones)
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    env._done.copy_(dev(mask))
    env.reset_(is_force=False)
    ref.reset(mask)
    for k in range(env.sdim):
        assert np.array_equal(host(env.raw_state()[k]), ref.s[k])
    assert np.array_equal(host(env._episode), ref.episode.view(np.int32))


@pytest.mark.parametrize("kind,continuous,T", ENV_CASES)
@pytest.mark.parametrize("n", [4096, 1001])
def test_env_step_teacher_forced(rl, kind, continuous, T, n):
    """Same input state + action into kernel and oracle at every step: flags / counters
    bit-exact, states within 1e-6 rel (Float32) / 1e-12 (Float64); resets bit-exact (shared Philox)."""
    npdt = np.float32 if T == torch.float32 else np.float64
    env = rl.HipVecEnv(kind, n, T=T, continuous=continuous, seed=5, env_id_base=3, max_steps=50)
    ref = oracle.VecEnv(kind, n, seed=5, env_id_base=3, dtype=npdt, continuous=continuous, max_steps=50)
    rng = np.random.default_rng(0)
    rtol = 2e-6 if npdt == np.float32 else 1e-12
    n_bit_mismatch = 0
    total = 0
    for step in range(120):
        # teacher forcing: the oracle restarts from the kernel's current state
        ref.set_state([host(env.raw_state()[k]) for k in range(env.sdim)], host(env._t))
        ref.episode[:] = host(env._episode).view(np.uint32)
        a = _rand_actions(kind, continuous, n, rng, npdt)
        env.act0_(dev(a))
        ref.step(a)
        assert np.array_equal(host(env._done), ref.done), f"done flags differ at step {step}"
        assert np.array_equal(host(env._t), ref.t), f"t differs at step {step}"
        assert np.array_equal(host(env._episode).view(np.uint32), ref.episode)
        np.testing.assert_allclose(host(env.reward()), ref.reward, rtol=rtol, atol=1e-7)
        for k in range(env.sdim):
            g, o = host(env.raw_state()[k]), ref.s[k]
            np.testing.assert_allclose(g, o, rtol=rtol, atol=1e-7 if npdt == np.float32 else 1e-14)
            n_bit_mismatch += int((g != o).sum())
            total += g.size
        # Pendulum observations are sin/cos of an unwrapped angle (|theta| up to ~80): a 1-ulp difference
        # in theta (4e-6 absolute there) moves sin/cos by as much, hence the absolute tolerance.
        oatol = 2e-5 if kind == "pendulum" else 1e-7
        np.testing.assert_allclose(host(env.last_state()), ref.last_obs, rtol=rtol, atol=oatol)
        np.testing.assert_allclose(host(env.state()), ref.obs(), rtol=rtol, atol=oatol)
    # The kernel evaluates sin/cos in Float64 and rounds once (<= 0.5000001 ulp); the oracle calls the
    # host libm's sinf/cosf (glibc: <= 0.56 ulp, i.e. a wrong last bit in a fraction of a percent of
    # calls; Julia's own Float32 kernels, evaluated in Float64, sit in between).  So most -- not all --
    # values are bit-identical; every value is within the tolerance asserted above.
    limit = 1e-3 if kind == "cartpole" else 1e-2
    assert n_bit_mismatch / total < limit, f"{n_bit_mismatch}/{total} state values differ in the last bit"


@pytest.mark.parametrize("T", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [4096, 1001])
def test_cartpole_reference_test_config_wide_theta(rl, T, n):
    """The reference's OWN CartPole test configuration, `CartPoleEnv(; T = Float32, thetathreshold = 90.0)`
    (RLEnvs/test/environments/examples/cart_pole.jl:7; Float64 is the constructor default, `:57`): with a 90 degree
    threshold the pole swings through |theta| in (pi/4, pi/2] before the episode ends, so the kernel leaves its
    small-argument sin/cos polynomial for the general branch (csrc/env_device.h `Trig<float>::sincos_`).
    Teacher-forced against the oracle: (1) free fall from the reset states until |theta| > pi/2
    terminates the episodes, (2) states injected over the whole range |theta| in (pi/4, pi) incl. the far side of the
    threshold -- flags / counters bit-exact, states within 2e-6 rel (Float32) / 1e-12 (Float64)."""
    npdt = np.float32 if T == torch.float32 else np.float64
    kw = dict(thetathreshold=90.0)
    env = rl.HipVecEnv("cartpole", n, T=T, seed=21, env_id_base=9, **kw)
    ref = oracle.VecEnv("cartpole", n, seed=21, env_id_base=9, dtype=npdt, **kw)
    assert env.cfg.thetathreshold_deg == 90.0
    rng = np.random.default_rng(4)
    rtol = 2e-6 if npdt == np.float32 else 1e-12
    atol = 1e-7 if npdt == np.float32 else 1e-14
    seen_wide = 0
    seen_done_by_theta = 0

    def forced_step(step):
        nonlocal seen_wide, seen_done_by_theta
        s_in = [host(env.raw_state()[k]) for k in range(4)]
        ref.set_state(s_in, host(env._t))
        ref.episode[:] = host(env._episode).view(np.uint32)
        seen_wide += int((np.abs(s_in[2]) > math.pi / 4).sum())
        a = rng.integers(0, 2, n).astype(np.int32)
        env.act0_(dev(a))
        ref.step(a)
        assert np.array_equal(host(env._done), ref.done), f"done flags differ at step {step}"
        assert np.array_equal(host(env._t), ref.t), f"t differs at step {step}"
        assert np.array_equal(host(env._episode).view(np.uint32), ref.episode)
        assert np.array_equal(host(env.reward()), ref.reward)
        for k in range(4):
            np.testing.assert_allclose(host(env.raw_state()[k]), ref.s[k], rtol=rtol, atol=atol)
        np.testing.assert_allclose(host(env.last_state()), ref.last_obs, rtol=rtol, atol=atol)
        seen_done_by_theta += int((ref.done.astype(bool) & (np.abs(ref.last_obs[2]) > math.pi / 2)).sum())

    # (1) free fall: random pushes, the pole tips over in ~60-100 steps
    for step in range(140):
        forced_step(step)
    assert seen_wide > 10 * n, "the free-running phase never left the small-angle branch"
    assert seen_done_by_theta > n // 2, "episodes did not end through the 90 degree threshold"
    # (2) injected states: |theta| uniform in (pi/4, pi), both signs, fast poles and carts
    seen_wide = 0
    for step in range(20):
        th = rng.uniform(math.pi / 4, math.pi, n) * rng.choice([-1.0, 1.0], n)
        s = np.stack([rng.uniform(-2.3, 2.3, n), rng.uniform(-3, 3, n), th, rng.uniform(-6, 6, n)]).astype(npdt)
        env.set_raw_state(s, t=rng.integers(0, 150, n))
        forced_step(1000 + step)
    assert seen_wide == 20 * n


@pytest.mark.parametrize("kind,continuous,T", ENV_CASES)
@pytest.mark.parametrize("n", [4096, 1001])
def test_packed_episode_counters_give_the_same_trajectories(rl, kind, continuous, T, n):
    """rlhip_env_state.episode == NULL: the reset counter shares the step-counter word (include/rlhip.h).  Same
    actions into an env with the separate array and a packed one: every state, reward, flag bit-identical at every
    step, and the unpacked (step, episode) pair equal to the two arrays -- over several auto-resets per env."""
    kw = dict(max_torque_noise=0.4) if kind == "acrobot" else {}
    a = rl.HipVecEnv(kind, n, T=T, continuous=continuous, seed=13, env_id_base=5, max_steps=37, **kw)
    b = rl.HipVecEnv(kind, n, T=T, continuous=continuous, seed=13, env_id_base=5, max_steps=37, packed_episode=True, **kw)
    assert b._episode is None and b.tbits == 6
    npdt = np.float32 if T == torch.float32 else np.float64
    rng = np.random.default_rng(3)

    def same():
        assert torch.equal(a.raw_state(), b.raw_state())
        assert torch.equal(a._t, b.step_counter()) and torch.equal(a._episode, b.episode_counter())
        assert torch.equal(a._done, b._done) and torch.equal(a.reward(), b.reward())
        assert torch.equal(a.state(), b.state())

    same()
    for step in range(150):
        act = dev(_rand_actions(kind, continuous, n, rng, npdt))
        a.act0_(act)
        b.act0_(act)
        if step % 10 == 0 or step > 140:
            same()
            assert torch.equal(a.last_state(), b.last_state())
    assert int(a._episode.min()) >= 4
    # masked and forced reset!, seed!
    a.reset_(is_force=False)
    b.reset_(is_force=False)
    same()
    a.reset_()
    b.reset_()
    same()
    a.seed_(99)
    b.seed_(99)
    a.reset_()
    b.reset_()
    same()
    # the fused policy kernels need the separate array and say so
    if kind == "cartpole" and T == torch.float32 and not continuous:
        from rlhip._lib import RLHipArgumentError

        pol = rl.PPOPolicy(b, update_freq=4)
        with pytest.raises(RLHipArgumentError):
            pol.rollout_()


def test_staggered_layout_of_a_large_env_matches_the_oracle(rl):
    """n >= 2^20: the state arrays are views of one allocation at staggered offsets (rlhip/envs.py) and the step kernel
    takes its non-temporal path -- teacher-forced against the oracle, packed and separate episode counters"""
    n = 1 << 20
    rng = np.random.default_rng(6)
    for packed in (False, True):
        env = rl.HipVecEnv("cartpole", n, seed=2, packed_episode=packed, max_steps=20)
        assert hasattr(env, "_backing") and not env.raw_state().is_contiguous()
        assert (env._s[1].data_ptr() - env._s[0].data_ptr()) % (4 * n) != 0
        ref = oracle.VecEnv("cartpole", n, seed=2, max_steps=20)
        for step in range(30):
            ref.set_state([host(env.raw_state()[k]) for k in range(4)], host(env.step_counter()))
            ref.episode[:] = host(env.episode_counter()).view(np.uint32)
            a = rng.integers(0, 2, n).astype(np.int32)
            env.act0_(dev(a))
            ref.step(a)
            assert np.array_equal(host(env._done), ref.done) and np.array_equal(host(env.step_counter()), ref.t)
            assert np.array_equal(host(env.episode_counter()).view(np.uint32), ref.episode)
            for k in range(4):
                np.testing.assert_allclose(host(env.raw_state()[k]), ref.s[k], rtol=2e-6, atol=1e-7)
        assert int(env.episode_counter().max()) >= 2
        twin = env.copy()
        assert torch.equal(twin.raw_state(), env.raw_state()) and torch.equal(twin._t, env._t)
        del env, twin
        torch.cuda.empty_cache()


@pytest.mark.parametrize("T", [torch.float32, torch.float64])
def test_acrobot_torque_noise_and_wrapper(rl, T):
    """AcrobotEnv: per-step torque noise from the shared Philox stream, reward = -1 after reset!,
    the RLBase spaces, the nips variant; teacher-forced against the oracle"""
    npdt = np.float32 if T == torch.float32 else np.float64
    n = 1000
    for kw in (dict(max_torque_noise=0.7), dict(book_or_nips="nips"), dict(max_torque_noise=0.3, dt=0.1, link_moi=0.8)):
        okw = {("nips" if k == "book_or_nips" else k): (1 if v == "nips" else v) for k, v in kw.items()}
        env = rl.AcrobotRK4Env(n, T=T, seed=9, env_id_base=40, max_steps=20, **kw)
        ref = oracle.VecEnv("acrobot", n, seed=9, env_id_base=40, dtype=npdt, max_steps=20, **okw)
        assert env.name == "AcrobotRK4Env" and len(env.action_space()) == 3 and len(env.state_space()) == 6
        assert (host(env.reward()) == -1).all() and not host(env.is_terminated()).any()
        assert host(env.state()) in env.state_space()
        rng = np.random.default_rng(2)
        for step in range(50):
            ref.set_state([host(env.raw_state()[k]) for k in range(4)], host(env._t))
            ref.episode[:] = host(env._episode).view(np.uint32)
            a = rng.integers(0, 3, n).astype(np.int32)
            env.act0_(dev(a))
            ref.step(a)
            assert np.array_equal(host(env._done), ref.done) and np.array_equal(host(env._t), ref.t)
            assert np.array_equal(host(env.reward()), ref.reward)
            for k in range(4):
                np.testing.assert_allclose(host(env.raw_state()[k]), ref.s[k], rtol=2e-6 if npdt == np.float32 else 1e-12,
                                           atol=1e-7 if npdt == np.float32 else 1e-14)
        assert host(env._episode).min() >= 2


@pytest.mark.parametrize("kind,continuous", [("cartpole", False), ("pendulum", True), ("mountaincar", False)])
def test_env_free_running_episode_statistics(rl, kind, continuous):
    """Free-running 4096 envs x 400 steps with auto-reset: done / t / episode counters identical to the
    oracle at every step (the drift of a chaotic system shows up, if at all, as a flag mismatch)."""
    n = 4096
    env = rl.HipVecEnv(kind, n, seed=11, continuous=continuous)
    ref = oracle.VecEnv(kind, n, seed=11, continuous=continuous)
    rng = np.random.default_rng(1)
    for step in range(400):
        a = _rand_actions(kind, continuous, n, rng, np.float32)
        env.act0_(dev(a))
        ref.step(a)
        if step % 50 == 49 or step < 5:
            assert np.array_equal(host(env._done), ref.done), f"step {step}"
            assert np.array_equal(host(env._t), ref.t), f"step {step}"
    assert np.array_equal(host(env._episode).view(np.uint32), ref.episode)
    for k in range(env.sdim):
        np.testing.assert_allclose(host(env.raw_state()[k]), ref.s[k], rtol=1e-4, atol=1e-5)


def test_env_interface_contract(rl):
    """RLBase.test_interfaces! / test_runnable! restated (RLBase/src/base.jl:77-227): copy determinism
    under equal seeds, state in state_space, actions in action_space, 1000 random steps with auto reset."""
    for kind, cont in (("cartpole", False), ("pendulum", True), ("mountaincar", False), ("mountaincar", True)):
        env = rl.HipVecEnv(kind, 64, seed=7, continuous=cont)
        twin = env.copy()
        g = torch.Generator(device="cpu").manual_seed(0)
        for _ in range(1000):
            assert env.state() in env.state_space()
            if cont:
                lim = env.action_space().hi[0]
                a = ((torch.rand(64, generator=g) * 2 - 1) * lim).cuda()
            else:
                a = torch.randint(1, len(env.action_space()) + 1, (64,), generator=g).cuda()
                assert a in env.action_space()
            env.act_(a)
            twin.act_(a)
        assert torch.equal(env.state(), twin.state())
        assert torch.equal(env.reward(), twin.reward())
        assert torch.equal(env.is_terminated(), twin.is_terminated())
        # invalid discrete action is an AssertionError like `@assert a in action_space(env)`
        if not cont:
            checked = rl.HipVecEnv(kind, 4, seed=7, validate_actions=True)
            with pytest.raises(AssertionError):
                checked.act_(torch.tensor([1, 2, 9, 1]).cuda())


def test_cartpole_reward_and_termination_rules(rl):
    # strict `>` on max_steps: an episode lasts max_steps + 1 steps; the terminal step pays 0
    env = rl.HipVecEnv("cartpole", 4, seed=1, max_steps=5, auto_reset=False, xthreshold=1e9, thetathreshold=1e9)
    a = torch.ones(4, dtype=torch.int64).cuda()
    rewards = []
    for i in range(6):
        env.act_(a)
        rewards.append(float(env.reward()[0]))
        assert bool(env.is_terminated()[0]) == (i == 5)
    assert rewards == [1, 1, 1, 1, 1, 0]


# ---------------------------------------------------------------------------------------- scans
def _cm(rows, dtype):
    from rlhip import ops

    a = np.array(rows, dtype=float)
    return ops.from_julia(a, dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_scans_golden_on_gpu(rl, dtype):
    from rlhip import ops
    from rlhip._lib import RLHipArgumentError

    S = load("scans")
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1.5e-8, atol=1e-12)
    for case in S["discount_rewards"]:
        r = _cm(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype)
        term = case.get("terminal_rows", case.get("terminal"))
        term = None if term is None else ops.from_julia(np.array(term), dtype=torch.uint8)
        init = case.get("init")
        if isinstance(init, list):
            init = dev(init, dtype)
        if case.get("expect_error"):
            with pytest.raises(RLHipArgumentError):
                ops.discount_rewards(r, case["gamma"], terminal=term, init=init, dims=case.get("dims", 0))
            continue
        out = ops.to_julia(ops.discount_rewards(r, case["gamma"], terminal=term, init=init, dims=case.get("dims", 0)))
        exp = np.array(case["expect_rows"] if "expect_rows" in case else case["expect"])
        np.testing.assert_allclose(host(out), exp, **tol, err_msg=case["src"])
    for case in S["discount_rewards_reduced"]:
        r = _cm(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype)
        term = case.get("terminal_rows", case.get("terminal"))
        term = None if term is None else ops.from_julia(np.array(term), dtype=torch.uint8)
        init = case.get("init")
        if isinstance(init, list):
            init = dev(init, dtype)
        if case.get("expect_error"):
            with pytest.raises(RLHipArgumentError):
                ops.discount_rewards_reduced(r, case["gamma"], terminal=term, init=init, dims=case.get("dims", 0))
            continue
        out = ops.discount_rewards_reduced(r, case["gamma"], terminal=term, init=init, dims=case.get("dims", 0))
        np.testing.assert_allclose(host(out), np.array(case["expect"]), **tol, err_msg=case["src"])
    for case in S["generalized_advantage_estimation"]:
        r = _cm(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype)
        v = _cm(case["values_rows"] if "values_rows" in case else case["values"], dtype)
        term = case.get("terminal_rows", case.get("terminal"))
        term = None if term is None else ops.from_julia(np.array(term), dtype=torch.uint8)
        if case.get("expect_error"):
            with pytest.raises(RLHipArgumentError):
                ops.generalized_advantage_estimation(r, v, case["gamma"], case["lam"], terminal=term,
                                                     dims=case.get("dims", 0))
            continue
        out = ops.to_julia(ops.generalized_advantage_estimation(r, v, case["gamma"], case["lam"], terminal=term,
                                                                dims=case.get("dims", 0)))
        exp = np.array(case["expect_rows"] if "expect_rows" in case else case["expect"])
        np.testing.assert_allclose(host(out), exp, **tol, err_msg=case["src"])


@pytest.mark.parametrize("npdt,tdt", [(np.float32, torch.float32), (np.float64, torch.float64)])
@pytest.mark.parametrize("dims", [1, 2])
def test_scans_random_bit_exact(rl, npdt, tdt, dims):
    """Same operation order, no FMA contraction -> bit-identical to the oracle, any shape."""
    from rlhip import ops

    rng = np.random.default_rng(7)
    n1, n2 = (37, 1000) if dims == 1 else (1000, 37)
    T = n1 if dims == 1 else n2
    r = rng.uniform(-16, 0, (n1, n2)).astype(npdt)
    vshape = (n1 + 1, n2) if dims == 1 else (n1, n2 + 1)
    v = rng.standard_normal(vshape).astype(npdt)
    term = (rng.random((n1, n2)) < 1 / 20)
    init = rng.standard_normal(n2 if dims == 1 else n1).astype(npdt)
    o_gae = oracle.generalized_advantage_estimation(r, v, 0.99, 0.95, terminal=term, dims=dims, dtype=npdt)
    g_gae = ops.to_julia(ops.generalized_advantage_estimation(ops.from_julia(r), ops.from_julia(v), 0.99, 0.95,
                                                              terminal=ops.from_julia(term), dims=dims))
    assert np.array_equal(host(g_gae), o_gae)
    o_d = oracle.discount_rewards(r, 0.99, terminal=term, init=init, dims=dims, dtype=npdt)
    g_d = ops.to_julia(ops.discount_rewards(ops.from_julia(r), 0.99, terminal=ops.from_julia(term), init=dev(init),
                                            dims=dims))
    assert np.array_equal(host(g_d), o_d)
    o_r = oracle.discount_rewards_reduced(r, 0.99, terminal=term, init=init, dims=dims, dtype=npdt)
    g_r = ops.discount_rewards_reduced(ops.from_julia(r), 0.99, terminal=ops.from_julia(term), init=dev(init), dims=dims)
    assert np.array_equal(host(g_r), o_r)
    assert T > 0


def test_gae_fused_returns_full_size(rl):
    """BASELINE config-3 shape (N = 4096, T = 128): fused advantages + returns, bit-exact vs the oracle."""
    from rlhip import ops

    rng = np.random.default_rng(7)
    n, T = 4096, 128
    r = rng.uniform(-16, 0, (T, n)).astype(np.float32)
    v = rng.standard_normal((T + 1, n)).astype(np.float32)
    term = rng.random((T, n)) < 1 / 200
    adv, ret = ops.gae_returns(dev(r), dev(v), dev(term), 0.99, 0.95)
    # time-major (T, n) storage == column-major (n, T): dims = 2
    o = oracle.generalized_advantage_estimation(r.T, v.T, 0.99, 0.95, terminal=term.T, dims=2, dtype=np.float32)
    assert np.array_equal(host(adv), o.T)
    assert np.array_equal(host(ret), (o.T + v[:T]).astype(np.float32))


@pytest.mark.parametrize("n,T,with_term", [(65536, 64, True), (131072, 37, True), (65536, 70, False), (65538, 64, True)])
def test_gae_streaming_four_envs_per_lane_bit_exact(rl, n, T, with_term):
    """scans beyond the L2 (n T >= 2^22) in the env-major Float32 layout take gae_vec4_kernel (four envs per lane,
    16-byte accesses, chunks of 8 steps); n not a multiple of 4 stays on the scalar kernel: both bit-exact vs the oracle,
    advantages and returns, with the chunk remainder (T mod 8 != 0) and without terminal flags"""
    from rlhip import ops

    rng = np.random.default_rng(n + T)
    r = rng.uniform(-16, 0, (T, n)).astype(np.float32)
    v = rng.standard_normal((T + 1, n)).astype(np.float32)
    term = (rng.random((T, n)) < 1 / 50) if with_term else None
    adv, ret = ops.gae_returns(dev(r), dev(v), dev(term) if with_term else None, 0.99, 0.95)
    o = oracle.generalized_advantage_estimation(r.T, v.T, 0.99, 0.95, terminal=term.T if with_term else None, dims=2,
                                                dtype=np.float32)
    assert np.array_equal(host(adv), o.T)
    assert np.array_equal(host(ret), (o.T + v[:T]).astype(np.float32))


# ------------------------------------------------------------------------------------ selection
def test_selection_golden_on_gpu(rl):
    from rlhip import ops

    S = load("select")
    c = S["greedy_plan"][0]
    v = dev(np.tile(np.array(c["values"], np.float32)[:, None], (1, 5)))
    a = ops.eps_greedy_select(v, 0.0, seed=1, step=1)
    assert (host(a) + 1 == c["expect"]).all()  # first-index tie rule, Julia 1-based
    p = S["get_eps_params"]
    for case in S["get_eps"]:
        e = ops.get_eps(case["kind"], p["eps_stable"], p["eps_init"], p["warmup_steps"], p["decay_steps"], case["step"])
        assert abs(e - case["expect"]) <= case.get("atol", 1e-12)


@pytest.mark.parametrize("na", [2, 3, 4, 18])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("tie", [False, True])
def test_eps_greedy_bit_exact(rl, na, masked, tie):
    from rlhip import ops

    rng = np.random.default_rng(na)
    n = 5000
    q = rng.standard_normal((na, n)).astype(np.float32)
    q[:, ::7] = np.round(q[:, ::7])  # ties
    q[0, 5] = np.nan
    q[:, 6] = -np.inf
    mask = None
    if masked:
        mask = (rng.random((na, n)) < 0.7).astype(np.uint8)
        mask[rng.integers(0, na, n), np.arange(n)] = 1  # at least one legal action
    for eps in (0.0, 0.3, 1.0):
        g = ops.eps_greedy_select(dev(q), eps, seed=42, step=17, env_id_base=1000,
                                  mask=None if mask is None else dev(mask), is_break_tie=tie)
        o = oracle.eps_greedy_select(q, eps, seed=42, step=17, env_id_base=1000, mask=mask, is_break_tie=tie)
        assert np.array_equal(host(g), o)
        # Julia (na, N) column-major layout gives the same answer
        g2 = ops.eps_greedy_select(dev(np.ascontiguousarray(q.T)), eps, seed=42, step=17, env_id_base=1000,
                                   mask=None if mask is None else dev(np.ascontiguousarray(mask.T)),
                                   is_break_tie=tie, soa=False)
        assert np.array_equal(host(g2), o)
        if masked:
            # a masked action is never selected -- except in the degenerate all -Inf column, where
            # findmax(ifelse.(mask, A, typemin(T))) returns index 1 whatever the mask (reference semantics)
            ok = mask[o, np.arange(n)].astype(bool)
            ok[6] = True
            assert ok.all()


@pytest.mark.parametrize("na", [2, 3, 6])
def test_categorical_bit_exact(rl, na):
    from rlhip import ops

    rng = np.random.default_rng(3)
    n = 20000
    logits = (rng.standard_normal((na, n)) * 2).astype(np.float32)
    a, lp = ops.categorical_sample(dev(logits), seed=9, step=123, env_id_base=5)
    oa, olp = oracle.categorical_sample(logits, seed=9, step=123, env_id_base=5)
    assert np.array_equal(host(a), oa)
    assert np.array_equal(host(lp), olp)
    mask = (rng.random((na, n)) < 0.6).astype(np.uint8)
    mask[0] = 1
    a, lp = ops.categorical_sample(dev(logits), seed=9, step=124, mask=dev(mask))
    oa, olp = oracle.categorical_sample(logits, seed=9, step=124, mask=mask)
    assert np.array_equal(host(a), oa)
    assert mask[oa, np.arange(n)].all()  # RLCore/test/utils/networks.jl:326-362: masked never sampled
    # sampling frequencies follow softmax
    if na == 2:
        l2 = np.tile(np.array([[0.0], [1.0]], np.float32), (1, n))
        a, _ = ops.categorical_sample(dev(l2), seed=1, step=0)
        assert abs(host(a).mean() - 1 / (1 + math.exp(-1.0))) < 0.02


# -------------------------------------------------------------------------------------- updates
def test_polyak_and_target_sync(rl):
    from rlhip import ops

    rng = np.random.default_rng(0)
    src = rng.standard_normal(17410).astype(np.float32)
    dst = rng.standard_normal(17410).astype(np.float32)
    for rho in (0.0, 0.5, 0.995):
        d = dev(dst.copy())
        ops.polyak_(d, dev(src), rho)
        o = oracle.polyak(dst.copy(), src, rho)
        assert np.array_equal(host(d), o)
    with pytest.raises(ValueError):
        ops.polyak_(dev(dst), dev(src), 1.5)  # @assert 0 <= rho <= 1 (target_network.jl:50)
    # the 16-byte streaming kernel (n >= 65536, three tail elements): bit-exact like the scalar one
    n = (1 << 17) + 3
    src, dst = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    d = dev(dst.copy())
    ops.polyak_(d, dev(src), 0.995)
    assert np.array_equal(host(d), oracle.polyak(dst.copy(), src, 0.995))


@pytest.mark.parametrize("n", [3331, 17410, 300000])
def test_clip_by_global_norm(rl, n):
    from rlhip import ops

    rng = np.random.default_rng(n)
    g = (rng.standard_normal(n) * 0.01).astype(np.float32)
    for clip in (0.5, 1e6):
        d = dev(g.copy())
        gn = ops.clip_by_global_norm_(d, clip)
        o = g.copy()
        ogn = oracle.clip_by_global_norm(o, clip)
        assert float(gn) == pytest.approx(ogn, rel=1e-6)
        np.testing.assert_allclose(host(d), o, rtol=1e-6, atol=0)
        if clip > ogn:
            assert np.array_equal(host(d), g)  # untouched when not clipped


@pytest.mark.parametrize("n", [3331, 100000, 65536 + 3, 1 << 20])
def test_adam_matches_oracle_and_torch(rl, n):
    from rlhip import ops

    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(n).astype(np.float32)
    p = dev(p0.copy())
    m = torch.zeros(n).cuda()
    v = torch.zeros(n).cuda()
    bp = torch.tensor([0.9, 0.999]).cuda()
    po, mo, vo = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    pt = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 6):
        g = (rng.standard_normal(n) * 0.1).astype(np.float32)
        ops.adam_(p, dev(g), m, v, bp)
        oracle.adam(po, g, mo, vo, 1e-3, 0.9, 0.999, 1e-8, t)
        pt.grad = torch.tensor(g)
        opt.step()
        np.testing.assert_allclose(host(p), po, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(host(p), pt.detach().numpy(), rtol=1e-5, atol=1e-6)  # torch fp32 reference
    np.testing.assert_allclose(host(bp), [0.9 ** 6, 0.999 ** 6], rtol=1e-6)


def test_fused_clip_adam_equals_unfused(rl):
    from rlhip import ops

    rng = np.random.default_rng(2)
    for n in (3331, 4097, 8195, 17410, 30001, 34435, 49152, 65536, 70000):  # scalar, vectorised (2, 4, 8, 16 x 4096) and grid-wide paths
        p0 = rng.standard_normal(n).astype(np.float32)
        g0 = rng.standard_normal(n).astype(np.float32)
        pa, pb = dev(p0.copy()), dev(p0.copy())
        ga, gb = dev(g0.copy()), dev(g0.copy())
        ma, va, mb, vb = (torch.zeros(n).cuda() for _ in range(4))
        bpa, bpb = torch.tensor([0.9, 0.999]).cuda(), torch.tensor([0.9, 0.999]).cuda()
        gn = torch.zeros(1).cuda()
        ops.clip_adam_(pa, ga, ma, va, bpa, grad_scale=0.5, clip_norm=0.5, gn_out=gn)
        gb *= 0.5
        gn_b = ops.clip_by_global_norm_(gb, 0.5)
        ops.adam_(pb, gb, mb, vb, bpb)
        assert float(gn) == pytest.approx(float(gn_b), rel=1e-6)
        np.testing.assert_allclose(host(pa), host(pb), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(host(bpa), host(bpb))


def test_normlogpdf_huber_td_target(rl):
    from rlhip import ops

    rng = np.random.default_rng(5)
    n = 4096
    mu, x = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    sg = rng.uniform(0.1, 3, n).astype(np.float32)
    g = host(ops.normlogpdf(dev(mu), dev(sg), dev(x)))
    o = np.array([oracle.normlogpdf(float(a), float(b), float(c)) for a, b, c in zip(mu[:300], sg[:300], x[:300])],
                 np.float32)
    np.testing.assert_allclose(g[:300], o, rtol=1e-5, atol=1e-6)
    ref = torch.distributions.Normal(torch.tensor(mu), torch.tensor(sg)).log_prob(torch.tensor(x)).numpy()
    np.testing.assert_allclose(g, ref, rtol=1e-4, atol=1e-5)
    d = 3
    MU, SG, X = (rng.standard_normal((d, n)).astype(np.float32), rng.uniform(0.2, 2, (d, n)).astype(np.float32),
                 rng.standard_normal((d, n)).astype(np.float32))
    gd = host(ops.diagnormlogpdf(dev(MU.T.copy()), dev(SG.T.copy()), dev(X.T.copy())))
    np.testing.assert_allclose(gd, oracle.diagnormlogpdf(MU, SG, X), rtol=1e-5, atol=1e-5)
    q, tg = rng.standard_normal(n).astype(np.float32) * 2, rng.standard_normal(n).astype(np.float32)
    loss, dq = ops.huber_loss(dev(q), dev(tg), 1.0)
    ol, odq = oracle.huber(q, tg, 1.0)
    assert float(loss) == pytest.approx(ol, rel=1e-6)
    assert np.array_equal(host(dq), odq)
    assert float(loss) == pytest.approx(float(torch.nn.HuberLoss(delta=1.0)(torch.tensor(q), torch.tensor(tg))), rel=1e-5)
    qn = rng.standard_normal((3, n)).astype(np.float32)
    r = rng.standard_normal(n).astype(np.float32)
    term = rng.random(n) < 0.1
    assert np.array_equal(host(ops.td_target(dev(qn), dev(r), dev(term), 0.99)), oracle.td_target(qn, r, term, 0.99))


# ----------------------------------------------------------------------------------------- ring
@pytest.mark.parametrize("n_env,obs_dim,capacity", [(1, 4, 7), (64, 4, 5), (4096, 4, 3), (3, 2, 10)])
def test_ring_push_sample_gather_bit_exact(rl, n_env, obs_dim, capacity):
    from rlhip.trajectory import CircularArraySARTSTraces

    rng = np.random.default_rng(0)
    tr = CircularArraySARTSTraces(capacity=capacity, n_env=n_env, obs_dim=obs_dim)
    ref = oracle.Ring(capacity, n_env, obs_dim)
    o0 = rng.standard_normal((obs_dim, n_env)).astype(np.float32)
    tr.push_state_(dev(o0))
    ref.push_state(o0)
    assert len(tr) == 0 == len(ref)  # RLCore/test/policies/agent.jl:27-34
    for step in range(2 * capacity + 3):  # wraps around twice
        o = rng.standard_normal((obs_dim, n_env)).astype(np.float32)
        a = rng.integers(0, 2, n_env).astype(np.int32)
        r = rng.standard_normal(n_env).astype(np.float32)
        t = (rng.random(n_env) < 0.2).astype(np.uint8)
        tr.push_transition_(dev(o), dev(a), dev(r), dev(t))
        ref.push_transition(o, a, r, t)
        assert len(tr) == len(ref) == min(step + 1, capacity)
        batch = 333
        idx = tr.sample_indices(batch, seed=77, draw_ctr=step)
        oidx = ref.sample_indices(batch, 77, step)
        assert np.array_equal(host(idx), oidx)
        s, aa, rr, tt, sn = tr.gather(idx)
        os_, oa, or_, ot, osn = ref.gather(oidx)
        assert np.array_equal(host(s), os_) and np.array_equal(host(sn), osn)
        assert np.array_equal(host(aa), oa) and np.array_equal(host(rr), or_) and np.array_equal(host(tt), ot)
    # multiplexed next_state: transition i's next state is transition i+1's state
    idx = dev(np.arange(0, (len(ref) - 1) * n_env, n_env, dtype=np.int64))
    s, _, _, _, sn = tr.gather(idx)
    idx2 = idx + n_env
    s2, _, _, _, _ = tr.gather(idx2)
    assert torch.equal(sn, s2)


def test_ring_u8_frames(rl):
    """Atari-style 84x84x4 u8 frames, n_env = 1 -> the streaming frame-gather kernel."""
    from rlhip.trajectory import CircularArraySARTSTraces

    rng = np.random.default_rng(11)
    fb = 84 * 84 * 4
    cap = 16
    tr = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=fb, dtype=torch.uint8)
    frames = rng.integers(0, 256, (cap + 12, fb), dtype=np.uint8)
    tr.push_state_(dev(frames[0]))
    for i in range(1, cap + 12):
        tr.push_transition_(dev(frames[i]), dev(np.array([i % 3], np.int32)), dev(np.array([float(i)], np.float32)),
                            dev(np.array([i % 5 == 0], np.uint8)))
    idx = tr.sample_indices(64, seed=1, draw_ctr=0)
    s, a, r, t, sn = tr.gather(idx)
    li = host(idx)
    first = (cap + 12 - 1) - cap  # logical transition 0 starts at this frame
    for b in range(64):
        f = first + li[b]
        assert np.array_equal(host(s[b]), frames[f])
        assert np.array_equal(host(sn[b]), frames[f + 1])
        assert host(r)[b] == float(f + 1) and host(a)[b] == (f + 1) % 3


# ------------------------------------------------------------------------------------------ MLP
@pytest.mark.parametrize("n_in,h,n_out,act", [(4, 256, 2, 0), (4, 128, 2, 0), (3, 256, 2, 1), (2, 64, 3, 0), (4, 100, 5, 1)])
def test_mlp2_forward_and_init(rl, n_in, h, n_out, act):
    from rlhip import ops

    p = ops.mlp2_init(n_in, h, n_out, seed=123, net_id=1)
    po = oracle.mlp2_init(n_in, h, n_out, 123, 1)
    assert np.array_equal(host(p), po)  # same Philox INIT stream, same Float32 expression
    rng = np.random.default_rng(0)
    p = dev((host(p) + rng.standard_normal(po.size) * 0.1).astype(np.float32))  # non-zero biases
    x = rng.standard_normal((n_in, 1000)).astype(np.float32)
    out = host(ops.mlp2_forward(p, n_in, h, n_out, act, dev(x)))
    o = oracle.mlp2_forward(host(p), n_in, h, n_out, act, x)
    np.testing.assert_allclose(out, o, rtol=1e-5, atol=1e-6)
    # torch fp32 reference of the same op
    W1 = torch.tensor(host(p)[: h * n_in].reshape(n_in, h).T)
    b1 = torch.tensor(host(p)[h * n_in: h * n_in + h])
    W2 = torch.tensor(host(p)[h * n_in + h: h * n_in + h + n_out * h].reshape(h, n_out).T)
    b2 = torch.tensor(host(p)[h * n_in + h + n_out * h:])
    hid = W1 @ torch.tensor(x) + b1[:, None]
    hid = torch.relu(hid) if act == 0 else torch.tanh(hid)
    ref = (W2 @ hid + b2[:, None]).numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)
