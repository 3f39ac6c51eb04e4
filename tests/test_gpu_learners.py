"""GPU parity tests of the learner path (policy forward + sampling, fused rollout, PPO / DQN loss and
gradient, update loop) through the C ABI, against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from conftest import F32_GRAD_TOL, assert_grad_close  # noqa: E402


@pytest.fixture(scope="module")
def rl():
    import rlhip

    return rlhip


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def host(t):
    return t.detach().cpu().numpy()


def make_pair(rl, kind, n, T, seed=3, continuous=None, hidden=256, **kw):
    """(gpu env, gpu policy, oracle env, oracle cfg) with identical seeds / params."""
    env = rl.HipVecEnv(kind, n, seed=seed, continuous=continuous)
    pol = rl.PPOPolicy(env, update_freq=T, hidden=hidden, **kw)
    oenv = oracle.VecEnv(kind, n, seed=seed, continuous=env.continuous)
    ocfg = oracle.ppo_default(continuous=int(env.continuous), hidden=hidden,
                              **{("lam" if k == "lam" else k): v for k, v in kw.items()})
    return env, pol, oenv, ocfg


@pytest.mark.parametrize("kind,continuous,hidden", [("cartpole", False, 256), ("cartpole", False, 128),
                                                    ("mountaincar", False, 64), ("pendulum", True, 256),
                                                    ("cartpole", False, 96)])
def test_policy_init_and_plan_vs_oracle(rl, kind, continuous, hidden):
    n = 2048
    env, pol, oenv, ocfg = make_pair(rl, kind, n, 8, continuous=continuous, hidden=hidden)
    ns, na = env.odim, pol.na
    nout = 2 * na if continuous else na
    # identical initial parameters (Philox INIT stream)
    pa = oracle.mlp2_init(ns, hidden, nout, pol.seed, 0)
    pc = oracle.mlp2_init(ns, hidden, 1, pol.seed, 1)
    assert np.array_equal(host(pol.params), np.concatenate([pa, pc]))
    assert pol.np == oracle.ppo_nparams(env.kind, ocfg)
    # perturb so that biases are non-zero
    rng = np.random.default_rng(0)
    p = (host(pol.params) + rng.standard_normal(pol.np) * 0.05).astype(np.float32)
    pol.params.copy_(dev(p))
    a = host(pol.plan_())
    obs = host(env.state())
    out = oracle.mlp2_forward(p[: pa.size], ns, hidden, nout, 0, obs)
    val = oracle.mlp2_forward(p[pa.size:], ns, hidden, 1, 0, obs)[0]
    np.testing.assert_allclose(host(pol._value), val, rtol=2e-5, atol=2e-6)
    if not continuous:
        oa, olp = oracle.categorical_sample(out, seed=pol.seed, step=0)
        agree = (a - 1 == oa).mean()
        assert agree > 0.999, f"only {agree:.5f} of the sampled actions agree"
        same = a - 1 == oa
        np.testing.assert_allclose(host(pol._logp)[same], olp[same], rtol=2e-5, atol=2e-6)
    else:
        # z = mu + sigma * noise; logp = normlogpdf
        mu, ls = out[0], out[1]
        lp = np.array([oracle.normlogpdf(float(m), float(np.exp(np.float32(s))), float(z))
                       for m, s, z in zip(mu[:200], ls[:200], a[:200])], np.float32)
        np.testing.assert_allclose(host(pol._logp)[:200], lp, rtol=1e-4, atol=1e-5)
        noise = (a - mu) / np.exp(ls)
        assert abs(noise.mean()) < 0.1 and abs(noise.std() - 1) < 0.1


@pytest.mark.parametrize("kind,continuous,hidden", [("cartpole", False, 256), ("pendulum", True, 256),
                                                    ("mountaincar", False, 128), ("cartpole", False, 96)])
def test_rollout_fused_equals_stepwise_bit_exact(rl, kind, continuous, hidden):
    """One launch for T vec-steps must reproduce the per-step plan!/push!/act!/push! protocol exactly."""
    n, T = 1000, 12
    envA, polA, _, _ = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden)
    envB, polB, _, _ = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden)
    for it in range(2):  # two consecutive update periods (counters carry over)
        polA.rollout_()
        for t in range(T):
            a = polB.plan_()
            polB.push_preact_()
            envB.act_(a)
            polB.push_postact_()
        polB.finish_rollout_()
        ta, tb = polA.trajectory, polB.trajectory
        for name in ("obs", "logp", "value", "reward", "terminal"):
            assert torch.equal(getattr(ta, name), getattr(tb, name)), f"{name} differs (period {it})"
        assert torch.equal(ta.action, tb.action)
        assert torch.equal(envA.raw_state(), envB.raw_state())
        assert torch.equal(envA._t, envB._t) and torch.equal(envA._episode, envB._episode)
        assert polA.vec_step == polB.vec_step == (it + 1) * T


@pytest.mark.parametrize("kind,continuous,hidden,act,n,T", [
    ("cartpole", False, 256, 0, 37, 40),      # two-action head; two noise-chunk boundaries (16, 32); n not a multiple of 16
    ("pendulum", False, 64, 1, 130, 33),      # three-action head, tanh, 4 lanes per env (64 envs per workgroup), one step past 32
    ("mountaincar", False, 128, 0, 16, 49),   # three chunks + one step; exactly half a workgroup of envs
    ("pendulum", True, 256, 1, 1000, 17),     # Gaussian head, tanh
    ("cartpole", True, 128, 0, 5, 16),        # Gaussian head on CartPole; T = exactly one chunk; fewer envs than one workgroup
    ("cartpole", False, 256, 0, 9, 1),        # a single step
])
def test_rollout_two_wave_kernel_equals_stepwise_bit_exact(rl, kind, continuous, hidden, act, n, T):
    """rollout_split_kernel (actor wave + critic wave per env group, step records and sampling noise handed over through LDS,
    noise produced one 16-step chunk ahead) against the per-step protocol, which shares none of that machinery: every trace,
    the env state and the fused GAE scan bit for bit, over chunk boundaries, ragged env counts and all head kinds."""
    envA, polA, _, _ = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden, act=act)
    envB, polB, _, _ = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden, act=act)
    for it in range(3):
        polA.rollout_()
        for t in range(T):
            a = polB.plan_()
            polB.push_preact_()
            envB.act_(a)
            polB.push_postact_()
        polB.finish_rollout_()
        polB.gae_()
        ta, tb = polA.trajectory, polB.trajectory
        for name in ("obs", "logp", "value", "reward", "terminal", "adv", "ret"):
            assert torch.equal(getattr(ta, name), getattr(tb, name)), f"{name} differs (period {it})"
        assert torch.equal(ta.action, tb.action)
        assert torch.equal(envA.raw_state(), envB.raw_state())
        assert torch.equal(envA._t, envB._t) and torch.equal(envA._episode, envB._episode)
        assert torch.equal(envA.reward(), envB.reward()) and torch.equal(envA._done, envB._done)


@pytest.mark.parametrize("kind,continuous", [("cartpole", False), ("mountaincar", False), ("pendulum", True)])
def test_rollout_vs_oracle(rl, kind, continuous):
    """Free-running comparison with the CPU restatement of the whole rollout: integer traces identical,
    Float32 traces within 1e-5 (the MLP sums are evaluated in a different order on the GPU)."""
    n, T = 512, 24
    env, pol, oenv, ocfg = make_pair(rl, kind, n, T, continuous=continuous)
    p = host(pol.params)
    pol.rollout_()
    otr = oracle.PPOTraj(env.kind, n, T, na=1, continuous=continuous)
    oracle.ppo_rollout(oenv, T, ocfg, p, otr, 0)
    tr = pol.trajectory
    # valid[t, i]: env i has taken the same actions as the oracle's env i in steps 0..t-1, so that its
    # step-t inputs (observation, value, RNG counters) are the same on both sides.  A near-tie of the
    # Gumbel-max sampler may flip an action (the logits differ in the last bits: different summation
    # order); from there on that ONE env legitimately diverges, every other env is still compared.
    if not continuous:
        flipped = host(tr.action_i) != otr.action_i  # (T, n)
        first_flip = np.where(flipped.any(0), flipped.argmax(0), T)  # per env
        assert (first_flip < T).sum() <= 2, f"{(first_flip < T).sum()} of {n} envs saw a flipped action"
        before = np.arange(T)[:, None] < first_flip[None, :]  # steps whose action agreed and all before it did
        upto = np.arange(T + 1)[:, None] <= first_flip[None, :]  # inputs of those steps (+ the flip step itself)
        assert np.array_equal(host(tr.terminal)[before], otr.terminal[before])
        assert np.array_equal(host(tr.reward)[before], otr.reward[before])
        tol = dict(rtol=1e-5, atol=1e-6)
    else:
        tol = dict(rtol=2e-3, atol=2e-3)  # continuous actions differ in the last bits -> mild drift over T steps
        np.testing.assert_allclose(host(tr.action_f), otr.action_f, **tol)
        assert np.array_equal(host(tr.terminal), otr.terminal)
        before = np.ones((T, n), bool)
        upto = np.ones((T + 1, n), bool)
    om = np.broadcast_to(upto[:, None, :], otr.obs.shape)
    np.testing.assert_allclose(host(tr.obs)[om], otr.obs[om], **tol)
    np.testing.assert_allclose(host(tr.logp)[before], otr.logp[before], rtol=1e-3, atol=1e-4 if continuous else 1e-5)
    np.testing.assert_allclose(host(tr.value)[upto], otr.value[upto], **tol)
    # GAE + returns on the GPU trajectory: bit-exact vs the oracle scan on the same inputs
    pol.gae_()
    o = oracle.generalized_advantage_estimation(host(tr.reward).T, host(tr.value).T, 0.99, 0.95,
                                                terminal=host(tr.terminal).T, dims=2, dtype=np.float32)
    assert np.array_equal(host(tr.adv), o.T)
    assert np.array_equal(host(tr.ret), (o.T + host(tr.value)[:T]).astype(np.float32))


def _oracle_microbatch(pol, tr, epoch_ctr, mb):
    n, T = tr.n, tr.T
    total = n * T
    bm = total // pol.cfg.n_microbatches
    perm = np.array([oracle.permute(pol.seed, epoch_ctr, total, mb * bm + b) for b in range(bm)])
    t, i = perm // n, perm % n
    obs = host(tr.obs)[t, :, i].T.copy()  # (ns, bm)
    flat = lambda x: host(x).reshape(-1)[perm]  # noqa: E731
    act = host(tr.action_f)[t, 0, i] if tr.continuous else host(tr.action_i).reshape(-1)[perm]
    return obs, act, flat(tr.logp), flat(tr.adv), flat(tr.ret), perm


@pytest.mark.parametrize("kind,continuous,hidden,act", [("cartpole", False, 256, 0), ("cartpole", False, 64, 1),
                                                        ("pendulum", True, 256, 0), ("mountaincar", False, 128, 0)])
def test_ppo_loss_and_gradient_vs_oracle(rl, kind, continuous, hidden, act):
    n, T = 256, 16  # micro-batch of 1024 samples, 16 tiles
    env, pol, oenv, ocfg = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden, act=act)
    rng = np.random.default_rng(1)
    p = (host(pol.params) + rng.standard_normal(pol.np) * 0.05).astype(np.float32)
    pol.params.copy_(dev(p))
    pol.rollout_()
    pol.gae_()
    # move the policy away from the behaviour policy so that ratios leave the clip range
    p2 = (p + rng.standard_normal(pol.np) * 0.02).astype(np.float32)
    pol.params.copy_(dev(p2))
    tr = pol.trajectory
    for epoch_ctr, mb in ((0, 0), (0, 3), (5, 1)):
        pol.grad_(epoch_ctr, mb)
        obs, a, lp, adv, ret, perm = _oracle_microbatch(pol, tr, epoch_ctr, mb)
        g, losses = oracle.ppo_loss_grad(ocfg, env.odim, pol.na, p2, obs, a, lp, adv, ret)
        assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, f"ppo_grad {kind} h={hidden} act={act} mb={mb}")
        np.testing.assert_allclose(host(pol.losses), losses, rtol=1e-4, atol=1e-6)
    # a permutation epoch covers every transition exactly once
    allidx = np.concatenate([_oracle_microbatch(pol, tr, 7, mb)[5] for mb in range(pol.cfg.n_microbatches)])
    assert np.array_equal(np.sort(allidx), np.arange(n * T))


def test_ppo_ragged_microbatch(rl):
    """n*T not divisible by the tile size / micro-batch count: partial tiles contribute nothing extra."""
    n, T = 100, 7  # 700 transitions, 3 micro-batches of 233 (1 dropped), 233 = 3 tiles + 41
    env, pol, oenv, ocfg = make_pair(rl, "cartpole", n, T, n_microbatches=3)
    ocfg.n_microbatches = 3
    pol.rollout_()
    pol.gae_()
    tr = pol.trajectory
    pol.grad_(2, 2)
    obs, a, lp, adv, ret, _ = _oracle_microbatch(pol, tr, 2, 2)
    g, losses = oracle.ppo_loss_grad(ocfg, 4, 2, host(pol.params), obs, a, lp, adv, ret)
    assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, "ppo_grad ragged")
    np.testing.assert_allclose(host(pol.losses), losses, rtol=1e-4, atol=1e-6)


def test_ppo_update_equals_manual_sequence_and_tracks_oracle(rl):
    n, T = 256, 16
    envA, polA, oenv, ocfg = make_pair(rl, "cartpole", n, T)
    envB, polB, _, _ = make_pair(rl, "cartpole", n, T)
    polA.rollout_()
    polB.rollout_()
    p0 = host(polA.params).copy()
    polA.update_()  # the single enqueue-everything entry point
    polB.gae_()
    for e in range(polB.cfg.n_epochs):  # the multi-GPU style sequence: grad -> (all-reduce) -> clip+Adam
        for mb in range(polB.cfg.n_microbatches):
            polB.grad_(e, mb)
            polB.apply_(1.0)
    polB.update_ctr += 1
    assert torch.equal(polA.params, polB.params)
    assert torch.equal(polA.m, polB.m) and torch.equal(polA.v, polB.v) and torch.equal(polA.beta_pow, polB.beta_pow)
    # oracle: same trajectory (copied from the GPU), same update loop on the CPU
    tr = polA.trajectory
    otr = oracle.PPOTraj(0, n, T)
    for name in ("obs", "logp", "value", "reward", "action_i", "terminal"):
        getattr(otr, name)[...] = host(getattr(tr, name))
    oracle.ppo_gae(ocfg, otr)
    assert np.array_equal(otr.adv, host(tr.adv))
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    steps, _ = oracle.ppo_update(0, ocfg, otr, po, mo, vo, 0, polA.seed, 0)
    assert steps == 16
    d = np.abs(host(polA.params) - po)
    moved = np.abs(po - p0)
    # Adam normalises the step to ~lr per parameter, so parameters whose gradient is a near-cancelling
    # sum may differ by O(lr); the bulk must agree tightly.
    assert np.quantile(d, 0.99) < 2e-4, f"99th percentile |dp| = {np.quantile(d, 0.99):.2e}"
    assert d.max() < 16 * 2 * 1e-3
    assert moved.max() > 1e-3  # something was learned at all


def test_ppo_learns_cartpole(rl):
    """End-to-end sanity of the whole GPU path: mean episode length rises within a few dozen updates."""
    n, T = 1024, 32
    env = rl.HipVecEnv("cartpole", n, seed=1)
    pol = rl.PPOPolicy(env, update_freq=T, lr=3e-3)
    first = None
    for it in range(40):
        pol.rollout_()
        pol.update_()
        ep_len = (n * T) / max(1.0, float(pol.trajectory.terminal.sum()))
        if it == 0:
            first = ep_len
    assert ep_len > 2.0 * first, f"episode length {first:.1f} -> {ep_len:.1f}"


# ------------------------------------------------------------------------------------------ DQN
def _fill_ring(rl, n_env, capacity, seed=0):
    from rlhip.trajectory import CircularArraySARTSTraces

    rng = np.random.default_rng(seed)
    tr = CircularArraySARTSTraces(capacity=capacity, n_env=n_env, obs_dim=4)
    ref = oracle.Ring(capacity, n_env, 4)
    o = rng.standard_normal((4, n_env)).astype(np.float32)
    tr.push_state_(dev(o))
    ref.push_state(o)
    for _ in range(capacity + 3):
        o = rng.standard_normal((4, n_env)).astype(np.float32)
        a = rng.integers(0, 2, n_env).astype(np.int32)
        r = rng.standard_normal(n_env).astype(np.float32)
        t = (rng.random(n_env) < 0.1).astype(np.uint8)
        tr.push_transition_(dev(o), dev(a), dev(r), dev(t))
        ref.push_transition(o, a, r, t)
    return tr, ref


@pytest.mark.parametrize("h,batch", [(128, 32), (128, 512), (256, 4096), (64, 100)])
def test_dqn_gradient_vs_oracle(rl, h, batch):
    from rlhip.dqn import dqn_grad

    tr, ref = _fill_ring(rl, 64, 9)
    rng = np.random.default_rng(2)
    ns, na = 4, 2
    p = (oracle.mlp2_init(ns, h, na, 5, 0) + rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.1).astype(np.float32)
    pt = (p + rng.standard_normal(p.size) * 0.05).astype(np.float32)
    grad, loss = dqn_grad(tr, h, na, 0, dev(p), dev(pt), batch, 0.99, 1.0, seed=11, draw_ctr=4)
    idx = ref.sample_indices(batch, 11, 4)
    s, a, r, t, sn = ref.gather(idx)
    ol, og = oracle.dqn_loss_grad(ns, h, na, 0, p, pt, s, a, r, t, sn, 0.99, 1.0)
    assert float(loss) == pytest.approx(ol, rel=1e-4)
    assert_grad_close(host(grad), og, F32_GRAD_TOL, f"dqn_grad h={h} batch={batch}")


@pytest.mark.parametrize("h", [128, 100])
def test_dqn_plan_vs_oracle(rl, h):
    from rlhip.dqn import dqn_plan

    rng = np.random.default_rng(3)
    n, ns, na = 4096, 4, 2
    p = (oracle.mlp2_init(ns, h, na, 5, 0) + rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.1).astype(np.float32)
    obs = rng.standard_normal((ns, n)).astype(np.float32)
    for eps in (0.0, 0.3):
        a, q = dqn_plan(dev(p), ns, h, na, 0, dev(obs), eps, seed=8, env_id_base=100, step=42)
        oq = oracle.mlp2_forward(p, ns, h, na, 0, obs)
        np.testing.assert_allclose(host(q), oq, rtol=2e-5, atol=2e-6)
        # integer selection is bit-exact given the same Q-values and the same Philox draws
        oa = oracle.eps_greedy_select(host(q), eps, seed=8, step=42, env_id_base=100)
        assert np.array_equal(host(a), oa)


def test_device_counter_path_and_graph_replay_equal_eager(rl):
    """rollout_() + update_() (host counters) == iteration_dc_() (device counters) == HIP-graph replay."""
    n, T = 512, 8
    pols = []
    for _ in range(3):
        env = rl.CartPoleEnv(n, seed=9)
        pols.append(rl.PPOPolicy(env, update_freq=T))
    a, b, c = pols
    for _ in range(2):  # two eager iterations first so that the counters are non-zero at capture time
        for p in pols:
            p.rollout_()
            p.update_()
    b.sync_counters_()
    c.capture_graph_(warmup=0)
    for it in range(3):
        a.rollout_()
        a.update_()
        b.iteration_dc_()
        c.replay_()
        torch.cuda.synchronize()
        assert torch.equal(a.params, b.params), f"device-counter path differs at iteration {it}"
        assert torch.equal(a.params, c.params), f"graph replay differs at iteration {it}"
        assert torch.equal(a.trajectory.action_i, c.trajectory.action_i)
        assert a.vec_step == b.vec_step == c.vec_step and a.update_ctr == c.update_ctr
    assert c.counters.tolist() == [c.vec_step, c.update_ctr]


def test_single_rank_nccl_group_graph_capture(rl):
    """The multi-GPU update sequence (grad -> all-reduce -> clip+Adam) with a 1-rank RCCL group, eager and
    captured in a HIP graph: must equal the fused single-GPU update."""
    import os

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n, T = 256, 8
        envs = [rl.CartPoleEnv(n, seed=3) for _ in range(3)]
        ref = rl.PPOPolicy(envs[0], update_freq=T)
        eager = rl.PPOPolicy(envs[1], update_freq=T, process_group=dist.group.WORLD)
        graph = rl.PPOPolicy(envs[2], update_freq=T, process_group=dist.group.WORLD)
        for p in (eager, graph):
            p._force_dist = True  # take the distributed branch although world == 1
        graph.capture_graph_(warmup=1)
        for p in (ref, eager):
            for _ in range(2):  # the graph policy did one warm-up iteration before the capture
                p.rollout_()
                p.update_()
        graph.replay_()
        torch.cuda.synchronize()
        assert torch.equal(ref.params, eager.params)
        assert torch.equal(ref.params, graph.params)
    finally:
        if created:
            dist.destroy_process_group()


def test_dqn_grad_on_explicit_indices_and_prioritized_learner():
    """rlhip_dqn_grad_idx_f32: the 2-layer DQN gradient on explicit (prioritized) indices equals the inline-draw
    variant on the same indices, returns the TD errors, and the prioritized learner writes priorities back."""
    import ctypes as C

    import rlhip
    from rlhip import dqn, ops
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    n, ns, h, na, batch = 128, 4, 64, 2, 300
    tr = rlhip.CircularArraySARTSTraces(capacity=32, n_env=n, obs_dim=ns)
    tr.records.normal_()  # every word of every 64-byte record: s, s_next and (overwritten below) a, r, t
    tr.action.random_(0, na)
    tr.reward.normal_()
    tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.1).to(torch.uint8))
    tr.rb.len_sa, tr.rb.len_rt = 33, 32
    p, tp = ops.mlp2_init(ns, h, na, 1, 0), ops.mlp2_init(ns, h, na, 2, 0)
    g0, l0 = dqn.dqn_grad(tr, h, na, 0, p, tp, batch, 0.99, 1.0, 9, 4)
    idx = tr.sample_indices(batch, 9, 4)
    ws = dqn.dqn_workspace(ns, h, na, batch)
    g1, l1, td = torch.empty_like(p), torch.empty(1, device="cuda"), torch.zeros(batch, device="cuda")
    call("rlhip_dqn_grad_idx_f32", C.byref(tr.rb), h, na, 0, ptr(p), ptr(tp), batch, ptr(idx), 0.99, 1.0, ptr(ws),
         ptr(g1), ptr(l1), ptr(td), stream_ptr())
    assert torch.equal(g0, g1) and torch.equal(l0, l1)
    s, a, r, t, sn = tr.gather(idx)
    q = ops.mlp2_forward(p, ns, h, na, 0, s)
    qn = ops.mlp2_forward(tp, ns, h, na, 0, sn)
    y = r + 0.99 * (1 - t.float()) * qn.max(0).values
    ref = (q.gather(0, a.long()[None])[0] - y).abs()
    torch.testing.assert_close(td, ref, rtol=1e-5, atol=1e-6)
    # the learner with prioritized traces
    env = rlhip.CartPoleEnv(n, seed=2)
    net = rlhip.HipApproximator(4, 64, 2, seed=2)
    learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=4), batchsize=64, min_replay_history=n, seed=2)
    policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.1, seed=2))
    traces = rlhip.CircularPrioritizedTraces(capacity=16, n_env=n, obs_dim=4, default_priority=5.0)
    rlhip.run(rlhip.Agent(policy, rlhip.Trajectory(traces)), env, rlhip.StopAfterNSteps(20))
    torch.cuda.synchronize()
    assert learner.n_updates >= 15 and torch.isfinite(net.params).all()
    leaves = traces.priorities[traces.priorities.numel() // 2:][:traces.n_leaves]
    assert (leaves != 5.0).any() and (leaves >= 0).all()


@pytest.mark.parametrize("kind,layers,hidden", [("cartpole", 2, 256), ("pendulum", 2, 64), ("mountaincar", 2, 100),
                                                ("cartpole", 3, 128), ("pendulum", 3, 128)])
def test_rollout_writes_gae_and_returns_bit_identical_to_the_scan(kind, layers, hidden):
    """the rollout kernels end with the GAE + returns scan of every env (gae_device.h): adv / ret after rollout_()
    equal what rlhip_ppo_gae_f32 computes from the same trajectory, bit for bit (wide, scalar and MFMA rollouts)"""
    import rlhip

    env = rlhip.HipVecEnv(kind, 333, seed=3)
    pol = rlhip.PPOPolicy(env, update_freq=37, hidden=hidden, seed=3, layers=layers)
    pol.rollout_()
    adv, ret = pol.trajectory.adv.clone(), pol.trajectory.ret.clone()
    assert adv.abs().sum() > 0
    pol.trajectory.adv.zero_()
    pol.trajectory.ret.zero_()
    pol.gae_()
    assert torch.equal(adv, pol.trajectory.adv) and torch.equal(ret, pol.trajectory.ret)


@pytest.mark.parametrize("ns,h,na,clip,batch", [(4, 64, 2, 0.5, 700), (4, 64, 2, 0.0, 700), (2, 100, 3, 1e6, 700), (3, 256, 3, 0.05, 700),
                                                 (4, 252, 4, 0.5, 700), (4, 256, 4, 0.5, 700), (4, 128, 2, 0.5, 32), (4, 128, 2, 0.5, 512),
                                                 (4, 128, 2, 0.5, 2048), (4, 256, 4, 0.5, 2048), (4, 128, 2, 0.5, 2049), (4, 128, 2, 0.5, 4096),
                                                 (3, 256, 3, 0.5, 40000), (2, 8, 1, 0.5, 100), (4, 200, 2, 0.5, 300)])
def test_dqn_update_is_bit_identical_to_grad_then_clip_adam(ns, h, na, clip, batch):
    """rlhip_dqn_update_f32 == rlhip_dqn_grad_f32 followed by rlhip_clip_adam_f32, bit for bit, over repeated calls (the departure
    counter re-arms itself).  Up to 32 tiles (2048 samples) the whole optimise! is ONE launch (dqn_grad_kernel<..., FUSE>: the
    workgroup that departs last folds the partial rows, clips, steps); beyond that the tail is dqn_reduce_apply_kernel."""
    import rlhip
    from rlhip import dqn, ops

    n = 64
    tr = rlhip.CircularArraySARTSTraces(capacity=32, n_env=n, obs_dim=ns)
    tr.records.normal_()  # every word of every 64-byte record: s, s_next and (overwritten below) a, r, t
    tr.action.random_(0, na)
    tr.reward.normal_()
    tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.1).to(torch.uint8))
    tr.rb.len_sa, tr.rb.len_rt = 33, 32
    tp = ops.mlp2_init(ns, h, na, 2, 0)
    st = []
    for _ in range(2):
        p = ops.mlp2_init(ns, h, na, 1, 0)
        st.append(dict(p=p, m=torch.zeros_like(p), v=torch.zeros_like(p), g=torch.empty_like(p),
                       bp=torch.tensor([0.9, 0.999], device="cuda"), loss=torch.empty(1, device="cuda"),
                       gn=torch.zeros(1, device="cuda"), ws=dqn.dqn_workspace(ns, h, na, batch)))
    a, b = st
    for it in range(5):
        dqn.dqn_grad(tr, h, na, it % 2, a["p"], tp, batch, 0.99, 1.0, 7, it, a["ws"], a["g"], a["loss"])
        ops.clip_adam_(a["p"], a["g"], a["m"], a["v"], a["bp"], 0.5, clip, 1e-2, 0.9, 0.999, 1e-8, a["gn"])
        dqn.dqn_update(tr, h, na, it % 2, b["p"], tp, batch, 0.99, 1.0, 7, it, b["ws"], b["g"], b["loss"], b["m"], b["v"],
                       b["bp"], 0.5, clip, 1e-2, 0.9, 0.999, 1e-8, b["gn"])
        for k in ("p", "m", "v", "g", "bp", "loss", "gn"):
            assert torch.equal(a[k], b[k]), (it, k)
    assert not torch.equal(a["p"], ops.mlp2_init(ns, h, na, 1, 0))


@pytest.mark.parametrize("kind,continuous,hidden,act", [("cartpole", False, 256, 0), ("pendulum", True, 256, 1),
                                                        ("mountaincar", False, 128, 0), ("pendulum", False, 64, 1)])
def test_two_layer_ppo_gradient_is_bit_deterministic_run_to_run(rl, kind, continuous, hidden, act):
    """The two-layer learner tile (csrc/ppo_grad_tile.h): layer 1 on the f32 MFMA, head sums by (wave, lane half) through
    LDS, weight gradients in registers -- fixed summation order, no atomics: eight launches on two alternating micro-batches
    must agree bit for bit (round 2's run-to-run sighting sat beside MFMAs at two waves per SIMD)."""
    n, T = 2048, 16
    env, pol, _, _ = make_pair(rl, kind, n, T, continuous=continuous, hidden=hidden, act=act)
    pol.rollout_()
    pol.gae_()
    runs = []
    for rep in range(8):
        pol.grad_(rep & 1, 1 + (rep & 1))
        torch.cuda.synchronize()
        runs.append((host(pol.grad).copy(), host(pol.losses).copy()))
    for rep in range(2, 8):
        assert np.array_equal(runs[rep][0], runs[rep & 1][0]), "gradient differs run to run"
        assert np.array_equal(runs[rep][1], runs[rep & 1][1]), "losses differ run to run"
    assert not np.array_equal(runs[0][0], runs[1][0])
