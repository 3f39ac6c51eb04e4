"""GPU-vs-oracle parity at the shapes `bench.py` TIMES ("next round" item 1).

The other learner suites compare with the oracle at sizes the oracle finishes in milliseconds (<= 1024-sample
micro-batches, <= 1000 envs).  The launches behind the numbers of the bench line are different launches:

  headline  BASELINE configs[3] per GPU -- 4096 CartPole envs x T = 32, 4 -> 256 -> {2, 1}: `rollout_split_kernel` over 4096 envs
            (256 workgroups), `ppo_grad_kernel<4,0,2,2>` on 32768-sample micro-batches (256 workgroups, 256 partial rows into
            `reduce_apply_kernel`), 16 optimiser steps per `update_()`;
  config 3  BASELINE configs[2], Float32 two-layer nets -- 4096 Pendulum envs x T = 128, clip 0.1, micro-batches of 131072:
            1024 team-pairs > MAX_GRAD_BLOCKS, i.e. the multi-trip persistent loop of `ppo_grad_kernel<3,...>`;
  config 2  BASELINE configs[1] -- DQN batch 4096 drawn from a replay ring of 2^16 slots x 4096 envs that has wrapped around.

Every comparison here is against `oracle/` (never GPU against GPU), with the seeds and constructor arguments of bench.py.
Measured margins are appended to gpurun_out/bench_shape_margins.jsonl (copied to profiles/r05_parity_margins.md)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from conftest import F32_GRAD_TOL, ROOT, SESSION_ID, assert_grad_close  # noqa: E402

MARGINS = os.path.join(ROOT, "gpurun_out", "bench_shape_margins.jsonl")


def note(tag, **kw):
    try:
        os.makedirs(os.path.dirname(MARGINS), exist_ok=True)
        with open(MARGINS, "a") as f:
            f.write(json.dumps({"session": SESSION_ID, "tag": tag, **kw}) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def rl():
    import rlhip

    oracle.use_all_cores(True)
    yield rlhip
    oracle.use_all_cores(False)


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def host(t):
    return t.detach().cpu().numpy()


def dev_stats(x, o, atol):
    """(max, q99.9) of |x - o| / (atol + |o|)"""
    x, o = np.asarray(x, np.float64), np.asarray(o, np.float64)
    d = np.abs(x - o) / (atol + np.abs(o))
    return (float(d.max()), float(np.quantile(d, 0.999))) if d.size else (0.0, 0.0)


def compare_rollout(tag, env, pol, oenv, ocfg, T, continuous, max_flipped, tol_rel, tol_abs):
    """Free-running rollout of the bench's policy against oracle.ppo_rollout.

    Discrete heads: an env whose Gumbel-max draw sits within an ulp of a tie may take the other action (the two sides sum the 256
    hidden units in different orders); from there on that ONE env legitimately diverges, every other env is compared entry by
    entry under `tol_rel`.

    Continuous heads: the actions differ in the last bits from step 0 on and the env amplifies that step after step (SURVEY A.7:
    the open-loop pendulum multiplies a perturbation by ~e^(4 t / s)), so a free-running comparison over T = 128 steps is a
    statement about the DISTRIBUTION of the deviation: the horizon of tests/test_gpu_learners.py::test_rollout_vs_oracle (24
    steps) under its bar, 99.9 % of ALL entries of every trace under the same bar, and the single worst entry bounded at 50 x
    (measured: see gpurun_out/bench_shape_margins.jsonl -> profiles/r05_parity_margins.md).  Terminal flags are exact."""
    n = env.n
    p = host(pol.params)
    pol.rollout_()
    otr = oracle.PPOTraj(env.kind, n, T, na=1, continuous=continuous)
    oracle.ppo_rollout(oenv, T, ocfg, p, otr, 0)
    tr = pol.trajectory
    at = tol_abs / tol_rel
    if not continuous:
        flipped = host(tr.action_i) != otr.action_i
        first_flip = np.where(flipped.any(0), flipped.argmax(0), T)
        n_flip = int((first_flip < T).sum())
        assert n_flip <= max_flipped, f"{n_flip} of {n} envs saw a flipped action"
        before = np.arange(T)[:, None] < first_flip[None, :]
        upto = np.arange(T + 1)[:, None] <= first_flip[None, :]
        assert np.array_equal(host(tr.action_i)[before], otr.action_i[before])
    else:
        n_flip = 0
        before = np.ones((T, n), bool)
        upto = np.ones((T + 1, n), bool)
    assert np.array_equal(host(tr.terminal)[before], otr.terminal[before])
    om = np.broadcast_to(upto[:, None, :], otr.obs.shape)
    stats = {"obs": dev_stats(host(tr.obs)[om], otr.obs[om], at), "value": dev_stats(host(tr.value)[upto], otr.value[upto], at),
             "reward": dev_stats(host(tr.reward)[before], otr.reward[before], at),
             "logp": dev_stats(host(tr.logp)[before], otr.logp[before], 1e-2)}
    if continuous:
        stats["action"] = dev_stats(host(tr.action_f), otr.action_f, at)
        stats["action_first_24_steps"] = dev_stats(host(tr.action_f)[:24], otr.action_f[:24], at)
    note(tag, envs=n, T=T, flipped_envs=n_flip, compared=int(before.sum()), bar=tol_rel,
         **{k: {"max": v[0], "q999": v[1]} for k, v in stats.items()})
    for k, (mx, q999) in stats.items():
        if continuous and k != "action_first_24_steps":
            assert q999 <= tol_rel and mx <= 50 * tol_rel, (k, mx, q999)
        else:  # discrete: every compared entry; continuous: the first 24 steps
            assert mx <= (1e-4 if k == "logp" and not continuous else tol_rel), (k, mx)
    # the scan the rollout launch fuses: bit-exact against the oracle's scan of the GPU's own traces
    o = oracle.generalized_advantage_estimation(host(tr.reward).T, host(tr.value).T, pol.cfg.gamma, pol.cfg.lam,
                                                terminal=host(tr.terminal).T, dims=2, dtype=np.float32)
    assert np.array_equal(host(tr.adv), o.T)
    assert np.array_equal(host(tr.ret), (o.T + host(tr.value)[:T]).astype(np.float32))
    return otr


def oracle_microbatch(pol, tr, epoch_ctr, mb):
    n, T = tr.n, tr.T
    total = n * T
    bm = total // pol.cfg.n_microbatches
    perm = np.array([oracle.permute(pol.seed, epoch_ctr, total, mb * bm + b) for b in range(bm)], dtype=np.int64)
    t, i = perm // n, perm % n
    obs = host(tr.obs)[t, :, i].T.copy()
    flat = lambda x: host(x).reshape(-1)[perm]  # noqa: E731
    act = host(tr.action_f)[t, 0, i] if tr.continuous else host(tr.action_i).reshape(-1)[perm]
    return obs, act, flat(tr.logp), flat(tr.adv), flat(tr.ret)


def copy_traj_to_oracle(kind, tr, continuous):
    otr = oracle.PPOTraj(kind, tr.n, tr.T, na=1, continuous=continuous)
    for name in ("obs", "logp", "value", "reward", "terminal") + (("action_f",) if continuous else ("action_i",)):
        getattr(otr, name)[...] = host(getattr(tr, name))
    return otr


def compare_update(tag, kind, pol, ocfg, continuous, n_steps):
    """one whole `update_()` (n_epochs x n_microbatches optimiser steps incl. clip + Adam) against oracle.ppo_update run on
    the GPU's own trajectory.  Adam normalises a step to ~lr per parameter, so a parameter whose gradient is a near-cancelling
    sum may move differently by O(lr) per step; the bulk must agree tightly."""
    tr = pol.trajectory
    p0 = host(pol.params).copy()
    otr = copy_traj_to_oracle(kind, tr, continuous)
    oracle.ppo_gae(ocfg, otr)
    assert np.array_equal(otr.adv, host(tr.adv))
    pol.update_()
    po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    steps, _ = oracle.ppo_update(kind, ocfg, otr, po, mo, vo, 0, pol.seed, 0)
    assert steps == n_steps
    d = np.abs(host(pol.params) - po)
    moved = np.abs(po - p0)
    q99, dmax = float(np.quantile(d, 0.99)), float(d.max())
    note(tag, steps=steps, params=int(p0.size), dp_q50=float(np.median(d)), dp_q99=q99, dp_max=dmax,
         moved_q50=float(np.median(moved)), lr=float(pol.cfg.lr))
    lr = float(pol.cfg.lr)
    assert float(np.median(moved)) > 0.5 * lr, "the update did not move the parameters"
    assert q99 < 0.2 * lr, f"99th percentile |dp| = {q99:.2e} (lr {lr:.0e})"
    assert dmax < n_steps * 2 * lr
    assert np.allclose(host(pol.m), mo, rtol=0, atol=1e-3 * max(1e-30, float(np.abs(mo).max())))


# ------------------------------------------------------------------------------------------ headline
def make_headline(rl):
    """bench.py main(): HipVecEnv("cartpole", 4096, seed = 123), PPOPolicy(update_freq = 32, hidden = 256, seed = 123)"""
    n, T = 4096, 32
    env = rl.HipVecEnv("cartpole", n, seed=123, env_id_base=0)
    pol = rl.PPOPolicy(env, update_freq=T, hidden=256, seed=123)
    oenv = oracle.VecEnv("cartpole", n, seed=123, continuous=False)
    ocfg = oracle.ppo_default(continuous=0, hidden=256)
    return n, T, env, pol, oenv, ocfg


def test_headline_rollout_4096x32_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_headline(rl)
    assert pol.np == 3331 and pol.cfg.n_epochs == 4 and pol.cfg.n_microbatches == 4
    compare_rollout("headline rollout 4096x32", env, pol, oenv, ocfg, T, False, max_flipped=8, tol_rel=1e-5, tol_abs=1e-6)


def test_headline_gradient_32768_sample_microbatches_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_headline(rl)
    rng = np.random.default_rng(1)
    # first at the bench's own operating point: the behaviour policy itself (ratio = 1 everywhere) ...
    pol.rollout_()
    tr = pol.trajectory
    p = host(pol.params)
    for epoch_ctr, mb in ((0, 0), (3, 2)):
        pol.grad_(epoch_ctr, mb)
        obs, a, lp, adv, ret = oracle_microbatch(pol, tr, epoch_ctr, mb)
        assert obs.shape[1] == 32768
        g, losses = oracle.ppo_loss_grad(ocfg, 4, 2, p, obs, a, lp, adv, ret)
        assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, f"bench-shape ppo_grad headline on-policy mb={mb}")
        np.testing.assert_allclose(host(pol.losses), losses, rtol=1e-4, atol=1e-6)
    # ... then moved away from it, so that ratios leave the clip range and biases are non-zero
    p2 = (p + rng.standard_normal(pol.np) * 0.03).astype(np.float32)
    pol.params.copy_(dev(p2))
    for epoch_ctr, mb in ((1, 1), (2, 3), (7, 0)):
        pol.grad_(epoch_ctr, mb)
        obs, a, lp, adv, ret = oracle_microbatch(pol, tr, epoch_ctr, mb)
        g, losses = oracle.ppo_loss_grad(ocfg, 4, 2, p2, obs, a, lp, adv, ret)
        assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, f"bench-shape ppo_grad headline off-policy mb={mb}")
        np.testing.assert_allclose(host(pol.losses), losses, rtol=1e-4, atol=1e-6)


def test_headline_update_16_steps_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_headline(rl)
    pol.rollout_()
    compare_update("headline update 16 x 32768", 0, pol, ocfg, False, 16)
    # and a second iteration on top (the bench's steady state: non-zero Adam moments, records re-packed)
    pol.rollout_()
    tr = pol.trajectory
    p = host(pol.params)
    pol.grad_(0, 1)
    obs, a, lp, adv, ret = oracle_microbatch(pol, tr, 0, 1)
    g, losses = oracle.ppo_loss_grad(ocfg, 4, 2, p, obs, a, lp, adv, ret)
    assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, "bench-shape ppo_grad headline second iteration")


# ------------------------------------------------------------------------------------------ config 3 (Float32 two-layer nets)
def make_config3(rl):
    """bench.py roofline_extras(): HipVecEnv("pendulum", 4096, seed = 7), PPOPolicy(update_freq = 128, hidden = 256, seed = 7,
    clip_range = 0.1)"""
    n, T = 4096, 128
    env = rl.HipVecEnv("pendulum", n, seed=7)
    pol = rl.PPOPolicy(env, update_freq=T, hidden=256, seed=7, clip_range=0.1)
    oenv = oracle.VecEnv("pendulum", n, seed=7, continuous=env.continuous)
    ocfg = oracle.ppo_default(continuous=int(env.continuous), hidden=256, clip_range=0.1)
    return n, T, env, pol, oenv, ocfg


def test_config3_rollout_4096x128_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_config3(rl)
    # continuous actions differ in the last bits (summation order) and Pendulum amplifies them over 128 steps: drift bar as in
    # tests/test_gpu_learners.py::test_rollout_vs_oracle, measured margin in the log
    compare_rollout("config3 rollout 4096x128", env, pol, oenv, ocfg, T, bool(env.continuous), max_flipped=8, tol_rel=2e-3,
                    tol_abs=2e-3)


def test_config3_gradient_131072_sample_microbatches_multi_trip_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_config3(rl)
    cont = bool(env.continuous)
    rng = np.random.default_rng(5)
    pol.rollout_()
    tr = pol.trajectory
    p = host(pol.params)
    p2 = (p + rng.standard_normal(pol.np) * 0.02).astype(np.float32)
    for params, cases in ((p, ((0, 0),)), (p2, ((1, 3), (6, 2)))):
        pol.params.copy_(dev(params))
        for epoch_ctr, mb in cases:
            pol.grad_(epoch_ctr, mb)
            obs, a, lp, adv, ret = oracle_microbatch(pol, tr, epoch_ctr, mb)
            assert obs.shape[1] == 131072
            g, losses = oracle.ppo_loss_grad(ocfg, 3, pol.na, params, obs, a, lp, adv, ret)
            assert_grad_close(host(pol.grad), g, F32_GRAD_TOL, f"bench-shape ppo_grad config3 f32 epoch={epoch_ctr} mb={mb}")
            np.testing.assert_allclose(host(pol.losses), losses, rtol=1e-4, atol=1e-6)
    assert cont


def test_config3_update_16_steps_of_131072_vs_oracle(rl):
    n, T, env, pol, oenv, ocfg = make_config3(rl)
    pol.rollout_()
    compare_update("config3 f32 update 16 x 131072", 1, pol, ocfg, True, 16)


# ------------------------------------------------------------------------------------------ config 2 (DQN, wrapped 2^16-slot ring)
def test_config2_dqn_batch_4096_from_a_wrapped_ring_of_65536_slots_x_4096_envs(rl):
    """BASELINE configs[1] at replay size: a CircularArraySARTSTraces of 2^16 slots x 4096 envs (2.7e8 transitions, 17 GB of
    64-byte records) filled THROUGH the push ABI past its wrap-around, then (i) the BatchSampler draw + gather of a 4096-sample
    batch bit-exact against the oracle's sampler arithmetic and the pushed content, (ii) the two-layer DQN gradient on that
    batch (the inline-draw kernel: it samples and gathers from the ring itself) against oracle.dqn_loss_grad.

    The content of push p is a function of (p, env) the host can evaluate for any p -- frame p mod 61 of a random pool (61 and
    the 65537 state slots are coprime) with component 0 overwritten by p / 2^16 (exact in Float32) -- so no host copy of the
    ring is needed and a gather from a wrong slot cannot alias a right one."""
    from rlhip.dqn import dqn_grad
    from rlhip.trajectory import CircularArraySARTSTraces

    cap, n, ns, na, h, batch, extra, pool = 1 << 16, 4096, 4, 2, 128, 4096, 777, 61
    rng = np.random.default_rng(21)
    P_obs = rng.standard_normal((pool, ns, n)).astype(np.float32)
    P_a = rng.integers(0, na, (pool, n)).astype(np.int32)
    P_r = rng.standard_normal((pool, n)).astype(np.float32)
    P_t = (rng.random((pool, n)) < 0.05).astype(np.uint8)
    d_obs, d_a, d_r, d_t = dev(P_obs), dev(P_a), dev(P_r), dev(P_t)
    tr = CircularArraySARTSTraces(capacity=cap, n_env=n, obs_dim=ns)
    assert tr.records_layout and tr.records.numel() * 4 == (cap + 1) * n * 64
    frame = torch.empty((ns, n), dtype=torch.float32, device="cuda")
    n_push = cap + extra  # transitions pushed; push 0 is the PreEpisodeStage state
    for p in range(n_push + 1):
        frame.copy_(d_obs[p % pool])
        frame[0].fill_(p / 65536.0)  # exact in Float32 (p < 2^17)
        if p == 0:
            tr.push_state_(frame)
        else:
            tr.push_transition_(frame, d_a[p % pool], d_r[p % pool], d_t[p % pool])
    torch.cuda.synchronize()
    assert len(tr) == cap and tr.rb.len_sa == cap + 1 and tr.rb.head_rt == extra % cap and tr.rb.head_sa == extra % (cap + 1)

    def expected(idx):
        li, e = idx // n, idx % n
        q = (n_push - cap) + li  # transition q: state of push q -> state of push q + 1, with (a, r, t) of push q + 1
        s = P_obs[q % pool, :, e].T.copy()
        sn = P_obs[(q + 1) % pool, :, e].T.copy()
        s[0], sn[0] = (q / 65536.0).astype(np.float32), ((q + 1) / 65536.0).astype(np.float32)
        return s, P_a[(q + 1) % pool, e], P_r[(q + 1) % pool, e], P_t[(q + 1) % pool, e], sn

    orb = oracle.RingC()
    orb.capacity, orb.n_env, orb.obs_dim, orb.len_rt = cap, n, ns, cap
    for ctr in (0, 3):
        oidx = np.empty(batch, np.int64)
        oracle.lib().rlo_ring_sample_indices(C.byref(orb), C.c_int64(batch), C.c_uint64(11), C.c_uint32(ctr),
                                             oidx.ctypes.data_as(C.c_void_p))
        idx = tr.sample_indices(batch, seed=11, draw_ctr=ctr)
        assert np.array_equal(host(idx), oidx)
        assert oidx.max() > 0.99 * cap * n and (oidx // n < extra).any()  # the draw reaches both ends of the wrapped ring
        got = [host(x) for x in tr.gather(idx)]
        for g, o, name in zip(got, expected(oidx), ("state", "action", "reward", "terminal", "next_state")):
            assert np.array_equal(g, o), f"gathered {name} differs (draw {ctr})"
    # the learner's own draw + gather + gradient, bench network 4 -> 128 -> 2
    p = (oracle.mlp2_init(ns, h, na, 5, 0) + rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.1).astype(np.float32)
    pt = (p + rng.standard_normal(p.size) * 0.05).astype(np.float32)
    dp, dpt = dev(p), dev(pt)
    grad, loss = dqn_grad(tr, h, na, 0, dp, dpt, batch, 0.99, 1.0, seed=11, draw_ctr=3)
    s, a, r, t, sn = expected(oidx)
    ol, og = oracle.dqn_loss_grad(ns, h, na, 0, p, pt, s, a, r, t, sn, 0.99, 1.0)
    assert float(loss) == pytest.approx(ol, rel=1e-4)
    assert_grad_close(host(grad), og, F32_GRAD_TOL, "bench-shape dqn_grad batch 4096, wrapped 2^16 x 4096 ring")
    note("config2 dqn ring 65536 x 4096", pushes=n_push, batch=batch, ring_gb=round(tr.records.numel() * 4 / 1e9, 2))
