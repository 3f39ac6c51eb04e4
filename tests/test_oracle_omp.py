"""The -fopenmp build of the oracle (bench.py's all-cores cpu_baseline) against the sequential parity oracle:
rollouts are bit-identical (env instances are independent), the learner update agrees to Float64-accumulation
round-off (per-thread gradient accumulators combined in thread order)."""
import numpy as np

import oracle


def _iteration(n=96, T=8, hidden=32):
    env = oracle.VecEnv("cartpole", n, seed=3)
    cfg = oracle.ppo_default(hidden=hidden)
    params = np.concatenate([oracle.mlp2_init(4, hidden, 2, 3, 0), oracle.mlp2_init(4, hidden, 1, 3, 1)])
    m, v = np.zeros_like(params), np.zeros_like(params)
    traj = oracle.PPOTraj(0, n, T)
    oracle.ppo_rollout(env, T, cfg, params, traj, 0)
    oracle.ppo_gae(cfg, traj)
    oracle.ppo_update(0, cfg, traj, params, m, v, 0, 3, 0)
    return traj, params


def test_openmp_build_matches_sequential_oracle():
    t0, p0 = _iteration()
    threads = oracle.use_all_cores(True)
    try:
        t1, p1 = _iteration()
    finally:
        assert oracle.use_all_cores(False) == 1
    assert threads >= 1
    for name in ("obs", "action_i", "logp", "value", "reward", "terminal", "adv", "ret"):
        assert np.array_equal(getattr(t0, name), getattr(t1, name)), name
    np.testing.assert_allclose(p1, p0, rtol=1e-5, atol=1e-7)
