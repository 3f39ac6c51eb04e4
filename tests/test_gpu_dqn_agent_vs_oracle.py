"""The fused DQN vec-step (`rlhip_dqn_vec_step_f32`: plan! + act! + push! in one launch, the whole optimise! in a second) against
the ORACLE's per-stage agent loop (`oracle.dqn_run`: run.jl:52-67 + agent_base.jl:45-59 sequenced from the oracle's C functions)
over 230 vec-steps of BASELINE configs[1] -- 4096 CartPole envs, QBasedPolicy(DQN), eps-greedy exp (eps_stable 0.01, decay 500),
gamma 0.99, Huber 1, Adam 1e-3, batch 512, one update per vec-step, hard target sync every 100 updates (two syncs inside the run).

Until round 6 the fused call was compared with the GPU's own per-stage path only (tests/test_gpu_run.py); VERDICT r5 missing item 5.

What can be asked for: integer actions are bit-exact until an env's greedy decision sits within the two sides' Q difference of a
tie (the parameters differ in their last bits after the first Adam step: Float64 vs fixed-order Float32 batch sums); from that
vec-step on THAT env is a different (equally valid) trajectory -- CartPole is chaotic -- and is excluded; every other env must
match entry by entry for the whole run: actions / rewards / terminal flags exactly, states within 1e-5 (measured in the log).
Parameters: the bars of tests/test_gpu_bench_shapes.py::compare_update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from test_gpu_bench_shapes import host, note  # noqa: E402


def test_fused_dqn_vec_step_vs_oracle_agent_loop(layers=2, max_flipped=8):
    """free-running, Float32 two-layer Q-network (the bf16 three-layer one: teacher-forced below -- its Q values differ from the
    oracle's by ~1e-5 relative from the first step on, and a free run of 4096 chaotic envs x 230 steps turns that into hundreds
    of legitimately different trajectories: measured 817 envs, profiles/r06_parity_margins.md)"""
    import rlhip as rl

    n, cap, K, batch, h = 4096, 256, 230, 512, 128
    env = rl.CartPoleEnv(n, seed=5)
    net = rl.HipApproximator(4, h, 2, seed=5, layers=layers)
    tn = rl.TargetNetwork(net, sync_freq=100)
    learner = rl.DQNLearner(tn, batchsize=batch, min_replay_history=n, seed=5)
    explorer = rl.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5)
    policy = rl.QBasedPolicy(learner, explorer)
    traces = rl.CircularArraySARTSTraces(capacity=cap, n_env=n, obs_dim=4)
    agent = rl.Agent(policy, rl.Trajectory(traces))
    p0 = host(net.params).copy()
    rl.run_fused_dqn(agent, env, rl.StopAfterNSteps(K))
    torch.cuda.synchronize()

    oracle.use_all_cores(True)
    try:
        o = oracle.dqn_run(K, n=n, hidden=h, env_seed=5, net_seed=5, explorer_seed=5, sampler_seed=5, capacity=cap, batch=batch,
                           sync_freq=100, layers=layers)
    finally:
        oracle.use_all_cores(False)
    # counters of the loop
    assert (learner.n_updates, learner.draw_ctr, explorer.step, tn.n_optimise) == (o.n_updates, o.draw_ctr, o.explorer_step, o.n_optimise)
    assert o.n_updates == K and o.n_optimise == K % 100
    for name in ("head_sa", "len_sa", "head_rt", "len_rt"):
        assert getattr(traces.rb, name) == getattr(o.ring.rb, name), name
    # every stored transition, in logical order, through both gathers
    idx = np.arange(K * n, dtype=np.int64)
    gs, ga, gr, gt, gsn = (host(x) for x in traces.gather(torch.as_tensor(idx).cuda()))
    os_, oa, or_, ot, osn = o.ring.gather(idx)
    ga, oa = ga.reshape(K, n), oa.reshape(K, n)
    assert np.array_equal(oa, np.stack(o.actions))
    flipped = ga != oa
    first = np.where(flipped.any(0), flipped.argmax(0), K)          # first vec-step at which an env's action differs
    n_flip = int((first < K).sum())
    upto = np.arange(K)[:, None] <= first[None, :]                   # transitions whose STATE both sides still share
    before = np.arange(K)[:, None] < first[None, :]                  # ... and whose action agrees too
    assert np.array_equal(ga[before], oa[before])
    assert n_flip <= max_flipped, f"{n_flip} of {n} envs took a different greedy action somewhere in {K} vec-steps"
    assert np.array_equal(gr.reshape(K, n)[before], or_.reshape(K, n)[before])
    assert np.array_equal(gt.reshape(K, n)[before], ot.reshape(K, n)[before])
    ds = np.abs(gs.reshape(4, K, n) - os_.reshape(4, K, n))[:, upto] / (1e-1 + np.abs(os_.reshape(4, K, n))[:, upto])
    dn = np.abs(gsn.reshape(4, K, n) - osn.reshape(4, K, n))[:, before] / (1e-1 + np.abs(osn.reshape(4, K, n))[:, before])
    assert ds.max() <= 1e-5 and dn.max() <= 1e-5, (ds.max(), dn.max())
    # parameters after 230 Adam steps and two hard target syncs
    p, po = host(net.params), o.params
    d, moved, lr = np.abs(p - po), np.abs(po - p0), 1e-3
    q99, dmax = float(np.quantile(d, 0.99)), float(d.max())
    td = np.abs(host(tn.target) - o.target)
    note(f"fused DQN vec-step vs oracle agent loop, layers={layers}", vec_steps=K, envs=n, flipped_envs=n_flip,
         first_flip=int(first.min()), state_err_max=float(ds.max()), next_state_err_max=float(dn.max()), dp_q50=float(np.median(d)),
         dp_q99=q99, dp_max=dmax, moved_q50=float(np.median(moved)), target_dp_max=float(td.max()),
         loss_gpu=float(learner.loss), loss_oracle=float(o.losses[-1]))
    assert float(np.median(moved)) > 5 * lr, "the run did not move the parameters"
    bar = 0.2 * lr if layers == 2 else 2.0 * lr      # bf16 hidden layer: every Adam step sees a 1e-3-relative different gradient
    assert q99 < bar, f"99th percentile |dp| = {q99:.2e} (lr {lr:.0e})"
    assert dmax < K * 2 * lr
    assert float(td.max()) <= dmax + 1e-12             # the target is a copy of the parameters of update 200
    assert abs(float(learner.loss) - o.losses[-1]) <= (2e-3 if layers == 2 else 2e-2) * max(1.0, abs(o.losses[-1]))


@pytest.mark.parametrize("layers", [2, 3])
def test_fused_dqn_vec_step_teacher_forced_vs_oracle(layers):
    """Step-by-step form of the comparison above for BOTH networks: before every vec-step the oracle receives the GPU's state
    (env arrays incl. step / reset counters, parameters, target, Adam moments), then both sides run ONE vec-step --
    `rlhip_dqn_vec_step_f32` against oracle.DQNRun.vec_step with the GPU's actions forced into act! / push!:
      plan!     the oracle's own decision equals the GPU's wherever its Q gap exceeds the forward tolerance (and the explore
                branch, which does not look at Q, always);
      act!      reward / terminal / step counters exact, states within 1e-6;
      push!     the two rings hold the same transitions at the end (wrapped: capacity 64 < 130 steps);
      optimise! parameters after the step (same draw from rings that agree, gradient, Adam, target sync every 50) under a per-step bar.
    No chaos is inherited from step to step, so the bars are those of single launches."""
    import rlhip as rl

    n, cap, K, batch, h, sync = 1024, 64, 130, 256, 128, 50
    env = rl.CartPoleEnv(n, seed=9)
    net = rl.HipApproximator(4, h, 2, seed=9, layers=layers)
    tn = rl.TargetNetwork(net, sync_freq=sync)
    learner = rl.DQNLearner(tn, batchsize=batch, min_replay_history=n, seed=9)
    explorer = rl.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=40, seed=9)
    policy = rl.QBasedPolicy(learner, explorer)
    traces = rl.CircularArraySARTSTraces(capacity=cap, n_env=n, obs_dim=4)
    agent = rl.Agent(policy, rl.Trajectory(traces))
    o = oracle.DQNRun(n=n, hidden=h, env_seed=9, net_seed=9, explorer_seed=9, sampler_seed=9, capacity=cap, batch=batch,
                      sync_freq=sync, decay_steps=40, layers=layers)
    assert np.array_equal(host(net.params), o.params)
    lr = 1e-3
    qtol = 1e-6 if layers == 2 else 5e-5
    near_ties, worst_q99, worst_max, worst_state = 0, 0.0, 0.0, 0.0
    for k in range(K):
        o.env.set_state(host(env._s), host(env._t))
        o.env.episode[:] = host(env._episode).astype(np.uint32)
        o.env.done[:] = host(env._done)
        for dst, src in ((o.params, net.params), (o.target, tn.target), (o.m, net.m), (o.v, net.v)):
            dst[:] = host(src)
        rl.run_fused_dqn(agent, env, rl.StopAfterNSteps(1))
        ga = host(policy._actions).astype(np.int32)
        o.vec_step(force_actions=ga)
        # plan!
        gq = host(policy._q)
        assert np.all(np.abs(gq - o.last_q) <= (5e-6 if layers == 2 else 2e-5) * (1 + np.abs(o.last_q)))
        gap = np.abs(o.last_q[0] - o.last_q[1])
        decisive = gap > qtol * (1 + np.abs(o.last_q).max(0))
        near_ties += int((~decisive).sum())
        assert np.array_equal(ga[decisive], o.last_plan[decisive]), f"vec-step {k}: a decisive plan! differs"
        # act!
        assert np.array_equal(host(env._t), o.env.t) and np.array_equal(host(env._done), o.env.done)
        assert np.array_equal(host(env.reward()), o.env.reward.astype(np.float32))
        st = float(np.abs(host(env._s) - np.stack(o.env.s)).max())
        worst_state = max(worst_state, st)
        assert st <= 1e-6, f"vec-step {k}: env state differs by {st:.2e}"
        # optimise!
        assert learner.n_updates == o.n_updates == k + 1 and tn.n_optimise == o.n_optimise
        d = np.abs(host(net.params) - o.params)
        q99, dmax = float(np.quantile(d, 0.99)), float(d.max())
        worst_q99, worst_max = max(worst_q99, q99), max(worst_max, dmax)
        assert q99 <= (0.02 if layers == 2 else 0.2) * lr and dmax <= 2.5 * lr, f"vec-step {k}: |dp| q99 {q99:.2e} max {dmax:.2e}"
        assert np.abs(host(tn.target) - o.target).max() <= dmax + 1e-12
    assert o.n_optimise == K % sync
    idx = np.arange(cap * n, dtype=np.int64)
    gs, g_a, gr, gt, gsn = (host(x) for x in traces.gather(torch.as_tensor(idx).cuda()))
    os_, oa, or_, ot, osn = o.ring.gather(idx)
    assert np.array_equal(g_a, oa) and np.array_equal(gr, or_) and np.array_equal(gt, ot)
    assert np.abs(gs - os_).max() <= 1e-6 and np.abs(gsn - osn).max() <= 1e-6
    note(f"fused DQN vec-step teacher-forced vs oracle, layers={layers}", vec_steps=K, envs=n, near_tie_decisions=near_ties,
         decisions=K * n, dp_q99_worst_step=worst_q99, dp_max_worst_step=worst_max, env_state_err_max=worst_state)
