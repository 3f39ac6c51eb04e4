"""n-step TD targets on the device (VERDICT r5 missing item 4; SURVEY.md row L2 `R = r + gamma^n (1 - t) max Qt(s')`):
`rlhip_ring_sample_indices_nstep` + `rlhip_ring_fold_nstep` (the NStepBatchSampler of RLTrajectories 0.4 as a device fold into a
batch-sized record ring), the unchanged DQN gradient entry points on the folded ring with gamma^n, `rlhip_td_target_n_f32`, and the
per-stage agent loop with `DQNLearner(n_step = 3)` -- all against the oracle (oracle/rlo_buffer.c, oracle.dqn_run(n_step = 3)).
Integer / index / return arithmetic bit-exact, gradients under F32_GRAD_TOL / BF16_GRAD_TOL."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from conftest import BF16_GRAD_TOL, F32_GRAD_TOL, assert_grad_close  # noqa: E402
from test_gpu_bench_shapes import dev, host, note  # noqa: E402


def _filled(rl, ns, n_env=64, cap=40, pushes=57, na=2, seed=0, p_term=0.2):
    rng = np.random.default_rng(seed)
    tr = rl.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    obs = rng.standard_normal((ns, n_env)).astype(np.float32)
    tr.push_state_(dev(obs))
    oring.push_state(obs)
    for _ in range(pushes):
        nobs = rng.standard_normal((ns, n_env)).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = (rng.standard_normal(n_env) * 2).astype(np.float32)
        t = (rng.random(n_env) < p_term).astype(np.uint8)
        tr.push_transition_(dev(nobs), dev(a), dev(r), dev(t))
        oring.push_transition(nobs, a, r, t)
    return tr, oring


@pytest.mark.parametrize("ns", [4, 3, 2])
@pytest.mark.parametrize("n_step", [1, 2, 3, 5, 32])
def test_fold_nstep_bit_exact_vs_oracle(ns, n_step):
    import rlhip as rl

    tr, oring = _filled(rl, ns, seed=ns + n_step)
    batch, gamma = 300, 0.97
    smp = rl.NStepBatchSampler(n_step, gamma, batch, seed=7)
    assert smp.gamma_n == oracle.gamma_pow(gamma, n_step)
    for ctr in (0, 4):
        idx = smp.sample_indices(tr, ctr)
        oidx = oracle.ring_sample_indices_nstep(oring, batch, n_step, 7, ctr)
        assert np.array_equal(host(idx), oidx)
        folded, iota = smp.fold(tr, idx)
        assert np.array_equal(host(iota), np.arange(batch)) and len(folded) == 1 and folded.n_env == batch
        got = [host(x) for x in folded.gather(iota)]
        for g, o, name in zip(got, oracle.ring_gather_nstep(oring, oidx, n_step, gamma), ("state", "action", "return", "terminal", "next_state")):
            assert np.array_equal(g, o), f"n-step {name} differs (n = {n_step}, draw {ctr})"
        if n_step == 1:  # the fold of a 1-step window IS the stored transition
            for g, p in zip(got, (host(x) for x in tr.gather(idx))):
                assert np.array_equal(g, p)
    # pad words of the folded records are zero (a record ring's contract: csrc/ring_device.h)
    assert float(folded.records[0, :, 12:].abs().max()) == 0.0 and float(folded.records.view(torch.int32)[0, :, 7].abs().max()) == 0
    # every valid start index, incl. the newest full window
    allidx = np.arange((len(oring) - n_step + 1) * oring.rb.n_env, dtype=np.int64)
    smp2 = rl.NStepBatchSampler(n_step, gamma, allidx.size, seed=7)
    folded, iota = smp2.fold(tr, dev(allidx))
    for g, o in zip((host(x) for x in folded.gather(iota)), oracle.ring_gather_nstep(oring, allidx, n_step, gamma)):
        assert np.array_equal(g, o)


def test_fold_nstep_argument_validation():
    import rlhip as rl
    from rlhip._lib import RLHipError, call
    from rlhip.ops import ptr, stream_ptr

    tr, _ = _filled(rl, 4, pushes=3)          # 3 transitions stored
    idx = torch.zeros(8, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        rl.NStepBatchSampler(33, 0.9, 8)
    with pytest.raises(RLHipError):            # fewer than n_step transitions
        rl.NStepBatchSampler(5, 0.9, 8).sample_indices(tr)
    wrong = rl.CircularArraySARTSTraces(capacity=1, n_env=9, obs_dim=4)
    with pytest.raises(RLHipError):            # folded ring of the wrong width
        call("rlhip_ring_fold_nstep", C.byref(tr.rb), ptr(idx), 8, 2, 0.9, C.byref(wrong.rb), None, stream_ptr())
    frames = rl.CircularArraySARTSTraces(capacity=4, n_env=8, obs_dim=6)
    with pytest.raises(RLHipError):            # not a record ring
        call("rlhip_ring_fold_nstep", C.byref(frames.rb), ptr(idx), 8, 2, 0.9, C.byref(wrong.rb), None, stream_ptr())


@pytest.mark.parametrize("layers,h", [(2, 128), (3, 128), (3, 256)])
def test_dqn_gradient_on_nstep_batch_vs_oracle(layers, h):
    """the unchanged gradient entry points on the folded ring with gamma^n against oracle.dqn[3]_loss_grad on the oracle's n-step batch"""
    import rlhip as rl
    from rlhip import dqn
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    ns, na, n_step, gamma, batch = 4, 2, 3, 0.99, 1000
    tr, oring = _filled(rl, ns, n_env=96, cap=50, pushes=70, seed=11)
    rng = np.random.default_rng(5)
    smp = rl.NStepBatchSampler(n_step, gamma, batch, seed=3)
    idx = smp.sample_indices(tr, 2)
    folded, iota = smp.fold(tr, idx)
    s, a, R, t, sn = oracle.ring_gather_nstep(oring, host(idx), n_step, gamma)
    gn = oracle.gamma_pow(gamma, n_step)
    if layers == 2:
        p = (oracle.mlp2_init(ns, h, na, 5, 0) + rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.1).astype(np.float32)
        pt = (p + rng.standard_normal(p.size) * 0.05).astype(np.float32)
        dp, dpt = dev(p), dev(pt)
        ws = dqn.dqn_workspace(ns, h, na, batch)
        grad, loss, td = torch.empty_like(dp), torch.empty(1, device="cuda"), torch.empty(batch, device="cuda")
        call("rlhip_dqn_grad_idx_f32", C.byref(folded.rb), h, na, 0, ptr(dp), ptr(dpt), batch, ptr(iota), gn, 1.0, ptr(ws), ptr(grad),
             ptr(loss), ptr(td), stream_ptr())
        ol, og = oracle.dqn_loss_grad(ns, h, na, 0, p, pt, s, a, R, t, sn, gn, 1.0)
        assert float(loss) == pytest.approx(ol, rel=1e-5)
        assert_grad_close(host(grad), og, F32_GRAD_TOL, "n-step dqn_grad_idx on the folded ring")
    else:
        p, pt = oracle.mlp3_init(ns, h, na, 11, 0), oracle.mlp3_init(ns, h, na, 12, 0)
        dp, dpt = dev(p), dev(pt)
        pk, ptk = dqn.mlp3_pack(dp, ns, h, na), dqn.mlp3_pack(dpt, ns, h, na)
        grad, loss = dqn.dqn3_grad(folded, h, na, 0, dp, pk, dpt, ptk, batch, gn, 1.0, 0, 0, idx=iota)
        ol, og, _ = oracle.dqn3_loss_grad(ns, h, na, 0, p, pt, s, a, R, t, sn, gn, 1.0)
        assert abs(float(loss) - ol) <= 2e-5 * max(1.0, abs(ol))
        o = 0
        for name, n in (("W1", h * ns), ("b1", h), ("W2", h * h), ("b2", h), ("W3", na * h), ("b3", na)):
            assert_grad_close(host(grad)[o:o + n], og[o:o + n], BF16_GRAD_TOL, f"n-step dqn3 h={h} {name}")
            o += n


def test_td_target_n_vs_oracle():
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    rng = np.random.default_rng(2)
    n, na = 1000, 3
    q = rng.standard_normal((na, n)).astype(np.float32)
    r = rng.standard_normal(n).astype(np.float32)
    t = (rng.random(n) < 0.3).astype(np.uint8)
    out = torch.empty(n, device="cuda")
    dq, dr, dt = dev(q), dev(r), dev(t)   # (kept alive across the launches)
    for n_step in (1, 3, 10):
        call("rlhip_td_target_n_f32", ptr(dq), na, n, n, 1, ptr(dr), ptr(dt), 0.99, n_step, ptr(out), stream_ptr())
        ref = oracle.td_target(q, r, t, oracle.gamma_pow(0.99, n_step))
        assert np.array_equal(host(out), ref)


def test_per_stage_agent_loop_with_nstep_learner_vs_oracle():
    """run(agent, env) with DQNLearner(n_step = 3) against oracle.dqn_run(n_step = 3): 512 CartPole envs x 80 vec-steps, free-running
    (Float32 two-layer net: the bars of tests/test_gpu_dqn_agent_vs_oracle.py)"""
    import rlhip as rl

    n, cap, K, batch, h = 512, 32, 80, 256, 128
    env = rl.CartPoleEnv(n, seed=3)
    net = rl.HipApproximator(4, h, 2, seed=3)
    tn = rl.TargetNetwork(net, sync_freq=25)
    learner = rl.DQNLearner(tn, batchsize=batch, min_replay_history=n, seed=3, n_step=3)
    explorer = rl.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=50, seed=3)
    traces = rl.CircularArraySARTSTraces(capacity=cap, n_env=n, obs_dim=4)
    agent = rl.Agent(rl.QBasedPolicy(learner, explorer), rl.Trajectory(traces))
    p0 = host(net.params).copy()
    rl.run(agent, env, rl.StopAfterNSteps(K))
    torch.cuda.synchronize()
    o = oracle.dqn_run(K, n=n, hidden=h, env_seed=3, net_seed=3, explorer_seed=3, sampler_seed=3, capacity=cap, batch=batch,
                       sync_freq=25, decay_steps=50, n_step=3)
    assert learner.n_updates == o.n_updates == K - 2 and tn.n_optimise == o.n_optimise
    idx = np.arange(cap * n, dtype=np.int64)
    ga = host(traces.gather(dev(idx))[1]).reshape(cap, n)
    oa = o.ring.gather(idx)[1].reshape(cap, n)
    n_flip = int((ga != oa).any(0).sum())
    d = np.abs(host(net.params) - o.params)
    note("per-stage DQN agent loop, n_step = 3, vs oracle", vec_steps=K, envs=n, envs_with_a_different_action_in_the_ring=n_flip,
         dp_q99=float(np.quantile(d, 0.99)), dp_max=float(d.max()), moved_q50=float(np.median(np.abs(o.params - p0))))
    assert n_flip <= 4
    assert np.quantile(d, 0.99) < 0.2e-3 and d.max() < K * 2e-3 and np.median(np.abs(o.params - p0)) > 1e-3
    with pytest.raises(NotImplementedError):
        rl.run_fused_dqn(agent, env, rl.StopAfterNSteps(1))
