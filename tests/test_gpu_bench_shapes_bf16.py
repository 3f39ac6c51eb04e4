"""GPU-vs-oracle parity of the four bf16 MFMA learners at the shapes `bench.py` TIMES ("next round" item 1b).

tests/test_gpu_bench_shapes.py did this for the Float32 two-layer paths in round 5; the MFMA learners' `roofline_extra` entries
were still quoted at launches no oracle comparison reached:

  dqn3_grad_mfma               rlhip_dqn3_grad_f32, hidden 128, batch 131072 out of a 256-slot x 4096-env record ring
                               (`dqn3_grad32_kernel` on 4096 32-sample tiles over its persistent workgroups + `d3_reduce_kernel`)
  dqn3w_grad_mfma_hidden256    the same entry point at hidden 256: `dqn3w_gather_kernel` + `ppo3w_fwd_kernel` x 2 + `ppo3w_bwd_kernel`
                               + `ppo3w_dw2_kernel` + `ppo3w_reduce_kernel` on 2048 64-sample tiles
  ppo3_grad_mfma               PPOPolicy(layers = 3, hidden = 128) on 4096 Pendulum envs x T = 128, clip 0.1: micro-batches of
                               131072 samples through `ppo3_gradT_kernel<3, relu, gaussian>`
  ppo3w (hidden 256)           the same policy at hidden 256 (csrc/ppo3w.hip), 2048 tiles per net

Every comparison is against `oracle/` (oracle.dqn3_loss_grad / oracle.ppo_loss_grad with layers = 3: the same bf16 roundings,
Float64 accumulation) under the bars of tests/conftest.py (BF16_GRAD_TOL per tensor, q99 bulk bar, small-tensor bar); the
constructor arguments and seeds are bench.py's.  Measured margins go to gpurun_out/bench_shape_margins.jsonl
(-> profiles/r06_parity_margins.md)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from conftest import BF16_GRAD_TOL, assert_grad_close  # noqa: E402
from test_gpu_bench_shapes import dev, host, note, oracle_microbatch  # noqa: E402


@pytest.fixture(scope="module")
def rl():
    import rlhip

    oracle.use_all_cores(True)
    yield rlhip
    oracle.use_all_cores(False)


def _net3(ns, h, na, seed):
    p = oracle.mlp3_init(ns, h, na, seed, 0)
    rng = np.random.default_rng(seed)
    o = 0
    for n, isb in ((h * ns, 0), (h, 1), (h * h, 0), (h, 1), (na * h, 0), (na, 1)):
        if isb:  # non-zero biases: every bias path of the tile carries a signal
            p[o:o + n] = rng.standard_normal(n).astype(np.float32) * 0.1
        o += n
    return p


def _tensors3(ns, h, nout):
    return (("W1", h * ns), ("b1", h), ("W2", h * h), ("b2", h), ("W3", nout * h), ("b3", nout))


# ------------------------------------------------------------------------------------------ DQN, 3-layer Q-network
@pytest.mark.parametrize("h", [128, 256])
def test_dqn3_grad_batch_131072_vs_oracle(rl, h):
    """bench.py roofline_extras(): `_dqn.dqn3_grad(tr2, h, 2, relu, ..., bm = 131072, gamma 0.99, delta 1, seed 1, ctr 0)` on the
    replay of the 4096-env CartPole agent (256 slots).  Here the ring holds pushed random transitions mirrored in oracle.Ring
    (pushed past its wrap), the nets carry non-zero biases, and the inline draw is the oracle's sampler."""
    from rlhip import dqn

    ns, na, n_env, cap, batch = 4, 2, 4096, 256, 131072
    rng = np.random.default_rng(100 + h)
    tr = rl.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    obs = (rng.standard_normal((ns, n_env)) * 0.5).astype(np.float32)
    tr.push_state_(dev(obs))
    oring.push_state(obs)
    for _ in range(cap + 37):  # wraps
        nobs = (rng.standard_normal((ns, n_env)) * 0.5).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = rng.standard_normal(n_env).astype(np.float32)
        t = (rng.random(n_env) < 0.05).astype(np.uint8)
        tr.push_transition_(dev(nobs), dev(a), dev(r), dev(t))
        oring.push_transition(nobs, a, r, t)
    p, tp = _net3(ns, h, na, 11), _net3(ns, h, na, 12)
    pd, tpd = dev(p), dev(tp)
    packed, tpacked = dqn.mlp3_pack(pd, ns, h, na), dqn.mlp3_pack(tpd, ns, h, na)
    ws = dqn.dqn3_workspace(ns, h, na, batch)
    td = torch.zeros(batch, device="cuda")
    g, loss = dqn.dqn3_grad(tr, h, na, 0, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 1, 0, workspace=ws, td=td)
    idx = oring.sample_indices(batch, 1, 0)
    s, a, r, t, sn = oring.gather(idx)
    ol, og, oq = oracle.dqn3_loss_grad(ns, h, na, 0, p, tp, s, a, r, t, sn, 0.99, 1.0)
    assert abs(float(loss) - ol) <= 2e-5 * max(1.0, abs(ol)), (float(loss), ol)
    gh, o, worst = host(g), 0, {}
    for name, n in _tensors3(ns, h, na):
        assert_grad_close(gh[o:o + n], og[o:o + n], BF16_GRAD_TOL, f"bench-shape dqn3 h={h} {name} batch 131072")
        worst[name] = float(np.abs(gh[o:o + n] - og[o:o + n]).max() / max(np.abs(og[o:o + n]).max(), 1e-30))
        o += n
    # TD errors of all 131072 samples (relu: no transcendental in the way)
    qn = oracle.mlp3_forward(tp, ns, h, na, 0, sn)
    y = r + 0.99 * (1 - t.astype(np.float32)) * qn.max(0)
    ref_td = np.abs(oq[a, np.arange(batch)] - y)
    terr = np.abs(host(td) - ref_td) / (1 + ref_td)
    assert terr.max() <= 1e-4, terr.max()
    # fixed summation order: a second launch is bit-identical
    g2, loss2 = dqn.dqn3_grad(tr, h, na, 0, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 1, 0, workspace=ws)
    assert torch.equal(g2, g) and torch.equal(loss2, loss)
    note(f"dqn3 h={h} batch 131072", loss=float(loss), oracle_loss=ol, td_err_max=float(terr.max()), grad_err_over_max=worst)


# ------------------------------------------------------------------------------------------ PPO, 3-layer actor / critic
@pytest.mark.parametrize("h", [128, 256])
def test_ppo3_gradient_131072_sample_microbatches_vs_oracle(rl, h):
    """bench.py roofline_extras(): HipVecEnv("pendulum", 4096, seed = 7), PPOPolicy(update_freq = 128, hidden = h, seed = 7,
    clip_range = 0.1, layers = 3): 4 epochs x 4 micro-batches of 131072.  On-policy (ratio = 1) and moved away from the
    behaviour policy (ratios leave the clip range)."""
    n, T = 4096, 128
    env = rl.HipVecEnv("pendulum", n, seed=7)
    pol = rl.PPOPolicy(env, update_freq=T, hidden=h, seed=7, clip_range=0.1, layers=3)
    cont = bool(env.continuous)
    assert cont and pol.cfg.n_epochs == 4 and pol.cfg.n_microbatches == 4
    ocfg = oracle.ppo_default(continuous=1, hidden=h, clip_range=0.1, layers=3)
    assert pol.np == oracle.ppo_nparams(oracle.KIND["pendulum"], ocfg)
    rng = np.random.default_rng(h)
    pol.rollout_()
    tr = pol.trajectory
    p = host(pol.params).copy()
    p2 = (p + rng.standard_normal(pol.np) * 0.01).astype(np.float32)
    ns = env.odim
    for params, cases in ((p, ((0, 0),)), (p2, ((5, 3),))):
        pol.params.copy_(dev(params))
        for epoch_ctr, mb in cases:
            pol.grad_(epoch_ctr, mb)
            g, losses = host(pol.grad).copy(), host(pol.losses).copy()
            obs, a, lp, adv, ret = oracle_microbatch(pol, tr, epoch_ctr, mb)
            assert obs.shape[1] == 131072
            og, ol = oracle.ppo_loss_grad(ocfg, ns, pol.na, params, obs, a[None, :], lp, adv, ret)
            assert np.all(np.abs(losses - ol) <= 2e-4 * (1 + np.abs(ol))), (losses, ol)
            np_a, worst = pol.np_actor, {}
            for name, ga, gb, nout in (("actor", g[:np_a], og[:np_a], 2), ("critic", g[np_a:], og[np_a:], 1)):
                o = 0
                for tname, sz in _tensors3(ns, h, nout):
                    assert_grad_close(ga[o:o + sz], gb[o:o + sz], BF16_GRAD_TOL,
                                      f"bench-shape ppo3 h={h} {name} {tname} epoch={epoch_ctr} mb={mb}")
                    worst[f"{name}.{tname}"] = float(np.abs(ga[o:o + sz] - gb[o:o + sz]).max() / max(np.abs(gb[o:o + sz]).max(), 1e-30))
                    o += sz
                assert o == ga.size
            note(f"ppo3 h={h} micro-batch 131072 epoch={epoch_ctr} mb={mb}", losses=[float(x) for x in losses],
                 oracle_losses=[float(x) for x in ol], grad_err_over_max=worst)
            pol.grad_(epoch_ctr, mb)  # deterministic
            assert np.array_equal(host(pol.grad), g)
