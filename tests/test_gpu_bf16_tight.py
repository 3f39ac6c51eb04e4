"""Flip-free, TIGHT parity cases for every bf16 MFMA learner instantiation.

The general bf16 cases (test_gpu_dqn3 / dqn3w / ppo3 / ppo3w) need a per-check bar of one bf16 ulp because a last-bit
difference of the MFMA's summation order can land on a DECISION -- a layer-2 relu' at a pre-activation within an ulp of
zero, a Huber branch, a PPO clip edge -- and a flipped decision changes one sample's whole backward pass.  A wrong tile or
a wrong fragment slot in one instantiation at 1e-3 of max|g| would hide under that bar.  Here every instantiation gets one
case whose inputs keep the decisions away from their edges:

  * relu nets: b2 of every net shifted by +6 (|W2 h1| stays below ~3), so every layer-2 unit is active for every sample;
    layer 1 is f32 and bit-identical on both sides (tools/micro/mfma_f32_l1), so its decisions cannot differ;
  * head weights scaled by 1/8 so that the shifted hidden activations do not blow up logits / log sigma;
  * PPO: |advantage| <= 3, 4096-sample micro-batch (no sample dominates a sum), parameters = the rollout's parameters,
    so every ratio is 1 +- 1e-6: far from the clip edges 1 +- 0.2;
  * DQN: Huber delta = 1000 (every sample on the quadratic branch: no |e| < delta decision), 4096 samples.

What is left is the MFMA's f32 summation order and bf16 rounding ties of individual dz2 / h1 elements (each moves ONE
element by a bf16 ulp of itself: ~2^-9 / batch of a sum).  Bar: max|g - o| <= BF16_TIGHT_TOL = 2e-5 of max|o| per tensor;
measured margins are in profiles/r04_parity_margins.md.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import BF16_TIGHT_TOL, assert_grad_close  # noqa: E402

pytestmark = pytest.mark.gpu

SHIFT, HEAD_SCALE = 6.0, 0.125


def _layout(ns, h, nout):
    return (("W1", h * ns), ("b1", h), ("W2", h * h), ("b2", h), ("W3", nout * h), ("b3", nout))


def _decision_free(p, ns, h, nout, relu):
    """shift b2, shrink the head (in place on a host copy of ONE net's flat parameters)"""
    p = p.copy()
    o = 0
    for name, sz in _layout(ns, h, nout):
        if name == "b2" and relu:
            p[o:o + sz] += SHIFT
        if name == "W3":
            p[o:o + sz] *= HEAD_SCALE
        o += sz
    assert o == p.size
    return p


def _check(g, ref, ns, h, nout, tag):
    o = 0
    for name, sz in _layout(ns, h, nout):
        assert_grad_close(g[o:o + sz], ref[o:o + sz], BF16_TIGHT_TOL, f"tight {tag} {name}")
        o += sz
    assert o == g.size


# ------------------------------------------------------------------------------------------------ PPO (ppo3 / ppo3w)
@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,cont", [("cartpole", False), ("pendulum", True)])
@pytest.mark.parametrize("hidden", [128, 256])
def test_ppo3_grad_tight(hidden, kind, cont, act):
    import rlhip

    n, T = 128, 32  # one micro-batch of 4096 samples = 32 tiles of 128 (ppo3.hip) / 64 tiles of 64 (ppo3w.hip)
    a = {"relu": 0, "tanh": 1}[act]
    env0 = rlhip.HipVecEnv(kind, n, seed=21)
    pol0 = rlhip.PPOPolicy(env0, update_freq=T, hidden=hidden, seed=21, layers=3, n_microbatches=1, act=a)
    ns, np_a = env0.odim, pol0.np_actor
    nout_a = 2
    p = pol0.params.cpu().numpy()
    p = np.concatenate([_decision_free(p[:np_a], ns, hidden, nout_a, act == "relu"),
                        _decision_free(p[np_a:], ns, hidden, 1, act == "relu")])
    del pol0, env0
    env = rlhip.HipVecEnv(kind, n, seed=21)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=hidden, seed=21, layers=3, n_microbatches=1, act=a, params=p)
    pol.rollout_()
    pol.gae_()
    tr = pol.trajectory
    tr.adv.clamp_(-3.0, 3.0)
    tr.ret.clamp_(-10.0, 10.0)
    pol.grad_(0, 0)
    g = pol.grad.cpu().numpy()
    losses = pol.losses.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    ocfg = oracle.ppo_default(hidden=hidden, continuous=int(cont), layers=3, n_microbatches=1, act=a)
    total = n * T
    f = np.array([oracle.permute(pol.seed, 0, total, b) for b in range(total)])
    t, i = f // n, f % n
    obs = tr.obs.cpu().numpy()[t, :, i].T.copy()
    action = tr.action_f.cpu().numpy().reshape(T, n)[t, i][None, :] if cont else tr.action_i.cpu().numpy()[t, i]
    na = 1 if cont else 2
    og, ol = oracle.ppo_loss_grad(ocfg, ns, na, p, obs, action, tr.logp.cpu().numpy()[t, i], tr.adv.cpu().numpy()[t, i],
                                  tr.ret.cpu().numpy()[t, i])
    assert np.all(np.abs(losses - ol) <= 2e-5 * (1 + np.abs(ol))), (losses, ol)
    tag = f"ppo3{'w' if hidden == 256 else ''} {kind} {act}"
    _check(g[:np_a], og[:np_a], ns, hidden, nout_a, tag + " actor")
    _check(g[np_a:], og[np_a:], ns, hidden, 1, tag + " critic")


# ------------------------------------------------------------------------------------------------ DQN (dqn3 / dqn3w)
@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("ns,na", [(4, 2), (2, 3), (3, 3)])
@pytest.mark.parametrize("hidden", [128, 256])
def test_dqn3_grad_tight(hidden, ns, na, act):
    import rlhip
    from rlhip import dqn

    batch, n_env, cap = 4096, 64, 80
    rng = np.random.default_rng(100 * hidden + 10 * ns + act)
    traces = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    obs = rng.standard_normal((ns, n_env)).astype(np.float32)
    traces.push_state_(torch.as_tensor(obs, device="cuda"))
    oring.push_state(obs)
    for _ in range(97):  # wraps
        nobs = rng.standard_normal((ns, n_env)).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = (rng.standard_normal(n_env) * 0.02).astype(np.float32)
        term = (rng.random(n_env) < 0.2).astype(np.uint8)
        traces.push_transition_(torch.as_tensor(nobs, device="cuda"), torch.as_tensor(a, device="cuda"),
                                torch.as_tensor(r, device="cuda"), torch.as_tensor(term, device="cuda"))
        oring.push_transition(nobs, a, r, term)
    # online net = target net + a small perturbation of the head
    base = _decision_free(oracle.mlp3_init(ns, hidden, na, 31, 0), ns, hidden, na, act == 0)
    o = 0
    for name, sz in _layout(ns, hidden, na):  # non-zero biases everywhere (every bias path exercised)
        if name in ("b1", "b3"):
            base[o:o + sz] = rng.standard_normal(sz).astype(np.float32) * 0.05
        o += sz
    tp = base.copy()
    p = base.copy()
    p[-(na * hidden + na):-na] *= 1.0 + 0.01 * rng.standard_normal(na * hidden).astype(np.float32)
    gamma, delta = 0.9, 1000.0
    pd, tpd = torch.as_tensor(p, device="cuda"), torch.as_tensor(tp, device="cuda")
    packed, tpacked = dqn.mlp3_pack(pd, ns, hidden, na), dqn.mlp3_pack(tpd, ns, hidden, na)
    td = torch.zeros(batch, device="cuda")
    g, loss = dqn.dqn3_grad(traces, hidden, na, act, pd, packed, tpd, tpacked, batch, gamma, delta, 7, 3, td=td)
    idx = oring.sample_indices(batch, 7, 3)
    s, a, r, t, sn = oring.gather(idx)
    rl, rg, rq = oracle.dqn3_loss_grad(ns, hidden, na, act, p, tp, s, a, r, t, sn, gamma, delta)
    tdh = td.cpu().numpy()
    assert tdh.max() < 0.5 * delta, f"the case is meant to stay on the quadratic Huber branch: max |td| = {tdh.max()}"
    assert abs(float(loss) - rl) <= 2e-5 * max(1.0, abs(rl))
    _check(g.cpu().numpy(), rg, ns, hidden, na, f"dqn3{'w' if hidden == 256 else ''} ns={ns} na={na} act={act}")
