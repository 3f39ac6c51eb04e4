"""Decision-proof, TIGHT parity cases for every bf16 MFMA learner instantiation.

The general bf16 cases (test_gpu_dqn3 / dqn3w / ppo3 / ppo3w) need a per-check max bar of one bf16 ulp because the MFMA's
f32 summation order differs from the oracle's (Float64, rounded once) in the last bits, and a last-bit difference can land on
a DECISION: a relu' at a pre-activation within an ulp of zero, a Huber branch, a PPO clip edge, and -- the frequent one --
the bf16 ROUNDING of a dz2 / h1 element (the two sides' head outputs differ by ~1e-6 relative, so ~2e-4 of all dz2 elements
round the other way; measured with random weights: 5e-5 .. 1.5e-4 of max|g| whenever the gradient sum cancels like a random
walk).  A wrong tile or a wrong fragment slot in ONE instantiation at 1e-3 of max|g| would hide under such a bar.

Here every instantiation gets one case in which NO decision can differ, because everything up to the head outputs is EXACT
on both sides, whatever the summation order:

  * observations are small integers, W1 / b1 multiples of 1/2, W2 ternary {-1/4, 0, 1/4} (exact in bf16), b2 multiples of
    1/2, W3 ternary x 2^-7: every product and every partial sum of layer 1 (fmaf chain / f32 MFMA), layer 2 (bf16 MFMA, f32
    accumulate) and the head is a dyadic rational that fits 24 bits -- relu nets produce bit-identical z1, h1, z2, h2, logits,
    mu, log sigma, V on the GPU and in the oracle (tanh nets: exact up to h1 = tanh(z1), where ocml and glibc differ by an ulp
    on some inputs; their layer-2 sums then differ in the last bits only);
  * DQN: rewards are multiples of 1/4, gamma = 1/2, batch = 4096 = 2^12: TD targets, TD errors, the Huber branch (delta = 1,
    BOTH branches occur) and dL/dq = e / batch are exact, so dz2 is bit-identical and rounds identically;
  * PPO: log pi_old comes from the oracle's forward of the same parameters (ratio = 1 +- 1e-7: far from the clip edges),
    the advantage is +-2.5 with the sign of the action's direction and the return is V + 1 (coherent sums: no cancellation
    that would amplify f32 accumulation noise); dL/dlogits differs only by expf / logf ulps.

What is left for relu nets is the f32 summation order of the gradient sums (the kernels accumulate tiles in f32, the oracle
in Float64): bar max|g - o| <= BF16_TIGHT_RELU_TOL = 1e-6 of max|o| per tensor (measured <= 1.5e-7: f32-level parity on the
bf16 paths).  tanh nets keep one inexact step, h2 = tanh(z2) (z2 itself is exact: the 57 distinct dyadic z1 values round to
the same bf16 h1 on both sides): ocml / glibc ulps in h2 move dL/dz2 by ~1e-7 relative, so ~3e-5 of the dz2 elements still
round the other way, each worth 2^-8 of ONE sample's term = 1e-6 .. 1e-5 of a coherent 4096-term sum; bars: max <= BF16_TIGHT_TOL
= 5e-5 (measured <= 1.4e-5, always the "one column" signature of a single rounding) and, for the h x h matrices, q99 <=
BF16_TIGHT_Q99_TOL = 1e-5 (measured <= 3.3e-6) -- a wrong tile / fragment slot moves at least 1/64 of a matrix by O(1).
Measured margins: profiles/r04_parity_margins.md.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import BF16_TIGHT_RELU_TOL, BF16_TIGHT_TOL, assert_grad_close  # noqa: E402

pytestmark = pytest.mark.gpu


def _layout(ns, h, nout):
    return (("W1", h * ns), ("b1", h), ("W2", h * h), ("b2", h), ("W3", nout * h), ("b3", nout))


def _dyadic_net(rng, ns, h, nout, tanh):
    """one net's flat parameters (Flux.destructure order), all entries dyadic rationals as described above"""
    parts = []
    for name, sz in _layout(ns, h, nout):
        if name == "W1":
            v = rng.choice([-0.125, -0.0625, 0.0, 0.0625, 0.125] if tanh else [-1.0, -0.5, 0.0, 0.5, 1.0], sz)
        elif name == "b1":
            v = rng.choice([0.5, 0.625, 0.75] if tanh else [0.0, 0.5, 1.0], sz)  # tanh: h1 positive (coherent dW2 sums), |z1| < 2
        elif name == "W2":
            # tanh: |z1| < 2 and |z2| < 1.5 -- near saturation the derivative 1 - h^2 is a cancellation that turns the one-ulp
            # difference of ocml's and glibc's tanhf into 1e-5 .. 1e-3 of that unit's whole dz column (seen: one column of dW2
            # at 2.5e-5 with |z2| ~ 4, a broad 1e-5 on dW1 with |z1| ~ 3); the reference's own tanh' is 1 - y^2 too
            v = rng.choice([-0.0625, 0.0, 0.0625] if tanh else [-0.25, 0.0, 0.25], sz, p=[0.125, 0.75, 0.125])
        elif name == "b2":
            v = rng.choice([-0.25, 0.0, 0.25] if tanh else [0.0, 0.5, 1.0, 2.0], sz)
        elif name == "W3":
            # every hidden unit feeds at most ONE head output: dh2[j] = sum_o W3[o, j] dL/dout[o] with two outputs of opposite
            # weight is a cancellation (seen: mu and log sigma on one unit, dh2 = (dls - dmu) / 8: the 1e-7 differences of the
            # two sides' expf / division become 1e-4 of that unit's dz2, 2.4 % of its elements round the other way)
            v = rng.choice([-1.0, 1.0], (h, nout)) * (2.0 ** -3 if tanh else 2.0 ** -7)
            v = np.where(rng.integers(0, nout + 1, h)[:, None] == np.arange(nout)[None, :], v, 0.0).reshape(-1)  # W3[o + nout j]
        else:
            v = rng.choice([-0.25, 0.0, 0.25], sz)
        parts.append(v.astype(np.float32))
    return np.concatenate(parts)


def _check(g, ref, ns, h, nout, tag, relu):
    o = 0
    tol = BF16_TIGHT_RELU_TOL if relu else BF16_TIGHT_TOL
    for name, sz in _layout(ns, h, nout):
        assert_grad_close(g[o:o + sz], ref[o:o + sz], tol, f"tight {tag} {name}")
        o += sz
    assert o == g.size


# ------------------------------------------------------------------------------------------------ PPO (ppo3 / ppo3w)
@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,cont", [("cartpole", False), ("pendulum", True)])
@pytest.mark.parametrize("hidden", [128, 256])
def test_ppo3_grad_tight(hidden, kind, cont, act):
    import rlhip

    import os

    n, T = int(os.environ.get("RLHIP_TIGHT_N", 128)), 32  # one micro-batch of 4096 samples = 32 tiles of 128 (ppo3.hip) / 64 tiles of 64 (ppo3w.hip)
    a = {"relu": 0, "tanh": 1}[act]
    rng = np.random.default_rng(1000 * hidden + 10 * int(cont) + a)
    env = rlhip.HipVecEnv(kind, n, seed=21)
    ns, nout_a, na = env.odim, 2, (1 if cont else 2)
    pa, pc = _dyadic_net(rng, ns, hidden, nout_a, act == "tanh"), _dyadic_net(rng, ns, hidden, 1, act == "tanh")
    p = np.concatenate([pa, pc])
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=hidden, seed=21, layers=3, n_microbatches=1, act=a, params=p)
    assert pol.np_actor == pa.size and pol.np == p.size
    # a synthetic trajectory (the gradient kernels read obs, action, logp, adv, ret; nothing else)
    tr = pol.trajectory
    total = n * T
    obs = rng.integers(-3, 4, (T + 1, ns, n)).astype(np.float32)
    x = np.ascontiguousarray(obs[:T].transpose(1, 0, 2).reshape(ns, total))  # (ns, T * n): sample index t * n + i
    out = oracle.mlp3_forward(pa, ns, hidden, nout_a, a, x).astype(np.float64)  # exact head outputs (relu) on both sides
    v = oracle.mlp3_forward(pc, ns, hidden, 1, a, x)[0]
    if cont:
        mu, ls = out[0], out[1]
        assert np.abs(ls).max() < 8
        # within 1/2 of mu (rounded to 1/4): no heavy tail of (a - mu) / sigma^2 -- a single sample must not carry percents of a sum
        action = (np.round(mu * 4) / 4 + rng.choice([-0.5, -0.25, 0.25, 0.5], total)).astype(np.float32)
        se = np.exp(ls) + 1e-8
        zz = (action - mu) / se
        logp = (-(zz * zz + np.log(2 * np.pi)) / 2 - np.log(se)).astype(np.float32)
        adv = np.where(action >= mu, 2.5, -2.5).astype(np.float32)
        tr.action_f.copy_(torch.as_tensor(action.reshape(T, 1, n)))
    else:
        lse = np.log(np.exp(out - out.max(0)).sum(0)) + out.max(0)
        action = rng.integers(0, 2, total).astype(np.int32)
        logp = (out[action, np.arange(total)] - lse).astype(np.float32)
        adv = np.where(action == 1, 2.5, -2.5).astype(np.float32)
        tr.action_i.copy_(torch.as_tensor(action.reshape(T, n)))
    ret = (v + 1.0).astype(np.float32)
    tr.obs.copy_(torch.as_tensor(obs))
    tr.logp.copy_(torch.as_tensor(logp.reshape(T, n)))
    tr.adv.copy_(torch.as_tensor(adv.reshape(T, n)))
    tr.ret.copy_(torch.as_tensor(ret.reshape(T, n)))
    pol.grad_(0, 0)
    g = pol.grad.cpu().numpy()
    losses = pol.losses.cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    ocfg = oracle.ppo_default(hidden=hidden, continuous=int(cont), layers=3, n_microbatches=1, act=a)
    f = np.array([oracle.permute(pol.seed, 0, total, b) for b in range(total)])  # the kernels' sample order (any order: one batch)
    act_arg = action[f][None, :] if cont else action[f]
    og, ol = oracle.ppo_loss_grad(ocfg, ns, na, p, x[:, f].copy(), act_arg, logp[f], adv[f], ret[f])
    assert np.all(np.abs(losses - ol) <= 2e-5 * (1 + np.abs(ol))), (losses, ol)
    np_a = pa.size
    if os.environ.get("RLHIP_TIGHT_DUMP"):
        np.savez(os.path.join(os.environ["RLHIP_TIGHT_DUMP"], f"tight_{hidden}_{kind}_{act}_{n}.npz"), g=g, og=og, x=x[:, f], adv=adv[f],
                 action=action[f], out=out[:, f])
    for name, gg in (("actor", og[:np_a]), ("critic", og[np_a:])):  # the case must exercise every tensor
        o = 0
        for tname, sz in _layout(ns, hidden, nout_a if name == "actor" else 1):
            assert np.abs(gg[o:o + sz]).max() > 0, f"{name} {tname}: zero reference gradient"
            o += sz
    tag = f"ppo3{'w' if hidden == 256 else ''} {kind} {act}"
    _check(g[:np_a], og[:np_a], ns, hidden, nout_a, tag + " actor", act == "relu")
    _check(g[np_a:], og[np_a:], ns, hidden, 1, tag + " critic", act == "relu")


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
    obs = rng.integers(-3, 4, (ns, n_env)).astype(np.float32)
    traces.push_state_(torch.as_tensor(obs, device="cuda"))
    oring.push_state(obs)
    for _ in range(97):  # wraps
        nobs = rng.integers(-3, 4, (ns, n_env)).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = (rng.integers(0, 9, n_env) / 4.0).astype(np.float32)
        term = (rng.random(n_env) < 0.2).astype(np.uint8)
        traces.push_transition_(torch.as_tensor(nobs, device="cuda"), torch.as_tensor(a, device="cuda"),
                                torch.as_tensor(r, device="cuda"), torch.as_tensor(term, device="cuda"))
        oring.push_transition(nobs, a, r, term)
    p, tp = _dyadic_net(rng, ns, hidden, na, act == 1), _dyadic_net(rng, ns, hidden, na, act == 1)
    gamma, delta = 0.5, 1.0
    pd, tpd = torch.as_tensor(p, device="cuda"), torch.as_tensor(tp, device="cuda")
    packed, tpacked = dqn.mlp3_pack(pd, ns, hidden, na), dqn.mlp3_pack(tpd, ns, hidden, na)
    td = torch.zeros(batch, device="cuda")
    g, loss = dqn.dqn3_grad(traces, hidden, na, act, pd, packed, tpd, tpacked, batch, gamma, delta, 7, 3, td=td)
    idx = oring.sample_indices(batch, 7, 3)
    s, a, r, t, sn = oring.gather(idx)
    rl, rg, rq = oracle.dqn3_loss_grad(ns, hidden, na, act, p, tp, s, a, r, t, sn, gamma, delta)
    tdh = td.cpu().numpy()
    share = float((tdh < delta).mean())
    assert 0.02 < share < 0.98, f"both Huber branches are meant to occur: {share:.3f} of the samples on the quadratic one"
    if act == 0:  # relu: the whole forward is exact -> the TD errors agree bit for bit
        qn = oracle.mlp3_forward(tp, ns, hidden, na, act, sn)
        y = r + gamma * (1 - t.astype(np.float32)) * qn.max(0)
        assert np.array_equal(tdh, np.abs(rq[a, np.arange(batch)] - y).astype(np.float32))
    assert abs(float(loss) - rl) <= 2e-6 * max(1.0, abs(rl))
    _check(g.cpu().numpy(), rg, ns, hidden, na, f"dqn3{'w' if hidden == 256 else ''} ns={ns} na={na} act={act}", act == 0)
