"""CPU tests of the oracle's 3-layer bf16 Q-network (oracle/rlo_mlp3.c) against a torch autograd model with the
same roundings (straight-through bf16 on the hidden-layer operands)."""
import numpy as np
import pytest
import torch

import oracle


def _split(p, ns, h, na):
    o = 0
    out = []
    for shape in ((h, ns), (h,), (h, h), (h,), (na, h), (na,)):
        n = int(np.prod(shape))
        t = p[o:o + n]
        out.append(t.reshape(shape[::-1]).T if len(shape) == 2 else t)  # Flux arrays are column-major
        o += n
    return out


def _ste_bf16(t):
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def _torch_q(p, ns, h, na, act, x):
    W1, b1, W2, b2, W3, b3 = _split(p, ns, h, na)
    f = torch.relu if act == 0 else torch.tanh
    h1 = f(W1 @ x + b1[:, None])
    h2 = f(_ste_bf16(W2) @ _ste_bf16(h1) + b2[:, None])
    return W3 @ h2 + b3[:, None]


def test_bf16_round_matches_torch():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * 10.0 ** rng.integers(-30, 30, 4000),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.005859375, 3.3895314e38, np.inf, -np.inf], np.float32)])
    ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(oracle.bf16_round(x), ref)


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("ns,h,na", [(4, 128, 2), (2, 32, 3), (3, 256, 3)])
def test_forward_matches_torch(ns, h, na, act):
    p = oracle.mlp3_init(ns, h, na, 3, 0)
    assert p.size == oracle.mlp3_nparams(ns, h, na) == h * ns + h + h * h + h + na * h + na
    p[h * ns:h * ns + h] = 0.1  # non-zero biases
    rng = np.random.default_rng(1)
    x = rng.standard_normal((ns, 64)).astype(np.float32)
    q = oracle.mlp3_forward(p, ns, h, na, act, x)
    ref = _torch_q(torch.from_numpy(p).double(), ns, h, na, act, torch.from_numpy(x).double()).numpy()
    # torch rounds h1 to bf16 from float64 values, the oracle from float32: compare loosely
    np.testing.assert_allclose(q, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("h", [128, 256])
@pytest.mark.parametrize("act", [0, 1])
def test_dqn3_loss_and_gradient_match_autograd(act, h):
    ns, na, b = 4, 2, 96
    rng = np.random.default_rng(2)
    p = oracle.mlp3_init(ns, h, na, 5, 0)
    tp = oracle.mlp3_init(ns, h, na, 6, 0)
    s = rng.standard_normal((ns, b)).astype(np.float32)
    sn = rng.standard_normal((ns, b)).astype(np.float32)
    a = rng.integers(0, na, b).astype(np.int32)
    r = rng.standard_normal(b).astype(np.float32) * 2
    term = rng.random(b) < 0.2
    loss, grad, q = oracle.dqn3_loss_grad(ns, h, na, act, p, tp, s, a, r, term, sn, 0.99, 1.0)
    pt = torch.from_numpy(p).float().requires_grad_(True)
    qt = _torch_q(pt, ns, h, na, act, torch.from_numpy(s))
    with torch.no_grad():
        qn = _torch_q(torch.from_numpy(tp), ns, h, na, act, torch.from_numpy(sn))
        y = torch.from_numpy(r) + 0.99 * (1 - torch.from_numpy(term.astype(np.float32))) * qn.max(0).values
    qa = qt.gather(0, torch.from_numpy(a.astype(np.int64))[None, :])[0]
    l = torch.nn.functional.huber_loss(qa, y, delta=1.0)
    l.backward()
    assert abs(loss - float(l)) < 1e-4 * max(1.0, abs(float(l)))
    g = pt.grad.numpy()
    # the oracle rounds dz2 to bf16 for the two backward GEMMs (autograd does not): ~2^-9 relative per term
    scale = np.abs(g).max()
    assert np.abs(grad - g).max() < 1e-2 * scale
    np.testing.assert_allclose(q, qt.detach().numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("h", [128, 256])
@pytest.mark.parametrize("continuous,kind", [(False, "cartpole"), (True, "pendulum")])
def test_ppo_loss_gradient_with_three_layer_nets_matches_autograd(continuous, kind, h):
    """oracle PPO loss / gradient with cfg.layers = 3 (actor and critic ns -> h -> h -> nout, bf16 hidden layer; h = 128 and
    the 256 of csrc/ppo3w.hip) against torch autograd with straight-through bf16 roundings (tolerance: the oracle also rounds
    dz2 to bf16)."""
    ns = 4 if kind == "cartpole" else 3
    na, b = (2, 200) if not continuous else (1, 200)
    nout_a = 2 * na if continuous else na
    cfg = oracle.ppo_default(hidden=h, continuous=int(continuous), layers=3)
    rng = np.random.default_rng(7)
    pa, pc = oracle.mlp3_init(ns, h, nout_a, 1, 0), oracle.mlp3_init(ns, h, 1, 1, 1)
    params = np.concatenate([pa, pc])
    assert params.size == oracle.ppo_nparams(oracle.KIND[kind], cfg)
    obs = rng.standard_normal((ns, b)).astype(np.float32)
    adv = rng.standard_normal(b).astype(np.float32)
    ret = rng.standard_normal(b).astype(np.float32)
    logp_old = (-0.7 + 0.1 * rng.standard_normal(b)).astype(np.float32)
    act_i = rng.integers(0, max(na, 2), b).astype(np.int32)
    act_f = rng.standard_normal((na, b)).astype(np.float32)
    grad, losses = oracle.ppo_loss_grad(cfg, ns, na, params, obs, act_f if continuous else act_i, logp_old, adv, ret)
    pt = torch.from_numpy(params).float().requires_grad_(True)
    x = torch.from_numpy(obs)
    out = _torch_q(pt[:pa.size], ns, h, nout_a, 0, x)
    v = _torch_q(pt[pa.size:], ns, h, 1, 0, x)[0]
    A, lo = torch.from_numpy(adv), torch.clamp(torch.from_numpy(logp_old), min=float(np.log(1e-8)))
    if continuous:
        mu, ls = out[:na], out[na:]
        se = torch.exp(ls) + 1e-8
        z = torch.from_numpy(act_f)
        lp = (-(((z - mu) / se) ** 2 + np.log(2 * np.pi)) / 2 - torch.log(se)).sum(0)
        ent = ((na * (np.log(2 * np.pi) + 1) + ls.sum(0)) / 2).mean()
    else:
        logp = torch.log_softmax(out, 0)
        lp = logp.gather(0, torch.from_numpy(act_i.astype(np.int64))[None])[0]
        ent = -(logp.exp() * logp).sum(0).mean()
    ratio = torch.exp(lp - lo)
    actor = -torch.min(ratio * A, torch.clamp(ratio, 1 - cfg.clip_range, 1 + cfg.clip_range) * A).mean()
    critic = ((torch.from_numpy(ret) - v) ** 2).mean()
    loss = cfg.actor_loss_weight * actor + cfg.critic_loss_weight * critic - cfg.entropy_loss_weight * ent
    loss.backward()
    assert abs(losses[0] - float(loss.detach())) < 2e-4 * max(1.0, abs(float(loss.detach())))
    g = pt.grad.numpy()
    assert np.abs(grad - g).max() < 1e-2 * np.abs(g).max()
