"""The parts of the oracle whose arithmetic the reference delegates to un-vendored packages (Flux Dense /
Zygote backward / Optimisers.Adam / Flux.Losses.huber_loss, removed Zoo PPO & DQN losses) are "parity
unpinned" by the reference; they are cross-checked here against PyTorch fp32 on the CPU."""
import numpy as np
import pytest
import torch

import oracle


def split(p, n_in, h, n_out):
    o = 0
    W1 = p[o:o + h * n_in].reshape(n_in, h).T; o += h * n_in  # noqa: E702  (col-major h x n_in)
    b1 = p[o:o + h]; o += h  # noqa: E702
    W2 = p[o:o + n_out * h].reshape(h, n_out).T; o += n_out * h  # noqa: E702
    b2 = p[o:o + n_out]
    return W1, b1, W2, b2


def torch_mlp(p, n_in, h, n_out, act, x):
    W1, b1, W2, b2 = split(p, n_in, h, n_out)
    hid = W1 @ x + b1[:, None]
    hid = torch.relu(hid) if act == 0 else torch.tanh(hid)
    return W2 @ hid + b2[:, None]


@pytest.mark.parametrize("act", [0, 1])
def test_mlp_forward_backward(act):
    rng = np.random.default_rng(0)
    n_in, h, n_out, B = 4, 32, 3, 50
    p = (rng.standard_normal(oracle.mlp2_nparams(n_in, h, n_out)) * 0.3).astype(np.float32)
    x = rng.standard_normal((n_in, B)).astype(np.float32)
    dout = rng.standard_normal((n_out, B)).astype(np.float32)
    out = oracle.mlp2_forward(p, n_in, h, n_out, act, x)
    pt = torch.tensor(p, requires_grad=True)
    ref = torch_mlp(pt, n_in, h, n_out, act, torch.tensor(x))
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    (ref * torch.tensor(dout)).sum().backward()
    g = oracle.mlp2_backward(p, n_in, h, n_out, act, x, dout)
    np.testing.assert_allclose(g, pt.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("continuous", [False, True])
@pytest.mark.parametrize("act", [0, 1])
def test_ppo_loss_and_gradient(continuous, act):
    rng = np.random.default_rng(1)
    ns, na, h, B = (3, 1, 16, 200) if continuous else (4, 3, 16, 200)
    cfg = oracle.ppo_default(hidden=h, act=act, continuous=int(continuous))
    nout = 2 * na if continuous else na
    np_a = oracle.mlp2_nparams(ns, h, nout)
    np_c = oracle.mlp2_nparams(ns, h, 1)
    p = (rng.standard_normal(np_a + np_c) * 0.3).astype(np.float32)
    obs = rng.standard_normal((ns, B)).astype(np.float32)
    adv = rng.standard_normal(B).astype(np.float32)
    ret = rng.standard_normal(B).astype(np.float32)
    logp_old = (rng.standard_normal(B) * 0.3 - 1.0).astype(np.float32)
    if continuous:
        a = rng.standard_normal((1, B)).astype(np.float32)
    else:
        a = rng.integers(0, na, B).astype(np.int32)
    g, losses = oracle.ppo_loss_grad(cfg, ns, na, p, obs, a, logp_old, adv, ret)

    pt = torch.tensor(p, requires_grad=True)
    X = torch.tensor(obs)
    out = torch_mlp(pt[:np_a], ns, h, nout, act, X)
    v = torch_mlp(pt[np_a:], ns, h, 1, act, X)[0]
    lo = torch.clamp(torch.tensor(logp_old), min=float(np.log(1e-8)))
    A, R = torch.tensor(adv), torch.tensor(ret)
    if continuous:
        mu, ls = out[0], out[1]
        sg = torch.exp(ls)
        z = torch.tensor(a[0])
        log2pi = torch.log(torch.tensor(2.0 * np.float32(np.pi)))
        zz = (z - mu) / (sg + 1e-8)
        lp = -(zz ** 2 + log2pi) / 2 - torch.log(sg + 1e-8)
        ent = ((log2pi + 1) + ls).mean() / 2
    else:
        logp = torch.log_softmax(out, dim=0)
        lp = logp[torch.tensor(a, dtype=torch.long), torch.arange(B)]
        ent = -(torch.softmax(out, 0) * logp).sum() / B
    ratio = torch.exp(lp - lo)
    surr1, surr2 = ratio * A, torch.clamp(ratio, 1 - cfg.clip_range, 1 + cfg.clip_range) * A
    actor_loss = -torch.minimum(surr1, surr2).mean()
    critic_loss = ((R - v) ** 2).mean()
    loss = cfg.actor_loss_weight * actor_loss + cfg.critic_loss_weight * critic_loss - cfg.entropy_loss_weight * ent
    loss.backward()
    np.testing.assert_allclose(losses, [loss.item(), actor_loss.item(), critic_loss.item(), ent.item()], rtol=1e-4,
                               atol=1e-6)
    np.testing.assert_allclose(g, pt.grad.numpy(), rtol=2e-3, atol=2e-5 * np.abs(g).max())


def test_dqn_loss_and_gradient():
    rng = np.random.default_rng(2)
    ns, h, na, B = 4, 24, 2, 128
    p = (rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.4).astype(np.float32)
    pt_ = (p + rng.standard_normal(p.size).astype(np.float32) * 0.1).astype(np.float32)
    s, sn = rng.standard_normal((ns, B)).astype(np.float32), rng.standard_normal((ns, B)).astype(np.float32)
    a = rng.integers(0, na, B).astype(np.int32)
    r = (rng.standard_normal(B) * 2).astype(np.float32)
    t = rng.random(B) < 0.2
    loss, g = oracle.dqn_loss_grad(ns, h, na, 0, p, pt_, s, a, r, t, sn, 0.99, 1.0)
    P = torch.tensor(p, requires_grad=True)
    q = torch_mlp(P, ns, h, na, 0, torch.tensor(s))[torch.tensor(a, dtype=torch.long), torch.arange(B)]
    with torch.no_grad():
        qn = torch_mlp(torch.tensor(pt_), ns, h, na, 0, torch.tensor(sn)).max(0).values
        G = torch.tensor(r) + 0.99 * (1 - torch.tensor(t, dtype=torch.float32)) * qn
    ref = torch.nn.HuberLoss(delta=1.0)(q, G)
    ref.backward()
    assert loss == pytest.approx(ref.item(), rel=1e-5)
    np.testing.assert_allclose(g, P.grad.numpy(), rtol=1e-3, atol=1e-6)


def test_adam_matches_torch():
    rng = np.random.default_rng(3)
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    po, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    pt = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    for t in range(1, 11):
        g = rng.standard_normal(n).astype(np.float32)
        oracle.adam(po, g, m, v, 1e-3, 0.9, 0.999, 1e-8, t)
        pt.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(po, pt.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_huber_and_clip_and_gaussian():
    rng = np.random.default_rng(4)
    q, tg = (rng.standard_normal(333) * 2).astype(np.float32), rng.standard_normal(333).astype(np.float32)
    loss, dq = oracle.huber(q, tg, 1.0)
    Q = torch.tensor(q, requires_grad=True)
    ref = torch.nn.HuberLoss(delta=1.0)(Q, torch.tensor(tg))
    ref.backward()
    assert loss == pytest.approx(ref.item(), rel=1e-6)
    np.testing.assert_allclose(dq, Q.grad.numpy(), rtol=1e-6, atol=1e-9)
    g = rng.standard_normal(777).astype(np.float32)
    g2 = g.copy()
    gn = oracle.clip_by_global_norm(g2, 0.5)
    assert gn == pytest.approx(np.linalg.norm(g.astype(np.float64)), rel=1e-6)
    np.testing.assert_allclose(np.linalg.norm(g2.astype(np.float64)), 0.5, rtol=1e-5)
    G = [torch.tensor(g.copy(), requires_grad=True)]
    G[0].grad = torch.tensor(g.copy())
    torch.nn.utils.clip_grad_norm_(G, 0.5)
    np.testing.assert_allclose(g2, G[0].grad.numpy(), rtol=1e-4)
    assert oracle.normlogpdf(0.3, 1.7, -0.4) == pytest.approx(
        torch.distributions.Normal(0.3, 1.7).log_prob(torch.tensor(-0.4)).item(), rel=1e-5)


def test_gaussian_sampler_statistics_and_permutation():
    z = []
    for i in range(2000):
        w = oracle.philox(11, i, 0, 5, oracle.TAG["NORMAL"])
        import ctypes as C

        a, b = C.c_float(), C.c_float()
        oracle.lib().rlo_normal_pair_f32(C.c_uint32(w[0]), C.c_uint32(w[1]), C.byref(a), C.byref(b))
        z += [a.value, b.value]
    z = np.array(z)
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05
    for n in (1, 2, 7, 64, 1000, 4097):
        p = oracle.permutation(5, 3, n)
        assert np.array_equal(np.sort(p), np.arange(n))
    assert not np.array_equal(oracle.permutation(5, 3, 1000), oracle.permutation(5, 4, 1000))


def test_weighted_dqn_loss_and_is_weights_match_torch():
    """round 4: importance-sampling weights of prioritized replay -- w = 1 / (p + 1e-10)^beta, normalised by the maximum;
    loss = mean(w .* huber(td)) (removed Zoo PrioritizedDQN, parity unpinned): oracle vs PyTorch autograd"""
    rng = np.random.default_rng(7)
    ns, h, na, B = 4, 24, 2, 128
    p = (rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.4).astype(np.float32)
    pt_ = (p + rng.standard_normal(p.size).astype(np.float32) * 0.1).astype(np.float32)
    s, sn = rng.standard_normal((ns, B)).astype(np.float32), rng.standard_normal((ns, B)).astype(np.float32)
    a = rng.integers(0, na, B).astype(np.int32)
    r = (rng.standard_normal(B) * 2).astype(np.float32)
    t = rng.random(B) < 0.2
    prio = (rng.random(B).astype(np.float32) + 1e-3) ** 0.6
    for beta in (0.4, 1.0):
        w = oracle.per_is_weights(prio, beta)
        wt = 1.0 / (torch.tensor(prio, dtype=torch.float64) + 1e-10) ** beta
        wt = (wt / wt.max()).float()
        np.testing.assert_allclose(w, wt.numpy(), rtol=2e-7, atol=0)
        assert w.max() == 1.0 and (w > 0).all()
        loss, g = oracle.dqn_loss_grad(ns, h, na, 0, p, pt_, s, a, r, t, sn, 0.99, 1.0, weights=w)
        P = torch.tensor(p, requires_grad=True)
        q = torch_mlp(P, ns, h, na, 0, torch.tensor(s))[torch.tensor(a, dtype=torch.long), torch.arange(B)]
        with torch.no_grad():
            qn = torch_mlp(torch.tensor(pt_), ns, h, na, 0, torch.tensor(sn)).max(0).values
            G = torch.tensor(r) + 0.99 * (1 - torch.tensor(t, dtype=torch.float32)) * qn
        ref = (torch.tensor(w) * torch.nn.HuberLoss(delta=1.0, reduction="none")(q, G)).mean()
        ref.backward()
        assert loss == pytest.approx(ref.item(), rel=1e-5)
        np.testing.assert_allclose(g, P.grad.numpy(), rtol=1e-3, atol=1e-6)
    # beta = 0: all weights 1 -> the unweighted loss, bit for bit
    w0 = oracle.per_is_weights(prio, 0.0)
    assert np.array_equal(w0, np.ones(B, np.float32))
    l0, g0 = oracle.dqn_loss_grad(ns, h, na, 0, p, pt_, s, a, r, t, sn, 0.99, 1.0, weights=w0)
    l1, g1 = oracle.dqn_loss_grad(ns, h, na, 0, p, pt_, s, a, r, t, sn, 0.99, 1.0)
    assert l0 == l1 and np.array_equal(g0, g1)
