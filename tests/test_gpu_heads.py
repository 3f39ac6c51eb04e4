"""Gaussian policy heads on the device (csrc/heads.hip) against the oracle (rlo_heads.c), through the C ABI."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rl():
    import rlhip
    return rlhip


def _inputs(d, n, seed=0):
    rng = np.random.default_rng(seed)
    mu = rng.normal(size=(d, n)).astype(np.float32)
    raw = np.log1p(np.exp(rng.normal(size=(d, n)))).astype(np.float32)
    return mu, raw


def _head(rl, mu, raw, soft, squash, lo, hi, seed, base):
    m, s = torch.tensor(mu, device="cuda"), torch.tensor(raw, device="cuda")
    cls = rl.SoftGaussianNetwork if soft else rl.GaussianNetwork
    kw = {} if soft else {"squash": "tanh" if squash else "identity"}
    return cls(pre=None, mu=lambda x: m, sigma=lambda x: s, min_sigma=lo, max_sigma=hi, seed=seed, env_id_base=base, **kw)


@pytest.mark.parametrize("d,n,K", [(1, 4099, 1), (10, 3, 5), (6, 1000, 3), (64, 17, 2)])
@pytest.mark.parametrize("soft,squash", [(0, 0), (0, 1), (1, 1)])
def test_head_matches_oracle(rl, d, n, K, soft, squash):
    mu, raw = _inputs(d, n, seed=d)
    lo, hi = 0.2, 1.7
    gn = _head(rl, mu, raw, soft, squash, lo, hi, seed=11, base=5)
    gn.step = 4
    state = torch.zeros((1, 1, n), device="cuda")
    a, lp = gn.sample(state, K)
    assert a.shape == (d, K, n) and lp.shape == (1, K, n) and gn.step == 5
    ao, lpo = oracle.gaussian_head_sample(mu, raw, K, lo, hi, squash, soft, seed=11, env_id_base=5, step=4)
    a, lp = a.cpu().numpy(), lp.cpu().numpy()[0]
    if not squash:
        assert np.array_equal(a, ao)                       # mu + sigma * noise: Float32 arithmetic on the same draws
    else:
        np.testing.assert_allclose(a, ao, rtol=0, atol=2.5e-7)   # tanhf: ocml vs glibc, <= 2 ulp near 1
    # log / tanh / log1p / exp differ in the last ulp between ocml and glibc; 1 - tanh^2 amplifies that for large |z|
    err = np.abs(lp - lpo)
    tol = 2e-5 + 2e-6 * np.abs(lpo) * d
    if squash and not soft:
        assert (err < 1e-3 + tol).mean() > 0.97 and err.max() < 0.5
    else:
        assert (err < tol * 4).all(), err.max()
    # (model)(state, action) on the oracle's actions
    lp2 = gn.logp(state, torch.tensor(ao, device="cuda")).cpu().numpy()[0]
    lp2o = oracle.gaussian_head_logp(mu, raw, ao, lo, hi, squash, soft)
    err2 = np.abs(lp2 - lp2o)
    if squash:
        assert (err2 < 1e-3 + tol).mean() > 0.97 and err2.max() < 0.5
    else:
        assert (err2 < tol * 4).all()


def test_reference_test_properties_and_call_forms(rl):
    """RLCore/test/utils/networks.jl:57-72 on the device head: shapes, logp ≈ diagnormlogpdf(m, L, a), logp ≈ gn(state, a)"""
    from rlhip import ops

    d, n = 10, 3
    mu, raw = _inputs(d, n, seed=3)
    gn = _head(rl, mu, raw, 0, 0, 0.0, float("inf"), seed=1, base=0)
    state = torch.zeros((20, n), device="cuda")
    m, L = gn(state)
    assert m.shape == L.shape == (d, n)
    a, logp = gn(state, is_sampling=True, is_return_log_prob=True)
    assert a.shape == (d, n) and logp.shape == (1, n)
    ref = ops.diagnormlogpdf(m.t().contiguous(), L.t().contiguous(), a.t().contiguous())
    assert torch.equal(logp[0], ref)
    assert torch.equal(logp, gn(state, a))
    only_a = gn(state, is_sampling=True)
    assert only_a.shape == (d, n) and not torch.equal(only_a, a)          # the step counter advanced
    acts, logps = gn.sample(state.unsqueeze(1), 5)
    assert acts.shape == (d, 5, n) and logps.shape == (1, 5, n)
    assert torch.equal(gn.logp(state.unsqueeze(1), acts), logps)
    with pytest.raises(ValueError):
        rl.GaussianNetwork(squash="sigmoid")
    with pytest.raises(rl.RLHipError):
        _head(rl, *_inputs(65, 2), 0, 0, 0.0, 1.0, 0, 0).sample(torch.zeros((1, 1, 2), device="cuda"), 1)


def test_head_throughput_is_streaming(rl):
    """2^20 states x d = 8: one launch, timed with HIP events (sanity bound only: > 0.5 TB/s of algorithmic traffic)"""
    d, n = 8, 1 << 20
    m = torch.randn((n, d), device="cuda")
    s = torch.rand((n, d), device="cuda") + 0.1
    gn = rl.GaussianNetwork(mu=lambda x: m.t(), sigma=lambda x: s.t(), squash="tanh")
    st = torch.zeros((1, 1, n), device="cuda")
    for _ in range(3):
        gn.sample(st, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gn.sample(st, 1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    byts = n * (3 * d + 1) * 4
    print(f"gaussian head d={d} n={n}: {us:.1f} us per call (incl. two transposes), {byts / us / 1e3:.1f} GB/s")
    assert byts / us / 1e3 > 100


def test_categorical_network_call_forms(rl):
    """CategoricalNetwork (networks.jl:405-432, masked :459-472): logits, one-hot Gumbel-max sample, (z, logits), mask"""
    na, n = 5, 2000
    lg = torch.randn((na, n), device="cuda")
    net = rl.CategoricalNetwork(lambda s: lg, seed=3, env_id_base=7)
    state = torch.zeros((4, n), device="cuda")
    assert torch.equal(net(state), lg)
    z, logits = net(state, is_sampling=True, is_return_log_prob=True)
    assert z.shape == (na, n) and torch.equal(logits, lg) and torch.equal(z.sum(0), torch.ones(n, device="cuda"))
    a_ref, _ = oracle.categorical_sample(lg.cpu().numpy(), seed=3, step=0, env_id_base=7)
    assert np.array_equal(net.actions.cpu().numpy(), a_ref) and np.array_equal(z.argmax(0).cpu().numpy(), a_ref)
    # empirical frequencies follow softmax(logits) on a constant-logit batch
    lg2 = torch.tensor([0.0, 1.0, 2.0, -1.0, 0.5], device="cuda").unsqueeze(1).expand(na, 20000).contiguous()
    net2 = rl.CategoricalNetwork(lambda s: lg2, seed=4)
    freq = net2(torch.zeros((1, 20000), device="cuda"), is_sampling=True).mean(1)
    np.testing.assert_allclose(freq.cpu().numpy(), torch.softmax(lg2[:, 0], 0).cpu().numpy(), atol=0.015)
    # masked: -Inf logits, never sampled
    mask = torch.ones((na, n), dtype=torch.bool, device="cuda")
    mask[1] = False
    mask[3, ::2] = False
    zm, lm = net(state, mask, is_sampling=True, is_return_log_prob=True)
    assert torch.isinf(lm[1]).all() and (zm[1] == 0).all() and (zm[3, ::2] == 0).all()
    assert torch.equal(net(state, mask)[0], lg[0])


def test_kernels_match_the_frozen_oracle_pins(rl):
    """tests/golden/oracle_pins/pins.json (frozen oracle outputs for the parts without a reference KAT): the Acrobot env
    kernel and the Gaussian head kernel reproduce them from the same seeds"""
    import json
    import os

    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_pins", "pins.json")))

    def unhx(xs, dtype):
        return np.array([float.fromhex(x) for x in xs], np.float64).astype(dtype)

    for case in pins["acrobot"]:
        dt, T = (np.float64, torch.float64) if case["dtype"] == "f64" else (np.float32, torch.float32)
        kw = dict(case["kw"])
        if "nips" in kw:
            kw = {"book_or_nips": "nips"}
        env = rl.AcrobotRK4Env(6, T=T, seed=21, env_id_base=3, **kw)
        for a, want in zip(case["actions"], case["steps"]):
            env.act0_(torch.tensor(a, dtype=torch.int32, device="cuda"))
            for k in range(4):
                np.testing.assert_allclose(env.raw_state()[k].cpu().numpy(), unhx(want["s"][k], dt),
                                           rtol=1e-12 if dt == np.float64 else 1e-6, atol=0)
            assert env.is_terminated().cpu().numpy().astype(int).tolist() == want["done"]
            assert np.array_equal(env.reward().cpu().numpy(), unhx(want["reward"], dt))
    h = pins["heads"]
    d, n = h["shape"]
    mu, raw = unhx(h["mu"], np.float32).reshape(d, n), unhx(h["raw_sigma"], np.float32).reshape(d, n)
    for c in h["cases"]:
        gn = _head(rl, mu, raw, c["soft"], c["squash"], 0.2, 1.5, seed=9, base=1)
        gn.step = 4
        a, lp = gn.sample(torch.zeros((1, 1, n), device="cuda"), 2)
        np.testing.assert_allclose(a.cpu().numpy().ravel(), unhx(c["action"], np.float32), rtol=3e-7, atol=0)
        np.testing.assert_allclose(lp.cpu().numpy().ravel(), unhx(c["logp"], np.float32), rtol=2e-5, atol=2e-5)
