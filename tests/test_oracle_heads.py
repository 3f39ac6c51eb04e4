"""Gaussian policy heads oracle (rlo_heads.c) against an independent Float64 numpy evaluation and the properties the
reference's own test asserts (RLCore/test/utils/networks.jl:55-72): logp == diagnormlogpdf(m, L, a) for the identity
squash, logp == gn(state, a) (sample / evaluate consistency), K-sample shapes; plus the tanh correction and the
SoftGaussianNetwork form, which are the same quantity written two ways."""
import numpy as np
import pytest

import oracle


def _inputs(d, n, seed=0):
    rng = np.random.default_rng(seed)
    mu = rng.normal(size=(d, n)).astype(np.float32)
    raw = np.log1p(np.exp(rng.normal(size=(d, n)))).astype(np.float32)   # softplus output, like the reference test
    return mu, raw


def _ref_logp(mu, sg, z, squash):
    mu, sg, z = (x.astype(np.float64) for x in (mu, sg, z))
    v = (sg + 1e-8) ** 2
    lp = -0.5 * (np.log(v.prod(0)) + (((z - mu) ** 2) / v).sum(0) + mu.shape[0] * np.log(2 * np.pi))
    if squash:
        lp = lp - np.log(1 - np.tanh(z) ** 2).sum(0)
    return lp


@pytest.mark.parametrize("d,n,K", [(1, 33, 1), (10, 3, 1), (10, 3, 5), (4, 100, 3)])
@pytest.mark.parametrize("squash", [0, 1])
def test_sample_and_logp_against_float64(d, n, K, squash):
    mu, raw = _inputs(d, n)
    lo, hi = 0.3, 1.5
    a, lp = oracle.gaussian_head_sample(mu, raw, K, lo, hi, squash, 0, seed=5, env_id_base=9, step=2)
    assert a.shape == (d, K, n) and lp.shape == (K, n)                 # networks.jl test :66-68
    sg = np.clip(raw, lo, hi)
    z = np.arctanh(a.astype(np.float64)) if squash else a
    for j in range(K):
        np.testing.assert_allclose(lp[j], _ref_logp(mu, sg, z[:, j], squash), rtol=2e-4, atol=2e-4)
    if not squash and K == 1:                                          # :65  logp ≈ diagnormlogpdf(m, L, a)
        dn = np.zeros(n, np.float32)
        oracle.lib().rlo_diagnormlogpdf_f32(oracle.binding._p(np.ascontiguousarray(mu.T)),
                                            oracle.binding._p(np.ascontiguousarray(sg.T)),
                                            oracle.binding._p(np.ascontiguousarray(a[:, 0].T)),
                                            oracle.binding.C.c_int64(d), oracle.binding.C.c_int64(n), oracle.binding._p(dn))
        assert np.array_equal(lp[0], dn)
    # :66,71  logp ≈ gn(state, a)
    lp2 = oracle.gaussian_head_logp(mu, raw, a, lo, hi, squash, 0)
    np.testing.assert_allclose(lp2, lp, rtol=1e-4, atol=2e-3 if squash else 1e-6)
    # the noise is N(0, 1): standardised residuals over all draws
    if d * n * K >= 1200 and not squash:
        r = (a - mu[:, None, :]) / sg[:, None, :]
        assert abs(r.mean()) < 0.1 and abs(r.std() - 1) < 0.1
    # deterministic in (seed, env, step); different steps differ
    a2, _ = oracle.gaussian_head_sample(mu, raw, K, lo, hi, squash, 0, seed=5, env_id_base=9, step=2)
    a3, _ = oracle.gaussian_head_sample(mu, raw, K, lo, hi, squash, 0, seed=5, env_id_base=9, step=3)
    assert np.array_equal(a, a2) and not np.array_equal(a, a3)
    if K > 1:
        assert not np.array_equal(a[:, 0], a[:, 1])
    # columns are keyed by env id: shifting env_id_base by one shifts the samples by one column
    a4, _ = oracle.gaussian_head_sample(mu[:, 1:], raw[:, 1:], K, lo, hi, squash, 0, seed=5, env_id_base=10, step=2)
    assert np.array_equal(a4, a[:, :, 1:])


def test_soft_head_equals_the_tanh_corrected_gaussian_head():
    """log(1 - tanh(z)^2) = 2 (log 2 - z - softplus(-2 z)): SoftGaussianNetwork (:156) and GaussianNetwork with
    squash = tanh (:74) give the same actions and the same log-probability up to Float32 rounding"""
    mu, raw = _inputs(6, 200, seed=1)
    a_s, lp_s = oracle.gaussian_head_sample(mu, raw, 2, 0.1, 2.0, 1, 1, seed=3, step=7)
    a_g, lp_g = oracle.gaussian_head_sample(mu, raw, 2, 0.1, 2.0, 1, 0, seed=3, step=7)
    assert np.array_equal(a_s, a_g) and (np.abs(a_s) <= 1).all()
    # 1 - tanh(z)^2 cancels in Float32 for |z| >~ 4 (the reference's GaussianNetwork form is the less stable one)
    err = np.abs(lp_s - lp_g)
    assert (err < 1e-3 + 1e-4 * np.abs(lp_g)).mean() > 0.98 and err.max() < 0.2
    err = np.abs(oracle.gaussian_head_logp(mu, raw, a_s, 0.1, 2.0, 1, 1) - lp_s)      # atanh(tanh(z)) in Float32
    assert (err < 3e-3 + 1e-4 * np.abs(lp_s)).mean() > 0.98 and err.max() < 0.2


def test_sigma_clamp_and_no_logp():
    mu, raw = _inputs(3, 50, seed=2)
    a, lp = oracle.gaussian_head_sample(mu, raw, 1, 0.0, 0.0, 0, 0, want_logp=False)   # sigma clamped to 0: a == mu
    assert lp is None and np.array_equal(a[:, 0], mu)
