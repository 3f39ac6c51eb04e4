"""CPU tests of the oracle's remaining explorers (oracle/rlo_select.c): hand-computed cases of the restated
StatsBase.sample walk, masks, sampling laws, UCB behaviour (RLCore/src/policies/explorers/*.jl; the reference has no
tests for these explorers: test/policies/explorers/explorers.jl is a one-line include list)."""
import numpy as np

import oracle


def _u(seed, i, step):
    w = oracle.philox(seed, i, 0, step, oracle.TAG["EXPLORE"])
    return oracle.u01_f64(w[0], w[1])


def test_weighted_walk_hand_cases():
    vals = np.array([[1.0, 2.0, 3.0, 4.0]] * 64, np.float32).T  # (4, 64), weights 1..4, sum 10
    a = oracle.explorer_select("weighted", vals, 5, 9)
    for i in range(64):
        t = _u(5, i, 9) * 10.0
        expect = 0 if 1.0 >= t else 1 if 3.0 >= t else 2 if 6.0 >= t else 3  # `while cw < t` walk
        assert a[i] == expect
    # is_normalized = true takes the stored sum 1 (Weights(values, one(T)))
    p = vals / 10.0
    assert np.array_equal(oracle.explorer_select("weighted", p, 5, 9, is_normalized=True), a)


def test_masks():
    rng = np.random.default_rng(0)
    vals = rng.random((5, 2000)).astype(np.float32)
    mask = rng.random((5, 2000)) < 0.6
    mask[0, ~mask.any(0)] = True
    for kind in ("weighted", "weighted_softmax", "gumbel_softmax"):
        a = oracle.explorer_select(kind, vals, 1, 2, mask=mask)
        assert mask[a, np.arange(2000)].all(), kind  # an illegal action is never drawn


def test_sampling_laws():
    n = 200_000
    logits = np.array([0.3, -1.0, 1.2, 0.0], np.float32)
    p = np.exp(logits - logits.max())
    p /= p.sum()
    vals = np.repeat(logits[:, None], n, 1)
    for kind in ("weighted_softmax", "gumbel_softmax"):
        a = oracle.explorer_select(kind, vals, 3, 7)
        cnt = np.bincount(a, minlength=4)
        chi2 = ((cnt - p * n) ** 2 / (p * n)).sum()
        assert chi2 < 30, (kind, chi2)  # 3 dof
    w = np.array([0.5, 0.0, 2.5, 1.0], np.float32)
    a = oracle.explorer_select("weighted", np.repeat(w[:, None], n, 1), 3, 8)
    cnt = np.bincount(a, minlength=4)
    assert cnt[1] == 0
    q = w / w.sum()
    nz = q > 0
    assert (((cnt - q * n) ** 2)[nz] / (q * n)[nz]).sum() < 30


def test_ucb_tries_every_action_first_then_prefers_the_best():
    na, n = 3, 50
    vals = np.tile(np.array([[0.1], [0.9], [0.5]], np.float32), (1, n))
    counts = np.full((na, n), 1e-10)
    seen = np.zeros((na, n), bool)
    for step in range(1, 4):  # with eps counts the bonus of an untried action is huge: all three get tried
        a = oracle.ucb_select(vals, 2.0, counts, step, 4)
        seen[a, np.arange(n)] = True
    assert seen.all()
    picks = np.zeros((na, n))
    for step in range(4, 400):
        a = oracle.ucb_select(vals, 2.0, counts, step, 4)
        picks[a, np.arange(n)] += 1
    assert (picks.argmax(0) == 1).all()
    assert np.allclose(counts.sum(0), 399 + 3e-10)
