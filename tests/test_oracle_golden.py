"""The oracle against the reference's own known-answer tests (tests/golden/*.json, transcribed from
RLCore/test/utils/base.jl and RLCore/test/policies/explorers/epsilon_greedy_explorer.jl) and the
Random123 Philox vectors.  This is what pins the oracle."""
import json
import math
import os

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name + ".json")) as f:
        return json.load(f)


def rows(x):
    return None if x is None else np.array(x)


SCANS = load("scans")
SELECT = load("select")


def _num(v):
    return {"inf": math.inf, "-inf": -math.inf}.get(v, v) if isinstance(v, str) else v


@pytest.mark.parametrize("case", SCANS["discount_rewards"], ids=lambda c: c["src"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_discount_rewards(case, dtype):
    r = np.array(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype=float)
    term = case.get("terminal_rows", case.get("terminal"))
    kw = dict(terminal=term, init=case.get("init"), dims=case.get("dims", 0), dtype=dtype)
    if case.get("expect_error"):
        with pytest.raises(TypeError):
            oracle.discount_rewards(r, case["gamma"], **kw)
        return
    out = oracle.discount_rewards(r, case["gamma"], **kw)
    exp = np.array(case["expect_rows"] if "expect_rows" in case else case["expect"])
    np.testing.assert_allclose(out, exp, rtol=1e-6 if dtype == np.float32 else 1.5e-8)


@pytest.mark.parametrize("case", SCANS["discount_rewards_reduced"], ids=lambda c: c["src"])
def test_discount_rewards_reduced(case):
    r = np.array(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype=float)
    term = case.get("terminal_rows", case.get("terminal"))
    kw = dict(terminal=term, init=case.get("init"), dims=case.get("dims", 0))
    if case.get("expect_error"):
        with pytest.raises(TypeError):
            oracle.discount_rewards_reduced(r, case["gamma"], **kw)
        return
    out = oracle.discount_rewards_reduced(r, case["gamma"], **kw)
    np.testing.assert_allclose(out, np.array(case["expect"]), rtol=1.5e-8)


@pytest.mark.parametrize("case", SCANS["generalized_advantage_estimation"], ids=lambda c: c["src"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gae(case, dtype):
    r = np.array(case["reward_rows"] if "reward_rows" in case else case["reward"], dtype=float)
    v = np.array(case["values_rows"] if "values_rows" in case else case["values"], dtype=float)
    term = case.get("terminal_rows", case.get("terminal"))
    kw = dict(terminal=term, dims=case.get("dims", 0), dtype=dtype)
    if case.get("expect_error"):
        with pytest.raises(TypeError):
            oracle.generalized_advantage_estimation(r, v, case["gamma"], case["lam"], **kw)
        return
    out = oracle.generalized_advantage_estimation(r, v, case["gamma"], case["lam"], **kw)
    exp = np.array(case["expect_rows"] if "expect_rows" in case else case["expect"])
    np.testing.assert_allclose(out, exp, rtol=2e-6 if dtype == np.float32 else 1.5e-8, atol=1e-12 if dtype == np.float64 else 1e-6)


@pytest.mark.parametrize("case", SELECT["find_all_max"], ids=lambda c: c["src"])
def test_find_all_max(case):
    x = [_num(v) for v in case["x"]]
    vmax, idx = oracle.find_all_max(x, case.get("mask"))
    assert vmax == _num(case["vmax"])
    assert list(idx + 1) == case["idx"]  # reference indices are 1-based


@pytest.mark.parametrize("case", SELECT["get_eps"], ids=lambda c: c["src"])
def test_get_eps(case):
    p = SELECT["get_eps_params"]
    e = oracle.get_eps(case["kind"], p["eps_stable"], p["eps_init"], p["warmup_steps"], p["decay_steps"],
                       case["step"])
    if "atol" in case:
        assert abs(e - case["expect"]) <= case["atol"]
    else:
        assert e == pytest.approx(case["expect"], rel=1.5e-8)


def test_eps_default_constructor_is_stable():
    # EpsilonGreedyExplorer(eps): kind = linear, eps_init = 1.0, warmup = decay = 0, step = 1
    # -> step >= warmup + decay -> always eps_stable  (epsilon_greedy_explorer.jl:47-78)
    for step in (1, 2, 1000):
        assert oracle.get_eps("linear", 0.3, 1.0, 0, 0, step) == 0.3


@pytest.mark.parametrize("case", SELECT["prob"], ids=lambda c: c["src"])
def test_eps_greedy_prob(case):
    p = oracle.eps_greedy_prob(case["values"], case["eps"], is_break_tie=bool(case["is_break_tie"]))
    np.testing.assert_allclose(p, case["expect"], rtol=1e-12, atol=1e-15)


def test_greedy_first_index_tie_rule():
    c = SELECT["greedy_plan"][0]
    assert oracle.findmax(c["values"]) + 1 == c["expect"]
    assert oracle.findmax(c["values"], dtype=np.float32) + 1 == c["expect"]
    # findmax semantics: NaN is maximal, first NaN wins; masked entries become typemin
    assert oracle.findmax([1.0, float("nan"), 3.0, float("nan")]) == 1
    assert oracle.findmax([1.0, 5.0, 3.0], mask=[1, 0, 1]) == 2
    # greedy eps-greedy (eps = 0) through the batched selector, all envs, no randomness involved
    v = np.tile(np.array(c["values"], np.float32)[:, None], (1, 7))
    a = oracle.eps_greedy_select(v, 0.0, seed=1, step=1)
    assert (a + 1 == c["expect"]).all()


@pytest.mark.parametrize("case", SELECT["target_sync"], ids=lambda c: c["src"])
def test_target_sync_counter(case):
    n = 0
    seen = []
    due_at = []
    for k in range(len(case["counters"])):
        due, n = oracle.target_sync_due(n, case["sync_freq"])
        seen.append(n)
        due_at.append(due)
    assert seen == case["counters"]
    assert due_at == [c == 0 for c in case["counters"]]


def test_polyak_hard_sync_and_average():
    rng = np.random.default_rng(0)
    src = rng.standard_normal(33).astype(np.float32)
    dst = rng.standard_normal(33).astype(np.float32)
    d0 = dst.copy()
    oracle.polyak(dst, src, 0.0)  # rho = 0: target replaced exactly (target_network.jl:74-102 equality test)
    assert (dst == src).all()
    dst = d0.copy()
    oracle.polyak(dst, src, 0.5)
    np.testing.assert_array_equal(dst, np.float32(0.5) * d0 + np.float32(0.5) * src)


def test_philox_kat():
    for c in load("philox")["cases"]:
        ctr = [int(x, 16) for x in c["ctr"]]
        key = [int(x, 16) for x in c["key"]]
        out = oracle.philox(key[0] | (key[1] << 32), *ctr)
        assert ["%08x" % w for w in out] == c["out"]


def test_normlogpdf_closed_form():
    # RLCore/test/utils/distributions.jl:19-27: logpdf(Normal(10, 5), 4) ~ normlogpdf(10, 5, 4)
    from scipy.stats import multivariate_normal, norm

    assert oracle.normlogpdf(10.0, 5.0, 4.0) == pytest.approx(norm(10, 5).logpdf(4.0), rel=1e-6)
    # :42-60 diagonal gaussian, 2-D: mu = [10 10; 1 1], sigma = [5 5; 6 6]
    mu = np.array([[10, 10], [1, 1]], np.float32)
    sg = np.array([[5, 5], [6, 6]], np.float32)
    x = np.array([[4, 11], [0.5, 3]], np.float32)
    out = oracle.diagnormlogpdf(mu, sg, x)
    for j in range(2):
        ref = multivariate_normal(mu[:, j], np.diag(sg[:, j].astype(np.float64) ** 2)).logpdf(x[:, j])
        assert out[j] == pytest.approx(ref, rel=1e-5)


def test_stop_after_n_steps_semantics():
    # core/stop_conditions.jl:55-69 restated in the host mirror is tested in test_host_run.py; here
    # the golden number itself: 11 `true`s in 20 calls for n = 10.
    c = SELECT["stop_after_n_steps"]
    cur, trues = 1, 0
    for _ in range(c["calls"]):
        trues += cur >= c["n"]
        cur += 1
    assert trues == c["n_true"]
