"""GPU parity (bit-exact integer actions) of the remaining explorers (select.hip) against the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,cls", [("weighted", "WeightedExplorer"), ("weighted_softmax", "WeightedSoftmaxExplorer"),
                                     ("gumbel_softmax", "GumbelSoftmaxExplorer")])
@pytest.mark.parametrize("na,n", [(1, 7), (2, 4096), (3, 1000), (18, 513), (64, 300)])
@pytest.mark.parametrize("masked", [False, True])
def test_sampling_explorers_bit_exact(kind, cls, na, n, masked):
    import rlhip

    rng = np.random.default_rng(na * 1000 + n)
    vals = (rng.random((na, n)) if kind == "weighted" else rng.standard_normal((na, n)) * 2).astype(np.float32)
    mask = None
    if masked:
        mask = rng.random((na, n)) < 0.7
        mask[0, ~mask.any(0)] = True
    ex = getattr(rlhip, cls)(seed=11)
    vd = torch.as_tensor(vals, device="cuda")
    md = None if mask is None else torch.as_tensor(mask, device="cuda")
    for step in (1, 2):
        a = ex.plan_(vd, md, env_id_base=3) if masked else ex.plan_(vd, env_id_base=3)
        ref = oracle.explorer_select(kind, vals, 11, step, env_id_base=3, mask=mask) + 1
        assert np.array_equal(a.cpu().numpy(), ref)
    assert ex.step == 3


def test_weighted_normalized_and_strided_values():
    import rlhip

    rng = np.random.default_rng(5)
    p = rng.random((4, 257)).astype(np.float32)
    p /= p.sum(0, keepdims=True)
    ex = rlhip.WeightedExplorer(is_normalized=True, seed=2)
    big = torch.zeros((8, 300), device="cuda")
    view = big[:4, :257]  # non-contiguous view: the kernel takes the strides
    view.copy_(torch.as_tensor(p, device="cuda"))
    a = ex.plan_(view)
    assert np.array_equal(a.cpu().numpy(), oracle.explorer_select("weighted", p, 2, 1, is_normalized=True) + 1)


def test_ucb_bit_exact_over_many_steps():
    import rlhip

    na, n = 4, 333
    rng = np.random.default_rng(9)
    ex = rlhip.UCBExplorer(na, n_env=n, c=2.0, seed=6)
    counts = np.full((na, n), 1e-10)
    for step in range(1, 40):
        vals = rng.standard_normal((na, n)).astype(np.float32)
        if step % 7 == 0:
            vals[1] = vals[2]  # exact ties between two actions
        a = ex.plan_(torch.as_tensor(vals, device="cuda"), env_id_base=1)
        ref = oracle.ucb_select(vals, 2.0, counts, step, 6, env_id_base=1) + 1
        assert np.array_equal(a.cpu().numpy(), ref), step
    assert np.array_equal(ex.actioncounts.cpu().numpy(), counts)


def test_batch_explorer_vector_and_matrix():
    import rlhip

    ex = rlhip.BatchExplorer(rlhip.GumbelSoftmaxExplorer(seed=1))
    v = torch.tensor([0.1, 2.0, -1.0], device="cuda")
    a1 = ex.plan_(v)
    assert a1.shape == (1,) and 1 <= int(a1[0]) <= 3
    a2 = ex.plan_(torch.randn((3, 10), device="cuda"))
    assert a2.shape == (10,)


# --------------------------------------------------------------------------------- prob(explorer, values[, mask])
def test_eps_greedy_prob_golden_and_random_bit_exact():
    """rlhip_eps_greedy_prob_f32 = RLBase.prob(::EpsilonGreedyExplorer, values[, mask]) (epsilon_greedy_explorer.jl:141-194) on
    device values: the reference's own expectations (RLCore/test/policies/explorers/epsilon_greedy_explorer.jl:45-73 ->
    tests/golden/select.json "prob") in every column, then random values with exact ties, NaNs and masks against the oracle --
    Float64 bit for bit, in both addressing modes (SoA and the storage of a Julia (na, N) matrix)."""
    import json
    import os

    from rlhip import ops
    from rlhip.dqn import EpsilonGreedyExplorer

    S = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "select.json")))
    for case in S["prob"]:
        v = np.array(case["values"], np.float32)
        assert np.array_equal(v.astype(np.float64) == v.astype(np.float64).max(), np.array(case["values"]) == max(case["values"]))
        V = torch.as_tensor(np.tile(v[:, None], (1, 7))).cuda()
        p = ops.eps_greedy_prob(V, case["eps"], is_break_tie=bool(case["is_break_tie"])).cpu().numpy()
        for i in range(7):
            np.testing.assert_allclose(p[:, i], case["expect"], rtol=0, atol=1e-15)  # the reference's `≈`
            assert np.array_equal(p[:, i], oracle.eps_greedy_prob(v, case["eps"], is_break_tie=bool(case["is_break_tie"])))
    rng = np.random.default_rng(4)
    for na, n in ((2, 4096), (3, 1000), (18, 333), (1, 5)):
        v = rng.integers(-2, 3, (na, n)).astype(np.float32) / 4  # many exact ties
        v[rng.random((na, n)) < 0.02] = np.nan
        mask = (rng.random((na, n)) < 0.7).astype(np.uint8)
        mask[rng.integers(0, na, n), np.arange(n)] = 1  # at least one legal action per env
        for eps in (0.0, 0.1, 0.9):
            for tie in (False, True):
                for m in (None, mask):
                    V = torch.as_tensor(v).cuda()
                    M = None if m is None else torch.as_tensor(m).cuda()
                    p = ops.eps_greedy_prob(V, eps, M, tie).cpu().numpy()
                    exp = np.stack([oracle.eps_greedy_prob(v[:, i], eps, None if m is None else m[:, i], tie) for i in range(n)], 1)
                    assert np.array_equal(p, exp, equal_nan=True), (na, n, eps, tie, m is not None)
                    # Julia (na, N) column-major storage = torch (n, na)
                    Vt = torch.as_tensor(np.ascontiguousarray(v.T)).cuda()
                    Mt = None if m is None else torch.as_tensor(np.ascontiguousarray(m.T)).cuda()
                    pt = ops.eps_greedy_prob(Vt, eps, Mt, tie, soa=False).cpu().numpy()
                    assert np.array_equal(pt.T, exp, equal_nan=True)
                    ok = ~np.isnan(v).any(0)  # (a NaN maximum has no `==` ties: find_all_max returns no index, as in Julia)
                    np.testing.assert_allclose(exp.sum(0)[ok], 1.0, atol=1e-12)
    ex = EpsilonGreedyExplorer(0.25, is_break_tie=True, seed=3)
    V = torch.as_tensor(np.array([[0.5, 0.1], [0.5, 0.3]], np.float32)).cuda()
    step = ex.step
    p = ex.prob(V).cpu().numpy()
    assert ex.step == step  # prob does not advance the explorer
    assert np.array_equal(p[:, 0], [0.5, 0.5]) and np.array_equal(p[:, 1], [0.125, 0.125 + 0.75])
