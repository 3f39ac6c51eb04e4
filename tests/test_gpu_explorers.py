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
