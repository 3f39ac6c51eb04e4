"""pytest configuration: `gpu` marker, import paths.

CPU suite:  python -m pytest tests/ -x -q -m "not gpu"
GPU suite:  python -m pytest tests/ -x -q -m gpu      (needs an MI355X; runs through the C-ABI)
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "reinforcementlearning.jl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (HIP kernels through the C-ABI)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------------------------- gradient parity bar
# north_star: "within 1e-5 rel for fp32".  For a gradient vector the meaningful relative measure is against the
# largest entry of the tensor (entries near zero are sums with cancellation): |g - o| <= tol * max|o|.
# F32_GRAD_TOL is the bar of every f32 VALU path (fixed summation order on the GPU, double accumulation in the
# oracle): measured <= 9.4e-8 (17 checks), bar 1e-6.
# The bf16 MFMA paths round operands to bf16 exactly like the oracle does and differ only by the MFMA's internal
# summation order -- EXCEPT where that last-bit difference lands on a decision: a layer-2 pre-activation within an ulp of
# zero (relu' flips for one (sample, unit) pair), a bf16 rounding tie, an ocml-vs-glibc tanh ulp.  A flip changes ONE
# sample's backward pass: the flipped unit's W2 row by up to a few 1e-4 of max|g|, and -- through dH1 -- every layer-1
# element by ~1 / batch of it.  Measured over the 420 bf16 checks of the GPU suite (round 3, profiles/r03_parity_margins.md):
# median of max|g - o| / max|o| 1.5e-7, 5 % of the checks above 2e-5 (all with the flip signature), worst 1.7e-4
# (round 2 saw 4.6e-4 once).  Hence two bars (round-2 verdict: one number for both hid a 2x regression and sat 1.09x above
# the worst flip):
#   per check   max|g - o| / max|o| <= BF16_GRAD_TOL = 2^-8 = 3.9e-3: ONE bf16 ulp.  A flipped rounding of dz2[s][j] moves
#               column j of dW2 by ulp(dz2[s][j]) x h1[s][:], i.e. by at most one bf16 ulp of that sample's share of max|g|
#               -- when one sample dominates a small micro-batch (432 samples, an advantage of -65 against a typical -5:
#               tests/test_gpu_ppo3w.py pendulum / tanh, measured 2.6e-3 with the signature "one column, every row")
#               the share approaches 1.  A wrong unit / wrong tile is O(1e-1): 25x above the bar.
#   per suite   tests/test_gpu_zz_margins.py: the MEDIAN over all bf16 checks of a run <= BF16_MEDIAN_TOL = 5e-7 (3.3x
#               measured) and at most BF16_FLIP_SHARE = 15 % of them above 2e-5 (3x measured) -- an arithmetic regression
#               moves every check, a decision flip moves a few
# Round 4: the max bar alone cannot see a wrong tile / fragment slot at 1e-3 of max|g| in ONE instantiation,
# so (a) every bf16 check also asserts its BULK: tensors of >= 4096 elements (the hidden x hidden matrices, where a wrong
# fragment slot is at least 1/64 of the elements) q99 <= BF16_Q99_TOL = 1e-4 (measured <= 1.5e-5 outside one conditioning
# outlier); smaller tensors (sums over all samples: a flip reaches every element through dH1, q99 ~ max) max <= BF16_SMALL_TOL
# = 1e-3 (measured <= 4.6e-4); and (b) tests/test_gpu_bf16_tight.py gives every instantiation a decision-free case with the
# per-check bars 1e-6 (relu: exact forward) / 5e-5 max + 1e-5 q99 (tanh) (profiles/r04_parity_margins.md).  2^-8 stays the max bar of the flip-prone cases only.
# Every check appends what it measured to gpurun_out/grad_err.jsonl (scratch), tagged with the pytest session.
import uuid

SESSION_ID = uuid.uuid4().hex[:12]
F32_GRAD_TOL = 1e-6
BF16_GRAD_TOL = 2.0 ** -8
BF16_TIGHT_TOL = 5e-5        # decision-proof tanh cases (tests/test_gpu_bf16_tight.py): max; measured <= 1.4e-5 (isolated roundings)
BF16_TIGHT_Q99_TOL = 1e-5    # ... and their bulk (tensors of >= 4096 elements): measured <= 3.3e-6
BF16_TIGHT_RELU_TOL = 1e-6   # decision-proof relu cases: everything up to the head outputs is exact on both sides; measured <= 1.6e-7
BF16_Q99_TOL = 1e-4
BF16_SMALL_TOL = 1e-3
BF16_BULK_MIN_N = 4096
BF16_MEDIAN_TOL = 5e-7
BF16_FLIP_LEVEL = 2e-5
BF16_FLIP_SHARE = 0.15
GRAD_ERR_LOG = os.path.join(ROOT, "gpurun_out", "grad_err.jsonl")


def assert_grad_close(g, o, tol, tag="", q99_tol=None):
    import json

    import numpy as np

    g = np.asarray(g, np.float64).reshape(-1)
    o = np.asarray(o, np.float64).reshape(-1)
    assert g.shape == o.shape, (g.shape, o.shape)
    assert np.isfinite(g).all(), f"{tag}: non-finite gradient"
    scale = max(float(np.abs(o).max()), 1e-30)
    e = np.abs(g - o) / scale
    err = float(e.max())
    q99 = float(np.quantile(e, 0.99))
    try:
        os.makedirs(os.path.dirname(GRAD_ERR_LOG), exist_ok=True)
        with open(GRAD_ERR_LOG, "a") as f:
            f.write(json.dumps({"session": SESSION_ID, "tag": tag, "n": int(g.size), "err_over_max": err, "tol": tol,
                                "q99": q99, "n_over_2e-5": int((e > BF16_FLIP_LEVEL).sum())}) + "\n")
    except OSError:
        pass
    assert err <= tol, f"{tag}: max|g - o| / max|o| = {err:.3e} > {tol:.1e}"
    if tol == BF16_TIGHT_TOL and g.size >= BF16_BULK_MIN_N:
        assert q99 <= BF16_TIGHT_Q99_TOL, f"{tag}: q99 of |g - o| / max|o| = {q99:.3e} > {BF16_TIGHT_Q99_TOL:.0e} (n = {g.size})"
    if tol == BF16_GRAD_TOL:  # the flip-prone bf16 cases: the max bar admits a flipped decision, the bulk must not move
        if g.size >= BF16_BULK_MIN_N:
            qt = BF16_Q99_TOL if q99_tol is None else q99_tol  # (one documented override: tests/test_gpu_ppo3w.py)
            assert q99 <= qt, f"{tag}: q99 of |g - o| / max|o| = {q99:.3e} > {qt:.0e} (n = {g.size})"
        else:
            assert err <= BF16_SMALL_TOL, f"{tag}: max|g - o| / max|o| = {err:.3e} > {BF16_SMALL_TOL:.0e} (n = {g.size} < {BF16_BULK_MIN_N})"
