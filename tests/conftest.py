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
# oracle): measured <= 9.4e-8 (gpurun_out/grad_err.jsonl), bar 1e-6.
# The bf16 MFMA paths round operands to bf16 exactly like the oracle does and differ only by the MFMA's internal
# summation order -- EXCEPT where that last-bit difference lands on a decision: a pre-activation within an ulp of zero
# (relu' flips for one (sample, unit) pair), a bf16 rounding tie, an ocml-vs-glibc tanh ulp.  A flip moves the rows /
# columns of ONE hidden unit by up to a few 1e-4 of max|g| and leaves every other element at the 1e-7 level.  So the bar
# has two parts (round-2 verdict: one number for both hid a 2x regression and sat 1.09x above the worst flip):
#   bulk   the BF16_BULK_Q quantile of |g - o| / max|o| stays below BF16_BULK_TOL  -- arithmetic regressions move this
#   flips  the elements above the bulk bar are few (a unit's share of the tensor) and below BF16_FLIP_TOL -- a wrong
#          unit / a wrong tile moves them by O(1e-1)
# Every check appends what it measured (max, quantiles, count over the bulk bar) to gpurun_out/grad_err.jsonl (scratch)
# so that the margins are known numbers, not guesses; the values measured in round 3 are in profiles/r03_parity_margins.md.
F32_GRAD_TOL = 1e-6
BF16_GRAD_TOL = 5e-4   # kept as the name the tests pass; assert_grad_close applies the two-part bar for it
BF16_BULK_Q = 0.99
BF16_BULK_TOL = 2e-5
BF16_FLIP_TOL = 2e-3
BF16_FLIP_SHARE = 0.01


def assert_grad_close(g, o, tol, tag=""):
    import json

    import numpy as np

    g = np.asarray(g, np.float64).reshape(-1)
    o = np.asarray(o, np.float64).reshape(-1)
    assert g.shape == o.shape, (g.shape, o.shape)
    assert np.isfinite(g).all(), f"{tag}: non-finite gradient"
    scale = max(float(np.abs(o).max()), 1e-30)
    e = np.abs(g - o) / scale
    err = float(e.max())
    two_part = tol == BF16_GRAD_TOL
    bulk = float(np.quantile(e, BF16_BULK_Q))
    n_over = int((e > BF16_BULK_TOL).sum())
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "grad_err.jsonl"), "a") as f:
            f.write(json.dumps({"tag": tag, "n": int(g.size), "err_over_max": err, "tol": tol, "q99": bulk,
                                "q999": float(np.quantile(e, 0.999)), "n_over_bulk": n_over}) + "\n")
    except OSError:
        pass
    if not two_part:
        assert err <= tol, f"{tag}: max|g - o| / max|o| = {err:.3e} > {tol:.1e}"
        return
    assert bulk <= BF16_BULK_TOL, f"{tag}: {BF16_BULK_Q} quantile of |g - o| / max|o| = {bulk:.3e} > {BF16_BULK_TOL:.1e}"
    assert err <= BF16_FLIP_TOL, f"{tag}: max|g - o| / max|o| = {err:.3e} > {BF16_FLIP_TOL:.1e} (no decision flip is that large)"
    assert n_over <= max(4, int(BF16_FLIP_SHARE * g.size)), \
        f"{tag}: {n_over} of {g.size} elements above {BF16_BULK_TOL:.1e} -- more than the rows of a few flipped units"
