"""Runs LAST (file name): the distribution of the bf16 MFMA gradient errors over everything this pytest session checked
(tests/conftest.py: assert_grad_close logs every check).  The per-check bar has to admit a relu / rounding decision flip
(a few 1e-4 of max|g|); the typical check sits three orders of magnitude lower, and that is what an arithmetic regression
would move -- so the suite-level bar is on the median and on the share of checks that show a flip at all."""
import json
import os

import numpy as np
import pytest

from conftest import (BF16_FLIP_LEVEL, BF16_FLIP_SHARE, BF16_GRAD_TOL, BF16_MEDIAN_TOL, F32_GRAD_TOL, GRAD_ERR_LOG,
                      SESSION_ID)

pytestmark = pytest.mark.gpu
_collected_files = set()


@pytest.fixture(autouse=True)
def _note_collected(request):
    _collected_files.update(os.path.basename(str(i.fspath)) for i in request.session.items)


def test_bf16_gradient_error_distribution_of_this_session():
    if not os.path.exists(GRAD_ERR_LOG):
        pytest.skip("no gradient checks were logged")
    rows = [json.loads(ln) for ln in open(GRAD_ERR_LOG) if ln.strip()]
    rows = [r for r in rows if r.get("session") == SESSION_ID]
    bf = np.array([r["err_over_max"] for r in rows if r["tol"] == BF16_GRAD_TOL])
    f32 = np.array([r["err_over_max"] for r in rows if r["tol"] == F32_GRAD_TOL])
    if bf.size < 100:
        # a subset run (one file, -k): the distribution bar is defined over the whole suite; every check of the subset still
        # carried its own max + bulk bars (conftest.assert_grad_close) and the tight cases theirs.  As part of the FULL suite
        # (all four bf16 learner files collected) too few checks is a failure, not a skip.
        full = {"test_gpu_dqn3.py", "test_gpu_dqn3w.py", "test_gpu_ppo3.py", "test_gpu_ppo3w.py"} <= _collected_files
        assert not full, f"full GPU suite but only {bf.size} bf16 gradient checks were logged"
        pytest.skip(f"only {bf.size} bf16 gradient checks in this session (subset run)")
    share = float((bf > BF16_FLIP_LEVEL).mean())
    print(f"bf16 checks {bf.size}: median {np.median(bf):.2e}, share above {BF16_FLIP_LEVEL:.0e}: {share:.3f}, max {bf.max():.2e}; "
          f"f32 checks {f32.size}: max {f32.max() if f32.size else 0:.2e}")
    assert np.median(bf) <= BF16_MEDIAN_TOL, f"median bf16 gradient error {np.median(bf):.2e} > {BF16_MEDIAN_TOL:.0e}"
    assert share <= BF16_FLIP_SHARE, f"{share:.1%} of the bf16 checks show an error above {BF16_FLIP_LEVEL:.0e}"
