"""The persistent whole-update kernel (csrc/ppo_persist.hip) against the two-launch-per-step path it replaces and
against the oracle: same tile code, same summation orders => the same bits whenever both run the same grid.

Every case runs in a subprocess (the switches RLHIP_PPO_PERSIST / RLHIP_PERSIST_* are read once per process)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (kind, continuous, n, T, hidden, act, n_microbatches): headline shape first; then every template axis of the kernel
# (NS = 4 / 3 / 2, relu / tanh, 2 / 3 actor outputs, one / two teams per workgroup, ragged micro-batch, several slices
# per workgroup when the grid is small)
CASES = [
    ("cartpole", False, 4096, 32, 256, 0, 4),
    ("pendulum", True, 1024, 16, 256, 0, 4),
    ("pendulum", False, 512, 8, 64, 1, 2),
    ("mountaincar", False, 2048, 16, 128, 1, 4),
    ("cartpole", False, 256, 16, 256, 1, 4),
    ("cartpole", False, 100, 7, 96, 0, 3),
]

DRIVER = r"""
import hashlib, json, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r)
import torch
import rlhip
out = []
for (kind, cont, n, T, hidden, act, nmb) in %(cases)r:
    env = rlhip.HipVecEnv(kind, n, seed=5, continuous=cont)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=hidden, act=act, n_microbatches=nmb, seed=5)
    h = hashlib.sha256()
    for it in range(%(iters)d):
        pol.rollout_()
        pol.update_()
    torch.cuda.synchronize()
    for t in (pol.params, pol.m, pol.v, pol.beta_pow, pol.grad, pol.losses):
        h.update(t.cpu().numpy().tobytes())
    out.append({"hash": h.hexdigest(), "status": pol.update_status(), "finite": bool(torch.isfinite(pol.params).all()),
                "loss": float(pol.losses[0])})
print("RESULT", json.dumps(out))
"""


def _run(cases, iters=2, **env):
    src = DRIVER % {"root": ROOT, "pkg": os.path.join(ROOT, "reinforcementlearning.jl_amd"), "cases": cases, "iters": iters}
    r = subprocess.run([sys.executable, "-c", src], env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    return json.loads(line[len("RESULT"):])


def test_persistent_update_is_bit_identical_to_the_two_launch_path():
    """params, Adam moments, beta powers, the last clipped gradient and the loss line after two rollout + update
    iterations: identical bytes with and without the persistent kernel, for every template axis"""
    a = _run(CASES, RLHIP_PPO_PERSIST="1")
    b = _run(CASES, RLHIP_PPO_PERSIST="0")
    for case, x, y in zip(CASES, a, b):
        assert x["status"] == 0 and x["finite"], (case, x)
        assert x["hash"] == y["hash"], (case, x, y)


def test_persistent_update_on_a_smaller_grid_tracks_the_full_grid():
    """RLHIP_PERSIST_MAX_GRID=24 (a device with 24 CUs): every workgroup walks several tiles and owns several parameter
    slices; the partial sums are grouped differently, so the results agree to rounding, not to the bit"""
    cases = CASES[:2]
    a = _run(cases, iters=1, RLHIP_PPO_PERSIST="1")
    b = _run(cases, iters=1, RLHIP_PPO_PERSIST="1", RLHIP_PERSIST_MAX_GRID="24")
    for case, x, y in zip(cases, a, b):
        assert y["status"] == 0 and y["finite"], (case, y)
        assert abs(x["loss"] - y["loss"]) <= 1e-5 * max(1.0, abs(x["loss"])), (case, x, y)


ORACLE_DRIVER = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(tests)r)
import numpy as np
import torch
import oracle
import rlhip

n, T = 4096, 32
env = rlhip.HipVecEnv("cartpole", n, seed=3)
pol = rlhip.PPOPolicy(env, update_freq=T, hidden=256, seed=3)
ocfg = oracle.ppo_default(continuous=0, hidden=256)
pol.rollout_()
p0 = pol.params.cpu().numpy().copy()
pol.update_()
assert pol.update_status() == 0
tr = pol.trajectory
otr = oracle.PPOTraj(0, n, T)
for name in ("obs", "logp", "value", "reward", "action_i", "terminal"):
    getattr(otr, name)[...] = getattr(tr, name).cpu().numpy()
oracle.ppo_gae(ocfg, otr)
po, mo, vo = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
steps, _ = oracle.ppo_update(0, ocfg, otr, po, mo, vo, 0, pol.seed, 0)
assert steps == 16
d = np.abs(pol.params.cpu().numpy() - po)
assert np.quantile(d, 0.99) < 2e-4, f"99th percentile |dp| = {np.quantile(d, 0.99):.2e}"
assert d.max() < 16 * 2 * 1e-3
assert np.abs(po - p0).max() > 1e-3
print("ORACLE_OK")
"""


def test_persistent_update_vs_oracle():
    """the whole 16-step update against the CPU oracle's update loop on the same trajectory"""
    src = ORACLE_DRIVER % {"root": ROOT, "pkg": os.path.join(ROOT, "reinforcementlearning.jl_amd"),
                           "tests": os.path.join(ROOT, "tests")}
    r = subprocess.run([sys.executable, "-c", src], env=dict(os.environ, RLHIP_PPO_PERSIST="1"), capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "ORACLE_OK" in r.stdout, r.stderr[-3000:]


def test_a_withheld_row_aborts_loudly_not_silently():
    """fault injection: workgroup 1 never publishes its last partial row -> its readers give up after the spin limit,
    the update ends (no hang), the parameters are NaN and the status word says RLHIP_ETIMEOUT"""
    r = _run(CASES[:1], iters=1, RLHIP_PPO_PERSIST="1", RLHIP_PERSIST_TEST_FAULT="2", RLHIP_PERSIST_SPIN_LIMIT="4096")
    assert r[0]["status"] == -4 and not r[0]["finite"], r
