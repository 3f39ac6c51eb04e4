"""The oracle's n-step sampler / fold (oracle/rlo_buffer.c: NStepBatchSampler of the un-vendored RLTrajectories 0.4 restated) against
an independent numpy statement of the same published algorithm, against the reference's own `discount_rewards_reduced`
(RLCore/src/utils/basic.jl:237-319, pinned by tests/golden/scans.json) for the return, and n = 1 against the 1-step gather."""
import numpy as np
import pytest

import oracle


def _ring(seed=0, cap=12, n_env=5, d=3, pushes=31, p_term=0.2):
    rng = np.random.default_rng(seed)
    ring = oracle.Ring(cap, n_env, d)
    frames = [rng.standard_normal((d, n_env)).astype(np.float32)]
    ring.push_state(frames[0])
    art = []
    for _ in range(pushes):
        f = rng.standard_normal((d, n_env)).astype(np.float32)
        a, r = rng.integers(0, 3, n_env).astype(np.int32), rng.standard_normal(n_env).astype(np.float32)
        t = (rng.random(n_env) < p_term).astype(np.uint8)
        ring.push_transition(f, a, r, t)
        frames.append(f)
        art.append((a, r, t))
    return ring, frames, art, pushes - cap  # logical transition li = push (first + li) -> (first + li + 1)


@pytest.mark.parametrize("n_step", [1, 2, 3, 7, 12])
def test_nstep_gather_matches_numpy_statement(n_step):
    ring, frames, art, first = _ring()
    cap, n_env = ring.rb.capacity, ring.rb.n_env
    gamma = np.float32(0.9)
    idx = oracle.ring_sample_indices_nstep(ring, 64, n_step, 3, 1)
    assert idx.min() >= 0 and idx.max() < (cap - n_step + 1) * n_env and (n_step == cap or len(set(idx // n_env)) > 1)
    idx = np.concatenate([idx, np.arange((cap - n_step + 1) * n_env)])   # and every valid start
    s, a, R, t, sn = oracle.ring_gather_nstep(ring, idx, n_step, float(gamma))
    for b, fj in enumerate(idx):
        li, e = fj // n_env, fj % n_env
        window = []
        for k in range(n_step):
            window.append(first + li + k)
            if art[first + li + k][2][e]:
                break
        ns = len(window)
        assert np.array_equal(s[:, b], frames[first + li][:, e]) and a[b] == art[first + li][0][e]
        assert np.array_equal(sn[:, b], frames[first + li + ns][:, e])
        assert t[b] == art[window[-1]][2][e] and t[b] == max(art[q][2][e] for q in window)
        gain = np.float32(0.0)
        for q in reversed(window):
            gain = np.float32(art[q][1][e] + np.float32(gamma * gain))
        assert R[b] == gain
        # the reference's own scan over the window (basic.jl:237-319)
        ref = oracle.discount_rewards_reduced(np.array([art[q][1][e] for q in window], np.float32), float(gamma), dtype=np.float32)
        assert R[b] == np.float32(ref)


def test_nstep_one_is_the_plain_gather():
    ring, *_ = _ring(seed=4)
    idx = ring.sample_indices(50, 9, 2)
    assert np.array_equal(idx, oracle.ring_sample_indices_nstep(ring, 50, 1, 9, 2))
    for x, y in zip(ring.gather(idx), oracle.ring_gather_nstep(ring, idx, 1, 0.99)):
        assert np.array_equal(x, y)


def test_gamma_pow():
    assert oracle.gamma_pow(0.99, 1) == np.float32(0.99)
    for n in (2, 3, 5, 10, 32):
        assert oracle.gamma_pow(0.99, n) == np.float32(np.float64(np.float32(0.99)) ** n)


def test_dqn_run_with_nstep_targets_runs():
    r = oracle.dqn_run(12, n=64, batch=32, capacity=16, n_step=3)
    assert r.n_updates == 10 and np.isfinite(r.params).all()   # the first two vec-steps hold fewer than 3 transitions
