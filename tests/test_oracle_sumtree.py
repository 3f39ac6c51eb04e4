"""CPU tests of the oracle's priority sum-tree (oracle/rlo_buffer.c): the published SumTree algorithm of the
un-vendored CircularArrayBuffers 0.1.12 restated -- hand-computed known answers, structural invariants and the
sampling law P(i) = p_i / sum(p).  (Parity unpinned upstream: there is no SumTree test in /root/reference.)"""
import numpy as np
import pytest

import oracle


def test_tree_layout_and_hand_computed_sums():
    st = oracle.SumTree(5)  # P = 8: heap of 16 floats, leaves at 8..12
    assert st.tree.size == 16 and st.P == 8
    st.update([0, 1, 2, 3, 4], [1.0, 2.0, 3.0, 4.0, 5.0])
    t = st.tree
    assert list(t[8:13]) == [1, 2, 3, 4, 5] and list(t[13:16]) == [0, 0, 0]
    assert list(t[4:8]) == [3, 7, 5, 0]
    assert list(t[2:4]) == [10, 5]
    assert t[1] == 15


def test_descent_hand_cases():
    # leaves 1,2,3,4,5: cumulative 1,3,6,10,15.  v <= left goes left (so v = 1 -> leaf 0, v = 3 -> leaf 1).
    st = oracle.SumTree(5)
    st.update(np.arange(5), [1.0, 2.0, 3.0, 4.0, 5.0])
    import ctypes as C

    lib = oracle.binding.lib()

    def leaf_for(v):
        # reproduce the walk in python on the oracle's tree
        node, P, t = 1, st.P, st.tree
        v = np.float32(v)
        while node < P:
            l, r = t[2 * node], t[2 * node + 1]
            right = (v > l and r > 0) or l == 0
            if right:
                v = np.float32(v - l)
            node = 2 * node + int(right)
        return node - P

    assert [leaf_for(v) for v in (0.0, 1.0, 1.5, 3.0, 3.5, 6.0, 9.9, 10.0, 10.5, 15.0)] == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]
    # the C sampler follows the same walk
    leaf, prio = st.sample(4096, 3, 0)
    for b in (0, 1, 17, 4095):
        w = oracle.philox(3, b, 0, 0, oracle.TAG["SAMPLER"])
        v = np.float32(oracle.u01_f32(w[2])) * st.tree[1]
        assert leaf[b] == leaf_for(v)
        assert prio[b] == st.leaves()[leaf[b]]


def test_parents_are_exact_child_sums_after_any_update_order():
    rng = np.random.default_rng(0)
    n = 1000
    a, b = oracle.SumTree(n), oracle.SumTree(n)
    keys = rng.integers(0, n, 5000)
    pr = rng.random(5000).astype(np.float32)
    a.update(keys, pr)
    # same final leaf values written in a different order / in pieces
    last = {}
    for k, p in zip(keys, pr):
        last[int(k)] = p
    ks = np.array(sorted(last), np.int64)
    b.update(ks[::-1], np.array([last[int(k)] for k in ks[::-1]], np.float32))
    assert np.array_equal(a.tree, b.tree)
    t = a.tree
    for node in range(1, a.P):
        assert t[node] == np.float32(t[2 * node] + t[2 * node + 1])


def test_duplicate_keys_last_wins_and_out_of_range_ignored():
    st = oracle.SumTree(8)
    st.update([3, 3, 3, -1, 8, 5], [1.0, 2.0, 7.0, 9.0, 9.0, 4.0])
    assert list(st.leaves()) == [0, 0, 0, 7, 0, 4, 0, 0]
    assert st.tree[1] == 11


def test_fill_range_matches_update():
    a, b = oracle.SumTree(300), oracle.SumTree(300)
    a.fill_range(17, 200, 2.5)
    b.update(np.arange(17, 217), np.full(200, 2.5, np.float32))
    assert np.array_equal(a.tree, b.tree)
    a.fill_range(0, 0, 1.0)
    assert np.array_equal(a.tree, b.tree)


def test_sampling_law_and_zero_priority_never_sampled():
    n = 64
    st = oracle.SumTree(n)
    p = np.zeros(n, np.float32)
    p[::2] = np.arange(1, 33, dtype=np.float32)  # odd leaves have zero priority
    st.update(np.arange(n), p)
    draws = 400_000
    leaf, prio = st.sample(draws, 9, 1)
    assert np.all(p[leaf] > 0)
    assert np.array_equal(prio, p[leaf])
    cnt = np.bincount(leaf, minlength=n).astype(np.float64)
    expect = p.astype(np.float64) / p.sum() * draws
    nz = expect > 0
    chi2 = ((cnt[nz] - expect[nz]) ** 2 / expect[nz]).sum()
    assert chi2 < 80.0  # 31 dof: P(chi2 > 80) < 1e-5


def test_single_leaf_and_all_zero_tree():
    st = oracle.SumTree(1)
    st.update([0], [3.0])
    leaf, prio = st.sample(10, 0, 0)
    assert np.all(leaf == 0) and np.all(prio == 3.0)
    z = oracle.SumTree(6)
    leaf, prio = z.sample(10, 0, 0)  # total 0: lands on the last leaf (documented), priority 0
    assert np.all(prio == 0) and np.all((leaf >= 0) & (leaf < 6))


def test_ring_prioritized_keys_follow_the_wrap():
    ring = oracle.Ring(4, 2, 1)
    st = oracle.SumTree(4 * 2)
    ring.push_state(np.zeros((1, 2), np.float32))
    for step in range(7):  # wraps: 7 pushes into capacity 4
        ring.push_transition(np.full((1, 2), step + 1.0, np.float32), [step, step], [float(step)] * 2, [0, 0])
        oracle.ring_push_priority(ring, st, 1.0 + step)
    # physical slots hold steps {4,5,6,3}; leaves = default priority of the step stored there
    assert list(st.leaves()) == [5, 5, 6, 6, 7, 7, 4, 4]
    idx, key, prio = oracle.ring_sample_prioritized(ring, st, 256, 1, 0)
    s, a, r, t, sn = ring.gather(idx)
    # the transition gathered through the logical index is the one whose leaf was drawn
    assert np.array_equal(prio, 1.0 + a.astype(np.float32))
    assert np.array_equal(key % 2, idx % 2)
