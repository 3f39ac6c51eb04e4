"""GPU parity of the device priority sum-tree (sumtree.hip) against the oracle: bit-exact trees, keys, logical
indices and priorities on the same seeded inputs; full-size (2^20 leaves) structural properties."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _dev_tree(n_leaves):
    import rlhip

    nodes = int(rlhip._lib.lib.rlhip_sumtree_nodes(n_leaves))
    return torch.zeros(nodes, dtype=torch.float32, device="cuda")


def _update(tree, n_leaves, keys, prio):
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    k = torch.as_tensor(np.asarray(keys, np.int64), device="cuda")
    p = torch.as_tensor(np.asarray(prio, np.float32), device="cuda")
    call("rlhip_sumtree_update", ptr(tree), n_leaves, ptr(k), ptr(p), k.numel(), stream_ptr())


def _sample(tree, n_leaves, batch, seed, ctr):
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    leaf = torch.empty(batch, dtype=torch.int64, device="cuda")
    prio = torch.empty(batch, dtype=torch.float32, device="cuda")
    call("rlhip_sumtree_sample", ptr(tree), n_leaves, batch, seed, ctr, ptr(leaf), ptr(prio), stream_ptr())
    return leaf.cpu().numpy(), prio.cpu().numpy()


@pytest.mark.parametrize("n_leaves", [1, 2, 5, 64, 1000, 4096, 70000])
@pytest.mark.parametrize("n_upd", [1, 37, 1024, 9000, 20000])
def test_update_bit_exact_with_duplicates(n_leaves, n_upd):
    rng = np.random.default_rng(n_leaves * 31 + n_upd)
    keys = rng.integers(-1, n_leaves + 1, n_upd)  # includes out-of-range keys (ignored) and duplicates
    prio = rng.random(n_upd).astype(np.float32) ** 0.6
    ref = oracle.SumTree(n_leaves)
    ref.update(keys, prio)
    tree = _dev_tree(n_leaves)
    _update(tree, n_leaves, keys, prio)
    assert np.array_equal(tree.cpu().numpy(), ref.tree)
    # a second update on top of the first
    keys2 = rng.integers(0, n_leaves, max(1, n_upd // 3))
    prio2 = rng.random(keys2.size).astype(np.float32)
    ref.update(keys2, prio2)
    _update(tree, n_leaves, keys2, prio2)
    assert np.array_equal(tree.cpu().numpy(), ref.tree)


@pytest.mark.parametrize("n_leaves,start,count", [(1, 0, 1), (300, 17, 200), (4096, 0, 4096), (100000, 99000, 1000),
                                                   (1 << 20, 0, 1 << 20), (1 << 20, 12345, 600000), (77, 5, 0)])
def test_fill_range_bit_exact(n_leaves, start, count):
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    ref = oracle.SumTree(n_leaves)
    ref.fill_range(start, count, 0.75)
    tree = _dev_tree(n_leaves)
    call("rlhip_sumtree_fill_range", ptr(tree), n_leaves, start, count, 0.75, stream_ptr())
    assert np.array_equal(tree.cpu().numpy(), ref.tree)


@pytest.mark.parametrize("n_leaves", [1, 5, 64, 1000, 1 << 16])
def test_sample_bit_exact(n_leaves):
    rng = np.random.default_rng(n_leaves)
    prio = (rng.random(n_leaves).astype(np.float32) ** 0.6) * (rng.random(n_leaves) < 0.7)  # ~30 % zeros
    if not prio.any():
        prio[0] = 1.0
    ref = oracle.SumTree(n_leaves)
    ref.update(np.arange(n_leaves), prio)
    tree = _dev_tree(n_leaves)
    _update(tree, n_leaves, np.arange(n_leaves), prio)
    for ctr in (0, 5):
        leaf, p = _sample(tree, n_leaves, 4096, 11, ctr)
        rl, rp = ref.sample(4096, 11, ctr)
        assert np.array_equal(leaf, rl)
        assert np.array_equal(p, rp)
        assert np.all(prio[leaf] > 0)  # a zero-priority leaf is never drawn


def test_sample_errors_and_empty():
    from rlhip._lib import RLHipError, call
    from rlhip.ops import ptr, stream_ptr

    tree = _dev_tree(8)
    out = torch.empty(0, dtype=torch.int64, device="cuda")
    call("rlhip_sumtree_sample", ptr(tree), 8, 0, 1, 0, ptr(out), None, stream_ptr())  # batch 0: no-op
    with pytest.raises(RLHipError):
        call("rlhip_sumtree_fill_range", ptr(tree), 8, 4, 5, 1.0, stream_ptr())  # range out of bounds
    with pytest.raises(RLHipError):
        call("rlhip_sumtree_fill_range", ptr(tree), 8, 0, 1, -1.0, stream_ptr())  # negative priority


def test_full_size_config5_properties():
    """2^20 leaves (BASELINE config 5), priorities U(0,1)^0.6: parents are exact child sums on every level,
    the root equals the fixed-order pairwise sum, sampled frequencies follow p / sum(p) on a coarse binning."""
    import rlhip
    from rlhip import ops

    n = 1 << 20
    u = ops.fill_uniform(n, 11, 0, 7)  # Philox SYNTH stream, seed 11
    prio = u.to(torch.float32) ** 0.6
    tree = _dev_tree(n)
    _update(tree, n, np.arange(n), prio.cpu().numpy())
    t = tree
    P = n
    level = t[P:2 * P]
    while level.numel() > 1:
        parent = level[0::2] + level[1::2]
        lo = level.numel() // 2
        assert torch.equal(parent, t[lo:2 * lo])
        level = parent
    draws = 1 << 20
    leaf, p = _sample(tree, n, draws, 7, 0)
    assert np.array_equal(p, prio.cpu().numpy()[leaf])
    bins = 64
    mass = prio.double().view(bins, -1).sum(1).cpu().numpy()
    cnt = np.bincount(leaf // (n // bins), minlength=bins)
    expect = mass / mass.sum() * draws
    chi2 = ((cnt - expect) ** 2 / expect).sum()
    assert chi2 < 140.0  # 63 dof


def test_prioritized_traces_round_trip_vs_oracle():
    """push -> default priority -> prioritized sample -> gather -> priority write-back, through the host mirror
    (CircularPrioritizedTraces / BatchSampler) against the oracle ring + sum-tree, with wrap-around."""
    import rlhip

    cap, n_env, od = 6, 3, 4
    tr = rlhip.CircularPrioritizedTraces(capacity=cap, n_env=n_env, obs_dim=od, default_priority=2.0)
    ring, st = oracle.Ring(cap, n_env, od), oracle.SumTree(cap * n_env)
    rng = np.random.default_rng(3)
    obs = rng.standard_normal((od, n_env)).astype(np.float32)
    tr.push_state_(torch.as_tensor(obs, device="cuda"))
    ring.push_state(obs)
    sampler = rlhip.BatchSampler(64, seed=5)
    for step in range(17):
        nobs = rng.standard_normal((od, n_env)).astype(np.float32)
        a = rng.integers(0, 2, n_env).astype(np.int32)
        r = rng.random(n_env).astype(np.float32)
        term = (rng.random(n_env) < 0.2).astype(np.uint8)
        tr.push_transition_(torch.as_tensor(nobs, device="cuda"), torch.as_tensor(a, device="cuda"),
                            torch.as_tensor(r, device="cuda"), torch.as_tensor(term, device="cuda"))
        ring.push_transition(nobs, a, r, term)
        oracle.ring_push_priority(ring, st, 2.0)
        batch = sampler.sample(tr)
        idx, key, prio = oracle.ring_sample_prioritized(ring, st, 64, 5, step)
        s, a_o, r_o, t_o, sn = ring.gather(idx)
        assert np.array_equal(batch["key"].cpu().numpy(), key)
        assert np.array_equal(batch["priority"].cpu().numpy(), prio)
        assert np.array_equal(batch["state"].cpu().numpy(), s)
        assert np.array_equal(batch["next_state"].cpu().numpy(), sn)
        assert np.array_equal(batch["action"].cpu().numpy(), a_o + 1)
        assert np.array_equal(batch["reward"].cpu().numpy(), r_o)
        # write back |td|-like priorities for the sampled keys (duplicates: last wins on both sides)
        newp = (np.abs(r_o) + 0.01).astype(np.float32)
        tr.set_priority_(batch["key"], torch.as_tensor(newp, device="cuda"))
        st.update(key, newp)
        assert np.array_equal(tr.priorities.cpu().numpy(), st.tree)
    assert abs(tr.total_priority() - float(st.tree[1])) == 0.0


@pytest.mark.parametrize("n_leaves,n_upd,spread", [(1 << 20, 4096, "uniform"), (1 << 20, 4096, "one_block"), (1 << 20, 4096, "few_blocks"),
                                                   (1 << 20, 32, "uniform"), (1 << 20, 512, "uniform"), (1 << 20, 9000, "one_block"),
                                                   (1 << 22, 4096, "uniform"), ((1 << 21) + 12345, 3000, "few_blocks"),
                                                   (1 << 14, 2048, "uniform"), (1 << 13, 5000, "uniform")])
def test_update_block_recompute_paths_bit_exact(n_leaves, n_upd, spread):
    """round 4 update kernel: election in LDS (a workgroup's share <= 1024 items) and on the leaf tags (above), whole
    128-leaf blocks recomputed by one wave, the path walk above the blocks for trees beyond 2^20 leaves -- every path
    against the sequential oracle, bit for bit, on a tree that already holds priorities everywhere"""
    rng = np.random.default_rng(n_upd + (n_leaves % 1000))
    ref = oracle.SumTree(n_leaves)
    tree = _dev_tree(n_leaves)
    base_k = np.arange(0, n_leaves, max(1, n_leaves // 50000))
    base_p = rng.random(base_k.size).astype(np.float32)
    ref.update(base_k, base_p)
    _update(tree, n_leaves, base_k, base_p)
    if spread == "uniform":
        keys = rng.integers(0, n_leaves, n_upd)
    elif spread == "one_block":  # every key inside one 128-leaf block: ONE workgroup owns all of them, heavy duplication
        keys = 77 * 128 + rng.integers(0, 128, n_upd)
    else:  # a handful of blocks, some keys repeated many times
        keys = rng.choice(rng.integers(0, n_leaves, 40), n_upd) + rng.integers(0, 3, n_upd)
        keys = np.minimum(keys, n_leaves - 1)
    prio = (rng.random(n_upd).astype(np.float32) + 0.01) ** 0.6
    ref.update(keys, prio)
    _update(tree, n_leaves, keys, prio)
    assert np.array_equal(tree.cpu().numpy(), ref.tree)
    _update(tree, n_leaves, keys[::-1].copy(), prio)  # the same keys in the opposite order: other winners
    ref.update(keys[::-1], prio)
    assert np.array_equal(tree.cpu().numpy(), ref.tree)
    assert tree[0].item() == 0.0  # the arrival counter is re-armed


@pytest.mark.parametrize("layout", ["frames_u8", "frames_u8_big_tree", "cartpole_f32", "wide_f32"])
def test_fused_sample_gather_equals_the_two_launches(layout):
    """rlhip_ring_sample_gather_prioritized: indices, keys, priorities and the gathered batch are bit-identical to
    rlhip_ring_sample_prioritized + rlhip_ring_gather (frame-major u8 ring, Float32 ring with 4 components = the fused
    kernels; 6 components = the documented two-launch route behind the same entry point)"""
    import rlhip

    if layout == "frames_u8":
        cap, n_env, od, dt = 300, 1, 84 * 84, torch.uint8
    elif layout == "frames_u8_big_tree":  # >= 2^12 leaves: a 13-level tree, drawn by a whole wavefront four levels per round trip (csrc/ring.hip)
        cap, n_env, od, dt = 5000, 1, 84 * 84, torch.uint8
    elif layout == "cartpole_f32":
        cap, n_env, od, dt = 64, 37, 4, torch.float32
    else:
        cap, n_env, od, dt = 64, 5, 6, torch.float32
    tr = rlhip.CircularPrioritizedTraces(capacity=cap, n_env=n_env, obs_dim=od, dtype=dt, default_priority=1.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    if dt == torch.uint8:
        tr.state.random_(0, 256, generator=g)
    else:
        (tr.records if tr.records_layout else tr.state).normal_(generator=g)
    tr.action.random_(0, 3, generator=g)
    tr.reward.normal_(generator=g)
    tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda", generator=g) < 0.1).to(torch.uint8))
    tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
    tr.rb.head_sa, tr.rb.head_rt = 5, 5  # a wrapped ring
    n = cap * n_env
    keys = torch.arange(n, dtype=torch.int64, device="cuda")
    tr.set_priority_(keys, torch.rand(n, device="cuda", generator=g) ** 0.6 + 0.01)
    for batch, ctr in ((1, 0), (33, 1), (512, 2), (4096, 3)):
        idx, key, prio = tr.sample_prioritized(batch, 9, ctr)
        ref = tr.gather(idx)
        (idx2, key2, prio2), got = tr.sample_gather_prioritized(batch, 9, ctr)
        assert torch.equal(idx, idx2) and torch.equal(key, key2) and torch.equal(prio, prio2)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)


def test_is_weights_bit_exact_and_weighted_gradients_match_oracle():
    """rlhip_per_is_weights_f32 against the oracle (bit-exact: Float64 power, rounded once, one division), and the three DQN
    gradient paths with weights -- 2-layer (rlhip_dqn_grad_idx_w_f32), 3-layer h = 128 and h = 256 (rlhip_dqn3_grad_w_f32)
    -- against oracle.dqn*_loss_grad(weights = w): same bars as the unweighted cases; weights of 1 reproduce the unweighted
    entry points bit for bit"""
    import ctypes as C

    import rlhip
    from conftest import BF16_GRAD_TOL, F32_GRAD_TOL, assert_grad_close
    from rlhip import dqn
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    rng = np.random.default_rng(12)
    for n in (1, 32, 1000, 4096, 70000):
        prio = ((rng.random(n).astype(np.float32) + 1e-4) ** 0.6).astype(np.float32)
        for beta in (0.0, 0.4, 1.0):
            w, prio_d = torch.empty(n, device="cuda"), torch.as_tensor(prio, device="cuda")
            call("rlhip_per_is_weights_f32", ptr(prio_d), n, beta, ptr(w), stream_ptr())
            assert np.array_equal(w.cpu().numpy(), oracle.per_is_weights(prio, beta)), (n, beta)
    ns, na, n_env, cap, batch = 4, 2, 64, 40, 1000
    traces = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    obs = rng.standard_normal((ns, n_env)).astype(np.float32)
    traces.push_state_(torch.as_tensor(obs, device="cuda"))
    oring.push_state(obs)
    for _ in range(57):
        nobs = rng.standard_normal((ns, n_env)).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = (rng.standard_normal(n_env) * 2).astype(np.float32)
        term = (rng.random(n_env) < 0.2).astype(np.uint8)
        traces.push_transition_(torch.as_tensor(nobs, device="cuda"), torch.as_tensor(a, device="cuda"),
                                torch.as_tensor(r, device="cuda"), torch.as_tensor(term, device="cuda"))
        oring.push_transition(nobs, a, r, term)
    idx = oring.sample_indices(batch, 7, 3)
    s, a, r, t, sn = oring.gather(idx)
    idx_d = torch.as_tensor(idx, device="cuda")
    w = oracle.per_is_weights((rng.random(batch).astype(np.float32) + 1e-3) ** 0.6, 0.5)
    w_d = torch.as_tensor(w, device="cuda")
    ones = torch.ones(batch, device="cuda")
    # ---- 2-layer Q-network (f32 VALU path)
    h = 128
    p = (rng.standard_normal(oracle.mlp2_nparams(ns, h, na)) * 0.3).astype(np.float32)
    tp = (p + rng.standard_normal(p.size).astype(np.float32) * 0.1).astype(np.float32)
    pd, tpd = torch.as_tensor(p, device="cuda"), torch.as_tensor(tp, device="cuda")
    ws = dqn.dqn_workspace(ns, h, na, batch)
    g, loss, td = torch.empty_like(pd), torch.empty(1, device="cuda"), torch.empty(batch, device="cuda")

    def grad2(weights):
        call("rlhip_dqn_grad_idx_w_f32", C.byref(traces.rb), h, na, 0, ptr(pd), ptr(tpd), batch, ptr(idx_d), ptr(weights), 0.99,
             1.0, ptr(ws), ptr(g), ptr(loss), ptr(td), stream_ptr())
        return g.cpu().numpy().copy(), float(loss), td.cpu().numpy().copy()

    gw, lw, tdw = grad2(w_d)
    ol, og = oracle.dqn_loss_grad(ns, h, na, 0, p, tp, s, a, r, t, sn, 0.99, 1.0, weights=w)
    assert abs(lw - ol) <= 2e-6 * max(1.0, abs(ol))
    assert_grad_close(gw, og, F32_GRAD_TOL, "weighted dqn_grad h=128")
    g1, l1, td1 = grad2(ones)
    call("rlhip_dqn_grad_idx_f32", C.byref(traces.rb), h, na, 0, ptr(pd), ptr(tpd), batch, ptr(idx_d), 0.99, 1.0, ptr(ws), ptr(g),
         ptr(loss), ptr(td), stream_ptr())
    assert np.array_equal(g.cpu().numpy(), g1) and float(loss) == l1 and np.array_equal(td.cpu().numpy(), td1)
    assert np.array_equal(tdw, td1)  # |Q(s, a) - y| is reported unweighted
    # ---- 3-layer Q-networks on the MFMA: hidden 128 (dqn3.hip, both tile kernels) and 256 (ppo3w.hip)
    for hh, bb in ((128, 1000), (128, 20000), (256, 1000)):
        idx3 = oring.sample_indices(bb, 7, 5)
        s3, a3, r3, t3, sn3 = oring.gather(idx3)
        w3 = oracle.per_is_weights((rng.random(bb).astype(np.float32) + 1e-3) ** 0.6, 0.5)
        p3, tp3 = oracle.mlp3_init(ns, hh, na, 11, 0), oracle.mlp3_init(ns, hh, na, 12, 0)
        p3d, tp3d = torch.as_tensor(p3, device="cuda"), torch.as_tensor(tp3, device="cuda")
        pk, tpk = dqn.mlp3_pack(p3d, ns, hh, na), dqn.mlp3_pack(tp3d, ns, hh, na)
        ws3 = dqn.dqn3_workspace(ns, hh, na, bb)
        g3, td3 = torch.empty_like(p3d), torch.empty(bb, device="cuda")
        idx3_d, w3_d = torch.as_tensor(idx3, device="cuda"), torch.as_tensor(w3, device="cuda")  # (named: they must outlive the call)
        call("rlhip_dqn3_grad_w_f32", C.byref(traces.rb), hh, na, 0, ptr(p3d), ptr(pk), ptr(tp3d), ptr(tpk), bb,
             ptr(idx3_d), ptr(w3_d), 0.99, 1.0, ptr(ws3), ptr(g3), ptr(loss), ptr(td3), stream_ptr())
        ol3, og3, _ = oracle.dqn3_loss_grad(ns, hh, na, 0, p3, tp3, s3, a3, r3, t3, sn3, 0.99, 1.0, weights=w3)
        assert abs(float(loss) - ol3) <= 2e-5 * max(1.0, abs(ol3)), (hh, bb)
        o = 0
        for name, sz in (("W1", hh * ns), ("b1", hh), ("W2", hh * hh), ("b2", hh), ("W3", na * hh), ("b3", na)):
            assert_grad_close(g3.cpu().numpy()[o:o + sz], og3[o:o + sz], BF16_GRAD_TOL, f"weighted dqn3 h={hh} b={bb} {name}")
            o += sz


@pytest.mark.parametrize("n_leaves", [2, 3, 96, 1 << 20, (1 << 20) - 5])
@pytest.mark.parametrize("n_upd", [1, 2, 3, 31, 32, 33, 63, 64])
def test_small_batch_update_kernel_bit_exact(n_leaves, n_upd):
    """rlhip_sumtree_update with <= 64 keys takes sumtree_update_small_kernel (one wavefront: sort, election of the last
    duplicate, the winners' paths walked up together with neighbouring paths merging) -- against the oracle's sequential
    update, on key patterns that exercise the merging: sibling leaves, whole blocks, one leaf repeated, both ends of the tree,
    out-of-range keys, ascending and descending order; on top of a filled tree so that untouched siblings carry mass"""
    rng = np.random.default_rng(n_leaves * 7 + n_upd)
    ref = oracle.SumTree(n_leaves)
    tree = _dev_tree(n_leaves)
    base_k = np.arange(n_leaves)[:: max(1, n_leaves // 5000)]
    base_p = rng.random(base_k.size).astype(np.float32) + 0.05
    ref.update(base_k, base_p)
    _update(tree, n_leaves, base_k, base_p)  # (the general kernel: more than 64 keys, or the small one for tiny trees)
    assert np.array_equal(tree.cpu().numpy(), ref.tree)
    c = int(rng.integers(0, n_leaves))
    patterns = {
        "random": rng.integers(0, n_leaves, n_upd),
        "with out-of-range": rng.integers(-3, n_leaves + 3, n_upd),
        "one block": (c // 64 * 64 + rng.integers(0, 64, n_upd)) % n_leaves,
        "consecutive ascending": (c + np.arange(n_upd)) % n_leaves,
        "consecutive descending": (c - np.arange(n_upd)) % n_leaves,
        "one leaf": np.full(n_upd, c),
        "both ends": np.where(np.arange(n_upd) % 2 == 0, np.arange(n_upd) // 2, n_leaves - 1 - np.arange(n_upd) // 2) % n_leaves,
        "sibling pairs": (2 * rng.integers(0, max(1, n_leaves // 2), n_upd) + (np.arange(n_upd) % 2)) % n_leaves,
    }
    for name, keys in patterns.items():
        prio = (rng.random(n_upd).astype(np.float32) ** 0.6) * (rng.random(n_upd) < 0.9)  # some zero priorities
        ref.update(keys, prio)
        _update(tree, n_leaves, keys, prio)
        assert np.array_equal(tree.cpu().numpy(), ref.tree), name


@pytest.mark.parametrize("cap,n_upd,batch", [(300, 32, 32), (5000, 64, 32), (5000, 1, 512), (300, 7, 4096), (5000, 65, 32), (5000, 0, 32)])
def test_fused_update_sample_gather_equals_the_two_calls_and_the_oracle(cap, n_upd, batch):
    """rlhip_ring_update_sample_gather_prioritized (round 6): the previous batch's priority write-back, the draw and the frame gather in
    ONE launch for <= 64 keys (n_upd = 65 / 0: the documented two-call route behind the same entry point).  Tree, indices, keys,
    priorities and the gathered batch bit-identical to set_priority_ + sample_gather_prioritized on a copy of the same ring, and to
    the oracle's SumTree.update + ring_sample_prioritized; duplicates among the keys (the last one wins); repeated calls re-arm the
    two sync words."""
    import rlhip

    od = 84 * 84
    g = torch.Generator(device="cuda").manual_seed(cap + n_upd)
    rng = np.random.default_rng(cap + n_upd + batch)
    trs = []
    for _ in range(2):
        tr = rlhip.CircularPrioritizedTraces(capacity=cap, n_env=1, obs_dim=od, dtype=torch.uint8, default_priority=1.0)
        trs.append(tr)
    a, b = trs
    a.state.random_(0, 256, generator=g)
    a.action.random_(0, 3, generator=g)
    a.reward.normal_(generator=g)
    a.terminal.copy_((torch.rand(a.terminal.shape, device="cuda", generator=g) < 0.1).to(torch.uint8))
    for name in ("state", "action", "reward", "terminal"):
        getattr(b, name).copy_(getattr(a, name))
    prio0 = torch.rand(cap, device="cuda", generator=g) + 0.05
    keys0 = torch.arange(cap, dtype=torch.int64, device="cuda")
    for tr in trs:
        tr.rb.len_sa, tr.rb.len_rt, tr.rb.head_rt, tr.rb.head_sa = cap + 1, cap, 17 % cap, 17 % (cap + 1)
        tr.set_priority_(keys0, prio0)
    ost = oracle.SumTree(cap)
    ost.update(np.arange(cap, dtype=np.int64), prio0.cpu().numpy())
    oring = oracle.Ring(cap, 1, 1)
    oring.rb.len_sa, oring.rb.len_rt, oring.rb.head_rt, oring.rb.head_sa = cap + 1, cap, 17 % cap, 17 % (cap + 1)
    assert np.array_equal(a.priorities.cpu().numpy(), ost.tree)
    for call_no in range(3):
        uk = rng.integers(0, cap, n_upd).astype(np.int64)
        if n_upd >= 4:
            uk[-1] = uk[0]          # a duplicate: the last occurrence wins
            uk[1] = -1 if call_no == 1 else uk[1]   # an out-of-range key is ignored (as rlhip_sumtree_update)
        up = (rng.random(n_upd) * 3 + 0.01).astype(np.float32)
        dk, dp = torch.as_tensor(uk).cuda(), torch.as_tensor(up).cuda()
        (i1, k1, p1), got1 = a.update_sample_gather_prioritized(dk if n_upd else None, dp if n_upd else None, batch, 9, call_no)
        if n_upd:
            b.set_priority_(dk, dp)
        (i2, k2, p2), got2 = b.sample_gather_prioritized(batch, 9, call_no)
        assert torch.equal(a.priorities, b.priorities), f"call {call_no}: the trees differ"
        assert torch.equal(i1, i2) and torch.equal(k1, k2) and torch.equal(p1, p2)
        for x, y in zip(got1, got2):
            assert torch.equal(x, y)
        if n_upd:
            ost.update(uk, up)
        assert np.array_equal(a.priorities.cpu().numpy(), ost.tree)
        oi, ok_, op = oracle.ring_sample_prioritized(oring, ost, batch, 9, call_no)
        assert np.array_equal(i1.cpu().numpy(), oi) and np.array_equal(k1.cpu().numpy(), ok_) and np.array_equal(p1.cpu().numpy(), op)
        assert a._sync.tolist() == [0, 0]
