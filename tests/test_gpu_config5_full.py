"""BASELINE configs[4] at its FULL size against the oracle ("next round" item 1a).

`bench.py` times the u8 frame kernels on a 2^20-slot ring of 84 x 84 x 4 frames (29.6 GB of states); until round 6 the only oracle
comparisons of those kernels used capacities 4 .. 37.  Here the full-size ring is filled THROUGH the ABI
(`rlhip_ring_push_state_maxpool` / `rlhip_ring_push_transition_maxpool` + `rlhip_ring_push_priority`, 2^20 + 777 transitions: past
its wrap-around), and every read path the bench times is compared bit for bit:

  * `rlhip_ring_sample_indices` + `rlhip_ring_gather`                 uniform BatchSampler, batch in {32, 512, 4096}
  * `rlhip_ring_sample_prioritized`, `rlhip_ring_sample_gather_prioritized`   (draw vs oracle.ring_sample_prioritized over the 2^20-leaf
    tree, gathered frames, both launch forms)
  * `rlhip_ring_gather_stacked` at n_stack = 4 across episode boundaries (StackFrames at sample time,
    RLCore/src/utils/stack_frames.jl:11-44)
  * `rlhip_sumtree_update` round trip (whole tree bit-exact against the oracle's, then a further draw)
  * explicit indices on both sides of the ring's wrap, of the physical end of the storage and of the 4 GiB byte offset
    (slot 152174 | 152175: every offset above is computed in 64 bits or read garbage).

Index arithmetic is the ORACLE's: an oracle.Ring of the same capacity receives the same pushes with a one-component "frame" that
holds the push number + 1, so its gather / stacked gather / prioritized draw say WHICH pushed frame (or the all-zero frame) every
output slot must hold.  Content is a function of the push number the host can evaluate -- frame(p) = max.(A[p mod 1021], B[p mod 1031])
of two pools of random screens (the AtariEnv 2-frame max-pool, RLEnvs/src/environments/3rd_party/atari.jl:104-107; 1021 x 1031 >
number of pushes, so no two pushes share a pair) -- so no host copy of the 29.6 GB exists.  Large batches are compared on the
device against torch.maximum of the pools at the oracle's ids (torch indexing, not the library); small ones also on the host in numpy.
"""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from test_gpu_bench_shapes import host, note  # noqa: E402

FB = 84 * 84 * 4          # bytes per stored state (BASELINE configs[4]: 84 x 84 x 4 u8)
CAP = 1 << 20
EXTRA = 777               # transitions pushed beyond the capacity
KA, KB = 1021, 1031       # coprime pool sizes, KA * KB > CAP + EXTRA + 1
N_PUSH = CAP + EXTRA      # transitions; push 0 is the PreEpisodeStage state


@pytest.fixture(scope="module")
def full():
    import rlhip
    from rlhip import _lib, ops
    from rlhip.ops import ptr, stream_ptr

    assert KA * KB > N_PUSH + 1
    rng = np.random.default_rng(11)
    A = rng.integers(0, 256, (KA, FB), dtype=np.uint8)
    B = rng.integers(0, 256, (KB, FB), dtype=np.uint8)
    a_all = rng.integers(0, 18, N_PUSH + 1).astype(np.int32)           # ALE's 18 actions
    r_all = rng.integers(-1, 2, N_PUSH + 1).astype(np.float32)         # clipped Atari rewards
    t_all = (rng.random(N_PUSH + 1) < 1 / 64).astype(np.uint8)         # an episode boundary every ~64 frames
    dA, dB = torch.as_tensor(A).cuda(), torch.as_tensor(B).cuda()
    da, dr, dt = torch.as_tensor(a_all).cuda(), torch.as_tensor(r_all).cuda(), torch.as_tensor(t_all).cuda()
    tr = rlhip.CircularPrioritizedTraces(capacity=CAP, n_env=1, obs_dim=FB, dtype=torch.uint8, default_priority=1.0)
    assert tr.state.numel() == (CAP + 1) * FB and tr.frame_major
    oring = oracle.Ring(CAP, 1, 1)      # the index oracle: "frame" = push number + 1 (exact in Float32 below 2^24)
    ost = oracle.SumTree(CAP)
    s = stream_ptr()
    pA, pB, pa, pr, pt = dA.data_ptr(), dB.data_ptr(), da.data_ptr(), dr.data_ptr(), dt.data_ptr()
    rb, tree = C.byref(tr.rb), ptr(tr.priorities)
    t0 = time.perf_counter()
    _lib.call("rlhip_ring_push_state_maxpool", rb, C.c_void_p(pA), C.c_void_p(pB), s)
    for p in range(1, N_PUSH + 1):
        _lib.call("rlhip_ring_push_transition_maxpool", rb, C.c_void_p(pA + (p % KA) * FB), C.c_void_p(pB + (p % KB) * FB),
                  C.c_void_p(pa + 4 * p), C.c_void_p(pr + 4 * p), C.c_void_p(pt + p), s)
        _lib.call("rlhip_ring_push_priority", rb, tree, C.c_float(1.0), s)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    # the oracle's pushes (its own counter arithmetic; rlo_buffer.c), frame id = push + 1
    ol = oracle.lib()
    ids = np.arange(1, N_PUSH + 2, dtype=np.float32)
    orb = C.byref(oring.rb)
    ol.rlo_ring_push_state(orb, C.c_void_p(ids.ctypes.data))
    for p in range(1, N_PUSH + 1):
        ol.rlo_ring_push_transition(orb, C.c_void_p(ids.ctypes.data + 4 * p), C.c_void_p(a_all.ctypes.data + 4 * p),
                                    C.c_void_p(r_all.ctypes.data + 4 * p), C.c_void_p(t_all.ctypes.data + p))
    ost.fill_range(0, CAP, 1.0)   # every leaf received the default priority (the 777 rewrites write 1.0 over 1.0)
    # the tree after 2^20 + 777 default-priority pushes: bit-exact against the oracle's (checked here, before any test re-prioritises)
    assert np.array_equal(tr.priorities.cpu().numpy(), ost.tree)
    assert float(tr.priorities[1]) == float(CAP)
    note("config5 full ring filled", pushes=N_PUSH, ring_gb=round(tr.state.numel() / 1e9, 2), gpu_push_loop_s=round(t_gpu, 1))

    class Full:
        pass

    f = Full()
    f.rl, f.tr, f.oring, f.ost, f.A, f.B, f.dA, f.dB = rlhip, tr, oring, ost, A, B, dA, dB
    f.a_all, f.r_all, f.t_all, f.ops = a_all, r_all, t_all, ops
    yield f
    del tr, f.tr
    torch.cuda.empty_cache()


def frames_host(A, B, ids):
    """ids: push number + 1, 0 = the all-zero frame StackFrames holds before an episode's first frame"""
    p = np.asarray(ids, np.int64) - 1
    out = np.maximum(A[p % KA], B[p % KB])
    out[np.asarray(ids) == 0] = 0
    return out


def frames_dev(dA, dB, ids):
    ids = torch.as_tensor(np.asarray(ids, np.int64)).cuda()
    p = ids - 1
    out = torch.maximum(dA[p % KA], dB[p % KB])
    out[ids == 0] = 0
    return out


def check_gather(f, idx, got, tag):
    """got = (s, a, r, t, sn) device tensors of rlhip_ring_gather for flat logical indices idx (numpy)"""
    os_, oa, or_, ot, osn = f.oring.gather(idx)      # (1, b) frame ids, a, r, t
    sid, nid = os_[0].astype(np.int64), osn[0].astype(np.int64)
    assert np.array_equal(nid, sid + 1) and sid.min() >= 1
    s, a, r, t, sn = got
    assert np.array_equal(host(a), oa) and np.array_equal(host(r), or_) and np.array_equal(host(t), ot), tag
    assert torch.equal(s, frames_dev(f.dA, f.dB, sid)), f"{tag}: gathered state frames differ"
    assert torch.equal(sn, frames_dev(f.dA, f.dB, nid)), f"{tag}: gathered next-state frames differ"
    k = min(len(idx), 64)                                # and on the host, in numpy, for a slice of the batch
    assert np.array_equal(host(s[:k]), frames_host(f.A, f.B, sid[:k])), tag
    assert np.array_equal(host(sn[-k:]), frames_host(f.A, f.B, nid[-k:])), tag
    assert np.array_equal(oa, f.a_all[sid]) and np.array_equal(ot, f.t_all[sid])  # transition q carries (a, r, t) of push q + 1 = id


def phys_state_slot(f, li):
    return (f.tr.rb.head_sa + li) % (CAP + 1)


def test_counters_after_the_wrap(full):
    f = full
    rb, orb = f.tr.rb, f.oring.rb
    assert len(f.tr) == CAP == len(f.oring)
    for name in ("head_sa", "len_sa", "head_rt", "len_rt"):
        assert getattr(rb, name) == getattr(orb, name), name
    assert rb.head_rt == EXTRA % CAP and rb.head_sa == EXTRA % (CAP + 1) and rb.len_sa == CAP + 1


def test_explicit_indices_around_wrap_storage_end_and_4gib(full):
    f = full
    head = f.tr.rb.head_sa
    four_gib_slot = (1 << 32) // FB          # 152174: this slot straddles byte offset 2^32
    want_phys = [0, 1, CAP - 1, CAP, four_gib_slot - 1, four_gib_slot, four_gib_slot + 1, 2 * four_gib_slot + 1,
                 head - 1, head, head + 1, (1 << 31) // FB, (1 << 31) // FB + 1]
    li = sorted({(p - head) % (CAP + 1) for p in want_phys} | {0, 1, CAP - 2, CAP - 1})
    li = np.array([x for x in li if x < CAP], np.int64)   # a transition needs its next state: li <= CAP - 1
    assert {phys_state_slot(f, int(x)) for x in li} >= {CAP, four_gib_slot, four_gib_slot + 1}
    assert any(phys_state_slot(f, int(x)) == CAP and phys_state_slot(f, int(x) + 1) == 0 for x in li)  # s at the storage's end, s' at its start
    n_bad, _ = f.tr.check_indices(torch.as_tensor(li).cuda())
    assert n_bad == 0
    got = f.tr.gather(torch.as_tensor(li).cuda())
    check_gather(f, li, got, "explicit boundary indices")
    bad, first = f.tr.check_indices(torch.as_tensor(np.array([0, CAP, -1], np.int64)).cuda())
    assert (bad, first) == (2, 1)


@pytest.mark.parametrize("batch", [32, 512, 4096])
def test_uniform_sample_and_gather(full, batch):
    f = full
    four_gib_slot = (1 << 32) // FB
    for ctr in (0, 5):
        oidx = f.oring.sample_indices(batch, 11, ctr)
        idx = f.tr.sample_indices(batch, seed=11, draw_ctr=ctr)
        assert np.array_equal(host(idx), oidx)
        check_gather(f, oidx, f.tr.gather(idx), f"uniform batch {batch} draw {ctr}")
    if batch == 4096:
        phys = (f.tr.rb.head_sa + oidx) % (CAP + 1)
        assert (phys < four_gib_slot).any() and (phys > four_gib_slot).any() and (oidx > CAP - EXTRA - 1).any()
        note("config5 uniform gather 4096", below_4gib=int((phys < four_gib_slot).sum()), wrapped=int((oidx >= CAP - EXTRA).sum()))


def test_prioritized_draw_gather_and_priority_round_trip(full):
    """priorities U(0,1)^0.6 (bench.py: `tr.set_priority_(keys, fill_uniform(cap, 11, 0, 7) ** 0.6)`), then for batch in
    {32, 512, 4096}: the draw (idx, key, priority) against oracle.ring_sample_prioritized, the gathered frames, the fused
    sample+gather launch against the two-launch form; then new priorities for the sampled keys through rlhip_sumtree_update
    (duplicates in the batch: the last one wins on both sides), the WHOLE tree against the oracle's, and a further draw."""
    f = full
    tr = f.tr
    keys = torch.arange(CAP, dtype=torch.int64, device="cuda")
    prio = f.ops.fill_uniform(CAP, 11, 0, 7) ** 0.6
    tr.set_priority_(keys, prio)
    f.ost.update(np.arange(CAP, dtype=np.int64), host(prio))   # same Float32 priorities on both sides (inputs, not results)
    assert np.array_equal(host(tr.priorities), f.ost.tree), "sum-tree after 2^20 priority writes differs from the oracle's"
    four_gib_slot = (1 << 32) // FB
    rng = np.random.default_rng(3)
    ctr = 0
    for batch in (32, 512, 4096):
        for rep in range(2):
            oidx, okey, oprio = oracle.ring_sample_prioritized(f.oring, f.ost, batch, 11, ctr)
            idx, key, pr = tr.sample_prioritized(batch, 11, ctr)
            assert np.array_equal(host(idx), oidx) and np.array_equal(host(key), okey) and np.array_equal(host(pr), oprio)
            check_gather(f, oidx, tr.gather(idx), f"prioritized batch {batch}")
            (idx2, key2, pr2), got2 = tr.sample_gather_prioritized(batch, 11, ctr)
            assert torch.equal(idx2, idx) and torch.equal(key2, key) and torch.equal(pr2, pr)
            check_gather(f, oidx, got2, f"fused prioritized draw + gather, batch {batch}")
            ctr += 1
            # priority write-back of the sampled keys (PrioritizedDQN: p = (|td| + eps)^alpha), then the trees must still agree
            td = rng.standard_normal(batch).astype(np.float32)
            newp = oracle.per_priority(td, 1e-6, 0.6)
            tr.set_priority_(key, torch.as_tensor(newp).cuda())
            f.ost.update(okey, newp)
            assert np.array_equal(host(tr.priorities), f.ost.tree), f"tree differs after the write-back of batch {batch}"
        if batch == 4096:
            phys = (tr.rb.head_sa + oidx) % (CAP + 1)
            assert (phys < four_gib_slot).any() and (phys > four_gib_slot).any()
            assert (oidx >= CAP - EXTRA).any(), "no sample from the wrapped part of the ring"
    # round 6: the write-back of the previous batch inside the next batch's launch (<= 64 keys), on the 2^21-node tree
    for batch in (32, 512):
        uk = rng.integers(0, CAP, 32).astype(np.int64)
        up = oracle.per_priority(rng.standard_normal(32).astype(np.float32), 1e-6, 0.6)
        (idx, key, pr), got = tr.update_sample_gather_prioritized(torch.as_tensor(uk).cuda(), torch.as_tensor(up).cuda(), batch, 11, ctr)
        f.ost.update(uk, up)
        assert np.array_equal(host(tr.priorities), f.ost.tree), "tree differs after the fused write-back"
        oidx, okey, oprio = oracle.ring_sample_prioritized(f.oring, f.ost, batch, 11, ctr)
        assert np.array_equal(host(idx), oidx) and np.array_equal(host(key), okey) and np.array_equal(host(pr), oprio)
        check_gather(f, oidx, got, f"fused write-back + draw + gather, batch {batch}")
        ctr += 1
    note("config5 prioritized", draws=ctr, total_priority=float(tr.priorities[1]))


@pytest.mark.parametrize("batch", [32, 512, 4096])
def test_stacked_gather_n_stack_4_across_episode_boundaries(full, batch):
    f = full
    n_stack = 4
    oidx = f.oring.sample_indices(batch, 13, batch)
    if batch == 512:  # plus the boundary slots of the explicit test and the oldest frames (history cut by the ring's start)
        head, g4 = f.tr.rb.head_sa, (1 << 32) // FB
        special = [(p - head) % (CAP + 1) for p in (0, 1, 2, 3, CAP, g4, g4 + 1, g4 + 3)] + [0, 1, 2, 3, CAP - 1]
        special = np.array([x for x in special if x < CAP], np.int64)
        oidx[:special.size] = special
    os_, oa, or_, ot, osn = oracle.ring_gather_stacked(f.oring, oidx, n_stack)   # (b, n_stack, 1) frame ids, 0 = zero frame
    sid, nid = os_[:, :, 0].astype(np.int64), osn[:, :, 0].astype(np.int64)
    s, a, r, t, sn = f.tr.gather_stacked(torch.as_tensor(oidx).cuda(), n_stack)
    assert np.array_equal(host(a), oa) and np.array_equal(host(r), or_) and np.array_equal(host(t), ot)
    exp_s = frames_dev(f.dA, f.dB, sid.reshape(-1)).reshape(batch, n_stack, FB)
    exp_n = frames_dev(f.dA, f.dB, nid.reshape(-1)).reshape(batch, n_stack, FB)
    assert torch.equal(s, exp_s) and torch.equal(sn, exp_n)
    k = min(batch, 16)
    assert np.array_equal(host(s[:k]), frames_host(f.A, f.B, sid[:k].reshape(-1)).reshape(k, n_stack, FB))
    zero_stacks = int((sid == 0).any(1).sum())
    if batch >= 512:   # with a boundary every ~64 frames ~ 4.6 % of the stacks are cut by one
        assert zero_stacks > 0 and (sid[:, -1] > 0).all()
    note(f"config5 stacked gather batch {batch}", stacks_cut_by_an_episode_boundary=zero_stacks)
