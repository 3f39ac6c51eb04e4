"""Index contracts of the two-layer PPO learner tile (csrc/ppo_grad_tile.h) restated in numpy -- CPU only.

The tile runs layer 1 of both of its phases on v_mfma_f32_32x32x2_f32.  What can go wrong silently is an index map: which
hidden unit a record slot holds, which (sample, unit) pair an accumulator register of a lane is, which lanes of a wave hold
the two halves of a unit's sums.  The maps below are the ones the kernel comments state; the test checks that each of them
covers what it must exactly once (the GPU parity tests check the arithmetic; tools/micro/mfma_f32_l1.hip the operand images).
"""
import numpy as np
import pytest

NW, TILE = 8, 64


def mfma_row(q, kb):  # accumulator register q of a lane in half kb = lane >> 5  (mfma_common.h)
    return (q & 3) + 8 * (q >> 2) + 4 * kb


@pytest.mark.parametrize("h", [8, 64, 72, 128, 200, 256])
def test_record_slots_hold_every_unit_once_and_pad_with_nothing(h):
    """stage_records / record_lds_slot: slot 32 w + m <- unit w hq + m for m < hq = h / 8; slots m >= hq stay zero"""
    hq = h // NW
    assert hq * NW == h and hq <= 32
    slot_of_unit = [(j // hq) * 32 + (j % hq) for j in range(h)]
    assert len(set(slot_of_unit)) == h and max(slot_of_unit) < NW * 32
    unit_of_slot = {}
    for slot in range(NW * 32):
        w, m = slot >> 5, slot & 31
        if m < hq:
            unit_of_slot[slot] = w * hq + m
    assert sorted(unit_of_slot.values()) == list(range(h))
    assert all(unit_of_slot[s] == j for j, s in enumerate(slot_of_unit))


def test_phase_1a_every_sample_unit_pair_is_one_register_of_one_lane():
    """phase 1a: D[unit slot 32 w + m][sample], lane (r, kb), half rt: register q of the lane is slot 32 w + mfma_row(q, kb)
    of sample 32 rt + r; the 16 partial head sums of a sample come from (wave, kb)"""
    seen = np.zeros((NW * 32, TILE), np.int32)
    part_rows = set()
    for w in range(NW):
        for lane in range(64):
            r, kb = lane & 31, lane >> 5
            for rt in range(2):
                for q in range(16):
                    seen[32 * w + mfma_row(q, kb), 32 * rt + r] += 1
                part_rows.add((2 * w + kb, 32 * rt + r))
    assert (seen == 1).all()
    assert len(part_rows) == 2 * NW * TILE  # L.part[(2 w + kb) * TILE + sample]: every cell written by exactly one lane


def test_phase_2_every_sample_unit_pair_is_one_register_of_one_lane_and_the_fold_joins_the_right_lanes():
    """phase 2: D[sample][unit slot 32 w + r]: lane (r, kb), half rt, register q <-> sample 32 rt + mfma_row(q, kb); the two
    lanes r and r + 32 of a wave hold complementary rows of the same unit (joined by v_permlane32_swap in grad_fold)"""
    seen = np.zeros((TILE, NW * 32), np.int32)
    rows_of_lane = {}
    for w in range(NW):
        for lane in range(64):
            r, kb = lane & 31, lane >> 5
            rows = []
            for rt in range(2):
                for q in range(16):
                    s = 32 * rt + mfma_row(q, kb)
                    seen[s, 32 * w + r] += 1
                    rows.append(s)
            rows_of_lane[(w, lane)] = set(rows)
    assert (seen == 1).all()
    for w in range(NW):
        for r in range(32):
            a, b = rows_of_lane[(w, r)], rows_of_lane[(w, r + 32)]
            assert not (a & b) and (a | b) == set(range(TILE))


def test_row_operand_fetch_order_matches_the_register_order():
    """phase 2's pipelined fetch: row P = 16 rt + q reads sample 32 (P >> 4) + (q & 3) + 8 (q >> 2) + 4 kb"""
    for kb in range(2):
        for P in range(32):
            q, rt = P & 15, P >> 4
            assert 32 * (P >> 4) + (q & 3) + 8 * (q >> 2) + 4 * kb == 32 * rt + mfma_row(q, kb)


def test_sample_records_are_the_planes_transposed():
    """the sample records (written by the first gradient launch of an update call, ppo_grad_kernel `samples_out`; rounds 3 - 4:
    pack_update_kernel): record f = t n + i holds {obs[(t ns + k) n + i], k < ns; 0 ...}, {logp[f], adv[f], ret[f], action[f]}:
    gathering record f equals gathering the eight planes at (t, i) -- for every ns the fused learner supports"""
    rng = np.random.default_rng(3)
    for ns in (2, 3, 4):
        n, T = 37, 5
        obs = rng.standard_normal(((T + 1) * ns * n)).astype(np.float32)
        logp, adv, ret = (rng.standard_normal(T * n).astype(np.float32) for _ in range(3))
        act = rng.integers(0, 3, T * n).astype(np.int32)
        rec = np.zeros((T * n, 8), np.float32)
        for f in range(T * n):
            t, i = divmod(f, n)
            for k in range(ns):
                rec[f, k] = obs[(t * ns + k) * n + i]
            rec[f, 4:7] = logp[f], adv[f], ret[f]
            rec[f, 7] = act[f:f + 1].view(np.float32)[0]
        for f in rng.permutation(T * n)[:50]:
            t, i = divmod(int(f), n)
            x = [obs[(t * ns + k) * n + i] if k < ns else 0.0 for k in range(4)]
            assert list(rec[f, :4]) == [np.float32(v) for v in x]
            assert rec[f, 7:8].view(np.int32)[0] == act[f]
            assert (rec[f, 4], rec[f, 5], rec[f, 6]) == (logp[f], adv[f], ret[f])


def test_norm_partial_granules_round_trip_and_never_match_a_stale_epoch():
    """reduce_apply_kernel<APPLY_GRID> / d3_apply_kernel: a workgroup's Float64 sum of squares travels as two 8-byte granules
    {epoch << 32 | high half}, {epoch << 32 | low half}; the reader accepts a pair only when BOTH tags equal this launch's epoch"""
    rng = np.random.default_rng(4)
    vals = np.concatenate([rng.standard_normal(64) ** 2 * 1e3, [0.0, 1e-300, 1e300, np.inf]])
    for epoch in (1, 2, 0x7FFFFFFF, 0xFFFFFFFF):
        for v in vals:
            bits = np.float64(v).view(np.uint64)
            hi = (np.uint64(epoch) << np.uint64(32)) | (bits >> np.uint64(32))
            lo = (np.uint64(epoch) << np.uint64(32)) | (bits & np.uint64(0xFFFFFFFF))
            assert int(hi >> np.uint64(32)) == epoch and int(lo >> np.uint64(32)) == epoch
            back = ((hi & np.uint64(0xFFFFFFFF)) << np.uint64(32)) | (lo & np.uint64(0xFFFFFFFF))
            assert back == bits
            # a torn pair (one granule from the previous launch) is never accepted
            stale = (np.uint64((epoch - 1) & 0xFFFFFFFF) << np.uint64(32)) | (bits >> np.uint64(32))
            assert int(stale >> np.uint64(32)) != epoch
    # a zero-initialised workspace never matches the first launch (epoch = stored count + 1 >= 1)
    assert int(np.uint64(0) >> np.uint64(32)) != 1


# ---------------------------------------------------------------------------------------------------------------------------------
# The two-layer DQN learner tile (csrc/dqn.hip, round 5: 1024 threads around the 16-lane DPP row).  Same idea: the maps the kernel
# comments state, checked for exact coverage.
DQN_THREADS, DQN_TILE = 1024, 64


@pytest.mark.parametrize("h", [4, 8, 64, 100, 128, 200, 252, 256])
def test_dqn_phase_1_every_sample_unit_pair_belongs_to_one_lane_and_a_row_is_one_sample(h):
    """phase 1: wave w, row = lane >> 4, c = lane & 15: sample 4 w + row, hidden units c, c + 16, ... < h; the DPP row sum
    (group_sum_dpp<16>) adds the sixteen lanes of ONE sample"""
    seen = np.zeros((DQN_TILE, h), np.int32)
    for tid in range(DQN_THREADS):
        w, lane = tid >> 6, tid & 63
        row, c = lane >> 4, lane & 15
        smp = 4 * w + row
        assert (lane // 16) * 16 <= lane < (lane // 16) * 16 + 16  # the row is an aligned group of sixteen lanes
        for jj in range(c, h, 16):
            seen[smp, jj] += 1
    assert (seen == 1).all()


@pytest.mark.parametrize("h", [4, 8, 64, 100, 128, 200, 252, 256])
def test_dqn_phase_2_every_sample_unit_pair_belongs_to_one_lane_and_a_row_is_one_unit(h):
    """phase 2: unit j = 4 w + row + 64 p for p < UPL = 2 (h <= 128) or 4, samples c, c + 16, c + 32, c + 48; only lane c == 0
    of a row publishes, every unit exactly once"""
    upl = 2 if h <= 128 else 4
    seen = np.zeros((DQN_TILE, h), np.int32)
    publishers = []
    for tid in range(DQN_THREADS):
        w, lane = tid >> 6, tid & 63
        row, c = lane >> 4, lane & 15
        for p in range(upl):
            j = 4 * w + row + 64 * p
            if 64 * p < h and j < h:
                for i in range(4):
                    seen[c + 16 * i, j] += 1
                if c == 0:
                    publishers.append(j)
    assert (seen == 1).all()
    assert sorted(publishers) == list(range(h))


@pytest.mark.parametrize("nb", [1, 2, 3, 8, 9, 31, 32])
def test_dqn_fused_tail_fold_visits_every_partial_row_once_in_the_reduce_kernels_order(nb):
    """dqn_fused_tail: four groups of per = ceil(nb / 4) rows, chunks of four rows, predicates r0 + r < per and b < nb -- the rows
    of a group in ascending order, every row exactly once (= dqn_reduce_kernel's b0 .. b1 ranges)"""
    per = (nb + 3) // 4
    order = [[] for _ in range(4)]
    for r0 in range(0, per, 4):
        for grp in range(4):
            for r in range(4):
                b = grp * per + r0 + r
                if r0 + r < per and b < nb:
                    order[grp].append(b)
    ref = [list(range(g * per, min(nb, g * per + per))) for g in range(4)]
    assert order == ref and sorted(sum(order, [])) == list(range(nb))
