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
