"""Index contracts between the kernels of csrc/ppo3w.hip, restated in numpy (no GPU): the bf16 hand-over of dZ2 in MFMA
B-fragment order (written by ppo3w_fwd_kernel from the MFMA D layout, read by ppo3w_dw2_kernel as B operand), the W2 fragment
images (ppo3w_pack_kernel / mlp3w_pack_kernel vs the parameter-centric re-pack of ppo3w_adam_pack_kernel) and the XCD pairing of
the dW2 workgroups.  These are the formulas of the source restated once more; the GPU tests pin the kernels against the oracle,
this file pins the formulas against each other and against the MFMA register layout of csrc/mfma_common.h."""
import numpy as np

HW, WV, RW = 256, 8, 64


def mfma_row(q, kb):
    return (q & 3) + 8 * (q >> 2) + 4 * kb


def test_dz_fragment_store_is_the_mfma_b_operand_of_dw2():
    # what ppo3w_fwd_kernel stores: wave w, lane (r, kb), row tile rt, register group gq holds dZ2[sample][col] for
    # sample = 32 rt + mfma_row(4 gq + i, kb), col = 32 w + r  ->  8 bytes at element slot * 8 + 4 kb + i
    tile = 3
    buf = -np.ones(((tile + 1) * (RW // 16) * WV * 64 * 8, 2), dtype=np.int64)  # element -> (sample, col)
    for w in range(WV):
        for lane in range(64):
            r, kb = lane & 31, lane >> 5
            for rt in range(2):
                for gq in range(4):
                    slot = ((tile * (RW // 16) + 2 * rt + (gq >> 1)) * WV + w) * 64 + 32 * (gq & 1) + r
                    for i in range(4):
                        e = slot * 8 + 4 * kb + i
                        assert buf[e, 0] < 0  # every element written once
                        buf[e] = (32 * rt + mfma_row(4 * gq + i, kb), 32 * w + r)
    base = tile * (RW // 16) * WV * 64 * 8
    assert (buf[base:, 0] >= 0).all() and (buf[:base, 0] < 0).all()  # exactly this tile's 64 x 256 elements
    # what ppo3w_dw2_kernel loads (load_dz_frags): k-step ks, column tile = wave w, lane l, element u must be
    # B[k = 16 ks + 8 (l >> 5) + u][col = 32 w + (l & 31)]  (mfma_common.h: B operand of v_mfma_f32_32x32x16_bf16)
    for ks in range(RW // 16):
        for w in range(WV):
            for l in range(64):
                for u in range(8):
                    e = (((tile * (RW // 16) + ks) * WV + w) * 64 + l) * 8 + u
                    assert tuple(buf[e]) == (16 * ks + 8 * (l >> 5) + u, 32 * w + (l & 31))


def _tr16_b64(seg_of_lane):
    """ds_read_b64_tr_b16 as pinned on the GPU by tools/micro/tr16_probe.hip: within a 16-lane group, lane g passes the address of a
    4-element segment and receives out[g][i] = segment of lane 4 i + (g >> 2), element g & 3."""
    out = [None] * 64
    for base in range(0, 64, 16):
        for g in range(16):
            out[base + g] = [seg_of_lane[base + 4 * i + (g >> 2)][g & 3] for i in range(4)]
    return out


def _dzf_copy_slot(c, pad):  # store_dzf_tile: 16-byte slot c of the fragment tile -> slot of the LDS copy
    G, Hh = (72, 36) if pad else (64, 32)
    return G * (c >> 6) + Hh * ((c >> 5) & 1) + (c & 31)


def test_bwd_transposing_reads_of_the_fragment_tile_are_the_mfma_a_operand_of_dh1():
    """round 6: dZ2 leaves the forward kernel ONLY as the fragment image; ppo3w_bwd_kernel copies a tile's 2048 16-byte slots into LDS
    (store_dzf_tile, optionally padded) and builds its A operand -- lane (m = sample row 32 rt + (l & 31), k half l >> 5), eight consecutive
    columns 16 ks + 8 (l >> 5) + 0 .. 7 -- with two transposing reads per fragment (dzf_lane_base / dzf_a_frag).  Restated here with the
    instruction's lane map: every fragment element is the right (sample, column), for both LDS layouts; and the padded layout puts the
    16 segments of a 16-lane group on 16 different 8-byte bank pairs of the 64-bank LDS where the unpadded one puts them on 8."""
    # the fragment tile as the forward kernel stores it (tile-relative): element -> (sample, col)
    img = -np.ones((RW // 16 * WV * 64 * 8, 2), dtype=np.int64)
    for w in range(WV):
        for lane in range(64):
            r, kb = lane & 31, lane >> 5
            for rt in range(2):
                for gq in range(4):
                    slot = ((2 * rt + (gq >> 1)) * WV + w) * 64 + 32 * (gq & 1) + r
                    for i in range(4):
                        img[slot * 8 + 4 * kb + i] = (32 * rt + mfma_row(4 * gq + i, kb), 32 * w + r)
    assert (img[:, 0] >= 0).all()
    for pad in (False, True):
        G, Hh = (72, 36) if pad else (64, 32)
        lds = -np.ones(((RW // 16) * WV * G * 8, 2), dtype=np.int64)
        for c in range(2048):
            lds[8 * _dzf_copy_slot(c, pad):8 * _dzf_copy_slot(c, pad) + 8] = img[8 * c:8 * c + 8]
        worst = 0
        for rt in range(2):
            for ks in range(HW // 16):
                const = (2 * rt * WV * G + (ks >> 1) * G + 16 * (ks & 1)) * 8
                frag = [[None] * 8 for _ in range(64)]
                for h in range(2):  # the two reads of a fragment: + 4 slots
                    addr = []
                    for lane in range(64):
                        Gq, g = lane >> 4, lane & 15
                        i, q = g >> 2, g & 3
                        base = (((Gq & 1) * WV * G) + Hh * (q >> 1) + 8 * (Gq >> 1) + i) * 8 + 4 * (q & 1)
                        addr.append(base + const + 4 * 8 * h)
                    got = _tr16_b64([[tuple(lds[a + e]) for e in range(4)] for a in addr])
                    for lane in range(64):
                        for e in range(4):
                            frag[lane][4 * h + e] = got[lane][e]
                    for b16 in range(0, 64, 16):  # bank pairs (8-byte granules of the 256-byte bank row) hit by one 16-lane group
                        pairs = {(2 * a // 8) % 32 for a in addr[b16:b16 + 16]}
                        worst = max(worst, 16 // len(pairs))
                for lane in range(64):
                    for u in range(8):
                        assert frag[lane][u] == (32 * rt + (lane & 31), 16 * ks + 8 * (lane >> 5) + u), (pad, rt, ks, lane, u)
        assert worst == (1 if pad else 2), (pad, worst)


def test_w2_fragment_images_pack_kernel_equals_parameter_centric_repack():
    # ppo3w_pack_kernel: element q of an image -> which W2[j + HW k] it holds
    q = np.arange(HW * HW)
    u, l, f = q & 7, (q >> 3) & 63, q >> 9
    t, ks = f % WV, f // WV
    col, kk = 32 * t + (l & 31), 16 * ks + 8 * (l >> 5) + u
    jk = col + HW * kk      # "W2jk": B[k = kk][col j]  = W2[j + HW k]
    kj = kk + HW * col      # "W2kj": B[j = kk][col k]  = W2[j + HW k]
    assert np.array_equal(np.sort(jk), q) and np.array_equal(np.sort(kj), q)  # bijections
    # ppo3w_adam_pack_kernel: parameter e = j + HW k -> its slot in each image
    e = np.arange(HW * HW)
    j, k = e & (HW - 1), e // HW
    q1 = ((((k >> 4) * WV + (j >> 5)) * 64) + ((j & 31) + 32 * ((k >> 3) & 1))) * 8 + (k & 7)
    q2 = ((((j >> 4) * WV + (k >> 5)) * 64) + ((k & 31) + 32 * ((j >> 3) & 1))) * 8 + (j & 7)
    assert np.array_equal(jk[q1], e) and np.array_equal(kj[q2], e)
    # and the forward MFMA consumes image "jk" as B[k][col = j]: wave w reads fragment (ks, w), lane l, element u
    for ks, w, l, u in ((0, 0, 0, 0), (5, 3, 37, 6), (15, 7, 63, 7)):
        qq = ((ks * WV + w) * 64 + l) * 8 + u
        assert jk[qq] == (32 * w + (l & 31)) + HW * (16 * ks + 8 * (l >> 5) + u)


def test_dw2_workgroup_mapping_pairs_the_k_halves_on_one_xcd():
    for nsr in (8, 64, 128):
        seen = set()
        for b in range(2 * nsr):
            xcd, jj = b & 7, b >> 3
            kh, sr = jj & 1, (jj >> 1) * 8 + xcd
            assert 0 <= sr < nsr and (sr, kh) not in seen
            seen.add((sr, kh))
        assert len(seen) == 2 * nsr
        # the two halves of a sample range differ by 8 in the workgroup index: same b % 8 = same XCD, adjacent in its order
        where = {}
        for b in range(2 * nsr):
            xcd, jj = b & 7, b >> 3
            where[((jj >> 1) * 8 + xcd, jj & 1)] = b
        for sr in range(nsr):
            assert where[(sr, 1)] - where[(sr, 0)] == 8


def test_reduce_mapping_small_and_w2_ranges_cover_every_parameter_once():
    for ns, nout_a in ((3, 2), (4, 2)):
        per = lambda nout: HW * ns + HW + HW * HW + HW + nout * HW + nout  # noqa: E731
        small = lambda nout: HW * ns + 2 * HW + nout * HW + nout           # noqa: E731
        np_a, np_all = per(nout_a), per(nout_a) + per(1)
        nS_a = small(nout_a)
        hits_s = np.zeros(nS_a + small(1), dtype=int)
        hits_w = np.zeros((2, HW * HW), dtype=int)
        nA = HW * ns + HW
        for p in range(np_all):
            net = 1 if p >= np_a else 0
            q = p - net * np_a
            if q < nA:
                hits_s[(nS_a if net else 0) + q] += 1
            elif q < nA + HW * HW:
                hits_w[net, q - nA] += 1
            else:
                hits_s[(nS_a if net else 0) + q - HW * HW] += 1
        assert (hits_s == 1).all() and (hits_w == 1).all()
