"""GPU parity of the stack-at-sample gather and the max-pool push (ring.hip) against the oracle, byte-exact."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,d", [(torch.uint8, 84 * 84), (torch.uint8, 16), (torch.float32, 8)])
@pytest.mark.parametrize("n_stack", [1, 2, 4, 8])
def test_gather_stacked_vs_oracle(dtype, d, n_stack):
    import rlhip

    rng = np.random.default_rng(d + n_stack)
    cap, steps = 37, 90  # wraps
    tr = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=d, dtype=dtype)
    ring = oracle.Ring(cap, 1, d)

    def frame():
        f = rng.integers(1, 255, d)
        return f.astype(np.float32)

    def dev(f):
        return torch.as_tensor(f.astype(np.uint8) if dtype == torch.uint8 else f, device="cuda").reshape(d, 1)

    f0 = frame()
    tr.push_state_(dev(f0))
    ring.push_state(f0[:, None])
    for t in range(steps):
        f = frame()
        a, r, term = np.array([t % 5], np.int32), np.array([t * 0.5], np.float32), np.array([rng.random() < 0.15], np.uint8)
        tr.push_transition_(dev(f), torch.as_tensor(a, device="cuda"), torch.as_tensor(r, device="cuda"),
                            torch.as_tensor(term, device="cuda"))
        ring.push_transition(f[:, None], a, r, term)
    idx = np.concatenate([np.arange(len(ring)), rng.integers(0, len(ring), 64)])
    s, a, r, t, sn = tr.gather_stacked(torch.as_tensor(idx, device="cuda"), n_stack)
    rs, ra, rr, rt, rsn = oracle.ring_gather_stacked(ring, idx, n_stack)
    assert np.array_equal(s.cpu().numpy().astype(np.float32), rs)
    assert np.array_equal(sn.cpu().numpy().astype(np.float32), rsn)
    assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(r.cpu().numpy(), rr)
    assert np.array_equal(t.cpu().numpy(), rt)
    # zero frames appear exactly where an episode boundary / the ring start cuts the history
    assert (rs.reshape(len(idx), n_stack, d)[:, -1, :] != 0).all()


def test_maxpool_push_identity():
    """states[i] == max.(s1, s2)  (RLEnvs/test/environments/3rd_party/atari.jl:39-59) for the fused push."""
    import rlhip

    d, cap = 84 * 84, 5
    tr = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=d, dtype=torch.uint8)
    g = torch.Generator(device="cpu").manual_seed(0)
    stored = []
    s1 = torch.randint(0, 256, (d,), generator=g, dtype=torch.uint8).cuda()
    s2 = torch.randint(0, 256, (d,), generator=g, dtype=torch.uint8).cuda()
    tr.push_state_maxpool_(s1, s2)
    stored.append(torch.maximum(s1, s2))
    for t in range(8):
        s1 = torch.randint(0, 256, (d,), generator=g, dtype=torch.uint8).cuda()
        s2 = torch.randint(0, 256, (d,), generator=g, dtype=torch.uint8).cuda()
        tr.push_transition_maxpool_(s1, s2, torch.tensor([t], dtype=torch.int32, device="cuda"),
                                    torch.tensor([1.0], device="cuda"), torch.tensor([0], dtype=torch.uint8, device="cuda"))
        stored.append(torch.maximum(s1, s2))
    assert len(tr) == cap
    idx = torch.arange(cap, device="cuda")
    s, a, r, t, sn = tr.gather(idx)  # frame-major (batch, d)
    first = len(stored) - 1 - cap
    for li in range(cap):
        assert torch.equal(s[li], stored[first + li]) and torch.equal(sn[li], stored[first + li + 1])
    assert a.tolist() == list(range(8 - cap, 8))


def test_stacked_gather_argument_validation():
    import rlhip
    from rlhip._lib import RLHipError

    tr = rlhip.CircularArraySARTSTraces(capacity=4, n_env=2, obs_dim=16, dtype=torch.uint8)
    idx = torch.zeros(1, dtype=torch.int64, device="cuda")
    with pytest.raises(RLHipError):
        tr.gather_stacked(idx, 4)  # n_env != 1
    tr = rlhip.CircularArraySARTSTraces(capacity=4, n_env=1, obs_dim=16, dtype=torch.uint8)
    with pytest.raises(RLHipError):
        tr.gather_stacked(idx, 9)  # n_stack > 8
