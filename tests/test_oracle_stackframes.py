"""Stack-at-sample gather of the oracle against a literal restatement of the reference's StackFrames
(RLCore/src/utils/stack_frames.jl:11-44) driven over an episode stream, plus the reference's own known-answer test
(RLCore/test/utils/stack_frames.jl:1-20)."""
import numpy as np

import oracle


class StackFramesSim:
    """StackFrames(T, d...): CircularArrayBuffer of the latest d[end] frames, zero-filled at construction (:22-26)
    and by reset! (:33-36); calling it pushes a frame (:28-31)."""

    def __init__(self, *d):
        self.buf = np.zeros(d, np.float32)  # last axis = time, newest last

    def __call__(self, frame):
        self.buf = np.concatenate([self.buf[..., 1:], np.asarray(frame, np.float32)[..., None]], axis=-1)
        return self

    def reset(self):
        self.buf[...] = 0


def test_reference_known_answers():
    # RLCore/test/utils/stack_frames.jl:3-9
    s = StackFramesSim(2, 3, 2)
    s(np.ones((2, 3), np.float32))
    assert np.array_equal(s.buf[:, :, 0], np.zeros((2, 3))) and np.array_equal(s.buf[:, :, 1], np.ones((2, 3)))
    # :14-20  one dimension lower: three pushes give the columns 1 2 3
    s = StackFramesSim(2, 3)
    for v in (1, 2, 3):
        s(v * np.ones(2))
    assert np.array_equal(s.buf, np.array([[1, 2, 3], [1, 2, 3]], np.float32))


def _episode_stream(rng, d, steps, p_term):
    """frames and flags of a single env with auto-reset: obs[t+1] follows transition t; after a terminal
    transition the next frame is the first frame of a new episode"""
    frames = [rng.integers(1, 255, d).astype(np.float32)]
    term = []
    for _ in range(steps):
        term.append(rng.random() < p_term)
        frames.append(rng.integers(1, 255, d).astype(np.float32))
    return frames, term


def test_stack_at_sample_equals_stackframes_on_the_way_in():
    rng = np.random.default_rng(0)
    d, n_stack, steps, cap = 6, 4, 300, 64
    frames, term = _episode_stream(rng, d, steps, 0.08)
    # the reference pipeline: a StackFrames in front of the agent; the stacked observation is what gets stored
    sf = StackFramesSim(d, n_stack)
    sf(frames[0])
    stacked = [sf.buf.copy()]
    for t in range(steps):
        if term[t]:
            sf.reset()  # reset!(env) -> reset!(StackFrames) at the episode boundary
        sf(frames[t + 1])
        stacked.append(sf.buf.copy())
    # our pipeline: single frames in the ring, stacks rebuilt by the gather
    ring = oracle.Ring(cap, 1, d)
    ring.push_state(frames[0][:, None])
    for t in range(steps):
        ring.push_transition(frames[t + 1][:, None], [t % 3], [float(t)], [term[t]])
    n = len(ring)
    idx = np.arange(n)
    s, a, r, tt, sn = oracle.ring_gather_stacked(ring, idx, n_stack)
    first = steps - n  # global index of logical transition 0
    checked = 0
    for li in range(n):
        g = first + li
        assert r[li] == float(g) and bool(tt[li]) == term[g]
        if li < n_stack:  # history partly overwritten by the wrap: only the frames still stored can match
            continue
        assert np.array_equal(s[li].T, stacked[g]), li        # (n_stack, d) oldest first == buf[:, k]
        assert np.array_equal(sn[li].T, stacked[g + 1]), li
        checked += 1
    assert checked > 50
    # the oldest transitions of a wrapped ring: frames that are gone read as zeros, the rest still matches
    assert np.array_equal(s[0][-1], stacked[first][:, -1]) and not s[0][:-1].any()


def test_n_stack_one_is_the_plain_gather():
    rng = np.random.default_rng(1)
    ring = oracle.Ring(8, 1, 5)
    ring.push_state(rng.standard_normal((5, 1)).astype(np.float32))
    for t in range(11):
        ring.push_transition(rng.standard_normal((5, 1)).astype(np.float32), [t], [1.0], [t % 4 == 3])
    idx = np.arange(len(ring))
    s, a, r, t, sn = oracle.ring_gather_stacked(ring, idx, 1)
    s0, a0, r0, t0, sn0 = ring.gather(idx)
    assert np.array_equal(s[:, 0, :].T, s0) and np.array_equal(sn[:, 0, :].T, sn0)
    assert np.array_equal(a, a0) and np.array_equal(t, t0)
