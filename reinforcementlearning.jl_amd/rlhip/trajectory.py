"""Trajectory / CircularArraySARTSTraces / BatchSampler / InsertSampleRatioController -- host mirror of
the un-vendored ReinforcementLearningTrajectories 0.4 surface the reference re-exports
(src/ReinforcementLearningCore/src/ReinforcementLearningCore.jl:10), with the storage in HBM.

Constructor forms follow the in-tree call sites: RLCore/test/policies/q_based_policy.jl:41-47,
docs/src/How_to_implement_a_new_algorithm.md:90-112.  Push protocol: RLCore/src/policies/agent/agent_base.jl:45-59.
The ring arithmetic itself is ring.hip behind the C ABI; head/length counters are host integers.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call
from .ops import ptr, stream_ptr


class CircularArraySARTSTraces:
    """CircularArraySARTSTraces(; capacity, state = T => (obs_dim, n_env), action = Int => (n_env,),
    reward = Float32 => (n_env,), terminal = Bool => (n_env,)).  One frame = one vec-step."""

    def __init__(self, capacity, n_env=1, obs_dim=1, dtype=torch.float32, device="cuda"):
        if dtype not in (torch.float32, torch.uint8):
            raise TypeError("state eltype must be Float32 or UInt8")
        dev = torch.device(device)
        self.capacity, self.n_env, self.obs_dim, self.dtype = capacity, n_env, obs_dim, dtype
        # Float32 observations with <= 4 components live in a RECORD ring (csrc/ring_device.h, include/rlhip.h RLHIP_RING_RECORDS):
        # one 64-byte record {s[4], action, reward, terminal, spare, s_next[4], pad[4]} per (state slot, env) -- the whole
        # transition that LEAVES that state -- so that a sampled transition is ONE cache line.  `records` is the storage (and
        # what a checkpoint holds); `state` / `action` / `reward` / `terminal` / `next_state` are strided VIEWS of it, indexed
        # by the record's slot (capacity + 1 slots; the newest slot holds a state only).  Other rings (UInt8 frames, wider
        # observations) keep one tensor per trace, every frame as pushed.
        self.records_layout = dtype == torch.float32 and obs_dim <= 4
        self.rb = _lib.Ring()
        if self.records_layout:
            self.records = torch.zeros((capacity + 1, n_env, 16), dtype=torch.float32, device=dev)
            assert self.records.numel() * 4 == int(_lib.lib.rlhip_ring_state_bytes(capacity, n_env, obs_dim, 4))
            call("rlhip_ring_init", C.byref(self.rb), capacity, n_env, obs_dim, 4, ptr(self.records), None, None, None)
        else:
            self.state = torch.zeros((capacity + 1, obs_dim, n_env), dtype=dtype, device=dev)
            self.action = torch.zeros((capacity, n_env), dtype=torch.int32, device=dev)
            self.reward = torch.zeros((capacity, n_env), dtype=torch.float32, device=dev)
            self.terminal = torch.zeros((capacity, n_env), dtype=torch.uint8, device=dev)
            call("rlhip_ring_init", C.byref(self.rb), capacity, n_env, obs_dim, 4 if dtype == torch.float32 else 1,
                 ptr(self.state), ptr(self.action), ptr(self.reward), ptr(self.terminal))
        assert int(_lib.lib.rlhip_ring_layout(C.byref(self.rb))) == (2 if self.records_layout else 0)
        self.frame_major = bool(_lib.lib.rlhip_ring_gather_is_frame_major(C.byref(self.rb)))

    def __getattr__(self, name):
        # record rings only (frame rings own real tensors of these names): strided views of `records`
        if name in ("state", "action", "reward", "terminal", "next_state") and self.__dict__.get("records_layout"):
            rec = self.__dict__["records"]
            if name == "state":
                return rec[:, :, :self.obs_dim]          # (capacity + 1, n_env, obs_dim)
            if name == "next_state":
                return rec[:, :, 8:8 + self.obs_dim]     # s' of the transition that leaves the slot's state
            if name == "action":
                return rec.view(torch.int32)[:, :, 4]    # (capacity + 1, n_env), by the slot of the transition's s
            if name == "reward":
                return rec[:, :, 5]
            return rec.view(torch.uint8)[:, :, 24]       # low byte of the terminal word (byte 24 of the 64-byte record)
        raise AttributeError(name)

    def push_state_(self, obs):
        """push!(traces, (state = s,))"""
        call("rlhip_ring_push_state", C.byref(self.rb), ptr(obs), stream_ptr())

    def push_transition_(self, next_obs, action0, reward, terminal):
        """push!(traces, (state = s', action = a, reward = r, terminal = t)); action0 is 0-based int32."""
        if terminal.dtype == torch.bool:
            terminal = terminal.view(torch.uint8)
        call("rlhip_ring_push_transition", C.byref(self.rb), ptr(next_obs), ptr(action0), ptr(reward),
             ptr(terminal), stream_ptr())

    def __len__(self):
        return int(_lib.lib.rlhip_ring_length(C.byref(self.rb)))

    def n_transitions(self):
        return len(self) * self.n_env

    def sample_indices(self, batch, seed, draw_ctr):
        idx = torch.empty(batch, dtype=torch.int64, device=self.state.device)
        call("rlhip_ring_sample_indices", C.byref(self.rb), batch, seed, draw_ctr, ptr(idx), stream_ptr())
        return idx

    def check_indices(self, idx):
        """(number of flat logical indices outside [0, length * n_env), position of the first one or -1): the debugging aid behind the
        bounds-checked build (rlhip_ring_check_indices; one launch + a stream synchronisation)."""
        n_bad, first = C.c_int64(0), C.c_int64(-1)
        call("rlhip_ring_check_indices", C.byref(self.rb), ptr(idx), idx.numel(), C.byref(n_bad), C.byref(first), stream_ptr())
        return int(n_bad.value), int(first.value)

    def gather(self, idx):
        """traces[inds] -> (state, action0, reward, terminal, next_state)."""
        b = idx.numel()
        dev = self.state.device
        shape = (b, self.obs_dim) if self.frame_major else (self.obs_dim, b)
        s = torch.empty(shape, dtype=self.dtype, device=dev)
        sn = torch.empty(shape, dtype=self.dtype, device=dev)
        a = torch.empty(b, dtype=torch.int32, device=dev)
        r = torch.empty(b, dtype=torch.float32, device=dev)
        t = torch.empty(b, dtype=torch.uint8, device=dev)
        call("rlhip_ring_gather", C.byref(self.rb), ptr(idx), b, ptr(s), ptr(a), ptr(r), ptr(t), ptr(sn),
             stream_ptr())
        return s, a, r, t, sn


def _gather_stacked(self, idx, n_stack):
    """traces[inds] with StackFrames applied at sample time (single-env rings of single frames):
    -> (state (b, n_stack, obs_dim), action0, reward, terminal, next_state), stacks oldest-first."""
    b, dev = idx.numel(), self.state.device
    s = torch.empty((b, n_stack, self.obs_dim), dtype=self.dtype, device=dev)
    sn = torch.empty((b, n_stack, self.obs_dim), dtype=self.dtype, device=dev)
    a = torch.empty(b, dtype=torch.int32, device=dev)
    r = torch.empty(b, dtype=torch.float32, device=dev)
    t = torch.empty(b, dtype=torch.uint8, device=dev)
    call("rlhip_ring_gather_stacked", C.byref(self.rb), ptr(idx), b, n_stack, ptr(s), ptr(a), ptr(r), ptr(t), ptr(sn),
         stream_ptr())
    return s, a, r, t, sn


def _push_state_maxpool_(self, screen1, screen2):
    """push!(traces, (state = max.(screen1, screen2),)) -- AtariEnv's 2-frame max-pool fused into the push"""
    call("rlhip_ring_push_state_maxpool", C.byref(self.rb), ptr(screen1), ptr(screen2), stream_ptr())


def _push_transition_maxpool_(self, screen1, screen2, action0, reward, terminal):
    if terminal.dtype == torch.bool:
        terminal = terminal.view(torch.uint8)
    call("rlhip_ring_push_transition_maxpool", C.byref(self.rb), ptr(screen1), ptr(screen2), ptr(action0),
         ptr(reward), ptr(terminal), stream_ptr())


CircularArraySARTSTraces.gather_stacked = _gather_stacked
CircularArraySARTSTraces.push_state_maxpool_ = _push_state_maxpool_
CircularArraySARTSTraces.push_transition_maxpool_ = _push_transition_maxpool_


class CircularPrioritizedTraces(CircularArraySARTSTraces):
    """CircularPrioritizedTraces(CircularArraySARTSTraces(...); default_priority): every pushed transition
    enters the device sum-tree with `default_priority`; `traces.set_priority_(keys, p)` is
    `trajectory[:priority, keys] = p` (un-vendored RLTrajectories 0.4; sumtree.hip)."""

    def __init__(self, capacity, n_env=1, obs_dim=1, dtype=torch.float32, default_priority=100.0, device="cuda"):
        super().__init__(capacity, n_env, obs_dim, dtype, device)
        self.default_priority = float(default_priority)
        if not self.default_priority > 0.0:  # a tree without mass cannot be sampled (the draw would land on an empty slot)
            raise ValueError("default_priority must be > 0")
        self.n_leaves = capacity * n_env
        nodes = int(_lib.lib.rlhip_sumtree_nodes(self.n_leaves))
        self.priorities = torch.zeros(nodes, dtype=torch.float32, device=self.state.device)  # zero-init contract

    def push_transition_(self, next_obs, action0, reward, terminal):
        super().push_transition_(next_obs, action0, reward, terminal)
        call("rlhip_ring_push_priority", C.byref(self.rb), ptr(self.priorities), self.default_priority, stream_ptr())

    def sample_prioritized(self, batch, seed, draw_ctr):
        """-> (logical flat indices for gather, physical keys, priorities)"""
        dev = self.state.device
        idx = torch.empty(batch, dtype=torch.int64, device=dev)
        key = torch.empty(batch, dtype=torch.int64, device=dev)
        prio = torch.empty(batch, dtype=torch.float32, device=dev)
        call("rlhip_ring_sample_prioritized", C.byref(self.rb), ptr(self.priorities), batch, seed, draw_ctr,
             ptr(idx), ptr(key), ptr(prio), stream_ptr())
        return idx, key, prio

    def sample_gather_prioritized(self, batch, seed, draw_ctr):
        """the prioritized draw and the gather of its batch in ONE launch (bit-identical to sample_prioritized + gather)
        -> (idx, key, priority), (state, action0, reward, terminal, next_state)"""
        dev = self.state.device
        idx = torch.empty(batch, dtype=torch.int64, device=dev)
        key = torch.empty(batch, dtype=torch.int64, device=dev)
        prio = torch.empty(batch, dtype=torch.float32, device=dev)
        shape = (batch, self.obs_dim) if self.frame_major else (self.obs_dim, batch)
        s = torch.empty(shape, dtype=self.dtype, device=dev)
        sn = torch.empty(shape, dtype=self.dtype, device=dev)
        a = torch.empty(batch, dtype=torch.int32, device=dev)
        r = torch.empty(batch, dtype=torch.float32, device=dev)
        t = torch.empty(batch, dtype=torch.uint8, device=dev)
        call("rlhip_ring_sample_gather_prioritized", C.byref(self.rb), ptr(self.priorities), batch, seed, draw_ctr, ptr(idx),
             ptr(key), ptr(prio), ptr(s), ptr(a), ptr(r), ptr(t), ptr(sn), stream_ptr())
        return (idx, key, prio), (s, a, r, t, sn)

    def update_sample_gather_prioritized(self, upd_keys, upd_prio, batch, seed, draw_ctr):
        """`trajectory[:priority, upd_keys] = upd_prio`, then the prioritized draw and the gather of its batch -- ONE launch for
        <= 64 keys on a frame-major ring (bit-identical to set_priority_ + sample_gather_prioritized, which is what runs otherwise)
        -> (idx, key, priority), (state, action0, reward, terminal, next_state)"""
        dev = self.state.device
        if not hasattr(self, "_sync"):
            self._sync = torch.zeros(2, dtype=torch.int32, device=dev)  # zero before the first call; every call re-arms it
        n_upd = 0 if upd_keys is None else upd_keys.numel()
        if n_upd and (upd_keys.dtype != torch.int64 or upd_prio.dtype != torch.float32 or upd_prio.numel() != n_upd):
            raise TypeError("keys must be int64 and priorities float32 of the same length")
        idx = torch.empty(batch, dtype=torch.int64, device=dev)
        key = torch.empty(batch, dtype=torch.int64, device=dev)
        prio = torch.empty(batch, dtype=torch.float32, device=dev)
        shape = (batch, self.obs_dim) if self.frame_major else (self.obs_dim, batch)
        s = torch.empty(shape, dtype=self.dtype, device=dev)
        sn = torch.empty(shape, dtype=self.dtype, device=dev)
        a = torch.empty(batch, dtype=torch.int32, device=dev)
        r = torch.empty(batch, dtype=torch.float32, device=dev)
        t = torch.empty(batch, dtype=torch.uint8, device=dev)
        call("rlhip_ring_update_sample_gather_prioritized", C.byref(self.rb), ptr(self.priorities), ptr(upd_keys) if n_upd else None,
             ptr(upd_prio) if n_upd else None, n_upd, batch, seed, draw_ctr, ptr(idx), ptr(key), ptr(prio), ptr(s), ptr(a), ptr(r), ptr(t),
             ptr(sn), ptr(self._sync), stream_ptr())
        return (idx, key, prio), (s, a, r, t, sn)

    def set_priority_(self, keys, prio):
        """trajectory[:priority, keys] = prio  (sequential semantics: the last duplicate key wins)"""
        if keys.dtype != torch.int64 or prio.dtype != torch.float32:
            raise TypeError("keys must be int64 and priorities float32")
        if keys.numel() != prio.numel():
            raise ValueError("keys and priorities differ in length")
        call("rlhip_sumtree_update", ptr(self.priorities), self.n_leaves, ptr(keys), ptr(prio), keys.numel(),
             stream_ptr())

    def total_priority(self):
        return float(self.priorities[1])


class BatchSampler:
    """BatchSampler(batchsize; rng): uniform indices with replacement (Philox SAMPLER stream); over
    CircularPrioritizedTraces: `inds, priorities = rand(rng, sumtree, batchsize)` and the batch also carries
    `key` (for the priority write-back) and `priority`."""

    def __init__(self, batchsize, seed=0):
        self.batchsize, self.seed, self.draw_ctr = batchsize, seed, 0

    def sample(self, traces):
        extra = {}
        if isinstance(traces, CircularPrioritizedTraces):
            (idx, key, prio), (s, a, r, t, sn) = traces.sample_gather_prioritized(self.batchsize, self.seed, self.draw_ctr)
            self.draw_ctr += 1
            return dict(state=s, action=a + 1, reward=r, terminal=t.view(torch.bool), next_state=sn, key=key, priority=prio)
        else:
            idx = traces.sample_indices(self.batchsize, self.seed, self.draw_ctr)
            extra = dict(key=idx)
        self.draw_ctr += 1
        s, a, r, t, sn = traces.gather(idx)
        return dict(state=s, action=a + 1, reward=r, terminal=t.view(torch.bool), next_state=sn, **extra)


class NStepBatchSampler:
    """NStepBatchSampler(n, gamma, batchsize; rng) of RLTrajectories 0.4 (un-vendored; oracle/rlo_buffer.c restates it): start
    indices with n transitions ahead of them, each window folded on the device into ONE transition
    (s_i, a_i, R = discount_rewards_reduced(r_i .. r_{i+ns-1}, gamma), any(terminal), s_{i+ns}) -- ns = n unless a terminal flag
    ends the window earlier.  `sample` returns the batch dictionary of BatchSampler; `fold` returns the folded record ring +
    indices the DQN gradient entry points take unchanged (with gamma^n = `gamma_n` as their discount).  Record rings only."""

    def __init__(self, n, gamma, batchsize, seed=0):
        if not 1 <= int(n) <= 32:
            raise ValueError("n must be in 1..32")
        self.n, self.gamma, self.batchsize, self.seed, self.draw_ctr = int(n), float(gamma), int(batchsize), seed, 0
        self.gamma_n = float(_lib.lib.rlhip_gamma_pow(self.gamma, self.n))
        self._folded = None

    def sample_indices(self, traces, draw_ctr=None):
        ctr = self.draw_ctr if draw_ctr is None else draw_ctr
        idx = torch.empty(self.batchsize, dtype=torch.int64, device=traces.state.device)
        call("rlhip_ring_sample_indices_nstep", C.byref(traces.rb), self.batchsize, self.n, self.seed, ctr, ptr(idx), stream_ptr())
        return idx

    def fold(self, traces, idx=None):
        """-> (folded traces: a CircularArraySARTSTraces of capacity 1 x batchsize envs holding the n-step transitions, iota)"""
        if idx is None:
            idx = self.sample_indices(traces)
            self.draw_ctr += 1
        b = idx.numel()
        if self._folded is None or self._folded.n_env != b or self._folded.obs_dim != traces.obs_dim:
            self._folded = CircularArraySARTSTraces(capacity=1, n_env=b, obs_dim=traces.obs_dim, device=traces.state.device)
            self._iota = torch.empty(b, dtype=torch.int64, device=traces.state.device)
        call("rlhip_ring_fold_nstep", C.byref(traces.rb), ptr(idx), b, self.n, self.gamma, C.byref(self._folded.rb), ptr(self._iota),
             stream_ptr())
        return self._folded, self._iota

    def sample(self, traces):
        folded, iota = self.fold(traces)
        s, a, r, t, sn = folded.gather(iota)
        return dict(state=s, action=a + 1, reward=r, terminal=t.view(torch.bool), next_state=sn)


class InsertSampleRatioController:
    """InsertSampleRatioController(ratio, threshold; n_inserted = 0, n_sampled = 0): sampling is allowed
    once n_inserted >= threshold and while n_sampled <= (n_inserted - threshold) * ratio
    (docs/src/How_to_implement_a_new_algorithm.md:108)."""

    def __init__(self, ratio=1.0, threshold=1, n_inserted=0, n_sampled=0):
        self.ratio, self.threshold, self.n_inserted, self.n_sampled = ratio, threshold, n_inserted, n_sampled

    def on_insert_(self, n=1):
        self.n_inserted += n

    def on_sample_(self):
        if self.n_inserted >= self.threshold and self.n_sampled <= (self.n_inserted - self.threshold) * self.ratio:
            self.n_sampled += 1
            return True
        return False


class Trajectory:
    """Trajectory(container, sampler, controller): `push_` inserts, iteration yields sampled batches while
    the controller allows (the `for batch in trajectory` protocol, RLCore/src/policies/learners/td_learner.jl:85-92)."""

    def __init__(self, container, sampler=None, controller=None):
        self.container = container
        self.sampler = sampler or BatchSampler(32)
        self.controller = controller or InsertSampleRatioController()

    def push_state_(self, obs):
        self.container.push_state_(obs)

    def push_transition_(self, next_obs, action0, reward, terminal):
        self.container.push_transition_(next_obs, action0, reward, terminal)
        self.controller.on_insert_(1)

    def __len__(self):
        return len(self.container)

    def __iter__(self):
        while len(self.container) > 0 and self.controller.on_sample_():
            yield self.sampler.sample(self.container)
