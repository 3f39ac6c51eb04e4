"""HipVecEnv -- the vectorised classic-control environment behind the reference's AbstractEnv surface.

Mirrors, for N instances at once (the role the historical `MultiThreadEnv` played, blog
docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:347-376):

    CartPoleEnv      src/ReinforcementLearningEnvironments/src/environments/examples/CartPoleEnv.jl
    PendulumEnv      .../PendulumEnv.jl
    MountainCarEnv   .../MountainCarEnv.jl

API names follow RLBase/src/interface.jl with Julia's `!` spelled as a trailing underscore:
`reset_`, `act_`, `state`, `reward`, `is_terminated`, `action_space`, `state_space`, `seed_`, `copy`.
Julia is 1-based: discrete actions here are 1..na exactly like the reference (`act_` takes 1-based
actions, the C ABI below is 0-based; the shift happens in this glue, as the Julia glue would do).

All state lives in HBM as SoA arrays (one wavefront lane per env instance); `state(env)` returns a
device tensor view -- "may be reused and mutated at each step; copy if needed" like the reference
(RLBase/src/interface.jl:515-517).
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import call
from .ops import ptr, stream_ptr

KIND = {"cartpole": 0, "pendulum": 1, "mountaincar": 2, "acrobot": 3}


def _make_cfg(kind, continuous, kw):
    if kind == 0:
        cfg = _lib.CartPoleCfg()
        call("rlhip_cartpole_default", C.byref(cfg))
        cfg.continuous = int(bool(continuous))
    elif kind == 1:
        cfg = _lib.PendulumCfg()
        call("rlhip_pendulum_default", C.byref(cfg))
        cfg.continuous = int(bool(continuous))
    elif kind == 2:
        cfg = _lib.MountainCarCfg()
        call("rlhip_mountaincar_default", C.byref(cfg), int(bool(continuous)))
    else:
        if continuous:
            raise TypeError("AcrobotEnv has a discrete action space (Base.OneTo(3))")
        cfg = _lib.AcrobotCfg()
        call("rlhip_acrobot_default", C.byref(cfg))
        if "book_or_nips" in kw:  # AcrobotEnv.jl:38
            kw = dict(kw)
            kw["nips"] = {"book": 0, "nips": 1}[kw.pop("book_or_nips")]
    rename = {"thetathreshold": "thetathreshold_deg"}
    for k, v in kw.items():
        k = rename.get(k, k)
        if not hasattr(cfg, k):
            raise TypeError(f"unknown keyword argument {k}")  # MethodError in the reference
        setattr(cfg, k, v)
    return cfg


class Space:
    """Minimal descriptor of action_space / state_space: closed interval box or 1..n."""

    def __init__(self, lo=None, hi=None, n=None):
        self.lo, self.hi, self.n = lo, hi, n

    def __contains__(self, x):
        if self.n is not None:
            x = torch.as_tensor(x)
            return bool(((x >= 1) & (x <= self.n)).all())
        x = torch.as_tensor(x, dtype=torch.float64).reshape(len(self.lo), -1)
        lo = torch.tensor(self.lo, dtype=torch.float64).reshape(-1, 1)
        hi = torch.tensor(self.hi, dtype=torch.float64).reshape(-1, 1)
        return bool(((x.cpu() >= lo) & (x.cpu() <= hi)).all())

    def __len__(self):
        return self.n if self.n is not None else len(self.lo)

    def __repr__(self):
        return f"Base.OneTo({self.n})" if self.n is not None else f"Box({self.lo}, {self.hi})"


class HipVecEnv:
    """N independent instances of one classic-control env stepped by HIP kernels.

    auto_reset=True is the MultiThreadEnv protocol: after `act_`, `reward(env)` / `is_terminated(env)`
    describe the step just taken while `state(env)` of a terminated instance is already the first
    state of its next episode.  auto_reset=False gives the scalar-env semantics (caller resets).
    """

    def __init__(self, kind, n_envs=1, T=torch.float32, continuous=None, seed=0, env_id_base=0,
                 auto_reset=True, device="cuda", validate_actions=False, packed_episode=False, **kwargs):
        """packed_episode: keep each instance's reset counter in the spare bits of its step-counter word (the packed mode
        of rlhip_env_state, include/rlhip.h) instead of a separate `episode` array: same trajectories, and an auto-reset
        touches no memory the step kernel does not stream anyway.  For the stand-alone env (`act_` / `reset_` / `state`);
        the fused policy kernels (PPOPolicy.rollout_, the fused DQN step) need the separate array."""
        self.kind = KIND[kind] if isinstance(kind, str) else int(kind)
        self.name = {0: "CartPoleEnv", 1: "PendulumEnv", 2: "MountainCarEnv", 3: "AcrobotRK4Env"}[self.kind]
        if T not in (torch.float32, torch.float64):
            raise TypeError("T must be torch.float32 or torch.float64")
        self.T = T
        self.is_f64 = int(T == torch.float64)
        self.n = int(n_envs)
        if continuous is None:
            continuous = self.kind == 1  # PendulumEnv defaults to continuous = true
        self.continuous = bool(continuous)
        self.cfg = _make_cfg(self.kind, self.continuous, kwargs)
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        self.auto_reset = bool(auto_reset)
        self.validate_actions = validate_actions
        self.packed_episode = bool(packed_episode)
        self.tbits = 1
        while self.tbits < 31 and (1 << self.tbits) <= int(self.cfg.max_steps) + 1:
            self.tbits += 1
        if self.packed_episode and self.tbits > 20:
            raise ValueError("packed_episode needs max_steps < 2^20 - 1")
        # packed mode: an instance resets at most once per reset! / act! call, so this many calls cannot saturate a counter
        self._reset_budget = int(_lib.lib.rlhip_env_packed_episode_capacity(int(self.cfg.max_steps))) if self.packed_episode else None
        self._reset_calls = 0
        self.device = torch.device(device)
        self.sdim = int(_lib.lib.rlhip_env_state_dim(self.kind))
        self.odim = int(_lib.lib.rlhip_env_obs_dim(self.kind))
        self._alloc_state()
        self._bind_state()
        self._obs_valid = False
        self.reset_(is_force=True)  # the constructors call reset! once (CartPoleEnv.jl:77)

    # Streamed arrays of a large vector env are carved from ONE allocation at staggered offsets: with the allocator's
    # default placement every array starts a multiple of 64 MB from the others (2^24 Float32 envs), so that the 13
    # streams of a step launch walk the HBM channels in lock-step; 4352 bytes (17 x 256 B) between consecutive arrays
    # spreads them: 150 -> 144 us per CartPole step at 2^24 envs (tools/envstep_layout.py, profiles/r02_env_step.md).
    STAGGER_BYTES = 4352
    STAGGER_MIN_ENVS = 1 << 20

    def _alloc_state(self):
        n, T, dev = self.n, self.T, self.device
        es = 8 if self.is_f64 else 4
        if n < self.STAGGER_MIN_ENVS:
            self._s = torch.zeros((self.sdim, n), dtype=T, device=dev)
            self._t = torch.zeros(n, dtype=torch.int32, device=dev)
            self._done = torch.zeros(n, dtype=torch.uint8, device=dev)
            self._reward = torch.zeros(n, dtype=T, device=dev)
            self._episode = None if self.packed_episode else torch.zeros(n, dtype=torch.int32, device=dev)
        else:
            def up(b):  # next array start: its bytes + the stagger, rounded to 256 B (16-byte vector accesses stay aligned)
                return (b + self.STAGGER_BYTES + 255) // 256 * 256

            row = up(es * n)  # uniform spacing of the state components -> one strided (sdim, n) view
            o_t = self.sdim * row
            o_r = o_t + up(4 * n)
            o_d = o_r + up(es * n)
            o_e = o_d + up(n)
            total = o_e + (0 if self.packed_episode else up(4 * n))
            self._backing = torch.zeros(total + 256, dtype=torch.uint8, device=dev)
            base = (-self._backing.data_ptr()) % 256

            def view(off, dtype, count):
                nb = count * torch.empty(0, dtype=dtype).element_size()
                return self._backing[base + off: base + off + nb].view(dtype)

            self._s = torch.as_strided(view(0, T, (self.sdim - 1) * (row // es) + n), (self.sdim, n), (row // es, 1))
            self._t = view(o_t, torch.int32, n)
            self._reward = view(o_r, T, n)
            self._done = view(o_d, torch.uint8, n)
            self._episode = None if self.packed_episode else view(o_e, torch.int32, n)
        self._obs = torch.zeros((self.odim, n), dtype=T, device=dev)
        self._last_obs = torch.zeros((self.odim, n), dtype=T, device=dev)

    def _bind_state(self):
        self._st = _lib.EnvState()
        for k in range(self.sdim):
            self._st.s[k] = self._s[k].data_ptr()
        self._st.t = self._t.data_ptr()
        self._st.done = self._done.data_ptr()
        self._st.reward = self._reward.data_ptr()
        self._st.episode = None if self._episode is None else self._episode.data_ptr()

    def step_counter(self):
        """t of every instance (i32[n]); unpacked when the episode counters share the word"""
        return (self._t & ((1 << self.tbits) - 1)) if self.packed_episode else self._t

    def episode_counter(self):
        """number of resets of every instance so far (i32[n])"""
        return ((self._t >> self.tbits) & ((1 << (32 - self.tbits)) - 1)) if self.packed_episode else self._episode

    # ------------------------------------------------------------------ RLBase env API
    def reset_(self, is_force=True):
        """reset!(env).  is_force=False resets only the terminated instances (MultiThreadEnv.reset!)."""
        mask = None if is_force else self._done
        self._count_reset_call()
        call("rlhip_env_reset", self.kind, self.is_f64, C.byref(self.cfg), C.byref(self._st), self.n,
             self.seed, self.env_id_base, ptr(mask), stream_ptr())
        self._obs_valid = False

    def act_(self, actions):
        """act!(env, actions): actions is a length-N device tensor (1-based ints, or T for continuous)."""
        if self.continuous:
            a = actions.to(self.T) if actions.dtype != self.T else actions
            if self.validate_actions:
                lim = self.cfg.max_torque if self.kind == 1 else 1.0
                if not bool(((a >= -lim) & (a <= lim)).all()):
                    raise AssertionError("a in action_space(env)")  # @assert in act!
        else:
            if self.validate_actions and actions not in self.action_space():
                raise AssertionError("a in action_space(env)")
            a = (actions - 1).to(torch.int32)  # 1-based -> 0-based ABI
        self.act0_(a.contiguous())

    def act0_(self, actions0):
        """ABI-level act!: 0-based int32 (discrete) or T (continuous) device tensor, no checks."""
        if self.auto_reset:
            self._count_reset_call()
        call("rlhip_env_step", self.kind, self.is_f64, C.byref(self.cfg), C.byref(self._st), self.n,
             ptr(actions0), int(self.auto_reset), self.seed, self.env_id_base, ptr(self._last_obs),
             ptr(self._obs), stream_ptr())
        self._obs_valid = True

    def _count_reset_call(self):
        """packed episode counters saturate after rlhip_env_packed_episode_capacity resets of one instance (every further
        reset would re-draw the same initial state): refuse the call that could get there instead of going on silently"""
        if self._reset_budget is None:
            return
        self._reset_calls += 1
        if self._reset_calls > self._reset_budget:
            raise _lib.RLHipArgumentError(
                f"packed_episode: {self._reset_calls} reset opportunities exceed the {self._reset_budget} episodes the packed "
                "counter can number (include/rlhip.h, rlhip_env_state): create the env with packed_episode=False")

    def state(self):
        """state(env): (obs_dim, N) device tensor (component-major)."""
        if not self._obs_valid:
            call("rlhip_env_obs", self.kind, self.is_f64, C.byref(self._st), self.n, ptr(self._obs),
                 stream_ptr())
            self._obs_valid = True
        return self._obs

    def last_state(self):
        """Observation produced by the last act! BEFORE any auto-reset (the terminal observation)."""
        return self._last_obs

    def reward(self):
        return self._reward

    def is_terminated(self):
        return self._done.view(torch.bool)

    def action_space(self):
        if self.continuous:
            lim = self.cfg.max_torque if self.kind == 1 else 1.0
            return Space([-lim], [lim])
        return Space(n=self.cfg.n_actions if self.kind == 1 else (2 if self.kind == 0 else 3))

    def state_space(self):
        inf = math.inf
        c = self.cfg
        if self.kind == 0:  # CartPoleEnv.jl:88-93
            th = c.thetathreshold_deg * math.pi / 180
            return Space([-2 * c.xthreshold, -inf, -2 * th, -inf], [2 * c.xthreshold, inf, 2 * th, inf])
        if self.kind == 1:  # PendulumEnv.jl:75-79
            return Space([-1.0, -1.0, -c.max_speed], [1.0, 1.0, c.max_speed])
        if self.kind == 3:  # AcrobotEnv.jl:77-86
            return Space([-1.0, -1.0, -1.0, -1.0, -c.max_vel_a, -c.max_vel_b], [1.0, 1.0, 1.0, 1.0, c.max_vel_a, c.max_vel_b])
        return Space([c.min_pos, -c.max_speed], [c.max_pos, c.max_speed])  # MountainCarEnv.jl:83-86

    def seed_(self, seed):
        """Random.seed!(env, seed): re-keys the Philox streams and restarts the episode counters."""
        self.seed = int(seed)
        if self.packed_episode:
            self._t &= (1 << self.tbits) - 1
            self._reset_calls = 0
        else:
            self._episode.zero_()

    def copy(self):
        """copy(env): deep copy, same seed and counters -> identical future under identical actions."""
        other = object.__new__(HipVecEnv)
        other.__dict__.update(self.__dict__)
        for name in ("_s", "_t", "_done", "_reward", "_episode", "_obs", "_last_obs"):
            v = getattr(self, name)
            setattr(other, name, None if v is None else v.clone())
        other.__dict__.pop("_backing", None)
        kw = {f: getattr(self.cfg, f) for f, _ in self.cfg._fields_}
        other.cfg = type(self.cfg)(**kw)
        other._bind_state()
        return other

    def __len__(self):
        return self.n

    # ------------------------------------------------------------------ low-level access
    def raw_state(self):
        """(state_dim, N) internal state (theta is NOT an observation for Pendulum)."""
        return self._s

    def set_raw_state(self, s, t=None):
        self._s.copy_(torch.as_tensor(s, dtype=self.T, device=self.device))
        if t is not None:
            t = torch.as_tensor(t, dtype=torch.int32, device=self.device)
            if self.packed_episode:
                t = t | (self._t & ~((1 << self.tbits) - 1))
            self._t.copy_(t)
        self._obs_valid = False


def CartPoleEnv(n_envs=1, **kw):
    """CartPoleEnv(; T, continuous, gravity, masscart, masspole, halflength, forcemag, max_steps, dt,
    thetathreshold, xthreshold)  (CartPoleEnv.jl:57-79) x n_envs."""
    return HipVecEnv("cartpole", n_envs, **kw)


def PendulumEnv(n_envs=1, **kw):
    """PendulumEnv(; T, max_speed, max_torque, g, m, l, dt, max_steps, continuous, n_actions)
    (PendulumEnv.jl:24-66) x n_envs."""
    return HipVecEnv("pendulum", n_envs, **kw)


def MountainCarEnv(n_envs=1, **kw):
    """MountainCarEnv(; T, continuous, min_pos, max_pos, max_speed, goal_pos, max_steps, goal_velocity,
    power, gravity)  (MountainCarEnv.jl:51-81) x n_envs."""
    return HipVecEnv("mountaincar", n_envs, **kw)


def AcrobotRK4Env(n_envs=1, **kw):
    """NOT the reference's AcrobotEnv: its `act!` integrates with OrdinaryDiffEq's adaptive `solve(ode, RK4())`
    (AcrobotEnv.jl:128-129, un-vendored step-size control); this env takes ONE classic RK4 step of dt over the same
    dsdt -- velocities differ by up to ~0.03 rad/s from a converged solution (tests/test_oracle_acrobot.py) -- hence the
    name.  AcrobotEnv(; T, link_length_a, ..., max_torque_noise, max_vel_a, max_vel_b, g, dt, max_steps, book_or_nips)
    (3rd_party/AcrobotEnv.jl:22-70) x n_envs.  One classic RK4 step per act! (parity unpinned: include/rlhip.h)."""
    return HipVecEnv("acrobot", n_envs, **kw)


def ContinuousMountainCarEnv(n_envs=1, **kw):
    return HipVecEnv("mountaincar", n_envs, continuous=True, **kw)
