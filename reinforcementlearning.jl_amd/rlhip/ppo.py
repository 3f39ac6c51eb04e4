"""PPOPolicy on the vectorised env -- host mirror of the (removed) Zoo `PPOPolicy` + `PPOTrajectory`.

Reference pointers: hyper-parameters and network shapes
docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15257-15287; the vector-env run loop
docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:351-374;
`ActorCritic` RLCore/src/utils/networks.jl:15-20; GAE RLCore/src/utils/basic.jl:334-417;
clip_by_global_norm! :19-29; `optimise!(::FluxApproximator, grad)` flux_approximator.jl:46.

Everything numeric is a HIP kernel behind the C ABI (ppo.hip / scans.hip / optim.hip); this class owns
device buffers (torch) and counters, and -- for multi-GPU -- inserts the RCCL all-reduce of the flat
gradient between the gradient kernel and the clip+Adam kernel.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import call
from .ops import ptr, stream_ptr


def make_ppo_cfg(**kw):
    cfg = _lib.PPOCfg()
    call("rlhip_ppo_default", C.byref(cfg))
    rename = {"lambda_": "lam", "λ": "lam", "γ": "gamma", "update_freq": None}
    for k, v in kw.items():
        k = rename.get(k, k)
        if k is None:
            continue
        if not hasattr(cfg, k):
            raise TypeError(f"unknown PPOPolicy keyword {k}")
        setattr(cfg, k, v)
    return cfg


class PPOTrajectory:
    """PPOTrajectory(; capacity = T, state = (ns, N), action = (N,), action_log_prob = (N,), reward = (N,),
    terminal = (N,)) (blog index.html:15280-15286), time-major SoA in HBM, plus value / advantage /
    return traces that the reference recomputes at update time."""

    def __init__(self, ns, n, T, continuous=False, device="cuda"):
        dev = torch.device(device)
        f = torch.float32
        self.ns, self.n, self.T, self.continuous = ns, n, T, continuous
        self.obs = torch.zeros((T + 1, ns, n), dtype=f, device=dev)
        self.logp = torch.zeros((T, n), dtype=f, device=dev)
        self.value = torch.zeros((T + 1, n), dtype=f, device=dev)
        self.reward = torch.zeros((T, n), dtype=f, device=dev)
        self.adv = torch.zeros((T, n), dtype=f, device=dev)
        self.ret = torch.zeros((T, n), dtype=f, device=dev)
        self.action_f = torch.zeros((T, 1, n), dtype=f, device=dev)
        self.action_i = torch.zeros((T, n), dtype=torch.int32, device=dev)
        self.terminal = torch.zeros((T, n), dtype=torch.uint8, device=dev)
        self.c = _lib.PPOTraj()
        for name in ("obs", "logp", "value", "reward", "adv", "ret", "action_f", "action_i", "terminal"):
            setattr(self.c, name, getattr(self, name).data_ptr())

    @property
    def action(self):
        """1-based actions like the reference's trace (discrete), raw actions (continuous)."""
        return self.action_f[:, 0, :] if self.continuous else self.action_i + 1


class PPOPolicy:
    """PPOPolicy(approximator = ActorCritic(actor = ns->hidden->na, critic = ns->hidden->1, Adam(lr)), ...)."""

    def __init__(self, env, update_freq=32, seed=None, params=None, process_group=None, **kw):
        self.env = env
        self.kind = env.kind
        self.cfg = make_ppo_cfg(continuous=int(env.continuous), **kw)
        self.T = int(update_freq)
        self.seed = env.seed if seed is None else int(seed)
        dev = env.device
        self.np = int(_lib.lib.rlhip_ppo_nparams(self.kind, C.byref(self.cfg)))
        if self.np <= 0:
            raise _lib.RLHipArgumentError(_lib.last_error())
        ns = env.odim
        self.na = 1 if env.continuous else len(env.action_space())
        nout_a = 2 * self.na if env.continuous else self.na
        # cfg.layers = 3: actor / critic ns -> h -> h -> nout, h = 128 (ppo3.hip) or 256 (ppo3w.hip), hidden layer on the bf16 MFMA
        self.layers = 3 if self.cfg.layers == 3 else 2
        nparams_fn, init_fn = ("rlhip_mlp2_nparams", "rlhip_mlp2_init_f32") if self.layers == 2 else \
            ("rlhip_mlp3_nparams", "rlhip_mlp3_init_f32")
        self.np_actor = int(getattr(_lib.lib, nparams_fn)(ns, self.cfg.hidden, nout_a))
        if params is None:
            self.params = torch.empty(self.np, dtype=torch.float32, device=dev)
            # glorot_uniform(rng) stand-in: actor net_id 0, critic net_id 1 (Philox INIT stream)
            call(init_fn, ptr(self.params), ns, self.cfg.hidden, nout_a, self.seed, 0, stream_ptr())
            call(init_fn, C.c_void_p(self.params.data_ptr() + 4 * self.np_actor), ns,
                 self.cfg.hidden, 1, self.seed, 1, stream_ptr())
        else:
            self.params = torch.as_tensor(params, dtype=torch.float32, device=dev).clone().contiguous()
            assert self.params.numel() == self.np
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.beta_pow = torch.tensor([self.cfg.beta1, self.cfg.beta2], dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.params)
        self.losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self.gn = torch.zeros(1, dtype=torch.float32, device=dev)
        self.trajectory = PPOTrajectory(ns, env.n, self.T, env.continuous, dev)
        ws = int(_lib.lib.rlhip_ppo_workspace_bytes(self.kind, C.byref(self.cfg), env.n, self.T))
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        # zero-fills it and registers its size: every later call checks its own (n, T) against it (ABI 2)
        call("rlhip_ppo_workspace_init", ptr(self.workspace), ws, stream_ptr())
        self.vec_step = 0      # global vec-step counter (Philox t of the sampling streams)
        self.update_ctr = 0    # number of update_ calls so far
        self.n_pushed = 0      # per-step protocol: vec-steps pushed since the last update
        self.process_group = process_group
        # device-resident mirror of (vec_step, update_ctr) for HIP-graph replay of whole iterations
        self.counters = torch.zeros(2, dtype=torch.int32, device=dev)
        self._graph = None
        self._a_i = torch.zeros(env.n, dtype=torch.int32, device=dev)
        self._a_f = torch.zeros(env.n, dtype=torch.float32, device=dev)
        self._logp = torch.zeros(env.n, dtype=torch.float32, device=dev)
        self._value = torch.zeros(env.n, dtype=torch.float32, device=dev)

    def __del__(self):
        # drop the workspace's entry from the library's size table before the allocator recycles the block
        try:
            _lib.lib.rlhip_ppo_workspace_release(ptr(self.workspace))
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    # ----------------------------------------------------------------- per-step protocol (drop-in)
    def plan_(self, env=None):
        """plan!(policy, env): returns the action tensor (1-based ints for discrete envs)."""
        env = env or self.env
        obs = env.state()
        call("rlhip_ppo_plan_f32", self.kind, C.byref(self.cfg), ptr(self.params), ptr(obs), env.n, self.seed,
             env.env_id_base, self.vec_step, ptr(self._a_i), ptr(self._a_f), ptr(self._logp), ptr(self._value),
             stream_ptr())
        return self._a_f if env.continuous else self._a_i + 1

    def push_preact_(self, env=None):
        """PreActStage push: state, action, action_log_prob (+ value) of the step about to be taken (one launch)."""
        env = env or self.env
        call("rlhip_ppo_push_preact_f32", C.byref(self.trajectory.c), self.n_pushed, env.odim, env.n, ptr(env.state()),
             ptr(self._value), ptr(self._logp), None if env.continuous else ptr(self._a_i),
             ptr(self._a_f) if env.continuous else None, stream_ptr())

    def push_postact_(self, env=None):
        """PostActStage push: reward, terminal (one launch)."""
        env = env or self.env
        call("rlhip_ppo_push_postact_f32", C.byref(self.trajectory.c), self.n_pushed, env.n, ptr(env.reward()),
             ptr(env._done), stream_ptr())
        self.n_pushed += 1
        self.vec_step += 1

    def finish_rollout_(self, env=None):
        """Bootstrap state/value after the last pushed step (the reference pushes state T+1 lazily)."""
        env = env or self.env
        self.plan_(env)  # fills self._value for the current state; draws are not consumed (same step redrawn)
        call("rlhip_ppo_push_preact_f32", C.byref(self.trajectory.c), self.T, env.odim, env.n, ptr(env.state()),
             ptr(self._value), None, None, None, stream_ptr())
        self.n_pushed = 0
        self._adv_ready = False  # per-step protocol: adv / ret are computed by gae_() in update_()

    # ----------------------------------------------------------------- fused protocol
    def rollout_(self, env=None):
        """T vec-steps of plan!/push!/act!/push! in ONE kernel launch."""
        env = env or self.env
        call("rlhip_ppo_rollout_f32", self.kind, C.byref(env.cfg), C.byref(env._st), env.n, self.T,
             C.byref(self.cfg), ptr(self.params), self.seed, env.env_id_base, self.vec_step,
             C.byref(self.trajectory.c), stream_ptr())
        env._obs_valid = False
        self.vec_step += self.T
        self._adv_ready = True  # the rollout kernels finish with the GAE + returns scan of every env (gae_device.h)

    def gae_(self):
        call("rlhip_ppo_gae_f32", C.byref(self.cfg), self.trajectory.n, self.T, C.byref(self.trajectory.c),
             stream_ptr())

    def grad_(self, epoch_ctr, mb, records_fresh=False):
        """records_fresh: the previous optimiser step was apply_() (which refreshes the learner's weight records), so
        the gradient launch can skip its re-pack."""
        call("rlhip_ppo_grad_fresh_f32" if records_fresh else "rlhip_ppo_grad_f32", self.kind, C.byref(self.cfg), self.trajectory.n, self.T,
             C.byref(self.trajectory.c), ptr(self.params), self.seed, epoch_ctr, mb, ptr(self.workspace),
             ptr(self.grad), ptr(self.losses), stream_ptr())

    def apply_(self, grad_scale=1.0):
        """[grad_scale] -> clip_by_global_norm! -> Adam -> weight-record refresh, one launch"""
        call("rlhip_ppo_apply_f32", self.kind, C.byref(self.cfg), self.trajectory.n, self.T, ptr(self.params),
             ptr(self.grad), ptr(self.m), ptr(self.v), ptr(self.beta_pow), grad_scale, ptr(self.workspace),
             ptr(self.gn), stream_ptr())

    def update_(self):
        """optimise!(policy): GAE (unless the fused rollout already wrote adv / ret), then n_epochs x n_microbatches of
        grad -> [all-reduce] -> clip -> Adam."""
        if not getattr(self, "_adv_ready", False):
            self.gae_()
        self._adv_ready = False
        world = 1
        if self.process_group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(self.process_group)
        if world == 1 and not getattr(self, "_force_dist", False):
            call("rlhip_ppo_update_f32", self.kind, C.byref(self.cfg), self.trajectory.n, self.T,
                 C.byref(self.trajectory.c), ptr(self.params), ptr(self.m), ptr(self.v), ptr(self.beta_pow),
                 self.seed, self.update_ctr, ptr(self.workspace), ptr(self.grad), ptr(self.losses), stream_ptr())
        else:
            import torch.distributed as dist

            comm = self._comm(world)
            if comm.ok and os.environ.get("RLHIP_P2P_HOST_LOOP", "0") != "1":
                # the whole sharded update as ONE C call (csrc/ppo_grad.hip rlhip_ppo_update_comm_f32): per optimiser
                # step gradient -> exchange -> clip + Adam; the exchange is the fused peer-to-peer kernel when every
                # rank validated it, ncclAllReduce on this stream otherwise -- no torch.distributed in the data path
                call("rlhip_ppo_update_comm_f32", self.kind, C.byref(self.cfg), self.trajectory.n, self.T,
                     C.byref(self.trajectory.c), ptr(self.params), ptr(self.m), ptr(self.v), ptr(self.beta_pow),
                     self.seed, self.update_ctr, ptr(self.workspace), ptr(self.grad), ptr(self.losses), comm.h,
                     stream_ptr())
                comm.check()  # a peer that never arrived at an earlier exchange: RLHipTimeoutError (host-pinned word, no sync)
                self.update_ctr += 1
                return
            for e in range(self.cfg.n_epochs):
                epoch_ctr = self.update_ctr * self.cfg.n_epochs + e
                for mb in range(self.cfg.n_microbatches):
                    self.grad_(epoch_ctr, mb, records_fresh=(e > 0 or mb > 0))
                    # gradient all-reduce BEFORE the global-norm clip, so the clip sees the global
                    # gradient (mean over shards == single-GPU semantics with a world-times larger batch)
                    if comm.ok:
                        comm.all_reduce_(self.grad)  # rlhip_allreduce_grads on this stream
                    else:  # no transport behind the ABI (e.g. a gloo group whose peer-to-peer validation failed)
                        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.process_group)
                    self.apply_(grad_scale=1.0 / world)
            if comm.ok:  # the torch.distributed fallback never touched the communicator's status word
                comm.check()
        self.update_ctr += 1

    def _comm(self, world):
        """The communicator of this policy's gradient exchange (rlhip.dist.HipComm over csrc/comm.hip), created on
        first use -- a collective: every rank reaches its first update_ together."""
        if not hasattr(self, "_hipcomm"):
            from .dist import HipComm

            self._hipcomm = HipComm.create(self.process_group, self.np, self.params.device)
        return self._hipcomm

    @property
    def _p2p(self):
        """the communicator if its peer-to-peer path is active, else None (round-1 attribute, bench.py / tests)"""
        c = getattr(self, "_hipcomm", None)
        return c if (c is not None and c.p2p_active) else None

    # ----------------------------------------------------------------- HIP-graph protocol
    def sync_counters_(self):
        """Copy the host counters to their device mirror (call before switching to the graph path)."""
        self.counters.copy_(torch.tensor([self.vec_step, self.update_ctr], dtype=torch.int32))

    def _world(self):
        if self.process_group is None:
            return 1
        import torch.distributed as dist

        return dist.get_world_size(self.process_group)

    def iteration_dc_(self, env=None):
        """One whole iteration -- rollout, GAE, n_epochs x n_microbatches updates (with the gradient
        all-reduce when a process group is set), counter advance -- enqueued with every counter read from
        device memory: identical results to rollout_() + update_(), but capturable in a HIP graph."""
        env = env or self.env
        tr = self.trajectory
        call("rlhip_ppo_rollout_dc_f32", self.kind, C.byref(env.cfg), C.byref(env._st), env.n, self.T,
             C.byref(self.cfg), ptr(self.params), self.seed, env.env_id_base, ptr(self.counters), C.byref(tr.c),
             stream_ptr())
        env._obs_valid = False
        self.gae_()
        world = self._world()
        if world == 1 and not getattr(self, "_force_dist", False):
            call("rlhip_ppo_update_dc_f32", self.kind, C.byref(self.cfg), tr.n, self.T, C.byref(tr.c),
                 ptr(self.params), ptr(self.m), ptr(self.v), ptr(self.beta_pow), self.seed, ptr(self.counters),
                 ptr(self.workspace), ptr(self.grad), ptr(self.losses), stream_ptr())
        else:
            import torch.distributed as dist

            for e in range(self.cfg.n_epochs):
                for mb in range(self.cfg.n_microbatches):
                    call("rlhip_ppo_grad_dc_f32", self.kind, C.byref(self.cfg), tr.n, self.T, C.byref(tr.c),
                         ptr(self.params), self.seed, e, ptr(self.counters), mb, ptr(self.workspace),
                         ptr(self.grad), ptr(self.losses), stream_ptr())
                    dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.process_group)
                    self.apply_(grad_scale=1.0 / world)
        call("rlhip_counters_advance", ptr(self.counters), self.T, 1, stream_ptr())
        self.vec_step += self.T
        self.update_ctr += 1

    def capture_graph_(self, env=None, warmup=2):
        """Capture iteration_dc_ into a HIP graph (torch.cuda.CUDAGraph owns the capture stream); `replay_`
        then costs one graph launch per iteration instead of ~35 kernel launches (+ 16 all-reduces)."""
        env = env or self.env
        self.sync_counters_()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # warm-up outside capture (lazy allocations, NCCL channel setup)
                self.iteration_dc_(env)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.iteration_dc_(env)
        self._graph = g
        # the capture itself executed nothing, but iteration_dc_ advanced the host mirrors once: undo
        self.vec_step -= self.T
        self.update_ctr -= 1
        return g

    def replay_(self):
        self._graph.replay()
        self.vec_step += self.T
        self.update_ctr += 1

    def n_updates_per_call(self):
        return self.cfg.n_epochs * self.cfg.n_microbatches
