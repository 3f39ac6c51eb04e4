"""QBasedPolicy / DQN learner / explorers / approximators on the vectorised env (host mirror).

Reference surface mirrored here:
    QBasedPolicy            src/ReinforcementLearningCore/src/policies/q_based_policy.jl:13-49
    EpsilonGreedyExplorer   .../policies/explorers/epsilon_greedy_explorer.jl:38-131 (GreedyExplorer :200-214)
    FluxApproximator        .../policies/learners/flux_approximator.jl:11-46
    TargetNetwork           .../policies/learners/target_network.jl:27-88
    BasicDQN/DQN learner    removed Zoo; spec docs/src/rlcore.md:28 and the blog config
                            docs/homepage/blog/a_practical_introduction_to_RL.jl/index.html:15121-15147
All numerics are HIP kernels behind the C ABI (dqn.hip, select.hip, optim.hip).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import call
from .ops import ptr, stream_ptr


# ----------------------------------------------------------------------------- functional layer
def dqn_workspace(ns, h, na, batch, device="cuda"):
    nbytes = int(_lib.lib.rlhip_dqn_workspace_bytes(ns, h, na, batch))
    return torch.zeros(nbytes, dtype=torch.uint8, device=device)  # zeroed: the tail holds rlhip_dqn_update_f32's counters


def dqn_update(traces, h, na, act, params, target_params, batch, gamma, delta, seed, draw_ctr, workspace, grad, loss, m, v,
               beta_pow, grad_scale, max_grad_norm, lr, beta1, beta2, eps, gn=None):
    """optimise!(learner, batch) in place: ONE launch up to 2048 samples (gradient; the workgroup that departs last reduces, clips, steps), two beyond."""
    call("rlhip_dqn_update_f32", C.byref(traces.rb), h, na, act, ptr(params), ptr(target_params), batch, gamma, delta,
         seed, draw_ctr, ptr(workspace), ptr(grad), ptr(loss), ptr(m), ptr(v), ptr(beta_pow), grad_scale, max_grad_norm,
         lr, beta1, beta2, eps, ptr(gn) if gn is not None else None, stream_ptr())
    return grad, loss


def dqn_grad(traces, h, na, act, params, target_params, batch, gamma, delta, seed, draw_ctr, workspace=None,
             grad=None, loss=None):
    """One DQN learner step up to the gradient: sample `batch` transitions, TD target, Huber, backward."""
    dev = params.device
    workspace = workspace if workspace is not None else dqn_workspace(traces.obs_dim, h, na, batch, dev)
    grad = grad if grad is not None else torch.empty_like(params)
    loss = loss if loss is not None else torch.empty(1, dtype=torch.float32, device=dev)
    call("rlhip_dqn_grad_f32", C.byref(traces.rb), h, na, act, ptr(params), ptr(target_params), batch, gamma,
         delta, seed, draw_ctr, ptr(workspace), ptr(grad), ptr(loss), stream_ptr())
    return grad, loss


def dqn_plan(params, ns, h, na, act, obs, eps, seed, env_id_base, step, actions=None, q_out=None):
    """plan!(QBasedPolicy, env): Q forward + eps-greedy for n envs; returns (0-based actions, q (na, n))."""
    n = obs.shape[1]
    actions = actions if actions is not None else torch.empty(n, dtype=torch.int32, device=obs.device)
    q_out = q_out if q_out is not None else torch.empty((na, n), dtype=torch.float32, device=obs.device)
    call("rlhip_dqn_plan_f32", ptr(params), ns, h, na, act, ptr(obs), n, float(eps), seed, env_id_base, step,
         ptr(actions), ptr(q_out), stream_ptr())
    return actions, q_out


def mlp3_init(ns, h, na, seed, net_id=0, device="cuda"):
    p = torch.empty(int(_lib.lib.rlhip_mlp3_nparams(ns, h, na)), dtype=torch.float32, device=device)
    call("rlhip_mlp3_init_f32", ptr(p), ns, h, na, seed, net_id, stream_ptr())
    return p


def mlp3_pack(params, ns, h, na, packed=None):
    """bf16 copies of W2 in both MFMA operand orders (refresh after every parameter update)."""
    if packed is None:
        packed = torch.empty(int(_lib.lib.rlhip_mlp3_packed_elems(h)), dtype=torch.int16, device=params.device)
    call("rlhip_mlp3_pack_bf16", ptr(params), ns, h, na, ptr(packed), stream_ptr())
    return packed


def dqn3_plan(params, packed, ns, h, na, act, obs, eps=0.0, seed=0, env_id_base=0, step=0, actions=None, q_out=None,
              want_actions=True):
    n = obs.shape[1]
    if want_actions and actions is None:
        actions = torch.empty(n, dtype=torch.int32, device=obs.device)
    q_out = q_out if q_out is not None else torch.empty((na, n), dtype=torch.float32, device=obs.device)
    call("rlhip_dqn3_plan_f32", ptr(params), ptr(packed), ns, h, na, act, ptr(obs), n, float(eps), seed, env_id_base,
         step, ptr(actions) if want_actions else None, ptr(q_out), stream_ptr())
    return actions, q_out


def dqn3_workspace(ns, h, na, batch, device="cuda"):
    # zeroed: the tail holds rlhip_dqn3_update_f32's counters
    return torch.zeros(int(_lib.lib.rlhip_dqn3_workspace_bytes(ns, h, na, batch)), dtype=torch.uint8, device=device)


def dqn3_update(traces, h, na, act, params, packed, target_params, target_packed, batch, gamma, delta, seed, draw_ctr,
                workspace, grad, loss, m, v, beta_pow, grad_scale, max_grad_norm, lr, beta1, beta2, eps, gn=None):
    """optimise!(learner, batch) of the 3-layer learner in two launches (params, moments, packed updated in place)."""
    call("rlhip_dqn3_update_f32", C.byref(traces.rb), h, na, act, ptr(params), ptr(packed), ptr(target_params),
         ptr(target_packed), batch, gamma, delta, seed, draw_ctr, ptr(workspace), ptr(grad), ptr(loss), ptr(m), ptr(v),
         ptr(beta_pow), grad_scale, max_grad_norm, lr, beta1, beta2, eps, ptr(gn) if gn is not None else None,
         stream_ptr())
    return grad, loss


def dqn3_grad(traces, h, na, act, params, packed, target_params, target_packed, batch, gamma, delta, seed, draw_ctr,
              idx=None, workspace=None, grad=None, loss=None, td=None):
    dev = params.device
    workspace = workspace if workspace is not None else dqn3_workspace(traces.obs_dim, h, na, batch, dev)
    grad = grad if grad is not None else torch.empty_like(params)
    loss = loss if loss is not None else torch.empty(1, dtype=torch.float32, device=dev)
    call("rlhip_dqn3_grad_f32", C.byref(traces.rb), h, na, act, ptr(params), ptr(packed), ptr(target_params),
         ptr(target_packed), batch, None if idx is None else ptr(idx), gamma, delta, seed, draw_ctr, ptr(workspace),
         ptr(grad), ptr(loss), None if td is None else ptr(td), stream_ptr())
    return grad, loss


# ----------------------------------------------------------------------------------- explorers
class EpsilonGreedyExplorer:
    """EpsilonGreedyExplorer(; ϵ_stable, kind = :linear, ϵ_init = 1.0, warmup_steps = 0, decay_steps = 0,
    step = 1, is_break_tie = false, rng) -- epsilon_greedy_explorer.jl:38-67.  One `step` per call, shared
    by all N lanes of a vector env (the historical `policy(env)` was a single call per vec-step)."""

    def __init__(self, eps_stable, kind="linear", eps_init=1.0, warmup_steps=0, decay_steps=0, step=1,
                 is_break_tie=False, seed=0):
        if kind not in ("linear", "exp"):
            raise ValueError("kind must be 'linear' or 'exp'")
        self.eps_stable, self.kind, self.eps_init = float(eps_stable), kind, float(eps_init)
        self.warmup_steps, self.decay_steps, self.step = int(warmup_steps), int(decay_steps), int(step)
        self.is_break_tie, self.seed = bool(is_break_tie), int(seed)

    def get_eps(self, step=None):
        """get_ϵ(s[, step])  :69-90"""
        step = self.step if step is None else step
        return _lib.lib.rlhip_get_eps(0 if self.kind == "linear" else 1, self.eps_stable, self.eps_init,
                                      self.warmup_steps, self.decay_steps, step)

    def plan_(self, values, mask=None, env_id_base=0):
        """plan!(s, values[, mask]) for a (na, N) device tensor of values -> 1-based actions."""
        from .ops import eps_greedy_select

        eps = self.get_eps()
        step = self.step
        self.step += 1  # :104,:110 -- incremented on every call, before the draw
        a0 = eps_greedy_select(values, eps, self.seed, step, env_id_base, mask, self.is_break_tie)
        return a0 + 1


    def prob(self, values, mask=None):
        """prob(s, values[, mask])  :141-194 -- the Float64 probability of every action of every env, (na, N) like `values`
        (the reference returns `Categorical(probs)` per env; the step counter does not advance)."""
        from .ops import eps_greedy_prob

        return eps_greedy_prob(values, self.get_eps(), mask, self.is_break_tie)


class GreedyExplorer(EpsilonGreedyExplorer):
    """GreedyExplorer()  :200-214 -- findmax first-index rule, no randomness."""

    def __init__(self):
        super().__init__(0.0)

    def get_eps(self, step=None):
        return 0.0


# -------------------------------------------------------------------------------- approximators
class HipApproximator:
    """FluxApproximator(model = Chain(Dense(ns, h, act), Dense(h, n_out)), optimiser = Adam(lr)).
    Flat parameters in Flux.destructure order + Adam state, all in HBM."""

    def __init__(self, n_in, hidden, n_out, act="relu", lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, seed=0,
                 net_id=0, device="cuda", params=None, layers=2):
        """layers = 3: Chain(Dense(n_in, h, act), Dense(h, h, act), Dense(h, n_out)) -- the blog's DQN model; the
        hidden x hidden layer runs in bf16 on the MFMA (dqn3.hip at hidden = 128, the streaming kernels of ppo3w.hip at 256)."""
        if layers not in (2, 3):
            raise ValueError("layers must be 2 or 3")
        self.n_in, self.hidden, self.n_out, self.layers = n_in, hidden, n_out, layers
        self.act = {"relu": 0, "tanh": 1}[act] if isinstance(act, str) else int(act)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        from .ops import mlp2_init

        init = mlp2_init if layers == 2 else mlp3_init
        self.params = init(n_in, hidden, n_out, seed, net_id, device) if params is None else \
            torch.as_tensor(params, dtype=torch.float32, device=device).clone()
        self.packed = mlp3_pack(self.params, n_in, hidden, n_out) if layers == 3 else None
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.beta_pow = torch.tensor([beta1, beta2], dtype=torch.float32, device=device)
        self.gn = torch.zeros(1, dtype=torch.float32, device=device)

    def forward(self, x):
        """forward(A, x) = A.model(x)  (flux_approximator.jl:43): x (n_in, batch) -> (n_out, batch)."""
        from .ops import mlp2_forward

        if self.layers == 3:
            return dqn3_plan(self.params, self.packed, self.n_in, self.hidden, self.n_out, self.act, x,
                             want_actions=False)[1]
        return mlp2_forward(self.params, self.n_in, self.hidden, self.n_out, self.act, x)

    def optimise_(self, grad, clip_norm=0.0, grad_scale=1.0):
        """optimise!(A, grad) = Flux.Optimise.update!(A.optimiser_state, A.model, grad)  (:46)."""
        from .ops import clip_adam_

        clip_adam_(self.params, grad, self.m, self.v, self.beta_pow, grad_scale, clip_norm, self.lr, self.beta1,
                   self.beta2, self.eps, self.gn)
        if self.layers == 3:
            mlp3_pack(self.params, self.n_in, self.hidden, self.n_out, self.packed)


class TargetNetwork:
    """TargetNetwork(network; sync_freq = 1, ρ = 0f0)  target_network.jl:27-60."""

    def __init__(self, network, sync_freq=1, rho=0.0):
        if not 0 <= rho <= 1:
            raise AssertionError("ρ must in [0,1]")  # :50
        self.network, self.sync_freq, self.rho, self.n_optimise = network, int(sync_freq), float(rho), 0
        self.target = network.params.clone()
        self.target_packed = network.packed.clone() if getattr(network, "layers", 2) == 3 else None

    def forward(self, x):
        return self.network.forward(x)

    def optimise_(self, grad, **kw):
        """optimise!(tn, grad)  :70-88: update the network, then every sync_freq calls
        dest = ρ·dest + (1-ρ)·src and reset the counter."""
        from .ops import polyak_

        self.network.optimise_(grad, **kw)
        self.n_optimise += 1
        if self.n_optimise % self.sync_freq == 0:
            polyak_(self.target, self.network.params, self.rho)
            self.n_optimise = 0
            net = self.network
            if getattr(net, "layers", 2) == 3:
                mlp3_pack(self.target, net.n_in, net.hidden, net.n_out, self.target_packed)


# -------------------------------------------------------------------------------------- learner
class DQNLearner:
    """BasicDQN / DQN learner on a device trajectory: every `update_freq` vec-steps (once
    `min_replay_history` transitions are stored, and while the trajectory's InsertSampleRatioController
    allows another batch) sample a batch, TD target with the target network, Huber loss, gradient,
    [all-reduce], Adam, target sync.  At most ONE batch per optimise! call (the reference's
    `for batch in trajectory` may yield several to catch up after a warm-up; a vec-step already inserts
    n_env transitions at once).

    Prioritized replay (CircularPrioritizedTraces): batches are drawn in proportion to the stored priorities,
    (|td| + per_eps)^per_alpha is written back, and -- round 4 -- the loss carries the importance-sampling weights
    of PrioritizedDQN when `per_beta > 0`: w = 1 ./ ((priority .+ 1f-10) .^ beta), w ./= maximum(w),
    loss = mean(w .* huber(td)) (`per_beta` may be a callable n_updates -> beta for the usual annealing towards 1;
    per_beta = 0, the default, is the proportional-sampling variant without weights)."""

    def __init__(self, approximator, batchsize=32, gamma=0.99, huber_delta=1.0, min_replay_history=100,
                 update_freq=1, max_grad_norm=0.0, seed=0, process_group=None, per_eps=1e-6, per_alpha=0.6, per_beta=0.0, n_step=1):
        """n_step > 1: batches come from NStepBatchSampler(n_step, gamma, batchsize) -- the window folded on the device into one
        transition -- and the TD target is R + gamma^n (1 - t) max Qt(s_{i+n}); uniform replay, per-stage loop."""
        self.n_step = int(n_step)
        self._nstep = None
        if self.n_step > 1:
            from .trajectory import NStepBatchSampler

            self._nstep = NStepBatchSampler(self.n_step, gamma, batchsize, seed=seed)
        self.per_eps, self.per_alpha, self.per_beta = float(per_eps), float(per_alpha), per_beta
        self.approximator = approximator  # a TargetNetwork
        net = approximator.network
        self.batchsize, self.gamma, self.delta = batchsize, gamma, huber_delta
        self.min_replay_history, self.update_freq, self.max_grad_norm = min_replay_history, update_freq, max_grad_norm
        if int(update_freq) < 1:
            raise ValueError("update_freq must be >= 1")
        self.seed, self.draw_ctr, self.n_updates = seed, 0, 0
        self.vec_steps = 0  # optimise! calls so far (one per vec-step): the update_freq gate
        self.process_group = process_group
        self.grad = torch.zeros_like(net.params)
        self.loss = torch.zeros(1, dtype=torch.float32, device=net.params.device)
        ws = dqn_workspace if net.layers == 2 else dqn3_workspace
        self.workspace = ws(net.n_in, net.hidden, net.n_out, batchsize, net.params.device)
        dev = net.params.device
        self.td = torch.zeros(batchsize, dtype=torch.float32, device=dev)
        self.is_weights = torch.ones(batchsize, dtype=torch.float32, device=dev)
        self._idx = self._key = self._prio = None

    def forward(self, x):
        return self.approximator.forward(x)

    def should_update_(self, trajectory, n_transitions=None):
        """the gate of optimise!: warm-up, every `update_freq`-th vec-step, the trajectory's sample / insert controller.
        Advances the vec-step counter (call once per vec-step; run_fused_dqn uses it for its `do_update` flag)."""
        self.vec_steps += 1
        n = trajectory.container.n_transitions() if n_transitions is None else n_transitions
        if n < self.min_replay_history or self.vec_steps % self.update_freq != 0:
            return False
        return bool(trajectory.controller.on_sample_())

    def optimise_(self, trajectory):
        traces = trajectory.container
        if not self.should_update_(trajectory):
            return False
        net = self.approximator.network
        prioritized = hasattr(traces, "sample_prioritized")
        if self._nstep is not None:
            if prioritized:
                raise NotImplementedError("n-step targets: uniform replay only")
            if len(traces) < self.n_step:  # not one full window yet
                return False
            folded, iota = self._nstep.fold(traces, self._nstep.sample_indices(traces, self.draw_ctr))
            if net.layers == 3:
                dqn3_grad(folded, net.hidden, net.n_out, net.act, net.params, net.packed, self.approximator.target,
                          self.approximator.target_packed, self.batchsize, self._nstep.gamma_n, self.delta, self.seed, self.draw_ctr,
                          iota, self.workspace, self.grad, self.loss, self.td)
            else:
                call("rlhip_dqn_grad_idx_f32", C.byref(folded.rb), net.hidden, net.n_out, net.act, ptr(net.params),
                     ptr(self.approximator.target), self.batchsize, ptr(iota), self._nstep.gamma_n, self.delta, ptr(self.workspace),
                     ptr(self.grad), ptr(self.loss), ptr(self.td), stream_ptr())
            self.draw_ctr += 1
            self.approximator.optimise_(self.grad, clip_norm=self.max_grad_norm, grad_scale=1.0)
            self.n_updates += 1
            return True
        beta = 0.0
        if prioritized:
            beta = float(self.per_beta(self.n_updates)) if callable(self.per_beta) else float(self.per_beta)
        if net.layers == 3:
            idx = None
            if prioritized:  # prioritized BatchSampler: keys + priorities from the device sum-tree
                idx, self._key, self._prio = traces.sample_prioritized(self.batchsize, self.seed, self.draw_ctr)
            if prioritized and beta > 0.0:
                call("rlhip_per_is_weights_f32", ptr(self._prio), self.batchsize, beta, ptr(self.is_weights), stream_ptr())
                call("rlhip_dqn3_grad_w_f32", C.byref(traces.rb), net.hidden, net.n_out, net.act, ptr(net.params), ptr(net.packed),
                     ptr(self.approximator.target), ptr(self.approximator.target_packed), self.batchsize, ptr(idx),
                     ptr(self.is_weights), self.gamma, self.delta, ptr(self.workspace), ptr(self.grad), ptr(self.loss),
                     ptr(self.td), stream_ptr())
            else:
                dqn3_grad(traces, net.hidden, net.n_out, net.act, net.params, net.packed, self.approximator.target,
                          self.approximator.target_packed, self.batchsize, self.gamma, self.delta, self.seed,
                          self.draw_ctr, idx, self.workspace, self.grad, self.loss, self.td)
            if prioritized:  # trajectory[:priority, keys] = (|td| + eps)^alpha  (PrioritizedDQN write-back)
                call("rlhip_per_priority_f32", ptr(self.td), self.batchsize, self.per_eps, self.per_alpha,
                     ptr(self.td), stream_ptr())
                traces.set_priority_(self._key, self.td)
        elif prioritized:
            idx, self._key, self._prio = traces.sample_prioritized(self.batchsize, self.seed, self.draw_ctr)
            if beta > 0.0:
                call("rlhip_per_is_weights_f32", ptr(self._prio), self.batchsize, beta, ptr(self.is_weights), stream_ptr())
                call("rlhip_dqn_grad_idx_w_f32", C.byref(traces.rb), net.hidden, net.n_out, net.act, ptr(net.params),
                     ptr(self.approximator.target), self.batchsize, ptr(idx), ptr(self.is_weights), self.gamma, self.delta,
                     ptr(self.workspace), ptr(self.grad), ptr(self.loss), ptr(self.td), stream_ptr())
            else:
                call("rlhip_dqn_grad_idx_f32", C.byref(traces.rb), net.hidden, net.n_out, net.act, ptr(net.params),
                     ptr(self.approximator.target), self.batchsize, ptr(idx), self.gamma, self.delta, ptr(self.workspace),
                     ptr(self.grad), ptr(self.loss), ptr(self.td), stream_ptr())
            call("rlhip_per_priority_f32", ptr(self.td), self.batchsize, self.per_eps, self.per_alpha, ptr(self.td),
                 stream_ptr())
            traces.set_priority_(self._key, self.td)
        else:
            dqn_grad(traces, net.hidden, net.n_out, net.act, net.params, self.approximator.target, self.batchsize,
                     self.gamma, self.delta, self.seed, self.draw_ctr, self.workspace, self.grad, self.loss)
        self.draw_ctr += 1
        scale = 1.0
        if self.process_group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(self.process_group)
            if world > 1:
                if not hasattr(self, "_hipcomm"):  # rlhip_comm_* (csrc/comm.hip): p2p kernel or RCCL behind one call
                    from .dist import HipComm

                    self._hipcomm = HipComm.create(self.process_group, self.grad.numel(), self.grad.device)
                if self._hipcomm.ok:
                    self._hipcomm.all_reduce_(self.grad)
                    self._hipcomm.check()
                else:
                    dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.process_group)
                scale = 1.0 / world
        self.approximator.optimise_(self.grad, clip_norm=self.max_grad_norm, grad_scale=scale)
        self.n_updates += 1
        return True


class QBasedPolicy:
    """QBasedPolicy(; learner, explorer)  q_based_policy.jl:13-49.  plan! on the vector env is one fused
    launch (Q forward + eps-greedy)."""

    def __init__(self, learner, explorer):
        self.learner, self.explorer = learner, explorer
        self._actions = None
        self._q = None

    def plan_(self, env):
        net = self.learner.approximator.network
        ex = self.explorer
        if ex.is_break_tie:  # tie-break variant: unfused path (forward, then the selection kernel)
            return ex.plan_(net.forward(env.state()), env_id_base=env.env_id_base)
        eps = ex.get_eps()
        step = ex.step
        ex.step += 1
        if net.layers == 3:
            self._actions, self._q = dqn3_plan(net.params, net.packed, net.n_in, net.hidden, net.n_out, net.act,
                                               env.state(), eps, ex.seed, env.env_id_base, step, self._actions,
                                               self._q)
        else:
            self._actions, self._q = dqn_plan(net.params, net.n_in, net.hidden, net.n_out, net.act, env.state(), eps,
                                              ex.seed, env.env_id_base, step, self._actions, self._q)
        return self._actions + 1

    def optimise_(self, trajectory):
        """optimise!(policy, stage, trajectory) -> optimise!(learner, stage, trajectory)  (:49)."""
        return self.learner.optimise_(trajectory)
