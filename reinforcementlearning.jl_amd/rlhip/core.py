"""run(policy, env, stop_condition, hook) for the vectorised env -- host mirror of the reference's
experiment loop, stages, stop conditions, hooks, Agent and RandomPolicy.

Reference files mirrored (src/ReinforcementLearningCore/src/...):
    core/run.jl:22-78            run / _run (scalar loop with 6 stages)
    core/stages.jl:10-37         stage types, default no-op push!/optimise!
    core/stop_conditions.jl      StopAfterNSteps (:48-76), StopAfterNEpisodes (:87-119), StopIfAny/All (:18-38)
    core/hooks.jl                EmptyHook, ComposedHook (:37-62), StepsPerEpisode, TotalRewardPerEpisode (:146-196),
                                 BatchStepsPerEpisode (:202-231), TimePerStep (:243-262), DoEveryNSteps
    policies/agent/agent_base.jl Agent: PreEpisode push (state,), PostAct push (state, action, reward, terminal)
    policies/random_policy.jl    RandomPolicy
and the vector-env specialisation of `_run` that the historical MultiThreadEnv used
(docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:351-374):
no episode stages; per vec-step  plan! -> PreAct push -> act! -> PostAct push -> stop check.

Host-side only: counters and callbacks.  Everything touching env state, trajectories or parameters is
a HIP kernel launched through the C ABI by the objects this loop drives.
"""
import time

import torch

# ------------------------------------------------------------------------------------- stages
PRE_EXPERIMENT_STAGE = "PreExperimentStage"
POST_EXPERIMENT_STAGE = "PostExperimentStage"
PRE_EPISODE_STAGE = "PreEpisodeStage"
POST_EPISODE_STAGE = "PostEpisodeStage"
PRE_ACT_STAGE = "PreActStage"
POST_ACT_STAGE = "PostActStage"


# ---------------------------------------------------------------------------- stop conditions
class StopAfterNSteps:
    """StopAfterNSteps(step; cur = 1): `check!` returns cur >= step, THEN increments (stop_conditions.jl:65-69),
    so exactly `step` steps are executed (test core/stop_conditions.jl:8: 11 trues in 20 calls for n = 10).
    On the vector env one check = one vec-step (the reference's MultiThreadEnv loop counted the same way)."""

    def __init__(self, step, cur=1):
        self.step, self.cur = int(step), int(cur)

    def check_(self, policy=None, env=None):
        res = self.cur >= self.step
        self.cur += 1
        return res


class StopAfterNEpisodes:
    """StopAfterNEpisodes(episode; cur = 0)  stop_conditions.jl:87-119.  On the vector env every terminated
    instance counts as one episode (one device->host read of the terminal flags per check)."""

    def __init__(self, episode, cur=0):
        self.episode, self.cur = int(episode), int(cur)

    def check_(self, policy, env):
        self.cur += int(env.is_terminated().sum())
        return self.cur >= self.episode


class StopIfAny:
    def __init__(self, *conds):
        self.conds = conds

    def check_(self, policy, env):
        return any([c.check_(policy, env) for c in self.conds])  # all are evaluated, like check!.(...)


class StopIfAll:
    def __init__(self, *conds):
        self.conds = conds

    def check_(self, policy, env):
        return all([c.check_(policy, env) for c in self.conds])


class StopAfterNSeconds:
    def __init__(self, seconds):
        self.deadline = time.time() + seconds

    def check_(self, policy, env):
        return time.time() > self.deadline


# -------------------------------------------------------------------------------------- hooks
class EmptyHook:
    def push_(self, stage, policy, env):
        return None

    def __add__(self, other):
        return ComposedHook(self, other)


class ComposedHook(EmptyHook):
    """ComposedHook(hooks...)  hooks.jl:37-62"""

    def __init__(self, *hooks):
        self.hooks = list(hooks)

    def push_(self, stage, policy, env):
        for h in self.hooks:
            h.push_(stage, policy, env)


class BatchStepsPerEpisode(EmptyHook):
    """BatchStepsPerEpisode(batchsize)  hooks.jl:202-231 -- vector-env aware; the per-env step counters are
    device tensors, finished episode lengths are pulled to the host only when episodes finish."""

    def __init__(self, batchsize, device="cuda"):
        self.steps = [[] for _ in range(batchsize)]
        self.step = torch.zeros(batchsize, dtype=torch.int64, device=device)

    def push_(self, stage, policy, env):
        if stage != POST_ACT_STAGE:
            return
        self.step += 1
        t = env.is_terminated()
        if bool(t.any()):
            idx = torch.nonzero(t).flatten().tolist()
            vals = self.step[t].tolist()
            for i, v in zip(idx, vals):
                self.steps[i].append(v)
            self.step[t] = 0


class TotalBatchRewardPerEpisode(EmptyHook):
    """TotalRewardPerEpisode (hooks.jl:146-196) for the vector env: running return per instance on the
    device, finished returns appended to `rewards[i]`."""

    def __init__(self, batchsize, device="cuda"):
        self.rewards = [[] for _ in range(batchsize)]
        self.reward = torch.zeros(batchsize, dtype=torch.float64, device=device)

    def push_(self, stage, policy, env):
        if stage != POST_ACT_STAGE:
            return
        self.reward += env.reward().to(torch.float64)
        t = env.is_terminated()
        if bool(t.any()):
            idx = torch.nonzero(t).flatten().tolist()
            vals = self.reward[t].tolist()
            for i, v in zip(idx, vals):
                self.rewards[i].append(v)
            self.reward[t] = 0


class DeviceEpisodeStats(EmptyHook):
    """TotalRewardPerEpisode + BatchStepsPerEpisode (hooks.jl:146-231) with the accumulators and the finished-episode
    log on the device (hooks.hip): `push_` is one kernel launch and never synchronises; `.steps[i]` / `.rewards[i]`
    (lists per env instance, as the reference hooks expose them) are materialised from the device log on access."""

    def __init__(self, batchsize, log_capacity=1 << 20, device="cuda"):
        import numpy as np

        self.n, self.cap, self.vec_step = int(batchsize), int(log_capacity), 0
        self._steps_acc = torch.zeros(self.n, dtype=torch.int32, device=device)
        self._ret_acc = torch.zeros(self.n, dtype=torch.float64, device=device)
        self._log = torch.zeros((self.cap, 3), dtype=torch.int64, device=device)  # 24-byte records
        self._count = torch.zeros(1, dtype=torch.int32, device=device)
        self._np = np

    def push_(self, stage, policy, env):
        if stage != POST_ACT_STAGE:
            return
        from ._lib import call
        from .ops import ptr, stream_ptr

        r = env.reward()
        if r.dtype != torch.float32:
            r = r.to(torch.float32)
        call("rlhip_hook_episode_stats", ptr(r), ptr(env._done), self.n, self.vec_step, ptr(self._steps_acc),
             ptr(self._ret_acc), ptr(self._log), self.cap, ptr(self._count), stream_ptr())
        self.vec_step += 1

    def records(self):
        """finished episodes as a structured numpy array sorted by (vec_step, env)"""
        np = self._np
        count = int(self._count.item())
        if count > self.cap:
            raise OverflowError(f"episode log overflow: {count} episodes finished, capacity {self.cap}")
        dt = np.dtype([("vec_step", np.uint32), ("env", np.uint32), ("steps", np.int32), ("pad", np.int32),
                       ("total_reward", np.float64)])
        rec = self._log[:count].cpu().numpy().view(dt).reshape(-1)
        return np.sort(rec, order=["vec_step", "env"])

    @property
    def steps(self):
        out = [[] for _ in range(self.n)]
        for e, s in zip(self.records()["env"].tolist(), self.records()["steps"].tolist()):
            out[e].append(s)
        return out

    @property
    def rewards(self):
        rec = self.records()
        out = [[] for _ in range(self.n)]
        for e, v in zip(rec["env"].tolist(), rec["total_reward"].tolist()):
            out[e].append(v)
        return out


class StepsPerEpisode(EmptyHook):
    """StepsPerEpisode  hooks.jl:64-101, for a single-instance env (n_envs = 1)."""

    def __init__(self):
        self.steps, self.count = [], 0

    def push_(self, stage, policy, env):
        if stage == POST_ACT_STAGE:
            self.count += 1
            if bool(env.is_terminated()[0]):
                self.steps.append(self.count)
                self.count = 0
        elif stage == POST_EXPERIMENT_STAGE and self.count > 0:
            self.steps.append(self.count)
            self.count = 0


class TimePerStep(EmptyHook):
    """TimePerStep(; max_steps = 100)  hooks.jl:243-262"""

    def __init__(self, max_steps=100):
        self.times, self.max_steps, self.t = [], max_steps, time.time()

    def push_(self, stage, policy, env):
        if stage == POST_ACT_STAGE:
            now = time.time()
            self.times.append(now - self.t)
            self.times = self.times[-self.max_steps:]
            self.t = now


class DoEveryNSteps(EmptyHook):
    """DoEveryNSteps(f; n = 1, t = 0)  hooks.jl:265-290"""

    def __init__(self, f, n=1, t=0):
        self.f, self.n, self.t = f, n, t

    def push_(self, stage, policy, env):
        if stage == POST_ACT_STAGE:
            self.t += 1
            if self.t % self.n == 0:
                self.f(self.t, policy, env)


# ------------------------------------------------------------------------------------ policies
class RandomPolicy:
    """RandomPolicy(action_space; rng)  random_policy.jl:18-32: plan! = rand(rng, action_space).  The draws
    come from the EXPLORE Philox stream (one block per env per call), selection by the eps-greedy kernel
    with eps = 1 (pure random branch)."""

    def __init__(self, action_space=None, seed=0):
        self.action_space, self.seed, self.step = action_space, int(seed), 1
        self._dummy = None

    def plan_(self, env):
        from .ops import eps_greedy_select

        sp = self.action_space or env.action_space()
        step = self.step
        self.step += 1
        if sp.n is None:  # continuous: uniform in the box from the SYNTH stream
            from .ops import fill_uniform

            u = fill_uniform(env.n, self.seed, step, 7, env.device)
            lo, hi = sp.lo[0], sp.hi[0]
            return (u * (hi - lo) + lo).to(env.T)
        if self._dummy is None or self._dummy.shape != (sp.n, env.n):
            self._dummy = torch.zeros((sp.n, env.n), dtype=torch.float32, device=env.device)
        return eps_greedy_select(self._dummy, 1.0, self.seed, step, env.env_id_base) + 1

    def push_(self, stage, env, action=None):
        return None

    def optimise_(self, stage):
        return None


class Agent:
    """Agent(policy, trajectory)  agent_base.jl:18-66 on the vector env: the PreEpisode push of (state,)
    happens once (there are no episode stages; instances auto-reset), PostAct pushes
    (state = s', action, reward, terminal) as ONE batched device push."""

    def __init__(self, policy, trajectory):
        self.policy, self.trajectory = policy, trajectory
        self._started = False

    def plan_(self, env):
        return self.policy.plan_(env)

    def push_(self, stage, env, action=None):
        if stage == PRE_EXPERIMENT_STAGE or (stage == PRE_ACT_STAGE and not self._started):
            if not self._started:
                self.trajectory.push_state_(env.state().to(torch.float32))
                self._started = True
        elif stage == POST_ACT_STAGE:
            a0 = (action - 1).to(torch.int32) if not env.continuous else action
            self.trajectory.push_transition_(env.state().to(torch.float32), a0.contiguous(),
                                             env.reward().to(torch.float32), env._done)

    def optimise_(self, stage):
        if stage == POST_ACT_STAGE and hasattr(self.policy, "optimise_"):
            return self.policy.optimise_(self.trajectory)
        return None


class PPOAgent:
    """Agent(policy = PPOPolicy, trajectory = PPOTrajectory) with the per-step drop-in protocol:
    PreAct push (state, action, action_log_prob), PostAct push (reward, terminal), update every
    `update_freq` vec-steps (blog index.html:15238-15287)."""

    def __init__(self, policy):
        self.policy = policy

    def plan_(self, env):
        return self.policy.plan_(env)

    def push_(self, stage, env, action=None):
        if stage == PRE_ACT_STAGE:
            self.policy.push_preact_(env)
        elif stage == POST_ACT_STAGE:
            self.policy.push_postact_(env)

    def optimise_(self, stage):
        if stage == POST_ACT_STAGE and self.policy.n_pushed == self.policy.T:
            self.policy.finish_rollout_()
            self.policy.update_()


# ----------------------------------------------------------------------------------------- run
def run(policy, env, stop_condition=None, hook=None):
    """run(policy, env, stop_condition, hook) -> hook   (core/run.jl:22-31) for a HipVecEnv.

    `_run(policy, env::HipVecEnv, ...)`: the MultiThreadEnv specialisation -- no episode stages:
        push!(hook/policy, PreExperimentStage)
        loop:  action = plan!(policy, env); push!(policy, PreActStage, env, action); push!(hook, PreActStage)
               act!(env, action)
               push!(policy, PostActStage, env, action); optimise!(policy, PostActStage); push!(hook, PostActStage)
               check!(stop_condition) -> break
        push!(policy/hook, PostExperimentStage)
    """
    stop_condition = stop_condition or StopAfterNSteps(1)
    hook = hook or EmptyHook()
    from .timing import timer as tm  # `@timeit_debug timer "<label>"` of run.jl:46-72, no-ops unless enabled

    hook.push_(PRE_EXPERIMENT_STAGE, policy, env)
    policy.push_(PRE_EXPERIMENT_STAGE, env)
    while True:
        with tm("plan!"):
            action = policy.plan_(env)
        with tm("push!(policy) PreActStage"):
            policy.push_(PRE_ACT_STAGE, env, action)
        with tm("push!(hook) PreActStage"):
            hook.push_(PRE_ACT_STAGE, policy, env)
        with tm("act!"):
            env.act_(action)
        with tm("push!(policy) PostActStage"):
            policy.push_(POST_ACT_STAGE, env, action)
        with tm("optimise! PostActStage"):
            policy.optimise_(POST_ACT_STAGE)
        with tm("push!(hook) PostActStage"):
            hook.push_(POST_ACT_STAGE, policy, env)
        if stop_condition.check_(policy, env):
            break
    policy.push_(POST_EXPERIMENT_STAGE, env)
    hook.push_(POST_EXPERIMENT_STAGE, policy, env)
    return hook


def run_fused_dqn(agent, env, stop_condition=None, hook=None):
    """`run(agent, env, stop, hook)` for Agent{QBasedPolicy{DQNLearner}} on a HipVecEnv with the whole loop body
    (plan! -> act! -> push! -> optimise!) as ONE C-ABI call per vec-step (rlhip_dqn_vec_step_f32).  Same kernels,
    same order, same counters as `run`: parameters, trajectory and explorer state end bit-identical
    (tests/test_gpu_run.py).  Hooks see the PostAct stage of every step."""
    import ctypes as C

    from . import _lib
    from ._lib import call
    from .ops import ptr, stream_ptr

    policy, traj = agent.policy, agent.trajectory
    learner, ex = policy.learner, policy.explorer
    tn = learner.approximator
    net = tn.network
    traces = traj.container
    if ex.is_break_tie or env.continuous or env.is_f64 or hasattr(traces, "sample_prioritized") \
            or learner.process_group is not None or getattr(learner, "n_step", 1) != 1:
        raise NotImplementedError("fused DQN step: plain eps-greedy, Float32 discrete env, uniform 1-step replay, 1 GPU")
    stop_condition = stop_condition or StopAfterNSteps(1)
    hook = hook or EmptyHook()
    hook.push_(PRE_EXPERIMENT_STAGE, agent, env)
    agent.push_(PRE_EXPERIMENT_STAGE, env)
    dev = env.device
    if policy._actions is None:
        policy._actions = torch.empty(env.n, dtype=torch.int32, device=dev)
        policy._q = torch.empty((net.n_out, env.n), dtype=torch.float32, device=dev)
    a = _lib.DqnStepArgs()
    a.kind, a.env_cfg, a.st, a.n = env.kind, C.addressof(env.cfg), C.addressof(env._st), env.n
    a.env_seed, a.env_id_base = env.seed, env.env_id_base
    a.obs, a.last_obs = ptr(env.state()), ptr(env._last_obs)
    a.ring = C.addressof(traces.rb)
    a.layers, a.h, a.na, a.act = net.layers, net.hidden, net.n_out, net.act
    a.params, a.target = ptr(net.params), ptr(tn.target)
    a.packed = ptr(net.packed) if net.layers == 3 else None
    a.target_packed = ptr(tn.target_packed) if net.layers == 3 else None
    a.m, a.v, a.beta_pow = ptr(net.m), ptr(net.v), ptr(net.beta_pow)
    a.lr, a.beta1, a.beta2, a.adam_eps = net.lr, net.beta1, net.beta2, net.eps
    a.max_grad_norm, a.grad_scale = learner.max_grad_norm, 1.0
    a.explorer_seed, a.batch, a.gamma, a.huber_delta = ex.seed, learner.batchsize, learner.gamma, learner.delta
    a.sampler_seed, a.rho = learner.seed, tn.rho
    a.workspace, a.grad, a.loss, a.gn = ptr(learner.workspace), ptr(learner.grad), ptr(learner.loss), ptr(net.gn)
    a.actions, a.q = ptr(policy._actions), ptr(policy._q)
    ctrl = traj.controller
    s = stream_ptr()
    while True:
        a.eps, a.explorer_step = ex.get_eps(), ex.step
        ex.step += 1
        # the transition pushed by this call counts towards min_replay_history / the sample-ratio controller
        ctrl.on_insert_(1)
        n_after = min(len(traces) + 1, traces.capacity) * traces.n_env
        a.do_update = int(learner.should_update_(traj, n_after))  # warm-up, update_freq, sample / insert controller
        a.draw_ctr = learner.draw_ctr
        a.do_sync = int(a.do_update and (tn.n_optimise + 1) % tn.sync_freq == 0)
        call("rlhip_dqn_vec_step_f32", C.byref(a), s)
        if a.do_update:
            learner.draw_ctr += 1
            learner.n_updates += 1
            tn.n_optimise = 0 if a.do_sync else tn.n_optimise + 1
        hook.push_(POST_ACT_STAGE, agent, env)
        if stop_condition.check_(agent, env):
            break
    env._obs_valid = True
    agent.push_(POST_EXPERIMENT_STAGE, env)
    hook.push_(POST_EXPERIMENT_STAGE, agent, env)
    return hook


def run_fused_ppo(policy, env, n_updates, hook=None):
    """The same loop with the T-step rollout fused into one launch per update period (hooks see one
    PostActStage per period with the LAST step's reward / terminal flags)."""
    hook = hook or EmptyHook()
    hook.push_(PRE_EXPERIMENT_STAGE, policy, env)
    for _ in range(n_updates):
        policy.rollout_(env)
        policy.update_()
        hook.push_(POST_ACT_STAGE, policy, env)
    hook.push_(POST_EXPERIMENT_STAGE, policy, env)
    return hook
