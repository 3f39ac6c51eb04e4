"""oracle.dqn_run -- the per-stage DQN agent loop on the oracle's vector env.  TEST INFRASTRUCTURE ONLY (see __init__.py).

What it restates (one vec-step = one pass of the loop body, the historical MultiThreadEnv specialisation of `_run`:
docs/homepage/blog/an_introduction_to_reinforcement_learning_jl_design_implementations_thoughts/index.md:351-374):

    RLCore/src/core/run.jl:52-67                 plan! -> push!(PreActStage) -> act! -> push!(PostActStage) -> optimise!(PostActStage) -> check!
    RLCore/src/policies/agent/agent_base.jl:45-59   PreEpisode/first PreAct: push (state,);  PostAct: push (state = s', action, reward, terminal)
    RLCore/src/policies/q_based_policy.jl:13-49  plan! = explorer(forward(learner, env)); optimise! forwards to the learner
    RLCore/src/policies/explorers/epsilon_greedy_explorer.jl:69-112   get_eps(step); step += 1; u >= eps ? argmax_first : rand(1:n)
    RLCore/src/policies/learners/target_network.jl:70-88              optimise!: update the network; every sync_freq calls dest = rho dest + (1 - rho) src
    RLCore/src/policies/learners/flux_approximator.jl:46              Flux.Optimise.update!(Adam)
    removed Zoo DQN learner (PARITY UNPINNED): batch from BatchSampler, y = r + gamma (1 - t) max Qt(s'),
        Huber(delta), mean over the batch, [clip_by_global_norm!], Adam, target sync
    InsertSampleRatioController / min_replay_history gate (docs/src/How_to_implement_a_new_algorithm.md:108)

Every numeric piece is one of the oracle's C functions (rlo_env_step, rlo_mlp2_forward_f32, rlo_eps_greedy_select_f32,
rlo_ring_push_*, rlo_ring_sample_indices, rlo_ring_gather, rlo_dqn_loss_grad_f32, rlo_clip_by_global_norm_f32, rlo_adam_f32,
rlo_polyak_f32, rlo_target_sync_due); this file only sequences them the way the reference's stages do -- it is the checker
of the fused `rlhip_dqn_vec_step_f32` (one C-ABI call per vec-step), which until round 6 was verified only against the
GPU's own per-stage path (missing item 5)."""
import numpy as np

from . import binding as B


class DQNRun:
    """State of the loop (so that a test can continue a run, inspect counters, or compare mid-way)."""

    def __init__(self, kind="cartpole", n=4096, ns=4, na=2, hidden=128, act=0, env_seed=5, net_seed=5, explorer_seed=5,
                 sampler_seed=5, capacity=256, batch=512, gamma=0.99, huber_delta=1.0, lr=1e-3, beta1=0.9, beta2=0.999,
                 adam_eps=1e-8, max_grad_norm=0.0, sync_freq=100, rho=0.0, min_replay_history=None, update_freq=1,
                 eps_stable=0.01, eps_kind="exp", eps_init=1.0, warmup_steps=0, decay_steps=500, env_id_base=0, params=None, layers=2, n_step=1):
        """layers = 3: the blog's Chain(Dense(ns, h, act), Dense(h, h, act), Dense(h, na)) with the bf16 hidden layer of rlo_mlp3.c"""
        self.env = B.VecEnv(kind, n, seed=env_seed, env_id_base=env_id_base, continuous=False)
        self.n, self.ns, self.na, self.h, self.act = n, ns, na, hidden, act
        self.layers, self.n_step = layers, n_step   # n_step > 1: NStepBatchSampler + gamma^n target (rlo_buffer.c)
        self._init, self._fwd = (B.mlp2_init, B.mlp2_forward) if layers == 2 else (B.mlp3_init, B.mlp3_forward)
        self.params = self._init(ns, hidden, na, net_seed, 0) if params is None else np.array(params, np.float32)
        self.target = self.params.copy()                      # TargetNetwork: deepcopy of the model (target_network.jl:56-58)
        self.m, self.v = np.zeros_like(self.params), np.zeros_like(self.params)
        self.ring = B.Ring(capacity, n, ns)
        self.capacity, self.batch = capacity, batch
        self.gamma, self.delta, self.max_grad_norm = gamma, huber_delta, max_grad_norm
        self.lr, self.beta1, self.beta2, self.adam_eps = lr, beta1, beta2, adam_eps
        self.sync_freq, self.rho = sync_freq, rho
        self.min_replay_history = n if min_replay_history is None else min_replay_history
        self.update_freq = update_freq
        self.eps = dict(kind=eps_kind, eps_stable=eps_stable, eps_init=eps_init, warmup_steps=warmup_steps, decay_steps=decay_steps)
        self.explorer_seed, self.sampler_seed, self.env_id_base = explorer_seed, sampler_seed, env_id_base
        self.explorer_step = 1      # EpsilonGreedyExplorer(; step = 1)
        self.draw_ctr = 0           # BatchSampler draws so far
        self.n_updates = 0          # Adam steps so far (t of the bias correction is n_updates + 1)
        self.n_optimise = 0         # TargetNetwork.n_optimise
        self.vec_steps = 0
        self.n_inserted = self.n_sampled = 0   # InsertSampleRatioController(ratio = 1, threshold = 1)
        self.started = False
        self.losses = []
        self.actions = []           # 0-based actions of every vec-step (for the comparison with the ring of the GPU run)

    def _controller_allows(self):
        if self.n_inserted >= 1 and self.n_sampled <= (self.n_inserted - 1) * 1.0:
            self.n_sampled += 1
            return True
        return False

    def vec_step(self, force_actions=None):
        """one vec-step.  force_actions (0-based int32, optional): teacher forcing -- the oracle still plans (self.last_q,
        self.last_plan hold its own Q values and decision) but act! / push! use the given actions, so that a step-by-step
        comparison with another implementation does not inherit the chaos of the env."""
        env = self.env
        if not self.started:        # push!(agent, PreEpisodeStage / first PreActStage): (state,)
            self.ring.push_state(env.obs())
            self.started = True
        # plan!(policy, env)
        eps = B.get_eps(self.eps["kind"], self.eps["eps_stable"], self.eps["eps_init"], self.eps["warmup_steps"],
                        self.eps["decay_steps"], self.explorer_step)
        step = self.explorer_step
        self.explorer_step += 1
        q = self._fwd(self.params, self.ns, self.h, self.na, self.act, env.obs())
        a0 = B.eps_greedy_select(q, eps, self.explorer_seed, step, env_id_base=self.env_id_base)
        self.last_q, self.last_plan, self.last_eps = q, a0, eps
        if force_actions is not None:
            a0 = np.ascontiguousarray(force_actions, np.int32)
        # act!(env, action) (instances auto-reset: the vector env has no episode stages)
        env.step(a0)
        # push!(agent, PostActStage, env, action): (state = s', action, reward, terminal)
        self.ring.push_transition(env.obs(), a0, env.reward.astype(np.float32), env.done)
        self.n_inserted += 1
        self.actions.append(a0.copy())
        # optimise!(agent, PostActStage)
        self.vec_steps += 1
        updated = False
        if len(self.ring) * self.n >= self.min_replay_history and self.vec_steps % self.update_freq == 0 and self._controller_allows():
            if self.n_step > 1 and len(self.ring) < self.n_step:
                return False
            gamma = self.gamma
            if self.n_step > 1:
                idx = B.ring_sample_indices_nstep(self.ring, self.batch, self.n_step, self.sampler_seed, self.draw_ctr)
                s, a, r, t, sn = B.ring_gather_nstep(self.ring, idx, self.n_step, self.gamma)
                gamma = B.gamma_pow(self.gamma, self.n_step)
            else:
                idx = self.ring.sample_indices(self.batch, self.sampler_seed, self.draw_ctr)
                s, a, r, t, sn = self.ring.gather(idx)
            self.draw_ctr += 1
            if self.layers == 2:
                loss, g = B.dqn_loss_grad(self.ns, self.h, self.na, self.act, self.params, self.target, s, a, r, t, sn, gamma, self.delta)
            else:
                loss, g, _ = B.dqn3_loss_grad(self.ns, self.h, self.na, self.act, self.params, self.target, s, a, r, t, sn, gamma, self.delta)
            if self.max_grad_norm > 0.0:
                B.clip_by_global_norm(g, self.max_grad_norm)
            B.adam(self.params, g, self.m, self.v, self.lr, self.beta1, self.beta2, self.adam_eps, self.n_updates + 1)
            self.n_updates += 1
            due, self.n_optimise = B.target_sync_due(self.n_optimise, self.sync_freq)
            if due:
                B.polyak(self.target, self.params, self.rho)
            self.losses.append(loss)
            updated = True
        return updated

    def run(self, k):
        for _ in range(k):
            self.vec_step()
        return self


def dqn_run(k, **kw):
    """k vec-steps of the per-stage DQN agent loop from a fresh state; returns the DQNRun (params, target, m, v, ring, counters)."""
    return DQNRun(**kw).run(k)
