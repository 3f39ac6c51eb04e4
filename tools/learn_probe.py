"""learning curves of the four learners VERDICT r5 item 4 names (probe for the thresholds of tests/test_gpu_learn.py)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "reinforcementlearning.jl_amd"); sys.path.insert(0, ".")
import rlhip as rl

def dqn(layers, batch=512, chunks=16, chunk=500, hidden=128):
    n = 4096
    env = rl.CartPoleEnv(n, seed=5)
    net = rl.HipApproximator(4, hidden, 2, seed=5, layers=layers)
    learner = rl.DQNLearner(rl.TargetNetwork(net, sync_freq=100), batchsize=batch, min_replay_history=n, seed=5)
    policy = rl.QBasedPolicy(learner, rl.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
    tr = rl.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)
    agent = rl.Agent(policy, rl.Trajectory(tr))
    out = []
    t0 = time.time()
    for c in range(chunks):
        rl.run_fused_dqn(agent, env, rl.StopAfterNSteps(chunk))
        torch.cuda.synchronize()
        L = min(len(tr), 256)
        term = float(tr.terminal[:257].sum())  # the record ring's terminal view (slot-indexed); all slots valid once full
        out.append(round(L * n / max(term, 1.0), 1))
    print(f"dqn layers={layers} batch={batch} hidden={hidden}: ep_len per {chunk} vec-steps:", out, f"({time.time()-t0:.1f}s)", flush=True)

def ppo(hidden, iters=60, lr=1e-3, **kw):
    n, T = 4096, 128
    env = rl.HipVecEnv("pendulum", n, seed=7)
    pol = rl.PPOPolicy(env, update_freq=T, hidden=hidden, seed=7, clip_range=0.1, layers=3, lr=lr, **kw)
    out = []
    t0 = time.time()
    for it in range(iters):
        pol.rollout_(); pol.update_()
        if it % 5 == 0 or it == iters - 1:
            out.append(round(float(pol.trajectory.reward.mean()), 3))
    print(f"ppo3 hidden={hidden} lr={lr} {kw}: mean reward/step every 5 iters:", out, f"({time.time()-t0:.1f}s)", flush=True)

dqn(2); dqn(3); dqn(2, batch=4096); dqn(2, batch=32)
ppo(128); ppo(256); ppo(128, iters=120, lr=3e-4); ppo(128, iters=60, entropy_loss_weight=0.0)
