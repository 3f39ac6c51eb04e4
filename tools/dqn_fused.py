"""the fused DQN vec-step loop alone (BASELINE configs[1]: 4096-env CartPole, 4 -> 128 -> 2, batch 512), for rocprofv3"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
n = 4096
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
env = rlhip.CartPoleEnv(n, seed=5)
net = rlhip.HipApproximator(4, 128, 2, seed=5, layers=layers)
learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=100), batchsize=batch, min_replay_history=n, seed=5)
policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
agent = rlhip.Agent(policy, rlhip.Trajectory(rlhip.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)))
rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(50))
torch.cuda.synchronize()
t0 = time.perf_counter()
rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(3000))
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"layers {layers} batch {batch}: {el / 3000 * 1e6:.2f} us per vec-step, {n * 3000 / el:.3e} env-steps/s")
