"""Unfused sequence grad -> reduce(no apply) -> clip_adam, for kernel-level comparison under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
env = rlhip.HipVecEnv("cartpole", 4096, seed=1)
pol = rlhip.PPOPolicy(env, update_freq=32)
pol.rollout_(); pol.gae_()
for i in range(200):
    pol.grad_(i // 4, i % 4)
    pol.apply_(1.0)
torch.cuda.synchronize()
print("done")
