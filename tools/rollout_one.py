"""the headline rollout (4096 CartPole envs, T = 32, actor / critic 4 -> 256 -> .) for profiling: python tools/rollout_one.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = rlhip.HipVecEnv("cartpole", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=32, hidden=256, seed=7)
for _ in range(iters):
    pol.rollout_()
torch.cuda.synchronize()
print("episode", float(env._st_episode.float().mean()) if hasattr(env, "_st_episode") else "")
