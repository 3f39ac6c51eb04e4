"""one config of the 3-layer PPO learner for profiling: python tools/ppo3_one.py [kind] [T] [iters] [act]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
kind = sys.argv[1] if len(sys.argv) > 1 else "pendulum"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 0 relu, 1 tanh
env = rlhip.HipVecEnv(kind, 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=T, hidden=128, seed=7, layers=3, clip_range=0.1, act=act)
for _ in range(iters):
    pol.rollout_(); pol.update_()
torch.cuda.synchronize()
print("loss", float(pol.losses[0]))
