"""the headline workload (4096 CartPole envs, T = 32, 2-layer actor / critic, 4 epochs x 4 micro-batches) for profiling:
python tools/ppo2_one.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
env = rlhip.HipVecEnv("cartpole", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=32, hidden=256, seed=7)
for _ in range(iters):
    pol.rollout_(); pol.update_()
torch.cuda.synchronize()
print("loss", float(pol.losses[0]))
