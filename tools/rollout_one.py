"""the headline rollout (4096 CartPole envs, T = 32, actor / critic 4 -> 256 -> .) for profiling:
python tools/rollout_one.py [iters] [kind = cartpole] [T = 32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kind = sys.argv[2] if len(sys.argv) > 2 else "cartpole"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
env = rlhip.HipVecEnv(kind, 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=T, hidden=256, seed=7)
for _ in range(iters):
    pol.rollout_()
torch.cuda.synchronize()
print("episode", float(env._st_episode.float().mean()) if hasattr(env, "_st_episode") else "")
