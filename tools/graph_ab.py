"""Headline iteration: eager launches (rollout_ + update_) vs one HIP-graph replay per iteration (capture_graph_ / replay_)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch  # noqa: E402

import rlhip  # noqa: E402

n, T, iters = 4096, 32, 300


def make():
    env = rlhip.HipVecEnv("cartpole", n, seed=123)
    return env, rlhip.PPOPolicy(env, update_freq=T, hidden=256, seed=123)


env, pol = make()
for _ in range(20):
    pol.rollout_()
    pol.update_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    pol.rollout_()
    pol.update_()
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / iters * 1e3
env, pol = make()
pol.capture_graph_(warmup=3)
for _ in range(20):
    pol.replay_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    pol.replay_()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / iters * 1e3
print(f"eager {t_eager:.4f} ms per iteration, graph replay {t_graph:.4f} ms per iteration")
