#!/usr/bin/env python3
"""Config 3 at hidden = 256 (layers = 3): wall-clock of rollout / update; run under rocprofv3 --kernel-trace --stats for the
per-kernel table.  usage: python tools/ppo3w_time.py [n_envs] [T] [iters] [hidden]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch  # noqa: E402

import rlhip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
hidden = int(sys.argv[4]) if len(sys.argv) > 4 else 256
act = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # 0 relu, 1 tanh
env = rlhip.HipVecEnv("pendulum", n, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=T, hidden=hidden, seed=7, clip_range=0.1, layers=3, act=act)
for _ in range(2):
    pol.rollout_()
    pol.update_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    pol.rollout_()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(iters):
    pol._adv_ready = True
    pol.update_()
torch.cuda.synchronize()
t2 = time.perf_counter()
bm = n * T // pol.cfg.n_microbatches
steps = pol.cfg.n_epochs * pol.cfg.n_microbatches
us = (t2 - t1) / iters / steps * 1e6
mf = 3 * 2 * hidden * hidden * 2 * bm
print(f"hidden {hidden} act {act} n {n} T {T}: rollout {(t1 - t0) / iters * 1e3:.3f} ms, update {(t2 - t1) / iters * 1e3:.3f} ms "
      f"({us:.1f} us per optimiser step of {bm} samples, {mf / us / 1e6:.1f} TFLOP/s of MFMA work), loss {float(pol.losses[0]):.5f}")
