"""grad kernel time vs problem size (HIP events)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from bench import event_time_ms
from rlhip.ops import stream_ptr
for n, T, nmb in [(64, 32, 4), (512, 32, 4), (2048, 32, 4), (4096, 32, 4), (4096, 32, 2), (4096, 32, 1), (4096, 128, 1)]:
    env = rlhip.HipVecEnv("cartpole", n, seed=1)
    pol = rlhip.PPOPolicy(env, update_freq=T, n_microbatches=nmb)
    pol.rollout_(); pol.gae_()
    for _ in range(3): pol.grad_(0, 0)
    torch.cuda.synchronize()
    ms = event_time_ms(lambda: pol.grad_(0, 0), 50, rlhip._lib.lib, stream_ptr())
    bm = n * T // nmb
    print(f"n={n:5d} T={T:4d} mb={nmb} samples={bm:7d} tiles={bm//64:5d}  pack+grad+reduce = {ms*1e3:8.2f} us")
