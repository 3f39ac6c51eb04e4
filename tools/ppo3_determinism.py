"""run-to-run bit-determinism of the three PPO layers = 3 gradient kernels (fixed summation order -> any difference is a hazard)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, rlhip
pick = rlhip._lib.lib.rlhip_debug_ppo3_variant
force = rlhip._lib.lib.rlhip_debug_ppo3_force128
for kind, n, T in (("cartpole", 1024, 16), ("pendulum", 4096, 128), ("cartpole", 4096, 128)):
    env = rlhip.HipVecEnv(kind, n, seed=1)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=128, seed=1, layers=3, n_microbatches=4, act=0)
    pol.rollout_(); pol.gae_()
    for name, f, v in (("round1", 1, 0), ("chained", 0, 0), ("prodcons", 0, 1)):
        force(f); pick(v)
        runs = []
        for rep in range(6):
            pol.grad_(rep % 2, rep % 4 if False else 1); torch.cuda.synchronize()
            runs.append(pol.grad.cpu().numpy().copy())
        ref = [runs[0], runs[1]]
        nd = [int((runs[k] != ref[k % 2]).sum()) for k in range(6)]
        mx = max(float(np.abs(runs[k] - ref[k % 2]).max()) for k in range(6))
        print(kind, n, T, name, "differing entries per run:", nd, "max abs diff", mx)
    force(0); pick(-1)
