"""dump losses / grad of the test configuration (run once per kernel variant, compare offline)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, rlhip
tag = sys.argv[1]
out = {}
for kind, act in (("cartpole", 1), ("cartpole", 0), ("pendulum", 1)):
    env = rlhip.HipVecEnv(kind, 96, seed=5)
    pol = rlhip.PPOPolicy(env, update_freq=9, hidden=128, seed=5, layers=3, n_microbatches=2, act=act)
    pol.rollout_(); pol.gae_()
    for rep in range(3):
        for mb, epoch in ((0, 0), (1, 3)):
            pol.grad_(epoch, mb)
            torch.cuda.synchronize()
            out[f"{kind}{act}_l_{mb}_{rep}"] = pol.losses.cpu().numpy().copy()
            out[f"{kind}{act}_g_{mb}_{rep}"] = pol.grad.cpu().numpy().copy()
np.savez(os.path.join(ROOT, "gpurun_out", f"ppo3cmp_{tag}.npz"), **out)
