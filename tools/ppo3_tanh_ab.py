"""chained tile with tanh (RLHIP_PPO3_CHAINED_TANH=1, experimental) against the round-1 tile: the instantiation that once
computed a wrong actor loss under one schedule"""
import os, sys, ctypes as C
os.environ["RLHIP_PPO3_CHAINED_TANH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, rlhip
force = rlhip._lib.lib.rlhip_debug_ppo3_force128
for kind, n, T in (("cartpole", 96, 9), ("cartpole", 1024, 16), ("pendulum", 2048, 24)):
    env = rlhip.HipVecEnv(kind, n, seed=5)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=128, seed=5, layers=3, n_microbatches=2, act=1)
    pol.rollout_(); pol.gae_()
    out = {}
    for f in (1, 0):
        force(f)
        for rep in range(2):
            pol.grad_(0, 1); torch.cuda.synchronize()
            out[(f, rep)] = (pol.grad.cpu().numpy().copy(), pol.losses.cpu().numpy().copy())
    force(0)
    g1, l1 = out[(1, 0)]; g0, l0 = out[(0, 0)]
    print(kind, n, T, "losses round-1", l1, "chained", l0, "max|dg|/max|g|", float(np.abs(g0 - g1).max() / np.abs(g1).max()),
          "chained run-to-run identical:", bool(np.array_equal(out[(0, 0)][0], out[(0, 1)][0])))
