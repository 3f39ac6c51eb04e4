"""HIP-event timing of the CartPole env-step kernel at 2^24 / 2^26 envs (dev tool)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from bench import roofline_env_step
for n in (1 << 22, 1 << 24, 1 << 26):
    for rep in range(3):
        r = roofline_env_step(torch, rlhip, n_envs=n, iters=20)
        print(n, r["us_per_launch"], r["achieved"], r["frac"])
