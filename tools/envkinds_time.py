"""achieved HBM rate of the env-step kernel for the three classic-control envs at 2^24 envs (dev tool): the lighter
the physics, the closer to the streaming limit -- separates instruction issue from bandwidth"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
from bench import event_time_ms
n = 1 << 24
for kind, cont, byts in (("mountaincar", False, 33), ("pendulum", False, 2 * 4 * 2 + 4 + 4 + 4 + 4 + 1), ("cartpole", False, 49)):
    env = rlhip.HipVecEnv(kind, n, seed=1, continuous=cont)
    na = 2 if kind == "cartpole" else 3
    actions = torch.randint(0, na, (n,), dtype=torch.int32, device="cuda")
    def step():
        call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptr(actions), 1, env.seed, 0, None, None, stream_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = [event_time_ms(step, 20, rlhip._lib.lib, stream_ptr()) * 1e3 for _ in range(4)]
    print(kind, [round(t, 1) for t in ts], "us;", round(byts * n / ts[-1] / 1e3, 1), "GB/s at", byts, "B/env-step")
    del env, actions
    torch.cuda.empty_cache()
