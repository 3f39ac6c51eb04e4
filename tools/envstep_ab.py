"""A/B of the CartPole env-step at 2^24 envs in ONE process, alternating: A = default thresholds (random actions, ~4.5 %
of the envs terminate and auto-reset per step), B = thresholds that never terminate (state re-seeded before every
batch so that theta stays on the small-angle path).  Separates the cost of the auto-reset from box / clock noise."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
n = 1 << 24
envs = {"A": rlhip.HipVecEnv("cartpole", n, seed=1),
        "B": rlhip.HipVecEnv("cartpole", n, seed=1, xthreshold=1e9, thetathreshold=1e9, max_steps=1 << 30)}
actions = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
def run(env, k):
    ts = []
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptr(actions), 1, env.seed, 0, None, None, stream_ptr())
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return [round(a.elapsed_time(b) * 1e3, 1) for a, b in ts]
for rep in range(4):
    for name in ("A", "B"):
        env = envs[name]
        env.reset_()
        t = run(env, 10)
        print(name, t, "median", sorted(t)[len(t) // 2])
