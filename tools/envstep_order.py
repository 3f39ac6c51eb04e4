"""per-launch times of the 2^24-env CartPole step for (random | never-terminating) envs created in either order (dev tool)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
n = 1 << 24
KW = {"A": {}, "B": dict(xthreshold=1e9, thetathreshold=1e9, max_steps=1 << 30)}
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
for order in (sys.argv[1] if len(sys.argv) > 1 else "AB"):
    env = rlhip.HipVecEnv("cartpole", n, seed=1, **KW[order])
    t = run(env, 24)
    th = env.raw_state()[2].abs()
    print(order, t, "done frac", round(float(env._done.float().mean()), 4), "|theta|>pi/4:", round(float((th > 0.785).float().mean()), 4),
          "nonfinite:", int((~torch.isfinite(env.raw_state())).sum()))
    del env
    torch.cuda.empty_cache()
