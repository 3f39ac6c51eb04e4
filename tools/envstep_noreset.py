"""CartPole env-step at 2^24 envs with thresholds that never terminate an episode: the full physics without the
auto-reset traffic (dev tool; theta stays small only for a few steps, so the state is re-seeded between batches)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
from bench import event_time_ms
n = 1 << 24
for kw, name in ((dict(), "default (random actions: ~4.5 % of the envs reset per step)"),
                 (dict(xthreshold=1e9, thetathreshold=1e9, max_steps=1 << 30), "no episode ever terminates")):
    env = rlhip.HipVecEnv("cartpole", n, seed=1, **kw)
    actions = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
    def step():
        call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptr(actions), 1, env.seed, 0, None, None, stream_ptr())
    ts = []
    for _ in range(4):
        env.reset_()
        for _ in range(3): step()
        torch.cuda.synchronize()
        ts.append(round(event_time_ms(step, 8, rlhip._lib.lib, stream_ptr()) * 1e3, 1))
    print(name, ts, "us;", round(49 * n / ts[-1] / 1e3, 1), "GB/s")
    del env, actions
    torch.cuda.empty_cache()
