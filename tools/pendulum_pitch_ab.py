"""Pendulum env-step at ~2^24 envs in a fresh process: (a) does the pitch of the (3, n) observation planes matter (n = 2^24 puts the
three planes exactly 64 MB apart, n = 2^24 + 4352 staggers them like the env's own arrays)?  (b) how many launches does the
first measurement of a process need before it reads the steady value (python tools/pendulum_pitch_ab.py [warmup launches])?"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip.ops import call, ptr, stream_ptr
sys.path.insert(0, ROOT)
from bench import event_time_ms

lib, s = rlhip._lib.lib, stream_ptr()
for rep in range(2):
    for n in ((1 << 24), (1 << 24) + 4352):
        env = rlhip.HipVecEnv("pendulum", n, seed=1, packed_episode=True)
        actions = torch.randint(0, 3, (8, n), dtype=torch.int32, device="cuda")
        a_ptrs = [ptr(actions[k]) for k in range(8)]
        buf = torch.empty(3 * n * 4 + 65536, dtype=torch.uint8, device="cuda")
        obs = buf[26368:26368 + 3 * n * 4].view(torch.float32).view(3, n)
        k = [0]
        def step():
            k[0] += 1
            call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, a_ptrs[k[0] & 7], 1, env.seed, 0, None, ptr(obs), s)
        for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
            step()
        torch.cuda.synchronize()
        ms = event_time_ms(step, 20, lib, s)
        print(f"n = {n}: {ms * 1e3:.1f} us per launch = {45 * n / (ms * 1e-3) / 8e12:.3f} of 8 TB/s", flush=True)
        del env, actions, obs, buf
        torch.cuda.empty_cache()
