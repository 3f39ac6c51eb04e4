"""run-to-run spread of the env-step roofline measurement: consecutive batches of 20 launches (dev tool)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
from bench import event_time_ms
n = 1 << 24
env = rlhip.HipVecEnv("cartpole", n, seed=1)
actions = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
def step():
    call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptr(actions), 1, env.seed, 0, None, None, stream_ptr())
for _ in range(3): step()
torch.cuda.synchronize()
print([round(event_time_ms(step, 20, rlhip._lib.lib, stream_ptr()) * 1e3, 1) for _ in range(12)])
