"""CartPole env-step kernel at 2^24 envs, 20 launches (for rocprofv3 PMC traffic passes)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
n = 1 << 24
env = rlhip.HipVecEnv("cartpole", n, seed=1)
a = torch.randint(0, 2, (n,), dtype=torch.int32, device="cuda")
for _ in range(20):
    call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptr(a), 1, env.seed, 0, None, None, stream_ptr())
torch.cuda.synchronize()
