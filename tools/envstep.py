"""CartPole env-step kernel at 2^24 envs under the protocol of bench.py's roofline leg (a fresh random action per env
and step from 16 pre-drawn vectors; 64 de-synchronising launches, then 20 steady-state launches) -- for the rocprofv3
PMC traffic passes: the LAST 20 launches of the kernel are the ones to average."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
n = 1 << 24
env = rlhip.HipVecEnv("cartpole", n, seed=1, packed_episode=os.environ.get("RLHIP_UNPACKED", "0") != "1")
a = torch.randint(0, 2, (16, n), dtype=torch.int32, device="cuda")
ptrs = [ptr(a[k]) for k in range(16)]
for i in range(84):
    call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, ptrs[i & 15], 1, env.seed, 0, None, None, stream_ptr())
torch.cuda.synchronize()
print("terminated in the last step:", float(env._done.float().mean()))
