"""Phase timeline of ppo_grad_kernel (debug stamps; run with RLHIP_GRAD_DEBUG=1 on the GPU box)."""
import os, sys
os.environ["RLHIP_GRAD_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import ctypes as C
import numpy as np, torch, rlhip
from rlhip import _lib
env = rlhip.HipVecEnv("cartpole", 4096, seed=1)
pol = rlhip.PPOPolicy(env, update_freq=32)
pol.rollout_(); pol.gae_()
for _ in range(3): pol.grad_(0, 0)
torch.cuda.synchronize()
np_ = pol.np
off = 512 * (np_ + 4) * 4
off = (off + 15) & ~15
off += 4096 * 8 + 16 * 4  # sumsq + counter pad
ws = pol.workspace
dbg = ws[off: off + 512 * 8 * 8].view(torch.int64).cpu().numpy().reshape(512, 8)
t0 = dbg[:, 0].min()
rel = dbg[:, :6] - t0
names = ["start", "prologue+gather", "phase1a", "phase1b", "phase2", "end"]
print("clock64 ticks relative to the first block start (median over 512 blocks / min / max):")
for k, nm in enumerate(names):
    print(f"  {nm:18s} median {np.median(rel[:,k]):9.0f}  min {rel[:,k].min():9.0f}  max {rel[:,k].max():9.0f}")
d = np.diff(dbg[:, :6], axis=1)
print("per-phase durations (median ticks):", dict(zip(names[1:], np.median(d, axis=0).astype(int))))
print("kernel span ticks:", dbg[:, 5].max() - t0)
