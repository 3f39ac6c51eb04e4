"""Phase timeline of the persistent PPO update (csrc/ppo_persist.hip): s_memtime stamps of thread 0 of every workgroup,
taken when RLHIP_PERSIST_DEBUG=1, read from the LAST bytes of the learner's workspace.

stamps per step: 0 step start | 1 weights + records ready (before the first tile) | 2 tile loop done | 3 fold + row
published | 4 R done (slices reduced + published) | 5 A sweep done (whole gradient here) | 6 norm + Adam done"""
import os
import sys

os.environ["RLHIP_PERSIST_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import numpy as np
import torch

import rlhip

n, T = 4096, 32
env = rlhip.HipVecEnv("cartpole", n, seed=123)
pol = rlhip.PPOPolicy(env, update_freq=T, hidden=256, seed=123)
for _ in range(5):
    pol.rollout_()
    pol.update_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pol.rollout_()
torch.cuda.synchronize()
e0.record()
pol._adv_ready = True
pol.update_()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
G, S = 256, 64
tail = 32 * n * T  # the update call's sample records sit behind the persistent kernel's area (csrc/ppo_grad.hip)
end = pol.workspace.numel() - tail
dbg = pol.workspace[end - G * S * 8 * 8:end].view(torch.int64).cpu().numpy().reshape(G, S, 8)
nsteps = pol.n_updates_per_call()
d = dbg[:, :nsteps, :7].astype(np.float64)
# s_memtime counters are per XCD (not synchronised across them): only differences inside one workgroup mean anything
spans = d[:, nsteps - 1, 6] - d[:, 0, 0]
span = float(np.median(spans))
tick_us = us / span  # the launch's event time over a workgroup's first-to-last stamp: microseconds per tick (upper bound)
print(f"update: {us:.1f} us by events; stamp span median {span:.0f} ticks (min {spans.min():.0f} max {spans.max():.0f}) -> "
      f"{tick_us * 1e3:.3f} ns per tick; status {pol.update_status()}")
names = ["weights+records", "tile loop", "fold+publish row", "R (reduce-scatter)", "A sweep (all-gather)", "norm+Adam"]
ph = np.diff(d, axis=2) * tick_us  # [G][steps][6]
print("phase                  mean    min    max   (us per step, over workgroups and steps 1..)")
for k, nm in enumerate(names):
    x = ph[:, 1:, k]
    print(f"{nm:22s} {x.mean():6.2f} {x.min():6.2f} {x.max():6.2f}")
step = (d[:, 1:, 0] - d[:, :-1, 0]) * tick_us
print(f"step period            {step.mean():6.2f} {step.min():6.2f} {step.max():6.2f}")
t7 = dbg[:, 1:nsteps, 7]
for k, nm in enumerate(["tile: phase 1a + barrier", "tile: phase 1b + barrier", "tile: phase 2 + barrier", "tile: publish first tile"]):
    x = ((t7 >> (16 * k)) & 0xFFFF).astype(np.float64) * tick_us
    print(f"{nm:26s} {x.mean():6.2f} {x.min():6.2f} {x.max():6.2f}")
own = np.arange(G) < (3331 + 15) // 16
print(f"R phase, slice owners: {ph[own, 1:, 3].mean():.2f} us; non-owners {ph[~own, 1:, 3].mean():.2f} us")
print(f"A sweep, slice owners: {ph[own, 1:, 4].mean():.2f} us; non-owners {ph[~own, 1:, 4].mean():.2f} us")
for k, nm in enumerate(names):
    print(f"{nm:22s} per step, workgroup 0:", " ".join(f"{v:5.2f}" for v in ph[0, :, k]))
