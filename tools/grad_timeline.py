"""Phase timeline of the two-launch PPO gradient kernel (csrc/ppo_grad.hip): s_memtime stamps of thread 0 of every workgroup
(RLHIP_GRAD_DEBUG=1), read from the debug words of the learner's workspace (layout: prepare_grad).

stamps: 0 prologue done (before publish_first_tile) | 1 tile loop entered | 2 phase 1a done | 3 phase 1b done | 4 phase 2 done |
5 (= 4) | 6 fold done.  What is NOT between stamps -- launch, prologue loads, the partial-row stores and the drain -- is the
launch's event time minus the stamped span."""
import os
import sys

os.environ["RLHIP_GRAD_DEBUG"] = "1"
os.environ["RLHIP_PPO_PERSIST"] = "0"
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
nsteps = pol.n_updates_per_call()
us_step = e0.elapsed_time(e1) * 1e3 / nsteps
MAXB = 512
np_ = int(pol.params.numel())
off = MAXB * np_ * 4 + MAXB * 4 * 4
off = (off + 15) & ~15
off += 4096 * 8 + 16 * 4
G = 256
d8 = pol.workspace[off:off + MAXB * 8 * 8].view(torch.int64).cpu().numpy().reshape(MAXB, 8)[:G].astype(np.float64)
d = d8[:, :7]
pro = (d8[:, 0] - d8[:, 7]) * float(os.environ.get("RLHIP_TICK_NS", "0.47")) * 1e-3
print(f"gradient launch, thread 0: kernel entry -> records staged (keys, gather issued, records) {pro.mean():6.2f} {pro.min():6.2f} {pro.max():6.2f}")
tick_us = 1e-3 * float(os.environ.get("RLHIP_TICK_NS", "0.47"))  # s_memtime tick (tools/persist_timeline.py measures it)
names = ["publish first tile (+ barrier)", "phase 1a (+ barrier)", "phase 1b (+ barrier)", "phase 2 (+ barrier)", "-", "fold"]
ph = np.diff(d, axis=1) * tick_us
print(f"update: {us_step:.2f} us per optimiser step by events (gradient + reduce/apply launches)")
for k, nm in enumerate(names):
    if nm != "-":
        print(f"{nm:32s} {ph[:, k].mean():6.2f} {ph[:, k].min():6.2f} {ph[:, k].max():6.2f}")
print(f"{'stamped span':32s} {(d[:, 6] - d[:, 0]).mean() * tick_us:6.2f}")

# reduce + norm barrier + Adam launch (reduce_apply_kernel<APPLY_GRID>): stamps of lane 0 of wave 0 of its 53 workgroups
R = (np_ + 63) // 64
e = pol.workspace[off + 256 * 8 * 8:off + (256 + R) * 8 * 8].view(torch.int64).cpu().numpy().reshape(R, 8)[:, :6].astype(np.float64)
pe = np.diff(e, axis=1) * tick_us
print("reduce_apply_kernel:")
for k, nm in enumerate(["partial-row loads + adds", "LDS hand-over + barrier", "group sums, grad store, sumsq", "grid barrier on the norm",
                        "norm + Adam + stores"]):
    print(f"{nm:32s} {pe[:, k].mean():6.2f} {pe[:, k].min():6.2f} {pe[:, k].max():6.2f}")
print(f"{'stamped span':32s} {(e[:, 5] - e[:, 0]).mean() * tick_us:6.2f}")
