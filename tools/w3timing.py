"""per-phase cycles of one steady-state tile of the three width-256 learner kernels (csrc/ppo3w.hip); needs a build with
RLHIP_EXTRA_FLAGS=-DRLHIP_W3_TIMING.  clock64() = s_memtime (100 MHz constant clock on gfx950: proportions only)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch  # noqa: E402

import rlhip  # noqa: E402

env = rlhip.HipVecEnv("pendulum", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=128, hidden=256, seed=7, layers=3, clip_range=0.1)
pol.rollout_()
pol.update_()
torch.cuda.synchronize()
st = (C.c_longlong * 48)()
fn = rlhip._lib.lib.rlhip_debug_w3_stamps
fn.restype = C.c_int32
assert fn(st) == 0
v = list(st)
names = [["gather (wave 0) + barrier", "layer 1 + barrier", "MFMA + bias / act", "head (DPP) + barrier", "loss (wave 0) + barrier",
          "dz + stores + barrier", "copy-out"],
         ["gather + dz tile load + barrier", "MFMA", "epilogue", "barrier"],
         ["gather + B loads + barrier", "layer 1^T + barrier", "MFMA", "barrier"]]
for k, (kn, nm) in enumerate(zip(("fwd (actor)", "bwd", "dw2"), names)):
    base = v[16 * k:16 * k + 16]
    print(kn)
    for i, n in enumerate(nm):
        print(f"   {n:34s} {base[i + 1] - base[i]:8d}")
    print(f"   {'tile total':34s} {base[len(nm)] - base[0]:8d}")
    print(f"   {'prologue':34s} {base[9] - base[8]:8d}")
    print(f"   {'tile loop':34s} {base[10] - base[9]:8d}")
    if base[11]:
        print(f"   {'epilogue':34s} {base[11] - base[10]:8d}")
