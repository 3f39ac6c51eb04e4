"""per-phase cycles of one steady-state round of ppo3_gradP_kernel (needs a build with RLHIP_EXTRA_FLAGS=-DRLHIP_P3P_TIMING)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
rlhip._lib.lib.rlhip_debug_ppo3_variant(1)
env = rlhip.HipVecEnv("pendulum", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=128, hidden=128, seed=7, layers=3, clip_range=0.1)
pol.rollout_(); pol.update_(); torch.cuda.synchronize()
st = (C.c_longlong * 48)()
fn = rlhip._lib.lib.rlhip_debug_p3p_stamps
fn.restype = C.c_int32
assert fn(st) == 0
v = list(st)
names = ["(1) H1X + H2b mfma", "H1Y transposition", "(2) head, loss, dzb", "(3) H2a + form a", "ring loads", "barrier A", "(4) dH1a mfma + ring", "dH1 epilogue", "barrier B"]
for wi, wn in enumerate(("wave 0 (SIMD 0, with wave 4)", "wave 2 (SIMD 2, with consumer 6)")):
    p = v[16 * wi:16 * wi + 16]
    print(wn)
    for k, nm in enumerate(names):
        print(f"   {nm:28s} {p[k + 1] - p[k]:8d}")
    print(f"   {'round':28s} {p[9] - p[0]:8d}")
c = v[32:48]
print("consumer wave 6: wait A", c[1] - c[0], " mfma", c[2] - c[1], " wait B", c[3] - c[2])
