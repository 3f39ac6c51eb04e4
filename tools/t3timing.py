"""per-phase cycles of one steady-state tile of ppo3_gradT_kernel (needs a build with RLHIP_EXTRA_FLAGS=-DRLHIP_T3_TIMING)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
env = rlhip.HipVecEnv("pendulum", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=128, hidden=128, seed=7, layers=3, clip_range=0.1)
pol.rollout_(); pol.update_(); torch.cuda.synchronize()
st = (C.c_longlong * 16)()
fn = rlhip._lib.lib.rlhip_debug_t3_stamps
fn.restype = C.c_int32
assert fn(st) == 0
v = list(st)
names = ["H1X", "L2 mfma x64", "epi-b head+loss", "dzb", "form a", "H1Y", "slab write+barrier", "dW2 mfma x32", "dH1 mfma x32", "dH1 epilogue", "barrier", ]
for k, nm in enumerate(names):
    print(f"{nm:24s} {v[k + 1] - v[k]:8d} cycles")
print(f"{'tile total':24s} {v[11] - v[0]:8d} cycles")
print(f"{'whole tile loop':24s} {v[13] - v[12]:8d} cycles for {-(-1024 // 128)} tiles")
