"""timing of the 3-layer MFMA PPO path (config 3: 4096 Pendulum envs, T = 128) (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip.ops import stream_ptr
from bench import event_time_ms
lib, s = rlhip._lib.lib, stream_ptr()
for kind, T in (("pendulum", 128), ("cartpole", 32)):
    env = rlhip.HipVecEnv(kind, 4096, seed=7)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=128, seed=7, layers=3, clip_range=0.1)
    for _ in range(2):
        pol.rollout_(); pol.update_()
    torch.cuda.synchronize()
    r = event_time_ms(pol.rollout_, 5, lib, s)
    g = event_time_ms(pol.gae_, 5, lib, s)
    u = event_time_ms(pol.update_, 3, lib, s)
    print(f"{kind} T={T} layers=3: rollout {r*1e3:8.1f} us  gae {g*1e3:6.1f} us  update(incl gae) {u*1e3:8.1f} us  -> {4096*T/((r+u)*1e-3):.3e} env-steps/s, np={pol.np}")
    bm = 4096 * T // 4
    fl = (4 * 2 * 128 * 128) * 2 * bm  # per micro-batch: (fwd + dH1 + dW2) x 2 nets ~ 3 GEMMs x 2 nets; count 3
    print(f"   per micro-batch {(u-g)/16*1e3:7.1f} us; MFMA flops/microbatch {3*2*128*128*2*bm/1e9:.2f} GF -> {3*2*128*128*2*bm/((u-g)/16*1e-3)/1e12:.1f} TFLOP/s")
