"""HIP-event timing of the Acrobot env-step kernel (one classic RK4 step in Float64 per env-step; dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
for T, name in ((torch.float32, "f32"), (torch.float64, "f64")):
    for n in (1 << 20, 1 << 22):
        env = rlhip.AcrobotEnv(n, T=T, seed=1)
        a = torch.randint(0, 3, (n,), dtype=torch.int32, device="cuda")
        for _ in range(3):
            env.act0_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            env.act0_(a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        eb = 4 if T == torch.float32 else 8
        byts = n * (4 * eb * 2 + 4 + 4 + 4 + eb + 1 + 2 * 6 * eb)   # state rw, action, t rw, reward, done, obs + last_obs
        print(f"acrobot {name} n={n}: {us:.1f} us/launch, {n / us:.1f} env-steps/us, {byts / us / 1e3:.1f} GB/s algorithmic")
