"""which preceding work leaves the device in the state the steady-state step needs?  20-step blocks (5 untimed + 20 timed, the
driver's form) right after: 1 s idle; 60 ms of f32 matmuls; 60 ms of HBM streaming; 30 iterations of the config-3 MFMA PPO learner."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip

env = rlhip.HipVecEnv("cartpole", 4096, seed=123)
pol = rlhip.PPOPolicy(env, update_freq=32, hidden=256, seed=123)
penv = rlhip.HipVecEnv("pendulum", 4096, seed=7)
ppol = rlhip.PPOPolicy(penv, update_freq=128, hidden=256, seed=7, clip_range=0.1, layers=3)


def block(k=20, w=5):
    for _ in range(w):
        pol.rollout_(); pol.update_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        pol.rollout_(); pol.update_()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
block(); torch.cuda.synchronize()


def spin(fn, secs):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn()
    torch.cuda.synchronize()


for rep in range(2):
    time.sleep(1.0); print("after 1 s idle        ", f"{block():.4f}", flush=True)
    spin(lambda: torch.mm(a, b), 0.06); print("after 60 ms matmul    ", f"{block():.4f}", flush=True)
    time.sleep(1.0); spin(lambda: x.add_(1.0), 0.06); print("after idle + 60 ms HBM", f"{block():.4f}", flush=True)
    time.sleep(1.0)
    for _ in range(30):
        ppol.rollout_(); ppol.update_()
    torch.cuda.synchronize(); print("after idle + 30 ppo3w iterations", f"{block():.4f}", flush=True)
    time.sleep(1.0); print("after 1 s idle, w = 30", f"{block(20, 30):.4f}", flush=True)
