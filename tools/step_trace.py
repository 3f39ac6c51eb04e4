"""per-step wall time of the headline step from a cold device: how long does the ramp to the steady-state rate take?
python tools/step_trace.py   (prints ms per step, averaged over windows, for: cold start, after 1 s idle, after 1 s of HBM-bound work)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip

env = rlhip.HipVecEnv("cartpole", 4096, seed=123)
pol = rlhip.PPOPolicy(env, update_freq=32, hidden=256, seed=123)


def trace(tag, n=320, win=(5, 20, 20, 20, 40, 80, 135)):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        pol.rollout_(); pol.update_()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    out, a = [], 0
    for w in win:
        out.append(f"[{a}:{a + w}) {sum(ms[a:a + w]) / w:.4f}")
        a += w
    print(tag, " ".join(out), f"| wall/step {wall / n * 1e3:.4f}", flush=True)


def block(k):  # the bench's protocol: W untimed, sync, K timed, sync
    for _ in range(5):
        pol.rollout_(); pol.update_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        pol.rollout_(); pol.update_()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


pol.rollout_(); pol.update_(); torch.cuda.synchronize()
trace("cold      ")
time.sleep(1.0)
trace("idle 1 s  ")
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    x.add_(1.0)
torch.cuda.synchronize()
trace("after HBM ")
time.sleep(1.0)
print("20-step blocks after 1 s idle:", " ".join(f"{block(20):.4f}" for _ in range(6)), flush=True)
time.sleep(1.0)
print("200-step block after 1 s idle:", f"{block(200):.4f}")
