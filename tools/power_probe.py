#!/usr/bin/env python3
"""Is the 256-wide learner's optimiser step running under the power cap?  Samples the GPU's hwmon power / sclk files from a thread while
the main thread runs update_() back to back for a few seconds, and prints the per-iteration update time beside them.
usage: python tools/power_probe.py [seconds] [mode: ppo3w|dqn3w|headline|envstep|adam|polyak|idle] [pad: 0|1 -- the backward kernel's LDS copy, csrc/ppo3w.hip RLHIP_W3_DZF_PAD]"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch  # noqa: E402

import rlhip  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
mode = sys.argv[2] if len(sys.argv) > 2 else "ppo3w"
if len(sys.argv) > 3:
    import ctypes as C
    from rlhip import _lib
    _lib.lib.rlhip_debug_w3_dzf_pad.argtypes = [C.c_int32]
    _lib.lib.rlhip_debug_w3_dzf_pad(int(sys.argv[3]))


def my_device_dirs():
    """sysfs device directories of the GPU this process computes on: matched by PCI address when torch reports one, else the render
    nodes this user may open (a box can show the hwmon files of GPUs that belong to other tenants)"""
    cands = sorted(glob.glob("/sys/class/drm/card*/device"))
    pr = torch.cuda.get_device_properties(0)
    bus, dev, dom = getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None), getattr(pr, "pci_domain_id", 0)
    if bus is not None and dev is not None:
        want = f"{dom:04x}:{bus:02x}:{dev:02x}."
        hit = [d for d in cands if os.path.basename(os.path.realpath(d)).startswith(want)]
        if hit:
            print("matched by PCI address", want, flush=True)
            return hit
    hit = []
    for d in cands:
        for r in glob.glob(os.path.join(d, "drm", "renderD*")):
            if os.access("/dev/dri/" + os.path.basename(r), os.R_OK | os.W_OK):
                hit.append(d)
    print("matched by openable render node:", hit, flush=True)
    return hit or cands


def sources():
    out = {}
    for d in my_device_dirs():
        for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "power1_cap", "temp1_input", "temp2_input"):
                p = os.path.join(h, name)
                if os.path.exists(p):
                    out.setdefault(name, p)
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


src = sources()
print("sources:", {k: v for k, v in src.items()}, flush=True)


def rd(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError as e:
        return f"err:{e.errno}"


for k in ("power1_cap", "pp_dpm_sclk", "pp_dpm_mclk"):
    if k in src:
        print(k, "=", rd(src[k]).replace("\n", " | "), flush=True)

samples = []
stop = False


def sampler():
    keys = [k for k in ("power1_average", "power1_input", "freq1_input", "temp1_input", "temp2_input") if k in src]
    while not stop:
        samples.append((time.perf_counter(), [rd(src[k]) for k in keys]))
        time.sleep(0.02)
    samples.append(("keys", keys))


th = threading.Thread(target=sampler)
th.start()
time.sleep(0.3)
marks = []
if mode == "ppo3w":
    env = rlhip.HipVecEnv("pendulum", 4096, seed=7)
    pol = rlhip.PPOPolicy(env, update_freq=128, hidden=256, seed=7, clip_range=0.1, layers=3, act=0)
    pol.rollout_()
    pol.update_()
    torch.cuda.synchronize()
    steps = pol.cfg.n_epochs * pol.cfg.n_microbatches
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(5):
            pol._adv_ready = True
            pol.update_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((t0, (t1 - t0) / 5 / steps * 1e6))
elif mode == "headline":  # bench.py's step: PPO / CartPole, 4096 envs x 32, two-layer f32 nets of 256
    env = rlhip.HipVecEnv("cartpole", 4096, seed=123)
    pol = rlhip.PPOPolicy(env, update_freq=32, hidden=256, seed=123)
    pol.rollout_()
    pol.update_()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(20):
            pol.rollout_()
            pol.update_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((t0, (t1 - t0) / 20 * 1e6))
elif mode == "dqn3w":  # the 256-wide DQN gradient at batch 131072
    from rlhip import dqn
    h, bm = 256, 131072
    tr = rlhip.CircularArraySARTSTraces(capacity=256, n_env=4096, obs_dim=4)
    tr.records.normal_()
    tr.action.random_(0, 2)
    tr.reward.normal_()
    tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.05).to(torch.uint8))
    tr.rb.len_sa, tr.rb.len_rt = 257, 256
    net = rlhip.HipApproximator(4, h, 2, seed=5, layers=3)
    tn = rlhip.TargetNetwork(net, sync_freq=100)
    ws = dqn.dqn3_workspace(4, h, 2, bm)
    g, l = torch.empty_like(net.params), torch.empty(1, device="cuda")
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(50):
            dqn.dqn3_grad(tr, h, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=l)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((t0, (t1 - t0) / 50 * 1e6))
elif mode == "envstep":  # the HBM-bound roofline kernel: 2^24 CartPole envs stepped back to back (bench.py's roofline_env_step)
    import ctypes as C
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr
    n_envs = 1 << 24
    env = rlhip.HipVecEnv("cartpole", n_envs, seed=1, packed_episode=True)
    actions = torch.randint(0, 2, (16, n_envs), dtype=torch.int32, device="cuda")
    a_ptrs = [ptr(actions[k]) for k in range(16)]
    k = 0
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(50):
            k += 1
            call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, a_ptrs[k & 15], 1, env.seed, 0, None, None, stream_ptr())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((t0, (t1 - t0) / 50 * 1e6))
elif mode in ("adam", "polyak"):  # the 2^26-parameter streaming launches of bench.roofline_hbm_side (28 / 12 bytes per parameter)
    from rlhip import ops
    n = 1 << 26
    p_, g_, m_ = (torch.randn(n, device="cuda") for _ in range(3))
    v_ = torch.rand(n, device="cuda") * 0.99 + 0.01
    bp = torch.tensor([0.9, 0.999], device="cuda")
    fn = (lambda: ops.adam_(p_, g_, m_, v_, bp)) if mode == "adam" else (lambda: ops.polyak_(p_, g_, 0.995))
    fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((t0, (t1 - t0) / 20 * 1e6))
elif mode == "idle":
    time.sleep(secs)
time.sleep(0.3)
stop = True
th.join()
keys = samples.pop()[1]
print("keys", keys)
t_first = samples[0][0]
# one line per 100 ms: the samples' mean beside the mean step time of the bursts that started in it
import collections
b = collections.defaultdict(list)
for t, v in samples:
    b[int((t - t_first) / 0.1)].append(v)
m = collections.defaultdict(list)
for t, us in marks:
    m[int((t - t_first) / 0.1)].append(us)
for k in sorted(b):
    cols = []
    for i, name in enumerate(keys):
        vals = [float(v[i]) for v in b[k] if not v[i].startswith("err")]
        if vals:
            scale = 1e6 if name.startswith("power") or name.startswith("freq") else 1e3
            cols.append(f"{name}={sum(vals) / len(vals) / scale:8.1f}")
    us = m.get(k)
    print(f"t={k * 0.1:5.1f}s  " + "  ".join(cols) + (f"  step_us={sum(us) / len(us):7.1f} (n={len(us)})" if us else ""))

# steady state: the samples / bursts of the second half of the busy window
if marks:
    t_a, t_b = marks[len(marks) // 2][0], marks[-1][0]
    ss = [v for t, v in samples if t_a <= t <= t_b]
    line = [f"STEADY mode={mode} lib={os.path.basename(os.environ.get('RLHIP_LIB_PATH', 'default'))} pad={sys.argv[3] if len(sys.argv) > 3 else 'default'}"]
    for i, name in enumerate(keys):
        vals = [float(v[i]) for v in ss if not v[i].startswith("err")]
        if vals:
            scale = 1e6 if name.startswith("power") or name.startswith("freq") else 1e3
            line.append(f"{name}={sum(vals) / len(vals) / scale:.1f}")
    us = [u for t, u in marks if t >= t_a]
    us.sort()
    line.append(f"step_us median={us[len(us) // 2]:.1f} min={us[0]:.1f}")
    print("  ".join(line))
