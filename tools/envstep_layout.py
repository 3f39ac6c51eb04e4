"""Does the placement of the 13 streamed arrays matter?  CartPole env-step at 2^24 envs with (a) the allocator's default
placement (every array a multiple of 64 MB apart), (b) all arrays carved from ONE buffer at staggered offsets, plus the
box's plain copy bandwidth for reference.  HIP events around 20 steady-state launches each."""
import os, sys, ctypes as C, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import _lib
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr

n = 1 << 24
BYTES = 49 * n


def timed(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def run(stagger):
    env = rlhip.HipVecEnv("cartpole", n, seed=1)
    keep = None
    if stagger is not None:
        # one buffer; array k starts at k * (4 n + stagger) bytes (16-byte aligned)
        per = 4 * n + stagger
        buf = torch.zeros(9 * per + 256, dtype=torch.uint8, device="cuda")
        base = buf.data_ptr()
        base += (-base) % 256
        st = env._st
        host = {"s": [env._s[k].clone() for k in range(4)], "t": env._t.clone(), "ep": env._episode.clone()}
        ptrs = [base + k * per for k in range(9)]
        def view(p, dtype, count):
            off = p - buf.data_ptr()
            return buf[off:off + count * torch.tensor([], dtype=dtype).element_size()].view(dtype)
        vs = [view(ptrs[k], torch.float32, n) for k in range(4)]
        for k in range(4):
            vs[k].copy_(host["s"][k]); st.s[k] = ptrs[k]
        vt = view(ptrs[4], torch.int32, n); vt.copy_(host["t"]); st.t = ptrs[4]
        vd = view(ptrs[5], torch.uint8, n); st.done = ptrs[5]
        vr = view(ptrs[6], torch.float32, n); st.reward = ptrs[6]
        ve = view(ptrs[7], torch.int32, n); ve.copy_(host["ep"]); st.episode = ptrs[7]
        keep = (buf, vs, vt, vd, vr, ve)
        done_t = vd
    else:
        done_t = env._done
    a = torch.randint(0, 2, (16, n), dtype=torch.int32, device="cuda")
    ap = [ptr(a[k]) for k in range(16)]
    c = [0]
    def step():
        c[0] += 1
        call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), n, ap[c[0] & 15], 1, env.seed, 0, None, None, stream_ptr())
    for _ in range(64):
        step()
    us = [timed(step) for _ in range(3)]
    frac = float(done_t.float().mean())
    del env, a, keep
    torch.cuda.empty_cache()
    return us, frac

out = {}
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
us = min(timed(lambda: y.copy_(x), 10) for _ in range(3))
out["copy_1GiB_TBps"] = round(2 * x.numel() * 4 / us / 1e6, 3)
del x, y
torch.cuda.empty_cache()
for stagger in (None, 0, 4096 + 256, 65536 + 1024, (1 << 20) + 8192, 3 * 4096 + 512):
    us, frac = run(stagger)
    out[str(stagger)] = {"us": [round(u, 1) for u in us], "TBps": round(BYTES / min(us) / 1e6, 3), "done_frac": round(frac, 4)}
    print(stagger, out[str(stagger)], flush=True)
print(json.dumps(out))
