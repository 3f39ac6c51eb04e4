"""A/B of the frame-gather kernels at the config-5 sizes (variant chosen by RLHIP_GS_VARIANT / RLHIP_GF_VARIANT)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import ops
from rlhip.trajectory import CircularArraySARTSTraces
sys.path.insert(0, ROOT)
from bench import event_time_ms
lib, s = rlhip._lib.lib, ops.stream_ptr()
cap, batch = 1 << 20, 4096
which = sys.argv[1]
if which in ("gae", "env"):
    pass
elif which == "frames":
    fb = 84 * 84 * 4
    tr = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=fb, dtype=torch.uint8)
    tr.state.random_(0, 256)
    tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
    idx = tr.sample_indices(batch, seed=11, draw_ctr=0)
    bufs = tr.gather(idx)
    ref = [b.clone() for b in bufs]
    c = [1]
    def smp():
        rlhip._lib.call("rlhip_ring_sample_indices", C.byref(tr.rb), batch, 11, c[0], ops.ptr(idx), s); c[0] += 1
    def g():
        rlhip._lib.call("rlhip_ring_gather", C.byref(tr.rb), ops.ptr(idx), batch, ops.ptr(bufs[0]), ops.ptr(bufs[1]), ops.ptr(bufs[2]), ops.ptr(bufs[3]), ops.ptr(bufs[4]), s)
    def both():
        smp(); g()
    gb = 2 * (2 * fb + 9) * batch / 1e9
    for rep in range(3):
        for v in (0, 1, 2, 3, 4, 5):
            lib.rlhip_debug_variant(1, v)
            for _ in range(3): both()
            ms = event_time_ms(both, 20, lib, s) - event_time_ms(smp, 20, lib, s)
            c[0] = 1; smp(); g(); torch.cuda.synchronize()
            print(f"frames variant {v}: {ms * 1e3:.1f} us  {gb / (ms * 1e-3):.0f} GB/s  checksum {int(bufs[0].sum())} {int(bufs[4].sum())}")
else:
    f1 = 84 * 84
    tr1 = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=f1, dtype=torch.uint8)
    tr1.state.random_(1, 256)
    tr1.terminal.copy_((torch.rand(cap, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) < 1 / 800).to(torch.uint8))
    tr1.rb.len_sa, tr1.rb.len_rt = cap + 1, cap
    idx1 = tr1.sample_indices(batch, seed=11, draw_ctr=0)
    outs = tr1.gather_stacked(idx1, 4)
    c = [1]
    def smp():
        rlhip._lib.call("rlhip_ring_sample_indices", C.byref(tr1.rb), batch, 11, c[0], ops.ptr(idx1), s); c[0] += 1
    def g():
        rlhip._lib.call("rlhip_ring_gather_stacked", C.byref(tr1.rb), ops.ptr(idx1), batch, 4, ops.ptr(outs[0]), ops.ptr(outs[1]), ops.ptr(outs[2]), ops.ptr(outs[3]), ops.ptr(outs[4]), s)
    def both():
        smp(); g()
    gb = (5 * f1 + 8 * f1 + 9 + 9) * batch / 1e9
    for rep in range(3):
        for v in (0, 2, 4, 5):
            lib.rlhip_debug_variant(0, v)
            for _ in range(3): both()
            ms = event_time_ms(both, 20, lib, s) - event_time_ms(smp, 20, lib, s)
            c[0] = 1; smp(); g(); torch.cuda.synchronize()
            print(f"stacked variant {v}: {ms * 1e3:.1f} us  {gb / (ms * 1e-3):.0f} GB/s  checksum {int(outs[0].sum())} {int(outs[4].sum())}")

if which == "gae":
    n, T = 1 << 20, 32
    r = torch.rand((T, n), device="cuda") * -16
    v = torch.randn((T + 1, n), device="cuda")
    term = torch.rand((T, n), device="cuda") < 1 / 200
    gb = (17 * n * T + 4 * n) / 1e9
    for rep in range(3):
        for var in (0, 1):
            lib.rlhip_debug_variant(2, var)
            a, rt = ops.gae_returns(r, v, term, 0.99, 0.95)
            ms = event_time_ms(lambda: ops.gae_returns(r, v, term, 0.99, 0.95), 20, lib, s)
            print(f"gae variant {var}: {ms * 1e3:.1f} us  {gb / (ms * 1e-3):.0f} GB/s  checksum {float(a.double().sum()):.6e} {float(rt.double().sum()):.6e}")
if which == "env":
    n_envs = 1 << 24
    env = rlhip.HipVecEnv("cartpole", n_envs, seed=1, packed_episode=True)
    actions = torch.randint(0, 2, (16, n_envs), dtype=torch.int32, device="cuda")
    a_ptrs = [ops.ptr(actions[k]) for k in range(16)]
    cnt = [0]
    def step():
        cnt[0] += 1
        rlhip._lib.call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, a_ptrs[cnt[0] & 15], 1, env.seed, 0, None, None, s)
    for _ in range(70): step()
    for rep in range(3):
        for var in (1, 2):
            lib.rlhip_debug_variant(3, var)
            for _ in range(3): step()
            ms = event_time_ms(step, 20, lib, s)
            print(f"env variant {var}: {ms * 1e3:.1f} us  {49 * n_envs / (ms * 1e-3) / 1e9:.0f} GB/s")
