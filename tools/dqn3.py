"""timing of the 3-layer MFMA DQN kernels (dev tool)"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import dqn
from rlhip.ops import stream_ptr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import event_time_ms

ns, na, H = 4, 2, 128
n_env, cap = 4096, 64
tr = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
tr.records.normal_()
tr.action.random_(0, 2)
tr.reward.normal_()
tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
p = dqn.mlp3_init(ns, H, na, 1)
tp = dqn.mlp3_init(ns, H, na, 2)
pk, tpk = dqn.mlp3_pack(p, ns, H, na), dqn.mlp3_pack(tp, ns, H, na)
lib, s = rlhip._lib.lib, stream_ptr()
for batch in (32, 512, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    ws = dqn.dqn3_workspace(ns, H, na, batch)
    g = torch.empty_like(p); loss = torch.empty(1, device="cuda")
    f = lambda: dqn.dqn3_grad(tr, H, na, 0, p, pk, tp, tpk, batch, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=loss)
    f(); torch.cuda.synchronize()
    ms = event_time_ms(f, 20, lib, s)
    fl = 4 * 2 * H * H * batch
    print(f"dqn3_grad batch {batch:7d}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.2f} TFLOP/s (MFMA flops only)")
p2 = rlhip.ops.mlp2_init(ns, H, na, 1, 0)
for batch in (512, 4096, 32768):
    ws = dqn.dqn_workspace(ns, H, na, batch)
    g = torch.empty_like(p2); loss = torch.empty(1, device="cuda")
    f = lambda: dqn.dqn_grad(tr, H, na, 0, p2, p2, batch, 0.99, 1.0, 1, 0, ws, g, loss)
    f(); torch.cuda.synchronize()
    ms = event_time_ms(f, 20, lib, s)
    print(f"dqn2_grad (4->128->2, VALU) batch {batch:7d}: {ms*1e3:8.1f} us")
for n in (4096, 1 << 17, 1 << 20):
    obs = torch.randn((ns, n), device="cuda")
    a = torch.empty(n, dtype=torch.int32, device="cuda"); q = torch.empty((na, n), device="cuda")
    f = lambda: dqn.dqn3_plan(p, pk, ns, H, na, 0, obs, 0.1, 1, 0, 3, a, q)
    f(); torch.cuda.synchronize()
    ms = event_time_ms(f, 20, lib, s)
    print(f"dqn3_plan n {n:8d}: {ms*1e3:8.1f} us  {2*H*H*n/ms/1e9:8.2f} TFLOP/s")
