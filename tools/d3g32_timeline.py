"""per-phase wall-clock stamps of the last tile of workgroup 0 of dqn3_grad32_kernel (thread 0; steady state at batch 131072 = 8 tiles per workgroup, the only tile at 4096), from a
library built with -DRLHIP_D3_TIMING (gpurun_ab/libT.so: RLHIP_LIB_PATH):  python tools/d3g32_timeline.py [batch]
Proportions only: s_memrealtime ticks at 100 MHz (10 ns), and the stamps pin the schedule."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch, rlhip
from rlhip import dqn
h, bm = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 131072
n = 4096
tr = rlhip.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)
tr.records.normal_(); tr.action.random_(0, 2); tr.reward.normal_()
tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.05).to(torch.uint8))
tr.rb.len_sa, tr.rb.len_rt = 257, 256
net = rlhip.HipApproximator(4, h, 2, seed=5, layers=3)
tn = rlhip.TargetNetwork(net, sync_freq=100)
ws = dqn.dqn3_workspace(4, h, 2, bm)
g, l = torch.empty_like(net.params), torch.empty(1, device="cuda")
for _ in range(5):
    dqn.dqn3_grad(tr, h, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=l)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
f = rlhip._lib.lib.rlhip_debug_d3_stamps
f.restype, f.argtypes = C.c_int32, [C.POINTER(C.c_longlong)]
assert f(out) == 0
st = [out[12 + k] for k in range(20)]
names = ["tile start -> transitions published", "barrier 1", "layer 1 (target)", "barrier 2", "layer 2 (target): 8 dependent MFMAs + H2 tile",
         "barrier 3 (+ the online fragments requested)", "head (target)", "layer 1 (online, both layouts)", "barrier 4", "layer 2 (online)",
         "barrier 5 (+ the W2kj fragments requested)", "head (online)", "barrier 6", "TD / Huber line (wave 0)", "barrier 7", "head backward -> dZ2 tiles",
         "barrier 8", "dH1 (8 dependent MFMAs) + dW1 / db1", "dW2 (8 MFMAs)"]
tot = (st[19] - st[0]) * 10.0
print(f"batch {bm}: one tile of workgroup 0 = {tot:.0f} ns")
for k, nm in enumerate(names):
    d = (st[k + 1] - st[k]) * 10.0
    print(f"  {nm:58s} {d:7.0f} ns  {100 * d / tot:5.1f} %")
