"""per-phase ticks of one steady-state tile of ppo3w_fwd_kernel<.., NET = 2> (the DQN target network, forward only) from a
-DRLHIP_W3_TIMING -DRLHIP_W3_TIMING_NET=2 build (RLHIP_LIB_PATH); thread 0 of workgroup 0 = wave 0; proportions only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch, rlhip
from rlhip import dqn
h, bm, n = 256, 131072, 4096
tr = rlhip.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)
tr.records.normal_(); tr.action.random_(0, 2); tr.reward.normal_()
tr.rb.len_sa, tr.rb.len_rt = 257, 256
net = rlhip.HipApproximator(4, h, 2, seed=5, layers=3)
tn = rlhip.TargetNetwork(net, sync_freq=100)
ws = dqn.dqn3_workspace(4, h, 2, bm)
g, l = torch.empty_like(net.params), torch.empty(1, device="cuda")
for _ in range(3):
    dqn.dqn3_grad(tr, h, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=l)
torch.cuda.synchronize()
st = (C.c_longlong * 48)()
fn = rlhip._lib.lib.rlhip_debug_w3_stamps
fn.restype = C.c_int32
assert fn(st) == 0
b = list(st)[:16]
for i, j, nme in [(1, 2, "layer 1 + barrier A"), (2, 3, "MFMA + bias / act + in-lane head FMAs"), (3, 4, "half add + partial store + barrier C"), (4, 5, "TD line (wave 0) + next loss inputs")]:
    print(f"   {nme:60s} {b[j] - b[i]:7d}")
print(f"   {'stamped span of the tile':60s} {b[5] - b[1]:7d}")
print(f"   {'prologue':60s} {b[9] - b[8]:7d}   tile loop {b[10] - b[9]:7d}")
