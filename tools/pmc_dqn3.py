"""MFMA kernels of the 3-layer DQN path at large batch (for rocprofv3 PMC passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import dqn
ns, na, H = 4, 2, 128
tr = rlhip.CircularArraySARTSTraces(capacity=64, n_env=4096, obs_dim=ns)
tr.records.normal_(); tr.action.random_(0, 2); tr.reward.normal_()
tr.rb.len_sa, tr.rb.len_rt = 65, 64
p, tp = dqn.mlp3_init(ns, H, na, 1), dqn.mlp3_init(ns, H, na, 2)
pk, tpk = dqn.mlp3_pack(p, ns, H, na), dqn.mlp3_pack(tp, ns, H, na)
batch = 131072
ws = dqn.dqn3_workspace(ns, H, na, batch)
g, loss = torch.empty_like(p), torch.empty(1, device="cuda")
for i in range(10):
    dqn.dqn3_grad(tr, H, na, 0, p, pk, tp, tpk, batch, 0.99, 1.0, 1, i, workspace=ws, grad=g, loss=loss)
n = 1 << 20
obs = torch.randn((ns, n), device="cuda")
a, q = torch.empty(n, dtype=torch.int32, device="cuda"), torch.empty((na, n), device="cuda")
for i in range(10):
    dqn.dqn3_plan(p, pk, ns, H, na, 0, obs, 0.1, 1, 0, i, a, q)
torch.cuda.synchronize()
