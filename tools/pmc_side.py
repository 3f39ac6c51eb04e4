"""the HBM-bound side kernels at the bench sizes, fresh inputs per launch (for rocprofv3 PMC traffic passes):
GAE + returns (2^20 envs x 32), frame gather and stack-at-sample gather from the 2^20-slot rings, batch 4096."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import ops
from rlhip.trajectory import CircularArraySARTSTraces
n, T = 1 << 20, 32
r = torch.rand((T, n), device="cuda") * -16
v = torch.randn((T + 1, n), device="cuda")
term = torch.rand((T, n), device="cuda") < 1 / 200
for i in range(10):
    ops.gae_returns(r, v, term, 0.99, 0.95)
del r, v, term
torch.cuda.empty_cache()
cap, batch = 1 << 20, 4096
tr = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=84 * 84 * 4, dtype=torch.uint8)
tr.state.random_(0, 256)
tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
for i in range(10):
    tr.gather(tr.sample_indices(batch, 11, i))
del tr
torch.cuda.empty_cache()
tr1 = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=84 * 84, dtype=torch.uint8)
tr1.state.random_(1, 256)
tr1.terminal.copy_((torch.rand(cap, 1, device="cuda") < 1 / 800).to(torch.uint8))
tr1.rb.len_sa, tr1.rb.len_rt = cap + 1, cap
for i in range(10):
    tr1.gather_stacked(tr1.sample_indices(batch, 11, i), 4)
torch.cuda.synchronize()
