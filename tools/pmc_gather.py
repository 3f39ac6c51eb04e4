"""config 5 gather kernels on the full 2^20-slot ring, fresh indices per launch (for rocprofv3 PMC traffic passes)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip.trajectory import CircularArraySARTSTraces
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
tr1.rb.len_sa, tr1.rb.len_rt = cap + 1, cap
for i in range(10):
    tr1.gather_stacked(tr1.sample_indices(batch, 11, i), 4)
torch.cuda.synchronize()
