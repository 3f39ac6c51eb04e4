#!/usr/bin/env python3
"""The 256-wide DQN learner gradient at batch 131072 (bench.py's dqn3w_grad_mfma_hidden256 leg alone), for prof_cmd.sh / pmc.sh."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch  # noqa: E402

import rlhip  # noqa: E402
from rlhip import dqn  # noqa: E402

h = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bm = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n = 4096
tr = rlhip.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)
tr.records.normal_()
tr.action.random_(0, 2)
tr.reward.normal_()
tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.05).to(torch.uint8))
tr.rb.len_sa, tr.rb.len_rt = 257, 256
net = rlhip.HipApproximator(4, h, 2, seed=5, layers=3)
tn = rlhip.TargetNetwork(net, sync_freq=100)
ws = dqn.dqn3_workspace(4, h, 2, bm)
g, l = torch.empty_like(net.params), torch.empty(1, device="cuda")
for _ in range(3):
    dqn.dqn3_grad(tr, h, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=l)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dqn.dqn3_grad(tr, h, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0, workspace=ws, grad=g, loss=l)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 20 * 1e6
print(f"dqn3 grad hidden {h} batch {bm}: {us:.1f} us, {4 * 2 * h * h * bm / us / 1e6:.1f} TFLOP/s of MFMA work, loss {float(l):.5f}")
