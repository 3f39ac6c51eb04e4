"""the persistent Dense kernel (256 -> 256, batch 131072) and the vec4 GAE scan (2^20 envs x 32), 10 launches each, for PMC passes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip import ops
from rlhip.ops import stream_ptr
B, K, N = 131072, 256, 256
xr = torch.randn((B, K), device="cuda").to(torch.bfloat16)
wt = (torch.randn((N, K), device="cuda") / 16).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
wf = ops.dense_frag_weight_bf16(wt)
y = torch.empty((B, N), dtype=torch.bfloat16, device="cuda")
for _ in range(10):
    rlhip._lib.call("rlhip_dense_bf16_forward_tiled", ops.ptr(xr), ops.ptr(wf), ops.ptr(bias), 0, B, K, N, ops.ptr(y), 1, stream_ptr())
n, T = 1 << 20, 32
r = torch.rand((T, n), device="cuda") * -16
v = torch.randn((T + 1, n), device="cuda")
term = torch.rand((T, n), device="cuda") < 1 / 200
for _ in range(10):
    ops.gae_returns(r, v, term, 0.99, 0.95)
torch.cuda.synchronize()
