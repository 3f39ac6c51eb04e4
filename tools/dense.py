"""timing of the dense MFMA kernels (dev tool)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinforcementlearning.jl_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rlhip
from rlhip import ops
from rlhip.ops import stream_ptr
from bench import event_time_ms
lib, s = rlhip._lib.lib, stream_ptr()
for (B, K, N) in ((131072, 256, 256), (131072, 128, 128), (131072, 512, 512), (1 << 20, 256, 256), (4096, 256, 256)):
    xr = torch.randn((B, K), device="cuda").to(torch.bfloat16)
    wt = (torch.randn((N, K), device="cuda") / 16).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    wf = ops.dense_frag_weight_bf16(wt)
    y = torch.empty((B, N), dtype=torch.bfloat16, device="cuda")
    f1 = lambda: rlhip._lib.call("rlhip_dense_bf16_forward", ops.ptr(xr), ops.ptr(wt), ops.ptr(bias), 0, B, K, N, ops.ptr(y), 1, s)
    f2 = lambda: rlhip._lib.call("rlhip_dense_bf16_forward_tiled", ops.ptr(xr), ops.ptr(wf), ops.ptr(bias), 0, B, K, N, ops.ptr(y), 1, s)
    for f, name in ((f1, "simple"), (f2, "tiled ")):
        for _ in range(3): f()
        ms = event_time_ms(f, 20, lib, s)
        print(f"{name} B={B:8d} K={K} N={N}: {ms*1e3:8.1f} us {2.0*B*K*N/ms/1e9:8.1f} TFLOP/s  {2.0*B*(K+N)/ms/1e6:8.1f} GB/s")
