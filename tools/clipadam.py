"""timing of the fused clip+Adam kernel over parameter-vector sizes (dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip import ops
from rlhip.ops import stream_ptr
from bench import event_time_ms
lib, s = rlhip._lib.lib, stream_ptr()
for n in (770, 3331, 4097, 17410, 30001, 65536):
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    bp = torch.tensor([0.9, 0.999], device="cuda"); gn = torch.zeros(1, device="cuda")
    f = lambda: rlhip._lib.call("rlhip_clip_adam_f32", ops.ptr(p), ops.ptr(g), ops.ptr(m), ops.ptr(v), ops.ptr(bp), n, 1.0, 0.5, 1e-3, 0.9, 0.999, 1e-8, ops.ptr(gn), s)
    for _ in range(3): f()
    # enqueue 200 back-to-back so the device, not the host, paces
    ms = event_time_ms(f, 200, lib, s)
    print(f"n={n:6d}: {ms*1e3:7.2f} us per launch")
