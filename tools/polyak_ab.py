"""Polyak streaming launch alone (bench.roofline_hbm_side's layout: dst, src carved from one allocation, 4352-byte stagger), for the
RLHIP_POLYAK_VARIANT A / B of round 6 (one process per variant: the hook is read once)"""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "reinforcementlearning.jl_amd")]
import torch, rlhip, bench
from rlhip import ops
from rlhip.ops import stream_ptr

lib, s = rlhip._lib.lib, stream_ptr()
res = []
for logn in (26, 22):
    n = 1 << logn
    stag = 4352
    pitch = (4 * n + stag + 255) // 256 * 256
    buf = torch.empty(2 * pitch, dtype=torch.uint8, device="cuda")
    p, g = buf[:4 * n].view(torch.float32), buf[pitch:pitch + 4 * n].view(torch.float32)
    p.normal_(); g.normal_()
    ref = 0.995 * p + (1.0 - 0.995) * g if logn == 22 else None
    ops.polyak_(p, g, 0.995)
    if ref is not None:
        assert torch.equal(p, ref * 1.0) or (p - ref).abs().max() < 1e-6
    ms = bench.event_time_ms(lambda: ops.polyak_(p, g, 0.995), 20, lib, s, 0.05)
    res.append((logn, round(ms * 1e3, 2), round(12 * n / 1e9 / (ms * 1e-3) / 8000, 4)))
    del p, g, buf
print("variant", os.environ.get("RLHIP_POLYAK_VARIANT", "1"), res)
