"""HIP-event timing of the sum-tree priority update / draw (dev tool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
for n_leaves in (1 << 20, 1 << 24, 5000, 1 << 12):
    P = 1
    while P < n_leaves: P *= 2
    tree = torch.zeros(2 * P, dtype=torch.float32, device="cuda")
    call("rlhip_sumtree_fill_range", ptr(tree), n_leaves, 0, n_leaves, 1.0, stream_ptr())
    for n in (32, 512, 4096, 65536):
        leaf = torch.randint(0, n_leaves, (n,), dtype=torch.int64, device="cuda")
        pr = torch.rand(n, device="cuda")
        f = lambda: call("rlhip_sumtree_update", ptr(tree), n_leaves, ptr(leaf), ptr(pr), n, stream_ptr())
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"leaves={n_leaves:9d} keys={n:6d}: update {e0.elapsed_time(e1) * 50:.1f} us")
