"""rlhip_sumtree_update alone: us per update of 32 / 512 / 4096 / 65536 sampled keys on a 2^20-leaf tree (round 4 tuning)"""
import os, sys, ctypes as C
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"reinforcementlearning.jl_amd")]
import torch, rlhip
from rlhip import ops
from bench import event_time_ms
cap=1<<20
tr = rlhip.CircularPrioritizedTraces(capacity=cap, n_env=1, obs_dim=16, dtype=torch.uint8)
tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
keys = torch.arange(cap, dtype=torch.int64, device="cuda")
tr.set_priority_(keys, ops.fill_uniform(cap, 11, 0, 7) ** 0.6)
s=ops.stream_ptr(); lib=rlhip._lib.lib
res={}
for b in (8,32,64,512,4096,65536):
    idx,key,prio=tr.sample_prioritized(b,11,0)
    f=lambda: rlhip._lib.call("rlhip_sumtree_update", ops.ptr(tr.priorities), cap, ops.ptr(key), ops.ptr(prio), b, s)
    f(); res[b]=round(event_time_ms(f,20,lib,s)*1e3,2)
print(res)
