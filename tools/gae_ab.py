import sys; sys.path.insert(0,"reinforcementlearning.jl_amd"); sys.path.insert(0,".")
import torch, rlhip
from rlhip import ops
from bench import event_time_ms
from rlhip.ops import stream_ptr
T=32; n=1<<20
r=torch.rand((T,n),device="cuda")*-16; v=torch.randn((T+1,n),device="cuda"); term=torch.rand((T,n),device="cuda")<1/200
ops.gae_returns(r,v,term,0.99,0.95)
ts=[event_time_ms(lambda: ops.gae_returns(r,v,term,0.99,0.95),10,rlhip._lib.lib,stream_ptr()) for _ in range(5)]
print(sys.argv[1], [round(t*1e3,1) for t in ts])
