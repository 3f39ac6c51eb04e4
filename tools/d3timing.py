"""phase timestamps of one dqn3_grad tile (build with RLHIP_EXTRA_FLAGS=-DRLHIP_D3_TIMING)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinforcementlearning.jl_amd"))
import torch, rlhip
from rlhip import dqn
ns, na, H = 4, 2, 128
tr = rlhip.CircularArraySARTSTraces(capacity=64, n_env=4096, obs_dim=ns)
tr.records.normal_(); tr.action.random_(0, 2); tr.reward.normal_()
tr.rb.len_sa, tr.rb.len_rt = 65, 64
p = dqn.mlp3_init(ns, H, na, 1); tp = dqn.mlp3_init(ns, H, na, 2)
pk, tpk = dqn.mlp3_pack(p, ns, H, na), dqn.mlp3_pack(tp, ns, H, na)
for batch in (128, 4096):
    for it in range(3):
        dqn.dqn3_grad(tr, H, na, 0, p, pk, tp, tpk, batch, 0.99, 1.0, 1, it)
        torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    rlhip._lib.lib.rlhip_debug_d3_stamps(out)
    st = list(out)
    names = {0: "start", 1: "gathered", 2: "t:layer1", 3: "t:gemm1", 4: "t:head", 5: "o:layer1(2x)", 6: "o:gemm1", 7: "o:head",
             8: "loss", 9: "headbwd+reduce", 12: "gemm2", 10: "dz1+dW1", 11: "gemm3+store"}
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10, 11]
    print("batch", batch)
    for a, b in zip(order[:-1], order[1:]):
        print(f"  {names[b]:16s} {(st[b]-st[a])*0.01:8.2f} us")
    print(f"  total            {(st[11]-st[0])*0.01:8.2f} us")
