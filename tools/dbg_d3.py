import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinforcementlearning.jl_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, oracle, rlhip
from rlhip import dqn
import test_gpu_dqn3 as T
H=128
for act in (0,1):
    ns, na, n_env, cap, batch = 4, 2, 64, 40, 128
    rng = np.random.default_rng(batch + act)
    traces = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    T._fill_ring(traces, oring, ns, n_env, 57, rng)
    p, tp = T._net(ns, na, 11), T._net(ns, na, 12)
    pd, tpd = torch.as_tensor(p, device="cuda"), torch.as_tensor(tp, device="cuda")
    packed, tpacked = dqn.mlp3_pack(pd, ns, H, na), dqn.mlp3_pack(tpd, ns, H, na)
    g, loss = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 7, 3)
    idx = oring.sample_indices(batch, 7, 3)
    s, a, r, t, sn = oring.gather(idx)
    rl, rg, rq = oracle.dqn3_loss_grad(ns, H, na, act, p, tp, s, a, r, t, sn, 0.99, 1.0)
    g = g.cpu().numpy(); o = 0
    print("act", act, "loss", float(loss), rl)
    for name, n in (("W1", H * ns), ("b1", H), ("W2", H * H), ("b2", H), ("W3", na * H), ("b3", na)):
        a_, b_ = g[o:o + n], rg[o:o + n]
        d = np.abs(a_ - b_)
        print(f"  {name}: maxerr {d.max():.3e} scale {np.abs(b_).max():.3e} nbad {(d > 2e-3*np.abs(b_).max()).sum()} argmax {d.argmax()}")
        o += n
    if act == 0:
        d = np.abs(g[H*ns:H*ns+H] - rg[H*ns:H*ns+H])
        print("bad b1 units:", np.nonzero(d > 1e-6)[0])
        d = np.abs(g[:H*ns] - rg[:H*ns]).reshape(ns, H)
        print("bad W1 units:", np.nonzero(d.max(0) > 1e-6)[0])
