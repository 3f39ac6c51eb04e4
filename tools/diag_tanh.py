import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
import test_gpu_ppo3w as t
H = t.H
kind, cont, act = "pendulum", True, "tanh"
n, T = 96, 9
a = 1
env, pol = t._setup(kind, n, T, n_microbatches=2, act=a)
pol.rollout_(); pol.gae_()
ocfg = oracle.ppo_default(hidden=H, continuous=1, layers=3, n_microbatches=2, act=a)
total, bm = n * T, (n * T) // 2
tr = pol.trajectory
print("adv range", float(tr.adv.min()), float(tr.adv.max()), "value range", float(tr.value.min()), float(tr.value.max()), "logp range", float(tr.logp.min()), float(tr.logp.max()))
for mb, epoch in ((0, 0), (1, 3)):
    pol.grad_(epoch, mb)
    g = pol.grad.cpu().numpy(); og, ol = t._oracle_grad(pol, env, cont, ocfg, epoch, mb, bm, total, n, T)
    print("losses", pol.losses.cpu().numpy(), ol)
    ns = 3
    o = 0
    for tname, sz in (("W1", H * ns), ("b1", H), ("W2", H * H), ("b2", H), ("W3", 2 * H), ("b3", 2)):
        A, B = g[o:o+sz], og[o:o+sz]
        e = np.abs(A - B) / np.abs(B).max()
        print(tname, "max|o|", np.abs(B).max(), "err max", e.max(), "q99", np.quantile(e, 0.99), "argmax", int(e.argmax()))
        if tname == "W2":
            E = e.reshape(H, H)  # Flux order W2[j + H k] -> index = j + H*k -> reshape (k, j)
            print("  worst k rows:", np.argsort(-E.max(1))[:5], E.max(1)[np.argsort(-E.max(1))[:5]])
            print("  worst j cols:", np.argsort(-E.max(0))[:5], E.max(0)[np.argsort(-E.max(0))[:5]])
        o += sz
