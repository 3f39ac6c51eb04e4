"""GPU parity of the 3-layer MFMA Q-network path at hidden = 256 (the DQN half of csrc/ppo3w.hip, reached through the
unchanged rlhip_mlp3_* / rlhip_dqn3_* entry points) against the oracle (oracle/rlo_mlp3.c) -- tests/test_gpu_dqn3.py at
the other width.

Tolerances (stated once): the oracle applies the same bf16 roundings as the kernels, so what is left is the MFMA's
internal f32 summation order (the oracle accumulates in Float64 and rounds once) and the occasional bf16 rounding
flip of a dz2 element that differs in its last f32 bit: forward |dq| <= 2e-5 * (1 + |q|); gradients within
2e-3 * max|g| per tensor.  tanh: ocml tanhf and glibc tanhf differ by an ulp on some inputs; when that ulp straddles a
bf16 rounding boundary of h1 (probability ~ 2^-16 per element) one hidden unit moves by 2^-9 relative, so for tanh
99.9 % of the outputs must meet the bound above and all of them 5e-3 * (1 + |q|); integer outputs (greedy / eps-greedy actions) are compared where the top-2 Q gap
exceeds the forward tolerance.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import BF16_GRAD_TOL, assert_grad_close  # noqa: E402

pytestmark = pytest.mark.gpu

H = 256
WV = H // 32  # column tiles of the fragment layout


def _assert_q_close(q, ref, act):
    err = np.abs(q - ref) / (1 + np.abs(ref))
    if act == 0:
        assert err.max() <= 2e-5, err.max()
    else:
        assert (err <= 2e-5).mean() >= 0.999 and err.max() <= 5e-3, ((err <= 2e-5).mean(), err.max())


def _net(ns, na, seed, bias=True):
    p = oracle.mlp3_init(ns, H, na, seed, 0)
    if bias:  # non-zero biases so every bias path is exercised
        rng = np.random.default_rng(seed)
        o = 0
        for n, isb in ((H * ns, 0), (H, 1), (H * H, 0), (H, 1), (na * H, 0), (na, 1)):
            if isb:
                p[o:o + n] = rng.standard_normal(n).astype(np.float32) * 0.1
            o += n
    return p


def test_init_and_pack_bit_exact():
    from rlhip import dqn

    for ns, na in ((4, 2), (2, 3), (3, 3)):
        p = dqn.mlp3_init(ns, H, na, 9, 1)
        ref = oracle.mlp3_init(ns, H, na, 9, 1)
        assert np.array_equal(p.cpu().numpy(), ref)
        packed = dqn.mlp3_pack(p, ns, H, na).cpu().numpy().view(np.uint16)
        W2 = ref[H * ns + H:H * ns + H + H * H]  # Flux column-major: W2[j + H k]
        bf = (oracle.bf16_round(W2).view(np.uint32) >> 16).astype(np.uint16).reshape(H, H).T  # bf[j][k]
        # MFMA B-fragment order: fragment (ks, t), lane l, element u (ppo3w.hip mlp3w_pack_kernel: 8 column tiles)
        q = np.arange(H * H)
        u, l, f = q & 7, (q >> 3) & 63, q >> 9
        t, ks = f % WV, f // WV
        col, kk = 32 * t + (l & 31), 16 * ks + 8 * (l >> 5) + u
        assert np.array_equal(packed[:H * H], bf[col, kk])   # "W2jk": B(col = j, kk = k)
        assert np.array_equal(packed[H * H:], bf[kk, col])   # "W2kj": B(col = k, kk = j)


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("ns,na,n", [(4, 2, 4096), (4, 2, 1), (2, 3, 130), (3, 3, 257), (4, 2, 70001)])
def test_forward_and_plan(ns, na, n, act):
    from rlhip import dqn

    p = _net(ns, na, 3 + ns)
    rng = np.random.default_rng(n)
    x = rng.standard_normal((ns, n)).astype(np.float32)
    pd = torch.as_tensor(p, device="cuda")
    packed = dqn.mlp3_pack(pd, ns, H, na)
    xd = torch.as_tensor(x, device="cuda")
    _, q = dqn.dqn3_plan(pd, packed, ns, H, na, act, xd, want_actions=False)
    ref = oracle.mlp3_forward(p, ns, H, na, act, x)
    qh = q.cpu().numpy()
    _assert_q_close(qh, ref, act)
    # plan!: eps-greedy with the EXPLORE stream; compare with the oracle's selection on the GPU's own q values
    # (bit-exact integer parity), and with the oracle's q where the decision is not within tolerance of a tie
    for eps in (0.0, 0.3, 1.0):
        a, q2 = dqn.dqn3_plan(pd, packed, ns, H, na, act, xd, eps, 17, 5, 42)
        assert torch.equal(q2, q)
        sel = oracle.eps_greedy_select(qh.astype(np.float32), eps, 17, 42, env_id_base=5)
        assert np.array_equal(a.cpu().numpy(), sel)


def _fill_ring(traces, oring, ns, n_env, steps, rng, na=2):
    obs = rng.standard_normal((ns, n_env)).astype(np.float32)
    traces.push_state_(torch.as_tensor(obs, device="cuda"))
    oring.push_state(obs)
    for _ in range(steps):
        nobs = rng.standard_normal((ns, n_env)).astype(np.float32)
        a = rng.integers(0, na, n_env).astype(np.int32)
        r = (rng.standard_normal(n_env) * 2).astype(np.float32)
        term = (rng.random(n_env) < 0.2).astype(np.uint8)
        traces.push_transition_(torch.as_tensor(nobs, device="cuda"), torch.as_tensor(a, device="cuda"),
                                torch.as_tensor(r, device="cuda"), torch.as_tensor(term, device="cuda"))
        oring.push_transition(nobs, a, r, term)


def _check_grad(g, ref, ns, na):
    o = 0
    for name, n in (("W1", H * ns), ("b1", H), ("W2", H * H), ("b2", H), ("W3", na * H), ("b3", na)):
        a, b = g[o:o + n], ref[o:o + n]
        assert_grad_close(a, b, BF16_GRAD_TOL, f"dqn3w {name} ns={ns} na={na}")
        o += n


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("batch,ns,na", [(32, 4, 2), (512, 4, 2), (1000, 4, 2), (20000, 4, 2),
                                         (300, 2, 3), (300, 3, 3)])
def test_dqn3w_grad_vs_oracle(batch, ns, na, act):
    """every kernel instantiation (CartPole / MountainCar / Pendulum shapes x relu / tanh) against the oracle"""
    import rlhip
    from rlhip import dqn

    n_env, cap = 64, 40
    rng = np.random.default_rng(batch + act)
    traces = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    _fill_ring(traces, oring, ns, n_env, 57, rng, na)  # wraps
    p, tp = _net(ns, na, 11), _net(ns, na, 12)
    pd, tpd = torch.as_tensor(p, device="cuda"), torch.as_tensor(tp, device="cuda")
    packed, tpacked = dqn.mlp3_pack(pd, ns, H, na), dqn.mlp3_pack(tpd, ns, H, na)
    td = torch.zeros(batch, device="cuda")
    g, loss = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 7, 3, td=td)
    idx = oring.sample_indices(batch, 7, 3)
    s, a, r, t, sn = oring.gather(idx)
    rl, rg, rq = oracle.dqn3_loss_grad(ns, H, na, act, p, tp, s, a, r, t, sn, 0.99, 1.0)
    assert abs(float(loss) - rl) <= 2e-5 * max(1.0, abs(rl))
    _check_grad(g.cpu().numpy(), rg, ns, na)
    # explicit indices (the prioritized path) give the same result as the inline draw
    g2, loss2 = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 0, 0,
                              idx=torch.as_tensor(idx, device="cuda"))
    assert torch.equal(g2, g) and torch.equal(loss2, loss)
    # run-to-run determinism (fixed summation order, no atomics)
    g3, _ = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 7, 3)
    assert torch.equal(g3, g)
    # td errors
    qn = oracle.mlp3_forward(tp, ns, H, na, act, sn)
    y = r + 0.99 * (1 - t.astype(np.float32)) * qn.max(0)
    ref_td = np.abs(rq[a, np.arange(batch)] - y)
    terr = np.abs(td.cpu().numpy() - ref_td) / (1 + ref_td)
    if act == 0:
        assert terr.max() <= 1e-4, terr.max()
    else:  # tanh: the rare bf16 rounding flip of an h1 element (module docstring) reaches the TD error of its sample
        assert (terr <= 1e-4).mean() >= 0.999 and terr.max() <= 5e-3, ((terr <= 1e-4).mean(), terr.max())


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("batch,ns,na", [(1000, 4, 2), (20000, 4, 2), (300, 2, 3), (300, 3, 3)])
def test_dqn3w_padded_lds_copy_of_the_backward_kernel_is_bit_identical(batch, ns, na, act):
    """ppo3w_bwd_kernel<.., PAD = true> (rlhip_debug_w3_dzf_pad: the bank-conflict-free LDS copy, off by default) on the DQN learner's
    instantiations: same gradient, bit for bit"""
    import ctypes as C

    import rlhip
    from rlhip import _lib, dqn

    fn_ = _lib.lib.rlhip_debug_w3_dzf_pad_info
    fn_.restype, fn_.argtypes = C.c_int32, [C.c_int32, C.POINTER(C.c_double)]
    info = (C.c_double * 6)()

    def fn(on):
        return fn_(on, info)

    n_env, cap = 64, 40
    rng = np.random.default_rng(batch + act)
    traces = rlhip.CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=ns)
    oring = oracle.Ring(cap, n_env, ns)
    _fill_ring(traces, oring, ns, n_env, 57, rng, na)
    pd, tpd = torch.as_tensor(_net(ns, na, 11), device="cuda"), torch.as_tensor(_net(ns, na, 12), device="cuda")
    packed, tpacked = dqn.mlp3_pack(pd, ns, H, na), dqn.mlp3_pack(tpd, ns, H, na)
    fn(-1)
    prev = int(info[0])  # the mode (2 = by the chip's clock, the default), restored below
    fn(0)
    try:
        g0, l0 = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 7, 3)
        g0, l0 = g0.clone(), l0.clone()
        fn(1)
        g1, l1 = dqn.dqn3_grad(traces, H, na, act, pd, packed, tpd, tpacked, batch, 0.99, 1.0, 7, 3)
        g1, l1 = g1.clone(), l1.clone()
    finally:
        fn(prev)
    assert torch.equal(g0, g1) and torch.equal(l0, l1) and float(g0.abs().max()) > 0


def test_dqn3w_learner_trains_and_prioritized_write_back():
    """QBasedPolicy with the 3-layer net on CartPole: plan!/optimise! run, the target network re-packs on sync,
    priorities are written back for the sampled keys."""
    import rlhip

    n = 256
    env = rlhip.CartPoleEnv(n, seed=1)
    net = rlhip.HipApproximator(4, H, 2, seed=1, layers=3)
    tn = rlhip.TargetNetwork(net, sync_freq=5)
    learner = rlhip.DQNLearner(tn, batchsize=128, min_replay_history=n, seed=1)
    policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=50, seed=1))
    traces = rlhip.CircularPrioritizedTraces(capacity=64, n_env=n, obs_dim=4, default_priority=10.0)
    agent = rlhip.Agent(policy, rlhip.Trajectory(traces))
    p0 = net.params.clone()
    rlhip.run(agent, env, rlhip.StopAfterNSteps(30))
    torch.cuda.synchronize()
    assert learner.n_updates >= 20
    assert not torch.equal(net.params, p0) and torch.isfinite(net.params).all()
    assert torch.isfinite(learner.loss).all()
    # packed copies follow the master weights
    from rlhip import dqn

    assert torch.equal(net.packed, dqn.mlp3_pack(net.params, 4, H, 2))
    assert torch.equal(tn.target_packed, dqn.mlp3_pack(tn.target, 4, H, 2))
    # some leaves no longer carry the default priority, the tree is consistent
    leaves = traces.priorities[traces.priorities.numel() // 2:][:traces.n_leaves]
    assert (leaves != 10.0).any() and (leaves >= 0).all()
    t = traces.priorities
    P = t.numel() // 2
    assert torch.equal(t[1:P], t[2:2 * P:2] + t[3:2 * P:2])
    ref = oracle.per_priority(learner.td.cpu().numpy() * 0 + 0.5, 1e-6, 0.6)
    assert abs(float(ref[0]) - (0.5 + 1e-6) ** 0.6) < 1e-6


@pytest.mark.parametrize("kind,batch,clip", [("cartpole", 512, 0.5), ("cartpole", 100, 0.0), ("pendulum", 1000, 1e6),
                                             ("mountaincar", 4096, 0.05)])
def test_dqn3w_update_is_bit_identical_to_grad_clip_adam_pack(kind, batch, clip):
    """rlhip_dqn3_update_f32 at hidden 256 (gradient kernels, then reduce + sum of squares / clip + Adam + bf16 re-pack) ==
    rlhip_dqn3_grad_f32 + rlhip_clip_adam_f32 + rlhip_mlp3_pack_bf16, bit for bit, over repeated calls"""
    import rlhip
    from rlhip import dqn, ops

    ns, na = {"cartpole": (4, 2), "pendulum": (3, 3), "mountaincar": (2, 3)}[kind]
    n, h = 64, H
    tr = rlhip.CircularArraySARTSTraces(capacity=32, n_env=n, obs_dim=ns)
    tr.records.normal_()  # every word of every 64-byte record: s, s_next and (overwritten below) a, r, t
    tr.action.random_(0, na)
    tr.reward.normal_()
    tr.terminal.copy_((torch.rand(tr.terminal.shape, device="cuda") < 0.1).to(torch.uint8))
    tr.rb.len_sa, tr.rb.len_rt = 33, 32
    tp = dqn.mlp3_init(ns, h, na, 2, 1)
    tpk = dqn.mlp3_pack(tp, ns, h, na)
    st = []
    for _ in range(2):
        p = dqn.mlp3_init(ns, h, na, 1, 0)
        st.append(dict(p=p, pk=dqn.mlp3_pack(p, ns, h, na), m=torch.zeros_like(p), v=torch.zeros_like(p),
                       g=torch.empty_like(p), bp=torch.tensor([0.9, 0.999], device="cuda"),
                       loss=torch.empty(1, device="cuda"), gn=torch.zeros(1, device="cuda"),
                       ws=dqn.dqn3_workspace(ns, h, na, batch)))
    a, b = st
    for it in range(4):
        dqn.dqn3_grad(tr, h, na, it % 2, a["p"], a["pk"], tp, tpk, batch, 0.99, 1.0, 7, it, None, a["ws"], a["g"], a["loss"])
        ops.clip_adam_(a["p"], a["g"], a["m"], a["v"], a["bp"], 0.5, clip, 1e-2, 0.9, 0.999, 1e-8, a["gn"])
        dqn.mlp3_pack(a["p"], ns, h, na, a["pk"])
        dqn.dqn3_update(tr, h, na, it % 2, b["p"], b["pk"], tp, tpk, batch, 0.99, 1.0, 7, it, b["ws"], b["g"], b["loss"],
                        b["m"], b["v"], b["bp"], 0.5, clip, 1e-2, 0.9, 0.999, 1e-8, b["gn"])
        for k in ("p", "pk", "m", "v", "g", "bp", "loss", "gn"):
            assert torch.equal(a[k], b[k]), (it, k)
    assert not torch.equal(a["p"], dqn.mlp3_init(ns, h, na, 1, 0))
