"""GPU parity of the width-256 three-layer PPO actor / critic (csrc/ppo3w.hip: cfg.layers = 3, hidden = 256 -- the
streaming forward / backward / dW2 kernels and the 8-wave rollout) against the oracle (oracle/rlo_learn.c with
layers = 3, hidden = 256).  Same contract and bars as tests/test_gpu_ppo3.py (hidden = 128): head outputs
2e-5 * (1 + |x|), gradients BF16_GRAD_TOL * max|g| per tensor, integer actions bit-exact away from sampling ties."""
import numpy as np
import pytest
import torch

import oracle
from conftest import BF16_GRAD_TOL, assert_grad_close  # noqa: E402

pytestmark = pytest.mark.gpu
H = 256


def _setup(kind, n, T, seed=5, **kw):
    import rlhip

    env = rlhip.HipVecEnv(kind, n, seed=seed)
    pol = rlhip.PPOPolicy(env, update_freq=T, hidden=H, seed=seed, layers=3, **kw)
    return env, pol


def test_parameter_count_and_workspace():
    env, pol = _setup("pendulum", 64, 4)
    ns = 3
    per = lambda nout: H * ns + H + H * H + H + nout * H + nout  # noqa: E731
    assert pol.np_actor == per(2) and pol.np == per(2) + per(1)
    assert pol.np == oracle.ppo_nparams(oracle.KIND["pendulum"], oracle.ppo_default(hidden=H, continuous=1, layers=3))


@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,cont", [("cartpole", False), ("pendulum", True)])
def test_rollout_matches_oracle(kind, cont, act):
    n, T = 200, 6  # 7 workgroups of 32 envs, the last one ragged
    a = {"relu": 0, "tanh": 1}[act]
    env, pol = _setup(kind, n, T, act=a)
    params = pol.params.cpu().numpy()
    oenv = oracle.VecEnv(kind, n, seed=5)
    ocfg = oracle.ppo_default(hidden=H, continuous=int(cont), layers=3, act=a)
    otr = oracle.PPOTraj(oracle.KIND[kind], n, T, continuous=cont)
    oracle.ppo_rollout(oenv, T, ocfg, params, otr, 0)
    pol.rollout_()
    tr = pol.trajectory
    v, ov = tr.value.cpu().numpy(), otr.value
    vtol = 1e-4 if (cont or act == "tanh") else 2e-5
    if cont:
        np.testing.assert_allclose(tr.obs[0].cpu().numpy(), otr.obs[0], rtol=0, atol=2e-7)
    else:
        assert np.array_equal(tr.obs[0].cpu().numpy(), otr.obs[0])
    assert np.all(np.abs(v[0] - ov[0]) <= vtol * (1 + np.abs(ov[0]))), np.abs(v[0] - ov[0]).max()
    if cont:
        af, oaf = tr.action_f.cpu().numpy().reshape(T, n), otr.action_f.reshape(T, n)
        assert np.all(np.abs(af[0] - oaf[0]) <= 1e-4 * (1 + np.abs(oaf[0])))
        assert np.all(np.abs(tr.logp[0].cpu().numpy() - otr.logp[0]) <= 1e-3)
    else:
        ai, oai = tr.action_i.cpu().numpy(), otr.action_i
        agree = (ai == oai)
        assert agree[0].mean() >= 0.995
        same = agree.all(0)
        assert same.mean() >= 0.95
        for name in ("reward", "terminal"):
            assert np.array_equal(getattr(tr, name).cpu().numpy()[:, same], getattr(otr, name)[:, same])
        assert np.array_equal(tr.obs.cpu().numpy()[:, :, same], otr.obs[:, :, same])
        assert np.all(np.abs(v[:, same] - ov[:, same]) <= 1e-4 * (1 + np.abs(ov[:, same])))
    # the GAE scan fused into the rollout launch equals the stand-alone scan on the same arrays
    adv = tr.adv.cpu().numpy().copy()
    pol.gae_()
    assert np.array_equal(tr.adv.cpu().numpy(), adv)


def _oracle_grad(pol, env, cont, ocfg, epoch, mb, bm, total, n, T):
    tr = pol.trajectory
    ns = env.odim
    na = 1 if cont else 2
    f = np.array([oracle.permute(pol.seed, epoch, total, mb * bm + b) for b in range(bm)])
    t, i = f // n, f % n
    obs = tr.obs.cpu().numpy()[t, :, i].T.copy()
    action = tr.action_f.cpu().numpy().reshape(T, n)[t, i][None, :] if cont else tr.action_i.cpu().numpy()[t, i]
    return oracle.ppo_loss_grad(ocfg, ns, na, pol.params.cpu().numpy(), obs, action, tr.logp.cpu().numpy()[t, i],
                                tr.adv.cpu().numpy()[t, i], tr.ret.cpu().numpy()[t, i])


def _check_grad(pol, g, og, ns, label, q99_tol=None):
    np_a = pol.np_actor
    for name, a, b in (("actor", g[:np_a], og[:np_a]), ("critic", g[np_a:], og[np_a:])):
        o = 0
        nout = 2 if name == "actor" else 1
        for tname, sz in (("W1", H * ns), ("b1", H), ("W2", H * H), ("b2", H), ("W3", nout * H), ("b3", nout)):
            assert_grad_close(a[o:o + sz], b[o:o + sz], BF16_GRAD_TOL, f"ppo3w {label} {name} {tname} ns={ns}", q99_tol=q99_tol)
            o += sz
        assert o == a.size


@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,cont", [("cartpole", False), ("pendulum", True)])
def test_grad_matches_oracle(kind, cont, act):
    """one micro-batch gradient on a GPU-generated trajectory: 432 samples = 6 full 64-sample tiles + a ragged one"""
    n, T = 96, 9
    a = {"relu": 0, "tanh": 1}[act]
    env, pol = _setup(kind, n, T, n_microbatches=2, act=a)
    pol.rollout_()
    pol.gae_()
    ocfg = oracle.ppo_default(hidden=H, continuous=int(cont), layers=3, n_microbatches=2, act=a)
    total, bm = n * T, (n * T) // 2
    for mb, epoch in ((0, 0), (1, 3)):
        pol.grad_(epoch, mb)
        g = pol.grad.cpu().numpy()
        losses = pol.losses.cpu().numpy()
        og, ol = _oracle_grad(pol, env, cont, ocfg, epoch, mb, bm, total, n, T)
        assert np.all(np.abs(losses - ol) <= 2e-4 * (1 + np.abs(ol))), (losses, ol)
        # Gaussian + tanh at 432 samples is the ill-conditioned corner of the suite: hidden units that feed BOTH heads carry
        # dh2 = w (dls - dmu) (a cancellation that turns the 1e-7 differences of the two sides' expf / division into 1e-4 of
        # that unit's dz2), tanh' = 1 - h^2 cancels near saturation, and one sample with an advantage of -65 is most of
        # max|g| -- each pinned down while building tests/test_gpu_bf16_tight.py (which holds this instantiation to 5e-5 /
        # 1e-5 on well-conditioned inputs).  Its bulk bar is 3e-4 (measured 8.4e-5) instead of the suite's 1e-4.
        _check_grad(pol, g, og, env.odim, f"{kind} {act}", q99_tol=3e-4 if (cont and act == "tanh") else None)
    pol.grad_(3, 1)  # deterministic: fixed summation order, no atomics
    assert np.array_equal(pol.grad.cpu().numpy(), g)


def _w3_pad(on):
    """run-time choice of the backward kernel's LDS copy (csrc/ppo3w.hip: rlhip_debug_w3_dzf_pad_info; not part of the ABI): 0 / 1 force a
    kernel, 2 = by the chip's clock (the default), < 0 query.  Returns (kernel of the next launch, [mode, variant, MHz read last, top MHz,
    switches, sensor found])"""
    import ctypes as C

    from rlhip import _lib

    fn = _lib.lib.rlhip_debug_w3_dzf_pad_info
    fn.restype, fn.argtypes = C.c_int32, [C.c_int32, C.POINTER(C.c_double)]
    info = (C.c_double * 6)()
    v = fn(on, info)
    return v, list(info)


@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,n,T", [("cartpole", 96, 9), ("pendulum", 2048, 40)])
def test_padded_lds_copy_of_the_backward_kernel_is_bit_identical(kind, n, T, act):
    """ppo3w_bwd_kernel<.., PAD = true> (bank-conflict-free transposing reads; profiles/r06_ppo3w.md section 5) reads the same fragments in the
    same order as the unpadded one: the whole gradient, bit for bit -- a ragged 432-sample micro-batch and 640 tiles.  (Which of the two a
    training loop runs is chosen by the chip's clock by default: bit-identity is what makes that choice invisible.)"""
    a = {"relu": 0, "tanh": 1}[act]
    env, pol = _setup(kind, n, T, n_microbatches=2, act=a)
    pol.rollout_()
    pol.gae_()
    mode0 = int(_w3_pad(-1)[1][0])
    try:
        assert _w3_pad(0)[0] == 0
        pol.grad_(1, 1)
        g0, l0 = pol.grad.clone(), pol.losses.clone()
        assert _w3_pad(1)[0] == 1
        pol.grad_(1, 1)
        g1, l1 = pol.grad.clone(), pol.losses.clone()
        assert _w3_pad(-1)[0] == 1
    finally:
        _w3_pad(mode0)
    assert int(_w3_pad(-1)[1][0]) == mode0
    assert torch.equal(g0, g1) and torch.equal(l0, l1)
    assert float(g0.abs().max()) > 0


def test_clock_aware_choice_of_the_backward_kernel_runs_and_reports():
    """mode 2 (the default): the host reads the device's hwmon clock every 512 backward launches and picks the kernel; whatever it picks, the
    update is the same bits as with either kernel forced, and the report is sane (sensor found on these boxes, top clock >= 1 GHz, at most a
    few switches in a second of updates)"""
    env, pol = _setup("pendulum", 2048, 40, n_microbatches=2)
    pol.rollout_()
    pol.gae_()
    mode0 = int(_w3_pad(-1)[1][0])
    try:
        _w3_pad(0)
        pol.grad_(1, 1)
        g0 = pol.grad.clone()
        _w3_pad(2)
        for _ in range(1500):  # ~1500 x 2 backward launches: several readings
            pol.grad_(1, 1)
        torch.cuda.synchronize()
        v, info = _w3_pad(-1)
        assert torch.equal(pol.grad, g0)
        assert int(info[0]) == 2 and v in (0, 1) and int(info[1]) == v
        if info[5]:
            assert info[3] >= 1000.0 and 0.0 <= info[2] <= 1.05 * info[3] and info[4] <= 4, info
    finally:
        _w3_pad(mode0)


def test_grad_many_tiles_per_workgroup_matches_oracle():
    """2048 envs x 40 steps / 2 micro-batches = 40960 samples = 640 tiles: the 512 persistent forward / backward
    workgroups walk one or two tiles each, the 128 sample ranges of the dW2 kernel five tiles each"""
    kind, cont, n, T = "pendulum", True, 2048, 40
    env, pol = _setup(kind, n, T, n_microbatches=2)
    pol.rollout_()
    pol.gae_()
    ocfg = oracle.ppo_default(hidden=H, continuous=1, layers=3, n_microbatches=2)
    total, bm = n * T, (n * T) // 2
    pol.grad_(1, 1)
    g = pol.grad.cpu().numpy()
    losses = pol.losses.cpu().numpy()
    og, ol = _oracle_grad(pol, env, cont, ocfg, 1, 1, bm, total, n, T)
    assert np.all(np.abs(losses - ol) <= 2e-4 * (1 + np.abs(ol))), (losses, ol)
    _check_grad(pol, g, og, env.odim, "pendulum 640 tiles")
    for _ in range(3):
        pol.grad_(1, 1)
        assert np.array_equal(pol.grad.cpu().numpy(), g)


@pytest.mark.parametrize("kind,cont,n,T,nmb", [("pendulum", True, 16, 3, 1),    # 48 samples: one ragged tile, one workgroup
                                               ("cartpole", False, 33, 4, 2),   # 66 per micro-batch: a tile with 2 valid rows
                                               ("cartpole", False, 2048, 17, 2)])  # 272 tiles: one or two passes per workgroup
def test_grad_edge_shapes_match_oracle(kind, cont, n, T, nmb):
    """ragged / tiny micro-batches and a tile count just above the number of persistent workgroups (the clamped prefetch of
    the tail passes, the odd-pass remainder of the pass pairs, a dW2 grid that is not a multiple of 8)"""
    env, pol = _setup(kind, n, T, n_microbatches=nmb)
    pol.rollout_()
    pol.gae_()
    ocfg = oracle.ppo_default(hidden=H, continuous=int(cont), layers=3, n_microbatches=nmb)
    total, bm = n * T, (n * T) // nmb
    mb = nmb - 1
    pol.grad_(2, mb)
    g = pol.grad.cpu().numpy()
    losses = pol.losses.cpu().numpy()
    og, ol = _oracle_grad(pol, env, cont, ocfg, 2, mb, bm, total, n, T)
    assert np.all(np.abs(losses - ol) <= 2e-4 * (1 + np.abs(ol))), (losses, ol)
    _check_grad(pol, g, og, env.odim, f"{kind} n={n} T={T}")
    p0 = pol.params.clone()
    pol.update_()
    torch.cuda.synchronize()
    assert torch.isfinite(pol.params).all() and not torch.equal(pol.params, p0)


@pytest.mark.parametrize("kind", ["cartpole", "pendulum"])
def test_update_runs_and_equals_the_microbatch_protocol(kind):
    """full iterations through the unchanged rlhip_ppo_* entry points: parameters move, stay finite, the fused update
    equals grad -> clip + Adam micro-batch by micro-batch (the multi-GPU code path with one rank)"""
    n, T = 512, 16
    env, pol = _setup(kind, n, T)
    env2, pol2 = _setup(kind, n, T)
    p0 = pol.params.clone()
    pol.rollout_()
    pol2.rollout_()
    assert torch.equal(pol.trajectory.reward, pol2.trajectory.reward)
    pol.update_()
    pol2.gae_()
    for e in range(pol2.cfg.n_epochs):
        for mb in range(pol2.cfg.n_microbatches):
            pol2.grad_(pol2.update_ctr * pol2.cfg.n_epochs + e, mb)
            pol2.apply_(1.0)
    pol2.update_ctr += 1
    torch.cuda.synchronize()
    assert torch.isfinite(pol.params).all() and not torch.equal(pol.params, p0)
    assert torch.equal(pol.params, pol2.params)
    for _ in range(3):
        pol.rollout_()
        pol.update_()
    assert torch.isfinite(pol.params).all() and torch.isfinite(pol.losses).all()


@pytest.mark.parametrize("hidden", [128, 256])
def test_three_layer_ppo_learns_cartpole(hidden):
    """End-to-end sanity of the 3-layer MFMA paths (rollout kernel + learner kernels of the same policy): the mean episode
    length of CartPole rises within a few dozen updates (tests/test_gpu_learners.py::test_ppo_learns_cartpole for layers = 3)"""
    import rlhip

    n, T = 1024, 32
    env = rlhip.HipVecEnv("cartpole", n, seed=1)
    pol = rlhip.PPOPolicy(env, update_freq=T, lr=1e-3, hidden=hidden, layers=3, seed=1)
    first = None
    for it in range(40):
        pol.rollout_()
        pol.update_()
        ep_len = (n * T) / max(1.0, float(pol.trajectory.terminal.sum()))
        if it == 0:
            first = ep_len
    print(f"hidden {hidden}: episode length {first:.1f} -> {ep_len:.1f}")
    assert torch.isfinite(pol.params).all()
    assert ep_len > 2.0 * first, f"episode length {first:.1f} -> {ep_len:.1f}"


@pytest.mark.parametrize("act", ["relu", "tanh"])
@pytest.mark.parametrize("kind,cont,n,T", [("cartpole", False, 96, 9), ("pendulum", True, 96, 9), ("pendulum", True, 4096, 5)])
def test_value_trace_is_the_critic_forward_on_the_recorded_observations(kind, cont, n, T, act):
    """round 3: the rollout kernel runs the actor alone; V(s_0 .. s_T) come from ONE batched critic pass over the
    (T + 1) n recorded observations (ppo3w_fwd_kernel mode 4).  Every entry of the value trace -- every time step, every
    env, the ragged last tile -- against the oracle's 3-layer forward on the SAME observations."""
    a = {"relu": 0, "tanh": 1}[act]
    env, pol = _setup(kind, n, T, act=a)
    pol.rollout_()
    tr = pol.trajectory
    obs = tr.obs.cpu().numpy()                      # (T + 1, ns, n)
    ns = env.odim
    x = obs.transpose(1, 0, 2).reshape(ns, (T + 1) * n)
    pc = pol.params.cpu().numpy()[pol.np_actor:]
    ref = oracle.mlp3_forward(pc, ns, H, 1, a, x).reshape(T + 1, n)
    v = tr.value.cpu().numpy()
    err = np.abs(v - ref) / (1 + np.abs(ref))
    if act == "relu":
        assert err.max() <= 2e-5, err.max()
    else:  # tanh: the rare bf16 rounding flip of an h1 element (ocml vs glibc tanhf ulps)
        assert (err <= 2e-5).mean() >= 0.999 and err.max() <= 5e-3, ((err <= 2e-5).mean(), err.max())
