"""CPU-side checks: the C-ABI library loads and exports every symbol include/rlhip.h declares, argument
validation reports errors the documented way (no compute is launched: there is no GPU here), and the
host-side logic of the mirror (stop conditions, controllers, explorer schedule) follows the reference."""
import ctypes as C
import json
import os
import re
import sys

import pytest

import rlhip
from rlhip import _lib

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    syms = _lib.declared_symbols()
    assert len(syms) >= 60
    missing = [s for s in syms if not hasattr(_lib.lib, s)]
    assert not missing, missing
    # and every declared function has a ctypes prototype in the Python glue
    assert not [s for s in syms if s not in _lib._PROTOS]
    # INTEGRATION.md quotes the number of entry points (it said 141 while the header had grown): keep it honest
    import re

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"\((\d+) `extern \"C\"` entry points", text)
    assert m and int(m.group(1)) == len(syms), (m and m.group(1), len(syms))


def test_abi_version_and_error_reporting():
    assert _lib.lib.rlhip_abi_version() == 2 == _lib.EXPECTED_ABI
    # argument validation happens before any HIP call: NULL output pointer -> RLHIP_EINVAL + message
    with pytest.raises(_lib.RLHipArgumentError) as e:
        _lib.call("rlhip_fill_uniform_f32", None, 16, 0, 0, 0, None)
    assert "invalid argument" in str(e.value)
    rc = _lib.lib.rlhip_env_reset(7, 0, None, None, 1, 0, 0, None, None)
    assert rc == -1 and b"kind" in _lib.lib.rlhip_last_error()


def test_struct_layouts_match_header():
    """sizeof of the POD structs as the C compiler sees them (guards the ctypes mirrors)."""
    import subprocess
    import tempfile

    src = r'''
    #include <stdio.h>
    #include "rlhip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rlhip_cartpole_cfg), sizeof(rlhip_pendulum_cfg),
               sizeof(rlhip_mountaincar_cfg), sizeof(rlhip_env_state), sizeof(rlhip_ring), sizeof(rlhip_ppo_cfg),
               sizeof(rlhip_ppo_traj), sizeof(rlhip_comm_desc), sizeof(rlhip_dqn_step_args));
        return 0;
    }'''
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "t.c"), "w") as f:
            f.write(src)
        subprocess.run(["gcc", "-I", inc, os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    sizes = [int(x) for x in out]
    mirrors = [_lib.CartPoleCfg, _lib.PendulumCfg, _lib.MountainCarCfg, _lib.EnvState, _lib.Ring, _lib.PPOCfg,
               _lib.PPOTraj, _lib.CommDesc, _lib.DqnStepArgs]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_communicator_entry_points_validate_before_touching_a_device():
    """SURVEY 8b export list: rlhip_comm_init / rlhip_allreduce_grads exist and reject bad arguments with RLHIP_EINVAL
    (no GPU here: nothing may be launched); the plain-C host that drives them without PyTorch compiles against the
    header with -Wall (tests/abi_host/abi_host.c; it RUNS in tests/test_gpu_abi_host.py)."""
    import subprocess
    import tempfile

    h = C.c_void_p()
    for rank, world in ((2, 2), (-1, 4), (0, 0), (0, 17)):
        with pytest.raises(_lib.RLHipArgumentError):
            _lib.call("rlhip_comm_init", rank, world, None, 100, C.byref(h))
    with pytest.raises(_lib.RLHipArgumentError):
        _lib.call("rlhip_comm_init", 0, 2, None, 0, C.byref(h))  # cap
    for fn, args in (("rlhip_allreduce_grads", (None, None, 4, None)), ("rlhip_comm_check", (None,)),
                     ("rlhip_comm_destroy", (None,)), ("rlhip_comm_unique_id", (None,))):
        with pytest.raises(_lib.RLHipArgumentError):
            _lib.call(fn, *args)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c",
                            os.path.join(root, "tests", "abi_host", "abi_host.c"), "-o", os.path.join(d, "a.o")],
                           capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_default_configs_match_reference_defaults():
    c = _lib.CartPoleCfg()
    _lib.call("rlhip_cartpole_default", C.byref(c))
    # CartPoleEnv.jl:22-32
    assert (c.gravity, c.masscart, c.masspole, c.halflength, c.forcemag, c.dt, c.thetathreshold_deg, c.xthreshold,
            c.max_steps) == (9.8, 1.0, 0.1, 0.5, 10.0, 0.02, 12.0, 2.4, 200)
    p = _lib.PendulumCfg()
    _lib.call("rlhip_pendulum_default", C.byref(p))
    assert (p.max_speed, p.max_torque, p.g, p.m, p.l, p.dt, p.max_steps, p.continuous, p.n_actions) == \
        (8, 2, 10, 1, 1, 0.05, 200, 1, 3)  # PendulumEnv.jl:41-53
    m = _lib.MountainCarCfg()
    _lib.call("rlhip_mountaincar_default", C.byref(m), 0)
    assert (m.min_pos, m.max_pos, m.max_speed, m.goal_pos, m.power, m.gravity) == (-1.2, 0.6, 0.07, 0.5, 0.001, 0.0025)
    _lib.call("rlhip_mountaincar_default", C.byref(m), 1)
    assert (m.goal_pos, m.power) == (0.45, 0.0015)  # MountainCarEnv.jl:74
    q = _lib.PPOCfg()
    _lib.call("rlhip_ppo_default", C.byref(q))
    # blog index.html:15257-15278
    assert (round(q.gamma, 6), round(q.lam, 6), round(q.clip_range, 6), round(q.max_grad_norm, 6), q.n_epochs,
            q.n_microbatches, q.hidden) == (0.99, 0.95, 0.1, 0.5, 4, 4, 256)
    assert _lib.lib.rlhip_ppo_nparams(0, C.byref(q)) == 3331  # 3 331 parameters
    assert _lib.lib.rlhip_mlp2_nparams(4, 128, 2) == 4 * 128 + 128 + 2 * 128 + 2


def test_three_layer_widths_accepted_by_the_host_side_of_the_abi():
    """layers = 3 (MFMA hidden layer): hidden 128 (dqn3.hip / ppo3.hip) and 256 (ppo3w.hip) -- parameter counts equal the
    oracle's, workspace sizes are positive and grow with the micro-batch, other widths and 3-action PPO heads are refused
    (pure host logic: no device is touched)"""
    import oracle

    for kind, name, cont, ns in ((0, "cartpole", 0, 4), (1, "pendulum", 1, 3)):
        for h in (128, 256):
            q = _lib.PPOCfg()
            _lib.call("rlhip_ppo_default", C.byref(q))
            q.hidden, q.layers, q.continuous = h, 3, cont
            np_ = _lib.lib.rlhip_ppo_nparams(kind, C.byref(q))
            per = lambda nout: h * ns + h + h * h + h + nout * h + nout  # noqa: E731
            assert np_ == per(2) + per(1)
            assert np_ == oracle.ppo_nparams(oracle.KIND[name], oracle.ppo_default(hidden=h, continuous=cont, layers=3))
            w1 = _lib.lib.rlhip_ppo_workspace_bytes(kind, C.byref(q), 256, 16)
            w2 = _lib.lib.rlhip_ppo_workspace_bytes(kind, C.byref(q), 4096, 128)
            assert 0 < w1 < w2 < (1 << 31)
        q.hidden = 64
        assert _lib.lib.rlhip_ppo_nparams(kind, C.byref(q)) < 0 and "128 or 256" in _lib.last_error()
    q = _lib.PPOCfg()
    _lib.call("rlhip_ppo_default", C.byref(q))
    q.hidden, q.layers = 256, 3
    assert _lib.lib.rlhip_ppo_nparams(2, C.byref(q)) < 0  # MountainCar: three actions, not instantiated for layers = 3
    for h in (128, 256):
        assert _lib.lib.rlhip_mlp3_nparams(4, h, 2) == oracle.mlp3_nparams(4, h, 2)
        assert _lib.lib.rlhip_mlp3_packed_elems(h) == 2 * h * h
        a, b = _lib.lib.rlhip_dqn3_workspace_bytes(4, h, 2, 512), _lib.lib.rlhip_dqn3_workspace_bytes(4, h, 2, 131072)
        assert 0 < a < b


def test_fused_act_entry_points_say_what_they_support_and_refuse_the_rest_before_touching_a_device():
    """rlhip_dqn_act_supported / rlhip_dqn3_act_supported (pure host predicates) and the argument checks of rlhip_dqn3_act_f32:
    the vec-step call falls back to the separate launches exactly where these say no"""
    sup2, sup3 = _lib.lib.rlhip_dqn_act_supported, _lib.lib.rlhip_dqn3_act_supported
    assert sup2(0, 4096, 128) == 1 and sup2(1, 4096, 64) == 1 and sup2(2, 1, 256) == 1
    assert sup2(0, 4096, 100) == 0 and sup2(3, 4096, 128) == 0 and sup2(0, 0, 128) == 0
    # 3-layer net: hidden 128 only, the three classic-control envs with their discrete action sets, up to 32768 envs
    assert sup3(0, 4096, 128, 2) == 1 and sup3(1, 192, 128, 3) == 1 and sup3(2, 32768, 128, 3) == 1
    assert sup3(0, 4096, 256, 2) == 0 and sup3(0, 4096, 128, 3) == 0 and sup3(1, 4096, 128, 2) == 0
    assert sup3(2, 32769, 128, 3) == 0 and sup3(3, 4096, 128, 2) == 0 and sup3(0, 0, 128, 2) == 0
    rc = _lib.lib.rlhip_dqn3_act_f32(0, None, None, 4096, None, None, 128, 2, 0, 0.1, 1, 0, 1, 0, None, None, None, None, None, None)
    assert rc != 0 and "NULL" in _lib.last_error()


def test_get_eps_host_function_golden():
    with open(os.path.join(G, "select.json")) as f:
        S = json.load(f)
    p = S["get_eps_params"]
    for case in S["get_eps"]:
        e = _lib.lib.rlhip_get_eps(0 if case["kind"] == "linear" else 1, p["eps_stable"], p["eps_init"],
                                   p["warmup_steps"], p["decay_steps"], case["step"])
        assert abs(e - case["expect"]) <= case.get("atol", 1e-12), case["src"]
    ex = rlhip.EpsilonGreedyExplorer(0.1, kind="exp", eps_init=0.9, warmup_steps=100, decay_steps=100)
    assert ex.get_eps(150) == pytest.approx(0.5852245277701068)
    # EpsilonGreedyExplorer(eps): linear, eps_init 1.0, warmup = decay = 0 -> always eps_stable (:47-78)
    assert rlhip.EpsilonGreedyExplorer(0.3).get_eps(1) == 0.3


def test_stop_after_n_steps_golden():
    with open(os.path.join(G, "select.json")) as f:
        c = json.load(f)["stop_after_n_steps"]
    s = rlhip.StopAfterNSteps(c["n"])
    assert sum(bool(s.check_()) for _ in range(c["calls"])) == c["n_true"]  # core/stop_conditions.jl:8
    a, b = rlhip.StopAfterNSteps(3), rlhip.StopAfterNSteps(5)
    any_, all_ = rlhip.StopIfAny(a, b), rlhip.StopIfAll(rlhip.StopAfterNSteps(3), rlhip.StopAfterNSteps(5))
    assert [any_.check_(None, None) for _ in range(4)] == [False, False, True, True]
    assert [all_.check_(None, None) for _ in range(6)] == [False, False, False, False, True, True]


def test_insert_sample_ratio_controller():
    c = rlhip.InsertSampleRatioController(ratio=0.5, threshold=4)
    allowed = []
    for _ in range(10):
        c.on_insert_(1)
        allowed.append(c.on_sample_())
    # nothing before 4 inserts; afterwards n_sampled <= (n_inserted - threshold) * ratio
    assert allowed[:3] == [False, False, False]
    assert allowed[3] is True and c.n_sampled <= (c.n_inserted - 4) * 0.5 + 1


def test_target_network_counter_logic():
    # TargetNetwork.optimise! counter semantics (target_network.jl:74-86) without touching the device
    class FakeNet:
        def __init__(self):
            self.params = None
            self.calls = 0

        def optimise_(self, grad, **kw):
            self.calls += 1

    tn = rlhip.TargetNetwork.__new__(rlhip.TargetNetwork)
    tn.network, tn.sync_freq, tn.rho, tn.n_optimise, tn.target = FakeNet(), 3, 0.0, 0, None
    synced = []
    import rlhip.ops as ops

    orig = ops.polyak_
    ops.polyak_ = lambda dst, src, rho: synced.append(tn.network.calls)
    try:
        seen = []
        for _ in range(4):
            tn.optimise_(None)
            seen.append(tn.n_optimise)
    finally:
        ops.polyak_ = orig
    assert seen == [1, 2, 0, 1] and synced == [3]  # golden: tests/golden/select.json target_sync
    with pytest.raises(AssertionError):
        rlhip.TargetNetwork(FakeNet(), rho=1.5)


def test_space_membership():
    sp = rlhip.Space(n=2)
    assert [1, 2] in sp and [0] not in sp and [3] not in sp and len(sp) == 2
    box = rlhip.Space([-2.0], [2.0])
    assert [0.5] in box and [2.5] not in box


def test_no_cpu_fallback_in_product_sources():
    """The product package never imports the oracle and has no CPU fallback path."""
    pkg = os.path.dirname(rlhip.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle", src, re.M), fn
    csrc = os.path.join(os.path.dirname(pkg), "csrc")
    for fn in os.listdir(csrc):
        assert "rl_oracle" not in open(os.path.join(csrc, fn)).read(), fn


def test_checkpoint_walker_round_trip(tmp_path):
    """state_dict / load_state_dict (rlhip/checkpoint.py; the JLD2 hook recipe docs/src/How_to_use_hooks.md:122-167):
    tensors, scalars, C-struct fields (not pointers), nested objects; in-place restore; strictness"""
    import numpy as np
    import torch

    from rlhip import checkpoint as ck

    class Node:
        __module__ = "rlhip.fake"

        def __init__(self, k):
            self.w = torch.arange(6, dtype=torch.float32).reshape(2, 3) * k
            self.counter, self.rate, self.flag, self.name = 3 * k, 0.5 * k, bool(k % 2), f"n{k}"
            self.cfg = _lib.PendulumCfg()
            self.cfg.max_speed, self.cfg.n_actions = 8.0 + k, 3 + k
            self.st = _lib.EnvState()          # pointers only: must not be exported
            self.foreign = re.compile("x")     # not ours: ignored
            self.child = None

    a, b = Node(1), Node(2)
    a.child, a.items = b, [torch.ones(2), {"z": torch.zeros(1, dtype=torch.int32)}]
    a.again = b                                # second path to the same object: visited once
    d = ck.state_dict(a)
    assert set(d) == {"w", "counter", "rate", "flag", "name", "cfg.max_speed", "cfg.max_torque", "cfg.g", "cfg.m",
                      "cfg.l", "cfg.dt", "cfg.max_steps", "cfg.continuous", "cfg.n_actions", "items/0", "items/1/z"} | \
        {f"again/{k}" for k in ("w", "counter", "rate", "flag", "name", "cfg.max_speed", "cfg.max_torque", "cfg.g",
                                "cfg.m", "cfg.l", "cfg.dt", "cfg.max_steps", "cfg.continuous", "cfg.n_actions")}
    assert isinstance(d["w"], np.ndarray) and d["again/counter"] == 6 and d["cfg.n_actions"] == 4
    n = ck.save_checkpoint(tmp_path / "c.npz", a)
    assert n == len(d)
    a2, b2 = Node(5), Node(7)
    a2.child, a2.items, a2.again = b2, [torch.zeros(2), {"z": torch.ones(1, dtype=torch.int32)}], b2
    w_ptr = a2.w.data_ptr()
    ck.load_checkpoint(tmp_path / "c.npz", a2)
    assert a2.w.data_ptr() == w_ptr and torch.equal(a2.w, a.w) and torch.equal(b2.w, b.w)
    assert (a2.counter, a2.rate, a2.flag, a2.name) == (3, 0.5, True, "n1") and type(a2.counter) is int
    assert (b2.counter, b2.cfg.n_actions, b2.cfg.max_speed) == (6, 5, 10.0)
    assert torch.equal(a2.items[0], torch.ones(2)) and int(a2.items[1]["z"]) == 0
    del d["rate"]
    with pytest.raises(KeyError):
        ck.load_state_dict(a2, d)
    ck.load_state_dict(a2, d, strict=False)
    d["w"] = np.zeros((3, 2), np.float32)
    with pytest.raises(ValueError):
        ck.load_state_dict(a2, d, strict=False)
    assert "rate" not in ck.state_dict(a, skip=("rate",))


def test_debug_timer_sections_follow_the_reference_labels():
    """RLCore.timer + TimerOutputs.enable_debug_timings (RLCore/test/core/base.jl:41-57): disabled sections are a shared
    no-op; enabled ones count calls under the labels of run.jl:46-72 (host clock here, HIP events on a GPU box)"""
    import rlhip
    from rlhip import timing as tmod

    t = tmod.TimerOutput()
    assert t("plan!") is t("act!")            # disabled: the same null context
    t.enabled = True                           # host-clock only (no device on this box)
    for _ in range(3):
        with t("plan!"):
            pass
        with t("act!"):
            pass
    d = t.todict()
    assert list(d) == ["plan!", "act!"] and d["plan!"]["ncalls"] == 3 and d["act!"]["host_ns"] > 0
    assert "plan!" in str(t) and isinstance(rlhip.timer, tmod.TimerOutput)
    t.reset_()
    assert not t.sections


def test_packed_episode_counter_capacity_and_host_guard():
    """the packed reset counter has 32 - tbits bits.  The ABI states its capacity, the kernels saturate (GPU
    test), and the host mirror refuses the call that could reach it instead of silently repeating reset draws."""
    import ctypes as C

    from rlhip import _lib

    cap = _lib.lib.rlhip_env_packed_episode_capacity
    assert cap(200) == (1 << 24) - 1          # CartPole default: 8 step bits (t reaches 201), 24 counter bits
    assert cap(1) == (1 << 30) - 1            # 2 step bits
    assert cap((1 << 20) - 2) == (1 << 12) - 1
    assert cap((1 << 20) - 1) == -1 and cap(0) == -1


def test_bare_bench_with_gpus_n_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus 8` without WORLD_SIZE (the form the driver uses at N = 1) must not die on
    plumbing: it re-executes itself under torch.distributed.run, one rank per GPU, loopback rendezvous, same arguments"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class _Stop(Exception):
        pass

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, argv
        raise _Stop()

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    with pytest.raises(_Stop):
        bench.main()
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(a[a.index("--master-port") + 1]) < 65536
    assert a[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "3"]
    # under the launcher (WORLD_SIZE set) a mismatch is still an error, not a silent single-rank run
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit):
        bench.main()
