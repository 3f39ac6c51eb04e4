#!/usr/bin/env python3
"""bench.py -- env-steps/s + learner updates/s of the 4096-env CartPole PPO hot path on N x MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: re-executes itself under the launcher below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one PPO iteration of the hot path on every GPU: fused rollout of T = 32 vec-steps of 4096
CartPole envs (actor/critic forward, Gumbel-max sampling, env step + auto-reset, trajectory writes),
GAE + returns, then 4 epochs x 4 micro-batches of { clipped-surrogate loss + gradient ->
[all-reduce of the flat gradient when N > 1: one-shot peer-to-peer exchange fused into the reduce kernel, RCCL as fallback] -> clip_by_global_norm -> Adam }.  Nothing is
skipped inside the timed region; inputs (env state, parameters) are resident in HBM when it starts.
Workload = BASELINE.json configs[3] per GPU (the config the metric is quoted on; configs[1] is its
DQN sibling and is measured in the `extra` block), synthetic data: random-init weights, Philox-seeded
env states.  Weak scaling: 4096 envs per GPU, env ids / Philox streams disjoint across ranks.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the HBM-bound env-step kernel at the operating point where HBM matters (2^24 envs),
                achieved = 49 algorithmic bytes x envs / mean launch time (HIP events on the launch stream)
                (uniformly random actions; `without_terminations` = the same launch when no episode ends)
  cpu_baseline  the CPU oracle ("port") running the same PPO iteration on a bounded sample: OpenMP build on every
                CPU the container may use (cores stated) and the single-thread figure beside it
  kernels       mean per-launch time of every kernel class of the timed workload (HIP events)
  roofline_extra  the side kernels / configs of BASELINE.json (see roofline_extras)
  allreduce     (N > 1 only) latency of the gradient exchange and the library's bus bandwidth sweep
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

N_ENVS = 4096
T_ROLLOUT = 32
HIDDEN = 256
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured achievable
CARTPOLE_STEP_BYTES = 49     # 24 B read + 25 B written per env-step
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)


SETTLE_S = 0.05  # see event_time_ms


def event_time_ms(fn, iters, lib, stream, settle_s=0.0):
    """mean milliseconds per call of fn(), measured with HIP events on `stream`.  settle_s > 0: fn() is first repeated for that
    long (device busy, stream drained every four calls).  The device needs ~20 - 50 ms of work before a kernel runs at its steady
    rate -- the Pendulum env-step at 2^24 envs reads 159 us after 40 warm-up launches (6 ms), 134 after 300, 129 - 132 from then on
    (tools/pendulum_pitch_ab.py; same ramp as the timed PPO step's, profiles/r04_rollout.md section 3) -- and a roofline leg is
    about the kernel, not about where in the process it happens to run."""
    from rlhip._lib import call

    if settle_s > 0.0:
        t_end = time.perf_counter() + settle_s
        while time.perf_counter() < t_end:
            for _ in range(4):
                fn()
            call("rlhip_stream_sync", stream)
    e0, e1 = C.c_void_p(), C.c_void_p()
    call("rlhip_event_create", C.byref(e0))
    call("rlhip_event_create", C.byref(e1))
    call("rlhip_event_record", e0, stream)
    for _ in range(iters):
        fn()
    call("rlhip_event_record", e1, stream)
    ms = C.c_float(0)
    call("rlhip_event_elapsed_ms", e0, e1, C.byref(ms))
    call("rlhip_event_destroy", e0)
    call("rlhip_event_destroy", e1)
    return ms.value / iters


def roofline_env_step(torch, rlhip, n_envs=1 << 24, iters=20):
    """The HBM-bound env-step kernel at 2^24 CartPole envs, uniformly random actions.

    All envs start an episode together, so terminations come in waves for the first few dozen steps (none before
    step ~8, then a burst, ...).  Two well-defined operating points of the same launch are timed on ONE set of arrays:
      without_terminations  launches 2..6 after a forced reset: no env can have terminated yet (checked)
      steady state          after 60 more steps have de-synchronised the episodes: ~4.5 % of the envs terminate and
                            auto-reset per step (`terminated_per_step` is measured) (scattered episode-counter RMW + a Philox block each); this is `roofline`"""
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr

    # the layout a large vector env gets from the host (rlhip/envs.py): arrays staggered inside one allocation, reset
    # counters packed into the spare bits of the step-counter word (rlhip_env_state packed mode) -- an auto-reset then
    # moves no byte beyond the 49 algorithmic ones
    env = rlhip.HipVecEnv("cartpole", n_envs, seed=1, packed_episode=True)
    # a different random action for every env at every step (16 pre-drawn vectors, cycled): with ONE fixed vector each
    # env would push the same way for ever, episodes would last ~9 steps and stay synchronised in waves
    actions = torch.randint(0, 2, (16, n_envs), dtype=torch.int32, device="cuda")
    a_ptrs = [ptr(actions[k]) for k in range(16)]
    counter = [0]

    def step():
        # pure act!: no observation copies (state(env) IS the state arrays for CartPole)
        counter[0] += 1
        call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, a_ptrs[counter[0] & 15], 1,
             env.seed, 0, None, None, stream_ptr())

    def per_unit(ms):
        gbs = CARTPOLE_STEP_BYTES * n_envs / (ms * 1e-3) / 1e9
        return {"us_per_launch": round(ms * 1e3, 2), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

    for _ in range(3):  # clocks / TLB
        step()
    env.reset_()
    step()
    torch.cuda.synchronize()
    no_term = per_unit(event_time_ms(step, 5, rlhip._lib.lib, stream_ptr()))
    no_term["episodes_finished_during_these_launches"] = int((env.episode_counter() != 2).sum())  # constructor + this reset
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    ms = event_time_ms(step, iters, rlhip._lib.lib, stream_ptr(), SETTLE_S)
    done_frac = float(env._done.float().mean())
    main = per_unit(ms)
    del env, actions
    torch.cuda.empty_cache()
    traffic, traffic_note = measured_traffic(n_envs)
    return {"bound": "hbm", "kernel": "env_step_kernel<CartPole,f32,EPL=4,non-temporal in-place arrays / ordinary stores for reward + done,packed>", "n_envs": n_envs,
            "bytes_per_unit": CARTPOLE_STEP_BYTES, "us_per_launch": main["us_per_launch"],
            "achieved": main["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": main["frac"],
            # HBM bytes per launch from the PMC counters (separate rocprofv3 passes: FETCH_SIZE x2 gfx950
            # correction + WRITE_SIZE) for exactly this kernel / size; the figure is only reported while the kernel
            # sources still hash to what was profiled (PMC_TRAFFIC below) -- otherwise null, never a stale literal
            "traffic": traffic, "algorithmic_bytes": CARTPOLE_STEP_BYTES * n_envs, "traffic_source": traffic_note,
            "env_steps_per_sec": round(n_envs / (ms * 1e-3), 1),
            "layout": "arrays staggered by 4352 B inside one allocation; episode counters packed into the step-counter word",
            "actions": "uniformly random per env and step (16 pre-drawn vectors), episodes de-synchronised by 60 steps before the timed launches",
            "terminated_per_step": round(done_frac, 4), "without_terminations": no_term}


# PMC measurement of the env-step kernel (tools/pmc_env.sh + tools/envstep.py; profiles/r04_pmc.md), valid for
# the kernel sources whose sha256 (first 16 hex digits over csrc/envs.hip + csrc/env_device.h) is `sha`
PMC_TRAFFIC = {"sha": "23d6334bde658716", "n_envs": 1 << 24, "bytes": 822223052.8, "source": "profiles/r06_pmc.md"}

# The same for the other HBM-bound kernels of the bench line: HBM bytes per launch from the PMC counters
# (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, mean of the last launches: tools/pmc_all.sh), each valid only while the
# sources of ITS kernel still hash to what was profiled (`files`, sha256[:16] over them) -- otherwise the entry reports null.
# `fetch_x2: False`: a kernel of scattered 16-byte reads, for which the guide's x 2 streaming calibration of FETCH_SIZE does not
# hold (profiles/r05_pmc.md: the counter tallies 64 bytes per fabric request whatever its size).
PMC_SIDE = {
    # name in the line: (source files, sha over them, bytes per launch, note)      -- tools/pmc_all.sh on the final sources of round 6
    "gae_returns": (["scans.hip", "gae_device.h"], "a163648c638b5c15", 574705766.4, ""),
    "frame_gather_u8": (["ring.hip", "ring_device.h", "sumtree_device.h"], "c150ec0211a5e8a3", 473958707.2, ""),
    "frame_gather_u8_stack_at_sample": (["ring.hip", "ring_device.h", "sumtree_device.h"], "c150ec0211a5e8a3", 387121459.2, ""),
    "env_step_pendulum": (["envs.hip", "env_device.h"], "23d6334bde658716", 755415244.8, ""),
    "env_step_mountaincar": (["envs.hip", "env_device.h"], "23d6334bde658716", 553800192.0, ""),
    "adam_2p26": (["optim.hip", "optim_device.h"], "13f7cbf603a83f38", 1879104512.0, ""),
    "adam_2p22": (["optim.hip", "optim_device.h"], "13f7cbf603a83f38", 117662822.4, ""),
    "polyak_2p26": (["optim.hip", "optim_device.h"], "13f7cbf603a83f38", 805315584.0, ""),
    "polyak_2p22": (["optim.hip", "optim_device.h"], "13f7cbf603a83f38", 50340864.0, ""),
    "push_transition_maxpool": (["ring.hip", "ring_device.h", "sumtree_device.h"], "c150ec0211a5e8a3", 346919219.2, ""),
    "gather_small": (["ring.hip", "ring_device.h", "sumtree_device.h"], "c150ec0211a5e8a3", 111666892.8,
                     "FETCH_SIZE not doubled: scattered 16-byte reads reach the fabric as 64-byte requests, which the counter tallies at 64 bytes (TCC_EA0_RDREQ = 1.02 requests per sample, none of them 32-byte)"),
    "gather_small_hbm": (["ring.hip", "ring_device.h", "sumtree_device.h"], "c150ec0211a5e8a3", 114353049.6,
                     "FETCH_SIZE not doubled (as gather_small)"),
}
PMC_SIDE_SOURCE = "profiles/r06_pmc.md"


def sources_sha(files):
    import hashlib

    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "reinforcementlearning.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def attach_traffic(out):
    """adds `traffic` (PMC bytes per launch or None), `traffic_ratio` (traffic / algorithmic bytes) and `traffic_source` to the
    entries of `out` that PMC_SIDE knows"""
    for name, (files, sha, nbytes, note) in PMC_SIDE.items():
        e = out.get(name)
        if not isinstance(e, dict):
            continue
        now = sources_sha(files)
        if nbytes is not None and sha == now:
            e["traffic"] = nbytes
            if e.get("algorithmic_bytes"):
                e["traffic_ratio"] = round(nbytes / e["algorithmic_bytes"], 4)
            e["traffic_source"] = f"{PMC_SIDE_SOURCE} (sources {'+'.join(files)} sha {now}){'; ' + note if note else ''}"
        else:
            e["traffic"] = None
            e["traffic_source"] = f"not measured for the current kernel sources ({'+'.join(files)} sha {now}; last PMC pass: sha {sha})"
    return out


def env_kernel_sha():
    import hashlib

    h = hashlib.sha256()
    for f in ("envs.hip", "env_device.h"):
        with open(os.path.join(ROOT, "reinforcementlearning.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(n_envs):
    sha = env_kernel_sha()
    if PMC_TRAFFIC["bytes"] is not None and PMC_TRAFFIC["sha"] == sha and PMC_TRAFFIC["n_envs"] == n_envs:
        return PMC_TRAFFIC["bytes"], f"{PMC_TRAFFIC['source']} (kernel sources sha {sha})"
    return None, f"not measured for the current kernel sources (sha {sha}; last PMC pass: sha {PMC_TRAFFIC['sha'] or 'none'})"


def roofline_extras(torch, rlhip, hbm_only=False):
    """Side kernels at the sizes BASELINE.json names: GAE scan (2^20 envs x 32), u8 frame gather from the full 2^20-slot
    29.6 GB ring of config 5 (uniform + prioritized + stack-at-sample), the bf16 Dense layer, the DQN vec-step of config 2
    (2- and 3-layer networks, per-step protocol and one-call-per-step), Pendulum PPO of config 3 (2-layer fp32, 3-layer MFMA)."""
    from rlhip import ops
    from rlhip.ops import stream_ptr
    from rlhip.trajectory import CircularArraySARTSTraces

    lib, s = rlhip._lib.lib, stream_ptr()
    out = {}
    # GAE + returns, N = 2^20 envs x T = 32: 17 B per (env, t) + 4 B per env
    n, T = 1 << 20, 32
    r = torch.rand((T, n), device="cuda") * -16
    v = torch.randn((T + 1, n), device="cuda")
    term = torch.rand((T, n), device="cuda") < 1 / 200
    ops.gae_returns(r, v, term, 0.99, 0.95)
    ms = event_time_ms(lambda: ops.gae_returns(r, v, term, 0.99, 0.95), 10, lib, s, SETTLE_S)
    gb = (17 * n * T + 4 * n) / 1e9
    out["gae_returns"] = {"bound": "hbm", "n_envs": n, "T": T, "us_per_launch": round(ms * 1e3, 1),
                          "achieved": round(gb / (ms * 1e-3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(gb * 1e9)}
    del r, v, term
    # BASELINE configs[4]: 2^20-slot ring of 84x84x4 u8 frames (29.6 GB of states), prioritized sampling
    # (device sum-tree, priorities U(0,1)^0.6) + frame gather, batch 4096 -> 2 * (2 * 28224 + 9) B per sample
    fb, cap, batch = 84 * 84 * 4, int(os.environ.get("RLHIP_BENCH_RING_SLOTS", 1 << 20)), 4096
    tr = rlhip.CircularPrioritizedTraces(capacity=cap, n_env=1, obs_dim=fb, dtype=torch.uint8)
    tr.state.random_(0, 256)
    tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap  # mark the ring full (synthetic frames, no push loop needed)
    keys = torch.arange(cap, dtype=torch.int64, device="cuda")
    tr.set_priority_(keys, ops.fill_uniform(cap, 11, 0, 7) ** 0.6)
    idx, key, prio = tr.sample_prioritized(batch, 11, 0)
    tr.gather(idx)
    bufs = tr.gather(idx)
    ctr = [1]

    def smp():
        rlhip._lib.call("rlhip_ring_sample_prioritized", C.byref(tr.rb), ops.ptr(tr.priorities), batch, 11, ctr[0],
                        ops.ptr(idx), ops.ptr(key), ops.ptr(prio), s)
        ctr[0] += 1

    def g():
        rlhip._lib.call("rlhip_ring_gather", C.byref(tr.rb), ops.ptr(idx), batch, ops.ptr(bufs[0]), ops.ptr(bufs[1]),
                        ops.ptr(bufs[2]), ops.ptr(bufs[3]), ops.ptr(bufs[4]), s)

    def upd():
        rlhip._lib.call("rlhip_sumtree_update", ops.ptr(tr.priorities), cap, ops.ptr(key), ops.ptr(prio), batch, s)

    def sg():  # round 4: the prioritized draw inside the gather launch (one launch instead of two)
        rlhip._lib.call("rlhip_ring_sample_gather_prioritized", C.byref(tr.rb), ops.ptr(tr.priorities), batch, 11, ctr[0],
                        ops.ptr(idx), ops.ptr(key), ops.ptr(prio), ops.ptr(bufs[0]), ops.ptr(bufs[1]), ops.ptr(bufs[2]),
                        ops.ptr(bufs[3]), ops.ptr(bufs[4]), s)
        ctr[0] += 1

    def both():
        sg()
        upd()

    def fresh_gather():  # new indices every launch: no Infinity-Cache hits from a repeated batch
        smp()
        g()

    ms_s = event_time_ms(smp, 10, lib, s, SETTLE_S)
    ms = event_time_ms(fresh_gather, 10, lib, s) - ms_s
    ms_rep = event_time_ms(g, 10, lib, s)
    ms_u = event_time_ms(upd, 10, lib, s)
    ms_all = event_time_ms(both, 10, lib, s)
    ms_sg = event_time_ms(sg, 10, lib, s)
    gb = 2 * (2 * fb + 9) * batch / 1e9
    out["frame_gather_u8"] = {"bound": "hbm", "capacity": cap, "frame_bytes": fb, "batch": batch,
                              "ring_state_gb": round((cap + 1) * fb / 1e9, 2),
                              "us_per_launch": round(ms * 1e3, 1), "achieved": round(gb / (ms * 1e-3), 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 4),
                              "algorithmic_bytes": int(gb * 1e9), "samples_per_sec": round(batch / (ms * 1e-3), 1),
                              "us_per_launch_repeated_batch": round(ms_rep * 1e3, 1),
                              "prioritized_sample_us": round(ms_s * 1e3, 1),
                              "priority_update_us": round(ms_u * 1e3, 1),
                              "sample_gather_fused_us": round(ms_sg * 1e3, 1),
                              "sample_gather_update_us": round(ms_all * 1e3, 1),
                              "sample_gather_update_note": "two launches: rlhip_ring_sample_gather_prioritized (draw inside the gather) + rlhip_sumtree_update",
                              "prioritized_samples_per_sec": round(batch / (ms_all * 1e-3), 1)}
    # batch in {32, 512, 4096} -- the small batches are the latency regime of the same kernels
    # (32 samples = 32 workgroup pairs: a handful of CUs busy; the time is the launch + one HBM round trip per frame pair)
    small = {}
    for b in (32, 512):
        idx_b, key_b, prio_b = tr.sample_prioritized(b, 11, 0)
        bufs_b = tr.gather(idx_b)
        cb = [1]

        def smp_b():
            rlhip._lib.call("rlhip_ring_sample_prioritized", C.byref(tr.rb), ops.ptr(tr.priorities), b, 11, cb[0],
                            ops.ptr(idx_b), ops.ptr(key_b), ops.ptr(prio_b), s)
            cb[0] += 1

        def g_b():
            rlhip._lib.call("rlhip_ring_gather", C.byref(tr.rb), ops.ptr(idx_b), b, ops.ptr(bufs_b[0]), ops.ptr(bufs_b[1]),
                            ops.ptr(bufs_b[2]), ops.ptr(bufs_b[3]), ops.ptr(bufs_b[4]), s)

        def u_b():
            rlhip._lib.call("rlhip_sumtree_update", ops.ptr(tr.priorities), cap, ops.ptr(key_b), ops.ptr(prio_b), b, s)

        def sg_b():
            rlhip._lib.call("rlhip_ring_sample_gather_prioritized", C.byref(tr.rb), ops.ptr(tr.priorities), b, 11, cb[0],
                            ops.ptr(idx_b), ops.ptr(key_b), ops.ptr(prio_b), ops.ptr(bufs_b[0]), ops.ptr(bufs_b[1]),
                            ops.ptr(bufs_b[2]), ops.ptr(bufs_b[3]), ops.ptr(bufs_b[4]), s)
            cb[0] += 1

        def all_b():
            sg_b()
            u_b()

        sync_b = torch.zeros(2, dtype=torch.int32, device="cuda")
        uk_b, up_b = key_b.clone(), prio_b.clone()  # the PREVIOUS batch's keys / priorities: written back inside the next launch

        def fused_b():  # round 6: write-back + draw + gather in ONE launch (<= 64 keys; otherwise the two calls behind the same entry)
            rlhip._lib.call("rlhip_ring_update_sample_gather_prioritized", C.byref(tr.rb), ops.ptr(tr.priorities), ops.ptr(uk_b),
                            ops.ptr(up_b), b, b, 11, cb[0], ops.ptr(idx_b), ops.ptr(key_b), ops.ptr(prio_b), ops.ptr(bufs_b[0]),
                            ops.ptr(bufs_b[1]), ops.ptr(bufs_b[2]), ops.ptr(bufs_b[3]), ops.ptr(bufs_b[4]), ops.ptr(sync_b), s)
            cb[0] += 1

        def fresh_b():
            smp_b()
            g_b()

        t_s = event_time_ms(smp_b, 20, lib, s)
        t_g = event_time_ms(fresh_b, 20, lib, s) - t_s
        t_u = event_time_ms(u_b, 20, lib, s)
        t_all = event_time_ms(all_b, 20, lib, s)
        fused_b()
        t_fused = event_time_ms(fused_b, 20, lib, s)
        gbb = 2 * (2 * fb + 9) * b / 1e9
        small[str(b)] = {"batch": b, "us_per_launch": round(t_g * 1e3, 1), "achieved": round(gbb / (t_g * 1e-3), 1),
                         "unit": "GB/s", "frac": round(gbb / (t_g * 1e-3) / HBM_PEAK_GBS, 4),
                         "prioritized_sample_us": round(t_s * 1e3, 1), "priority_update_us": round(t_u * 1e3, 1),
                         "sample_gather_update_us": round(t_all * 1e3, 1),
                         "update_sample_gather_one_call_us": round(t_fused * 1e3, 1),
                         "update_sample_gather_note": ("rlhip_ring_update_sample_gather_prioritized: ONE launch (<= 64 keys)" if b <= 64 else
                                                       "rlhip_ring_update_sample_gather_prioritized: > 64 keys -> the two launches behind the same entry point"),
                         "prioritized_samples_per_sec": round(b / (min(t_all, t_fused) * 1e-3), 1)}
        del idx_b, key_b, prio_b, bufs_b, uk_b, up_b, sync_b
    out["frame_gather_u8"]["small_batches"] = small
    del key, prio, keys
    del tr, bufs, idx
    torch.cuda.empty_cache()
    # same replay as single 84x84 frames (7.4 GB instead of 29.6 GB) with StackFrames applied at sample time:
    # 5 source frames read once, 2 x 4 frames written per sample
    f1 = 84 * 84
    tr1 = CircularArraySARTSTraces(capacity=cap, n_env=1, obs_dim=f1, dtype=torch.uint8)
    tr1.state.random_(1, 256)
    tr1.terminal.copy_((torch.rand(cap, 1, device="cuda") < 1 / 800).to(torch.uint8))
    tr1.rb.len_sa, tr1.rb.len_rt = cap + 1, cap
    idx1 = tr1.sample_indices(batch, seed=11, draw_ctr=0)
    outs = tr1.gather_stacked(idx1, 4)
    cnt = [1]

    def gs():
        rlhip._lib.call("rlhip_ring_sample_indices", C.byref(tr1.rb), batch, 11, cnt[0], ops.ptr(idx1), s)
        cnt[0] += 1
        rlhip._lib.call("rlhip_ring_gather_stacked", C.byref(tr1.rb), ops.ptr(idx1), batch, 4, ops.ptr(outs[0]),
                        ops.ptr(outs[1]), ops.ptr(outs[2]), ops.ptr(outs[3]), ops.ptr(outs[4]), s)

    def gi():
        rlhip._lib.call("rlhip_ring_sample_indices", C.byref(tr1.rb), batch, 11, cnt[0], ops.ptr(idx1), s)

    ms = event_time_ms(gs, 10, lib, s, SETTLE_S) - event_time_ms(gi, 10, lib, s)
    gb = (5 * f1 + 8 * f1 + 9 + 9) * batch / 1e9
    out["frame_gather_u8_stack_at_sample"] = {
        "bound": "hbm", "capacity": cap, "frame_bytes": f1, "n_stack": 4, "batch": batch,
        "ring_state_gb": round((cap + 1) * f1 / 1e9, 2), "us_per_launch": round(ms * 1e3, 1),
        "achieved": round(gb / (ms * 1e-3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(gb * 1e9),
        "samples_per_sec": round(batch / (ms * 1e-3), 1)}
    del tr1, outs, idx1
    torch.cuda.empty_cache()
    if hbm_only:  # tools/pmc_all.py: the HBM-bound legs alone, for the PMC traffic passes
        return out
    # BASELINE configs[1]: 4096-way CartPole + QBasedPolicy(DQN, 4->128->2), batch 512, 1 update per vec-step
    n = N_ENVS
    env = rlhip.CartPoleEnv(n, seed=5)
    net = rlhip.HipApproximator(4, 128, 2, seed=5)
    learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=100), batchsize=512, min_replay_history=n, seed=5)
    policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
    agent = rlhip.Agent(policy, rlhip.Trajectory(CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)))
    rlhip.run(agent, env, rlhip.StopAfterNSteps(20))
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    rlhip.run(agent, env, rlhip.StopAfterNSteps(steps))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["dqn_cartpole_4096env"] = {"env_steps_per_sec": round(n * steps / el, 1), "updates_per_sec": round(steps / el, 1),
                                   "ms_per_vec_step": round(el / steps * 1e3, 4), "batch": 512,
                                   "note": "per-step drop-in protocol (plan!/act!/push!/optimise! = one ccall each, eager)"}
    steps_f = 2000
    t0 = time.perf_counter()
    rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(steps_f))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["dqn_cartpole_4096env"]["fused_vec_step"] = {
        "env_steps_per_sec": round(n * steps_f / el, 1), "updates_per_sec": round(steps_f / el, 1),
        "ms_per_vec_step": round(el / steps_f * 1e3, 4),
        "note": "rlhip_dqn_vec_step_f32: ONE C-ABI call per vec-step; plan! + act! + push! in one launch (dqn_act.hip), then the whole optimise! (gradient; the workgroup that departs last reduces, clips, steps) in a second one -- 2 launches per vec-step up to 2048 samples, 3 beyond; bit-identical to the per-step protocol"}
    del agent, policy, learner, net, env
    # the same two loops from a COMPILED host (tests/abi_host/abi_host.c `time`: no PyTorch, no interpreter -- the position of the
    # Julia glue's ccalls): what the per-stage protocol costs when the host is not Python
    try:
        import subprocess

        host = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "abi_host", "abi_host.bin")
        r = subprocess.run([host, "time", "2000"], capture_output=True, text=True, timeout=120,
                           env={k: v for k, v in os.environ.items() if not k.startswith("PYTHON")})
        out["dqn_cartpole_4096env"]["compiled_host"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as exc:  # noqa: BLE001  (the C host is test infrastructure: its absence must not cost the bench line)
        out["dqn_cartpole_4096env"]["compiled_host"] = {"error": repr(exc)[:200]}
    # the other batch sizes BASELINE config 2 names (32, 4096), fused loop, 2- and 3-layer network
    by_batch = {}
    for layers in (2, 3):
        for bsz in (32, 4096):
            env = rlhip.CartPoleEnv(n, seed=5)
            net = rlhip.HipApproximator(4, 128, 2, seed=5, layers=layers)
            learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=100), batchsize=bsz, min_replay_history=n, seed=5)
            policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
            agent = rlhip.Agent(policy, rlhip.Trajectory(CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)))
            rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(30))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(1000))
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            by_batch[f"layers{layers}_batch{bsz}"] = {"env_steps_per_sec": round(n * 1000 / el, 1),
                                                      "us_per_vec_step": round(el / 1000 * 1e6, 2)}
            del agent, policy, learner, net, env
    out["dqn_cartpole_4096env"]["fused_vec_step_other_batches"] = by_batch
    # same config with the blog's 3-layer Q-network 4 -> 128 -> 128 -> 2, hidden layer on the bf16 MFMA (dqn3.hip)
    env = rlhip.CartPoleEnv(n, seed=5)
    net = rlhip.HipApproximator(4, 128, 2, seed=5, layers=3)
    learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=100), batchsize=512, min_replay_history=n, seed=5)
    policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
    agent = rlhip.Agent(policy, rlhip.Trajectory(CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)))
    rlhip.run(agent, env, rlhip.StopAfterNSteps(20))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rlhip.run(agent, env, rlhip.StopAfterNSteps(steps))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["dqn3_mfma_cartpole_4096env"] = {"env_steps_per_sec": round(n * steps / el, 1),
                                         "updates_per_sec": round(steps / el, 1),
                                         "ms_per_vec_step": round(el / steps * 1e3, 4), "batch": 512,
                                         "net": "4->128->128->2 relu, hidden layer bf16 MFMA, f32 master weights"}
    t0 = time.perf_counter()
    rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(steps_f))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["dqn3_mfma_cartpole_4096env"]["fused_vec_step"] = {
        "env_steps_per_sec": round(n * steps_f / el, 1), "updates_per_sec": round(steps_f / el, 1),
        "ms_per_vec_step": round(el / steps_f * 1e3, 4)}
    # the MFMA learner kernel alone at a PPO-sized batch (131072 samples): 4 hidden GEMMs per sample
    # (target fwd, online fwd, dH1, dW2) = 4 * 2 * 128 * 128 flop
    from rlhip import dqn as _dqn

    bm = 131072
    ws = _dqn.dqn3_workspace(4, 128, 2, bm)
    tr2 = agent.trajectory.container
    gbuf, lbuf = torch.empty_like(net.params), torch.empty(1, device="cuda")
    tn = learner.approximator

    def gk():
        _dqn.dqn3_grad(tr2, 128, 2, 0, net.params, net.packed, tn.target, tn.target_packed, bm, 0.99, 1.0, 1, 0,
                       workspace=ws, grad=gbuf, loss=lbuf)

    gk()
    ms = event_time_ms(gk, 10, lib, s, SETTLE_S)
    tf = 4 * 2 * 128 * 128 * bm / (ms * 1e-3) / 1e12
    out["dqn3_grad_mfma"] = {"bound": "mfma", "kernel": "dqn3_grad32_kernel<4,2,relu,OCC=2> (32-sample tiles, persistent workgroups) + d3_reduce_kernel", "batch": bm,
                             "us_per_launch": round(ms * 1e3, 1), "achieved": round(tf, 1), "peak": 2500.0,
                             "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                             "note": "MFMA flops only; first layer, heads, loss and all bias/W1/W3 gradients run on the "
                                     "VALU in the same kernel"}
    # the same learner step at hidden = 256 (4 -> 256 -> 256 -> 2): the streaming kernels of csrc/ppo3w.hip behind the same entry
    # points (ring gather, target forward -> TD target, online forward -> Huber -> dZ2, dH1 -> dW1, dW2, reduce)
    net256 = rlhip.HipApproximator(4, 256, 2, seed=5, layers=3)
    tn256 = rlhip.TargetNetwork(net256, sync_freq=100)
    ws256 = _dqn.dqn3_workspace(4, 256, 2, bm)
    g256 = torch.empty_like(net256.params)

    def gk256():
        _dqn.dqn3_grad(tr2, 256, 2, 0, net256.params, net256.packed, tn256.target, tn256.target_packed, bm, 0.99, 1.0, 1, 0,
                       workspace=ws256, grad=g256, loss=lbuf)

    gk256()
    ms256 = event_time_ms(gk256, 5, lib, s)
    tf256 = 4 * 2 * 256 * 256 * bm / (ms256 * 1e-3) / 1e12
    out["dqn3w_grad_mfma_hidden256"] = {"bound": "mfma", "kernel": "dqn3w_gather_kernel + ppo3w_fwd_kernel<target> + ppo3w_fwd_kernel<online> + "
                                        "ppo3w_bwd_kernel + ppo3w_dw2_kernel + ppo3w_reduce_kernel (csrc/ppo3w.hip)", "batch": bm,
                                        "us_per_launch": round(ms256 * 1e3, 1), "achieved": round(tf256, 1), "peak": 2500.0,
                                        "unit": "TFLOP/s", "frac": round(tf256 / 2500.0, 4)}
    del agent, policy, learner, net, env, ws, gbuf
    # BASELINE configs[2]: 4096-way PendulumEnv + PPOPolicy (GAE lambda = 0.95), T = 128, clip 0.1, 4 x 4
    # micro-batches of 131072, actor 3 -> 256 -> (mu, log sigma), critic 3 -> 256 -> 1.  fp32 VALU: a
    # 2-layer net has no hidden x hidden GEMM (K = 3, N <= 2), see DESIGN.md section 5.
    penv = rlhip.HipVecEnv("pendulum", n, seed=7)
    ppol = rlhip.PPOPolicy(penv, update_freq=128, hidden=HIDDEN, seed=7, clip_range=0.1)
    for _ in range(3):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    iters = 20
    t0 = time.perf_counter()
    for _ in range(iters):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["ppo_pendulum_4096env_T128"] = {"env_steps_per_sec": round(n * 128 * iters / el, 1),
                                        "updates_per_sec": round(ppol.n_updates_per_call() * iters / el, 1),
                                        "ms_per_iteration": round(el / iters * 1e3, 4), "dtype": "f32",
                                        "final_loss": float(ppol.losses[0])}
    del ppol, penv
    # the same config with THREE-layer actor / critic 3 -> 128 -> 128 -> {(mu, log sigma), 1}: the hidden x hidden layers
    # run on the bf16 MFMA (ppo3.hip, cfg.layers = 3) -- "actor/critic MLP in bf16 MFMA", f32 master weights
    penv = rlhip.HipVecEnv("pendulum", n, seed=7)
    ppol = rlhip.PPOPolicy(penv, update_freq=128, hidden=128, seed=7, clip_range=0.1, layers=3)
    for _ in range(3):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms_r = event_time_ms(ppol.rollout_, 3, lib, s)

    def upd_only():  # the rollout launch leaves adv / ret ready (fused GAE scan): the 16 optimiser steps alone
        ppol._adv_ready = True
        ppol.update_()

    ms_u = event_time_ms(upd_only, 3, lib, s)
    bm3 = n * 128 // ppol.cfg.n_microbatches
    mf = 3 * 2 * 128 * 128 * 2 * bm3  # forward + dH1 + dW2 GEMMs of both nets per micro-batch (the USEFUL MFMA flops)
    per_step_us = ms_u * 1e3 / ppol.n_updates_per_call()
    out["ppo3_mfma_pendulum_4096env_T128"] = {
        "env_steps_per_sec": round(n * 128 * iters / el, 1),
        "updates_per_sec": round(ppol.n_updates_per_call() * iters / el, 1),
        "ms_per_iteration": round(el / iters * 1e3, 4), "dtype": "bf16 MFMA hidden layers, f32 master weights / accumulate",
        "n_params": ppol.np, "rollout_us": round(ms_r * 1e3, 1), "update_us": round(ms_u * 1e3, 1),
        "per_microbatch_us": round(per_step_us, 1), "microbatch": bm3,
        "kernels": "ppo3_gradT_kernel<3,relu,gaussian> (register-chained tile, persistent workgroups, csrc/ppo3t_kernel.h) + "
                   "d3_apply_kernel (reduce + loss + norm + clip + Adam + bf16 re-pack, one launch)",
        "learner_mfma_tflops": round(mf / (per_step_us * 1e-6) / 1e12, 1),
        "final_loss": float(ppol.losses[0])}
    out["ppo3_grad_mfma"] = {"bound": "mfma", "kernel": "ppo3_gradT_kernel<3,relu,gaussian> + d3_apply_kernel", "batch": bm3,
                             "us_per_launch": round(per_step_us, 1),
                             "achieved": round(mf / (per_step_us * 1e-6) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(mf / (per_step_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                             "note": "useful GEMM flops (3 GEMMs x 2 nets) of one optimiser step over the WHOLE step incl. the "
                                     "optimiser tail; the tile issues 144 MFMAs per 96 useful (layer 2 in both operand roles, "
                                     "one MFMA transposition); layer 1, heads, loss and every elementwise pass run on the VALU "
                                     "in the same kernel and bound it (profiles/r02_ppo3_gradT.md)"}
    del ppol, penv
    # config 3 at the width SURVEY 8(d) names (256): 3 -> 256 -> 256 -> {(mu, log sigma), 1}, csrc/ppo3w.hip -- three
    # streaming kernels per net (forward + loss + dZ2 / dH1 -> dW1 / dW2) with one operand resident in registers each
    penv = rlhip.HipVecEnv("pendulum", n, seed=7)
    ppol = rlhip.PPOPolicy(penv, update_freq=128, hidden=256, seed=7, clip_range=0.1, layers=3)
    for _ in range(2):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    it256 = 5
    t0 = time.perf_counter()
    for _ in range(it256):
        ppol.rollout_()
        ppol.update_()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms_r = event_time_ms(ppol.rollout_, 3, lib, s)

    def upd_only256():
        ppol._adv_ready = True
        ppol.update_()

    ms_u = event_time_ms(upd_only256, 2, lib, s)
    mf = 3 * 2 * 256 * 256 * 2 * bm3
    per_step_us = ms_u * 1e3 / ppol.n_updates_per_call()
    out["ppo3w_mfma_pendulum_4096env_T128_hidden256"] = {
        "env_steps_per_sec": round(n * 128 * it256 / el, 1),
        "updates_per_sec": round(ppol.n_updates_per_call() * it256 / el, 1),
        "ms_per_iteration": round(el / it256 * 1e3, 4), "dtype": "bf16 MFMA hidden layers, f32 master weights / accumulate",
        "n_params": ppol.np, "rollout_us": round(ms_r * 1e3, 1), "update_us": round(ms_u * 1e3, 1),
        "per_microbatch_us": round(per_step_us, 1), "microbatch": bm3,
        "kernels": "ppo3w_build_rec_kernel once per update; per optimiser step: ppo3w_gather_rec_kernel; per net ppo3w_fwd_kernel + "
                   "ppo3w_bwd_kernel; ppo3w_dw2_kernel (both nets, one launch) (csrc/ppo3w.hip: one operand register-resident per "
                   "kernel, persistent 8-wave workgroups, one per CU); ppo3w_reduce_sumsq_kernel; ppo3w_adam_pack_kernel; rollout: "
                   "ppo3w_rollout_kernel",
        "learner_mfma_tflops": round(mf / (per_step_us * 1e-6) / 1e12, 1),
        "frac_of_bf16_peak": round(mf / (per_step_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
        "final_loss": float(ppol.losses[0]),
        "note": "a burst of two update calls behind an idle gap; this learner's speed depends on the box's clock limiter (sustained_clock in this "
                "line times 1.5 s of the same step with the chip's clock beside it).  Round 6 vs round 5 is a SAME-BOX A / B in "
                "profiles/r06_ppo3w.md section 4: 212 -> 202 us per step (dZ2 written once, as the MFMA fragment image); boxes differ by "
                "+-4 % for the same code"}
    return out


def roofline_hbm_side(torch, rlhip):
    """The rest of SURVEY 8(d)'s HBM list, each at a size that leaves the caches, algorithmic bytes per
    unit from SURVEY 8(d): Pendulum / MountainCar env-step at 2^24 envs (45 / 33 B per env-step), Adam (28 B / param) and
    Polyak (12 B / param) at 2^22 and 2^26 parameters, the replay push with the 2-frame max-pool at config 5's frame size
    (84 x 84 x 4 u8, 4096 transitions per launch: 2 screens read, 1 frame + 9 B written per transition), and the small
    (CartPole) transition gather, 82 B per sample x 2^20 samples out of a 2^20-transition ring."""
    from rlhip import ops
    from rlhip._lib import call
    from rlhip.ops import ptr, stream_ptr
    from rlhip.trajectory import CircularArraySARTSTraces

    lib, s = rlhip._lib.lib, stream_ptr()
    out = {}

    def carve(sizes_bytes, dtype):
        """arrays of one streaming launch carved from ONE allocation at offsets staggered by 4352 B, like the env arrays of a
        large vector env (rlhip/envs.py): separately allocated 2^k-byte arrays start a multiple of 64 MB apart and their
        streams then walk the HBM channels in lock-step"""
        stag, offs, o = 4352, [], 0
        for b in sizes_bytes:
            offs.append(o)
            o += (b + stag + 255) // 256 * 256
        buf = torch.empty(o, dtype=torch.uint8, device="cuda")
        return [buf[a:a + b].view(dtype) for a, b in zip(offs, sizes_bytes)], buf

    def entry(gb, ms, **kw):
        d = {"bound": "hbm", "us_per_launch": round(ms * 1e3, 2), "achieved": round(gb / (ms * 1e-3), 1), "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(gb * 1e9)}
        d.update(kw)
        return d

    # --- env-step of the other two classic-control envs (same kernel template as `roofline`, same protocol: random actions,
    # episodes de-synchronised before the timed launches, packed episode counters)
    for kind, na, nbytes, with_obs in (("pendulum", 3, 45, True), ("mountaincar", 3, 33, False)):
        n = 1 << 24
        env = rlhip.HipVecEnv(kind, n, seed=1, packed_episode=True)
        actions = torch.randint(0, na, (8, n), dtype=torch.int32, device="cuda")
        a_ptrs = [ptr(actions[k]) for k in range(8)]
        # (the observation planes: three more write streams; offset from the allocation's natural alignment like the env's own arrays)
        obs = carve([4352 * 5, 3 * n * 4], torch.float32)[0][1].view(3, n) if with_obs else None
        k = [0]

        def step():
            k[0] += 1
            call("rlhip_env_step", env.kind, 0, C.byref(env.cfg), C.byref(env._st), env.n, a_ptrs[k[0] & 7], 1, env.seed, 0,
                 None, ptr(obs) if with_obs else None, s)

        for _ in range(40):  # TLB; the 200-step episodes of 2^24 envs stay synchronised (time limit only) -- see note
            step()
        torch.cuda.synchronize()
        ms = event_time_ms(step, 20, lib, s, SETTLE_S)
        out[f"env_step_{kind}"] = entry(nbytes * n / 1e9, ms, n_envs=n, bytes_per_unit=nbytes,
                                        kernel=f"env_step_kernel<{kind},f32,EPL=4,non-temporal in-place arrays / ordinary stores for the write-only ones,packed>",
                                        env_steps_per_sec=round(n / (ms * 1e-3), 1),
                                        note=("state(env) = (cos, sin, thetadot) written by the same launch (obs_out); three Float64 trig "
                                              "evaluations per env-step: the VALU work (not the 45 bytes) bounds this launch" if with_obs else
                                              "state(env) IS the state arrays") + "; uniformly random discrete actions")
        del env, actions, obs
        torch.cuda.empty_cache()
    # --- Adam (Optimisers.Adam, 28 B / param) and Polyak (TargetNetwork soft sync, 12 B / param)
    for logn in (22, 26):
        n = 1 << logn
        (p, g, m, v), _keep = carve([4 * n] * 4, torch.float32)
        for t_ in (p, g, m):
            t_.normal_()
        v.uniform_(0.01, 1.0)
        bp = torch.tensor([0.9, 0.999], device="cuda")
        ops.adam_(p, g, m, v, bp)
        ms = event_time_ms(lambda: ops.adam_(p, g, m, v, bp), 20, lib, s, SETTLE_S)
        out[f"adam_2p{logn}"] = entry(28 * n / 1e9, ms, n_params=n, bytes_per_unit=28,
                                      kernel=("adam_vec4_kernel<non-temporal stores, one 16-byte chunk per lane>: ONE launch (the beta-power advance runs in the "
                                              "workgroup that departs last; csrc/optim.hip)" if n <= (1 << 23) else
                                              "adam_vec4_kernel<non-temporal stores, one 16-byte chunk per lane> + beta_pow_advance_kernel (above 2^23 "
                                              "parameters one call = two launches)"),
                                      note="2^22 parameters (117 MB per call) fit the 256 MB Infinity Cache: the 2^26 entry is the HBM one" if logn == 22 else "")
        ops.polyak_(p, g, 0.995)
        ms = event_time_ms(lambda: ops.polyak_(p, g, 0.995), 20, lib, s)
        out[f"polyak_2p{logn}"] = entry(12 * n / 1e9, ms, n_params=n, bytes_per_unit=12, kernel="polyak_vec4_kernel")
        del p, g, m, v, _keep
        torch.cuda.empty_cache()
    # --- replay push with the 2-frame max-pool (AtariEnv.act! fused into push!), config 5's frame size, 4096 envs
    fb, n_env = 84 * 84 * 4, 4096
    tr = CircularArraySARTSTraces(capacity=6, n_env=n_env, obs_dim=fb, dtype=torch.uint8)
    (s1, s2), _keep = carve([fb * n_env] * 2, torch.uint8)
    s1, s2 = s1.view(fb, n_env), s2.view(fb, n_env)
    s1.random_(0, 256)
    s2.random_(0, 256)
    a = torch.zeros(n_env, dtype=torch.int32, device="cuda")
    r = torch.zeros(n_env, dtype=torch.float32, device="cuda")
    t = torch.zeros(n_env, dtype=torch.uint8, device="cuda")
    tr.push_state_maxpool_(s1, s2)
    for _ in range(8):
        tr.push_transition_maxpool_(s1, s2, a, r, t)
    ms = event_time_ms(lambda: tr.push_transition_maxpool_(s1, s2, a, r, t), 10, lib, s, SETTLE_S)
    out["push_transition_maxpool"] = entry((3 * fb + 18) * n_env / 1e9, ms, frame_bytes=fb, transitions_per_launch=n_env,
                                           bytes_per_unit=3 * fb + 18, kernel="push_transition_maxpool_kernel (one launch)",
                                           transitions_per_sec=round(n_env / (ms * 1e-3), 1))
    del tr, s1, s2, _keep
    torch.cuda.empty_cache()
    # --- small-observation gather: CartPole transitions (ns = 4), 2^20 samples, at BOTH operating points: out of
    # a 256 x 4096 ring (67 MB of records: Infinity-Cache resident) and out of a 16384 x 4096 ring (4.3 GB: every record comes from HBM)
    n_env, batch = 4096, 1 << 20
    for key, cap in (("gather_small", 256), ("gather_small_hbm", 16384)):
        tr = CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=4)
        tr.records.normal_()  # every word of every 64-byte record: s, s_next and (overwritten below) a, r, t
        tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
        idx = tr.sample_indices(batch, seed=11, draw_ctr=0)
        bufs = tr.gather(idx)
        c = [1]

        def smp():
            call("rlhip_ring_sample_indices", C.byref(tr.rb), batch, 11, c[0], ptr(idx), s)
            c[0] += 1

        def sg():
            smp()
            call("rlhip_ring_gather", C.byref(tr.rb), ptr(idx), batch, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(bufs[3]),
                 ptr(bufs[4]), s)

        ms = event_time_ms(sg, 10, lib, s, SETTLE_S) - event_time_ms(smp, 10, lib, s)
        ring_mb = round((cap + 1) * n_env * 64 / 1e6, 1)
        if key == "gather_small":
            out[key] = entry(82 * batch / 1e9, ms, batch=batch, bytes_per_unit=82, ring_transitions=cap * n_env, ring_mb=ring_mb,
                             kernel="gather_rec_kernel<4>", samples_per_sec=round(batch / (ms * 1e-3), 1),
                             requests_per_sec_g=round(1.02 * batch / (ms * 1e-3) / 1e9, 1),
                             note="the ring stores one 64-byte record {s[4], a, r, t, s'[4]} per (state slot, env) -- the whole transition in ONE "
                                  "cache line = one fabric request per sample (TCC_EA0_RDREQ 1.02 per sample: profiles/r05_pmc.md).  This 67 MB ring is "
                                  "Infinity-Cache resident: `frac` divides by the HBM peak only to keep one unit across the table -- the launch is bound by "
                                  "the L2s' 64-byte REQUEST rate (~45 G requests / s), not by HBM bytes; the HBM operating point is `gather_small_hbm`.  "
                                  "32-byte records (two lines per sample) took 43.5 us for this launch, round 4 (five lines) 79 us, rounds 1 - 3 (eleven) 164 - 168 us")
            out[key]["bound"] = "infinity-cache request rate"
        else:
            out[key] = entry(82 * batch / 1e9, ms, batch=batch, bytes_per_unit=82, ring_transitions=cap * n_env, ring_mb=ring_mb,
                             kernel="gather_rec_kernel<4>", samples_per_sec=round(batch / (ms * 1e-3), 1),
                             requests_per_sec_g=round(1.02 * batch / (ms * 1e-3) / 1e9, 1),
                             note="the same launch out of a 4.3 GB ring: every 64-byte record is an HBM read of one line (a random 64-byte read "
                                  "opens a DRAM page for one burst), the batch is written streaming")
        del tr, idx, bufs
        torch.cuda.empty_cache()
    return out


def kernel_breakdown(torch, rlhip, pol, env):
    """Device time of the three enqueue units of one step (HIP events on the launch stream).  Each unit is
    ONE C-ABI call, so the numbers are not host-paced: rollout (1 launch), GAE (1 launch), update
    (pack + n_epochs x n_microbatches x {grad kernel, reduce+clip+Adam kernel}).  The rollout launch includes the
    fused GAE + returns scan; gae_returns_us is the stand-alone scan (used by the per-step protocol)."""
    from rlhip.ops import stream_ptr

    s = stream_ptr()
    lib = rlhip._lib.lib
    out = {}
    saved = [t.clone() for t in (pol.params, pol.m, pol.v, pol.beta_pow)]
    pol.gae_()  # first launch of the stand-alone scan in this process (the timed workload uses the fused one)
    out["rollout_T32_us"] = round(event_time_ms(pol.rollout_, 5, lib, s) * 1e3, 2)
    out["gae_returns_us"] = round(event_time_ms(pol.gae_, 5, lib, s) * 1e3, 2)
    n_upd = pol.n_updates_per_call()
    def upd_only():  # the rollout leaves adv / ret ready (fused GAE scan): time the optimiser steps alone
        pol._adv_ready = True
        pol.update_()

    upd_ms = event_time_ms(upd_only, 5, lib, s)
    out["update_us"] = round(upd_ms * 1e3, 2)
    out["per_microbatch_us"] = round(upd_ms * 1e3 / n_upd, 2)
    for t, sv in zip((pol.params, pol.m, pol.v, pol.beta_pow), saved):
        t.copy_(sv)
    # f32 flops of one micro-batch: forward 2*h*((ns+nout_a)+(ns+1)) per sample, x3 with backward
    bm = (env.n * pol.T) // pol.cfg.n_microbatches
    flops = 3 * 2 * pol.cfg.hidden * ((env.odim + pol.na) + (env.odim + 1)) * bm
    out["learner_tflops_f32"] = round(flops / (out["per_microbatch_us"] * 1e-6) / 1e12, 2)
    out["learner_f32_peak_tflops"] = 157.3
    return out


def sustained_clock_probe(torch, rlhip, pol, env, seconds=1.5):  # noqa: C901
    """AFTER the timed region (nothing of the protocol in front of it changes): the chip's clock and socket power while (a) the headline step and
    (b) the 256-wide PPO optimiser step run back to back for `seconds` each -- hwmon freq1_input / power1_input of THIS device (matched by PCI
    address), sampled every 20 ms from a thread; steady state = the second half of each window.  Why it is in the line: on two of three
    boxes of round 6 the 256-wide learner ran under a firmware limiter (2.11 - 2.35 of 2.4 GHz at 1.05 - 1.16 kW, cap 1.4 kW) whose clock depends
    on the kernel mix (profiles/r06_ppo3w.md section 5; tools/power_probe.py is the stand-alone form), and a 16-ms burst after an idle gap
    reads 3 - 5 % slower than the same work sustained: a learner number is only comparable together with the clock it ran at."""
    import glob
    import threading

    pr = torch.cuda.get_device_properties(torch.cuda.current_device())
    want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
    files = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.basename(os.path.realpath(d)).startswith(want):
            for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for name in ("freq1_input", "power1_input", "power1_average", "power1_cap", "temp2_input"):
                    f = os.path.join(h, name)
                    if os.path.exists(f):
                        files.setdefault(name, f)
    if "freq1_input" not in files:
        return {"error": f"no hwmon freq1_input for PCI device {want}*"}

    def rd(name):
        try:
            with open(files[name]) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError, KeyError):
            return float("nan")

    def window(burst, per_burst):
        samples, marks, stop = [], [], [False]

        def sampler():
            while not stop[0]:
                samples.append((time.perf_counter(), rd("freq1_input"), rd("power1_input" if "power1_input" in files else "power1_average"),
                                rd("temp2_input")))
                time.sleep(0.02)

        th = threading.Thread(target=sampler)
        th.start()
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            burst()
            torch.cuda.synchronize()
            marks.append((t0, (time.perf_counter() - t0) / per_burst * 1e6))
        stop[0] = True
        th.join()
        t_a = marks[len(marks) // 2][0]
        ss = [x for x in samples if x[0] >= t_a]
        us = sorted(u for t, u in marks if t >= t_a)

        def mean(i, scale):
            v = [x[i] for x in ss if x[i] == x[i]]
            return round(sum(v) / len(v) / scale, 1) if v else None

        return {"sclk_mhz": mean(1, 1e6), "socket_w": mean(2, 1e6), "junction_c": mean(3, 1e3), "us_median": round(us[len(us) // 2], 1),
                "us_min": round(us[0], 1), "bursts": len(us)}

    out = {"power_cap_w": round(rd("power1_cap") / 1e6, 1) if "power1_cap" in files else None, "window_s": seconds,
           "note": "after the timed region; steady state = second half of each window; sclk tops out at 2400 MHz"}

    def head():
        for _ in range(20):
            pol.rollout_()
            pol.update_()

    out["headline_step"] = window(head, 20)
    penv = rlhip.HipVecEnv("pendulum", 4096, seed=7)
    ppol = rlhip.PPOPolicy(penv, update_freq=128, hidden=256, seed=7, clip_range=0.1, layers=3)
    ppol.rollout_()
    ppol.update_()
    torch.cuda.synchronize()
    nup = ppol.n_updates_per_call()

    def upd():
        for _ in range(5):
            ppol._adv_ready = True
            ppol.update_()

    out["ppo3w_optimiser_step"] = window(upd, 5 * nup)
    # the same step with each of the backward kernel's two LDS copies forced (csrc/ppo3w.hip RLHIP_W3_DZF_PAD: bit-identical kernels; the padded copy
    # is 3 us per launch faster and, on boxes with an active clock limiter, costs 6 - 8 % of the clock), then 3 s in the default mode (the host
    # picks by the device's clock reading): which kernel wins on THIS box, and what the default picked
    try:
        import ctypes as C

        from rlhip._lib import lib as _l

        fn = _l.rlhip_debug_w3_dzf_pad_info
        fn.restype, fn.argtypes = C.c_int32, [C.c_int32, C.POINTER(C.c_double)]
        info = (C.c_double * 6)()
        fn(-1, info)
        mode0 = int(info[0])
        out["ppo3w_optimiser_step"]["lds_copy_of_the_backward_kernel"] = {"mode": mode0, "padded": int(info[1])}
        try:
            fn(0, info)
            out["ppo3w_optimiser_step_unpadded_lds_copy"] = window(upd, 5 * nup)
            fn(1, info)
            out["ppo3w_optimiser_step_padded_lds_copy"] = window(upd, 5 * nup)
            fn(2, info)
            seconds = 3.0
            w = window(upd, 5 * nup)
            fn(-1, info)
            w.update({"picked_padded": int(info[1]), "switches": int(info[4]), "sensor": bool(info[5]), "last_reading_mhz": round(info[2], 1),
                      "top_mhz": round(info[3], 1)})
            out["ppo3w_optimiser_step_auto_3s"] = w
        finally:
            fn(mode0, info)
    except Exception as exc:  # noqa: BLE001
        out["ppo3w_optimiser_step_padded_lds_copy"] = {"error": repr(exc)}
    return out


def _oracle_ppo_iterations(oracle, np, n, budget_s, max_iters):
    env = oracle.VecEnv("cartpole", n, seed=1)
    cfg = oracle.ppo_default(hidden=HIDDEN)
    params = np.concatenate([oracle.mlp2_init(4, HIDDEN, 2, 1, 0), oracle.mlp2_init(4, HIDDEN, 1, 1, 1)])
    assert params.size == oracle.ppo_nparams(0, cfg)
    m, v = np.zeros_like(params), np.zeros_like(params)
    traj = oracle.PPOTraj(0, n, T_ROLLOUT)
    opt_step, iters = 0, 0
    t0 = time.perf_counter()
    while True:
        oracle.ppo_rollout(env, T_ROLLOUT, cfg, params, traj, iters * T_ROLLOUT)
        oracle.ppo_gae(cfg, traj)
        opt_step, _ = oracle.ppo_update(0, cfg, traj, params, m, v, opt_step, 1, iters)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= max_iters:
            break
    return iters, el


def cpu_baseline(budget_s=8.0):
    """The CPU oracle's PPO iteration (same algorithm, same hyper-parameters, "port" of the reference) on a
    bounded sample: all host cores (the -fopenmp build of the same C sources, env instances / samples split over
    threads) on the full 4096-env workload, and one core on a 256-env slice."""
    import numpy as np

    import oracle

    it1, el1 = _oracle_ppo_iterations(oracle, np, 256, budget_s * 0.6, 200)
    single = {"value": round(256 * T_ROLLOUT * it1 / el1, 1), "updates_per_sec": round(16 * it1 / el1, 2), "cores": 1,
              "sample": f"{it1} iterations of 256 envs x T={T_ROLLOUT} in {el1:.1f} s"}
    threads = oracle.use_all_cores(True)
    try:
        _oracle_ppo_iterations(oracle, np, N_ENVS, 0.0, 1)  # warm the thread pool
        it, el = _oracle_ppo_iterations(oracle, np, N_ENVS, budget_s, 200)
    finally:
        oracle.use_all_cores(False)
    return {"value": round(N_ENVS * T_ROLLOUT * it / el, 1), "unit": "env-steps/s", "cores": threads, "kind": "port",
            "updates_per_sec": round(16 * it / el, 2),
            "sample": f"{it} full PPO iterations of {N_ENVS} envs x T={T_ROLLOUT} (same net / epochs / micro-batches) in "
                      f"{el:.1f} s, oracle C restatement, gcc -O2 -fopenmp, {threads} threads = the CPUs this container may "
                      f"use (cgroup quota) of the host's {os.cpu_count()}",
            "single_thread": single}


def allreduce_report(torch, pol, world):
    """SURVEY 8(e) "what to report", N > 1 only, outside the timed region, every rank in lock-step: latency of the
    gradient exchange at the real gradient size (library all-reduce and, when it is active, the one-shot peer-to-peer
    kernel), and the library's bus bandwidth over 4 KB .. 256 MB (busbw = 2 (N - 1) / N x bytes / time)."""
    import torch.distributed as dist

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / iters], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = {"world": world, "gradient_bytes": int(pol.np) * 4}
    g = torch.zeros(int(pol.np), dtype=torch.float32, device="cuda")
    out["library_us_at_gradient_size"] = round(timed(lambda: dist.all_reduce(g), 50), 2)  # torch.distributed, for reference
    comm = getattr(pol, "_hipcomm", None)
    if comm is not None:
        d = comm.info()
        out["p2p"] = {"active": bool(d.p2p_active), "why": d.why.decode(), "rccl_behind_abi": bool(d.rccl_active),
                      "rccl_path": d.rccl_path.decode()}
        if comm.ok:  # what the product's optimiser step pays: rlhip_allreduce_grads on the compute stream
            out["abi_us_at_gradient_size"] = round(timed(lambda: comm.all_reduce_(g), 50), 2)
            out["abi_transport_at_gradient_size"] = "p2p kernel" if d.p2p_active else "ncclAllReduce"
        if d.p2p_active:
            out["p2p_us_at_gradient_size"] = out["abi_us_at_gradient_size"]
        if d.rccl_active:  # vectors beyond the exchange buffer take ncclAllReduce behind the same entry point
            big = torch.zeros(max(int(d.cap) + 1, 1 << 20), dtype=torch.float32, device="cuda")
            us = timed(lambda: comm.all_reduce_(big), 20)
            out["abi_rccl"] = {"bytes": big.numel() * 4, "us": round(us, 2),
                               "busbw_gbs": round(2.0 * (world - 1) / world * big.numel() * 4 / us / 1e3, 2)}
            del big
    SIZES = (4 << 10, 64 << 10, 1 << 20, 16 << 20, 256 << 20)
    sweep = []
    for nbytes in SIZES:
        x = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
        us = timed(lambda: dist.all_reduce(x), 20 if nbytes <= (1 << 20) else 5)
        sweep.append({"bytes": nbytes, "us": round(us, 2),
                      "busbw_gbs": round(2.0 * (world - 1) / world * nbytes / us / 1e3, 2)})
        del x
    out["library_sweep"] = sweep  # torch.distributed (RCCL through PyTorch), for reference
    # the same sweep through the PRODUCT's collective, rlhip_allreduce_grads (csrc/comm.hip), on a communicator whose
    # exchange buffer takes vectors up to 16 MB: the one-shot peer-to-peer kernel up to there (every rank reads every
    # peer's buffer: latency-optimal, not bandwidth-optimal), ncclAllReduce of the dlopen'ed RCCL beyond
    try:
        from rlhip.dist import HipComm

        big_comm = HipComm.create(pol.process_group, (16 << 20) // 4, g.device)
        bd = big_comm.info()
        abi = []
        for nbytes in SIZES:
            via_p2p = bool(bd.p2p_active) and nbytes // 4 <= int(bd.cap)
            if not via_p2p and not bd.rccl_active:  # e.g. a gloo group (no RCCL behind the ABI): nothing carries this size
                abi.append({"bytes": nbytes, "transport": "none (no RCCL communicator, beyond the exchange buffer)"})
                continue
            x = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
            us = timed(lambda: big_comm.all_reduce_(x), 20 if nbytes <= (1 << 20) else 5)
            abi.append({"bytes": nbytes, "us": round(us, 2), "transport": "p2p kernel" if via_p2p else "ncclAllReduce",
                        "busbw_gbs": round(2.0 * (world - 1) / world * nbytes / us / 1e3, 2)})
            del x
        out["abi_sweep"] = abi
        out["abi_sweep_p2p"] = {"active": bool(bd.p2p_active), "why": bd.why.decode(), "cap_bytes": int(bd.cap) * 4}
        if bd.rccl_active and bd.p2p_active:  # both transports at the real gradient size (the small communicator has the p2p one)
            os.environ["RLHIP_NO_P2P"] = "1"
            try:
                rc_comm = HipComm.create(pol.process_group, int(pol.np), g.device)
                out["abi_us_at_gradient_size_rccl"] = round(timed(lambda: rc_comm.all_reduce_(g), 50), 2)
                rc_comm.close()
            finally:
                del os.environ["RLHIP_NO_P2P"]
        big_comm.failed() and out.setdefault("abi_sweep_timeout", True)
        big_comm.close()
    except Exception as exc:  # noqa: BLE001  -- the report must not cost the bench line
        out["abi_sweep"] = {"error": repr(exc)}
    out["bounds"] = "ring: one xGMI link (~153 GB/s); direct reduce-scatter + all-gather over 7 links: ~1.07 TB/s egress per GPU"
    return out


def preflight(torch, rlhip, rank, local_rank, world, pg, backend):
    """`bench.py --gpus N --preflight`: everything the first multi-GPU run depends on, reported per rank in ONE JSON line (rank 0
    prints it) so that a plumbing failure is diagnosable from one log -- and exit status 0 either way (it is a report).
    Per rank: device, hipDeviceCanAccessPeer towards every peer's device, an IPC round trip with every peer (rlhip_p2p_export /
    import / probe of a known word / close -- each peer separately, so one bad link names itself), then the product's own set-up
    (rlhip_comm_init -> rlhip_p2p_setup: mapping, exact self-test, cross-rank agreement) with rlhip_comm_info().why, and one
    gradient-sized exchange through rlhip_allreduce_grads compared with the rank-order sum computed on the host."""
    import ctypes as C

    from rlhip import _lib
    from rlhip._lib import call

    rep = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "backend": backend,
           "device_name": torch.cuda.get_device_name(), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if world == 1:
        rep["note"] = "world = 1: nothing to exchange"
        print(json.dumps({"preflight": [rep]}), flush=True)
        return
    import torch.distributed as dist

    def gather(x):
        out = [None] * world
        dist.all_gather_object(out, x, group=pg)
        return out

    devices = gather(rep["device"])
    rep["can_access_peer"] = {str(p): (True if devices[p] == rep["device"] else bool(_lib.lib.rlhip_p2p_can_access(devices[p])))
                              for p in range(world) if p != rank}
    # one IPC round trip per peer with the building blocks of csrc/p2p.hip
    buf, hb = C.c_void_p(), (C.c_uint8 * 64)()
    ipc = {}
    try:
        call("rlhip_p2p_alloc", 4096, C.byref(buf))
        word = torch.tensor([0x5EED0000 + rank], dtype=torch.int32)
        call("rlhip_memcpy_h2d", buf, C.c_void_p(word.data_ptr()), 4, None)
        call("rlhip_stream_sync", None)
        call("rlhip_p2p_export", buf, hb)
        mine = bytes(hb)
    except _lib.RLHipError as exc:
        mine = None
        ipc["export"] = f"FAILED: {exc}"
    handles = gather(mine)
    for p in range(world):
        if p == rank:
            continue
        if handles[p] is None:
            ipc[str(p)] = "peer could not export"
            continue
        q = C.c_void_p()
        try:
            call("rlhip_p2p_import", (C.c_uint8 * 64).from_buffer_copy(handles[p]), C.byref(q))
            val = C.c_uint32(0)
            call("rlhip_p2p_probe", q, 0, C.byref(val))
            ipc[str(p)] = "ok" if val.value == 0x5EED0000 + p else f"mapped, but read {val.value:#x} instead of {0x5EED0000 + p:#x}"
            call("rlhip_p2p_close", q)
        except _lib.RLHipError as exc:
            ipc[str(p)] = f"FAILED: {exc}"
    rep["ipc_round_trip"] = ipc
    dist.barrier(group=pg)  # nobody frees a buffer a peer still has mapped
    if buf.value:
        call("rlhip_p2p_free", buf)
    # the product's set-up and one exchange
    try:
        from rlhip.dist import HipComm

        n = 3331
        comm = HipComm.create(pg, n, torch.device("cuda"))
        d = comm.info()
        rep["comm"] = {"p2p_active": bool(d.p2p_active), "why": d.why.decode(), "rccl_behind_abi": bool(d.rccl_active),
                       "rccl_path": d.rccl_path.decode(), "setup_error": comm.setup_error, "transport": comm.transport()}
        x = (torch.arange(n, dtype=torch.float32) % 251 + rank).cuda()  # integers: any summation order gives the same bits
        want = sum((torch.arange(n, dtype=torch.float32) % 251 + r) for r in range(world))
        if comm.ok:
            comm.all_reduce_(x)
        else:
            dist.all_reduce(x, group=pg)
        torch.cuda.synchronize()
        rep["exchange"] = {"via": "rlhip_allreduce_grads" if comm.ok else "torch.distributed (no transport behind the ABI)",
                           "correct": bool(torch.equal(x.cpu(), want)), "timeout": bool(comm.failed())}
        comm.close()
    except Exception as exc:  # noqa: BLE001 -- a report, not a run
        rep["comm"] = {"error": repr(exc)}
    reports = gather(rep)
    if rank == 0:
        ok = all(r.get("exchange", {}).get("correct") and not r.get("exchange", {}).get("timeout") for r in reports)
        print(json.dumps({"preflight": reports, "world": world, "exchange_ok_on_every_rank": ok,
                          "p2p_active_on_every_rank": all(r.get("comm", {}).get("p2p_active") for r in reports)}), flush=True)
    dist.barrier(group=pg)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / breakdown legs")
    ap.add_argument("--preflight", action="store_true",
                    help="N > 1 plumbing report (< 10 s of GPU work, no timed steps): peer-access matrix, per-peer IPC mapping, the "
                         "peer-to-peer self-test verdict and why, one exchange through the ABI checked on the host; exits 0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on a free
        # loopback port, same arguments (the ranks see WORLD_SIZE and take the branch below).  exec: the launcher's exit
        # status, stdout (rank 0's JSON line) and signals are this process's.
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch one rank per GPU (or run `python bench.py "
                         f"--gpus N` bare: it launches torch.distributed.run itself)")
    # test hooks (not used by the driver): all ranks on device 0 / gloo instead of RCCL, to exercise the
    # N > 1 code path on a single-GPU box
    if os.environ.get("RLHIP_BENCH_SINGLE_DEVICE", "0") == "1":
        local_rank = 0
    backend = os.environ.get("RLHIP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    import rlhip

    if args.preflight:
        preflight(torch, rlhip, rank, local_rank, world, pg, backend)
        return

    env = rlhip.HipVecEnv("cartpole", N_ENVS, seed=123, env_id_base=rank * N_ENVS)
    pol = rlhip.PPOPolicy(env, update_freq=T_ROLLOUT, hidden=HIDDEN, seed=123, process_group=pg)

    # Launch mode.  Default: eager enqueue (3 C-ABI calls per step at N = 1; grad / all-reduce / clip+Adam per
    # micro-batch at N > 1) -- the host runs ahead of the GPU, so the step is GPU-bound either way (measured:
    # 0.6556 ms graph vs 0.6581 ms eager at N = 1).  RLHIP_BENCH_GRAPH=1 replays one captured HIP graph per step
    # instead (device-resident counters; covered by tests/test_gpu_learners.py).
    mode = "eager"
    if os.environ.get("RLHIP_BENCH_GRAPH", "0") == "1":
        pol.capture_graph_(warmup=2)
        mode = "hip_graph"

    def step():
        if mode == "hip_graph":
            pol.replay_()
        else:
            pol.rollout_()
            pol.update_()

    def sync():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    # The measurement legs that do not touch this policy (CPU baseline first, then the roofline kernels) run BEFORE the timed
    # region: the step is ~0.4 ms of back-to-back 5 - 60 us launches, and after `--warmup 5` alone the device is still ramping
    # its clocks (measured: the 20-step form read 4 - 5 % slower than the 200-step form of the same build).  The timed region
    # itself is unchanged: W untimed steps, barrier + synchronize, exactly K steps, barrier + synchronize.
    # RLHIP_BENCH_EXTRAS_FIRST=0 restores the old order (legs after the timed region) for an A / B.
    extras = {}
    extras_first = os.environ.get("RLHIP_BENCH_EXTRAS_FIRST", "1") == "1"
    want_extras = rank == 0 and world == 1 and not args.no_extras

    def run_extras():
        # order: the idle GPU (CPU baseline) first, the HBM-bound kernels next, the learner legs (whole PPO / DQN iterations,
        # the kind of work the timed steps are) last -- tools/step_preheat.py: a 20-step block reads 0.416 ms / step after idle
        # or after HBM streaming, 0.403 - 0.407 right after learner iterations, 0.398 in steady state
        extras["cpu_baseline"] = cpu_baseline()
        side = roofline_hbm_side(torch, rlhip)
        extras["roofline"] = roofline_env_step(torch, rlhip)
        extras["roofline_extra"] = roofline_extras(torch, rlhip)
        extras["roofline_extra"].update(side)
        attach_traffic(extras["roofline_extra"])

    if want_extras and extras_first:
        run_extras()

    def timed(n_warm, n_steps):
        """the contract's region: n_warm untimed steps, barrier + synchronize, EXACTLY n_steps steps, barrier + synchronize; MAX over ranks"""
        for _ in range(n_warm):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist

            tmax = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    # PROTOCOL (frozen in round 6): BOTH forms are measured in this process and printed in the same line.
    #   1. `ms_per_step_no_preheat` / `value_no_preheat`: the contract taken literally -- W warm-up steps, K timed steps, nothing of
    #      this workload in front of them (the RLHIP_BENCH_PREHEAT_STEPS=0 form of round 5).
    #   2. `ms_per_step` / `value`: the same region again after the workload has run `preheat_steps` = 60 untimed steps in total (the
    #      W + K steps of measurement 1 count towards the 60): the round-5 form, comparable with BENCH_r05.  The 0.4-ms step of ~33
    #      back-to-back 5 - 50 us launches needs ~45 steps (~18 ms) before it runs at its steady rate (tools/step_preheat.py;
    #      profiles/r04_rollout.md section 3), so `--steps 20 --warmup 5` alone measures the ramp.
    # Nothing else runs between the two; no leg was added, moved or removed in front of them since round 5.
    preheat = int(os.environ.get("RLHIP_BENCH_PREHEAT_STEPS", "60"))
    elapsed_np = timed(args.warmup, args.steps)
    if preheat > 0:
        for _ in range(max(0, preheat - args.warmup - args.steps)):
            step()
        elapsed = timed(args.warmup, args.steps)
    else:
        elapsed = elapsed_np

    env_steps = world * N_ENVS * T_ROLLOUT * args.steps
    updates = pol.n_updates_per_call() * args.steps
    result = {
        "metric": "env_steps_per_sec",
        "value": round(env_steps / elapsed, 1),
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_no_preheat": round(elapsed_np / args.steps * 1e3, 4),
        "value_no_preheat": round(world * N_ENVS * T_ROLLOUT * args.steps / elapsed_np, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "updates_per_sec": round(updates / elapsed, 1),
        "config": {"workload": "ppo_cartpole_4096env_per_gpu (BASELINE.json configs[3] per GPU)",
                   "n_envs_per_gpu": N_ENVS, "rollout_T": T_ROLLOUT,
                   "actor": f"4->{HIDDEN}->2 relu", "critic": f"4->{HIDDEN}->1 relu", "n_params": pol.np,
                   "n_epochs": pol.cfg.n_epochs, "n_microbatches": pol.cfg.n_microbatches,
                   "microbatch": (N_ENVS * T_ROLLOUT) // pol.cfg.n_microbatches,
                   "parallelism": f"env-shards x{world}, flat-gradient all-reduce per optimiser step (see gradient_allreduce)" if world > 1
                   else "single GPU"},
        "launch_mode": mode,
        "preheat_steps": preheat,
        "gradient_allreduce": ("none (single GPU)" if world == 1 else
                               (pol._hipcomm.transport() if getattr(pol, "_hipcomm", None) is not None else "not initialised")),
        "final_loss": float(pol.losses[0]),
        "mean_episode_len_last_rollout": round(
            (N_ENVS * T_ROLLOUT) / max(1.0, float(pol.trajectory.terminal.sum())), 2),
    }
    if want_extras:
        result["kernels"] = kernel_breakdown(torch, rlhip, pol, env)
        # (under rocprofv3 -- ROCPROF_* variables / its tool library in LD_PRELOAD -- the leg is skipped: its ~25 000 back-to-back launches would be
        # most of the kernel trace of a command whose profile is meant to show the legs and the timed steps)
        profiled = any("rocprof" in k.lower() or "rocprof" in v.lower() for k, v in os.environ.items())
        if profiled:
            result["sustained_clock"] = {"skipped": "running under rocprofv3"}
        elif os.environ.get("RLHIP_BENCH_CLOCK_PROBE", "1") == "1":
            try:  # a sensor that cannot be read must not cost the bench line
                result["sustained_clock"] = sustained_clock_probe(torch, rlhip, pol, env)
            except Exception as exc:  # noqa: BLE001
                result["sustained_clock"] = {"error": repr(exc)}
        if not extras_first:
            run_extras()
        result["roofline"] = extras["roofline"]
        result["roofline_extra"] = extras["roofline_extra"]
        result["cpu_baseline"] = extras["cpu_baseline"]
        result["legs_order"] = (f"cpu_baseline, HBM rooflines, learner rooflines, [warmup, timed steps] -> ms_per_step_no_preheat, pre-heat up to {preheat} "
                                f"untimed steps in total, [warmup, timed steps] -> ms_per_step, kernel breakdown, sustained clock probe"
                                if extras_first else "[warmup, timed steps], kernel breakdown, cpu_baseline, rooflines")
    if world > 1 and not args.no_extras:
        try:  # collective on every rank; a local failure must not cost the bench line
            result["allreduce"] = allreduce_report(torch, pol, world)
        except Exception as exc:  # noqa: BLE001
            result["allreduce"] = {"error": repr(exc)}
    if getattr(pol, "_hipcomm", None) is not None:
        # must be false: a peer that never arrived poisons the step with NaN (csrc/comm.hip) -- also visible in final_loss
        result["p2p_timeouts"] = bool(pol._hipcomm.failed())
    if world > 1:
        # the sharded learner's invariant, checked on the parameters the timed steps produced: every replica applied the
        # same reduced gradient in the same order, so the parameter vectors are bit-identical (MIN / MAX all-reduce of a
        # checksum; outside the timed region)
        try:
            from rlhip.dist import params_checksum_equal

            result["replicas_bit_identical"] = bool(params_checksum_equal(pol.params, pg))
        except Exception as exc:  # noqa: BLE001
            result["replicas_bit_identical"] = repr(exc)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist

        if getattr(pol, "_hipcomm", None) is not None:
            pol._hipcomm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
