"""world_size = 2 `gloo` tests (CPU) of the multi-GPU path's host logic: env sharding with global Philox
ids, all-reduce(mean) of the flat gradient BEFORE the global-norm clip, replicas staying bit-identical,
max-over-ranks timing.  The per-shard compute in these tests is done by the oracle (the checker) because
there is no GPU here; on the GPU box the same host functions wrap the HIP kernels (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "reinforcementlearning.jl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    from rlhip import dist as rdist

    r, lr, w, group = rdist.init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    n_per, T, seed = 64, 8, 21
    base, n = rdist.env_shard(rank, n_per)
    cfg = oracle.ppo_default(hidden=32, n_microbatches=1, n_epochs=1)
    params = np.concatenate([oracle.mlp2_init(4, 32, 2, seed, 0), oracle.mlp2_init(4, 32, 1, seed, 1)])
    env = oracle.VecEnv("cartpole", n, seed=seed, env_id_base=base)
    traj = oracle.PPOTraj(0, n, T)
    oracle.ppo_rollout(env, T, cfg, params, traj, 0)
    oracle.ppo_gae(cfg, traj)
    # this rank's micro-batch gradient (whole shard, identity order to make the 2-rank == 1-rank check exact)
    obs = traj.obs[:T].transpose(1, 0, 2).reshape(4, T * n)
    g, _ = oracle.ppo_loss_grad(cfg, 4, 2, params, obs, traj.action_i.reshape(-1), traj.logp.reshape(-1),
                                traj.adv.reshape(-1), traj.ret.reshape(-1))
    flat = torch.tensor(g.copy())
    rdist.allreduce_mean_(flat, group)
    gmean = flat.numpy().copy()
    gn = oracle.clip_by_global_norm(gmean, cfg.max_grad_norm)  # clip AFTER the all-reduce
    m, v = np.zeros_like(params), np.zeros_like(params)
    oracle.adam(params, gmean, m, v, cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps, 1)
    same = rdist.params_checksum_equal(torch.tensor(params), group)
    tmax = rdist.max_over_ranks(1.0 + rank, group, device="cpu")
    q.put((rank, traj.obs.copy(), traj.action_i.copy(), traj.logp.copy(), traj.adv.copy(), traj.ret.copy(), g, gmean,
           gn, params, same, tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_single_rank_double_batch():
    import oracle

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, obs0, a0, lp0, adv0, ret0, g0, gm0, gn0, p0, same0, t0), (_, obs1, a1, lp1, adv1, ret1, g1, gm1, gn1, p1, same1,
                                                                 t1) = res
    # replicas: identical reduced gradient, identical parameters after the step, checksum agreement
    assert np.array_equal(gm0, gm1) and np.array_equal(p0, p1) and same0 and same1 and gn0 == gn1
    assert t0 == t1 == 2.0  # max over ranks
    assert not np.array_equal(g0, g1)  # the shards really differ
    # single process with both shards (global env ids 0..127) reproduces the shards' rollouts exactly ...
    n_per, T, seed = 64, 8, 21
    cfg = oracle.ppo_default(hidden=32, n_microbatches=1, n_epochs=1)
    params = np.concatenate([oracle.mlp2_init(4, 32, 2, seed, 0), oracle.mlp2_init(4, 32, 1, seed, 1)])
    env = oracle.VecEnv("cartpole", 2 * n_per, seed=seed, env_id_base=0)
    traj = oracle.PPOTraj(0, 2 * n_per, T)
    oracle.ppo_rollout(env, T, cfg, params, traj, 0)
    oracle.ppo_gae(cfg, traj)
    assert np.array_equal(traj.obs, np.concatenate([obs0, obs1], axis=2))
    assert np.array_equal(traj.action_i, np.concatenate([a0, a1], axis=1))
    assert np.array_equal(traj.adv, np.concatenate([adv0, adv1], axis=1))
    # ... and its gradient over the 2x batch equals the mean of the two shard gradients
    obs = traj.obs[:T].transpose(1, 0, 2).reshape(4, T * 2 * n_per)
    g, _ = oracle.ppo_loss_grad(cfg, 4, 2, params, obs, traj.action_i.reshape(-1), traj.logp.reshape(-1),
                                traj.adv.reshape(-1), traj.ret.reshape(-1))
    np.testing.assert_allclose(gm0, _pre_clip(g, cfg.max_grad_norm), rtol=2e-4,
                               atol=1e-6)


def _pre_clip(g, clip):
    import oracle

    g = g.copy()
    oracle.clip_by_global_norm(g, clip)
    return g


def test_single_process_helpers():
    from rlhip import dist as rdist

    assert rdist.env_shard(3, 4096) == (3 * 4096, 4096)
    assert rdist.max_over_ranks(1.5) == 1.5
