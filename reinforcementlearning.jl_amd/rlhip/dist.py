"""Multi-GPU sharding of the hot path: one process per GPU, independent env shards, one all-reduce of
the flat gradient per optimiser step (RCCL over xGMI on the GPU box; `gloo` in CPU tests).

The reference has no counterpart (no Distributed / MPI / NCCL anywhere).  Design:
  * rollout is embarrassingly parallel: rank r owns env ids [r * n_per_rank, (r + 1) * n_per_rank) and
    the Philox streams keyed by those GLOBAL ids, so a trajectory does not depend on the number of GPUs;
  * parameters and Adam state are replicated; every rank initialises them from the same seed;
  * per optimiser step: grad kernel -> all_reduce(SUM) of the 13 KB flat gradient -> clip+Adam kernel
    with grad_scale = 1 / world (the clip sees the GLOBAL mean gradient = single-GPU semantics with a
    world-times larger micro-batch); replicas stay bit-identical because the reduced buffer is.
"""
import os

import torch


def env_shard(rank, n_per_rank):
    """(env_id_base, n) of a rank's shard."""
    return rank * n_per_rank, n_per_rank


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world, group or None)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, 1, None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world, dist.group.WORLD


def allreduce_mean_(flat_grad, group=None):
    """In-place mean of a flat gradient buffer over the group (sum all-reduce, then scale)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.mul_(1.0 / world)
    return flat_grad


def max_over_ranks(value, group=None, device=None):
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if torch.cuda.is_available() else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def params_checksum_equal(params, group=None):
    """Debug check that replicas are bit-identical: all-reduce MIN and MAX of an integer checksum."""
    import torch.distributed as dist

    c = params.view(torch.int32).to(torch.int64).sum().reshape(1)
    lo, hi = c.clone(), c.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool((lo == hi).all())


class P2PAllReduce:
    """One-shot peer-to-peer sum all-reduce of a small flat f32 vector inside one kernel on the compute stream
    (p2p.hip): every rank publishes its vector in an IPC-mapped uncached buffer and sums the world's buffers in rank
    order.  `create()` returns None -- and the caller keeps using the library all-reduce -- unless every rank maps
    every peer AND a self-test against torch.distributed's all-reduce passes on all ranks."""

    TIMEOUT_POLLS = 1 << 24       # steady state: tens of seconds of polling before a rank gives up (status flag)
    SELFTEST_TIMEOUT_POLLS = 1 << 20  # self-test: a path that does not work must fail within a few seconds

    def __init__(self):
        self.ok = False

    @staticmethod
    def _agree(flag, group, device):
        import torch.distributed as dist

        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    @classmethod
    def create(cls, group, cap, device):
        import ctypes as C

        import torch.distributed as dist

        from ._lib import call, lib

        self = cls()
        self.group, self.cap, self.device = group, int(cap), device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.seq = 0
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self._own, self._imported = C.c_void_p(), []
        handle = None
        try:
            call("rlhip_p2p_alloc", int(lib.rlhip_p2p_comm_bytes(self.cap)), C.byref(self._own))
            h = (C.c_uint8 * 64)()
            call("rlhip_p2p_export", self._own, h)
            handle = bytes(h)
        except Exception:  # noqa: BLE001 -- any failure means "use the library collective"
            handle = None
        gathered = [None] * self.world
        my_dev = torch.cuda.current_device()
        dist.all_gather_object(gathered, (handle, my_dev), group=group)
        good = all(g is not None and g[0] is not None for g in gathered)
        self.peers = (C.c_void_p * self.world)()
        if good:
            try:
                # no kernel touches a peer buffer before (1) the runtime says the devices can address each other and
                # (2) a host-driven 4-byte copy from the mapped flags succeeded and read the zero they were set to
                good = all(int(lib.rlhip_p2p_can_access(int(g[1]))) == 1 for g in gathered)
                flags_off = 2 * self.cap * 4
                for p in range(self.world):
                    if not good:
                        break
                    if p == self.rank:
                        self.peers[p] = self._own
                    else:
                        q = C.c_void_p()
                        call("rlhip_p2p_import", (C.c_uint8 * 64).from_buffer_copy(gathered[p][0]), C.byref(q))
                        self._imported.append(q)
                        self.peers[p] = q
                        val = C.c_uint32(0xFFFFFFFF)
                        call("rlhip_p2p_probe", q, flags_off, C.byref(val))
                        good = good and val.value == 0
            except Exception:  # noqa: BLE001
                good = False
        if not cls._agree(good, group, device):
            return None
        # self-test against the library all-reduce: same sums (up to summation order), bit-identical across ranks.
        # Every rank issues the SAME sequence of library collectives whatever happens locally (no early exit, local
        # failures only clear `passed`): a rank that bailed out alone would leave the others inside a collective.
        passed = True
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        for _ in range(3):
            x = torch.randn(min(self.cap, 4099), generator=g).to(device)
            ref = x.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=group)
            y = x.clone()
            try:
                self.all_reduce_(y, timeout_polls=cls.SELFTEST_TIMEOUT_POLLS)
                torch.cuda.synchronize()
                if int(self.status.item()) != 0 or not torch.allclose(y, ref, rtol=1e-5, atol=1e-5):
                    passed = False
            except Exception:  # noqa: BLE001
                passed = False
            lo, hi = y.clone(), y.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if not torch.equal(lo, hi):
                passed = False
        if not cls._agree(passed, group, device):
            return None
        self.ok = True
        return self

    def all_reduce_(self, t, timeout_polls=None):
        """in-place SUM over the ranks; t: contiguous f32 device tensor with numel <= cap"""
        from ._lib import call
        from .ops import ptr, stream_ptr

        self.seq += 1
        call("rlhip_p2p_allreduce_f32", ptr(t), t.numel(), self.cap, self.rank, self.world, self.peers, self.seq,
             timeout_polls or self.TIMEOUT_POLLS, ptr(self.status), stream_ptr())
        return t

    def close(self):
        """unmap the peers, then (after every rank has unmapped) free the own buffer; collective"""
        import torch.distributed as dist

        from ._lib import call

        torch.cuda.synchronize()
        for q in self._imported:
            try:
                call("rlhip_p2p_close", q)
            except Exception:  # noqa: BLE001
                pass
        self._imported = []
        try:
            dist.barrier(group=self.group)
        except Exception:  # noqa: BLE001
            pass
        if self._own:
            try:
                call("rlhip_p2p_free", self._own)
            except Exception:  # noqa: BLE001
                pass
            self._own = None
        self.ok = False

    def failed(self):
        """True if any all-reduce since the last check timed out (synchronises)."""
        return int(self.status.item()) != 0
