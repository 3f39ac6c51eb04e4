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


class HipComm:
    """The sharded learner's collective, owned by librlhip.so (csrc/comm.hip, `rlhip_comm_*`): the one-shot
    peer-to-peer kernel over IPC-mapped buffers once every rank validated it, RCCL's ncclAllReduce on the compute
    stream otherwise -- both behind `rlhip_allreduce_grads`.  torch.distributed is ONLY the byte transport of the
    set-up (a 128-byte RCCL id from rank 0, one 64-byte IPC handle + device id per rank) -- what a Julia host would
    do with any transport it has.

    use_rccl: None = yes when the group's backend is nccl and RLHIP_COMM_NO_RCCL is unset (RCCL refuses two ranks on
    one device: the one-GPU multi-process tests run the peer-to-peer path alone, over a gloo group).
    RLHIP_NO_P2P=1 skips the peer-to-peer set-up (RCCL only); RLHIP_REQUIRE_P2P=1 makes a failed validation fatal."""

    def __init__(self):
        self.h = None

    @classmethod
    def create(cls, group, cap, device, use_rccl=None):
        import ctypes as C
        import sys

        import torch.distributed as dist

        from . import _lib
        from ._lib import call

        self = cls()
        self.group, self.cap, self.device = group, int(cap), device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if use_rccl is None:
            use_rccl = dist.get_backend(group) == "nccl" and os.environ.get("RLHIP_COMM_NO_RCCL", "0") != "1"
        uid = [None]
        if use_rccl:
            if self.rank == 0:
                b = (C.c_uint8 * 128)()
                call("rlhip_comm_unique_id", b)
                uid = [bytes(b)]
            dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = C.c_void_p()
        call("rlhip_comm_init", self.rank, self.world, (C.c_uint8 * 128).from_buffer_copy(uid[0]) if uid[0] else None,
             self.cap, C.byref(h))
        self.h = h
        # From here on a failure on ONE rank must not strand the others in a collective (round 5): an error
        # of this rank's export / set-up is caught, reported to every rank through the set-up transport (the object gathers
        # below), and answered everywhere by "peer-to-peer not active" -- the run continues over ncclAllReduce (or, without an
        # RCCL communicator, torch.distributed) and says so in `transport()`.
        self.setup_error = None
        hb, dev_id = (C.c_uint8 * 64)(), _lib.i32(0)
        try:
            call("rlhip_comm_export", h, hb, C.byref(dev_id))
            if os.environ.get("RLHIP_TEST_FAIL_EXPORT_RANK", "") == str(self.rank):  # test hook: tests/test_gpu_run.py
                raise _lib.RLHipError("injected rlhip_comm_export failure (RLHIP_TEST_FAIL_EXPORT_RANK)")
            mine = (bytes(hb), int(dev_id.value), None)
        except _lib.RLHipError as exc:
            self.setup_error = f"rank {self.rank}: {exc}"
            mine = (None, -1, self.setup_error)
        gathered = [None] * self.world
        dist.all_gather_object(gathered, mine, group=group)
        export_errors = [g[2] for g in gathered if g[2]]
        ran_setup = False
        if os.environ.get("RLHIP_NO_P2P", "0") != "1" and self.world > 1 and not export_errors:
            handles = (C.c_uint8 * (64 * self.world)).from_buffer_copy(b"".join(g[0] for g in gathered))
            devices = (_lib.i32 * self.world)(*[g[1] for g in gathered])
            active = _lib.i32(0)
            ran_setup = True
            try:
                call("rlhip_p2p_setup", h, handles, devices, C.byref(active))
            except _lib.RLHipError as exc:  # (HIP errors inside the set-up are votes, not returns: this is the RCCL agreement itself failing)
                self.setup_error = f"rank {self.rank}: {exc}"
        elif export_errors and self.world > 1:
            call("rlhip_comm_disable_p2p", h, ("a rank could not export its exchange buffer: " + export_errors[0])[:250].encode())
        if ran_setup:
            # the verdict of a clean set-up is already agreed inside the library; a rank whose call RAISED took no part in that
            # agreement, so agree once more over the set-up transport and switch the path off wherever it had validated
            verdicts = [None] * self.world
            dist.all_gather_object(verdicts, self.setup_error, group=group)
            errs = [v for v in verdicts if v]
            if errs:
                call("rlhip_comm_disable_p2p", h, ("set-up failed with an error on another rank: " + errs[0])[:250].encode())
        d = self.info()
        if self.world > 1 and not d.p2p_active:
            msg = (f"[rlhip] rank {self.rank}: peer-to-peer gradient exchange NOT active: {d.why.decode()} -> "
                   f"{'RCCL ncclAllReduce' if d.rccl_active else 'torch.distributed all_reduce'} per optimiser step")
            if os.environ.get("RLHIP_REQUIRE_P2P", "0") == "1":
                raise RuntimeError(msg)
            if self.rank == 0:
                print(msg, file=sys.stderr, flush=True)
        return self

    def info(self):
        import ctypes as C

        from . import _lib
        from ._lib import call

        d = _lib.CommDesc()
        call("rlhip_comm_info", self.h, C.byref(d))
        return d

    @property
    def p2p_active(self):
        return bool(self.info().p2p_active)

    @property
    def rccl_active(self):
        return bool(self.info().rccl_active)

    @property
    def ok(self):
        """a transport behind rlhip_allreduce_grads exists"""
        d = self.info()
        return self.world == 1 or bool(d.p2p_active or d.rccl_active)

    def transport(self):
        d = self.info()
        if self.world == 1:
            return "none (single rank)"
        if d.p2p_active:
            return "p2p one-shot kernel over IPC-mapped peer buffers (csrc/p2p.hip), validated on every rank at start-up" + \
                   ("; RCCL communicator behind the same ABI as fallback" if d.rccl_active else "")
        if d.rccl_active:
            return f"RCCL ncclAllReduce on the compute stream via rlhip_allreduce_grads (p2p not active: {d.why.decode()})"
        return f"torch.distributed all_reduce (no transport behind the ABI: {d.why.decode()})"

    def set_timeout(self, polls):
        from ._lib import call

        call("rlhip_comm_set_timeout", self.h, int(polls))

    def all_reduce_(self, t):
        """in-place SUM over the ranks on the current stream; t: contiguous f32 device tensor"""
        from ._lib import call
        from .ops import ptr, stream_ptr

        call("rlhip_allreduce_grads", self.h, ptr(t), t.numel(), stream_ptr())
        return t

    def check(self):
        """raises RLHipTimeoutError if a peer never arrived at an exchange (reads a host-pinned word: no sync)"""
        from ._lib import call

        call("rlhip_comm_check", self.h)

    def failed(self):
        """True if any exchange since the start timed out (synchronises first)."""
        from ._lib import RLHipTimeoutError

        torch.cuda.synchronize()
        try:
            self.check()
        except RLHipTimeoutError:
            return True
        return False

    def close(self):
        """collective: barrier -> every rank unmaps its peers -> barrier -> every rank frees its own (IPC-exported)
        buffer: no rank frees a buffer a slower rank still has mapped"""
        import torch.distributed as dist

        from ._lib import call

        if self.h is None:
            return
        torch.cuda.synchronize()

        def barrier():
            try:
                dist.barrier(group=self.group)
            except Exception:  # noqa: BLE001
                pass

        barrier()
        call("rlhip_comm_unmap", self.h)
        barrier()
        call("rlhip_comm_destroy", self.h)
        self.h = None


class P2PAllReduce:
    """Round-1 name of the validated peer-to-peer exchange: `create()` returns a HipComm whose peer-to-peer path is
    active, or None (then the caller uses the library all-reduce)."""

    @classmethod
    def create(cls, group, cap, device):
        comm = HipComm.create(group, cap, device)
        if comm.p2p_active:
            return comm
        comm.close()
        return None
