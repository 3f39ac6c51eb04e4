"""The reference's debug timer for the run loop, fed from device timings.

Reference: `const timer = TimerOutput()` (RLCore/src/ReinforcementLearningCore.jl:18); every call of `_run` is wrapped
in `@timeit_debug timer "<label>"` (RLCore/src/core/run.jl:46-72); `TimerOutputs.enable_debug_timings(RLCore)` switches
the sections on (docs/src/tips.md:23, test RLCore/test/core/base.jl:41-57).  Same labels here.  A section of a loop
whose body only ENQUEUES kernels measures nothing useful on the host clock, so every section is also bracketed by two
HIP events on the compute stream (rlhip_event_record, include/rlhip.h) and reports the device time between them;
events are resolved lazily, in batches, so that an enabled timer does not serialise the loop.  Disabled (the default)
a section is a shared no-op context manager."""
import contextlib
import ctypes as C
import time
from collections import OrderedDict

import torch

_NULL = contextlib.nullcontext()
_BATCH = 512  # event pairs kept in flight before one stream sync resolves them


class _Section:
    __slots__ = ("owner", "label", "t0", "e0")

    def __init__(self, owner, label):
        self.owner, self.label = owner, label

    def __enter__(self):
        o = self.owner
        self.e0 = o._record() if o._device else None
        self.t0 = time.perf_counter_ns()
        return self

    def __exit__(self, *exc):
        o = self.owner
        dt = time.perf_counter_ns() - self.t0
        rec = o.sections.setdefault(self.label, {"ncalls": 0, "host_ns": 0, "device_ms": 0.0})
        rec["ncalls"] += 1
        rec["host_ns"] += dt
        if self.e0 is not None:
            o._pending.append((self.label, self.e0, o._record()))
            if len(o._pending) >= _BATCH:
                o._resolve()
        return False


class TimerOutput:
    """TimerOutputs.TimerOutput stand-in: sections[label] = {ncalls, host_ns, device_ms}."""

    def __init__(self):
        self.enabled = False
        self.sections = OrderedDict()
        self._pending, self._pool = [], []
        self._device = False

    # -- TimerOutputs.enable_debug_timings(RLCore) / disable_debug_timings / reset_timer!
    def enable_debug_timings(self):
        self.enabled = True
        self._device = torch.cuda.is_available()

    def disable_debug_timings(self):
        self._resolve()
        self.enabled = False

    def reset_(self):
        self._resolve()
        self.sections.clear()

    def __call__(self, label):
        """`@timeit_debug timer label expr`  ->  `with timer(label): expr`"""
        return _Section(self, label) if self.enabled else _NULL

    # -- device side
    def _record(self):
        from ._lib import call
        from .ops import stream_ptr

        if self._pool:
            ev = self._pool.pop()
        else:
            ev = C.c_void_p()
            call("rlhip_event_create", C.byref(ev))
        call("rlhip_event_record", ev, stream_ptr())
        return ev

    def _resolve(self):
        if not self._pending:
            return
        from ._lib import call
        from .ops import stream_ptr

        call("rlhip_stream_sync", stream_ptr())
        ms = C.c_float()
        for label, e0, e1 in self._pending:
            call("rlhip_event_elapsed_ms", e0, e1, C.byref(ms))
            self.sections[label]["device_ms"] += float(ms.value)
            self._pool += [e0, e1]
        self._pending.clear()

    # -- reporting
    def todict(self):
        self._resolve()
        return {k: dict(v) for k, v in self.sections.items()}

    def __str__(self):
        d = self.todict()
        tot_h = sum(v["host_ns"] for v in d.values()) or 1
        tot_d = sum(v["device_ms"] for v in d.values()) or 1.0
        w = max([len("Section")] + [len(k) for k in d])
        lines = [f"{'Section'.ljust(w)}  ncalls   host time  %tot    avg    device time  %tot    avg",
                 "-" * (w + 70)]
        for k, v in d.items():
            h_ms, n = v["host_ns"] / 1e6, v["ncalls"]
            lines.append(f"{k.ljust(w)}  {n:6d}  {h_ms:8.2f}ms {100 * v['host_ns'] / tot_h:5.1f}% {1e3 * h_ms / n:6.1f}us"
                         f"  {v['device_ms']:9.2f}ms {100 * v['device_ms'] / tot_d:5.1f}% {1e3 * v['device_ms'] / n:6.1f}us")
        return "\n".join(lines)


timer = TimerOutput()  # `RLCore.timer`


def enable_debug_timings():
    timer.enable_debug_timings()


def disable_debug_timings():
    timer.disable_debug_timings()
