"""Launch plans: replay a training step's kernel launches without re-running the Python that issued them.

One step of the 3-D backbones is ~300 kernel launches issued from Python (shape checks, ``torch.empty``, ctypes
marshalling, event/stream calls): 4.5-6 ms of host time per step, more than the 4.8 ms the GPU needs for them, so the
step was bound by the host, and by however fast the host happened to be.  A plan records, during one ordinary (eager)
forward + backward, every launching call that goes through the C ABI together with the stream/event calls that order
the two streams, and afterwards replays that list: the same functions with the same arguments, in the same order, on
the same streams.  It is not a HIP graph (a captured two-stream step replays slower than eager here): the
launches are still individual ``hipLaunchKernel`` calls, only the Python between them is gone.

What makes the recorded arguments valid later:
  * every tensor the recorded step allocates comes from a ``torch.cuda.MemPool`` owned by the plan, so the addresses
    are never handed to anyone else; tensors the step frees go back to that pool and are reused only by the step's own
    later allocations, in an order the replay repeats exactly;
  * parameters, buffers and the flat gradient buffer are part of the plan's key (their addresses) or owned by it;
  * the step's inputs (x; the incoming gradient) are patched into the argument lists at each replay.
The outputs of a replayed step live in plan-owned memory: they are valid until the next forward of the same plan.
"""
from __future__ import annotations

import contextlib
import os
import weakref

import torch

from . import _lib

ENABLED = os.environ.get("DEVIS_LAUNCH_PLAN", "1") != "0"


class Recorder:
    def __init__(self):
        self.ops = []          # (callable, [args], is_lib_call)
        self.opaque = 0        # > 0: inside a host call that is noted as a whole


def current():
    return getattr(_lib._rec_tls, "rec", None)


def call(fn, *args):
    """Run a host-side action that belongs to the step (event record, stream wait, collective launch) and, when the
    thread is recording, note it as ONE replayable entry: launches made inside it are not noted a second time."""
    rec = getattr(_lib._rec_tls, "rec", None)
    if rec is None or rec.opaque:
        return fn(*args)
    rec.opaque += 1
    try:
        out = fn(*args)
    finally:
        rec.opaque -= 1
    rec.ops.append((fn, list(args), False))
    return out


class Segment:
    """A recorded list of calls + where the step's inputs appear among their arguments."""

    def __init__(self, ops, inputs):
        self.ops = ops
        self.patch = {}
        for name, ptr in inputs.items():
            sites = [(i, j) for i, (_, args, is_lib) in enumerate(ops) if is_lib
                     for j, a in enumerate(args) if type(a) is int and a == ptr]
            if not sites:
                raise _lib.DevisError("launch plan: input %r is not an argument of any recorded call" % name)
            self.patch[name] = sites

    def run(self, **ptrs):
        ops = self.ops
        for name, ptr in ptrs.items():
            for i, j in self.patch[name]:
                ops[i][1][j] = ptr
        for fn, args, is_lib in ops:
            rc = fn(*args)
            if is_lib and rc:
                _lib.check(rc, "replayed launch %s" % getattr(fn, "__name__", fn))


class StepPlan:
    """Forward + backward of one backbone at one shape.  States: new -> forward recorded -> ready (both recorded)."""

    def __init__(self, device):
        self.device = device
        self.pool = torch.cuda.MemPool()
        self.fwd = self.bwd = None
        self.gen = 0               # forwards run on this plan's memory so far
        self._pending = None       # weakref to the ctx whose backward has not run yet
        self.logits = None
        self.flat = self.dx = None
        self.keep = []

    # -- recording
    @contextlib.contextmanager
    def recording(self):
        rec = Recorder()
        _lib._rec_tls.rec = rec
        try:
            with torch.cuda.use_mem_pool(self.pool, self.device):
                yield rec
        finally:
            _lib._rec_tls.rec = None

    def ready(self) -> bool:
        return self.fwd is not None and self.bwd is not None

    def busy(self) -> bool:
        """True while a forward on this plan's memory still waits for its backward."""
        return self._pending is not None and self._pending() is not None

    def begin_step(self, ctx) -> int:
        self.gen += 1
        self._pending = weakref.ref(ctx)
        return self.gen

    def end_step(self) -> None:
        self._pending = None
