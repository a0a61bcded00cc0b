"""Data parallelism over RCCL/xGMI: one process per GPU, gradients averaged across ranks.

Semantics = DistributedDataParallel over the reference model: every rank computes
BatchNorm statistics and the global soft-dice on ITS OWN samples, gradients are averaged.  The
4.76 M fp32 gradients (19 MB) go out as a few flat buckets in reverse execution order; at this size
the all-reduce is latency-bound on xGMI, so few large buckets beat many small ones.
"""
from __future__ import annotations

from typing import Iterable, List

import os

import torch
import torch.distributed as dist


# The backbones mark three points of their backward at which a slice of the flat gradient buffer is final (decoder;
# the deepest stage; the rest).  The first slice has the whole encoder's backward left to hide behind, and every
# collective has a fixed cost of its own (launch, ring latency), so the first two slices go out as ONE all-reduce at the
# second mark.  (Measured in the one-rank rehearsal while RCCL's stream still shared the
# main stream's queue: three all-reduces +0.28 ms, two +0.21; with nccl_options two or three make no difference there,
# 4.606 / 4.603 ms.)  DEVIS_DP_BUCKETS=3 sends them separately.
_MERGE_FIRST_TWO = os.environ.get("DEVIS_DP_BUCKETS", "2") == "2"


class FlatGradSync:
    """Bucketed all-reduce overlapped with backward, for a backbone that writes its gradients into one flat
    buffer (devis_amd.unet3d.Unet).  The backbone calls ``begin(flat, bounds)`` at the start of its backward and
    ``bucket_ready(i)`` as soon as every gradient in ``flat[bounds[i][0]:bounds[i][1]]`` is final; the slice is
    all-reduced at once (RCCL runs it on its own stream, behind the kernels already queued) while the rest of
    backward keeps the compute stream busy.  ``finish()`` joins the streams and turns sums into means.
    The parameters' ``.grad`` are views of ``flat``, so there is no gather and no copy-back."""

    def __init__(self, world: int, group=None):
        self.world, self.group = world, group
        self.flat, self.bounds, self.works = None, None, []
        try:
            # RCCL averages in the collective; gloo cannot (SUM + one div_ there)
            self._avg = dist.get_backend(group) == "nccl"
        except Exception:
            self._avg = False

    def begin(self, flat: torch.Tensor, bounds) -> None:
        bounds = list(bounds)
        if _MERGE_FIRST_TWO and len(bounds) == 3:
            bounds = [(bounds[0][0], bounds[0][0]), (bounds[0][0], bounds[1][1]), bounds[2]]
        self.flat, self.bounds, self.works = flat, bounds, []

    def bucket_ready(self, i: int) -> None:
        lo, hi = self.bounds[i]
        if hi <= lo:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self.works.append(dist.all_reduce(self.flat[lo:hi], op=op, group=self.group, async_op=True))

    def finish(self) -> bool:
        """-> True if a flat buffer was reduced since the last call."""
        if self.flat is None:
            return False
        for w in self.works:
            w.wait()
        if not self._avg:
            self.flat.div_(self.world)
        self.flat, self.works = None, []
        return True


class GradientAllReducer:
    """Averages gradients across ranks.  With ``module`` (a TMSU or Unet whose backbone keeps a flat gradient
    buffer) the all-reduce is bucketed and overlapped with backward (FlatGradSync); parameters outside such a
    backbone, and any other model, go through flat buckets built after backward."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, group=None, module=None,
                 force: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force: run the collectives even in a one-rank group (rehearsal of the RCCL path on a one-GPU box)
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.flat_sync = None
        backbone = getattr(module, "backbone", module)
        if self.active and backbone is not None and hasattr(backbone, "_grad_sync"):
            self.flat_sync = FlatGradSync(self.world, group)
            backbone._grad_sync = self.flat_sync
            self._backbone = backbone
            own = {id(p) for p in backbone.parameters()}
            self.params = [p for p in self.params if id(p) not in own]
        # reverse order: decoder/head gradients are produced first in backward
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def all_reduce(self) -> None:
        """Average .grad across ranks (async launch per bucket, then copy back)."""
        if not self.active:
            return
        if self.flat_sync is not None:
            flat = self._backbone.last_flat_grad
            if self.flat_sync.finish() and flat is not None:
                # autograd normally adopts the views as .grad; if it cloned one, refresh the copy
                lo_ptr, hi_ptr = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
                for n, p in self._backbone.named_parameters():
                    if p.grad is not None and not (lo_ptr <= p.grad.data_ptr() < hi_ptr):
                        o, k, shape = self._backbone._grad_layout[n]
                        p.grad.copy_(flat[o:o + k].view(shape))
        works = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            flat.div_(self.world)
            works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat, bucket))
        for work, flat, bucket in works:
            work.wait()
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n


def nccl_options():
    """Options for ``dist.init_process_group("nccl", pg_options=...)`` of a data-parallel training process.
    torch launches RCCL's kernels on a stream of its own; taken from the normal-priority pool that stream shared the
    MAIN stream's hardware queue on MI355X (kernel trace: both on one queue id), and a hardware queue is in-order: a
    bucket's all-reduce, which has to wait for the side stream's weight gradients, held back every backward kernel the
    main stream had queued behind it (70-90 us of the main stream standing still per collective).  From the
    high-priority pool it gets a hardware queue of its own (a third queue id in the trace; main and side keep theirs) and
    waits there alone: one-rank rehearsal 4.713 -> 4.571 ms beside a plain step of 4.495.  DEVIS_NCCL_HIGH_PRIORITY=0:
    torch's default."""
    if os.environ.get("DEVIS_NCCL_HIGH_PRIORITY", "1") == "0":
        return None
    return dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


def shard_patches(n_patches: int, rank: int, world: int):
    """Patch indices this rank runs in sharded sliding-window inference (round-robin)."""
    return list(range(rank, n_patches, world))
